# pall (plain loads and stores in the v2 kernel) against the shipped library: the GPU suite on it, then 150-step runs (9.8 M examples) and the PMC traffic.
V=$PWD/build/variants
FWGPU_LIBRARY=$V/libfwgpu_pall.so python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | head -20
for rep in 1 2; do for L in "" $V/libfwgpu_pall.so; do
  FWGPU_LIBRARY=$L timeout 400 python3 bench.py --steps 150 --warmup 4 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('150 steps', '$(basename ${L:-shipped})', round(d['value']), round(d['final_logloss'],4), {k:round(v,4) for k,v in d['logloss_after_examples'].items()})"
done; done
for L in "" $V/libfwgpu_pall.so; do
  FWGPU_LIBRARY=$L timeout 600 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('traffic', '$(basename ${L:-shipped})', round(d['value']), round(d['final_logloss'],4), round(r['frac'],4), r.get('traffic'), r.get('traffic_detail') or '')"
done
