# On top of the shipped write-back weight stores: the gather's reads of w through L2 as well (-DFW_PLAIN_GATHER=1; an XCD then reads its own latest rows from its L2).
V=$PWD/build/variants
FWGPU_LIBRARY=$V/libfwgpu_pwg.so python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | head -5
for rep in 1 2; do for L in "" $V/libfwgpu_pwg.so; do
  FWGPU_LIBRARY=$L timeout 400 python3 bench.py --steps 150 --warmup 4 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('150 steps', '$(basename ${L:-shipped})', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d['logloss_after_examples'].items()})"
done; done
