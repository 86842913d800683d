# Write-back stores for ONE of the two tables only: pw = the weights (-DFW_PLAIN_STORES_W=1), pacc = the accumulators (-DFW_PLAIN_STORES_ACC=1); loads device-scope.
V=$PWD/build/variants
for L in $V/libfwgpu_pw.so $V/libfwgpu_pacc.so; do echo "== GPU suite on $(basename $L)"; FWGPU_LIBRARY=$L python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | head -5; done
for rep in 1 2; do for L in "" $V/libfwgpu_pw.so $V/libfwgpu_pacc.so; do
  FWGPU_LIBRARY=$L timeout 400 python3 bench.py --steps 150 --warmup 4 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('150 steps', '$(basename ${L:-shipped})', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d['logloss_after_examples'].items()})"
done; done
