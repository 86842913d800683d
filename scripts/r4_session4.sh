#!/bin/bash
# Round 4, fourth GPU session: the whole GPU suite on the slimmed kernels (opaque thread index / per-example LDS binding), the spread the
# statistical tolerances rest on, and bench regression checks.
OUT=gpurun_out/r4d; mkdir -p $OUT
echo "== gpu suite"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/gputest.txt
echo "== spread"; timeout 1500 python scripts/holdout_spread.py 8 2>&1 | grep -v amdgpu.ids | tee $OUT/holdout_spread.txt
run() { # name steps env...
  local name=$1; local steps=$2; shift; shift
  env "$@" timeout 900 python3 bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['final_logloss'],4), d.get('oracle_final_logloss'), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
for pass in 1 2 3; do
  run "default         " 20
  run "no prefetch     " 20 FWGPU_PREFETCH=0
  run "flush 0         " 20 FWGPU_WB_FLUSH_EVERY=0
done 2>&1 | tee $OUT/bench_ab.txt
