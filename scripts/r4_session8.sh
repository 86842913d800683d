#!/bin/bash
OUT=gpurun_out/r4h; mkdir -p $OUT
run() { # name steps env...
  local name=$1; local steps=$2; shift; shift
  env "$@" timeout 900 python3 bench.py --steps $steps --warmup 5 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d.get('logloss_after_examples',{}).items()})"
}
V=$PWD/build/variants
for pass in 1 2 3; do
  run "maxr14 (HEAD)" 20
  run "maxr16       " 20 FWGPU_LIBRARY=$V/libfwgpu_m16.so
  run "maxr18       " 20 FWGPU_LIBRARY=$V/libfwgpu_m18.so
  run "maxr18 ug2   " 20 FWGPU_LIBRARY=$V/libfwgpu_m18ug2.so
done 2>&1 | tee $OUT/maxr_ab.txt
run "L maxr14" 150 2>&1 | tee $OUT/maxr_long.txt
run "L maxr16" 150 FWGPU_LIBRARY=$V/libfwgpu_m16.so 2>&1 | tee -a $OUT/maxr_long.txt
run "L maxr18" 150 FWGPU_LIBRARY=$V/libfwgpu_m18.so 2>&1 | tee -a $OUT/maxr_long.txt
