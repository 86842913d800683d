#!/bin/bash
# Round 4, second GPU session: conservation table with the hot-row policies, the whole GPU suite on the new build, A/B of policies 1 / 2 / 3 / 4 with the
# write-back bound, table sizes of the recent-row filter, and 150-step loss curves of the candidates.
OUT=gpurun_out/r4b; mkdir -p $OUT
echo "== conservation"; timeout 1800 python -m pytest tests/test_gpu_conservation.py -q -s -p no:cacheprovider 2>&1 | tee $OUT/conservation.txt | grep -v "^E \|^    \|^$" | tail -80
echo "== gpu suite"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_conservation.py 2>&1 | tail -8 | tee $OUT/gputest.txt
run() { # name steps env...
  local name=$1; local steps=$2; shift; shift
  env "$@" timeout 900 python3 bench.py --steps $steps --warmup 5 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d.get('logloss_after_examples',{}).items()})"
}
V=$PWD/build/variants
for pass in 1 2; do
  run "p1_f0       " 20 FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=0
  run "p1_f64      " 20 FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=64
  run "p2_f64      " 20 FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=64
  run "p3_f0       " 20 FWGPU_STORE_POLICY=3 FWGPU_WB_FLUSH_EVERY=0
  run "p3_f64      " 20 FWGPU_STORE_POLICY=3 FWGPU_WB_FLUSH_EVERY=64
  run "p4_f64      " 20 FWGPU_STORE_POLICY=4 FWGPU_WB_FLUSH_EVERY=64
  run "p3_f64_r1024" 20 FWGPU_STORE_POLICY=3 FWGPU_WB_FLUSH_EVERY=64 FWGPU_LIBRARY=$V/libfwgpu_r1024.so
  run "p3_f64_r4096" 20 FWGPU_STORE_POLICY=3 FWGPU_WB_FLUSH_EVERY=64 FWGPU_LIBRARY=$V/libfwgpu_r4096.so
  run "p4_f64_r4096" 20 FWGPU_STORE_POLICY=4 FWGPU_WB_FLUSH_EVERY=64 FWGPU_LIBRARY=$V/libfwgpu_r4096.so
done 2>&1 | tee $OUT/policy_ab.txt
for pass in 1; do
  run "L p1_f0     " 150 FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=0
  run "L p2_f64    " 150 FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=64
  run "L p3_f64    " 150 FWGPU_STORE_POLICY=3 FWGPU_WB_FLUSH_EVERY=64
  run "L p4_f64    " 150 FWGPU_STORE_POLICY=4 FWGPU_WB_FLUSH_EVERY=64
  run "L p1_f64    " 150 FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=64
done 2>&1 | tee $OUT/policy_long.txt
