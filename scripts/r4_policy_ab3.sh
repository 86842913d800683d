#!/bin/bash
# Round 4, third GPU session: l2probe S5/S6, the nt (streaming) flavour of the write-back stores: conservation + bench.
OUT=gpurun_out/r4c; mkdir -p $OUT
echo "== l2probe"; timeout 120 tools/l2probe 2>&1 | head -12 | tee $OUT/l2probe.txt
V=$PWD/build/variants
echo "== conservation, default build"; timeout 900 python -m pytest tests/test_gpu_conservation.py -q -s -p no:cacheprovider 2>&1 | tee $OUT/conservation_default.txt | grep -v "^E \|^    \|^$" | tail -50
echo "== conservation, acc stores nt (policy 2 column = w plain, acc nt)"; FWGPU_LIBRARY=$V/libfwgpu_nta.so timeout 900 python -m pytest tests/test_gpu_conservation.py -q -s -p no:cacheprovider 2>&1 | tee $OUT/conservation_nta.txt | grep -v "^E \|^    \|^$" | tail -40
run() { # name steps env...
  local name=$1; local steps=$2; shift; shift
  env "$@" timeout 900 python3 bench.py --steps $steps --warmup 5 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d.get('logloss_after_examples',{}).items()})"
}
for pass in 1 2; do
  run "default(p1_f128)" 20
  run "p1_f0           " 20 FWGPU_WB_FLUSH_EVERY=0
  run "p2_f128         " 20 FWGPU_STORE_POLICY=2
  run "p2_f128 acc nt  " 20 FWGPU_STORE_POLICY=2 FWGPU_LIBRARY=$V/libfwgpu_nta.so
  run "p2_f0   acc nt  " 20 FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=0 FWGPU_LIBRARY=$V/libfwgpu_nta.so
  run "p2_f128 w+acc nt" 20 FWGPU_STORE_POLICY=2 FWGPU_LIBRARY=$V/libfwgpu_ntwa.so
  run "p1_f128 w nt    " 20 FWGPU_STORE_POLICY=1 FWGPU_LIBRARY=$V/libfwgpu_ntwa.so
done 2>&1 | tee $OUT/policy_ab.txt
run "L p2_f128 acc nt  " 150 FWGPU_STORE_POLICY=2 FWGPU_LIBRARY=$V/libfwgpu_nta.so 2>&1 | tee $OUT/policy_long.txt
run "L p2_f128 w+acc nt" 150 FWGPU_STORE_POLICY=2 FWGPU_LIBRARY=$V/libfwgpu_ntwa.so 2>&1 | tee -a $OUT/policy_long.txt
run "L default         " 150 2>&1 | tee -a $OUT/policy_long.txt
