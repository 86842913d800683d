#!/bin/bash
# Round 4, first GPU session: L2 semantics probe, the FFM-row conservation table, parity sanity, and an interleaved A/B of the
# store policies / write-back intervals / record prefetch through bench.py (env switches, one process per run).
OUT=gpurun_out/r4a; mkdir -p $OUT
echo "== l2probe"; timeout 120 tools/l2probe 2>&1 | tee $OUT/l2probe.txt
echo "== parity sanity"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/parity.txt
echo "== conservation"; timeout 1500 python -m pytest tests/test_gpu_conservation.py -x -q -s -p no:cacheprovider 2>&1 | tee $OUT/conservation.txt | tail -60
run() { # name env...
  local name=$1; shift
  env "$@" timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
for pass in 1 2; do
  run "p1_pf0      " FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=0 FWGPU_PREFETCH=0
  run "p1_pf1      " FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=0
  run "p0_pf1      " FWGPU_STORE_POLICY=0
  run "p2_f0       " FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=0
  run "p2_f1       " FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=1
  run "p2_f4       " FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=4
  run "p2_f16      " FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=16
  run "p2_f64      " FWGPU_STORE_POLICY=2 FWGPU_WB_FLUSH_EVERY=64
  run "p1_f4       " FWGPU_STORE_POLICY=1 FWGPU_WB_FLUSH_EVERY=4
done 2>&1 | tee $OUT/policy_ab.txt
