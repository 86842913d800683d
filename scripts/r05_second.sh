#!/bin/bash
# Round 5, second GPU-box call: parity after the generic kernel's update / head changes, the new launcher + timeout tests, config E rate,
# headless k = 16 with more overflow rows in flight, and the 16 Mi-example protocol (3 passes).
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_procs.py tests/test_gpu_dist.py -x -q -m gpu > $OUT/r05b_pytest_parity.log 2>&1; echo "parity+dist rc=$?"; tail -3 $OUT/r05b_pytest_parity.log
timeout 1500 python3 -m pytest tests/test_gpu_scale_launch.py -x -q -m gpu > $OUT/r05b_pytest_launch.log 2>&1; echo "launch rc=$?"; tail -3 $OUT/r05b_pytest_launch.log
for i in 1 2; do
timeout 600 python3 bench.py --k 16 --nn-layers 2 --batch 8192 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b > $OUT/r05b_configE_$i.json 2> $OUT/r05b_configE_$i.err
python3 -c "
import json;d=json.loads([l for l in open('$OUT/r05b_configE_$i.json') if l.startswith('{')][-1]);print('config E:', round(d['value']), d['final_logloss'], d.get('oracle_final_logloss'), d['roofline']['frac'])"
done
# headless k = 16 (two-chunk rows of the v2 kernel): overflow rows in flight per wave in the update
PASSES=2 bash scripts/ab_bench.sh "--k 16 --batch 16384 --steps 16 --warmup 4 --holdout 65536 --no-config-e --no-config-b" default uo2 uo3 uo4 2>&1 | tee $OUT/r05b_k16_uo_ab.txt
timeout 900 python3 bench.py --long > $OUT/r05b_long.json 2> $OUT/r05b_long.err; tail -c 1500 $OUT/r05b_long.json
# rate vs skew x store policy (what the hot rows cost under each policy)
for z in 0 1.05 1.3; do for pol in 0 1 2; do
  timeout 300 python3 bench.py --zipf $z --store-policy $pol --steps 12 --warmup 3 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zipf $z policy $pol:', round(d['value']), 'launch ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_logloss'],4))"
done; done | tee $OUT/r05b_skew_policy.txt
