#!/bin/bash
# Round 5, third GPU-box call: batched predict-only head, config E bisect of this round's generic-kernel changes, long-protocol A/Bs of the concurrency knobs
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config_e or deep_head or head or oversize or chunked" > $OUT/r05c_pytest_head.log 2>&1; echo "head tests rc=$?"; tail -3 $OUT/r05c_pytest_head.log
COMPARE=1 timeout 300 python3 scripts/e_predict_rate.py 2>&1 | tail -2 | tee $OUT/r05c_e_predict.txt
FWGPU_HEAD_PREDICT_PER_EXAMPLE=1 timeout 300 python3 scripts/e_predict_rate.py 2>&1 | tail -1 | tee -a $OUT/r05c_e_predict.txt
FWGPU_HEAD_GEMM_SPLITK_MAX_M=1000000 timeout 300 python3 scripts/e_predict_rate.py 2>&1 | tail -1 | sed 's/^/split-K GEMMs for every slab: /' | tee -a $OUT/r05c_e_predict.txt
PASSES=2 bash scripts/ab_bench.sh "--k 16 --nn-layers 2 --batch 8192 --steps 24 --warmup 4 --holdout 65536 --no-config-e --no-config-b" default e_cg1 e_fju4 e_cg1_fju4 2>&1 | tee $OUT/r05c_configE_bisect.txt
for flags in "--max-in-flight 256" "--max-in-flight 128" "--store-policy 0" "--store-policy 2"; do
  timeout 600 python3 bench.py --long --long-passes 2 $flags 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['logloss_after_examples']
print('$flags:', round(d['value']), 'final', [round(x,4) for x in d['final_logloss_passes']], 'curve', {k: round(v[0],4) for k,v in c.items() if int(k) % 4194304 == 0})"
done | tee $OUT/r05c_long_ab.txt
# phase shares of the headless k = 16 kernel (two-chunk rows) and of config E, -DFW_TICKS build
FWGPU_LIBRARY=$R/build/variants/libfwgpu_ticks.so K=16 QUICK=1 timeout 300 python3 scripts/perf_probe.py 2>&1 | grep -E "threads=" | sed 's/^/k16 headless: /' | tee $OUT/r05c_phase_ticks.txt
FWGPU_LIBRARY=$R/build/variants/libfwgpu_ticks.so K=16 NN_LAYERS=2 THREADS=0 QUICK=1 B=8192 timeout 300 python3 scripts/perf_probe.py 2>&1 | grep -E "threads=" | sed 's/^/config E: /' | tee -a $OUT/r05c_phase_ticks.txt
