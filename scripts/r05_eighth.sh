#!/bin/bash
# Round 5, eighth GPU-box call (every leg on a short leash): streaming owner-side apply after the LR block bound, E predict rate, policy 3 with m = 4, E in flight
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 400 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_delivers" > $OUT/r05h_pytest_owner.log 2>&1; echo "owner group rc=$?"; tail -4 $OUT/r05h_pytest_owner.log
timeout 300 python3 -m pytest tests/test_gpu_dist_procs.py -x -q -m gpu -k "streaming" > $OUT/r05h_pytest_owner_procs.log 2>&1; echo "owner procs rc=$?"; tail -4 $OUT/r05h_pytest_owner_procs.log
for n in 1 2 4; do RANKS=$n STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330; done | tee $OUT/r05h_owner_stream_rate.txt
for e in 3 6; do FWGPU_STREAM_CONSUMER_EIGHTHS=$e RANKS=2 STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330 | sed "s/^/consumer eighths $e: /"; done | tee -a $OUT/r05h_owner_stream_rate.txt
COMPARE=1 timeout 200 python3 scripts/e_predict_rate.py 2>&1 | tail -2 | tee $OUT/r05h_e_predict.txt
K=8 NN=0 timeout 200 python3 scripts/e_predict_rate.py 2>&1 | tail -1 | sed 's/^/config C predict-only: /' | tee -a $OUT/r05h_e_predict.txt
FWGPU_ACC_HOT_THETA=2 FWGPU_ACC_SAMPLE_LOG2=2 timeout 400 python3 -m pytest tests/test_gpu_conservation.py -x -q -m gpu -s -k "adagrad and 28" > $OUT/r05h_conservation_m4.log 2>&1; grep -A8 "surviving fraction" $OUT/r05h_conservation_m4.log | head -12
FWGPU_ACC_HOT_THETA=2 FWGPU_ACC_SAMPLE_LOG2=2 timeout 400 python3 bench.py --long --long-passes 2 --store-policy 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['logloss_after_examples']
print('long, policy 3 theta 2 m 4:', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'final', [round(x,4) for x in d['final_logloss_passes']], 'curve', {k: round(v[0],4) for k,v in c.items() if int(k) % 4194304 == 0})" | tee $OUT/r05h_policy3_long.txt
for fl in 128 192; do timeout 300 python3 bench.py --k 16 --nn-layers 2 --batch 8192 --steps 24 --warmup 4 --holdout 65536 --curve-every 1000 --max-in-flight $fl --no-cpu-baseline --no-traffic --no-config-e --no-config-b 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config E, max in flight $fl:', round(d['value']), round(d['final_logloss'],4), d.get('oracle_final_logloss'))"; done | tee $OUT/r05h_configE_in_flight.txt
