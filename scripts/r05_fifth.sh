#!/bin/bash
# Round 5, fifth GPU-box call: streaming owner-side apply (consumers inside the example kernel), store policy 3, oversize + head, E predict kernel trace
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "oversize or chunked" > $OUT/r05e_pytest_oversize.log 2>&1; echo "oversize rc=$?"; tail -3 $OUT/r05e_pytest_oversize.log
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "owner" > $OUT/r05e_pytest_owner.log 2>&1; echo "owner rc=$?"; tail -5 $OUT/r05e_pytest_owner.log
timeout 600 python3 -m pytest tests/test_gpu_dist_procs.py -x -q -m gpu -k "streaming or timeout" > $OUT/r05e_pytest_owner_procs.log 2>&1; echo "owner procs rc=$?"; tail -5 $OUT/r05e_pytest_owner_procs.log
for n in 1 2 4; do RANKS=$n STEPS=8 timeout 300 python3 scripts/owner_stream_rate.py 2>&1 | tail -1; done | tee $OUT/r05e_owner_stream_rate.txt
RANKS=4 STEPS=8 CWG=64 timeout 300 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | tee -a $OUT/r05e_owner_stream_rate.txt
# store policy 3 (thinned accumulator stores on hot rows): rate on the driver shape, then the long protocol
for th in 2 4 16; do for lg in 2 3; do
  FWGPU_ACC_HOT_THETA=$th FWGPU_ACC_SAMPLE_LOG2=$lg timeout 300 python3 bench.py --store-policy 3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-config-e --no-config-b 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy 3 theta $th m 2^$lg:', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_logloss'],4), 'oracle', d['oracle_final_logloss'])"
done; done | tee $OUT/r05e_policy3.txt
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-config-e --no-config-b 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy 1 (shipped):', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_logloss'],4))" | tee -a $OUT/r05e_policy3.txt
FWGPU_ACC_HOT_THETA=4 FWGPU_ACC_SAMPLE_LOG2=3 timeout 600 python3 bench.py --long --long-passes 2 --store-policy 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['logloss_after_examples']
print('long, policy 3 theta 4 m 8:', round(d['value']), 'final', [round(x,4) for x in d['final_logloss_passes']], 'curve', {k: round(v[0],4) for k,v in c.items() if int(k) % 4194304 == 0})" | tee -a $OUT/r05e_policy3.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r05e_epred_trace -o t -- python3 $R/scripts/e_predict_rate.py > $OUT/r05e_epred_trace.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/r05e_epred_trace -name "*.db") > $OUT/r05e_epred_kernels.txt 2>&1; head -12 $OUT/r05e_epred_kernels.txt
