#!/bin/bash
# Round 5, sixth GPU-box call: streaming owner-side apply again (closed-form LR check, two positions per consumer round, consumer share of the grid),
# store policy 3 on the conservation rig and on the long protocol with other thresholds
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "owner" > $OUT/r05f_pytest_owner.log 2>&1; echo "owner rc=$?"; tail -12 $OUT/r05f_pytest_owner.log
timeout 600 python3 -m pytest tests/test_gpu_dist_procs.py -x -q -m gpu -k "streaming" > $OUT/r05f_pytest_owner_procs.log 2>&1; echo "owner procs rc=$?"; tail -8 $OUT/r05f_pytest_owner_procs.log
for e in 3 5 6; do for n in 1 4; do FWGPU_STREAM_CONSUMER_EIGHTHS=$e RANKS=$n STEPS=6 timeout 300 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | sed "s/^/consumer eighths $e: /"; done; done | tee $OUT/r05f_owner_stream_rate.txt
timeout 900 python3 -m pytest tests/test_gpu_conservation.py -x -q -m gpu -s > $OUT/r05f_conservation.log 2>&1; echo "conservation rc=$?"; grep -A8 "surviving fraction" $OUT/r05f_conservation.log | head -60
for cfg in "2 3" "1.5 3" "2 4" "4 4"; do set -- $cfg
  FWGPU_ACC_HOT_THETA=$1 FWGPU_ACC_SAMPLE_LOG2=$2 timeout 600 python3 bench.py --long --long-passes 2 --store-policy 3 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['logloss_after_examples']
print('long, policy 3 theta $1 m 2^$2:', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'final', [round(x,4) for x in d['final_logloss_passes']], 'curve', {k: round(v[0],4) for k,v in c.items() if int(k) % 4194304 == 0})"
done | tee $OUT/r05f_policy3_long.txt
