#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 300 python3 -m pytest tests/test_gpu_dist.py -q -m gpu -k "streaming" -rs > $OUT/r05l_pytest_owner.log 2>&1; echo "owner group rc=$?"; tail -8 $OUT/r05l_pytest_owner.log | cut -c1-300
timeout 300 python3 -m pytest tests/test_gpu_dist_procs.py -q -m gpu -k "streaming" > $OUT/r05l_pytest_owner_procs.log 2>&1; echo "owner procs rc=$?"; tail -3 $OUT/r05l_pytest_owner_procs.log
for n in 1 2 4; do FWGPU_STREAM_CONSUMER_EIGHTHS=4 RANKS=$n STEPS=8 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330; done | tee $OUT/r05l_owner_stream_rate.txt
