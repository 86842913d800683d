#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 200 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_delivers and 4-8-9" > $OUT/r05k_pytest_owner.log 2>&1; echo "owner group 4 ranks rc=$?"; tail -4 $OUT/r05k_pytest_owner.log
timeout 200 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_learns and 4" -s > $OUT/r05k_pytest_owner_stat.log 2>&1; echo "owner stat 4 ranks rc=$?"; grep "owner-side apply, streaming" $OUT/r05k_pytest_owner_stat.log | sort -u
FWGPU_STREAM_CONSUMER_EIGHTHS=4 RANKS=4 STEPS=8 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330 | tee $OUT/r05k_owner_stream_rate.txt
