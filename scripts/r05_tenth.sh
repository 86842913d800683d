#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 400 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_delivers" > $OUT/r05j_pytest_owner.log 2>&1; echo "owner group rc=$?"; tail -4 $OUT/r05j_pytest_owner.log
RANKS=4 STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py > $OUT/r05j_rate_4.log 2>&1; tail -1 $OUT/r05j_rate_4.log | cut -c1-400
for e in 2 3 4; do FWGPU_STREAM_CONSUMER_EIGHTHS=$e RANKS=1 STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330 | sed "s/^/1 rank, consumer eighths $e: /"; done | tee $OUT/r05j_owner_stream_rate.txt
for e in 2 4; do FWGPU_STREAM_CONSUMER_EIGHTHS=$e RANKS=4 STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330 | sed "s/^/4 ranks, consumer eighths $e: /"; done | tee -a $OUT/r05j_owner_stream_rate.txt
timeout 300 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_learns" -s > $OUT/r05j_pytest_owner_stat.log 2>&1; echo "owner stat rc=$?"; grep "owner-side apply, streaming" $OUT/r05j_pytest_owner_stat.log | sort -u
