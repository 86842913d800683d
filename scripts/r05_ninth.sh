#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 400 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_delivers" > $OUT/r05i_pytest_owner.log 2>&1; echo "owner group rc=$?"; tail -4 $OUT/r05i_pytest_owner.log
timeout 300 python3 -m pytest tests/test_gpu_dist_procs.py -x -q -m gpu -k "streaming" > $OUT/r05i_pytest_owner_procs.log 2>&1; echo "owner procs rc=$?"; tail -4 $OUT/r05i_pytest_owner_procs.log
for n in 1 2 4; do RANKS=$n STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py > $OUT/r05i_rate_$n.log 2>&1; tail -2 $OUT/r05i_rate_$n.log | cut -c1-400; done | tee $OUT/r05i_owner_stream_rate.txt
for e in 3 6; do FWGPU_STREAM_CONSUMER_EIGHTHS=$e RANKS=2 STEPS=6 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-330 | sed "s/^/consumer eighths $e: /"; done | tee -a $OUT/r05i_owner_stream_rate.txt
timeout 300 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_learns" -s > $OUT/r05i_pytest_owner_stat.log 2>&1; echo "owner stat rc=$?"; grep "owner-side apply, streaming" $OUT/r05i_pytest_owner_stat.log; tail -3 $OUT/r05i_pytest_owner_stat.log
