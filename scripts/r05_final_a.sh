#!/bin/bash
# Round 5 evidence, part A: the whole GPU suite at this commit, then the conservation rig's table and the round profile
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -q -m gpu -x -rs > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; tail -6 $OUT/r05_gputest.log | cut -c1-250
timeout 600 python3 -m pytest tests/test_gpu_conservation.py -q -m gpu -s -k "hot_ffm" > $OUT/r05_conservation.log 2>&1; grep -A8 "surviving fraction" $OUT/r05_conservation.log > $OUT/r05_conservation.txt; tail -9 $OUT/r05_conservation.txt
bash scripts/profile_round.sh r05 > $OUT/r05_profile_round.log 2>&1; tail -5 $OUT/r05_profile_round.log | cut -c1-400
