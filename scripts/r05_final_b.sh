#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -q -m gpu -x -rs -v > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; grep -E "PASSED|FAILED|SKIPPED|ERROR" $OUT/r05_gputest.log | tail -4 | cut -c1-200; tail -25 $OUT/r05_gputest.log | cut -c1-200
