#!/bin/bash
OUT=gpurun_out/r4g; mkdir -p $OUT
echo "== owner apply"; timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_dist_procs.py -q -x -p no:cacheprovider -k "owner" -s 2>&1 | tail -40 | tee $OUT/owner.txt
echo "== rest"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/gputest.txt
