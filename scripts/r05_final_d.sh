#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_overlap.py tests/test_gpu_parity.py -q -m gpu -x -k "two_learners or below_4096 or oversize" 2>&1 | tail -4
