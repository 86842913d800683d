#!/bin/bash
OUT=gpurun_out/r4n; mkdir -p $OUT
V=$PWD/build/variants
FWGPU_LIBRARY=$V/libfwgpu_pipe20.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/tests_pipe20.txt
bash scripts/store_policy_ab.sh 3 20 -- "HEAD|" "pipelined kept rows (20)|FWGPU_LIBRARY=$V/libfwgpu_pipe20.so" "pipelined kept rows (18)|FWGPU_LIBRARY=$V/libfwgpu_pipe18.so" 2>&1 | tee $OUT/ab.txt
bash scripts/store_policy_ab.sh 1 150 -- "pipelined kept rows (20)|FWGPU_LIBRARY=$V/libfwgpu_pipe20.so" "HEAD|" 2>&1 | tee $OUT/long.txt
