#!/bin/bash
OUT=gpurun_out/r4o; mkdir -p $OUT
V=$PWD/build/variants
FWGPU_LIBRARY=$V/libfwgpu_pipe2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/tests_pipe2.txt
bash scripts/store_policy_ab.sh 3 20 -- "pipe kept|FWGPU_LIBRARY=$V/libfwgpu_pipe1.so" "pipe kept + 8 overflow slots|FWGPU_LIBRARY=$V/libfwgpu_pipe2.so" "pipe kept + 5 slots|FWGPU_LIBRARY=$V/libfwgpu_pipe2s5.so" "pipe kept + 12 slots|FWGPU_LIBRARY=$V/libfwgpu_pipe2s12.so" 2>&1 | tee $OUT/ab.txt
bash scripts/store_policy_ab.sh 1 150 -- "pipe kept + 8 overflow slots|FWGPU_LIBRARY=$V/libfwgpu_pipe2.so" 2>&1 | tee $OUT/long.txt
