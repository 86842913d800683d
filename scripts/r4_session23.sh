#!/bin/bash
OUT=gpurun_out/r4w; mkdir -p $OUT
V=$PWD/build/variants
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py -q -m gpu -p no:cacheprovider --tb=line 2>&1 | tail -25 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 2 20 -- "depth 3|" "HEAD|FWGPU_LIBRARY=$V/libfwgpu_lkm0.so" "depth 1|FWGPU_LIBRARY=$V/libfwgpu_pd1.so" "depth 2|FWGPU_LIBRARY=$V/libfwgpu_pd2.so" "depth 4|FWGPU_LIBRARY=$V/libfwgpu_pd4.so" "depth 6|FWGPU_LIBRARY=$V/libfwgpu_pd6.so" 2>&1 | tee $OUT/short.txt
