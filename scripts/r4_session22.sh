#!/bin/bash
OUT=gpurun_out/r4v; mkdir -p $OUT
V=$PWD/build/variants
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 3 20 -- "max 4 (auto 3)|" "max 0 (HEAD)|FWGPU_LIBRARY=$V/libfwgpu_lkm0.so" "max 3|FWGPU_LIBRARY=$V/libfwgpu_lkm3.so" 2>&1 | tee $OUT/short.txt
