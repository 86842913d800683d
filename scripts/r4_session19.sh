#!/bin/bash
OUT=gpurun_out/r4s; mkdir -p $OUT
V=$PWD/build/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py tests/test_gpu_config_a.py -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 3 20 -- "keep LAST 20, overflow first x8 (HEAD)|" "keep FIRST 20|FWGPU_LIBRARY=$V/libfwgpu_keepfirst.so" "keep last, overflow first x6|FWGPU_LIBRARY=$V/libfwgpu_ugf6.so" 2>&1 | tee $OUT/ab.txt
bash scripts/store_policy_ab.sh 1 150 -- "keep LAST (HEAD)|" "keep FIRST|FWGPU_LIBRARY=$V/libfwgpu_keepfirst.so" 2>&1 | tee $OUT/long.txt
