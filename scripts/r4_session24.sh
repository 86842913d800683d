#!/bin/bash
OUT=gpurun_out/r4x; mkdir -p $OUT
V=$PWD/build/variants
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=line 2>&1 | tail -12 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 2 150 -- "parked rows + depth 3|" "HEAD|FWGPU_LIBRARY=$V/libfwgpu_lkm0.so" 2>&1 | tee $OUT/long.txt
