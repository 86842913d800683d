#!/bin/bash
OUT=gpurun_out/r4t; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 2 150 -- "keep LAST (HEAD)|" "keep FIRST|FWGPU_LIBRARY=$V/libfwgpu_keepfirst.so" 2>&1 | tee $OUT/long.txt
