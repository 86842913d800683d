#!/bin/bash
OUT=gpurun_out/r4ab; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 3 20 -- "early 0 (shipped)|FWGPU_LIBRARY=$V/libfwgpu_parkdirect.so" "early 1|FWGPU_LIBRARY=$V/libfwgpu_ae1.so" "early 2|FWGPU_LIBRARY=$V/libfwgpu_ae2.so" "early 3|FWGPU_LIBRARY=$V/libfwgpu_ae3.so" 2>&1 | tee $OUT/short.txt
