#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 3 20 -- "shipped|" "scalar spills in scratch|FWGPU_LIBRARY=$V/libfwgpu_nolane.so" 2>&1 | tee $OUT/short.txt
timeout 2400 python scripts/holdout_spread.py 8 2>&1 | tee $OUT/spread.txt
