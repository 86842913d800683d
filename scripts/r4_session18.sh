#!/bin/bash
OUT=gpurun_out/r4r; mkdir -p $OUT
timeout 1500 python scripts/holdout_spread.py 8 2>&1 | grep -v amdgpu | tee $OUT/holdout_spread.txt
bash scripts/store_policy_ab.sh 2 20 -- "HEAD|" 2>&1 | tee $OUT/ab.txt
