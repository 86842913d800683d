#!/bin/bash
OUT=gpurun_out/r4m; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 2 20 -- "HEAD|" "acc S6 stores|FWGPU_LIBRARY=$V/libfwgpu_accs6.so" 2>&1 | tee $OUT/ab.txt
FWGPU_LIBRARY=$V/libfwgpu_accs6.so timeout 600 python -m pytest tests/test_gpu_conservation.py -q -s -p no:cacheprovider -k "adagrad" 2>&1 | grep -v "^E \|^    " | tail -30 | tee $OUT/conservation_s6.txt
bash scripts/store_policy_ab.sh 1 150 -- "acc S6 stores|FWGPU_LIBRARY=$V/libfwgpu_accs6.so" 2>&1 | tee $OUT/long.txt
