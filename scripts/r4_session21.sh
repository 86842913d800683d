#!/bin/bash
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 2 20 -- "lds_keep auto|" "lds_keep 0|FWGPU_LDS_KEEP=0" "lds_keep 1|FWGPU_LDS_KEEP=1" "lds_keep 2|FWGPU_LDS_KEEP=2" 2>&1 | tee $OUT/short.txt
bash scripts/store_policy_ab.sh 2 150 -- "lds_keep auto|" "lds_keep 0|FWGPU_LDS_KEEP=0" 2>&1 | tee $OUT/long.txt
