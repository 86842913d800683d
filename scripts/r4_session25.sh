#!/bin/bash
OUT=gpurun_out/r4y; mkdir -p $OUT
V=$PWD/build/variants
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py -q -m gpu -p no:cacheprovider --tb=line 2>&1 | tail -8 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 3 20 -- "parked rows LDS-direct|" "parked through registers (3ecd7dd)|FWGPU_LIBRARY=$V/libfwgpu_parkregs.so" 2>&1 | tee $OUT/short.txt
for b in 512 1024 2048 4096; do MODES=owner OWNER_BATCH=$b TOTAL=131072 timeout 600 python scripts/group_modes_run.py 2>&1 | tee -a $OUT/owner.txt; done
