#!/bin/bash
OUT=gpurun_out/r4q; mkdir -p $OUT
V=$PWD/build/variants
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py tests/test_gpu_overlap.py -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/tests.txt
bash scripts/store_policy_ab.sh 3 20 -- "LR pair kept from the forward pass (HEAD)|" "LR entries reloaded|FWGPU_LIBRARY=$V/libfwgpu_nolrkeep.so" 2>&1 | tee $OUT/ab.txt
bash scripts/store_policy_ab.sh 2 150 -- "LR pair kept (HEAD)|" "LR entries reloaded|FWGPU_LIBRARY=$V/libfwgpu_nolrkeep.so" 2>&1 | tee $OUT/long.txt
timeout 900 python scripts/holdout_spread.py 4 short_fused hogwild_96k config_b zipf13_noise zipf13_noise_k8_win 2>&1 | grep -v amdgpu | tee $OUT/spread.txt
