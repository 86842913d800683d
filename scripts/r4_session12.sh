#!/bin/bash
OUT=gpurun_out/r4l; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 3 20 -- "HEAD (w stores nt)|" "gather loads nt|FWGPU_LIBRARY=$V/libfwgpu_gnt.so" "acc loads nt|FWGPU_LIBRARY=$V/libfwgpu_ant.so" "both loads nt|FWGPU_LIBRARY=$V/libfwgpu_gant.so" "flush 256|FWGPU_WB_FLUSH_EVERY=256" 2>&1 | tee $OUT/ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.txt
