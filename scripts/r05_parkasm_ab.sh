#!/bin/bash
# A/B on one box: parked rows' LDS-direct loads issued through inline asm, in front of the register rows' loads (-DFW_PARK_ASM=1) against the shipped build
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
PASSES=3 bash scripts/ab_bench.sh "--steps 20 --warmup 5 --no-config-e --no-config-b" default parkasm 2>&1 | tee $OUT/r05_parkasm_ab.txt
FWGPU_LIBRARY=$R/build/variants/libfwgpu_parkasm.so timeout 1200 python3 -m pytest tests -q -m gpu -x -rs > $OUT/r05_parkasm_gputest.log 2>&1; echo "gpu suite (parkasm) rc=$?"; tail -3 $OUT/r05_parkasm_gputest.log | cut -c1-200
