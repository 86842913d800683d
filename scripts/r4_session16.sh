#!/bin/bash
OUT=gpurun_out/r4p; mkdir -p $OUT
V=$PWD/build/variants
QUICK=1 B=65536 FWGPU_LIBRARY=$V/libfwgpu_ticks.so timeout 600 python scripts/perf_probe.py 2>&1 | grep -v amdgpu | tee $OUT/ticks.txt
