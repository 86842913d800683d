#!/bin/bash
# What binds the headless k = 16 kernel (two-chunk rows, no kept rows): duration and HBM traffic per launch of the updating instantiation
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --k 16 --batch 16384 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/k16_trace -o trace -- $CMD > $OUT/k16_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/k16_fetch -o fetch -- $CMD > $OUT/k16_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/k16_write -o write -- $CMD > $OUT/k16_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/k16_trace $OUT/k16_fetch $OUT/k16_write -name "*.db" | sort) > $OUT/r05_k16_headless_rocprofv3.txt 2>&1
rm -rf $OUT/k16_trace $OUT/k16_fetch $OUT/k16_write
grep -n "^## \|kernel_r<300, true, 0, true, 2, 3>(fwgpu::KernelParams), [0-9]*, " $OUT/r05_k16_headless_rocprofv3.txt | cut -c1-170
grep "^{" $OUT/k16_trace.log | tail -1 | cut -c1-600
