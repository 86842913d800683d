#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_scale_launch.py -q -m gpu -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r05_trace -o trace -- $CMD > $OUT/r05_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/r05_fetch -o fetch -- $CMD > $OUT/r05_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/r05_write -o write -- $CMD > $OUT/r05_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write -name "*.db" | sort) > $OUT/r05_kernel_rocprofv3_full.txt 2>&1
rm -rf $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write
grep -n "^## \|kernel_r<300, true, 20, true, 1, 3>(fwgpu::KernelParams), [0-9]*, " $OUT/r05_kernel_rocprofv3_full.txt | cut -c1-170
tail -1 $OUT/r05_trace.log | cut -c1-300
