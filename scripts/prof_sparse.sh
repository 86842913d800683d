#!/bin/bash
# kernel trace of the row-sparse multi-GPU step on one rank (usage on the GPU box: bash scripts/prof_sparse.sh [batch])
set -u
B=${1:-2048}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/sparse_trace -o trace -- python3 $R/bench.py --force-dist --dp-mode sparse --batch $B --steps 24 --warmup 4 --no-cpu-baseline --no-traffic > $OUT/sparse_trace.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/sparse_trace -name "*.db" | sort) > $OUT/sparse_kernel_rocprofv3.txt 2>&1
head -30 $OUT/sparse_kernel_rocprofv3.txt | cut -c1-220
tail -2 $OUT/sparse_trace.log | cut -c1-300
