#!/bin/bash
# Kernel traces of the other paths: config E (generic kernel + deep head) and the one-rank replica exchange (delta kernels).
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_traceE -o traceE -- python3 $R/bench.py --k 16 --nn-layers 2 --nn-width 256 --steps 4 --warmup 1 --batch 8192 --no-cpu-baseline > $OUT/${TAG}_traceE.log 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_traceD -o traceD -- python3 $R/bench.py --force-dist --steps 8 --warmup 2 --sync-every 4 --no-cpu-baseline > $OUT/${TAG}_traceD.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/${TAG}_traceE $OUT/${TAG}_traceD -name "*.db" | sort) > $OUT/${TAG}_extra_kernels_rocprofv3.txt 2>&1
cat $OUT/${TAG}_extra_kernels_rocprofv3.txt | cut -c1-170
