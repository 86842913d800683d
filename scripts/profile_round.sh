#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel trace and the two PMC passes of the same command, single-rank RCCL run.
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh r01b
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
timeout 600 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace -- $CMD > $OUT/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/${TAG}_write -o write -- $CMD > $OUT/${TAG}_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write -name "*.db" | sort) > $OUT/${TAG}_kernel_rocprofv3.txt 2>&1
head -40 $OUT/${TAG}_kernel_rocprofv3.txt
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python3 bench.py --force-dist --no-cpu-baseline > $OUT/${TAG}_bench_dist1.json 2> $OUT/${TAG}_bench_dist1.err
tail -c 1200 $OUT/${TAG}_bench_dist1.json | head -c 1200
