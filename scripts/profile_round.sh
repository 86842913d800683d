#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel trace and the two PMC passes of the same command, single-rank RCCL run.
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh r01b
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
timeout 900 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
# the driver's shape, and the protocol at its stated length (16 Mi examples, three passes)
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_shape.json 2> $OUT/${TAG}_bench_driver_shape.err
timeout 900 python3 bench.py --long > $OUT/${TAG}_bench_long.json 2> $OUT/${TAG}_bench_long.err
tail -c 400 $OUT/${TAG}_bench_long.json
# ... run out: 64 Mi examples, two passes, against the 16-thread hogwild oracle's committed curve (round 6)
timeout 1200 python3 bench.py --long --examples 67108864 --long-passes 2 > $OUT/${TAG}_bench_long64.json 2> $OUT/${TAG}_bench_long64.err
tail -c 300 $OUT/${TAG}_bench_long64.json
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --no-cpu-baseline --no-traffic --no-config-e --no-config-b"  # (the default shape: 4 + 48 steps -- the kernel gets faster over the first steps as rows turn hot, store policy 4)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace -- $CMD > $OUT/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/${TAG}_write -o write -- $CMD > $OUT/${TAG}_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write -name "*.db" | sort) > $OUT/${TAG}_kernel_rocprofv3.txt 2>&1
head -40 $OUT/${TAG}_kernel_rocprofv3.txt
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write  # (the rocpd databases: tens of MB each; gpurun merges at most 64 MiB back -- round 6 lost a whole evidence run to them)
# the multi-GPU code paths with ONE rank over RCCL (library communicator): replica exchange + sharded leg, then the sharded step as the timed mode
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python3 bench.py --force-dist --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_dist1_replica.json 2> $OUT/${TAG}_bench_dist1_replica.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python3 bench.py --force-dist --dp-mode sharded --steps 192 --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench_dist1_sharded.json 2> $OUT/${TAG}_bench_dist1_sharded.err
tail -c 700 $OUT/${TAG}_bench_dist1_replica.json; echo; tail -c 400 $OUT/${TAG}_bench_dist1_sharded.json; echo
# config E (configs[4]): exact head (the default) and the mini-batched MFMA head
timeout 600 python3 bench.py --k 16 --nn-layers 2 --batch 8192 --steps 24 --warmup 2 > $OUT/${TAG}_configE_exact.json 2> $OUT/${TAG}_configE_exact.err
timeout 600 python3 bench.py --k 16 --nn-layers 2 --head minibatch --batch 1024 --steps 192 --warmup 4 --no-cpu-baseline > $OUT/${TAG}_configE_minibatch.json 2> $OUT/${TAG}_configE_minibatch.err
tail -c 500 $OUT/${TAG}_configE_exact.json | head -c 300; echo
# config E under the profiler: kernel trace + the two PMC passes of the exact-head run, and the predict-only (batched head) launches
cd /tmp
ECMD="python3 $R/bench.py --k 16 --nn-layers 2 --batch 8192 --steps 12 --warmup 2 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_E_trace -o trace -- $ECMD > $OUT/${TAG}_E_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/${TAG}_E_fetch -o fetch -- $ECMD > $OUT/${TAG}_E_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/${TAG}_E_write -o write -- $ECMD > $OUT/${TAG}_E_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/${TAG}_E_trace $OUT/${TAG}_E_fetch $OUT/${TAG}_E_write -name "*.db" | sort) > $OUT/${TAG}_configE_rocprofv3.txt 2>&1
head -12 $OUT/${TAG}_configE_rocprofv3.txt
rm -rf $OUT/${TAG}_E_trace $OUT/${TAG}_E_fetch $OUT/${TAG}_E_write
