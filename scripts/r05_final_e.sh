#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -q -m gpu -x -rs -v > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/r05_gputest.log | cut -c1-200
timeout 900 python3 bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench_driver_shape.json 2> $OUT/r05_bench_driver_shape.err
timeout 900 python3 bench.py --long > $OUT/r05_bench_long.json 2> $OUT/r05_bench_long.err
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r05_trace -o trace -- $CMD > $OUT/r05_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/r05_fetch -o fetch -- $CMD > $OUT/r05_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/r05_write -o write -- $CMD > $OUT/r05_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write -name "*.db" | sort) > $OUT/r05_kernel_rocprofv3.txt 2>&1
rm -rf $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write
python3 - <<'PY'
import json
for n in ("r05_bench","r05_bench_driver_shape","r05_bench_long"):
    d=json.loads([l for l in open(f"gpurun_out/{n}.json") if l.startswith("{")][-1])
    print(n, round(d["value"]), round(d["roofline"]["frac"],4), d.get("final_logloss_passes") or round(d["final_logloss"],4))
PY
