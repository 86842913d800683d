#!/bin/bash
# Round 5, final evidence after the register work: protocol at full length, rocprofv3 trace + PMC passes of the default command, the same for config E
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python3 bench.py --long > $OUT/r05_bench_long.json 2> $OUT/r05_bench_long.err; echo "long rc=$?"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/r05_trace -o trace -- $CMD > $OUT/r05_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/r05_fetch -o fetch -- $CMD > $OUT/r05_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/r05_write -o write -- $CMD > $OUT/r05_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write -name "*.db" | sort) > $OUT/r05_kernel_rocprofv3.txt 2>&1
rm -rf $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write
cd /tmp
ECMD="python3 $R/bench.py --k 16 --nn-layers 2 --batch 8192 --steps 12 --warmup 2 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/r05_E_trace -o trace -- $ECMD > $OUT/r05_E_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $OUT/r05_E_fetch -o fetch -- $ECMD > $OUT/r05_E_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $OUT/r05_E_write -o write -- $ECMD > $OUT/r05_E_write.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/r05_E_trace $OUT/r05_E_fetch $OUT/r05_E_write -name "*.db" | sort) > $OUT/r05_configE_rocprofv3.txt 2>&1
rm -rf $OUT/r05_E_trace $OUT/r05_E_fetch $OUT/r05_E_write
grep -n "^## \|kernel_r<300, true, 20, true, 1, 3>(fwgpu::KernelParams), [0-9]*, " $OUT/r05_kernel_rocprofv3.txt | cut -c1-170
head -14 $OUT/r05_configE_rocprofv3.txt | cut -c1-170
tail -1 $OUT/r05_trace.log | cut -c1-300
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_bench_long.json") if l.startswith("{")][-1])
print("long", round(d["value"]), round(d["roofline"]["frac"],4), d.get("final_logloss_passes"), d.get("final_logloss_vs_oracle"))
PY
