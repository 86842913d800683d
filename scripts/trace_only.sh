R=$PWD; OUT=$R/gpurun_out; TAG=r03e
timeout 600 python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o trace -- $CMD > $OUT/${TAG}_trace.log 2>&1
cd $R
python3 scripts/rocprof_summary.py $(find $OUT/${TAG}_trace -name "*.db" | sort) > $OUT/${TAG}_kernel_trace.txt 2>&1
head -6 $OUT/${TAG}_kernel_trace.txt | cut -c1-150
tail -1 $OUT/${TAG}_trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced run itself:', d['value'], d['roofline']['avg_launch_ms'], d['table_placement'])"
python3 -c "
import json
d=json.loads([l for l in open('$OUT/${TAG}_bench.json') if l.startswith('{')][-1]); print('bench:', d['value'], d['final_logloss'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'], d['table_placement'])"
