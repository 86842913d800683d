#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout 300 python3 bench.py --k 16 --batch 16384 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), d['roofline']['frac'] if 'roofline' in d else '')"; }
run "after the change, default" "" | tee $OUT/r05_k16_after.txt
run "after the change, threads 1024" "--threads 1024" | tee -a $OUT/r05_k16_after.txt
timeout 1200 python3 -m pytest tests -q -m gpu -x -rs -v > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; grep -E "FAILED|ERROR" $OUT/r05_gputest.log | head -5 | cut -c1-300; tail -4 $OUT/r05_gputest.log | cut -c1-200
