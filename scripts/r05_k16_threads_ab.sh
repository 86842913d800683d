#!/bin/bash
# headless k = 16 (two-chunk rows): the default launch shape (one 1024-thread workgroup per CU on the generic kernel) against 512 threads (the v2 kernel, NC = 2)
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout 300 python3 bench.py --k 16 --batch 16384 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), d['roofline']['frac'] if 'roofline' in d else '')"; }
for p in 1 2; do
  run "pass $p default" ""
  run "pass $p threads512" "--threads 512"
  run "pass $p threads512 wgs2" "--threads 512 --wgs-per-cu 2"
done 2>&1 | tee $OUT/r05_k16_threads_ab.txt
