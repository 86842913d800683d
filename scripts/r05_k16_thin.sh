#!/bin/bash
# headless k = 16: store policy 3 on the re-read two-chunk rows (thinned accumulator stores of hot rows) against FWGPU_THIN_REREAD=0; then the whole GPU suite
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout 300 python3 bench.py --k 16 --batch 16384 --steps 48 --warmup 4 --holdout 65536 --no-cpu-baseline --no-config-e --no-config-b $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$1:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(r.get('frac',0),4), 'traffic GB', round((r.get('traffic') or 0)/1e9,2))"; }
for p in 1 2; do
  run "pass $p shipped (hot re-read rows thinned)" ""
  FWGPU_THIN_REREAD=0 run "pass $p FWGPU_THIN_REREAD=0" ""
done 2>&1 | tee $OUT/r05_k16_thin_ab.txt
timeout 1200 python3 -m pytest tests -q -m gpu -x -rs -v > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; grep -E "FAILED|ERROR" $OUT/r05_gputest.log | head -5 | cut -c1-300; tail -4 $OUT/r05_gputest.log | cut -c1-200
