#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for th in 0.25 0.5 1 2; do
  FWGPU_ACC_HOT_THETA=$th timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-config-e --no-config-b 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver shape, theta $th:', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_logloss'],4))"
done | tee $OUT/r05_theta_driver.txt
for th in 0.5 1; do
  FWGPU_ACC_HOT_THETA=$th timeout 400 python3 bench.py --long --long-passes 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['logloss_after_examples']
print('long, theta $th:', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'final', [round(x,4) for x in d['final_logloss_passes']], 'curve', {k: round(v[0],4) for k,v in c.items() if int(k) % 4194304 == 0})"
done | tee $OUT/r05_theta_long.txt
