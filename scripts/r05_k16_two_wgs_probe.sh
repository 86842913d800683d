#!/bin/bash
# What would a second workgroup per CU be worth at k = 16?  A stream with fewer features per example (--mean-extra 4: the entries' own slots then leave room for two workgroups'
# LDS on a CU), one workgroup per CU forced against the automatic choice.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout 300 python3 bench.py --k 16 --mean-extra 4 --batch 16384 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-config-e --no-config-b $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$1:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(r.get('frac',0),4), 'traffic/algorithmic', round((r.get('traffic') or 0)/max(1,r.get('bytes_per_launch',0) or 1),3) if r.get('bytes_per_launch') else r.get('traffic'))"; }
for p in 1 2; do
  run "pass $p automatic" ""
  run "pass $p one workgroup per CU" "--wgs-per-cu 1"
done 2>&1 | tee $OUT/r05_k16_two_wgs_probe.txt
