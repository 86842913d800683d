#!/bin/bash
# headless k = 16 after the entries' own slots left the LDS of concurrent whole-line updates (two workgroups per CU), against the LDS copy forced back (FWGPU_SELFW_LDS=1: one workgroup per CU);
# then the whole GPU suite and the default bench line
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { timeout 300 python3 bench.py --k 16 --batch 16384 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-config-e --no-config-b 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$1:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(r.get('frac',0),4), 'traffic GB', round((r.get('traffic') or 0)/1e9,2))"; }
for p in 1 2; do
  run "pass $p shipped (own slots from the re-read row)"
  FWGPU_SELFW_LDS=1 run "pass $p FWGPU_SELFW_LDS=1 (own slots in LDS)"
done 2>&1 | tee $OUT/r05_k16_selfw_ab.txt
timeout 1200 python3 -m pytest tests -q -m gpu -x -rs -v > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; grep -E "FAILED|ERROR" $OUT/r05_gputest.log | head -5 | cut -c1-300; tail -4 $OUT/r05_gputest.log | cut -c1-200
timeout 900 python3 bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_bench.json") if l.startswith("{")][-1])
print("r05_bench", round(d["value"]), round(d["roofline"]["frac"],4), round(d["final_logloss"],4), {k:(round(v["value"]), v.get("final_logloss")) for k,v in d.items() if k.startswith("config_") and isinstance(v,dict) and "value" in v})
PY
