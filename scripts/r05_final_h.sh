#!/bin/bash
# the bench lines at the round's final commit
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python3 bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err; echo "bench rc=$?"
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config-e --no-config-b > $OUT/r05_bench_driver_shape.json 2> $OUT/r05_bench_driver_shape.err; echo "driver shape rc=$?"
python3 - <<'PY'
import json
for n in ("r05_bench","r05_bench_driver_shape"):
    d=json.loads([l for l in open(f"gpurun_out/{n}.json") if l.startswith("{")][-1])
    print(n, round(d["value"]), round(d["roofline"]["frac"],4), round(d["final_logloss"],4), {k:(round(v["value"]), v.get("final_logloss")) for k,v in d.items() if k.startswith("config_") and isinstance(v,dict) and "value" in v})
PY
