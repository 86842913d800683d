#!/bin/bash
# Round 5, after the scalar-spill work: the whole GPU suite, then the default bench line and the driver's shape
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_conservation.py -q -m gpu -x -s 2>&1 | grep -E "passed|failed|adagrad|^ +[0-9]" | cut -c1-150
timeout 1200 python3 -m pytest tests -q -m gpu -x -rs -v > $OUT/r05_gputest.log 2>&1; echo "gpu suite rc=$?"; grep -E "FAILED|ERROR" $OUT/r05_gputest.log | head -5 | cut -c1-300; tail -4 $OUT/r05_gputest.log | cut -c1-200
timeout 900 python3 bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err; echo "bench rc=$?"
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_driver_shape.json 2> $OUT/r05_bench_driver_shape.err; echo "driver shape rc=$?"
python3 - <<'PY'
import json
for n in ("r05_bench","r05_bench_driver_shape"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{n}.json") if l.startswith("{")][-1])
        print(n, round(d["value"]), round(d["roofline"]["frac"],4), round(d["final_logloss"],4), {k:(round(v["value"]), v.get("final_logloss")) for k,v in d.items() if k.startswith("config_") and isinstance(v,dict) and "value" in v})
    except Exception as e:
        print(n, "unreadable", e)
PY
