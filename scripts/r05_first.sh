#!/bin/bash
# Round 5, first GPU-box call: (a) the 16-thread hogwild oracle curves at SURVEY 8d's length on the box's 256 host cores (3 runs, concurrent,
# CPU only), while (b) the GPU runs the rate-vs-skew sweep of the shipped kernel with traffic counters and (c) rocprofv3's counter list is kept.
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
make -C oracle native > /dev/null 2>&1
for r in 1 2 3; do
  CURVE_PRED_THREADS=32 CURVE_GEN_THREADS=8 nohup python3 scripts/make_bench_oracle_curve.py hog16 $r > $OUT/r05_oracle_hog16_r$r.log 2>&1 &
done
rocprofv3 -L > $OUT/r05_rocprofv3_list_avail.txt 2>&1
grep -i -E "mall|umc|dram|hbm|df_|fabric|EA0|EA_" $OUT/r05_rocprofv3_list_avail.txt | head -80 > $OUT/r05_memory_side_counters.txt
for z in 0 0.8 1.05 1.3; do
  timeout 600 python3 bench.py --zipf $z --steps 20 --warmup 5 --no-cpu-baseline --no-config-e > $OUT/r05_skew_z$z.json 2> $OUT/r05_skew_z$z.err
  python3 - $OUT/r05_skew_z$z.json $z <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"zipf {sys.argv[2]}: {d['value']/1e6:.3f} M ex/s  launch {r['avg_launch_ms']:.3f} ms  frac {r['frac']:.4f}  alg {r['algorithmic_bytes_per_launch']/1e9:.2f} GB  traffic {(r['traffic'] or 0)/1e9:.2f} GB  loss {d['final_logloss']:.4f}  | {r['traffic_source']}")
except Exception as e:
    print("zipf", sys.argv[2], "failed", e)
PY
done | tee $OUT/r05_skew_sweep.txt
# memory-side counter groups at the two extreme skews
for z in 0 1.3; do
  ZIPF=$z B=65536 bash scripts/pmc_probe.sh r05_pmc_z$z "shipped:window=1" > $OUT/r05_pmc_z$z.out 2>&1
done
wait
tail -3 $OUT/r05_oracle_hog16_r*.log
cp tests/golden/bench_oracle_curve_hog16_r*.json $OUT/
