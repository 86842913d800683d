mkdir -p gpurun_out
E="--k 16 --nn-layers 2 --nn-width 256 --head exact --batch 8192 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
for rep in 1 2 3 4 5 6 7 8; do
  timeout 300 python bench.py $E > gpurun_out/e.json 2> gpurun_out/e.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/e.json") if l.startswith("{")][-1])
print("E rep $rep", round(d["value"]), round(d["final_logloss"],5), d.get("oracle_final_logloss"), round(d["roofline"]["frac"],4))
PY
done
echo "=== predict-only E at two launch sizes"
timeout 300 python scripts/e_predict_rate.py 2>&1 | tail -5
B=8192 timeout 300 python scripts/e_predict_rate.py 2>&1 | tail -5
