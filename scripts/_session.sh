mkdir -p gpurun_out
echo "=== FW_PARK_ASM A/B: bench.py default shape (48 + 4 launches of 65 536), interleaved"
for rep in 1 2 3; do
for lib in fwumious_wabbit_amd/lib/libfwgpu.so build/variants/libfwgpu_parkasm.so; do
  FWGPU_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-config-e --no-config-b > gpurun_out/x.json 2> gpurun_out/x.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/x.json") if l.startswith("{")][-1])
print("$lib rep $rep", round(d["value"]), round(d["final_logloss"],5), round(d["roofline"]["frac"],4))
PY
done; done
echo "=== config B table"
timeout 1500 python scripts/configB_table.py 2>&1 | tee gpurun_out/r06_configB_table.txt
