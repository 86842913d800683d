mkdir -p gpurun_out
E="--k 16 --nn-layers 2 --nn-width 256 --head exact --batch 8192 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
for rep in 1 2; do
for lib in fwumious_wabbit_amd/lib/libfwgpu.so build/variants/libfwgpu_e640.so build/variants/libfwgpu_e768.so build/variants/libfwgpu_e1024.so; do
  FWGPU_LIBRARY=$lib timeout 300 python bench.py $E > gpurun_out/e.json 2> gpurun_out/e.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/e.json") if l.startswith("{")][-1])
    print("E   $lib rep $rep", round(d["value"]), round(d["final_logloss"],5), round(d["roofline"]["frac"],4))
except Exception as e:
    print("E failed $lib", e); print(open("gpurun_out/e.err").read()[-800:])
PY
done; done
