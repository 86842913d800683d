mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conservation.py -x -q 2>&1 | tail -2)
E="--k 16 --nn-layers 2 --nn-width 256 --head exact --batch 8192 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
K="--k 16 --batch 16384 --steps 24 --warmup 4 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
for rep in 1 2; do
for lib in fwumious_wabbit_amd/lib/libfwgpu.so build/variants/libfwgpu_uo3.so build/variants/libfwgpu_uo4.so; do
  FWGPU_LIBRARY=$lib timeout 300 python bench.py $E > gpurun_out/e.json 2> gpurun_out/e.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/e.json") if l.startswith("{")][-1])
    print("E   $lib rep $rep", round(d["value"]), round(d["final_logloss"],5), round(d["roofline"]["frac"],4))
except Exception as e:
    print("E failed", e); print(open("gpurun_out/e.err").read()[-1500:])
PY
  FWGPU_LIBRARY=$lib timeout 300 python bench.py $K > gpurun_out/e.json 2> gpurun_out/e.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/e.json") if l.startswith("{")][-1])
    print("k16 $lib rep $rep", round(d["value"]), round(d["final_logloss"],5), round(d["roofline"]["frac"],4))
except Exception as e:
    print("k16 failed", e); print(open("gpurun_out/e.err").read()[-1500:])
PY
done; done
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-config-e --no-config-b > gpurun_out/x.json 2> gpurun_out/x.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/x.json") if l.startswith("{")][-1])
print("config C default shape", round(d["value"]), round(d["final_logloss"],5), round(d["roofline"]["frac"],4))
PY
