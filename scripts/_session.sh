mkdir -p gpurun_out
B="--fields 10 --k 4 --bits 22 --ffm-bits 22 --mean-extra 0 --zipf 1.1 --ids 100000 --p-weighted 0 --seed 20240611 --lr 0.1 --power-t 0.5 --batch 4096 --steps 200 --warmup 20 --holdout 65536 --no-cpu-baseline --no-traffic --no-config-e --no-config-b"
for rep in 1 2 3; do for ss in 1 0; do
  FWGPU_SMALL_SHAPE=$ss timeout 300 python bench.py $B > gpurun_out/b.json 2> gpurun_out/b.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/b.json") if l.startswith("{")][-1])
print("B small_shape=$ss rep $rep", round(d["value"]), round(d["final_logloss"],5), d.get("oracle_final_logloss"))
PY
done; done
timeout 900 python -m pytest tests/test_zz_gpu_hogwild_quality.py tests/test_gpu_config_a.py -q 2>&1 | tail -5
