#!/bin/bash
OUT=gpurun_out/r4e; mkdir -p $OUT
echo "== failing dist test"; timeout 600 python -m pytest "tests/test_gpu_dist_procs.py" -q -x -p no:cacheprovider -k "True" 2>&1 | tail -60 | tee $OUT/dist_fail.txt
echo "== new tests"; timeout 1500 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_scale_launch.py -q -x -p no:cacheprovider 2>&1 | tail -30 | tee $OUT/new_tests.txt
echo "== config A in flight"; python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/config_a.txt
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, test_zz_gpu_hogwild_quality as Q
for inf in (64, 32, 16, 8, 4):
    v=[Q.scenario_config_a_hogwild(in_flight=inf) for _ in range(6)]
    g=np.array([a for a,_ in v]); print(inf, v[0][1], g.min(), g.max(), np.abs(g-v[0][1]).max(), flush=True)
PY
run() { # name steps env...
  local name=$1; local steps=$2; shift; shift
  env "$@" timeout 900 python3 bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
V=$PWD/build/variants
for pass in 1 2 3; do
  run "diet (HEAD)     " 20
  run "pre-diet kernels" 20 FWGPU_LIBRARY=$V/libfwgpu_prediet.so
done 2>&1 | tee $OUT/diet_ab.txt
echo "== config E"; timeout 600 python3 bench.py --k 16 --nn-layers 2 --nn-width 256 --batch 8192 --steps 24 --warmup 4 --no-cpu-baseline --no-traffic --no-config-e 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config E', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4))" | tee $OUT/configE.txt
FWGPU_LIBRARY=$V/libfwgpu_prediet.so timeout 600 python3 bench.py --k 16 --nn-layers 2 --nn-width 256 --batch 8192 --steps 24 --warmup 4 --no-cpu-baseline --no-traffic --no-config-e 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config E prediet', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4))" | tee -a $OUT/configE.txt
