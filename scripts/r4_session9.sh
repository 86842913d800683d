#!/bin/bash
OUT=gpurun_out/r4i; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 2 20 -- "maxr14|" "maxr16|FWGPU_LIBRARY=$V/libfwgpu_m16.so" "maxr18ug2|FWGPU_LIBRARY=$V/libfwgpu_m18ug2.so" "maxr20ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m20ug2ua1.so" "maxr20|FWGPU_LIBRARY=$V/libfwgpu_m20.so" 2>&1 | tee $OUT/maxr_ab.txt
bash scripts/store_policy_ab.sh 1 150 -- "maxr18ug2|FWGPU_LIBRARY=$V/libfwgpu_m18ug2.so" "maxr20ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m20ug2ua1.so" "maxr20|FWGPU_LIBRARY=$V/libfwgpu_m20.so" "maxr14|" "maxr16|FWGPU_LIBRARY=$V/libfwgpu_m16.so" 2>&1 | tee $OUT/maxr_long.txt
e() { env "$@" timeout 900 python3 bench.py --k 16 --nn-layers 2 --nn-width 256 --batch 8192 --steps $STEPS --warmup 4 --no-cpu-baseline --no-traffic --no-config-e 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d.get('logloss_after_examples',{}).items()})"; }
for pass in 1 2; do
  STEPS=24; echo -n "E default  "; e A=1; echo -n "E dense wb "; e FWGPU_LIBRARY=$V/libfwgpu_nnwb.so
done 2>&1 | tee $OUT/configE_ab.txt
STEPS=120; echo -n "E long default  "; e A=1 2>&1 | tee -a $OUT/configE_ab.txt; echo -n "E long dense wb "; e FWGPU_LIBRARY=$V/libfwgpu_nnwb.so 2>&1 | tee -a $OUT/configE_ab.txt
