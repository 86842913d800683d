#!/bin/bash
OUT=gpurun_out/r4j; mkdir -p $OUT
V=$PWD/build/variants
bash scripts/store_policy_ab.sh 2 20 -- "maxr20ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m20.so" "maxr22ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m22.so" "maxr24ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m24.so" "maxr20ug1ua1|FWGPU_LIBRARY=$V/libfwgpu_m20ug1.so" 2>&1 | tee $OUT/maxr_ab.txt
bash scripts/store_policy_ab.sh 1 150 -- "maxr22ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m22.so" "maxr24ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m24.so" "maxr20ug2ua1|FWGPU_LIBRARY=$V/libfwgpu_m20.so" 2>&1 | tee $OUT/maxr_long.txt
