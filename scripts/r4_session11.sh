#!/bin/bash
OUT=gpurun_out/r4k; mkdir -p $OUT
V=$PWD/build/variants
echo "== gpu suite on the 20-kept-rows build"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | tee $OUT/gputest.txt
bash scripts/store_policy_ab.sh 2 20 -- "shipped (p1 f128, 20 kept)|" "w stores nt|FWGPU_LIBRARY=$V/libfwgpu_ntw.so" "p2 f128|FWGPU_STORE_POLICY=2" 2>&1 | tee $OUT/ab.txt
bash scripts/store_policy_ab.sh 2 150 -- "shipped|" "w stores nt|FWGPU_LIBRARY=$V/libfwgpu_ntw.so" 2>&1 | tee $OUT/long.txt
bash scripts/store_policy_ab.sh 1 150 -- "p2 f128|FWGPU_STORE_POLICY=2" 2>&1 | tee -a $OUT/long.txt
