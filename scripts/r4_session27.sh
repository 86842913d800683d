#!/bin/bash
OUT=gpurun_out/r4aa; mkdir -p $OUT
for n in 98304 131072 196608; do echo "n_train $n"; Z13_K8_TRAIN=$n timeout 1200 python scripts/holdout_spread.py 8 zipf13_noise_k8_win 2>&1 | tail -1; done | tee $OUT/z13.txt
