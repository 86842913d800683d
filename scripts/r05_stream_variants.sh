#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for v in default kr1 kr4 kr4pu8; do
  if [ "$v" = default ]; then unset FWGPU_LIBRARY; else export FWGPU_LIBRARY=$R/build/variants/libfwgpu_$v.so; fi
  for n in 1 4; do FWGPU_STREAM_CONSUMER_EIGHTHS=4 RANKS=$n STEPS=8 timeout 150 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$v', d['ranks'], 'ranks:', round(d['examples_per_sec']), 'hold-out', round(d['holdout_logloss_65536'],4))
except Exception as e: print('$v $n ranks: failed')"; done
done | tee $OUT/r05_stream_variants.txt
