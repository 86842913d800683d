#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
echo "A: config C shape, 20-bit tables, 4 ranks, global step 16384"; BITS=20 B=16384 RANKS=4 STEPS=4 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-200
echo "B: same, global step 3000"; BITS=20 B=3000 RANKS=4 STEPS=4 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-200
echo "C: same, small regions 2^8 / 2^9"; BITS=20 B=3000 RANKS=4 STEPS=4 LG_ROWS=8 LG_LR=9 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-200
echo "D: same, 20 consumer workgroups"; BITS=20 B=3000 RANKS=4 STEPS=4 LG_ROWS=8 LG_LR=9 CWG=20 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-200
echo "E: 2 ranks, small regions, 10 consumer workgroups"; BITS=20 B=3000 RANKS=2 STEPS=4 LG_ROWS=8 LG_LR=9 CWG=10 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-200
