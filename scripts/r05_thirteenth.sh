#!/bin/bash
set -u
echo "A: 4 ranks, host records, config C shape 20-bit"; RECORDS=1 BITS=20 B=3000 RANKS=4 STEPS=3 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-160
echo "B: 4 ranks, device batches, 6 fields k=4, one feature per field"; FIELDS=6 K=4 MEAN_EXTRA=0.5 BITS=20 B=3000 RANKS=4 STEPS=3 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-160
echo "C: same with host records"; RECORDS=1 FIELDS=6 K=4 MEAN_EXTRA=0.5 BITS=20 B=3000 RANKS=4 STEPS=3 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-160
echo "D: same, regions 2^8 / 2^9, 20 consumer workgroups"; RECORDS=1 FIELDS=6 K=4 MEAN_EXTRA=0.5 BITS=20 B=3000 RANKS=4 STEPS=3 LG_ROWS=8 LG_LR=9 CWG=20 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-160
echo "E: 10 fields k=4, 18-bit, step 16384"; FIELDS=10 K=4 MEAN_EXTRA=0 BITS=18 B=16384 RANKS=4 STEPS=3 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-160
