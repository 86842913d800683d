#!/bin/bash
set -u
echo "A: torch imported + asked for the GPU after the library"; TORCH_AFTER=1 RECORDS=1 FIELDS=6 K=4 MEAN_EXTRA=0.5 BITS=20 B=3000 RANKS=4 STEPS=3 LG_ROWS=8 LG_LR=9 CWG=20 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -2 | cut -c1-160
echo "B: GPU_MAX_HW_QUEUES=4 forced"; GPU_MAX_HW_QUEUES=4 RECORDS=1 FIELDS=6 K=4 MEAN_EXTRA=0.5 BITS=20 B=3000 RANKS=4 STEPS=3 LG_ROWS=8 LG_LR=9 CWG=20 timeout 60 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | cut -c1-160
echo "C: the pytest case"; timeout 100 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_delivers and 4-8-9" 2>&1 | tail -3
echo "D: the pytest case, no torch in conftest"; FWGPU_TEST_ASSUME_GPU=1 timeout 100 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "streaming_form_delivers and 4-8-9" 2>&1 | tail -3
