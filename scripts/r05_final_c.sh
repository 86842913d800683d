#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash scripts/profile_round.sh r05 > $OUT/r05_profile_round.log 2>&1
rm -rf $OUT/r05_trace $OUT/r05_fetch $OUT/r05_write $OUT/r05_E_trace $OUT/r05_E_fetch $OUT/r05_E_write   # (the rocprofv3 databases: summarised into the *_rocprofv3.txt files above)
timeout 600 python3 -m pytest tests/test_gpu_conservation.py -q -m gpu -s -k "hot_ffm" > $OUT/r05_conservation.log 2>&1; grep -A8 "surviving fraction" $OUT/r05_conservation.log > $OUT/r05_conservation.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ls -la $OUT | head -40; du -sh $OUT
