#!/bin/bash
# Round 5, fourth GPU-box call: streaming owner-side apply (tests + rate), oversize example with a head, kept-rows variants on the long protocol, the default line
set -u
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "oversize or chunked" > $OUT/r05d_pytest_oversize.log 2>&1; echo "oversize rc=$?"; tail -3 $OUT/r05d_pytest_oversize.log
timeout 900 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "owner" > $OUT/r05d_pytest_owner.log 2>&1; echo "owner rc=$?"; tail -5 $OUT/r05d_pytest_owner.log
for n in 1 2 4; do RANKS=$n STEPS=10 timeout 600 python3 scripts/owner_stream_rate.py 2>&1 | tail -1; done | tee $OUT/r05d_owner_stream_rate.txt
RANKS=4 STEPS=10 CWG=96 timeout 600 python3 scripts/owner_stream_rate.py 2>&1 | tail -1 | tee -a $OUT/r05d_owner_stream_rate.txt
for v in maxr14 keeplast; do
  FWGPU_LIBRARY=$R/build/variants/libfwgpu_$v.so timeout 600 python3 bench.py --long --long-passes 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['logloss_after_examples']
print('$v:', round(d['value']), 'final', [round(x,4) for x in d['final_logloss_passes']], 'curve', {k: round(v[0],4) for k,v in c.items() if int(k) % 4194304 == 0})"
done | tee $OUT/r05d_long_kept_rows.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/r05d_bench_driver_shape.json 2> $OUT/r05d_bench_driver_shape.err; tail -c 3000 $OUT/r05d_bench_driver_shape.json
