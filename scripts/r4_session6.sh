#!/bin/bash
OUT=gpurun_out/r4f; mkdir -p $OUT
echo "== targeted"; timeout 900 python -m pytest tests/test_gpu_dist_procs.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "True or beyond_4096" 2>&1 | tail -30 | tee $OUT/targeted.txt
echo "== full suite"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/gputest.txt
echo "== dist N=1 vs plain"; for i in 1 2; do
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain      ', round(d['value']), round(d['final_logloss'],4), d.get('oracle_final_logloss'))"
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --force-dist --no-other-modes 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist ', round(d['value']), round(d['final_logloss'],4), d.get('rccl_ranks'))"
done 2>&1 | tee $OUT/dist1.txt
