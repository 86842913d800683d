#!/bin/bash
# Interleaved A/B of library variants through bench.py.  usage: scripts/ab_bench.sh "bench args" name1 name2 ...  (default = in-tree build)
ARGS=$1; shift
PASSES=${PASSES:-2}
for p in $(seq $PASSES); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset FWGPU_LIBRARY; else export FWGPU_LIBRARY=$PWD/build/variants/libfwgpu_$v.so; fi
    timeout 400 python3 bench.py $ARGS --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pass $p $v:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3))"
  done
done
