#!/bin/bash
# Interleaved A/B of the FFM row store policy, its write-back interval, the record prefetch and library variants through bench.py
# (one process per run; env switches FWGPU_STORE_POLICY / FWGPU_WB_FLUSH_EVERY / FWGPU_PREFETCH / FWGPU_LDS_KEEP / FWGPU_LIBRARY).  Round 4's tables
# (profiles/r04a_policy_ab.txt, r04b_*, r04c_*) were produced by lists of this form.
# usage: scripts/store_policy_ab.sh [passes=2] [steps=20] -- "name|ENV=val ENV=val" ...
#   e.g. scripts/store_policy_ab.sh 2 20 -- "p1_f128|" "p2_f128|FWGPU_STORE_POLICY=2" "p1_f0|FWGPU_WB_FLUSH_EVERY=0" "m16|FWGPU_LIBRARY=$PWD/build/variants/libfwgpu_m16.so"
PASSES=${1:-2}; STEPS=${2:-20}; shift; shift; [ "$1" = "--" ] && shift
for p in $(seq $PASSES); do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    env $envs timeout 900 python3 bench.py --steps $STEPS --warmup 5 --curve-every 30 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pass $p $name:', round(d['value']), round(d['final_logloss'],4), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:round(v,4) for k,v in d.get('logloss_after_examples',{}).items()})"
  done
done
