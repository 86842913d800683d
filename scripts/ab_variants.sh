#!/bin/bash
# Interleaved A/B of library variants (build/variants/libfwgpu_NAME.so): PASSES passes over the list, one process per run,
# so that box drift (clocks, co-tenants) hits every variant alike.  usage: scripts/ab_variants.sh "spec" name1 name2 ...
SPEC=$1; shift
PASSES=${PASSES:-3}
for p in $(seq $PASSES); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset FWGPU_LIBRARY; else export FWGPU_LIBRARY=$PWD/build/variants/libfwgpu_$v.so; fi
    r=$(ROUNDS=2 timeout 300 python3 scripts/ab_probe.py "$SPEC" 2>&1 | grep "learn ms" | sed 's/.*median \([0-9.]*\) min.*/\1/' | tr '\n' ' ')
    echo "pass $p $v: $r"
  done
done
