# After kp_fresh() (argument block read where it is used: 200 -> 85 spilled scalars in the config-C kernel): the failing 728 B layout again, ranks NOT
# ordered; then A/B of the default bench and of config E against the library from before the change (build/variants/libfwgpu_before_split.so).
run() { name=$1; shift; n=$1; shift; ok=0; bad=0; fault=0; for i in 1 2 3 4 5 6 7 8; do out=$(env "$@" timeout 300 python3 scripts/group_repro.py $n 2048 8 2>&1 | grep -E "final|fault" | tail -1); if echo "$out" | grep -q fault; then fault=$((fault+1)); elif [ "$out" = "$(cat /tmp/ref_$n)" ]; then ok=$((ok+1)); else bad=$((bad+1)); fi; done; echo "$name n=$n: exact $ok wrong $bad fault $fault"; }
timeout 300 python3 scripts/group_repro.py 4 2048 8 2>&1 | grep final | tail -1 > /tmp/ref_4
V=$PWD/build/variants
run "728 B layout with kp_fresh(), unordered" 4 FWGPU_LIBRARY=$V/libfwgpu_kp0ncfresh.so FWGPU_GROUP_CONCURRENT=local
run "shipped library with kp_fresh(), unordered" 4 FWGPU_GROUP_CONCURRENT=local
for rep in 1 2 3; do for L in $V/libfwgpu_before_split.so ""; do
  FWGPU_LIBRARY=$L timeout 300 python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config C', '${L:+before}' or 'kp_fresh', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4))"
done; done
for rep in 1 2; do for L in $V/libfwgpu_before_split.so ""; do
  FWGPU_LIBRARY=$L timeout 300 python3 bench.py --k 16 --nn-layers 2 --batch 8192 --steps 40 --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config E', '${L:+before}' or 'kp_fresh', round(d['value']), round(d['final_logloss'],4))"
done; done
