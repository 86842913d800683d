# Kept rows per wave in the config-C learn kernel after kp_fresh() freed registers: 14 (shipped) against 16 / 18 (build/variants/libfwgpu_maxrN.so =
# kernels.hip with -DFW_MAXR_WIN=N).  Three interleaved passes of the default bench: examples/s, hold-out after 1.64 M examples, roofline.frac.
V=$PWD/build/variants
for rep in 1 2 3; do for L in "" $V/libfwgpu_maxr16.so $V/libfwgpu_maxr18.so; do
  FWGPU_LIBRARY=$L timeout 300 python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename ${L:-shipped_14})', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4))"
done; done
