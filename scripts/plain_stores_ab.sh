# The v2 kernel's row STORES plain (write-back L2, flushed at the end of the launch) instead of device-scope write-through: ps = stores only
# (-DFW_PLAIN_STORES_W=1 -DFW_PLAIN_STORES_ACC=1), pall = loads and stores (the same + -DFW_PLAIN_GATHER=1 -DFW_PLAIN_UPD_LOADS=1).  Another XCD then sees a row when the line leaves this
# XCD's L2 (~20 us at this traffic).  Three interleaved passes of the default bench, with the PMC traffic of the run.
V=$PWD/build/variants
for rep in 1 2 3; do for L in "" $V/libfwgpu_ps.so $V/libfwgpu_pall.so; do
  FWGPU_LIBRARY=$L timeout 400 python3 bench.py --no-cpu-baseline ${TRAFFIC:---no-traffic} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename ${L:-shipped})', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4), d['roofline'].get('traffic'), d['logloss_after_examples'])"
done; done
