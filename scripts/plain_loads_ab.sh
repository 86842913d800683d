# The v2 kernel's table LOADS through L2 (plain) instead of device scope in updating launches; stores stay device-scope write-through.
# pg = the gather's reads of w only (-DFW_PLAIN_GATHER=1), pa = + the update phase's reads of acc / w (-DFW_PLAIN_UPD_LOADS=1).  A stale L2 line lives
# ~20 us at this traffic, an example ~150 us: what a plain load can miss is less than what a row kept from the gather already ignores.
V=$PWD/build/variants
for rep in 1 2 3; do for L in "" $V/libfwgpu_pg.so $V/libfwgpu_pa.so; do
  FWGPU_LIBRARY=$L timeout 300 python3 bench.py --no-cpu-baseline ${TRAFFIC:---no-traffic} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename ${L:-shipped})', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4), d['roofline'].get('traffic'), d['logloss_after_examples'])"
done; done
