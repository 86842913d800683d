# (1) AdaGrad LUT as a compile-time LDS pointer in the v2 kernel (ds_read instead of flat_load: no vmcnt(0) in front of every lookup) against the
# build before it (build/variants/libfwgpu_kpfresh.so) and the one before kp_fresh / the two translation units (libfwgpu_before_split.so): default bench,
# three interleaved passes.  (2) the one-rank RCCL legs (sparse / sharded steps) of the three builds.
V=$PWD/build/variants
for rep in 1 2 3; do for L in $V/libfwgpu_before_split.so $V/libfwgpu_kpfresh.so ""; do
  FWGPU_LIBRARY=$L timeout 300 python3 bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config C', '$(basename ${L:-shipped_lds_lut})', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],4))"
done; done
for L in $V/libfwgpu_before_split.so $V/libfwgpu_kpfresh.so ""; do
  FWGPU_LIBRARY=$L MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python3 bench.py --force-dist --no-cpu-baseline --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['dp_modes']; print('dist1', '$(basename ${L:-shipped_lds_lut})', round(d['value']), 'sparse', round(m['sparse']['value']), 'sharded', round(m['sharded']['value']))"
done
