for v in 1 0; do
echo "=== ticks, FWGPU_NN_V2=$v"
FWGPU_NN_V2=$v FWGPU_LIBRARY=build/variants/libfwgpu_ticks.so K=16 NN_LAYERS=2 B=8192 THREADS=$([ $v = 1 ] && echo 512 || echo 1024) WGS=0 QUICK=1 LUTG=1 timeout 300 python scripts/perf_probe.py 2>&1 | grep "update=True"
done
