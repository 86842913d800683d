# config E (BASELINE configs[4]: F=30, k=16, 2x256 ReLU head): deferred dense steps A/B
# FWGPU_NN_DEFER = examples pending per workgroup, FWGPU_NN_DEFER_SUM = summed gradient (1) / step per example (0), FWGPU_NN_PLAIN = L2 reads
run() {
echo "== $*"
env "$@" timeout 600 python3 bench.py --k 16 --nn-layers 2 --batch 8192 --steps ${E_STEPS:-24} --warmup 4 --no-traffic --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('E', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],3), {k: round(v,4) for k,v in d['logloss_after_examples'].items()})"
}
run FWGPU_NN_DEFER=0
run FWGPU_NN_DEFER=8 FWGPU_NN_DEFER_SUM=1
run FWGPU_NN_DEFER=8 FWGPU_NN_DEFER_SUM=1 FWGPU_NN_PLAIN=1
run FWGPU_NN_DEFER=4 FWGPU_NN_DEFER_SUM=1 FWGPU_NN_PLAIN=1
run FWGPU_NN_DEFER=8 FWGPU_NN_DEFER_SUM=0 FWGPU_NN_PLAIN=1
run FWGPU_NN_DEFER=0 FWGPU_NN_PLAIN=1
