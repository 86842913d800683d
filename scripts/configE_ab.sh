# config E (BASELINE configs[4]: F=30, k=16, 2x256 ReLU head): thinned dense steps A/B
# FWGPU_NN_UPD_EVERY = m: dense steps for every m-th example of a workgroup; FWGPU_NN_PLAIN = 1: forward / input gradients read W through L2
run() {
echo "== $*"
for rep in 1 2; do
env "$@" timeout 600 python3 bench.py --k 16 --nn-layers 2 --batch 8192 --steps ${E_STEPS:-48} --warmup 4 --no-traffic --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('E', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],3), {k: round(v,4) for k,v in d['logloss_after_examples'].items()})"
done
}
run FWGPU_NN_UPD_EVERY=1
run FWGPU_NN_UPD_EVERY=4
run FWGPU_NN_UPD_EVERY=8
run FWGPU_NN_UPD_EVERY=8 FWGPU_NN_PLAIN=1
run FWGPU_NN_UPD_EVERY=16 FWGPU_NN_PLAIN=1
run FWGPU_NN_UPD_EVERY=32 FWGPU_NN_PLAIN=1
