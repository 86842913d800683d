# config E (BASELINE configs[4]: F=30, k=16, 2x256 ReLU head) and the same FFM without the head, bench lines in short
run() {
echo "== $*"
timeout 600 python3 bench.py "$@" --batch 8192 --steps ${E_STEPS:-24} --warmup 4 --no-traffic --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('E', round(d['value']), round(d['final_logloss'],4), round(d['roofline']['frac'],3), {k: round(v,4) for k,v in d['logloss_after_examples'].items()})"
}
run --k 16 --nn-layers 2
run --k 16
