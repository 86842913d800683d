# A/B of the hot LR entry route (kernels.hip hot_lr_flush): FWGPU_HOT_LR_EVERY = 0 (plain read-modify-writes) / N examples between flushes;
# headline config C, config B (10 fields, k = 4, 22-bit tables), LR-only.  Output: examples/s and hold-out log-loss.
for e in ${HOT_LR_SET:-0 1 4 32}; do
echo "== FWGPU_HOT_LR_EVERY=$e"
FWGPU_HOT_LR_EVERY=$e timeout 300 python3 bench.py --no-other-modes --no-traffic --no-cpu-baseline --steps ${C_STEPS:-25} 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('C', d['value'], d['final_logloss'])"
FWGPU_HOT_LR_EVERY=$e timeout 300 python3 bench.py --no-other-modes --no-traffic --no-cpu-baseline --fields 10 --k 4 --bits 22 --ffm-bits 22 --mean-extra 0 --zipf 1.1 --ids 100000 --p-weighted 0 --batch 65536 --steps 60 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['value'], d['final_logloss'])"
FWGPU_HOT_LR_EVERY=$e timeout 300 python3 bench.py --no-other-modes --no-traffic --no-cpu-baseline --k 0 --steps 60 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('LR-only', d['value'], d['final_logloss'])"
done
