#!/bin/bash
# equal-examples comparison of the update rules on one GPU (786 432 examples of the config-C stream, same hold-out tail):
# hogwild (per-occurrence steps, concurrent), synchronous micro-batch (per-occurrence steps, frozen weights per batch),
# row-sparse buckets (ONE summed-gradient step per row and batch).  usage on the GPU box: bash scripts/sparse_quality.sh
set -u
OUT=gpurun_out/sparse_quality.txt
mkdir -p gpurun_out
: > $OUT
run() {  # label, args...
    local label=$1; shift
    timeout 900 python3 bench.py "$@" --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$label', 'examples', d['steps']*d['config']['global_batch'], 'batch', d['config']['global_batch'], 'ex/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'holdout_logloss %.4f' % d['final_logloss'])" >> $OUT
}
run hogwild_16384 --batch 16384 --steps 48 --warmup 0
run sync_2048 --sync --batch 2048 --steps 384 --warmup 0
run sparse_512 --force-dist --dp-mode sparse --batch 512 --steps 1536 --warmup 0
run sparse_2048 --force-dist --dp-mode sparse --batch 2048 --steps 384 --warmup 0
run sparse_4096 --force-dist --dp-mode sparse --batch 4096 --steps 192 --warmup 0
run sparse_8192 --force-dist --dp-mode sparse --batch 8192 --steps 96 --warmup 0
cat $OUT
