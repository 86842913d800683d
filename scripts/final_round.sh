#!/bin/bash
# End-of-round evidence run (on the GPU box, from the repo root): full GPU suite with the HEAD hash, round profile (bench, rocprofv3 trace + PMC,
# one-rank RCCL legs, config E), the hold-out spread table the statistical tolerances rest on, config B / LR-only rates.
# usage: bash scripts/final_round.sh r03b <git-hash>
TAG=${1:-r03}; HASH=${2:-unknown}
OUT=gpurun_out
echo "# pytest -m gpu at $HASH" > $OUT/${TAG}_gputest.log
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 >> $OUT/${TAG}_gputest.log
bash scripts/profile_round.sh $TAG > $OUT/${TAG}_profile_round.log 2>&1
python scripts/holdout_spread.py 8 > $OUT/${TAG}_holdout_spread.txt 2>&1
HOT_LR_SET="0 1" C_STEPS=20 bash scripts/hot_lr_ab.sh > $OUT/${TAG}_hot_lr_ab.txt 2>&1
bash scripts/group_exp.sh > $OUT/${TAG}_group_exp.txt 2>&1
tail -3 $OUT/${TAG}_gputest.log
