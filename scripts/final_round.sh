#!/bin/bash
# End-of-round evidence run: the full -m gpu suite with the HEAD hash, then scripts/profile_round.sh (default bench line with cpu_baseline, live traffic
# and the config-E leg; rocprofv3 kernel trace + the two PMC passes; one-rank RCCL legs; config E).  usage: bash scripts/final_round.sh r04 <hash>
TAG=${1:-r04}; HASH=${2:-unknown}
OUT=gpurun_out; mkdir -p $OUT
echo "HEAD $HASH" > $OUT/${TAG}_gputest.log
timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 >> $OUT/${TAG}_gputest.log
tail -4 $OUT/${TAG}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_gputest.log
bash scripts/profile_round.sh $TAG
timeout 600 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_shape.json 2> $OUT/${TAG}_bench_driver_shape.err
tail -c 400 $OUT/${TAG}_bench_driver_shape.json
