# Launch shape for small examples (config B: 10 features per example): workgroup size x workgroups per CU, examples/s and hold-out log-loss.
B="--no-other-modes --no-traffic --no-cpu-baseline --fields 10 --k 4 --bits 22 --ffm-bits 22 --mean-extra 0 --zipf 1.1 --ids 100000 --p-weighted 0 --batch 65536 --steps 60"
for shape in "0 0" "256 4" "128 8" "64 8" "64 16" "64 32" "128 16"; do
set -- $shape
timeout 300 python3 bench.py $B --threads $1 --wgs-per-cu $2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads $1 wgs/CU $2:', round(d['value']/1e6,2), 'M ex/s, hold-out', round(d['final_logloss'],4))"
done
