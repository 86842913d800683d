# hold-out loss at a longer horizon (150 steps x 65 536 = 9.8 M examples) for the resident-row variants (build/variants/libfwgpu_maxrNwin.so)
for v in ${LH_SET:-0 2 6}; do
  for rep in 1 2; do
    echo -n "maxr$v rep $rep: "
    FWGPU_LIBRARY=$PWD/build/variants/libfwgpu_maxr${v}win.so timeout 900 python3 bench.py --steps 150 --warmup 4 --no-traffic --no-cpu-baseline --curve-every 30 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['final_logloss'],4), {k: round(v,4) for k,v in d['logloss_after_examples'].items()})"
  done
done
