#!/usr/bin/env python3
"""BASELINE configs[1] (10 fields, k = 4, 22-bit tables, micro-batch 4096, AdagradLUT lr 0.1 / power_t 0.5; SURVEY 8d "Config B"): the time-to-loss table of its
launch shape.  One child run of bench.py per row -- the side leg's own command (bench.config_b_leg: 20 + 200 launches, 901 120 examples learned,
65 536-example hold-out) with the workgroup size, the cap on the examples in flight and the launch size varied:

  threads     workgroup size (the shipped choice: 512; a 10-feature example occupies 10 of its 8 waves' rows)
  in flight   fwgpu_set_max_in_flight (0: what the device holds: two to eight workgroups per CU depending on the size)
  batch       examples per launch (4096 = the config's micro-batch; 32 768 = eight micro-batches per launch: what a queue of micro-batches behind one
              persistent launch would give at best -- the launch gap removed, the in-flight count unchanged)

Columns: examples/s, hold-out log-loss after the leg's 901 120 examples (sequential oracle 0.593945, 16-thread hogwild oracle 0.59393-0.59397:
tests/golden/bench_oracle_curve_configb_*.json), seconds and examples to the hold-out loss 0.5945.  usage: python scripts/configB_table.py [reps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
BASE = ["--fields", "10", "--k", "4", "--bits", "22", "--ffm-bits", "22", "--mean-extra", "0", "--zipf", "1.1", "--ids", "100000", "--p-weighted", "0", "--seed", "20240611",
        "--lr", "0.1", "--power-t", "0.5", "--holdout", "65536", "--no-cpu-baseline", "--no-traffic", "--no-config-e", "--no-config-b", "--target-logloss", "0.5945"]
ROWS = [(512, 0, 4096), (256, 0, 4096), (128, 0, 4096), (128, 512, 4096), (128, 1024, 4096), (256, 512, 4096), (256, 1024, 4096), (512, 256, 4096),
        (512, 0, 32768), (256, 512, 32768), (128, 512, 32768), (128, 1024, 32768)]
print(f"{'threads':>7s} {'in flight':>9s} {'batch':>6s} {'M ex/s':>8s} {'hold-out':>9s} {'s to 0.5945':>11s} {'examples to it':>14s}")
for th, inf, b in ROWS:
    for _ in range(reps):
        steps, warm = (200, 20) if b == 4096 else (25, 2)  # (the same ~900 k examples)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + BASE + ["--batch", str(b), "--steps", str(steps), "--warmup", str(warm), "--curve-every", str(max(1, steps // 25))]
        if th != 512:
            cmd += ["--threads", str(th)]
        if inf:
            cmd += ["--max-in-flight", str(inf)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode or not lines:
            print(f"{th:7d} {inf:9d} {b:6d}  failed: {(p.stderr or p.stdout)[-200:]!r}", flush=True)
            continue
        d = json.loads(lines[-1])
        s2l = d.get("seconds_to_logloss") or {}
        sec, ex = s2l.get("seconds"), s2l.get("examples")
        print(f"{th:7d} {inf or 'auto':>9} {b:6d} {d['value'] / 1e6:8.2f} {d['final_logloss']:9.5f} {(f'{sec:.4f}' if sec is not None else 'not reached'):>11s} {(str(ex) if ex is not None else '-'):>14s}", flush=True)
