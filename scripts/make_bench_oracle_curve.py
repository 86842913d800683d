#!/usr/bin/env python3
"""Hold-out log-loss of the SEQUENTIAL CPU oracle (the reference's single-thread algorithm, oracle/fw_oracle.c) on bench.py's fixed-seed
config-C stream, after every 65 536 examples: the reference point bench.py prints next to the GPU's hogwild loss (`oracle_logloss_after_examples`).
Writes tests/golden/bench_oracle_curve.json (data; bench.py reads the numbers, never the oracle).  ~15 minutes on one core.
usage: python scripts/make_bench_oracle_curve.py [n_steps=52]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (hyper-parameters and the stream generator are bench.py's own)
import fwumious_wabbit_amd as fw  # noqa: E402
from oracle import fwo  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 52
B = 65536
sys.argv = [sys.argv[0]]


class A:  # bench.py's defaults for config C
    fields, k, bits, ffm_bits = 30, 8, 28, 28
    mean_extra, zipf, ids, p_weighted, seed, holdout = 5.67, 1.05, 10_000_000, 0.1, 20240612, 8192


args = A()
F = args.fields
ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=bench.LR, ffm_learning_rate=bench.LR, power_t=bench.POWER_T,
                       ffm_power_t=bench.POWER_T, init_acc_gradient=bench.INIT_ACC, ffm_init_acc_gradient=bench.INIT_ACC,
                       bit_precision=args.bits, num_combos=F + 1, ffm_k=args.k, ffm_bit_precision=args.ffm_bits, ffm_num_fields=F)
ots = fwo.TranslatorSpec([([(i, False)], 1.0) for i in range(F)], [[(i, False)] for i in range(F)], True, args.bits, args.k, args.ffm_bits)
try:
    om = fwo.Model(ocfg, native=True)
except Exception:
    om = fwo.Model(ocfg, native=False)
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, args.holdout)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
out = {"what": "sequential CPU oracle, hold-out log-loss after N training examples of bench.py's default stream",
       "config": {k: getattr(args, k) for k in ("fields", "k", "bits", "ffm_bits", "mean_extra", "zipf", "ids", "p_weighted", "seed", "holdout")},
       "hyper": {"lr": bench.LR, "power_t": bench.POWER_T, "init_acc": bench.INIT_ACC}, "examples": [], "logloss": []}
t0 = time.time()
for s in range(n_steps):
    recs, off = bench.gen_records(fw, args, s * B, B)
    om.run_stream(ots, recs, off, holdout_after=0, nthreads=1, want_preds=False)
    _, p = om.run_stream(ots, hrecs, hoff, holdout_after=1, nthreads=1)
    out["examples"].append((s + 1) * B)
    out["logloss"].append(round(bench.logloss(p, hy), 6))
    print(s + 1, out["logloss"][-1], f"{time.time() - t0:.0f}s", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "bench_oracle_curve.json"), "w") as f:
        json.dump(out, f, indent=1)
