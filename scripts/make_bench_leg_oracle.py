#!/usr/bin/env python3
"""CPU-oracle reference values for bench.py's side legs, on the legs' own streams (committed as data under tests/golden/, read by bench.py as numbers):

  python scripts/make_bench_leg_oracle.py configb   BASELINE configs[1]: 10 fields, k = 4, 22-bit tables, micro-batch 4096, lr 0.1 / power_t 0.5;
                                                     220 steps (bench.py's config_b leg: 20 warm-up + 200 timed), sequential AND 16-thread hogwild (3 runs)
  python scripts/make_bench_leg_oracle.py confige   BASELINE configs[4]: config C's stream with k = 16 + the 2 x 256 ReLU head, 28 steps of 8192 (the config_e leg),
                                                     sequential (the reference's single thread)

Hold-out: 65 536 examples of the stream's tail, predicted and never learned (main.rs:238-241).  Writes tests/golden/bench_oracle_curve_<leg>_<mode>.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fwumious_wabbit_amd as fw  # noqa: E402
from oracle import fwo  # noqa: E402

leg = sys.argv[1]
sys.argv = [sys.argv[0]]
PT = int(os.environ.get("CURVE_PRED_THREADS", "8"))


class A:
    pass


args = A()
if leg == "configb":
    args.fields, args.k, args.bits, args.ffm_bits = 10, 4, 22, 22
    args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed, args.holdout = 0.0, 1.1, 100_000, 0.0, 20240611, 65536
    args.lr, args.power_t, args.nn_layers, args.nn_width = 0.1, 0.5, 0, 256
    B, n_steps, every, modes = 4096, 220, 20, [("seq", 1, 1), ("hog16", 16, 1), ("hog16", 16, 2), ("hog16", 16, 3)]
elif leg == "confige":
    args.fields, args.k, args.bits, args.ffm_bits = 30, 16, 28, 28
    args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed, args.holdout = 5.67, 1.05, 10_000_000, 0.1, 20240612, 65536
    args.lr, args.power_t, args.nn_layers, args.nn_width = bench.LR, bench.POWER_T, 2, 256
    B, n_steps, every, modes = 8192, 28, 4, [("seq", 1, 1)]
else:
    raise SystemExit(__doc__)
F = args.fields
ots = fwo.TranslatorSpec([([(i, False)], 1.0) for i in range(F)], [[(i, False)] for i in range(F)], True, args.bits, args.k, args.ffm_bits)
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, args.holdout)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
for mode, nthreads, run in modes:
    ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=args.lr, ffm_learning_rate=args.lr, power_t=args.power_t, ffm_power_t=args.power_t,
                           init_acc_gradient=bench.INIT_ACC, ffm_init_acc_gradient=bench.INIT_ACC, bit_precision=args.bits, num_combos=F + 1, ffm_k=args.k,
                           ffm_bit_precision=args.ffm_bits, ffm_num_fields=F)
    nn = fwo.make_nn_config([(args.nn_width, "relu", "hu")] * args.nn_layers, "one", bench.NN_LR, bench.NN_POWER_T, bench.NN_INIT_ACC) if args.nn_layers else None
    om = fwo.Model(ocfg, native=True, nn=nn)
    out = {"what": f"CPU oracle, {'the reference single thread' if nthreads == 1 else f'hogwild mode, {nthreads} threads on {os.cpu_count()} host cores, run {run}'}: hold-out log-loss after N "
                   f"training examples of bench.py's {leg} leg",
           "config": {k: getattr(args, k) for k in ("fields", "k", "bits", "ffm_bits", "mean_extra", "zipf", "ids", "p_weighted", "seed", "holdout", "nn_layers", "nn_width")},
           "hyper": {"lr": args.lr, "power_t": args.power_t, "init_acc": bench.INIT_ACC}, "threads": nthreads, "examples": [], "logloss": [], "train_seconds": 0.0}
    for s in range(n_steps):
        recs, off = bench.gen_records(fw, args, s * B, B)
        dt, _ = om.run_stream(ots, recs, off, holdout_after=0, nthreads=nthreads, want_preds=False)
        out["train_seconds"] += dt
        if (s + 1) % every == 0 or s + 1 == n_steps:
            p = om.predict_stream(ots, hrecs, hoff, nthreads=PT)
            out["examples"].append((s + 1) * B)
            out["logloss"].append(round(bench.logloss(p, hy), 6))
            print(leg, mode, run, s + 1, out["logloss"][-1], flush=True)
    om.close()
    name = f"bench_oracle_curve_{leg}_{mode}" + (f"_r{run}" if nthreads > 1 else "") + ".json"
    with open(os.path.join(ROOT, "tests", "golden", name), "w") as f:
        json.dump(out, f, indent=1)
