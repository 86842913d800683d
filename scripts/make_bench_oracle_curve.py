#!/usr/bin/env python3
"""Hold-out log-loss of the CPU oracle (oracle/fw_oracle.c, the reference's algorithm) on bench.py's fixed-seed config-C stream at SURVEY 8d's
protocol length: 16 Mi training examples (256 steps of 65 536), hold-out = 262 144 examples of the same stream's tail that are predicted and
never learned (main.rs:184-185, 238-241; loss as benchmark/calc_loss.py:5-25).

  python scripts/make_bench_oracle_curve.py seq            the reference's single thread (main.rs:213-270): deterministic
  python scripts/make_bench_oracle_curve.py hog16 <run>    the reference's hogwild mode, 16 threads (main.rs:189-194, hogwild.rs:89-103);
                                                           racy by definition: <run> only names the output, every run interleaves differently

Checkpoints: after every step up to 52 (bench.py's default and driver shapes end at 25 / 52 steps), then every 4 steps up to 256, every 16 beyond.
Writes tests/golden/bench_oracle_curve_<mode>[_r<run>].json (data: bench.py and the tests read the numbers, never the oracle).
Environment: CURVE_STEPS (256; 1024 = round 6's 64 Mi-example protocol), CURVE_FAMILY=2 (the second stream family at config C's size: another teacher seed
4242, Zipf 1.3 ids, 5 % of the labels flipped -- bench.py --family 2; file names get `_fam2`), CURVE_OUT (write somewhere else first: a run of hours
should not sit half-written among the committed curves).
The 8 192-example prefix of the hold-out (round 1-4's yardstick) is kept beside the 262 144-example loss.
Sequential: ~1 h on one core (+ the hold-out passes on `--pred-threads` threads)."""
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

mode = sys.argv[1] if len(sys.argv) > 1 else "seq"
run = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_steps = int(os.environ.get("CURVE_STEPS", "256"))
pred_threads = int(os.environ.get("CURVE_PRED_THREADS", "4"))
gen_threads = int(os.environ.get("CURVE_GEN_THREADS", "2"))
B = 65536
HOLDOUT = 262144
sys.argv = [sys.argv[0]]


family = int(os.environ.get("CURVE_FAMILY", "1"))


class A:  # bench.py's defaults for config C
    fields, k, bits, ffm_bits = 30, 8, 28, 28
    mean_extra, zipf, ids, p_weighted, seed, holdout = 5.67, 1.05, 10_000_000, 0.1, 20240612, HOLDOUT
    label_flip = 0.0


if family == 2:
    bench.apply_family(A, 2)


args = A()
F = args.fields
nthreads = {"seq": 1, "hog16": 16}[mode]
ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=bench.LR, ffm_learning_rate=bench.LR, power_t=bench.POWER_T,
                       ffm_power_t=bench.POWER_T, init_acc_gradient=bench.INIT_ACC, ffm_init_acc_gradient=bench.INIT_ACC,
                       bit_precision=args.bits, num_combos=F + 1, ffm_k=args.k, ffm_bit_precision=args.ffm_bits, ffm_num_fields=F)
ots = fwo.TranslatorSpec([([(i, False)], 1.0) for i in range(F)], [[(i, False)] for i in range(F)], True, args.bits, args.k, args.ffm_bits)
try:
    om = fwo.Model(ocfg, native=True)
except Exception:
    om = fwo.Model(ocfg, native=False)
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, HOLDOUT, threads=gen_threads)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
name = "bench_oracle_curve_" + ("fam2_" if family == 2 else "") + mode + (f"_r{run}" if mode != "seq" else "") + ".json"
path = os.environ.get("CURVE_OUT") or os.path.join(ROOT, "tests", "golden", name)
out = {"what": ("CPU oracle, the reference's single thread (main.rs:213-270)" if mode == "seq" else
                f"CPU oracle in the reference's hogwild mode, {nthreads} threads on {os.cpu_count()} host cores (hogwild.rs:89-103), run {run}")
               + ": hold-out log-loss after N training examples of bench.py's default stream",
       "config": {k: getattr(args, k) for k in ("fields", "k", "bits", "ffm_bits", "mean_extra", "zipf", "ids", "p_weighted", "seed", "holdout") + (("label_flip",) if args.label_flip else ())},
       "hyper": {"lr": bench.LR, "power_t": bench.POWER_T, "init_acc": bench.INIT_ACC}, "threads": nthreads,
       "holdout_prior_logloss": bench.logloss(np.full(len(hy), float(np.mean(hy == 1))), hy),
       "examples": [], "logloss": [], "logloss_first_8192": [], "train_seconds": 0.0}
t0 = time.time()
skip_before = int(os.environ.get("CURVE_SKIP_BEFORE", "0"))  # no checkpoints before this step (the sequential curve's first 256 steps are committed already: deterministic)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
ahead = ThreadPoolExecutor(max_workers=1)  # the next step's records are generated while this one trains (ctypes releases the GIL)
nxt = ahead.submit(bench.gen_records, fw, args, 0, B, gen_threads)
for s in range(n_steps):
    recs, off = nxt.result()
    if s + 1 < n_steps:
        nxt = ahead.submit(bench.gen_records, fw, args, (s + 1) * B, B, gen_threads)
    dt, _ = om.run_stream(ots, recs, off, holdout_after=0, nthreads=nthreads, want_preds=False)
    out["train_seconds"] += dt
    if s + 1 < skip_before:
        continue
    if s + 1 <= 52 or ((s + 1) % 4 == 0 and s + 1 <= 256) or (s + 1) % 16 == 0 or s + 1 == n_steps:
        p = om.predict_stream(ots, hrecs, hoff, nthreads=pred_threads)
        out["examples"].append((s + 1) * B)
        out["logloss"].append(round(bench.logloss(p, hy), 6))
        out["logloss_first_8192"].append(round(bench.logloss(p[:8192], hy[:8192]), 6))
        print(s + 1, out["logloss"][-1], out["logloss_first_8192"][-1], f"{time.time() - t0:.0f}s", flush=True)
        with open(path + ".tmp", "w") as f:
            json.dump(out, f, indent=1)
        os.replace(path + ".tmp", path)
