#!/usr/bin/env python3
"""An EMULATION of the device's concurrent mode on the CPU oracle (oracle/fw_oracle.c fwo_learn_window_emulation; analysis only -- not a reference code path, not a parity yardstick):
bench.py's stream cut into windows of WINDOW examples (the examples the device has in flight), every example of a window scored against the tables of the window's start, the steps
applied in order with  flags bit 0: FFM weights written back as start value - step (the device's rows kept from the gather: last writer wins)  and  bit 1: the same for the accumulators
(lossy write-through stores; without it every g^2 is counted: store policy 4).  The question it answers: does THIS -- staleness of one window plus last-writer-wins on
the rows many examples hold -- produce the curve the GPU shows on this stream (a minimum at 5-8 M examples well below the sequential reference, then a rise onto a plateau), and which of the
two ingredients does?   usage: python scripts/oracle_window_emulation.py WINDOW FLAGS [steps=256]   -> profiles/r06_window_emulation_w<WINDOW>_f<FLAGS>.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
window, flags = int(sys.argv[1]), int(sys.argv[2])
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import fwumious_wabbit_amd as fw  # noqa: E402
from oracle import fwo  # noqa: E402

B, HOLDOUT = 65536, 262144


class A:
    fields, k, bits, ffm_bits = 30, 8, 28, 28
    mean_extra, zipf, ids, p_weighted, seed, holdout, label_flip = 5.67, 1.05, 10_000_000, 0.1, 20240612, HOLDOUT, 0.0


args = A()
F = args.fields
ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=bench.LR, ffm_learning_rate=bench.LR, power_t=bench.POWER_T, ffm_power_t=bench.POWER_T,
                       init_acc_gradient=bench.INIT_ACC, ffm_init_acc_gradient=bench.INIT_ACC, bit_precision=args.bits, num_combos=F + 1, ffm_k=args.k,
                       ffm_bit_precision=args.ffm_bits, ffm_num_fields=F)
ots = fwo.TranslatorSpec([([(i, False)], 1.0) for i in range(F)], [[(i, False)] for i in range(F)], True, args.bits, args.k, args.ffm_bits)
om = fwo.Model(ocfg, native=True)
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, HOLDOUT, threads=1)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
path = os.path.join(ROOT, "profiles", f"r06_window_emulation_w{window}_f{flags}.json")
out = {"what": f"CPU emulation of the device's concurrent mode: windows of {window} examples, flags {flags} (bit 0: FFM weights last-writer-wins over the window; bit 1: accumulators too); "
               "hold-out log-loss after N examples of bench.py's stream (oracle/fw_oracle.c fwo_learn_window_emulation; analysis only)",
       "window": window, "flags": flags, "examples": [], "logloss": []}
t0 = time.time()
for s in range(n_steps):
    recs, off = bench.gen_records(fw, args, s * B, B, threads=1)
    om.learn_window_emulation(ots, recs, off, window, flags)
    if (s + 1) % 16 == 0 or s + 1 == n_steps:
        p = om.predict_stream(ots, hrecs, hoff, nthreads=2)
        out["examples"].append((s + 1) * B)
        out["logloss"].append(round(bench.logloss(p, hy), 6))
        print(s + 1, out["logloss"][-1], f"{time.time() - t0:.0f}s", flush=True)
        with open(path + ".tmp", "w") as f:
            json.dump(out, f, indent=1)
        os.replace(path + ".tmp", path)
