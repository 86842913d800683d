#!/usr/bin/env python3
"""The hold-out protocol of `bench.py --long` for a LIST of launch settings over ONE device-resident copy of the stream: every variant starts from freshly
initialised weights, learns the same batches, and is scored on the same hold-out tail every `--every` steps -- the two-sided yardstick of DESIGN 6
(GPU curve against both oracle modes' curves, tests/golden/bench_oracle_curve_*.json).  Generating and uploading the records is what takes minutes
(64 Mi examples: 116 GB); a variant is seconds.

  python scripts/long_variants.py --steps 1024 --family 1 --out gpurun_out/x.json  "name:policy=4,theta=0.05,m=3,inflight=512"  "shipped:"  ...

Variant keys: policy (store policy 0-4), theta / m (hot-row threshold, log2 of the sampling: options 9 / 10), inflight (fwgpu_set_max_in_flight),
wb (write-back interval, option 6), keep (rows parked in LDS, option 8), lrthin (store policy 4 on hot LR entries, option 12), kept (0: no rows kept from the gather, option 13), reps (passes of this variant, default 1), lib (path of a variant libfwgpu.so:
run in a child process -- NOT supported here: use FWGPU_LIBRARY on the whole script).
Prints one JSON document: per variant examples/s (wall clock around the training launches, checkpoints excluded by events) and the hold-out curve."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--every", type=int, default=16)
    ap.add_argument("--family", type=int, default=1)
    ap.add_argument("--holdout", type=int, default=262144)
    ap.add_argument("--out", default=None)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()

    import torch

    import fwumious_wabbit_amd as fw
    from fwumious_wabbit_amd import _capi as capi

    class A:  # bench.py's defaults (config C)
        fields, k, bits, ffm_bits = 30, 8, 28, 28
        mean_extra, zipf, ids, p_weighted, seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
        label_flip, nn_layers, nn_width, lr, power_t = 0.0, 0, 256, bench.LR, bench.POWER_T

    args = A()
    args.holdout = a.holdout
    bench.apply_family(args, a.family)
    torch.cuda.set_device(0)
    K, B = a.steps, a.batch
    mi = bench.build_model_instance(fw, args, 0)
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    t0 = time.time()
    gen_threads = min(128, os.cpu_count() or 8)
    batches = []
    for s in range(K):
        recs, off = bench.gen_records(fw, args, s * B, B, threads=gen_threads)
        batches.append(re.record_batch(fbt, recs, off))
    hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, a.holdout, threads=gen_threads)
    hbatch = re.record_batch(fbt, hrecs, hoff)
    hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
    prep_s = time.time() - t0
    print(f"prepared {K * B} examples in {prep_s:.0f} s", file=sys.stderr, flush=True)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    oc = bench.oracle_reference_curves(args, 1)
    out = {"family": a.family, "steps": K, "batch": B, "holdout": a.holdout, "prep_seconds": prep_s,
           "holdout_prior_logloss": bench.logloss(np.full(len(hy), float(np.mean(hy == 1)), dtype=np.float64), hy),
           "oracle_seq": {str(n): v for n, v in sorted(oc["seq"].items()) if n % (a.every * B) == 0 and n <= K * B},
           "oracle_hog16": [{str(n): v for n, v in sorted(c.items()) if n % (a.every * B) == 0 and n <= K * B} for c in oc["hog16"]],
           "variants": []}
    for spec in a.variants:
        name, _, kv = spec.partition(":")
        opt = dict(x.split("=") for x in kv.split(",") if x)
        reps = int(opt.pop("reps", 1))
        re.set_store_policy(int(opt.get("policy", -1)), int(opt.get("wb", -1)))
        re.set_hot_row_sampling(float(opt.get("theta", -1)), int(opt.get("m", -1)))
        re.set_max_in_flight(int(opt.get("inflight", 0)))
        re.set_lds_keep(int(opt.get("keep", -1)))
        capi.check(capi.lib().fwgpu_debug_set_option(re.h, 12, int(opt.get("lrthin", -1))))
        capi.check(capi.lib().fwgpu_debug_set_option(re.h, 13, int(opt.get("kept", -1))))
        for rep in range(reps):
            re.allocate_and_init_weights()
            torch.cuda.synchronize()
            curve, train_ms = {}, 0.0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i0 in range(0, K, a.every):
                e0.record(stream)
                for i in range(i0, min(K, i0 + a.every)):
                    re.learn_batch(batches[i], capi.MODE_HOGWILD, True, sptr)
                e1.record(stream)
                n = min(K, i0 + a.every)
                re.learn_batch(hbatch, capi.MODE_HOGWILD, False, sptr)
                curve[str(n * B)] = bench.logloss(hbatch.predictions(sptr), hy)
                train_ms += e0.elapsed_time(e1)
            v = {"name": name, "rep": rep, "options": opt, "examples_per_sec": K * B / (train_ms * 1e-3), "final": curve[str(K * B)],
                 "min": min(curve.values()), "min_at": min(curve, key=curve.get), "curve": curve}
            out["variants"].append(v)
            print(f"{name:28s} rep {rep}: {v['examples_per_sec'] / 1e6:6.3f} M ex/s  final {v['final']:.4f}  min {v['min']:.4f} at {int(v['min_at']) / 1e6:.1f} M", file=sys.stderr, flush=True)
            if a.out:
                with open(a.out + ".tmp", "w") as f:
                    json.dump(out, f, indent=1)
                os.replace(a.out + ".tmp", a.out)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
