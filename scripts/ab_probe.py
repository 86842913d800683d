"""Interleaved A/B of launch options on config-C learn launches (one process, same batches, N rounds).
usage: python3 scripts/ab_probe.py "name:opt=val,opt=val;name2:..."   options: window, lutg, threads, wgs, kv (kernel version)
env: B (examples per launch, default 16384), ROUNDS (default 5), FIELDS, K, ZIPF (id skew, default 1.05), PREDICT=1 adds a predict-only timing."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench


class A:
    pass


args = A()
args.fields, args.k, args.bits, args.ffm_bits = int(os.environ.get("FIELDS", 30)), int(os.environ.get("K", 8)), 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, float(os.environ.get("ZIPF", 1.05)), 10_000_000, 0.1, 20240612
B = int(os.environ.get("B", 16384))
NB = 3
ROUNDS = int(os.environ.get("ROUNDS", 5))
mi = bench.build_model_instance(fw, args, 0)
re = fw.Regressor(mi)
fbt = fw.FeatureBufferTranslator(mi)
L = capi.lib()
recs, off = bench.gen_records(fw, args, 0, NB * B)
batches = [re.record_batch(fbt, recs[int(off[s * B]):int(off[(s + 1) * B])], off[s * B:(s + 1) * B + 1] - off[s * B]) for s in range(NB)]

variants = []
for spec in (sys.argv[1] if len(sys.argv) > 1 else "win:window=2;nowin:window=0").split(";"):
    name, _, opts = spec.partition(":")
    d = dict(window=1, lutg=0, threads=512, wgs=0, kv=0)
    for kv in filter(None, opts.split(",")):
        a, b = kv.split("=")
        d[a] = int(b)
    variants.append((name, d))


def apply(d):
    capi.check(L.fwgpu_debug_set_option(re.h, 1, d["lutg"]))
    capi.check(L.fwgpu_debug_set_option(re.h, 2, d["window"]))
    capi.check(L.fwgpu_debug_set_kernel_version(re.h, d["kv"]))
    re.set_launch(d["threads"], d["wgs"])


def timed(update, reps=6):
    re.learn_batch(batches[0], capi.MODE_HOGWILD, update)
    batches[0].predictions()
    t0 = time.perf_counter()
    for i in range(reps):
        re.learn_batch(batches[i % NB], capi.MODE_HOGWILD, update)
    batches[(reps - 1) % NB].predictions()
    return (time.perf_counter() - t0) / reps


res = {n: [] for n, _ in variants}
resp = {n: [] for n, _ in variants}
for r in range(ROUNDS):
    for n, d in variants:
        apply(d)
        res[n].append(timed(True))
        if os.environ.get("PREDICT"):
            resp[n].append(timed(False))
for n, d in variants:
    t = np.array(res[n]) * 1e3
    line = f"{n:>16s}: learn ms/launch median {np.median(t):.3f} min {t.min():.3f} max {t.max():.3f} -> {B / np.median(t) / 1e3:.2f} Mex/s"
    if resp[n]:
        tp = np.array(resp[n]) * 1e3
        line += f" | predict median {np.median(tp):.3f} ms"
    print(line, flush=True)
p = batches[NB - 1].predictions()
print("finite:", bool(np.all(np.isfinite(p))), "mean p", float(p.mean()))
