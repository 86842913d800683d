"""Owner-side apply, STREAMING form (fwgpu_dist_group_learn_owner_stream), N in-process ranks sharing ONE GPU, config C (30 fields, k = 8, 28-bit tables):
examples/s over device-resident record batches and the hold-out loss of the gathered model, next to the sequential oracle's committed curve.
usage: python3 scripts/owner_stream_rate.py   env: RANKS (4), B (global examples per step, 65536), STEPS (12), LG_ROWS (15), LG_LR (16), CWG (consumer workgroups, 48), BITS (28)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
if os.environ.get("TORCH_AFTER") == "1":  # (what pytest's conftest does between loading the library and the first test)
    import torch
    print("torch sees a GPU:", torch.cuda.is_available(), flush=True)
from fwumious_wabbit_amd import _capi as capi
from fwumious_wabbit_amd.dist import DistGroup
import bench


class A:
    pass


args = A()
args.fields, args.k = int(os.environ.get("FIELDS", 30)), int(os.environ.get("K", 8))
args.bits = args.ffm_bits = int(os.environ.get("BITS", 28))
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = float(os.environ.get("MEAN_EXTRA", 5.67)), 1.05, 10_000_000, 0.1, 20240612
N = int(os.environ.get("RANKS", 4))
B = int(os.environ.get("B", 65536))
K = int(os.environ.get("STEPS", 12))
lgr, lgl, cwg = int(os.environ.get("LG_ROWS", 15)), int(os.environ.get("LG_LR", 16)), int(os.environ.get("CWG", 0))
mi = bench.build_model_instance(fw, args, 0)
regs = [fw.Regressor(mi) for _ in range(N)]
fbt = fw.FeatureBufferTranslator(mi)
g = DistGroup(regs)
g.set_mode(capi.MODE_HOGWILD)
per = B // N
batches, host = [], []
HOST = os.environ.get("RECORDS") == "1"  # feed host records (the entry point the tests use) instead of device-resident batches
for s in range(K):
    recs, off = bench.gen_records(fw, args, s * B, B)
    if HOST:
        host.append(([recs[int(off[j * per]):int(off[(j + 1) * per])] for j in range(N)], [off[j * per:(j + 1) * per + 1] - off[j * per] for j in range(N)]))
    batches.append([regs[j].record_batch(fbt, recs[int(off[j * per]):int(off[(j + 1) * per])], off[j * per:(j + 1) * per + 1] - off[j * per]) for j in range(N)])
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, 65536)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
def step(s):
    if HOST:
        g.learn_owner_stream(fbt, host[s][0], host[s][1], log2_rows=lgr, log2_lr=lgl, consumer_workgroups=cwg)
    else:
        g.learn_owner_stream(fbt, batches=batches[s], log2_rows=lgr, log2_lr=lgl, consumer_workgroups=cwg)


step(0)  # warm-up: regions allocated, kernels loaded
t0 = time.perf_counter()
for s in range(1, K):
    step(s)
dt = time.perf_counter() - t0
g.gather_tables()
hb = regs[0].record_batch(fbt, hrecs, hoff)
regs[0].learn_batch(hb, capi.MODE_HOGWILD, False)
ll = bench.logloss(hb.predictions(), hy)
p_last = np.concatenate([b.predictions() for b in batches[K - 1]]) if not HOST else np.zeros(1)
print(json.dumps({"ranks": N, "examples_per_step": B, "steps_timed": K - 1, "examples_per_sec": (K - 1) * B / dt, "ms_per_step": 1e3 * dt / (K - 1),
                  "holdout_logloss_65536": ll, "examples_learned": K * B, "finite": bool(np.all(np.isfinite(p_last))), "log2_rows": lgr, "log2_lr": lgl, "consumer_workgroups": cwg}), flush=True)
