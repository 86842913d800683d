"""What does the replica exchange do to the model?  N replicas emulated on ONE GPU (sequentially, no RCCL): each trains its
own shard of bench.py's stream; at a sync point  table_r <- snapshot + combine_r(table_r - snapshot)  with combine = sum
(what dist_sync does) or mean.  Prints the hold-out log-loss of the synced model for N = 1, 2, 4, 8."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench

class A: pass
args = A(); args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
K = int(os.environ.get("STEPS", 48)); SYNC = int(os.environ.get("SYNC", 32)); B = int(os.environ.get("B", 16384))
TABLES = (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, 8192)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)

def run(N, rule, K=K):
    mi = bench.build_model_instance(fw, args, 0)
    reps = [fw.Regressor(mi) for _ in range(N)]
    fbt = fw.FeatureBufferTranslator(mi)
    views = [[r.table_as_torch(t) for t in TABLES] for r in reps]
    snap = [v.clone() for v in views[0]]
    def sync():
        for ti in range(len(TABLES)):
            tot = torch.zeros_like(snap[ti])
            for r in range(N):
                tot += views[r][ti] - snap[ti]
            if rule == "mean":
                tot /= N
            snap[ti] += tot
            for r in range(N):
                views[r][ti].copy_(snap[ti])
        torch.cuda.synchronize()
    for s0 in range(0, K, SYNC):
        for r in range(N):
            recs, off = bench.gen_records(fw, args, (r * K + s0) * B, min(SYNC, K - s0) * B)
            for s in range(min(SYNC, K - s0)):
                lo, hi = s * B, (s + 1) * B
                b = reps[r].record_batch(fbt, recs[int(off[lo]):int(off[hi])], off[lo:hi + 1] - off[lo])
                reps[r].learn_batch(b, capi.MODE_HOGWILD, True)
                b.predictions(); b.close()
        if N > 1:
            sync()
    hb = reps[0].record_batch(fbt, hrecs, hoff)
    reps[0].learn_batch(hb, capi.MODE_HOGWILD, False)
    p = hb.predictions()
    ll = bench.logloss(p, hy)
    sat = float(np.mean((p < 1e-6) | (p > 1 - 1e-6)))
    print(f"N={N} rule={rule}: hold-out log-loss {ll:.4f} after {N * K * B} examples (saturated predictions {sat:.3f})", flush=True)
    hb.close()
    for r in reps: r.close()
    del views, snap
    torch.cuda.empty_cache()

if os.environ.get("EQUAL_TOTAL"):
    # the comparison that matters: N replicas x M examples each against ONE learner on the same N x M examples
    print(f"# equal total examples: one learner on N*M examples vs N replicas (mean rule, exchange every {SYNC} steps) on M = {K * B} each")
    run(1, "mean", K)
    for N in (2, 4, 8):
        run(1, "mean", K * N)
        run(N, "mean", K)
elif os.environ.get("ONLY_N"):
    run(int(os.environ["ONLY_N"]), "mean")
elif os.environ.get("ONLY_MEAN"):
    for N in (1, 4, 8):
        run(N, "mean")
else:
    run(1, "sum")
    for N in (2, 4, 8):
        for rule in ("sum", "mean"):
            run(N, rule)
