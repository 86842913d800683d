"""Functional multi-rank runs of the library's synchronous multi-GPU modes on ONE GPU: N ranks inside one process
(fwgpu_dist_group_*: the same phases as the RCCL path, collectives done by device copies), config C (28-bit tables, every rank a full-size model).
Prints, per mode and N: examples/s of the emulation (all ranks share the one GPU: a functional figure, not a scaling one),
hold-out log-loss of rank 0's model, and whether all ranks ended with the same model."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from fwumious_wabbit_amd.dist import DistGroup
import bench


class A:
    pass


args = A()
args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
GLOBAL = {"sparse": 8192, "sharded": int(os.environ.get("SHARDED_BATCH", 1024)), "owner": int(os.environ.get("OWNER_BATCH", 8192))}
TOTAL = int(os.environ.get("TOTAL", 262144))
mi = bench.build_model_instance(fw, args, 0)
recs, off = bench.gen_records(fw, args, 0, TOTAL)
hrecs, hoff = bench.gen_records(fw, args, 1_000_000_000, 16384)
hy = hrecs[hoff[:-1].astype(np.int64) + 1].astype(np.float32)
for mode in os.environ.get("MODES", "sparse,sharded,owner").split(","):
    for n in (1, 2, 4):
        regs = [fw.Regressor(mi) for _ in range(n)]
        fbt = fw.FeatureBufferTranslator(mi)
        g = DistGroup(regs)
        if mode == "owner":  # owner-side apply: hogwild kernels push gradient rows to the rows' owners, every owner applies concurrently
            g.set_mode(capi.MODE_HOGWILD)
        gb = GLOBAL[mode]
        per = gb // n
        steps = TOTAL // gb
        t0 = time.perf_counter()
        for s in range(steps):
            rr, oo = [], []
            for j in range(n):
                a, b = s * gb + j * per, s * gb + (j + 1) * per
                rr.append(recs[int(off[a]):int(off[b])])
                oo.append(off[a:b + 1] - off[a])
            {"sparse": g.learn_sparse, "sharded": g.learn_sharded, "owner": g.learn_owner}[mode](fbt, rr, oo)
        dt = time.perf_counter() - t0
        if mode in ("sharded", "owner"):
            g.gather_tables()
        hb = regs[0].record_batch(fbt, hrecs, hoff)
        regs[0].learn_batch(hb, capi.MODE_HOGWILD, False)
        ll = bench.logloss(hb.predictions(), hy)
        sums = [[r.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)] for r in regs]
        same = all(x == sums[0] for x in sums)
        print(f"{mode:8s} ranks {n}: global batch {gb} = {n} x {per}, {steps} steps, {TOTAL / dt:,.0f} examples/s (host records in, one GPU for all ranks), "
              f"hold-out log-loss {ll:.4f}, all ranks' models identical: {same}, checksum(ffm_w) {sums[0][1]}", flush=True)
        hb.close()
        g.close()
        for r in regs:
            r.close()
