"""Reproducibility of the in-process sparse step: N ranks, config C at full table size, `steps` steps; prints the table checksums of every rank
(FWGPU_GROUP_CONCURRENT=1 lets the ranks' phases overlap on their streams, the schedule that was seen to misbehave on some boxes)."""
import sys, os
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
mi = bench.build_model_instance(fw, args, 0)
n, per, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
in_flight = int(sys.argv[4]) if len(sys.argv) > 4 else int(os.environ.get("GRID", "0"))  # cap on each rank's persistent grid (n x in_flight <= 768: all ranks' workgroups co-resident)
recs, off = bench.gen_records(fw, args, 0, n * per * steps)
regs = [fw.Regressor(mi) for _ in range(n)]
if in_flight:
    for r in regs:
        r.set_max_in_flight(in_flight)
fbt = fw.FeatureBufferTranslator(mi)
g = DistGroup(regs)
for s in range(steps):
    rr, oo = [], []
    for j in range(n):
        a, b = (s * n + j) * per, (s * n + j + 1) * per
        rr.append(recs[int(off[a]):int(off[b])])
        oo.append(off[a:b + 1] - off[a])
    outs = g.learn_sparse(fbt, rr, oo)
    if os.environ.get("REPRO_SAVE"):
        np.savez(os.environ["REPRO_SAVE"] + f"_step{s}.npz", *outs)
print("final", [tuple(r.table_checksum(t) % 1000000 for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)) for r in regs])
