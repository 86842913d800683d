"""Are the fused kernels exposed to the concurrency fault of the phase kernels?  Regressor A runs READ-ONLY launches of one fixed
batch (deterministic: every launch must give the same predictions, bit for bit) on one stream while regressor B learns (hogwild, its own
tables) on another stream of the same device: two different kernels from two hardware queues sharing the CUs.
usage: overlap_exactness.py [launches=200] [batch=16384] [bits=26]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench


class A:
    pass


launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
per = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 26
args = A()
args.fields, args.k, args.bits, args.ffm_bits = 30, 8, bits, bits
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
mi = bench.build_model_instance(fw, args, 0)
recs, off = bench.gen_records(fw, args, 0, 3 * per)
ra, rb = fw.Regressor(mi), fw.Regressor(mi)
fbt = fw.FeatureBufferTranslator(mi)
o = off[:per + 1]
ba = ra.record_batch(fbt, recs[:int(o[-1])], o)
bbs = []
for j in (1, 2):
    oo = off[j * per:(j + 1) * per + 1]
    bbs.append(rb.record_batch(fbt, recs[int(oo[0]):int(oo[-1])], oo - oo[0]))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
# A learns a little first so that its tables are not the initial ones, then stays read-only
ra.learn_batch(ba, capi.MODE_HOGWILD, True, sa.cuda_stream)
sa.synchronize()
ra.learn_batch(ba, capi.MODE_HOGWILD, False, sa.cuda_stream)
ref = ba.predictions(sa.cuda_stream).copy()
sa.synchronize()
ref = ba.predictions(sa.cuda_stream).copy()


def run(overlap):
    bad_launches, bad_examples = 0, 0
    t0 = time.time()
    for i in range(launches):
        if overlap:
            for b in bbs:
                rb.learn_batch(b, capi.MODE_HOGWILD, True, sb.cuda_stream)
        ra.learn_batch(ba, capi.MODE_HOGWILD, False, sa.cuda_stream)
        p = ba.predictions(sa.cuda_stream)
        d = int((p.view(np.uint32) != ref.view(np.uint32)).sum())
        if d:
            bad_launches += 1
            bad_examples += d
    sb.synchronize()
    return bad_launches, bad_examples, time.time() - t0


for name, ov in (("alone", False), ("with another regressor learning on a second stream", True), ("alone again", False)):
    bl, be, dt = run(ov)
    print(f"read-only launches {name}: {launches} launches x {per} examples, launches with a differing prediction: {bl} ({be} examples), {dt:.1f} s", flush=True)
