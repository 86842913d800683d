"""Does the constant feature (one LR entry read-modify-written by EVERY example) hold the config-C kernel back?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench
class A: pass
args = A(); args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
B = 16384
recs, off = bench.gen_records(fw, args, 0, 3 * B)
for const in (True, False, True, False):
    mi = bench.build_model_instance(fw, args, 0)
    mi.add_constant_feature = const
    re = fw.Regressor(mi); fbt = fw.FeatureBufferTranslator(mi)
    bs = [re.record_batch(fbt, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(3)]
    re.learn_batch(bs[0], capi.MODE_HOGWILD, True); bs[0].predictions()
    t0 = time.perf_counter()
    for i in range(12): re.learn_batch(bs[i % 3], capi.MODE_HOGWILD, True)
    bs[2].predictions(); dt = (time.perf_counter() - t0) / 12
    print(f"add_constant_feature={const}: {dt*1e3:.3f} ms/launch {B/dt/1e6:.2f} M ex/s", flush=True)
    for b in bs: b.close()
    re.close()
