"""Does the learn-launch time depend on WHICH physical memory the tables got?  Re-creates the regressor several times in one
process (free + allocate again) and times the same launches each time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench


class A:
    pass


args = A()
args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
B = 16384
mi = bench.build_model_instance(fw, args, 0)
recs, off = bench.gen_records(fw, args, 0, 2 * B)
for rep in range(int(os.environ.get("REPS", 8))):
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    batches = [re.record_batch(fbt, recs[int(off[s * B]):int(off[(s + 1) * B])], off[s * B:(s + 1) * B + 1] - off[s * B]) for s in range(2)]
    for i in range(4):
        re.learn_batch(batches[i % 2], capi.MODE_HOGWILD, True)
    batches[1].predictions()
    t0 = time.perf_counter()
    for i in range(16):
        re.learn_batch(batches[i % 2], capi.MODE_HOGWILD, True)
    batches[1].predictions()
    dt = (time.perf_counter() - t0) / 16
    ptrs = [hex(re.table_device_ptr(t)) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    print(f"regressor #{rep}: {dt * 1e3:.3f} ms/launch  tables (lr, w, acc) at {ptrs}", flush=True)
    for b in batches:
        b.close()
    re.close()
