"""Throughput of the in-order mode (FWGPU_MODE_SEQUENTIAL: one workgroup walks the batch in example order = the reference's single thread, exactly)."""
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
args.fields, args.k, args.bits, args.ffm_bits = 30, int(os.environ.get("K", 8)), 28, 28
args.nn_layers, args.nn_width = 0, 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
N = int(os.environ.get("N", 4096))
mi = bench.build_model_instance(fw, args, 0)
re = fw.Regressor(mi)
fbt = fw.FeatureBufferTranslator(mi)
recs, off = bench.gen_records(fw, args, 0, N)
b = re.record_batch(fbt, recs, off)
for thr in (512, 1024):
    re.set_launch(thr, 0)
    for update in (True, False):
        re.learn_batch(b, capi.MODE_SEQUENTIAL, update)
        b.predictions()
        t0 = time.perf_counter()
        re.learn_batch(b, capi.MODE_SEQUENTIAL, update)
        b.predictions()
        dt = time.perf_counter() - t0
        print(f"threads {thr} {'learn' if update else 'predict'}: {N / dt:,.0f} examples/s in order ({dt / N * 1e6:.1f} us per example)", flush=True)
