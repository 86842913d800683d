"""Config E (k = 16 + 2 x 256 ReLU head), PREDICT-ONLY launches of a device-resident record batch: examples/s of the batched head (x on the v2 kernel, layers as
MFMA GEMMs over the batch) -- or, with FWGPU_HEAD_PREDICT_PER_EXAMPLE=1, of the per-example forward inside the generic kernel -- and the largest difference
between the two forms' predictions when both are run (COMPARE=1: one process, the per-example form through the in-order launch).
usage: python3 scripts/e_predict_rate.py   env: B (65536), K (16), NN (2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench


class A:
    pass


args = A()
args.fields, args.k, args.bits, args.ffm_bits = 30, int(os.environ.get("K", 16)), 28, 28
args.nn_layers, args.nn_width = int(os.environ.get("NN", 2)), 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
B = int(os.environ.get("B", 65536))
mi = bench.build_model_instance(fw, args, 0)
re = fw.Regressor(mi)
fbt = fw.FeatureBufferTranslator(mi)
recs, off = bench.gen_records(fw, args, 0, B)
b = re.record_batch(fbt, recs, off)
tb = re.record_batch(fbt, recs[: int(off[8192])], off[:8193])
for _ in range(3):
    re.learn_batch(tb, capi.MODE_HOGWILD, True)  # some training, so that the head is not at its initial state
tb.predictions()
for _ in range(3):
    re.learn_batch(b, capi.MODE_HOGWILD, False)
p = b.predictions().copy()
t0 = time.perf_counter()
R = 8
for _ in range(R):
    re.learn_batch(b, capi.MODE_HOGWILD, False)
b.predictions()
dt = (time.perf_counter() - t0) / R
form = "per-example forward" if os.environ.get("FWGPU_HEAD_PREDICT_PER_EXAMPLE") else "batched head"
print(f"{form}: {B} examples in {dt * 1e3:.3f} ms -> {B / dt / 1e6:.2f} M examples/s", flush=True)
if os.environ.get("COMPARE"):
    n = 4096
    sb = re.record_batch(fbt, recs[: int(off[n])], off[: n + 1])
    re.learn_batch(sb, capi.MODE_SEQUENTIAL, False)  # in-order launches always take the per-example forward
    q = sb.predictions()
    print(f"max |p_batched - p_per_example| over {n} examples: {np.abs(p[:n] - q).max():.3e}")
