"""How does the launch time of the config-C learn kernel evolve from a cold box?  Prints ms/launch against wall time since the
first launch (continuous launches), then again after an idle pause."""
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
re = fw.Regressor(mi)
fbt = fw.FeatureBufferTranslator(mi)
recs, off = bench.gen_records(fw, args, 0, 2 * B)
batches = [re.record_batch(fbt, recs[int(off[s * B]):int(off[(s + 1) * B])], off[s * B:(s + 1) * B + 1] - off[s * B]) for s in range(2)]


def burst(seconds, label):
    t_start = time.perf_counter()
    i = 0
    while time.perf_counter() - t_start < seconds:
        t0 = time.perf_counter()
        for _ in range(8):
            re.learn_batch(batches[i % 2], capi.MODE_HOGWILD, True)
            i += 1
        batches[(i - 1) % 2].predictions()
        dt = (time.perf_counter() - t0) / 8
        print(f"{label} t={time.perf_counter() - t_start:6.2f}s  {dt * 1e3:.3f} ms/launch", flush=True)
        time.sleep(float(os.environ.get("GAP", 0)))


burst(float(os.environ.get("T1", 12)), "cold")
time.sleep(float(os.environ.get("IDLE", 5)))
burst(3, "after-idle")
