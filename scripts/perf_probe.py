"""GPU perf exploration: config-C learn launches with per-phase shader-clock breakdown and launch-shape sweeps."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench

class A: pass
args = A(); args.fields, args.k, args.bits, args.ffm_bits = 30, 8, 28, 28
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
B = int(os.environ.get("B", 16384)); NB = 3
mi = bench.build_model_instance(fw, args, 0)
re = fw.Regressor(mi); fbt = fw.FeatureBufferTranslator(mi); L = capi.lib()
recs, off = bench.gen_records(fw, args, 0, NB * B)
batches = [re.batch_from_records(fbt, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(NB)]
names = ["stage", "scan", "gather", "dot+lrfwd+sigm", "lr_update", "ffm_update", "wait_slowest", "examples"]

def run(threads, wgs, update=True, reps=4, label=""):
    re.set_launch(threads, wgs)
    re.learn_batch(batches[0], capi.MODE_HOGWILD, update); batches[0].predictions()
    capi.check(L.fwgpu_debug_phase_ticks(re.h, 1, None))
    t0 = time.perf_counter()
    for i in range(reps):
        re.learn_batch(batches[i % NB], capi.MODE_HOGWILD, update)
    batches[(reps - 1) % NB].predictions()
    dt = (time.perf_counter() - t0) / reps
    out = (C.c_uint64 * 8)()
    capi.check(L.fwgpu_debug_phase_ticks(re.h, 0, out))
    t = np.array(list(out), dtype=np.float64)
    tot = t[:7].sum()
    per_ex = tot / max(t[7], 1)
    br = " ".join(f"{n}={100*v/tot:.0f}%" for n, v in zip(names[:7], t[:7]))
    print(f"{label} threads={threads} wgs/cu={wgs or 'auto'} update={update}: {dt*1e3:.3f} ms/launch {B/dt/1e6:.2f} Mex/s | ticks/example={per_ex:.0f} | {br}", flush=True)

run(512, 0, label="base")
run(512, 0, update=False, label="predict")
for th, w in ((256, 0), (1024, 0), (512, 2), (512, 1), (1024, 1), (256, 4)):
    run(th, w)

# ---- isolate the LR block's cost
print("--- FFM only (LR block off)")
mi2 = bench.build_model_instance(fw, args, 0); mi2.wiring = capi.WIRING_FFM_ONLY
re_full, batches_full = re, batches
re = fw.Regressor(mi2); fbt2 = fw.FeatureBufferTranslator(mi2)
batches = [re.batch_from_records(fbt2, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(NB)]
run(512, 0, label="ffm_only"); run(512, 0, update=False, label="ffm_only predict")
print("--- LR only (no FFM block)")
mi3 = bench.build_model_instance(fw, args, 0); mi3.ffm_k = 0; mi3.ffm_fields = []
re = fw.Regressor(mi3); fbt3 = fw.FeatureBufferTranslator(mi3)
batches = [re.batch_from_records(fbt3, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(NB)]
run(512, 0, label="lr_only"); run(256, 8, label="lr_only"); run(512, 0, update=False, label="lr_only predict")
