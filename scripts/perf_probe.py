"""GPU perf exploration: config-C learn launches with per-phase shader-clock breakdown and launch-shape sweeps.
The phase stamps exist in -DFW_TICKS builds of kernels.hip only (both example kernels, round 5): `scripts/build_variant.sh ticks -DFW_TICKS`, then
`FWGPU_LIBRARY=build/variants/libfwgpu_ticks.so python3 scripts/perf_probe.py`; with the shipped library the breakdown reads zero."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
import bench

class A: pass
args = A(); args.fields, args.k, args.bits, args.ffm_bits = int(os.environ.get("FIELDS", 30)), int(os.environ.get("K", 8)), 28, 28
args.nn_layers, args.nn_width = int(os.environ.get("NN_LAYERS", 0)), 256
args.mean_extra, args.zipf, args.ids, args.p_weighted, args.seed = 5.67, 1.05, 10_000_000, 0.1, 20240612
B = int(os.environ.get("B", 16384)); NB = 3
mi = bench.build_model_instance(fw, args, 0)
re = fw.Regressor(mi); fbt = fw.FeatureBufferTranslator(mi); L = capi.lib()
recs, off = bench.gen_records(fw, args, 0, NB * B)
batches = [re.batch_from_records(fbt, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(NB)]
names = ["stage", "scan", "gather", "dot+lrfwd+sigm", "lr_update", "ffm_update", "wait_slowest", "examples"]

def run(threads, wgs, update=True, reps=6, label=""):
    re.set_launch(threads, wgs)
    re.learn_batch(batches[0], capi.MODE_HOGWILD, update); batches[0].predictions()
    # 1) wall time without instrumentation
    t0 = time.perf_counter()
    for i in range(reps):
        re.learn_batch(batches[i % NB], capi.MODE_HOGWILD, update)
    batches[(reps - 1) % NB].predictions()
    dt = (time.perf_counter() - t0) / reps
    # 2) phase breakdown (instrumented run)
    capi.check(L.fwgpu_debug_phase_ticks(re.h, 1, None))
    for i in range(2):
        re.learn_batch(batches[i % NB], capi.MODE_HOGWILD, update)
    batches[1].predictions()
    out = (C.c_uint64 * 16)()
    capi.check(L.fwgpu_debug_phase_ticks(re.h, 0, out))
    t = np.array(list(out), dtype=np.float64)
    tot = t[:7].sum()
    per_ex = tot / max(t[7], 1)
    br = " ".join(f"{n}={100*v/tot:.0f}%" for n, v in zip(names[:7], t[:7]))
    sub = " ".join(f"s{j}={100*t[8+j]/tot:.1f}%" for j in range(4) if t[8+j])
    if t[15]:
        sub += f" | per update batch: store-drain {t[12]/t[15]:.0f} load-wait {t[13]/t[15]:.0f} issue+compute {t[14]/t[15]:.0f} ticks, {t[15]/max(t[7],1):.1f} batches/example"
    if args.nn_layers:
        sub = "head: " + " ".join(f"{nm}={100*t[8+j]/tot:.1f}%" for j, nm in enumerate(["x", "fwd_l0", "fwd_l1", "fwd_final", "bwd_final", "bwd_l1", "bwd_l0"]))
    print(f"{label} threads={threads} wgs/cu={wgs or 'auto'} update={update}: {dt*1e3:.3f} ms/launch {B/dt/1e6:.2f} Mex/s | ticks/example={per_ex:.0f} | {br} | stage parts: {sub}", flush=True)

use_records = os.environ.get("RECORDS", "1") == "1"
if use_records:
    batches = [re.record_batch(fbt, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(NB)]
quick = os.environ.get("QUICK", "0") == "1"
for lutg in ((int(os.environ.get("LUTG", 0)),) if quick else (0, 1)):
    capi.check(L.fwgpu_debug_set_option(re.h, 1, lutg))
    capi.check(L.fwgpu_debug_set_option(re.h, 2, int(os.environ.get("WINDOW", 1))))
    print(f"--- kernel v2, lut_global={lutg}, records={use_records}")
    for th, w in (((int(os.environ.get("THREADS", 512)), int(os.environ.get("WGS", 0))),) if quick else ((512, 2), (384, 2), (320, 3), (384, 3), (256, 4), (448, 2))):
        run(th, w)
        if quick:
            run(th, w, update=False)
