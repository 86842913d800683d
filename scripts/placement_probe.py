"""Is the run-to-run spread of the learn launch (3.5 vs 3.85 ms on one box) a property of WHERE the tables landed or of WHEN the
launches ran?  Several regressors alive at once in one process; bursts of timed launches interleaved between them."""
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
NREG = int(os.environ.get("NREG", 4))
mi = bench.build_model_instance(fw, args, 0)
recs, off = bench.gen_records(fw, args, 0, 2 * B)
regs = []
for r in range(NREG):
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    batches = [re.record_batch(fbt, recs[int(off[s * B]):int(off[(s + 1) * B])], off[s * B:(s + 1) * B + 1] - off[s * B]) for s in range(2)]
    regs.append((re, batches))
    print(f"regressor {r}: placement search (tries, fastest ms, slowest ms) {re.placement()}", flush=True)
    print(f"regressor {r}: tables (lr, w, acc) at {[hex(re.table_device_ptr(t)) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]}", flush=True)
for rnd in range(int(os.environ.get("ROUNDS", 5))):
    row = []
    for re, batches in regs:
        for i in range(3):
            re.learn_batch(batches[i % 2], capi.MODE_HOGWILD, True)
        batches[0].predictions()
        t0 = time.perf_counter()
        for i in range(12):
            re.learn_batch(batches[i % 2], capi.MODE_HOGWILD, True)
        batches[1].predictions()
        row.append((time.perf_counter() - t0) / 12 * 1e3)
    print(f"round {rnd}: " + "  ".join(f"{x:.3f}" for x in row), flush=True)
print("tries: " + " ".join(str(re.placement()[0]) for re, _ in regs))
