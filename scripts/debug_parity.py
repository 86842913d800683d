"""Debug helper (GPU): where does the sequential stream first deviate from the oracle?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import make_pair, record_labels
from oracle import fwo


def run(name, n_ns, k, bits, ffm_bits, n, mean_extra, p_weighted, ids, seed, interactions=(), per_example=False, opt=fw.Optimizer.AdagradLUT):
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, opt, interactions=interactions)
    recs, off = fw.synth_records(n_ns, mean_extra, 1.1, ids, p_weighted, seed, 0, n)
    om = fwo.Model(ocfg)
    _, p_ref = om.run_stream(ots, recs, off, nthreads=1)
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    if per_example:
        p_gpu = np.zeros(n, np.float32)
        for i in range(n):
            fb = fbt.translate(recs[int(off[i]):int(off[i + 1])])
            p_gpu[i] = re.learn(fb, None, True)
    else:
        b = re.batch_from_records(fbt, recs, off)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        p_gpu = b.predictions()
    d = np.abs(p_gpu - p_ref)
    bad = np.nonzero(d > 1e-5)[0]
    dw = np.abs(re.table_read(capi.TABLE_FFM_W) - om.ffm_weights).max() if k else 0
    da = np.abs(re.table_read(capi.TABLE_FFM_ACC) - om.ffm_acc).max() if k else 0
    dl = np.abs(re.table_read(capi.TABLE_LR) - om.lr_table).max()
    print(f"{name:40s} per_example={per_example!s:5s} max|dp|={d.max():.3e} first_bad={bad[0] if len(bad) else -1} "
          f"nbad={len(bad)} dW={dw:.2e} dAcc={da:.2e} dLR={dl:.2e}", flush=True)
    re.close()


for pe in (True, False):
    run("b-like 1feat/field collisions", 10, 4, 12, 12, 600, 0.0, 0.0, 3000, 1, per_example=pe)
    run("b-like big table (no collisions)", 10, 4, 20, 20, 600, 0.0, 0.0, 3000, 1, per_example=pe)
    run("multi-feature fields, big table", 10, 4, 20, 20, 300, 2.0, 0.0, 100000, 2, per_example=pe)
    run("weighted values, big table", 10, 4, 20, 20, 300, 0.0, 0.5, 100000, 3, per_example=pe)
    run("c-like", 30, 8, 18, 18, 150, 5.67, 0.1, 100000, 2, per_example=pe)
    run("c-like no weights", 30, 8, 18, 18, 150, 5.67, 0.0, 100000, 2, per_example=pe)
    run("lr only collisions", 8, 0, 10, 10, 600, 1.0, 0.2, 2000, 7, per_example=pe)
    run("sgd b-like collisions", 10, 4, 12, 12, 600, 0.0, 0.0, 3000, 1, per_example=pe, opt=fw.Optimizer.SGD)
