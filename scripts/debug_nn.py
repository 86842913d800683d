"""Deep-head parity diagnostics: per-table max deviations GPU vs oracle after a sequential stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import capi
from oracle import fwo
from helpers import make_pair, logloss, record_labels

def run(n_ns, k, bits, ffm_bits, opt, layers, topo, n, seed, interactions=(), **kw):
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, opt, interactions=interactions, **kw)
    mi.nn_layers = [dict(width=w, activation=a, init=i) for w, a, i in layers]
    mi.nn_topology = topo
    nn = fwo.make_nn_config(layers, topo)
    recs, off = fw.synth_records(n_ns, 1.0, 1.1, 3000, 0.1, seed, 0, n)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg, nn=nn)
    L = len(layers)
    w0 = np.concatenate([om.nn_weights(l).copy() for l in range(L + 1)])
    _, p_ref = om.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
    w1 = np.concatenate([om.nn_weights(l) for l in range(L + 1)])
    a1 = np.concatenate([om.nn_acc(l) for l in range(L + 1)])
    re = fw.Regressor(mi)
    re.table_write(capi.TABLE_NN_W, w0)
    b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    p = b.predictions()
    dp = np.abs(p - p_ref)
    print("pred max diff", dp.max(), "at", dp.argmax(), "first >1e-6:", np.argmax(dp > 1e-6) if (dp > 1e-6).any() else None)
    for name, g, o in (("nn_w", re.table_read(capi.TABLE_NN_W), w1), ("nn_acc", re.table_read(capi.TABLE_NN_ACC), a1),
                       ("lr", re.table_read(capi.TABLE_LR), om.lr_table), ("ffm_w", re.table_read(capi.TABLE_FFM_W), om.ffm_weights),
                       ("ffm_acc", re.table_read(capi.TABLE_FFM_ACC), om.ffm_acc)):
        d = np.abs(g - o)
        i = d.argmax()
        print(f"{name}: max|d|={d.max():.3e} at {i} gpu={g[i]} ref={o[i]} rel={d.max() / (abs(o[i]) + 1e-30):.2e} n>2e-5: {(d > 2e-5 + 1e-5 * np.abs(o)).sum()}")

run(6, 4, 12, 12, fw.Optimizer.AdagradLUT, [(12, "relu", "hu"), (8, "relu", "hu")], "one", int(sys.argv[1]) if len(sys.argv) > 1 else 600, 21, interactions=[(0, 1)])
