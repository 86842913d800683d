"""CPU: does the sequential reference algorithm converge on the config-C synthetic stream, and at which lr?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from helpers import make_pair, logloss, record_labels
from oracle import fwo
n_train, n_hold = int(sys.argv[1]), 4000
recs, off = fw.synth_records(30, 5.67, 1.05, 10_000_000, 0.1, 20240612, 0, n_train + n_hold)
y = record_labels(recs, off)
print("label mean", y.mean())
for lr, threads in ((0.1, 1), (0.025, 1), (0.01, 1), (0.025, 8)):
    mi, ocfg, ots = make_pair(30, 8, 24, 24, fw.Optimizer.AdagradLUT, lr=lr, ffm_lr=lr)
    om = fwo.Model(ocfg, native=True)
    t = time.time()
    _, p = om.run_stream(ots, recs, off, holdout_after=n_train + 1, nthreads=threads)
    ll_hold = logloss(p[n_train:], y[n_train:]).mean()
    ll_train = logloss(p[n_train // 2:n_train], y[n_train // 2:n_train]).mean() if threads == 1 else float('nan')
    print(f"lr={lr} threads={threads}: holdout logloss {ll_hold:.4f}  progressive(train 2nd half) {ll_train:.4f}  ({time.time()-t:.1f}s)", flush=True)
    om.close()
