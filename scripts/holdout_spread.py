"""Spread of the hogwild hold-out loss of tests/test_gpu_parity.py's trainer and hogwild tests over repeated runs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import capi
from helpers import make_pair, logloss, record_labels
import test_gpu_parity as T

n_train, n_hold = 12000, 2000
mi, ocfg, ots = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 77, 0, n_train + n_hold)
y = record_labels(recs, off)
ref_hold, _ = T._holdout_loss_oracle(ocfg, ots, recs, off, n_train)
vals = []
for rep in range(12):
    re = fw.Regressor(mi)
    tr = fw.HogwildTrainer(re, mi, micro_batch=1024)
    tr.digest_records(recs[:int(off[n_train])], off[:n_train + 1])
    tr.block_until_workers_finished()
    hb = re.batch_from_records(fw.FeatureBufferTranslator(mi), recs[int(off[n_train]):], off[n_train:] - off[n_train])
    re.learn_batch(hb, capi.MODE_HOGWILD, False)
    vals.append(float(logloss(hb.predictions(), y[n_train:]).mean()))
    tr.close(); re.close()
print("oracle sequential", ref_hold, "gpu hogwild", np.round(vals, 4), "max gap", max(abs(v - ref_hold) for v in vals))
