import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import make_pair
from oracle import fwo

def trial(F, k, feats, label=1.0, reps=2, tag=""):
    mi, ocfg, _ = make_pair(F, k, 12, 14, fw.Optimizer.AdagradLUT)
    om = fwo.Model(ocfg); re = fw.Regressor(mi)
    worst = 0
    for r in range(reps):
        pg = re.learn(fw.lr_and_ffm_vec([], feats, label), None, True)
        po = om.learn(None, fwo.ffm_entries(feats), label, 1.0, True)
        worst = max(worst, abs(pg - po))
    w_g, w_o = re.table_read(capi.TABLE_FFM_W), om.ffm_weights
    a_g, a_o = re.table_read(capi.TABLE_FFM_ACC), om.ffm_acc
    dW, dA = np.abs(w_g - w_o), np.abs(a_g - a_o)
    print(f"{tag:38s} F={F} k={k} dp={worst:.2e} dW={dW.max():.2e}@{dW.argmax()} dA={dA.max():.2e}@{dA.argmax()} nW>1e-5={int((dW>1e-5).sum())}")
    re.close()

for F, k in ((5, 4), (10, 4), (30, 8)):
    R = F * k
    trial(F, k, [(64, 1.0, 0), (1000, 1.0, 1 * k)], tag="no overlap")
    trial(F, k, [(64, 1.0, 0), (64 + k, 1.0, 1 * k)], tag="overlap shift k, fields 0,1")
    trial(F, k, [(64, 1.0, 0), (64 + 2 * k, 1.0, 2 * k)], tag="overlap shift 2k, fields 0,2")
    trial(F, k, [(64, 1.0, 0), (64, 1.0, 1 * k)], tag="same row two fields")
    trial(F, k, [(64, 1.0, 0), (2000, 1.0, 1 * k), (64 + k, 1.0, 2 * k)], tag="i, other, j overlaps i")
    trial(F, k, [(64 + k, 1.0, 0), (64, 1.0, 1 * k)], tag="later row starts lower")
