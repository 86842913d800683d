import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import make_pair
from oracle import fwo

n_ns, k, bits = 10, 4, 12
mi, ocfg, ots = make_pair(n_ns, k, bits, bits, fw.Optimizer.AdagradLUT)
recs, off = fw.synth_records(n_ns, 0.0, 1.1, 3000, 0.0, 1, 0, 100)
om = fwo.Model(ocfg); re = fw.Regressor(mi); fbt = fw.FeatureBufferTranslator(mi)
R = n_ns * k
for i in range(100):
    fb = fbt.translate(recs[int(off[i]):int(off[i + 1])])
    w_before = om.ffm_weights.copy(); a_before = om.ffm_acc.copy()
    pg = re.learn(fb, None, True)
    po = om.learn(fb.lr_buffer, fb.ffm_buffer, fb.label, fb.example_importance, True)
    wg, ag = re.table_read(capi.TABLE_FFM_W), re.table_read(capi.TABLE_FFM_ACC)
    dW = np.abs(wg - om.ffm_weights)
    if dW.max() > 1e-5 or abs(pg - po) > 1e-5:
        print(f"example {i}: dp={abs(pg-po):.3e} label={fb.label}")
        hs = [int(e['hash']) for e in fb.ffm_buffer]
        print("ffm hashes:", hs, "fields:", [int(e['contra_field_index']) // k for e in fb.ffm_buffer])
        bad = np.nonzero(dW > 1e-5)[0]
        print("bad addrs:", bad[:40])
        for a in bad[:12]:
            owners = [(j, a - h) for j, h in enumerate(hs) if h <= a < h + R]
            print(f"  a={a} owners(feature, elem)={owners} w_before={w_before[a]:.6e} w_ref={om.ffm_weights[a]:.6e} w_gpu={wg[a]:.6e} "
                  f"acc_before={a_before[a]:.4e} acc_ref={om.ffm_acc[a]:.4e} acc_gpu={ag[a]:.4e}")
        break
else:
    print("no deviation in 100 examples")
