import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from fwumious_wabbit_amd.dist import DistGroup
from helpers import logloss, make_pair, record_labels
from oracle import fwo
n_ns,k,bits,ffm_bits,extra,ids = 30,8,16,18,3.0,50000
mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
steps, gb = 6, 96
recs, off = fw.synth_records(n_ns, extra, 1.1, ids, 0.1, 81, 0, steps * gb)
om = fwo.Model(ocfg)
re = fw.Regressor(mi); fbt = fw.FeatureBufferTranslator(mi); sp = re.split_buffers(gb, 512)
re2 = fw.Regressor(mi); g = DistGroup([re2]); g.set_mode(capi.MODE_SEQUENTIAL)
for s in range(steps):
    sub, so = recs[int(off[s*gb]):int(off[(s+1)*gb])], off[s*gb:(s+1)*gb+1]-off[s*gb]
    p_ref = om.learn_minibatch(ots, sub, so)
    b = re.record_batch(fbt, sub, so); re.learn_batch_sync(b, sp, capi.MODE_SEQUENTIAL); p1 = b.predictions()
    p2 = g.learn_sharded(fbt, [sub], [so])[0]
    print(s, "sync-oracle", np.abs(p1-p_ref).max(), "dist-oracle", np.abs(p2-p_ref).max(), "dist-sync", np.abs(p2-p1).max())
    for nm, t, ot in (("lr", capi.TABLE_LR, om.lr_table), ("w", capi.TABLE_FFM_W, om.ffm_weights), ("acc", capi.TABLE_FFM_ACC, om.ffm_acc)):
        a1, a2 = re.table_read(t), re2.table_read(t)
        print("   ", nm, "sync-oracle", np.abs(a1-ot).max(), "dist-oracle", np.abs(a2-ot).max(), "at", int(np.abs(a2-ot).argmax()))
