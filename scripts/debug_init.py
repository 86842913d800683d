import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import make_pair
from oracle import fwo
mi, ocfg, _ = make_pair(5, 8, 14, 14, fw.Optimizer.AdagradLUT, ffm_init_acc=0.5)
mi.ffm_init_width, mi.ffm_init_zero_band, mi.ffm_init_center = 0.2, 0.25, 0.01
ocfg.ffm_init_width, ocfg.ffm_init_zero_band, ocfg.ffm_init_center = 0.2, 0.25, 0.01
om = fwo.Model(ocfg); re = fw.Regressor(mi)
g, o = re.table_read(capi.TABLE_FFM_W), om.ffm_weights
bad = np.nonzero(g.view(np.uint32) != o.view(np.uint32))[0]
print("n bad", len(bad), "of", len(g), "first", bad[:10])
for i in bad[:8]:
    print(i, repr(float(g[i])), repr(float(o[i])), hex(g.view(np.uint32)[i]), hex(o.view(np.uint32)[i]))
