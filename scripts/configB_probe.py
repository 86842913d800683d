"""Config B (10 fields, k=4, 22-bit tables): what bounds it?  Same stream with / without the constant feature and with
uniform instead of Zipf ids (hot LR entries are read-modify-written by every example in flight)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi

def run(const, zipf, ids, lr_on=True, B=65536):
    F = 10
    mi = fw.ModelInstance(learning_rate=0.025, ffm_learning_rate=0.025, power_t=0.38, ffm_power_t=0.38, init_acc_gradient=1.0,
                          ffm_init_acc_gradient=1.0, bit_precision=22, ffm_bit_precision=22, ffm_k=4, add_constant_feature=const,
                          optimizer=fw.Optimizer.AdagradLUT,
                          feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(F if lr_on else 0)],
                          ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(F)])
    re = fw.Regressor(mi); fbt = fw.FeatureBufferTranslator(mi)
    recs, off = fw.synth_records(F, 0.0, zipf, ids, 0.0, 5, 0, 3 * B)
    bs = [re.record_batch(fbt, recs[int(off[s*B]):int(off[(s+1)*B])], off[s*B:(s+1)*B+1] - off[s*B]) for s in range(3)]
    re.learn_batch(bs[0], capi.MODE_HOGWILD, True); bs[0].predictions()
    t0 = time.perf_counter()
    for i in range(9): re.learn_batch(bs[i % 3], capi.MODE_HOGWILD, True)
    bs[2].predictions(); dt = (time.perf_counter() - t0) / 9
    re.learn_batch(bs[0], capi.MODE_HOGWILD, False); bs[0].predictions()
    t0 = time.perf_counter()
    for i in range(9): re.learn_batch(bs[i % 3], capi.MODE_HOGWILD, False)
    bs[2].predictions(); dti = (time.perf_counter() - t0) / 9
    print(f"constant={const} zipf={zipf} ids={ids} lr={lr_on}: train {B/dt/1e6:.1f} M ex/s ({dt*1e3:.2f} ms), predict {B/dti/1e6:.1f} M ex/s")
    for b in bs: b.close()
    re.close()

run(True, 1.1, 100000)
run(False, 1.1, 100000)
run(False, 0.0001, 4000000)
run(True, 0.0001, 4000000)
run(False, 1.1, 100000, lr_on=False)
