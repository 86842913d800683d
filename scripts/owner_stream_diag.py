"""Diagnostic: the streaming and the step-synchronous owner-side apply on the same small SGD job, next to the plain fused kernel: where do they part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from fwumious_wabbit_amd.dist import DistGroup

n_ranks = int(os.environ.get("RANKS", 1))
n_ns, k, bits, ffm_bits = 6, 4, 20, 20
combos = [fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(n_ns)]
mi = fw.ModelInstance(learning_rate=0.01, ffm_learning_rate=0.01, bit_precision=bits, power_t=0.5, ffm_power_t=0.5, add_constant_feature=False,
                      feature_combo_descs=combos, ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(n_ns)], ffm_k=k, ffm_bit_precision=ffm_bits,
                      init_acc_gradient=1.0, ffm_init_acc_gradient=1.0, optimizer=fw.Optimizer.SGD)
fbt = fw.FeatureBufferTranslator(mi)
n_ex = 3000 - 3000 % n_ranks
recs, off = fw.synth_records(n_ns, 0.5, 0.0, 10_000_000, 0.3, 77, 0, 2 * n_ex)
per = n_ex // n_ranks
init = fw.Regressor(mi)
w0 = init.table_read(capi.TABLE_FFM_W)
# plain fused kernel, hogwild, two launches
b1 = init.record_batch(fbt, recs[: int(off[n_ex])], off[: n_ex + 1])
init.learn_batch(b1, capi.MODE_HOGWILD, True)
p_plain1 = b1.predictions().copy()
w_plain1 = init.table_read(capi.TABLE_FFM_W)
print("plain kernel, step 1: p range", p_plain1.min(), p_plain1.max(), "max |w - w0|", np.abs(w_plain1 - w0).max(), "floats moved", int((w_plain1 != w0).sum()))
init.close()
res = {}
for form in ("stream", "sync"):
    regs = [fw.Regressor(mi) for _ in range(n_ranks)]
    g = DistGroup(regs)
    g.set_mode(capi.MODE_HOGWILD)
    ps = []
    for step in range(2):
        rr, oo = [], []
        for j in range(n_ranks):
            a_, b_ = step * n_ex + j * per, step * n_ex + (j + 1) * per
            rr.append(recs[int(off[a_]):int(off[b_])])
            oo.append(off[a_:b_ + 1] - off[a_])
        outs = g.learn_owner_stream(fbt, rr, oo, log2_rows=7, log2_lr=7, consumer_workgroups=5 * n_ranks) if form == "stream" else g.learn_owner(fbt, rr, oo)
        ps.append(np.concatenate(outs))
        g.gather_tables()
        w = regs[0].table_read(capi.TABLE_FFM_W)
        lrt = regs[0].table_read(capi.TABLE_LR)
        print(form, "step", step + 1, ": p range", ps[-1].min(), ps[-1].max(), "max |w - w0|", float(np.abs(w - w0).max()), "floats moved", int((w != w0).sum()),
              "nan", int(np.isnan(w).sum()))
    res[form] = (ps, w, lrt)
    g.close()
    for r in regs:
        r.close()
print("step 1 predictions: max |stream - plain|", float(np.abs(res["stream"][0][0] - p_plain1).max()), " max |sync - plain|", float(np.abs(res["sync"][0][0] - p_plain1).max()))
print("step 2 predictions: max |stream - sync|", float(np.abs(res["stream"][0][1] - res["sync"][0][1]).max()))
d = np.abs(res["stream"][1] - res["sync"][1])
print("FFM tables stream vs sync: floats differing > 2e-6:", int((d > 2e-6).sum()), "max", float(d.max()))
d1 = np.abs(res["sync"][1] - w0); d2 = np.abs(res["stream"][1] - w0)
print("moved floats: sync", int((d1 > 0).sum()), "stream", int((d2 > 0).sum()), " both", int(((d1 > 0) & (d2 > 0)).sum()))

a, b = res["stream"][2], res["sync"][2]
print("LR table sizes", a.size, "even slots nonzero: stream", int(np.count_nonzero(a[0::2])), "sync", int(np.count_nonzero(b[0::2])), " odd slots: min/max stream", a[1::2].min(), a[1::2].max(), "sync", b[1::2].min(), b[1::2].max())
d = np.abs(a - b)
idx = np.argsort(-d)[:8]
print("largest LR differences (index, stream, sync):", [(int(i), float(a[i]), float(b[i])) for i in idx])
print("differing > 2e-6: even slots", int((d[0::2] > 2e-6).sum()), "odd slots", int((d[1::2] > 2e-6).sum()))
y = recs[off[:-1].astype(np.int64) + 1].astype(np.float64)
for form in ("stream", "sync"):
    p_ = np.concatenate(res[form][0]).astype(np.float64)
    want = np.zeros(1 << bits)
    for e in range(2 * n_ex):
        lrb = np.asarray(fbt.translate(recs[int(off[e]):int(off[e + 1])]).lr_buffer)
        np.add.at(want, lrb["hash"].astype(np.int64), -0.01 * (p_[e] - y[e]) * lrb["value"].astype(np.float64))
    got = res[form][2][0::2]
    dd = np.abs(got - want)
    print(form, "LR vs closed form: entries off by > 2e-6:", int((dd > 2e-6 + 2e-4 * np.abs(want)).sum()), "max", float(dd.max()))
