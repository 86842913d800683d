"""Serving with the context cache: candidates/s of one request = context (20 of 30 namespaces) + N candidates
(the other 10 namespaces), through (a) the concatenation route (every candidate scored as context + candidate, whole line
translated on the device) and (b) the device-side context cache (fw_setup_cache once, candidates reduced to the entries the
cache does not cover).  Also the kernel-only rates of both on entry batches (no text parsing)."""
import sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi, persistence as P
from fwumious_wabbit_amd.feed import VwNamespaceMap
from fwumious_wabbit_amd.serving import Predictor

F, K, NCTX = 30, 8, 20
N = int(os.environ.get("N", 2000))
vw = VwNamespaceMap("".join(f"N{i:02d},ns{i}\n" for i in range(F)))
mi = fw.ModelInstance(learning_rate=0.025, ffm_learning_rate=0.025, power_t=0.38, ffm_power_t=0.38, bit_precision=24, ffm_k=K,
                      ffm_bit_precision=24, optimizer=fw.Optimizer.AdagradLUT, ffm_init_acc_gradient=1.0,
                      feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(F)],
                      ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(F)])
re = fw.Regressor(mi)
recs, off = fw.synth_records(F, 5.67, 1.05, 1_000_000, 0.1, 5, 0, 20000)
b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
re.learn_batch(b, capi.MODE_HOGWILD, True)
b.predictions()
rng = np.random.default_rng(1)


def ns_text(i):
    n = 1 + rng.poisson(5.67)
    return f"|N{i:02d} " + " ".join(f"f{rng.zipf(1.3) % 100000}" for _ in range(n))


ctx = " ".join(ns_text(i) for i in range(NCTX)) + " "
cands = [" ".join(ns_text(i) for i in range(NCTX, F)) + "\n" for _ in range(N)]
d = tempfile.mkdtemp()
path = os.path.join(d, "m.fw")
P.save_regressor_to_filename(path, mi, vw, re)
P.convert_inference_regressor(path, path + ".inf")
pr = Predictor(f"fw -i {path}.inf -t")
full = [ctx + c for c in cands]
want = pr.predict_batch(full)
t0 = time.perf_counter(); assert pr.setup_cache(ctx + "\n") == 0.0; t_setup = time.perf_counter() - t0
got = pr.predict_batch(cands, with_cache=True)
print(f"context {len(ctx.split()) - NCTX} features, candidates ~{np.mean([len(c.split()) - (F - NCTX) for c in cands]):.0f} features, N = {N}; "
      f"max |cached - concatenated| = {np.abs(got - want).max():.2e}; fw_setup_cache {t_setup * 1e6:.0f} us")
full_c, cands_c = pr.encode_batch(full), pr.encode_batch(cands)  # the char** of a C / Rust caller (Python's str -> bytes is not the library's time)
for name, fn in (("concatenation (whole lines, device translation)", lambda: pr.predict_batch(full_c)),
                 ("device context cache (context scanned once, records, kernel skips the cached namespaces)", lambda: pr.predict_batch(cands_c, with_cache=True))):
    fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print(f"text route, {name}: {np.median(ts) * 1e3:.2f} ms per request = {N / np.median(ts):,.0f} candidates/s")
enc = [t.encode() for t in full[:300]]
pr.L.fw_predict(pr.p, enc[0])
t0 = time.perf_counter()
for t in enc:
    pr.L.fw_predict(pr.p, t)
print(f"fw_predict, one whole line per call: {(time.perf_counter() - t0) / 300 * 1e6:.0f} us per call")
t0 = time.perf_counter()
for c in cands[:300]:
    pr.predict_with_cache(c)
print(f"fw_predict_with_cache, one candidate per call: {(time.perf_counter() - t0) / 300 * 1e6:.0f} us per call")

# kernel-only: entry batches (no text), predict-only launches
from fwumious_wabbit_amd.feed import VowpalParser
parser = VowpalParser(vw)
fbt = fw.FeatureBufferTranslator(mi)
re2 = pr_re = fw.Regressor(mi)
re2.overwrite_weights_from_buf(re.write_weights_to_buf())
fbs = [fbt.translate(parser.next_vowpal(l.encode())) for l in full]
cfb = fbt.translate(parser.next_vowpal((ctx + "\n").encode()))
cache = re2.setup_cache(cfb)
cut = [fw.FeatureBuffer(label=0.0, example_importance=1.0, example_number=0, lr_buffer=f.lr_buffer, ffm_buffer=cache.filter(f.ffm_buffer)) for f in fbs]
bf, bc = re2.batch(fbs), re2.batch(cut)
bc.set_cache(cache)
for name, bb in (("whole examples", bf), ("context cache + uncovered entries", bc)):
    re2.learn_batch(bb, capi.MODE_HOGWILD, False); bb.predictions()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(10):
            re2.learn_batch(bb, capi.MODE_HOGWILD, False)
        bb.predictions(); ts.append((time.perf_counter() - t0) / 10)
    print(f"kernel only, {name}: {np.median(ts) * 1e6:.0f} us per launch of {N} = {N / np.median(ts) / 1e6:.2f} M candidates/s")
print("max |kernel cached - whole| =", float(np.abs(bc.predictions() - bf.predictions()).max()))
