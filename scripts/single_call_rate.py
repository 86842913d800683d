"""Latency of the per-example entry points (Regressor::learn / predict, regressor.rs:356-395): one example per call."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fwumious_wabbit_amd as fw
from helpers import make_pair
mi, ocfg, ots = make_pair(30, 8, 24, 24, fw.Optimizer.AdagradLUT, lr=0.025, ffm_lr=0.025)
re = fw.Regressor(mi)
fbt = fw.FeatureBufferTranslator(mi)
recs, off = fw.synth_records(30, 5.67, 1.05, 1_000_000, 0.1, 5, 0, 600)
fbs = [fbt.translate(recs[int(off[i]):int(off[i + 1])]) for i in range(600)]
for name, fn in (("learn", lambda fb: re.learn(fb, True)), ("predict", lambda fb: re.predict(fb))):
    for fb in fbs[:50]:
        fn(fb)
    t0 = time.perf_counter()
    for fb in fbs[50:550]:
        fn(fb)
    print(f"{name}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call (~{len(fbs[100].ffm_buffer)} FFM / {len(fbs[100].lr_buffer)} LR entries)")
