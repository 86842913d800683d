"""Rates of the rows either side of the hot path, config C shaped data:
text parser (lines/s, MB/s), cache write / read (uncompressed and LZ4), model save/load, serving FFI latency and the batched call."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import capi, persistence as P
from fwumious_wabbit_amd.feed import RecordCache, VowpalParser, VwNamespaceMap

F = 30
gpu = "--no-gpu" not in sys.argv
vw = VwNamespaceMap("".join(f"A{i},ns{i}\n" for i in range(F)))
rng = np.random.default_rng(1)
N = 20000
lines = []
for i in range(N):
    parts = ["1" if rng.random() < 0.3 else "-1"]
    for ns in range(F):
        k = 1 + rng.poisson(5.67)
        parts.append(f"|A{ns} " + " ".join(f"{rng.integers(0, 10_000_000)}" + (f":{0.5 + 1.5 * rng.random():.3f}" if rng.random() < 0.1 else "") for _ in range(k)))
    lines.append(" ".join(parts) + "\n")
text = "".join(lines).encode()
p = VowpalParser(vw)
t0 = time.perf_counter(); words, off, used, rc = p.parse_buffer(text); dt = time.perf_counter() - t0
assert rc == 0 and len(off) == N + 1
print(f"parser: {N / dt:,.0f} lines/s, {len(text) / dt / 1e6:,.0f} MB/s of text, {words.nbytes / dt / 1e6:,.0f} MB/s of records ({words.nbytes / N:.0f} B/record, 1 thread)")
d = tempfile.mkdtemp()
for gz in (False, True):
    inp = os.path.join(d, "t.vw.gz" if gz else "t.vw")
    rc_ = RecordCache(inp, True, vw)
    t0 = time.perf_counter()
    for r in range(8):
        rc_.push_records(words)
    rc_.write_finish(); dtw = time.perf_counter() - t0; rc_.close()
    size = os.path.getsize(inp + ".fwcache")
    L = capi.lib(); import ctypes as C
    buf = np.zeros(1 << 24, dtype=np.uint32); ro = np.zeros((1 << 18) + 1, dtype=np.uint64); nr, nw = C.c_uint64(), C.c_uint64()
    for rep in range(2):  # second pass: file in the page cache (the first one waits for the write-back of what we just wrote)
        rc_ = RecordCache(inp, True, vw)
        t0 = time.perf_counter(); n = 0
        while True:
            capi.check(L.fwgpu_cache_next_records(rc_.h, capi.ptr(buf), buf.size, capi.ptr(ro), 1 << 18, C.byref(nr), C.byref(nw)))
            if nr.value == 0: break
            n += nr.value
        dtr = time.perf_counter() - t0; rc_.close()
    print(f"cache {'lz4' if gz else 'raw'}: write {8 * words.nbytes / dtw / 1e6:,.0f} MB/s, read {8 * words.nbytes / dtr / 1e6:,.0f} MB/s = {n / dtr:,.0f} records/s, file {size / (8 * words.nbytes):.2f} of raw")
if gpu:
    from fwumious_wabbit_amd.serving import Predictor
    mi = fw.ModelInstance(learning_rate=0.025, ffm_learning_rate=0.025, power_t=0.38, ffm_power_t=0.38, bit_precision=24, ffm_k=8,
                          ffm_bit_precision=24, optimizer=fw.Optimizer.AdagradLUT, ffm_init_acc_gradient=1.0,
                          feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(F)],
                          ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(F)])
    re = fw.Regressor(mi)
    b = re.record_batch(fw.FeatureBufferTranslator(mi), words, off)
    re.learn_batch(b, capi.MODE_HOGWILD, True); b.predictions()
    path = os.path.join(d, "m.fw")
    t0 = time.perf_counter(); P.save_regressor_to_filename(path, mi, vw, re); dts = time.perf_counter() - t0
    sz = os.path.getsize(path)
    t0 = time.perf_counter(); P.convert_inference_regressor(path, path + ".inf"); dtc = time.perf_counter() - t0
    t0 = time.perf_counter(); pr = Predictor(f"fw -i {path}.inf -t"); dtl = time.perf_counter() - t0
    print(f"model file {sz / 1e6:.0f} MB: save {sz / dts / 1e6:,.0f} MB/s, convert {sz / dtc / 1e6:,.0f} MB/s, load as predictor {dtl:.2f} s")
    req = [l.split(" ", 1)[1] for l in lines[:2000]]
    pr.predict(req[0])
    t0 = time.perf_counter()
    for r in req[:500]: pr.predict(r)
    dt1 = (time.perf_counter() - t0) / 500
    pr.predict_batch(req[:64])
    for nb in (64, 512, 2000):
        t0 = time.perf_counter(); pr.predict_batch(req[:nb]); dtb = time.perf_counter() - t0
        print(f"predict_batch({nb}): {dtb * 1e3:.2f} ms = {nb / dtb:,.0f} predictions/s")
    print(f"fw_predict (one request per call, ~200 features): {dt1 * 1e6:.0f} us per call")
if gpu:
    big = text * 24  # 480 000 lines, ~885 MB of text (long enough that the trainer's staging buffers have stopped growing)
    for th in (1, 8, 16, 32):
        re2 = fw.Regressor(mi)
        tr = fw.HogwildTrainer(re2, mi, micro_batch=16384)
        tr.digest_text(p, text, threads=th)  # warm-up: the trainer's pinned staging and device buffers are allocated on first use
        tr.block_until_workers_finished()
        t0 = time.perf_counter()
        n, used, rc = tr.digest_text(p, big, threads=th)
        tr.block_until_workers_finished()
        dt = time.perf_counter() - t0
        print(f"text -> trainer, native, {th} parser threads: {n / dt:,.0f} lines/s ({len(big) / dt / 1e6:,.0f} MB/s of text)")
        tr.close(); re2.close()
