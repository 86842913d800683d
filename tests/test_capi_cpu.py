"""CPU-side checks of the product library: it loads, exports every symbol include/fwgpu.h declares, its
host-only integer paths (murmur3, record translation, synthetic stream) are bit-exact against the oracle and
the reference's KATs, and it refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from oracle import fwo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
    KATS = json.load(f)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fwgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(fwgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    L = C.CDLL(capi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert capi.lib().fwgpu_abi_version() == 1
    # the reference's serving FFI (lib.rs:150-236) under its own names
    ffi = open(os.path.join(ROOT, "include", "fw_ffi.h")).read()
    ffi = re.sub(r"/\*.*?\*/", "", ffi, flags=re.S)
    ffi_names = re.findall(r"\b(\w+)\s*\(", ffi)
    assert {"new_fw_predictor_prototype", "clone_lite", "fw_predict", "fw_predict_with_cache", "fw_setup_cache",
            "free_predictor", "fwgpu_predictor_predict_batch"} <= set(ffi_names)
    assert not [n for n in ffi_names if not hasattr(L, n)]


def test_struct_layouts_match_header():
    assert capi.LR_ENTRY.itemsize == 12 and capi.FFM_ENTRY.itemsize == 12  # feature_buffer.rs:10-22
    assert C.sizeof(capi.Config) == 17 * 4


def test_murmur3_kats_and_oracle_agreement():
    L = capi.lib()
    for c in KATS["hash"]["cases"]:
        seed = L.fwgpu_murmur3_32(c["ns"].encode(), len(c["ns"]), 0)
        assert L.fwgpu_murmur3_32(c["feature"].encode(), len(c["feature"]), seed) & 0x7FFFFFFF == c["hash"]
    O = fwo.lib()
    rng = np.random.default_rng(3)
    for _ in range(2000):
        n = int(rng.integers(0, 24))
        s = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        seed = int(rng.integers(0, 1 << 32))
        assert L.fwgpu_murmur3_32(s, n, seed) == O.fwo_murmur3_32(s, n, seed)


def test_hash_masks_match_oracle():
    L, O = capi.lib(), fwo.lib()
    for b in range(1, 32):
        assert L.fwgpu_lr_hash_mask(b) == O.fwo_lr_hash_mask(b)
        for k in (1, 2, 3, 4, 7, 8, 10, 16, 17, 32):
            assert L.fwgpu_ffm_hash_mask(b, k) == O.fwo_ffm_hash_mask(b, k)


@pytest.mark.parametrize("tc", KATS["translation"], ids=[t["name"] for t in KATS["translation"]])
def test_translation_kats(tc):
    mi = fw.ModelInstance(add_constant_feature=tc["add_constant_feature"], bit_precision=tc["bit_precision"],
                          ffm_k=tc["ffm_k"], ffm_bit_precision=tc["ffm_bit_precision"],
                          feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(n, bool(f)) for n, f in m], w)
                                               for m, w in tc["combos"]],
                          ffm_fields=[[fw.NamespaceDescriptor(n, bool(f)) for n, f in m] for m in tc["fields"]])
    fbt = fw.FeatureBufferTranslator(mi)
    for case in tc["cases"]:
        fb = fbt.translate(case["record"])
        assert [[int(e["hash"]), float(e["value"]), int(e["combo_index"])] for e in fb.lr_buffer] == case["lr"]
        assert [[int(e["hash"]), float(e["value"]), int(e["contra_field_index"])] for e in fb.ffm_buffer] == case["ffm"]
        assert fb.label == 1.0 and fb.example_importance == 1.0


def test_synth_stream_is_deterministic_and_well_formed():
    recs, off = fw.synth_records(30, 5.67, 1.05, 10_000_000, 0.1, 20240612, 0, 300)
    recs2, off2 = fw.synth_records(30, 5.67, 1.05, 10_000_000, 0.1, 20240612, 0, 300)
    assert np.array_equal(recs, recs2) and np.array_equal(off, off2)
    # any sub-range of the stream can be generated on its own
    recs3, off3 = fw.synth_records(30, 5.67, 1.05, 10_000_000, 0.1, 20240612, 100, 50)
    assert np.array_equal(recs3, recs[int(off[100]):int(off[150])])
    nnz = []
    for i in range(300):
        r = recs[int(off[i]):int(off[i + 1])]
        assert r[0] == len(r) and r[1] in (0, 1) and r[2] == 0x3F800000  # parser.rs:57-74
        cnt = 0
        for ns in range(30):
            t = int(r[3 + ns])
            if t & 0x80000000:
                s, e = (t >> 16) & 0x3FFF, t & 0xFFFF
                assert 33 <= s <= e <= len(r) and (e - s) % 2 == 0
                cnt += (e - s) // 2
            else:
                cnt += 1
        nnz.append(cnt)
    assert 180 < np.mean(nnz) < 220  # ~200 nnz per example (BASELINE.json config C)
    labels = recs[off[:-1].astype(np.int64) + 1]
    assert 0.2 < labels.mean() < 0.8


def test_translation_of_synthetic_records_is_bit_exact_vs_oracle():
    from helpers import make_pair

    mi, ocfg, ots = make_pair(12, 8, 20, 20, fw.Optimizer.AdagradLUT, interactions=[(0, 1), (2, 5)])
    fbt = fw.FeatureBufferTranslator(mi)
    recs, off = fw.synth_records(12, 2.0, 1.1, 100000, 0.3, 11, 0, 200)
    for i in range(200):
        r = recs[int(off[i]):int(off[i + 1])]
        fb = fbt.translate(r)
        lr, ffm, label, imp = ots.translate(r)
        assert fb.lr_buffer.tobytes() == lr.tobytes()
        assert fb.ffm_buffer.tobytes() == ffm.tobytes()
        assert fb.label == label and fb.example_importance == imp


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.FwgpuError) as ei:
        fw.Regressor(fw.ModelInstance())
    assert ei.value.code == 2  # FWGPU_ERR_DEVICE


def test_bad_config_is_rejected_before_touching_the_device():
    mi = fw.ModelInstance(ffm_k=8, ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(80)])  # 8*80*80 > 41472
    with pytest.raises(capi.FwgpuError) as ei:
        fw.Regressor(mi)
    assert ei.value.code == 1 and "FFM_CONTRA_BUF_LEN" in str(ei.value)  # block_ffm.rs:96-101


def test_host_translator_survives_garbage_records():
    """FeatureBufferTranslator::translate on the host: corrupted record words (slot offsets pointing anywhere, wrong lengths)
    must be rejected or translated within bounds, never crash (the reference indexes unchecked: feature_buffer.rs:47-108)."""
    from helpers import make_pair

    mi, _, ots = make_pair(8, 4, 16, 16, fw.Optimizer.AdagradLUT, interactions=[(0, 1), (2, 3)])
    fbt = fw.FeatureBufferTranslator(mi)
    recs, off = fw.synth_records(8, 2.0, 1.1, 1000, 0.5, 5, 0, 50)
    rng = np.random.default_rng(7)
    ok = bad = 0
    for trial in range(3000):
        i = int(rng.integers(0, 50))
        r = recs[int(off[i]):int(off[i + 1])].copy()
        for _ in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 4))
            pos = int(rng.integers(0, len(r)))
            if kind == 0:
                r[pos] = rng.integers(0, 2 ** 32, dtype=np.uint64).astype(np.uint32)
            elif kind == 1:
                r[pos] ^= np.uint32(1) << np.uint32(rng.integers(0, 32))
            elif kind == 2:
                r = r[: max(1, pos)]
            else:
                if len(r) > 11:
                    r[3 + int(rng.integers(0, 8))] = np.uint32(0x80000000 | (int(rng.integers(0, 0x4000)) << 16) | int(rng.integers(0, 0x10000)))
        try:
            fb = fbt.translate(r)
            assert len(fb.lr_buffer) <= 8192 and len(fb.ffm_buffer) <= 8192
            ok += 1
        except capi.FwgpuError:
            bad += 1
    assert ok > 100 and bad > 100, (ok, bad)
