"""Pins the CPU oracle (oracle/fw_oracle.c) against the reference's own known-answer tests,
transcribed as data in tests/golden/reference_kats.json (builder: make_reference_kats.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import fwo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")
with open(GOLD) as f:
    KATS = json.load(f)

f32 = np.float32


def _cfg(d, wiring):
    return fwo.make_config(wiring=fwo.WIRING_FFM_ONLY if wiring == "ffm_only" else fwo.WIRING_REGRESSOR, **d)


def run_scenario(sc, model_factory):
    """Yields (step, got) for each step; model_factory(cfg, ffm_fill) -> object with learn/predict/forward_backward."""
    m = model_factory(sc)
    for st in sc["steps"]:
        lr = fwo.lr_entries(st["lr"])
        ffm = fwo.ffm_entries(st["ffm"])
        if st["op"] == "learn":
            got = m.learn(lr, ffm, st["label"], st["importance"], st["update"])
        elif st["op"] == "forward_backward":
            got = m.forward_backward(lr, ffm, st["label"], st["importance"], st["update"])
        else:
            got = m.predict(lr, ffm)
        yield st, got


def oracle_factory(sc):
    m = fwo.Model(_cfg(sc["config"], sc["wiring"]))
    if "ffm_fill" in sc:
        m.ffm_fill(sc["ffm_fill"])
    return m


@pytest.mark.parametrize("sc", KATS["scenarios"], ids=[s["name"] for s in KATS["scenarios"]])
def test_scenario(sc):
    for i, (st, got) in enumerate(run_scenario(sc, oracle_factory)):
        want = st.get("current_code", st["expect"]) if st.get("stale") else st["expect"]
        if st["cmp"] == "eq":
            assert f32(got) == f32(want), f"{sc['name']} step {i} ({st['op']}): got {got!r} want {want!r}"
        else:
            assert abs(got - want) < 5e-6, f"{sc['name']} step {i} ({st['op']}): got {got!r} want {want!r}"


def test_stale_assertions_are_documented():
    stale = [(s["name"], st) for s in KATS["scenarios"] for st in s["steps"] if st.get("stale")]
    assert len(stale) == 1 and stale[0][0] == "test_ffm_missing_field"
    assert "note" in stale[0][1]


def test_optimizer_kats():
    L = fwo.lib()
    o = KATS["optimizer"]
    assert f32(L.fwo_step_sgd(0.15, 0.1)) == f32(0.1) * f32(0.15)
    for c in o["flex"]:
        acc = C.c_float(c["acc"])
        p = L.fwo_step_flex(c["lr"], -c["power_t"], c["g"], C.byref(acc))
        if c["expect"] is not None:
            assert f32(p) == f32(c["expect"])
        want_acc = {"0.9+0.1*0.1": f32(0.9) + f32(0.1) * f32(0.1), "0.1*0.1": f32(0.1) * f32(0.1), "0.0": f32(0)}[c["acc_expr"]]
        assert f32(acc.value) == want_acc
    for c in o["lut"]:
        lut = np.zeros(2048, dtype=np.float32)
        L.fwo_lut_init(lut.ctypes.data_as(C.POINTER(C.c_float)), c["lr"], c["power_t"], c["init_acc"])
        acc = C.c_float(c["acc"])
        p = L.fwo_step_lut(lut.ctypes.data_as(C.POINTER(C.c_float)), c["g"], C.byref(acc))
        assert f32(p) == f32(c["expect"])
        want_acc = {"0.9+0.1*0.1": f32(0.9) + f32(0.1) * f32(0.1), "0.1*0.1": f32(0.1) * f32(0.1), "0.0": f32(0)}[c["acc_expr"]]
        assert f32(acc.value) == want_acc
    cmp_ = o["comparison"]
    lut = np.zeros(2048, dtype=np.float32)
    lp = lut.ctypes.data_as(C.POINTER(C.c_float))
    L.fwo_lut_init(lp, cmp_["lr"], cmp_["power_t"], cmp_["init_acc"])
    for g in cmp_["gradients"]:
        for a in cmp_["accumulations"]:
            a1, a2 = C.c_float(a), C.c_float(a)
            pf = L.fwo_step_flex(cmp_["lr"], -cmp_["power_t"], g, C.byref(a1))
            pl = L.fwo_step_lut(lp, g, C.byref(a2))
            err = abs(pf - pl)
            rel = err / abs(pl) if pl != 0 else err
            assert rel < cmp_["max_rel_err"]


def test_triangle_kat():
    L = fwo.lib()
    t = KATS["triangle"]
    fp = C.POINTER(C.c_float)
    inp = np.array(t["input"], dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    L.fwo_triangle_forward(inp.ctypes.data_as(fp), t["width"], out.ctypes.data_as(fp))
    assert out.tolist() == t["forward"]
    gin = np.zeros(4, dtype=np.float32)
    L.fwo_triangle_backward(out.ctypes.data_as(fp), t["width"], gin.ctypes.data_as(fp))
    assert gin.tolist() == t["backward"]


@pytest.mark.parametrize("tc", KATS["translation"], ids=[t["name"] for t in KATS["translation"]])
def test_translation_kat(tc):
    combos = [([tuple(m) for m in members], w) for members, w in tc["combos"]]
    fields = [[tuple(m) for m in members] for members in tc["fields"]]
    ts = fwo.TranslatorSpec(combos, fields, tc["add_constant_feature"], tc["bit_precision"], tc["ffm_k"],
                            tc["ffm_bit_precision"])
    for case in tc["cases"]:
        lr, ffm, label, imp = ts.translate(case["record"])
        assert [[int(e["hash"]), float(e["value"]), int(e["combo_index"])] for e in lr] == case["lr"]
        assert [[int(e["hash"]), float(e["value"]), int(e["contra_field_index"])] for e in ffm] == case["ffm"]
        assert label == 1.0 and imp == 1.0


def test_hash_kats():
    L = fwo.lib()
    for c in KATS["hash"]["cases"]:
        seed = L.fwo_murmur3_32(c["ns"].encode(), len(c["ns"]), 0)
        h = L.fwo_murmur3_32(c["feature"].encode(), len(c["feature"]), seed) & 0x7FFFFFFF
        assert h == c["hash"], c


def test_hash_masks():
    L = fwo.lib()
    # feature_buffer.rs:138-148; SURVEY 8(a1): k=10 clears 4 low bits, k=4 -> 2, k=8 -> 3, k=16 -> 4, k=1 -> 0
    assert L.fwo_lr_hash_mask(18) == (1 << 18) - 1
    assert L.fwo_ffm_hash_mask(18, 1) == (1 << 18) - 1
    assert L.fwo_ffm_hash_mask(18, 3) == ((1 << 18) - 1) ^ 3
    assert L.fwo_ffm_hash_mask(22, 4) == ((1 << 22) - 1) ^ 3
    assert L.fwo_ffm_hash_mask(28, 8) == ((1 << 28) - 1) ^ 7
    assert L.fwo_ffm_hash_mask(18, 10) == ((1 << 18) - 1) ^ 15
    assert L.fwo_ffm_hash_mask(28, 16) == ((1 << 28) - 1) ^ 15


def test_merand48_properties():
    """merand48 is parity-UNPINNED (no reference test observes an init weight); check the published
    LCG's structural properties and the init formula's range (block_ffm.rs:797-806)."""
    L = fwo.lib()
    vals = np.array([L.fwo_merand48(i) for i in range(0, 200000, 7)], dtype=np.float32)
    assert vals.min() >= 0.0 and vals.max() < 1.0
    assert 0.49 < vals.mean() < 0.51
    # one LCG step, by hand
    s = (0xEECE66D5DEECE66D * 12345 + 2147483647) & 0xFFFFFFFFFFFFFFFF
    want = np.array([((s >> 25) & 0x7FFFFF) | (127 << 23)], dtype=np.uint32).view(np.float32)[0] - f32(1.0)
    assert f32(L.fwo_merand48(12345)) == want
    m = fwo.Model(fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, ffm_k=4, ffm_num_fields=3, ffm_bit_precision=10))
    w = m.ffm_weights
    assert len(w) == 1024 + 12
    bound = 0.5 / np.sqrt(4.0) / 50.0
    assert np.abs(w).max() <= bound + 1e-9 and np.abs(w).max() > 0.5 * bound
    assert np.all(m.ffm_acc == 0.0)


# ------------------------------------------------------------------ deep head (a18)
def _layer_fb(opt, lr, w, acc, x, out_grad, n_in, n_out, update=True):
    L = fwo.lib()
    fp = C.POINTER(C.c_float)
    y = np.zeros(n_out, dtype=np.float32)
    L.fwo_neuron_layer_fb(opt, lr, 0.0, 0.0, w.ctypes.data_as(fp), acc.ctypes.data_as(fp), n_in, n_out,
                          x.ctypes.data_as(fp), y.ctypes.data_as(fp), out_grad.ctypes.data_as(fp), int(update))
    return y


def test_neuron_layer_kats():
    # block_neural.rs:507-537 test_simple: SGD lr 0.1, One-init (w=1, bias=0), input 2.0, upstream gradient 1.0
    w, acc = np.array([1.0, 0.0], dtype=np.float32), np.zeros(2, dtype=np.float32)
    x = np.array([2.0], dtype=np.float32)
    assert abs(_layer_fb(fwo.OPT_SGD, 0.1, w, acc, x, np.ones(1, np.float32), 1, 1)[0] - 2.0) < 5e-6
    x = np.array([2.0], dtype=np.float32)
    assert abs(_layer_fb(fwo.OPT_SGD, 0.1, w, acc, x, np.ones(1, np.float32), 1, 1)[0] - 1.5) < 5e-6
    # block_neural.rs:539-581 test_two_neurons: both neurons output 2.0, then 1.5 with update=false
    w, acc = np.array([1.0, 1.0, 0.0, 0.0], dtype=np.float32), np.zeros(4, dtype=np.float32)
    x = np.array([2.0], dtype=np.float32)
    y = _layer_fb(fwo.OPT_SGD, 0.1, w, acc, x, np.ones(2, np.float32), 1, 2)
    assert y.tolist() == [2.0, 2.0]
    assert x[0] == 2.0  # "on tape 0 input of 2.0 will be replaced with the gradient of 2.0" (1*1 + 1*1)
    x = np.array([2.0], dtype=np.float32)
    y = _layer_fb(fwo.OPT_SGD, 0.1, w, acc, x, np.ones(2, np.float32), 1, 2, update=False)
    assert abs(y[0] - 1.5) < 5e-6 and abs(y[1] - 1.5) < 5e-6


def test_deep_head_reduces_to_plain_regressor_when_transparent():
    """Structural check of the a18 wiring: with topology 'one', a hidden layer of zero weights and the final neuron's
    One-init (regressor.rs:312-319), the logit is 0*h + 1*x summed = the plain LR+FFM logit, so the first prediction
    equals the plain regressor's; and learning then changes the head's weights."""
    cfgk = dict(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=0.1, ffm_learning_rate=0.1, bit_precision=10,
                num_combos=3, ffm_k=4, ffm_bit_precision=10, ffm_num_fields=3, ffm_init_acc_gradient=1.0)
    lr = fwo.lr_entries([(5, 1.0, 0), (9, 2.0, 1), (11650396 & 1023, 1.0, 2)])
    ffm = fwo.ffm_entries([(8, 1.0, 0), (40, 2.0, 4), (100, 1.0, 8)])
    plain = fwo.Model(fwo.make_config(**cfgk))
    plain.lr_table[:] = np.linspace(-0.3, 0.3, plain.lr_table.size, dtype=np.float32)
    nn = fwo.make_nn_config([(6, "relu", "zero")], topology="one", nn_learning_rate=0.05, nn_power_t=0.0)
    deep = fwo.Model(fwo.make_config(**cfgk), nn=nn)
    deep.lr_table[:] = plain.lr_table
    assert np.array_equal(deep.ffm_weights, plain.ffm_weights)
    p0 = plain.predict(lr, ffm)
    assert abs(deep.predict(lr, ffm) - p0) < 1e-6
    assert abs(deep.learn(lr, ffm, 1.0, 1.0, True) - p0) < 1e-6
    assert deep.nn_weights(1).size == 6 + (3 + 6) + 1  # final neuron: h(6) + x(3 combos + 6 triangle) + bias
    assert not np.all(deep.nn_weights(1)[:-1] == 1.0)   # the One-initialised final neuron has learned
    p1 = deep.predict(lr, ffm)
    assert p1 > p0                                       # label 1: the prediction moved up
