"""GPU parity tests (run with -m gpu on the MI355X box).  Everything goes through the C ABI (libfwgpu.so);
the CPU oracle is only the checker.

Parity bar (BASELINE.json north_star): hash indices bit-exact; per-example log-loss within 1e-4 of the
reference CPU path on identical input.  Tolerances used here are written next to each assert.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import logloss, make_pair, mi_from_cfg, record_labels
from oracle import fwo

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
    KATS = json.load(f)

PRED_TOL = 1e-5      # |p_gpu - p_ref| on a single prediction (f32 summation-order noise is ~1e-7)
LOGLOSS_TOL = 1e-4   # north_star: per-example log-loss tolerance
# (statistical hold-out comparisons of the concurrent modes live in test_zz_gpu_hogwild_quality.py, which sorts last)


# ------------------------------------------------------------------ the reference's own KATs, on the GPU
@pytest.mark.parametrize("sc", KATS["scenarios"], ids=[s["name"] for s in KATS["scenarios"]])
def test_reference_kat_scenarios_on_gpu(sc):
    mi = mi_from_cfg(sc["config"], sc["wiring"])
    re = fw.Regressor(mi)
    if "ffm_fill" in sc:
        re.ffm_fill(sc["ffm_fill"])
    for i, st in enumerate(sc["steps"]):
        fb = fw.lr_and_ffm_vec(st["lr"], st["ffm"], st["label"], st["importance"])
        if st["op"] == "predict" or not st["update"]:
            got = re.predict(fb) if st["op"] == "predict" else re.learn(fb, None, False)
        else:
            got = re.learn(fb, None, True)
        want = st.get("current_code", st["expect"]) if st.get("stale") else st["expect"]
        assert abs(got - want) < PRED_TOL, f"{sc['name']} step {i} ({st['op']}): got {got!r} want {want!r}"
    re.close()


def test_smoke_entry():
    import __graft_entry__ as g

    g.smoke()


# ------------------------------------------------------------------ streams: sequential mode == reference single thread
def _stream_parity(n_ns, k, bits, ffm_bits, optimizer, n, mean_extra, p_weighted, ids, seed, interactions=(),
                   weight_tol=2e-5, whole_lines=None, lds_keep=None, kept_rows=None, **kw):
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, optimizer, interactions=interactions, **kw)
    recs, off = fw.synth_records(n_ns, mean_extra, 1.1, ids, p_weighted, seed, 0, n)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg)
    _, p_ref = om.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
    results = []
    # both ways of feeding the device: entries translated on the host, and raw records translated inside the kernel
    for kind in ("entries", "records"):
        re = fw.Regressor(mi)
        if whole_lines is not None:
            re.set_whole_line_updates(whole_lines)
        if lds_keep is not None:
            re.set_lds_keep(lds_keep)
        if kept_rows is not None:
            re.set_kept_rows(kept_rows)
        fbt = fw.FeatureBufferTranslator(mi)
        b = re.batch_from_records(fbt, recs, off) if kind == "entries" else re.record_batch(fbt, recs, off)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        p_gpu = b.predictions()
        d_ll = np.abs(logloss(p_gpu, y) - logloss(p_ref, y))
        assert d_ll.max() < LOGLOSS_TOL, f"{kind}: max per-example |d logloss| = {d_ll.max()} at {d_ll.argmax()}"
        assert np.abs(p_gpu - p_ref).max() < 5e-5
        # final tables: same state as the reference after the whole stream
        # (weights are O(0.1); accumulators grow to O(100), hence the relative term)
        def close(a, b_):
            return bool(np.all(np.abs(a - b_) <= weight_tol + 1e-5 * np.abs(b_)))

        assert close(re.table_read(capi.TABLE_LR), om.lr_table)
        if k:
            assert close(re.table_read(capi.TABLE_FFM_W), om.ffm_weights)
            acc_g, acc_o = re.table_read(capi.TABLE_FFM_ACC), om.ffm_acc
            assert close(acc_g, acc_o)
            # exactly the same set of accumulators was touched
            a0 = np.float32(mi.ffm_init_acc_gradient if optimizer == fw.Optimizer.AdagradFlex else 0.0)
            assert np.array_equal(acc_g != a0, acc_o != a0)
        results.append((p_gpu, [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]))
        b.close()
        re.close()
    # device-side translation == host translation, bit for bit (hash indices and everything downstream)
    assert np.array_equal(results[0][0], results[1][0]) and results[0][1] == results[1][1]
    return results[0][0], p_ref


def test_sequential_stream_config_b_like_with_collisions():
    # 10 fields, k=4 (16-byte vector path), tiny tables: heavy hash collisions and overlapping FFM rows
    _stream_parity(10, 4, 12, 12, fw.Optimizer.AdagradLUT, n=1500, mean_extra=0.0, p_weighted=0.0, ids=3000, seed=1)


def test_sequential_stream_config_c_like():
    # 30 fields, k=8, ~200 nnz per example, weighted features, LR interactions
    _stream_parity(30, 8, 18, 18, fw.Optimizer.AdagradLUT, n=300, mean_extra=5.67, p_weighted=0.1, ids=100000, seed=2,
                   interactions=[(0, 1), (3, 7)])


_PARKED_CASES = {
    "lut_lines": dict(args=(30, 8, 18, 18, fw.Optimizer.AdagradLUT), kw=dict(n=200, mean_extra=5.67, p_weighted=0.1, ids=100000, seed=61, whole_lines=2)),
    "lut_collisions": dict(args=(30, 8, 16, 16, fw.Optimizer.AdagradLUT), kw=dict(n=150, mean_extra=5.67, p_weighted=0.1, ids=3000, seed=62, whole_lines=3)),
    "flex": dict(args=(30, 8, 16, 16, fw.Optimizer.AdagradFlex),
                 kw=dict(n=150, mean_extra=6.5, p_weighted=0.2, ids=20000, seed=63, init_acc=1.0, ffm_init_acc=1.0, weight_tol=5e-5, whole_lines=2)),
    "sgd": dict(args=(30, 8, 16, 16, fw.Optimizer.SGD), kw=dict(n=150, mean_extra=6.5, p_weighted=0.2, ids=20000, seed=64, lr=0.05, ffm_lr=0.05, whole_lines=3)),
}


def test_large_table_kernel_without_kept_rows_in_order():
    """fwgpu_debug_set_option 13 = 0 (round 6): the large-table kernel with NO rows kept from the gather -- its own instantiation, every row re-read by the update.  In order it is the
    reference like every other path: per example and on the final tables, three optimizers, collisions and repeated hashes included."""
    _stream_parity(30, 8, 16, 16, fw.Optimizer.AdagradLUT, n=150, mean_extra=5.67, p_weighted=0.1, ids=3000, seed=62, whole_lines=3, kept_rows=0)
    _stream_parity(30, 8, 17, 17, fw.Optimizer.AdagradLUT, n=60, mean_extra=7.6, p_weighted=0.1, ids=2000, seed=76, whole_lines=2, kept_rows=0)
    _stream_parity(30, 8, 16, 16, fw.Optimizer.AdagradFlex, n=100, mean_extra=6.5, p_weighted=0.2, ids=20000, seed=63, init_acc=1.0, ffm_init_acc=1.0, weight_tol=5e-5, whole_lines=3, kept_rows=0)
    _stream_parity(30, 8, 16, 16, fw.Optimizer.SGD, n=100, mean_extra=6.5, p_weighted=0.2, ids=20000, seed=64, lr=0.05, ffm_lr=0.05, whole_lines=3, kept_rows=0)


def test_parked_rows_at_every_share_size_around_the_register_capacity():
    """~150 ... ~260 features per example over 8 waves = 18 ... 33 rows per wave: shares that end below, at and beyond the 20 register slots and the
    parked slots, with repeated hashes (ids = 2000 on a 17-bit table) falling on kept, parked and re-read positions alike."""
    for i, mean_extra in enumerate((4.0, 4.6, 5.0, 5.4, 6.0, 6.6, 7.6)):
        _stream_parity(30, 8, 17, 17, fw.Optimizer.AdagradLUT, n=60, mean_extra=mean_extra, p_weighted=0.1, ids=2000, seed=70 + i, whole_lines=3 if i % 2 else 2)


@pytest.mark.parametrize("case", list(_PARKED_CASES))
@pytest.mark.parametrize("rows", [0, 1, 3])
def test_rows_parked_in_lds_for_the_update_are_exact_in_order(rows, case):
    """Rows of a wave's share beyond the 20 register-kept ones keep their gather-time weights in LDS (option 8) instead of being re-read by the
    update phase.  ~200 features per example = ~25 rows per wave: with 0 / 1 / 3 parked rows per wave the in-order mode must stay the
    reference's single thread, with repeated and overlapping rows (small tables) taking the chained path as before, for the three optimizers."""
    c = _PARKED_CASES[case]
    _stream_parity(*c["args"], lds_keep=rows, **c["kw"])


@pytest.mark.parametrize("whole_lines", [0, 2])
def test_sequential_stream_whole_line_row_updates(whole_lines):
    """The whole-line ("window") update path of the v2 kernel (forced on: these tables are far smaller than the Infinity
    Cache, where it is off by default) must be exact in the in-order mode: rows whose 128 B line spans intersect are
    serialised, neighbouring weights pass through bit for bit.  Tiny tables: heavy collisions, overlaps and shared lines;
    k = 8 rows with all four start phases (0/32/64/96 B into a line, the last one spilling into a ninth line), k = 4 rows
    of 2-3 lines, k = 16 single-field rows."""
    _stream_parity(30, 8, 18, 18, fw.Optimizer.AdagradLUT, n=300, mean_extra=5.67, p_weighted=0.1, ids=100000, seed=41,
                   whole_lines=whole_lines)
    _stream_parity(30, 8, 15, 15, fw.Optimizer.AdagradLUT, n=200, mean_extra=2.0, p_weighted=0.1, ids=3000, seed=42,
                   whole_lines=whole_lines)
    _stream_parity(10, 4, 12, 12, fw.Optimizer.AdagradLUT, n=1500, mean_extra=0.0, p_weighted=0.0, ids=3000, seed=43,
                   whole_lines=whole_lines)
    _stream_parity(12, 16, 14, 14, fw.Optimizer.AdagradFlex, n=400, mean_extra=1.0, p_weighted=0.2, ids=2000, seed=44,
                   init_acc=1.0, ffm_init_acc=1.0, weight_tol=5e-5, whole_lines=whole_lines)
    _stream_parity(6, 4, 12, 12, fw.Optimizer.SGD, n=800, mean_extra=1.0, p_weighted=0.2, ids=500, seed=45, lr=0.05,
                   ffm_lr=0.05, whole_lines=whole_lines)


def test_sequential_stream_config_a_like_scalar_path():
    # examples/ffm: 2 fields, k=10 (scalar path: k % 4 != 0), interaction AB, tiny tables
    _stream_parity(2, 10, 13, 13, fw.Optimizer.AdagradLUT, n=2000, mean_extra=0.0, p_weighted=0.0, ids=300, seed=3,
                   interactions=[(0, 1)], power_t=0.0, ffm_power_t=0.0)


def test_sequential_stream_sgd_and_flex():
    _stream_parity(6, 4, 12, 12, fw.Optimizer.SGD, n=800, mean_extra=1.0, p_weighted=0.2, ids=500, seed=4, lr=0.05,
                   ffm_lr=0.05)
    _stream_parity(6, 8, 12, 12, fw.Optimizer.AdagradFlex, n=800, mean_extra=1.0, p_weighted=0.2, ids=500, seed=5,
                   init_acc=1.0, ffm_init_acc=1.0, weight_tol=5e-5)


@pytest.mark.parametrize("whole_lines", [None, 0, 2])
def test_sequential_stream_k16_two_chunk_rows(whole_lines):
    """R = 30*16 = 480 floats: two 16-byte chunks per lane and row, on the v2 kernel's two-chunk instantiation (static wave ranges,
    duplicate-row chains; whole_lines = 2: 2 KiB windows of whole lines); R = 40*8 = 320: a partly filled second chunk, rows that
    start at all four 32-byte phases of a line; heavy collisions on tiny tables"""
    _stream_parity(30, 16, 16, 18, fw.Optimizer.AdagradLUT, n=120, mean_extra=1.0, p_weighted=0.1, ids=20000, seed=6, whole_lines=whole_lines)
    _stream_parity(40, 8, 14, 14, fw.Optimizer.AdagradLUT, n=150, mean_extra=1.0, p_weighted=0.1, ids=3000, seed=46, whole_lines=whole_lines)
    _stream_parity(30, 16, 13, 13, fw.Optimizer.AdagradFlex, n=100, mean_extra=0.5, p_weighted=0.2, ids=800, seed=47, init_acc=1.0,
                   ffm_init_acc=1.0, weight_tol=5e-5, whole_lines=whole_lines)


def test_concurrent_two_chunk_rows_equal_the_in_order_result_on_disjoint_examples():
    """Concurrent whole-line updates of two-chunk rows (k = 16 x 30 fields) keep no LDS copy of the entries' own slots: the self-pair correction
    (block_ffm.rs:238) reads the slot from the row the update has just re-read -- its copy from before any step of the example, so a duplicate-row chain
    sees the pre-update value like the reference's forward pass (resolve_row_mode: KernelParams::no_selfw).  On examples that share no row and no LR
    entry the order of the examples cannot matter: the concurrent launch (two workgroups per CU, own slots from the re-read rows) must leave the SAME bits
    as the in-order launch (one workgroup, own slots from the gather's LDS copy), duplicates and weighted features included.  Store policy 1: policy 3's
    thinned accumulator stores are unbiased, not bit-equal."""
    F, k, n = 30, 16, 96
    mi, _, _ = make_pair(F, k, 16, 24, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    rng = np.random.default_rng(20241003)
    fbs = []
    for e in range(n):
        feats = []
        j = 0
        for f in range(F):
            for _ in range(int(rng.integers(1, 4))):
                feats.append(((e * 128 + j) * 512, float(rng.choice([1.0, 0.5, 2.0])), f * k))
                j += 1
            if f % 7 == 3:  # the same feature again in the same field: a chained duplicate row
                feats.append((feats[-1][0], 1.0, f * k))
        assert j < 128
        fbs.append(fw.lr_and_ffm_vec([((e * 4 + 1) * 8, 1.0, e % F)], feats, float(e & 1), 1.0 if e % 5 else 0.5))
    out = []
    for mode in (capi.MODE_SEQUENTIAL, capi.MODE_HOGWILD):
        re = fw.Regressor(mi)
        re.set_store_policy(1)
        b = re.batch(fbs)
        re.learn_batch(b, mode, True)
        out.append((b.predictions().copy(), re.table_read(capi.TABLE_FFM_W), re.table_read(capi.TABLE_FFM_ACC), re.table_read(capi.TABLE_LR)))
        b.close()
        re.close()
    for a, c in zip(out[0], out[1]):
        assert np.array_equal(a, c)
    fresh = fw.Regressor(mi)
    assert not np.array_equal(out[0][1], fresh.table_read(capi.TABLE_FFM_W))  # (the batch did step the weights)
    fresh.close()


def test_lr_only_model():
    _stream_parity(8, 0, 14, 14, fw.Optimizer.AdagradLUT, n=1500, mean_extra=1.0, p_weighted=0.2, ids=2000, seed=7,
                   interactions=[(0, 1)])


# ------------------------------------------------------------------ deep head (SURVEY a18, config E shape)
def _nn_stream_parity(n_ns, k, bits, ffm_bits, optimizer, layers, topology, n, seed, mean_extra=1.0, ids=3000,
                      interactions=(), nn_lr=0.02, nn_power_t=0.45, nn_init_acc=0.0, weight_tol=2e-5, setup=None, **kw):
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, optimizer, interactions=interactions, **kw)
    mi.nn_layers = [dict(width=w, activation=a, init=i) for w, a, i in layers]
    mi.nn_topology = topology
    mi.nn_learning_rate, mi.nn_power_t, mi.nn_init_acc_gradient = nn_lr, nn_power_t, nn_init_acc
    nn = fwo.make_nn_config(layers, topology, nn_lr, nn_power_t, nn_init_acc)
    recs, off = fw.synth_records(n_ns, mean_extra, 1.1, ids, 0.1, seed, 0, n)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg, nn=nn)
    L = len(layers)
    w0 = np.concatenate([om.nn_weights(l).copy() for l in range(L + 1)])
    _, p_ref = om.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
    w1 = np.concatenate([om.nn_weights(l) for l in range(L + 1)])
    a1 = np.concatenate([om.nn_acc(l) for l in range(L + 1)])
    assert not np.array_equal(w0, w1)  # the head did learn
    outs = []
    for kind in ("entries", "records"):
        re = fw.Regressor(mi)
        if setup:
            setup(re)
        assert re.table_len(capi.TABLE_NN_W) == w0.size
        re.table_write(capi.TABLE_NN_W, w0)  # Hu / Xavier draws are implementation-defined: load the same ones
        fbt = fw.FeatureBufferTranslator(mi)
        b = re.batch_from_records(fbt, recs, off) if kind == "entries" else re.record_batch(fbt, recs, off)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        p_gpu = b.predictions()
        d_ll = np.abs(logloss(p_gpu, y) - logloss(p_ref, y))
        assert d_ll.max() < LOGLOSS_TOL, f"{kind}: max per-example |d logloss| = {d_ll.max()} at {d_ll.argmax()}"

        def close(a, b_):
            return bool(np.all(np.abs(a - b_) <= weight_tol + 1e-5 * np.abs(b_)))

        assert close(re.table_read(capi.TABLE_NN_W), w1)
        if optimizer != fw.Optimizer.SGD:
            assert close(re.table_read(capi.TABLE_NN_ACC), a1)
        assert close(re.table_read(capi.TABLE_LR), om.lr_table)
        if k:
            assert close(re.table_read(capi.TABLE_FFM_W), om.ffm_weights)
            assert close(re.table_read(capi.TABLE_FFM_ACC), om.ffm_acc)
        outs.append((p_gpu, re.table_checksum(capi.TABLE_NN_W), re.table_checksum(capi.TABLE_FFM_W)))
        b.close()
        re.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]


def test_deep_head_topology_one_relu_layers():
    # config E in small: LR + FFM -> triangle -> [relu, relu] -> final neuron over [h, x]
    _nn_stream_parity(6, 4, 12, 12, fw.Optimizer.AdagradLUT, [(12, "relu", "hu"), (8, "relu", "hu")], "one", n=600,
                      seed=21, interactions=[(0, 1)])


def test_deep_head_k16_scalar_and_topology_two():
    _nn_stream_parity(5, 16, 13, 14, fw.Optimizer.AdagradLUT, [(25, "relu", "xavier")], "one", n=300, seed=22)
    _nn_stream_parity(3, 5, 12, 12, fw.Optimizer.AdagradLUT, [(9, "none", "hu"), (7, "relu", "xavier")], "two", n=400,
                      seed=23)


def test_deep_head_sgd_and_flex():
    _nn_stream_parity(4, 4, 12, 12, fw.Optimizer.SGD, [(10, "relu", "hu")], "one", n=400, seed=24, lr=0.05, ffm_lr=0.05)
    _nn_stream_parity(4, 8, 12, 12, fw.Optimizer.AdagradFlex, [(10, "relu", "hu")], "one", n=400, seed=25,
                      nn_init_acc=1.0, weight_tol=5e-5)


def test_deep_head_blob_round_trip_and_inference():
    mi, _, _ = make_pair(4, 4, 12, 12, fw.Optimizer.AdagradLUT)
    mi.nn_layers = [dict(width=6, activation="relu", init="hu"), dict(width=5, activation="relu", init="xavier")]
    recs, off = fw.synth_records(4, 1.0, 1.1, 3000, 0.1, 26, 0, 200)
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    b = re.record_batch(fbt, recs, off)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    blob = re.write_weights_to_buf()
    X = mi.num_combos + 4 * 5 // 2
    n_nn = (X + 1) * 6 + (6 + 1) * 5 + (5 + X) + 1
    assert re.table_len(capi.TABLE_NN_W) == n_nn
    lr_n, ffm_n = re.table_len(capi.TABLE_LR) // 2, re.table_len(capi.TABLE_FFM_W)  # LR table: {w, acc} pairs
    assert len(blob) == 8 + 8 * (lr_n + ffm_n + n_nn)
    assert int(np.frombuffer(blob[:8], dtype=np.uint64)[0]) == lr_n + ffm_n + n_nn
    # first dense layer: weights then accumulators, right after the FFM arrays
    o = 8 + 8 * (lr_n + ffm_n)
    l0 = (X + 1) * 6
    nn_w = re.table_read(capi.TABLE_NN_W)
    assert np.array_equal(np.frombuffer(blob[o:o + 4 * l0], dtype=np.float32), nn_w[:l0])
    assert np.array_equal(np.frombuffer(blob[o + 4 * l0:o + 8 * l0], dtype=np.float32), re.table_read(capi.TABLE_NN_ACC)[:l0])
    # inference pass == sequential no-update pass; a second regressor loaded from the blob predicts the same
    re.learn_batch(b, capi.MODE_HOGWILD, False)
    p1 = b.predictions().copy()
    re2 = fw.Regressor(mi)
    re2.overwrite_weights_from_buf(blob)
    b2 = re2.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re2.learn_batch(b2, capi.MODE_SEQUENTIAL, False)
    assert np.array_equal(p1, b2.predictions())
    for x in (b, b2, re, re2):
        x.close()


# ------------------------------------------------------------------ BASELINE configs at their real shapes
def test_config_e_real_geometry_sequential_parity():
    """BASELINE configs[4]: F = 30, k = 16 (two-chunk rows, R = 480), 2 x 256 ReLU head, topology one
    (regressor.rs:191-323, block_neural.rs:196-340, block_relu.rs:79-112) -- per-example parity with the oracle in
    the in-order mode, entries and records, NN / FFM / LR tables compared afterwards.  Tables are 20-bit so the
    oracle's copy stays small; the geometry (row length, head widths, ~200 nnz) is the real one."""
    _nn_stream_parity(30, 16, 20, 20, fw.Optimizer.AdagradLUT, [(256, "relu", "hu"), (256, "relu", "hu")], "one", n=96,
                      seed=31, mean_extra=5.67, ids=100000, lr=0.025, ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38,
                      nn_lr=0.025, nn_power_t=0.38, nn_init_acc=1.0)


def test_config_e_real_geometry_sequential_parity_on_the_large_table_kernel():
    """The same stream through the kernel that runs config E's CONCURRENT launches since round 6 -- the head as a phase of the large-table kernel's two-chunk
    instantiation (fw_example_kernel_r<..., NN = true>; fwgpu_debug_set_option 11 = 2 forces in-order launches onto it, option 2 = 3 its update path onto a
    20-bit table): per-example parity with the oracle and the final NN / FFM / LR tables, as above."""
    def setup(re):
        re.set_whole_line_updates(3)
        re.set_head_kernel(2)

    _nn_stream_parity(30, 16, 20, 20, fw.Optimizer.AdagradLUT, [(256, "relu", "hu"), (256, "relu", "hu")], "one", n=96,
                      seed=31, mean_extra=5.67, ids=100000, lr=0.025, ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38,
                      nn_lr=0.025, nn_power_t=0.38, nn_init_acc=1.0, setup=setup)
    _nn_stream_parity(30, 16, 20, 20, fw.Optimizer.SGD, [(256, "relu", "hu"), (256, "relu", "hu")], "one", n=48,
                      seed=33, mean_extra=5.67, ids=100000, lr=0.025, ffm_lr=0.025, nn_lr=0.025, setup=setup)


def test_config_e_real_geometry_hogwild_1024_thread_workgroups():
    """Same geometry in the concurrent mode (the library picks its k = 16 launch shape by itself): finite predictions,
    the loss falls, and a later predict-only pass over the same weights equals the in-order predict-only pass."""
    mi, ocfg, ots = make_pair(30, 16, 20, 20, fw.Optimizer.AdagradLUT, lr=0.025, ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38)
    mi.nn_layers = [dict(width=256, activation="relu", init="hu"), dict(width=256, activation="relu", init="hu")]
    mi.nn_learning_rate, mi.nn_power_t, mi.nn_init_acc_gradient = 0.025, 0.38, 1.0
    recs, off = fw.synth_records(30, 5.67, 1.05, 100000, 0.1, 32, 0, 12000)
    y = record_labels(recs, off)
    re = fw.Regressor(mi)
    b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re.learn_batch(b, capi.MODE_HOGWILD, True)
    p = b.predictions()
    assert np.all(np.isfinite(p))
    ll = logloss(p, y)
    assert ll[-3000:].mean() < ll[:3000].mean() and ll[-3000:].mean() < 0.6931
    re.learn_batch(b, capi.MODE_HOGWILD, False)
    p1 = b.predictions().copy()
    re.learn_batch(b, capi.MODE_SEQUENTIAL, False)
    assert np.abs(b.predictions() - p1).max() < PRED_TOL
    b.close()
    re.close()


def test_config_b_real_size_sequential_parity():
    """BASELINE configs[1]: 10 fields, k = 4, 22-bit FFM and LR tables, micro-batch 4096, seed 20240611.
    One 4096-example micro-batch in the in-order mode against the oracle per example and on the final tables (the hogwild
    hold-out half lives in test_zz_gpu_hogwild_quality.py)."""
    _stream_parity(10, 4, 22, 22, fw.Optimizer.AdagradLUT, n=4096, mean_extra=0.0, p_weighted=0.0, ids=100000,
                   seed=20240611)


def test_sequential_mode_is_deterministic():
    mi, _, _ = make_pair(10, 4, 12, 12, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 1.0, 1.1, 3000, 0.1, 9, 0, 600)
    outs = []
    for _ in range(2):
        re = fw.Regressor(mi)
        b = re.batch_from_records(fw.FeatureBufferTranslator(mi), recs, off)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        outs.append((b.predictions(), re.table_checksum(capi.TABLE_FFM_W), re.table_checksum(capi.TABLE_FFM_ACC),
                     re.table_checksum(capi.TABLE_LR)))
        re.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]


# ------------------------------------------------------------------ inference on identical weights
def test_batch_inference_matches_reference_predict():
    mi, ocfg, ots = make_pair(30, 8, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(30, 5.67, 1.05, 100000, 0.1, 21, 0, 600)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg)
    om.run_stream(ots, recs[: int(off[300])], off[:301], nthreads=1)  # train the oracle on the first half
    re = fw.Regressor(mi)
    re.table_write(capi.TABLE_LR, om.lr_table)                       # identical weights on the device
    re.table_write(capi.TABLE_FFM_W, om.ffm_weights)
    re.table_write(capi.TABLE_FFM_ACC, om.ffm_acc)
    fbt = fw.FeatureBufferTranslator(mi)
    b = re.batch_from_records(fbt, recs, off)
    before = [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    re.learn_batch(b, capi.MODE_HOGWILD, False)  # predict-only: all workgroups, cached loads
    p_gpu = b.predictions()
    after = [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    assert before == after  # inference never writes
    p_ref = np.zeros(600, dtype=np.float32)
    for i in range(600):
        lr, ffm, _, _ = ots.translate(recs[int(off[i]):int(off[i + 1])])
        p_ref[i] = om.predict(lr, ffm)
    assert np.abs(logloss(p_gpu, y) - logloss(p_ref, y)).max() < LOGLOSS_TOL
    assert np.abs(p_gpu - p_ref).max() < PRED_TOL
    # idempotence (the single-workgroup launch uses another workgroup size, hence another summation tree)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, False)
    assert np.abs(b.predictions() - p_gpu).max() < PRED_TOL
    re.learn_batch(b, capi.MODE_HOGWILD, False)
    assert np.array_equal(b.predictions(), p_gpu)
    re.close()


def test_device_translation_of_crafted_records_matches_host_translation():
    import struct

    def f32b(x):
        return struct.unpack("<I", struct.pack("<f", x))[0]

    NS, NOF = 0x80000000, 0x80000000
    ND = fw.NamespaceDescriptor
    mi = fw.ModelInstance(
        learning_rate=0.05, ffm_learning_rate=0.05, bit_precision=12, ffm_bit_precision=12, ffm_k=4,
        optimizer=fw.Optimizer.AdagradLUT, ffm_init_acc_gradient=1.0, add_constant_feature=True,
        feature_combo_descs=[fw.FeatureComboDesc([ND(0)]), fw.FeatureComboDesc([ND(0), ND(1)], 2.0),
                             fw.FeatureComboDesc([ND(2, True)]), fw.FeatureComboDesc([ND(0), ND(1), ND(3)], 0.5)],
        ffm_fields=[[ND(0)], [ND(0), ND(1)], [ND(3)], [ND(2, True)]])
    recs = []
    def rec(label, words):
        r = [3 + 4 + 0, label, f32b(1.0)] + words
        r[0] = len(r)
        return r
    # ns0: two weighted features, ns1: single, ns2 (f32): two, ns3: none
    recs.append(rec(1, [NS | (7 << 16) | 11, 0x123, NS | (11 << 16) | 15, NOF, 0xfea, f32b(2.0), 0xfeb, f32b(3.0), 0x77, f32b(5.0), 0x78, f32b(7.0)]))
    # everything single
    recs.append(rec(0, [0x10, 0x20, 0x30, 0x40]))
    # ns0 empty -> combos with ns0 vanish; ns3 three features
    recs.append(rec(1, [NOF, 0x21, NOF, NS | (7 << 16) | 13, 0x1, f32b(1.0), 0x2, f32b(0.5), 0x3, f32b(0.25)]))
    recs.append(rec(0, [NOF, NOF, NOF, NOF]))
    flat = np.array([w for r in recs for w in r], dtype=np.uint32)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    fbt = fw.FeatureBufferTranslator(mi)
    outs = []
    for kind in ("entries", "records"):
        re = fw.Regressor(mi)
        b = re.batch_from_records(fbt, flat, off) if kind == "entries" else re.record_batch(fbt, flat, off)
        for _ in range(3):
            re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        outs.append((b.predictions(), [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]))
        re.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    # malformed records are rejected on the host before anything reaches the device
    re = fw.Regressor(mi)
    bad = np.array(rec(1, [NS | (7 << 16) | 400, 0x1, 0x2, 0x3]), dtype=np.uint32)
    with pytest.raises(capi.FwgpuError):
        re.record_batch(fbt, bad, np.array([0, len(bad)], dtype=np.uint64))
    re.close()




# ------------------------------------------------------------------ synchronous micro-batches (split pipeline)
def _sync_parity(n_ns, k, bits, ffm_bits, optimizer, n, mb, seed, mean_extra=1.0, ids=3000, p_weighted=0.2, weight_tol=2e-5,
                 interactions=(), nn=None, **kw):
    """fwgpu_learn_batch_sync in the in-order mode == the oracle's micro-batch mode (fw_oracle.h): per micro-batch all
    examples are scored with the weights of the batch start, then the updates are applied in example order."""
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, optimizer, interactions=interactions, **kw)
    onn = None
    if nn:
        layers, topology, nn_lr, nn_pt, nn_acc = nn
        mi.nn_layers = [dict(width=w, activation=a, init=i) for w, a, i in layers]
        mi.nn_topology = topology
        mi.nn_learning_rate, mi.nn_power_t, mi.nn_init_acc_gradient = nn_lr, nn_pt, nn_acc
        onn = fwo.make_nn_config(layers, topology, nn_lr, nn_pt, nn_acc)
    recs, off = fw.synth_records(n_ns, mean_extra, 1.1, ids, p_weighted, seed, 0, n)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg, nn=onn)
    re = fw.Regressor(mi)
    if nn:
        L = len(nn[0])
        w0 = np.concatenate([om.nn_weights(l).copy() for l in range(L + 1)])
        re.table_write(capi.TABLE_NN_W, w0)
    fbt = fw.FeatureBufferTranslator(mi)
    sp = re.split_buffers(mb, 512)
    worst = 0.0
    for s0 in range(0, n, mb):
        e0 = min(n, s0 + mb)
        sub, so = recs[int(off[s0]):int(off[e0])], off[s0:e0 + 1] - off[s0]
        p_ref = om.learn_minibatch(ots, sub, so)
        b = re.record_batch(fbt, sub, so)
        re.learn_batch_sync(b, sp, capi.MODE_SEQUENTIAL)
        p_gpu = b.predictions()
        d = np.abs(logloss(p_gpu, y[s0:e0]) - logloss(p_ref, y[s0:e0])).max()
        worst = max(worst, d)
        assert d < LOGLOSS_TOL, f"micro-batch at {s0}: max per-example |d logloss| = {d}"
        b.close()

    def close(a, b_):
        return bool(np.all(np.abs(a - b_) <= weight_tol + 1e-5 * np.abs(b_)))

    assert close(re.table_read(capi.TABLE_LR), om.lr_table)
    if k:
        assert close(re.table_read(capi.TABLE_FFM_W), om.ffm_weights)
        assert close(re.table_read(capi.TABLE_FFM_ACC), om.ffm_acc)
    if nn:
        w1 = np.concatenate([om.nn_weights(l) for l in range(L + 1)])
        a1 = np.concatenate([om.nn_acc(l) for l in range(L + 1)])
        assert close(re.table_read(capi.TABLE_NN_W), w1), np.abs(re.table_read(capi.TABLE_NN_W) - w1).max()
        if optimizer != fw.Optimizer.SGD:
            assert close(re.table_read(capi.TABLE_NN_ACC), a1)
    sp.close()
    re.close()
    return worst


def test_sync_micro_batch_matches_oracle_micro_batch_mode():
    _sync_parity(10, 4, 14, 14, fw.Optimizer.AdagradLUT, n=1024, mb=128, seed=51)
    _sync_parity(30, 8, 18, 18, fw.Optimizer.AdagradLUT, n=192, mb=64, seed=52, mean_extra=5.67, ids=100000, p_weighted=0.1,
                 interactions=[(0, 1)])
    _sync_parity(2, 10, 13, 13, fw.Optimizer.AdagradLUT, n=600, mb=100, seed=53, mean_extra=0.0, ids=300, p_weighted=0.0,
                 interactions=[(0, 1)], power_t=0.0, ffm_power_t=0.0)
    _sync_parity(6, 8, 12, 12, fw.Optimizer.AdagradFlex, n=512, mb=64, seed=54, init_acc=1.0, ffm_init_acc=1.0, weight_tol=5e-5)
    # 30 fields x k = 16: the phase kernels keep T in the split record instead of LDS (launch_example_phase, KernelParams::t_global)
    _sync_parity(30, 16, 20, 20, fw.Optimizer.AdagradLUT, n=128, mb=64, seed=55, mean_extra=5.67, ids=100000, p_weighted=0.1, lr=0.025, ffm_lr=0.025,
                 power_t=0.38, ffm_power_t=0.38)
    _sync_parity(6, 4, 12, 12, fw.Optimizer.SGD, n=512, mb=64, seed=55, lr=0.05, ffm_lr=0.05)
    _sync_parity(8, 0, 14, 14, fw.Optimizer.AdagradLUT, n=600, mb=50, seed=56, interactions=[(0, 1)])
    _sync_parity(30, 16, 16, 18, fw.Optimizer.AdagradLUT, n=96, mb=32, seed=57, ids=20000, p_weighted=0.1)


def test_mini_batched_deep_head_matches_oracle_micro_batch_mode():
    """The MFMA head (head.hip): a deep head inside synchronous micro-batches = frozen dense weights per batch, gradients summed
    over the batch, ONE optimizer step per dense weight -- against the oracle's statement of that mode (fw_oracle.h), in the
    in-order mode.  Small shapes with ragged tiles, both topologies, identity and ReLU layers, three optimizers; then config E's
    real geometry (F = 30, k = 16, 2 x 256 ReLU, topology one)."""
    _sync_parity(6, 4, 12, 12, fw.Optimizer.AdagradLUT, n=384, mb=64, seed=61, interactions=[(0, 1)],
                 nn=([(12, "relu", "hu"), (8, "relu", "hu")], "one", 0.02, 0.45, 1.0))
    _sync_parity(5, 16, 13, 14, fw.Optimizer.AdagradLUT, n=200, mb=50, seed=62, nn=([(25, "relu", "xavier")], "one", 0.02, 0.45, 1.0))
    _sync_parity(3, 5, 12, 12, fw.Optimizer.AdagradLUT, n=300, mb=75, seed=63,
                 nn=([(9, "none", "hu"), (7, "relu", "xavier")], "two", 0.02, 0.45, 1.0))
    _sync_parity(4, 4, 12, 12, fw.Optimizer.SGD, n=256, mb=64, seed=64, lr=0.05, ffm_lr=0.05, nn=([(10, "relu", "hu")], "one", 0.01, 0.45, 0.0))
    _sync_parity(4, 8, 12, 12, fw.Optimizer.AdagradFlex, n=256, mb=64, seed=65, init_acc=1.0, ffm_init_acc=1.0, weight_tol=5e-5,
                 nn=([(10, "relu", "hu")], "one", 0.02, 0.45, 1.0))
    _sync_parity(30, 16, 20, 20, fw.Optimizer.AdagradLUT, n=128, mb=64, seed=66, mean_extra=5.67, ids=100000, p_weighted=0.1, lr=0.025,
                 ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38, nn=([(256, "relu", "hu"), (256, "relu", "hu")], "one", 0.025, 0.38, 1.0))


# ------------------------------------------------------------------ serving context cache
CACHE_TOL = 5e-6  # assert_epsilon! of the reference's *_with_cache tests (block_helpers.rs:30-40)


@pytest.mark.parametrize("sc", [s for s in KATS["scenarios"] if any(st["ffm"] for st in s["steps"])],
                         ids=[s["name"] for s in KATS["scenarios"] if any(st["ffm"] for st in s["steps"])])
def test_reference_with_cache_twins(sc):
    """Every reference FFM test has a *_with_cache twin (block_ffm.rs:1325-2037) asserting that setup_cache on PART of the
    example followed by predict_with_cache on the whole example gives the SAME numbers as the plain prediction.  Here: before
    every step of every FFM KAT scenario, a cache is set up from each single feature and from the first half of the features,
    and predict_with_cache must reproduce the step's expected prediction."""
    mi = mi_from_cfg(sc["config"], sc["wiring"])
    re = fw.Regressor(mi)
    if "ffm_fill" in sc:
        re.ffm_fill(sc["ffm_fill"])
    cache = None
    for i, st in enumerate(sc["steps"]):
        fb = fw.lr_and_ffm_vec(st["lr"], st["ffm"], st["label"], st["importance"])
        want = st.get("current_code", st["expect"]) if st.get("stale") else st["expect"]
        ffm = list(st["ffm"])
        subsets = [[f] for f in ffm] + ([ffm[: len(ffm) // 2]] if len(ffm) > 1 else []) + [[]]
        for sub in subsets:
            cache = re.setup_cache(fw.lr_and_ffm_vec(st["lr"], sub, st["label"], st["importance"]), cache)
            got = re.predict_with_cache(fb, None, cache)
            assert abs(got - want) < CACHE_TOL, f"{sc['name']} step {i} cache {sub}: got {got!r} want {want!r}"
        if st["op"] == "predict" or not st["update"]:
            re.predict(fb)
        else:
            re.learn(fb, None, True)
    if cache is not None:
        cache.close()
    re.close()


@pytest.mark.parametrize("k,n_ns", [(8, 30), (4, 10), (10, 3), (16, 30)])
def test_context_cache_random_contexts(k, n_ns):
    """predict_with_cache(context + candidate) == predict(context + candidate) on a trained model, for contexts made of
    whole namespaces, of single features, with duplicates, with fields that only the context / only the candidate fills;
    the cached route through an entry batch (all candidates of a request in one launch) gives the single calls' numbers."""
    mi, ocfg, ots = make_pair(n_ns, k, 16, 16, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    recs, off = fw.synth_records(n_ns, 2.0, 1.1, 3000, 0.3, 71, 0, 400)
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    b = re.record_batch(fbt, recs[: int(off[300])], off[:301])
    re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    b.close()
    rng = np.random.default_rng(k)
    worst = 0.0
    for i in range(300, 360):
        fb = fbt.translate(recs[int(off[i]):int(off[i + 1])])
        ffm = np.asarray(fb.ffm_buffer)
        plain = re.predict(fb)
        fields = np.unique(ffm["contra_field_index"])
        ctx_fields = rng.choice(fields, size=max(1, len(fields) // 2), replace=False) if len(fields) else fields
        # (a context holds ALL occurrences of its features: features_present is keyed by hash + field, regressor.rs:25-38)
        third = set(zip(ffm["hash"][::3].tolist(), ffm["contra_field_index"][::3].tolist()))
        in_third = np.array([(h, f) in third for h, f in zip(ffm["hash"].tolist(), ffm["contra_field_index"].tolist())], dtype=bool)
        for ctx in (ffm[np.isin(ffm["contra_field_index"], ctx_fields)], ffm[in_third], ffm[:0], ffm):
            cfb = fw.FeatureBuffer(label=0.0, example_importance=1.0, example_number=0, lr_buffer=fb.lr_buffer, ffm_buffer=ctx)
            cache = re.setup_cache(cfb)
            got = re.predict_with_cache(fb, None, cache)
            worst = max(worst, abs(got - plain))
            # the entries the cache does not cover: nothing of the context, everything else, order kept
            rest = cache.filter(ffm)
            keys = set(zip(ctx["hash"].tolist(), ctx["contra_field_index"].tolist()))
            keep = np.array([(h, f) not in keys for h, f in zip(ffm["hash"].tolist(), ffm["contra_field_index"].tolist())], dtype=bool)
            assert np.array_equal(rest, ffm[keep])
            cache.close()
    assert worst < CACHE_TOL, worst
    # one request: a context + 50 candidates through ONE launch of an entry batch that holds only the uncovered entries
    fb0 = fbt.translate(recs[int(off[360]):int(off[361])])
    ctx = np.asarray(fb0.ffm_buffer)
    ctx = ctx[ctx["contra_field_index"] < (n_ns // 2) * k]
    cache = re.setup_cache(fw.FeatureBuffer(label=0.0, example_importance=1.0, example_number=0, lr_buffer=fb0.lr_buffer, ffm_buffer=ctx))
    fulls, cut = [], []
    for i in range(361, 400):
        fbi = fbt.translate(recs[int(off[i]):int(off[i + 1])])
        cand = np.asarray(fbi.ffm_buffer)
        cand = cand[cand["contra_field_index"] >= (n_ns // 2) * k]
        whole = np.concatenate([ctx, cand])
        fulls.append(fw.FeatureBuffer(label=0.0, example_importance=1.0, example_number=0, lr_buffer=fbi.lr_buffer, ffm_buffer=whole))
        cut.append(fw.FeatureBuffer(label=0.0, example_importance=1.0, example_number=0, lr_buffer=fbi.lr_buffer, ffm_buffer=cache.filter(whole)))
    want = np.array([re.predict(f) for f in fulls], dtype=np.float32)
    single = np.array([re.predict_with_cache(f, None, cache) for f in fulls], dtype=np.float32)
    eb = re.batch(cut)
    eb.set_cache(cache)
    re.learn_batch(eb, capi.MODE_HOGWILD, False)
    batched = eb.predictions()
    assert np.abs(single - want).max() < CACHE_TOL and np.abs(batched - want).max() < CACHE_TOL
    eb.set_cache(None)
    eb.close()
    cache.close()
    re.close()

# ------------------------------------------------------------------ hogwild mode and the record-stream trainer
@pytest.mark.parametrize("shape", ["ffm_k4", "lr_only", "ffm_k8_generic_kernel"])
def test_hogwild_steps_on_the_constant_features_entry_are_all_applied(shape):
    """The hot LR entry (kernels.hip hot_lr_flush; steps on the GLOBAL accumulator by a returning atomic, weight deltas by
    fire-and-forget atomics): whatever the interleaving, the entry's accumulator after a hogwild launch is
    acc0 + the sum over ALL examples of g^2, g = -(label - prediction) * importance being what each example's sigmoid produced
    (block_loss_functions.rs:141, block_lr.rs:135-150 with value 1.0) -- computable from the launch's own predictions.  With
    plain per-example read-modify-writes (option 0) concurrent steps overwrite each other and most of the sum is missing."""
    n = 30000
    k = {"ffm_k4": 4, "lr_only": 0, "ffm_k8_generic_kernel": 8}[shape]
    mi, ocfg, ots = make_pair(10, k, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 7, 0, n)
    y = record_labels(recs, off)
    label = (y == 1).astype(np.float64)
    h = 11650396 & ((1 << 18) - 1)  # feature_buffer.rs:8, 270-276
    got = {}
    for every in (1, 5, 0):  # 1: every step sent at once (the default); 5: weight deltas pending 5 examples; 0: plain route
        re = fw.Regressor(mi)
        if shape == "ffm_k8_generic_kernel":
            capi.check(re.L.fwgpu_debug_set_kernel_version(re.h, 1))
        re.set_hot_lr_entry(every)
        b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
        re.learn_batch(b, capi.MODE_HOGWILD, True)
        p = b.predictions().astype(np.float64)
        w, acc = re.table_read(capi.TABLE_LR, 2 * h, 2)
        got[every] = (float(acc) - 1.0) / float(((label - p) ** 2).sum())
        assert np.isfinite(w) and w != 0.0
        b.close()
        re.close()
    assert abs(got[1] - 1.0) < 2e-3 and abs(got[5] - 1.0) < 2e-3, got  # (f32 sums in another order)
    assert got[0] < 0.9, got  # the plain route loses concurrent steps (measured: 0.1-0.5 of the sum survives)


def test_cross_xcd_visibility_of_sc1_accesses():
    L = capi.lib()
    stale, tmo = C.c_uint32(0), C.c_uint32(0)
    capi.check(L.fwgpu_debug_coherence_probe(0, 1, 2000, C.byref(stale), C.byref(tmo)))
    assert tmo.value == 0
    assert stale.value == 0, f"sc1 store -> sc1 load saw {stale.value} stale words"
    stale2, tmo2 = C.c_uint32(0), C.c_uint32(0)
    capi.check(L.fwgpu_debug_coherence_probe(0, 0, 2000, C.byref(stale2), C.byref(tmo2)))
    print(f"plain store/load variant: {stale2.value} stale words (informational), timeouts {tmo2.value}")


# ------------------------------------------------------------------ weight init, tables, serialisation
def _np_checksum(a):
    i = np.arange(a.size, dtype=np.uint64)
    x = (i << np.uint64(32)) ^ a.view(np.uint32).astype(np.uint64) ^ (i >> np.uint64(32))
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        return int(x.sum(dtype=np.uint64))


@pytest.mark.parametrize("k,bits,opt,width", [(8, 16, fw.Optimizer.AdagradLUT, 0.0), (4, 14, fw.Optimizer.AdagradFlex, 0.0),
                                              (10, 15, fw.Optimizer.SGD, 0.0), (8, 14, fw.Optimizer.AdagradLUT, 0.2)])
def test_weight_init_is_bit_exact_vs_oracle(k, bits, opt, width):
    mi, ocfg, _ = make_pair(5, k, 14, bits, opt, ffm_init_acc=0.5)
    mi.ffm_init_width, mi.ffm_init_zero_band, mi.ffm_init_center = width, 0.25, 0.01
    ocfg.ffm_init_width, ocfg.ffm_init_zero_band, ocfg.ffm_init_center = width, 0.25, 0.01
    om = fwo.Model(ocfg)
    re = fw.Regressor(mi)
    assert re.table_len(capi.TABLE_FFM_W) == (1 << bits) + 5 * k  # block_ffm.rs:92-94
    assert np.array_equal(re.table_read(capi.TABLE_FFM_W).view(np.uint32), om.ffm_weights.view(np.uint32))
    assert np.array_equal(re.table_read(capi.TABLE_FFM_ACC), om.ffm_acc)
    assert np.array_equal(re.table_read(capi.TABLE_LR), om.lr_table)
    assert re.table_checksum(capi.TABLE_FFM_W) == _np_checksum(om.ffm_weights)
    re.close()


def test_serialised_weights_follow_the_reference_layout():
    for opt in (fw.Optimizer.AdagradLUT, fw.Optimizer.SGD):
        mi, ocfg, ots = make_pair(4, 4, 10, 10, opt)
        recs, off = fw.synth_records(4, 1.0, 1.1, 200, 0.1, 5, 0, 200)
        re = fw.Regressor(mi)
        b = re.batch_from_records(fw.FeatureBufferTranslator(mi), recs, off)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        blob = re.write_weights_to_buf()
        lr, w, acc = re.table_read(capi.TABLE_LR), re.table_read(capi.TABLE_FFM_W), re.table_read(capi.TABLE_FFM_ACC)
        n_lr, n_ffm = 1 << 10, (1 << 10) + 16
        # regressor.rs:426-442: u64 total = sum of get_serialized_len, then LR block, FFM weights, FFM optimizer
        assert int(np.frombuffer(blob[:8], dtype="<u8")[0]) == n_lr + n_ffm
        body = np.frombuffer(blob[8:], dtype="<f4")
        if opt == fw.Optimizer.SGD:  # 4-byte LR entries, no optimizer part (optimizer.rs:20)
            assert len(body) == n_lr + n_ffm
            assert np.array_equal(body[:n_lr], lr[0::2]) and np.array_equal(body[n_lr:], w)
        else:
            assert len(body) == 2 * n_lr + 2 * n_ffm
            assert np.array_equal(body[: 2 * n_lr], lr)
            assert np.array_equal(body[2 * n_lr: 2 * n_lr + n_ffm], w) and np.array_equal(body[2 * n_lr + n_ffm:], acc)
        re2 = fw.Regressor(mi)
        re2.overwrite_weights_from_buf(blob)
        for t in (capi.TABLE_LR, capi.TABLE_FFM_W) + ((capi.TABLE_FFM_ACC,) if opt != fw.Optimizer.SGD else ()):
            assert re2.table_checksum(t) == re.table_checksum(t)
        with pytest.raises(capi.FwgpuError):
            re2.overwrite_weights_from_buf(blob[:-4])
        re.close()
        re2.close()


# ------------------------------------------------------------------ edge cases the reference tests
def test_edge_cases_against_oracle():
    k, F = 4, 5
    mi, ocfg, _ = make_pair(F, k, 10, 10, fw.Optimizer.AdagradLUT)
    om = fwo.Model(ocfg)
    re = fw.Regressor(mi)
    R = F * k
    cases = [
        ([], []),                                                   # no features at all
        ([(3, 1.0, 0)], []),                                        # LR only
        ([], [(8, 1.0, 2 * k)]),                                    # a single FFM feature: only self-pairs -> 0
        ([(7, 1.0, 0), (7, 2.0, 0), (7, -1.5, 1)], []),             # duplicate LR hash (regressor.rs:629-655)
        ([], [(16, 1.0, 0), (16, 2.0, 1 * k), (200, 1.0, 4 * k)]),  # same row in two fields
        ([], [(16, 1.0, 0), (16 + 4, 2.0, 1 * k), (16 + 8, 0.5, 3 * k)]),  # overlapping rows (stride 4 < R)
        ([(5, 1.0, 0)], [(40, 1.0, 1 * k), (44, 1.0, 1 * k), (40, 3.0, 1 * k), (400, 1.0, 3 * k)]),  # dup inside a field
        ([(9, 1.0, 5)], [(1020, 1.0, 0), (1020, 1.0, 4 * k)]),      # row reaching into the spill-over tail
    ]
    for rep in range(3):
        for ci, (lr, ffm) in enumerate(cases):
            for label, imp in ((1.0, 1.0), (0.0, 0.5), (1.0, 0.0)):
                fb = fw.lr_and_ffm_vec(lr, ffm, label, imp)
                pg = re.learn(fb, None, True)
                po = om.learn(fwo.lr_entries(lr), fwo.ffm_entries(ffm), label, imp, True)
                assert abs(pg - po) < PRED_TOL, (rep, ci, label, imp, pg, po)
    assert np.abs(re.table_read(capi.TABLE_FFM_W) - om.ffm_weights).max() < 1e-5
    assert np.abs(re.table_read(capi.TABLE_FFM_ACC) - om.ffm_acc).max() < 1e-5
    assert np.abs(re.table_read(capi.TABLE_LR) - om.lr_table).max() < 1e-5
    # unordered FFM entries / out-of-range hashes are rejected, not mis-trained
    with pytest.raises(capi.FwgpuError):
        re.learn(fw.ffm_vec([(8, 1.0, 2 * k), (8, 1.0, 0)]), None, True)
    with pytest.raises(capi.FwgpuError):
        re.learn(fw.ffm_vec([(1 << 10 + 1, 1.0, 0)]), None, True)
    with pytest.raises(capi.FwgpuError):
        re.learn(fw.lr_vec([(1 << 10, 1.0, 0)]), None, True)
    # empty batch
    b = re.batch([])
    re.learn_batch(b, capi.MODE_HOGWILD, True)
    assert len(b.predictions()) == 0
    re.close()


def test_saturated_and_nan_logits_follow_the_sigmoid_rules():
    # block_loss_functions.rs:125-141: |wsum| > 50 -> p = sigma(+-50), no update; NaN -> 0.5, no update
    mi = fw.ModelInstance(optimizer=fw.Optimizer.SGD, learning_rate=0.1, bit_precision=10)
    re = fw.Regressor(mi)
    om = fwo.Model(fwo.make_config(optimizer=fwo.OPT_SGD, learning_rate=0.1, bit_precision=10))
    for wv, want in ((100.0, 1.0), (-100.0, 1.9287e-22), (float("nan"), 0.5)):
        t = np.zeros(2 << 10, dtype=np.float32)
        t[2 * 5] = wv
        re.table_write(capi.TABLE_LR, t)
        om.lr_table[:] = t
        fb = fw.lr_vec([(5, 1.0, 0)], label=0.0)
        pg, po = re.learn(fb, None, True), om.learn(fwo.lr_entries([(5, 1.0, 0)]), None, 0.0, 1.0, True)
        assert (np.isnan(pg) and np.isnan(po)) or abs(pg - po) < 1e-7
        assert abs(pg - want) < 1e-6
        got = re.table_read(capi.TABLE_LR, 10, 1)[0]
        assert (np.isnan(got) and np.isnan(wv)) or got == np.float32(wv)  # general gradient 0: weight untouched
    re.close()


# ------------------------------------------------------------------ BASELINE.json full sizes: size-independent properties
def test_full_size_config_c_properties():
    # 30 fields, k=8, 28-bit hash: 2 x 1.07 GB FFM tables + 2.1 GB LR table.  The oracle cannot follow at this
    # size in seconds, so check properties: init checksum == closed form on a slice, inference is read-only and
    # idempotent, a training pass changes exactly the touched rows, sequential == oracle on the touched rows.
    mi, ocfg, ots = make_pair(30, 8, 28, 28, fw.Optimizer.AdagradLUT)
    re = fw.Regressor(mi)
    L = fwo.lib()
    n = (1 << 28) + 240
    # spot-check the init against the published LCG on slices across the table, including the tail
    for start in (0, 123456789, n - 4096):
        got = re.table_read(capi.TABLE_FFM_W, start, 4096)
        want = np.array([(1.0 * L.fwo_merand48(n + start + i) - 0.5) for i in range(4096)], dtype=np.float32)
        want = (want * np.float32(1.0 / np.sqrt(np.float32(8.0)) / np.float32(50.0))).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    recs, off = fw.synth_records(30, 5.67, 1.05, 10_000_000, 0.1, 20240612, 0, 2048)
    fbt = fw.FeatureBufferTranslator(mi)
    b = re.batch_from_records(fbt, recs, off)
    c0 = [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    re.learn_batch(b, capi.MODE_HOGWILD, False)
    p1 = b.predictions()
    re.learn_batch(b, capi.MODE_HOGWILD, False)
    p2 = b.predictions()
    assert np.array_equal(p1, p2) and np.all((p1 > 0) & (p1 < 1))
    assert c0 == [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    # sequential training of the first 64 examples vs the oracle restricted to the touched rows
    nseq = 64
    bs = re.batch_from_records(fbt, recs[: int(off[nseq])], off[: nseq + 1])
    re.learn_batch(bs, capi.MODE_SEQUENTIAL, True)
    p_gpu = bs.predictions()
    c1 = [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    assert all(a != b_ for a, b_ in zip(c0, c1))
    om = fwo.Model(ocfg)  # 4.3 GB on the host; init takes a few seconds
    _, p_ref = om.run_stream(ots, recs[: int(off[nseq])], off[: nseq + 1], nthreads=1)
    y = record_labels(recs, off)[:nseq]
    assert np.abs(logloss(p_gpu, y) - logloss(p_ref, y)).max() < LOGLOSS_TOL
    for i in range(0, nseq, 7):
        lr, ffm, _, _ = ots.translate(recs[int(off[i]):int(off[i + 1])])
        for e in ffm[::17]:
            h = int(e["hash"])
            assert np.abs(re.table_read(capi.TABLE_FFM_W, h, 240) - om.ffm_weights[h:h + 240]).max() < 2e-5
            assert np.abs(re.table_read(capi.TABLE_FFM_ACC, h, 240) - om.ffm_acc[h:h + 240]).max() < 1e-4
    om.close()
    re.close()


# ------------------------------------------------------------------ multi-GPU bookkeeping kernels
@pytest.mark.parametrize("tiled", [0, 1])
def test_head_products_match_a_torch_f32_reference(tiled):
    """The mini-batched head's products (head.hip) against plain torch f32 matmuls: the split-K MFMA kernel (one 32 x 32 tile per workgroup, eight waves
    over K) and the LDS-tiled one, for the three operand layouts the head uses, with ragged M / N, every epilogue, and shapes the split-K kernel must
    refuse (K % 8 != 0, odd leading dimensions: the tiled kernel takes them).  f32 products summed in another order: tolerance 2e-5 relative to |A||B| row sums."""
    import torch

    L = capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(7)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).cuda()

    def check(got, want, A2, B2):
        bound = 2e-5 * (A2.abs() @ B2.abs()) + 1e-6
        assert bool(((got - want).abs() <= bound).all()), float((got - want).abs().max())

    for M, N, K in ((1024, 256, 496), (100, 70, 64), (33, 31, 72), (64, 496, 256), (50, 40, 20), (37, 29, 13)):
        # forward: h = relu(x . W^T + b), mask        A[M, K] row-major, B = W[N, K] row-major (tb), epilogue 1
        x, W, b = rnd(M, K), rnd(N, K), rnd(N)
        h, mask = torch.full((M, N), 7.0, device="cuda"), torch.full((M, N), 7.0, device="cuda")
        capi.check(L.fwgpu_debug_head_gemm(x.data_ptr(), W.data_ptr(), h.data_ptr(), M, N, K, K, K, N, 0, 1, 1, b.data_ptr(), mask.data_ptr(), 1, tiled, st))
        z = x @ W.t() + b
        check(h, torch.relu(z), x, W.t())
        sure = z.abs() > 1e-3  # (a pre-activation within rounding of 0 may fall on either side)
        assert bool((mask[sure] == (z[sure] >= 0).float()).all())
        # weight gradients: dW[N2, K2] = dz^T . in      A = dz[Kb, M2] (ta: the batch is the slow dimension of both operands), epilogue 0
        Kb, M2, N2 = M, N, K
        dz, inp = rnd(Kb, M2), rnd(Kb, N2)
        dW = torch.full((M2, N2), 7.0, device="cuda")
        # (the split-K kernel needs Kb % 8 == 0; other batch sizes run on the tiled kernel: same call)
        capi.check(L.fwgpu_debug_head_gemm(dz.data_ptr(), inp.data_ptr(), dW.data_ptr(), M2, N2, Kb, M2, N2, N2, 1, 0, 0, None, None, 0, tiled, st))
        check(dW, dz.t() @ inp, dz.t(), inp)
        # input gradients: d_in = (dz . W) * mask (epilogue 2), and accumulated into an existing buffer (epilogue 3)
        dzz, W2, m2 = rnd(M, N), rnd(N, K), (rnd(M, K) > 0).float()
        din = torch.full((M, K), 7.0, device="cuda")
        capi.check(L.fwgpu_debug_head_gemm(dzz.data_ptr(), W2.data_ptr(), din.data_ptr(), M, K, N, N, K, K, 0, 0, 2, None, m2.data_ptr(), 0, tiled, st))
        check(din, (dzz @ W2) * m2, dzz, W2)
        base = rnd(M, K)
        acc = base.clone()
        capi.check(L.fwgpu_debug_head_gemm(dzz.data_ptr(), W2.data_ptr(), acc.data_ptr(), M, K, N, N, K, K, 0, 0, 3, None, None, 0, tiled, st))
        check(acc - base, dzz @ W2, dzz, W2)
    torch.cuda.synchronize()


def test_delta_kernels_match_torch_reference():
    import torch

    L = capi.lib()
    for n, scale in ((1, 1.0), (3, 0.5), (4, 1.0), (1027, 0.125), (1 << 20, 0.25)):
        g = torch.Generator(device="cpu").manual_seed(n)
        t = torch.randn(n, generator=g).cuda()
        s0 = torch.randn(n, generator=g).cuda()
        d = torch.empty(n, device="cuda")
        D = torch.empty(n, device="cuda")
        t0, s00 = t.clone(), s0.clone()
        st = torch.cuda.current_stream().cuda_stream
        capi.check(L.fwgpu_delta_start(t.data_ptr(), s0.data_ptr(), d.data_ptr(), D.data_ptr(), n, scale, st))
        assert torch.equal(d, t0 - s00) and torch.equal(D, d * scale)
        D += 0.5  # "others" contributed 0.5 everywhere
        t += 0.25  # local updates while the exchange was in flight
        capi.check(L.fwgpu_delta_finish(t.data_ptr(), s0.data_ptr(), d.data_ptr(), D.data_ptr(), n, st))
        torch.cuda.synchronize()
        assert torch.allclose(s0, s00 + scale * (t0 - s00) + 0.5, atol=1e-6)
        # the local delta is replaced by the agreed one; updates made meanwhile stay
        assert torch.allclose(t, s00 + scale * (t0 - s00) + 0.5 + 0.25, atol=1e-6)


def test_table_torch_view_is_zero_copy():
    import torch

    mi, _, _ = make_pair(4, 4, 10, 10, fw.Optimizer.AdagradLUT)
    re = fw.Regressor(mi)
    v = re.table_as_torch(capi.TABLE_FFM_W)
    assert v.is_cuda and v.dtype == torch.float32 and v.numel() == re.table_len(capi.TABLE_FFM_W)
    assert np.array_equal(v.cpu().numpy(), re.table_read(capi.TABLE_FFM_W))
    v[5] = 42.0
    torch.cuda.synchronize()
    assert re.table_read(capi.TABLE_FFM_W, 5, 1)[0] == np.float32(42.0)
    re.close()


# ------------------------------------------------------------------ the C++ host mirror, reference-style tests
def test_cpp_host_mirror_reference_tests():
    import subprocess

    exe = os.path.join(ROOT, "host", "test_reference_kats")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all 16 tests passed" in r.stdout


def test_very_large_examples():
    """An example with thousands of features (1024-thread workgroups, the generic kernel) next to ordinary ones: same results
    as the oracle in the in-order mode, whatever their size."""
    k, F = 4, 5
    mi, ocfg, _ = make_pair(F, k, 16, 16, fw.Optimizer.AdagradLUT, lr=0.02, ffm_lr=0.02)
    rng = np.random.default_rng(31)

    def example(n_ffm, n_lr):
        fields = np.sort(rng.integers(0, F, n_ffm))
        ffm = [(int(rng.integers(0, 1 << 14)) * 4, float(rng.choice([1.0, 0.5, 2.0])), int(f) * k) for f in fields]
        lr = [(int(rng.integers(0, 1 << 16)), float(rng.choice([1.0, 0.25])), int(rng.integers(0, F))) for _ in range(n_lr)]
        return lr, ffm

    sizes = [(5, 6), (1200, 1500), (8, 3), (2000, 10), (0, 1500), (12, 7)]
    exs = [example(a, b) for a, b in sizes]
    labels = [1.0, 0.0, 1.0, 1.0, 0.0, 1.0]
    om = fwo.Model(ocfg)
    p_ref = [om.learn(fwo.lr_entries(lr), fwo.ffm_entries(ffm), y, 1.0, True) for (lr, ffm), y in zip(exs, labels)]
    re = fw.Regressor(mi)
    b = re.batch([fw.lr_and_ffm_vec(lr, ffm, y, 1.0) for (lr, ffm), y in zip(exs, labels)])
    re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    p = b.predictions()
    assert np.abs(p - np.array(p_ref, dtype=np.float32)).max() < 2e-5, (p, p_ref)
    assert np.abs(re.table_read(capi.TABLE_FFM_W) - om.ffm_weights).max() < 2e-5
    assert np.abs(re.table_read(capi.TABLE_LR) - om.lr_table).max() < 1e-4
    # the concurrent mode copes with the same batch (results are order dependent: only sanity is checked)
    re.learn_batch(b, capi.MODE_HOGWILD, True)
    assert np.all(np.isfinite(b.predictions()))
    b.close()
    re.close()
    # more than a workgroup stages (4096 entries), or more than the LDS holds (3900 + 3900): the chunked path (regressor.cpp learn_one_chunked), same
    # results as the oracle -- round 3 refused these with FWGPU_ERR_RANGE
    om2, re2 = fwo.Model(ocfg), fw.Regressor(mi)
    for n_ffm, n_lr in ((4100, 1), (3900, 3900), (30, 5000)):
        lr, ffm = example(n_ffm, n_lr)
        want = om2.learn(fwo.lr_entries(lr), fwo.ffm_entries(ffm), 1.0, 1.0, True)
        got = re2.learn(fw.lr_and_ffm_vec(lr, ffm, 1.0, 1.0), None, True)
        assert abs(got - want) < 2e-5, (n_ffm, n_lr, got, want)
    assert np.abs(re2.table_read(capi.TABLE_FFM_W) - om2.ffm_weights).max() < 2e-5
    re2.close()


def test_trainer_holdout_and_testonly_protocol():
    """main.rs:238-241: `--holdout_after N` examples numbered >= N are predicted with the model as it is, never learned;
    `-t` learns nothing.  The kept predictions equal a predict-only pass over the same examples afterwards."""
    mi, _, _ = make_pair(8, 4, 16, 16, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    recs, off = fw.synth_records(8, 1.0, 1.1, 5000, 0.2, 91, 0, 12000)
    re = fw.Regressor(mi)
    tr = fw.HogwildTrainer(re, mi, micro_batch=1024)
    tr.set_holdout(holdout_after=10001)
    # fed in uneven slices, one of them straddling the boundary
    for a, b_ in ((0, 3000), (3000, 9990), (9990, 10030), (10030, 12000)):
        tr.digest_records(recs[int(off[a]):int(off[b_])], off[a:b_ + 1] - off[a])
    tr.block_until_workers_finished()
    assert tr.examples_seen() == 12000
    p_hold = tr.predictions()
    assert len(p_hold) == 2000
    sums = [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    hb = re.record_batch(fw.FeatureBufferTranslator(mi), recs[int(off[10000]):], off[10000:] - off[10000])
    re.learn_batch(hb, capi.MODE_HOGWILD, False)
    assert np.array_equal(hb.predictions(), p_hold)  # same model, same examples, same order
    # exactly the first 10 000 were learned: a model trained on those alone (in-order mode for determinism is not
    # needed: only WHICH rows were touched is compared) has touched the same accumulators
    re2 = fw.Regressor(mi)
    b2 = re2.record_batch(fw.FeatureBufferTranslator(mi), recs[: int(off[10000])], off[:10001])
    re2.learn_batch(b2, capi.MODE_HOGWILD, True)
    b2.predictions()
    assert np.array_equal(re.table_read(capi.TABLE_FFM_ACC) != 0, re2.table_read(capi.TABLE_FFM_ACC) != 0)
    # -t: nothing moves, every example gets a prediction
    tr2 = fw.HogwildTrainer(re, mi, micro_batch=700)
    tr2.set_holdout(testonly=True)
    tr2.digest_records(recs[: int(off[3000])], off[:3001])
    tr2.block_until_workers_finished()
    assert len(tr2.predictions()) == 3000
    assert [re.table_checksum(t) for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)] == sums
    for x in (hb, b2, tr, tr2, re, re2):
        x.close()


def test_one_example_in_flight_is_the_sequential_mode():
    """fwgpu_set_max_in_flight(1): the trainer (a HOGWILD launcher) then IS the reference's single-thread loop"""
    mi, _, _ = make_pair(6, 4, 12, 12, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(6, 1.0, 1.1, 3000, 0.2, 93, 0, 2500)
    re_a = fw.Regressor(mi)
    b = re_a.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re_a.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    b.predictions()
    re_b = fw.Regressor(mi)
    re_b.set_max_in_flight(1)
    tr = fw.HogwildTrainer(re_b, mi, micro_batch=600)
    tr.digest_records(recs, off)
    tr.block_until_workers_finished()
    for t in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC):
        assert re_a.table_checksum(t) == re_b.table_checksum(t)
    for x in (b, tr, re_a, re_b):
        x.close()


@pytest.mark.gpu
def test_table_placement_search_is_transparent(monkeypatch):
    """fwgpu_create looks for an accumulator allocation that does not contend with the weight table (tables beyond the
    Infinity Cache only; regressor.cpp place_ffm_acc).  Whatever it picks, the model is the same model: same initial tables,
    same sequential-mode results as with the search switched off; small tables are never searched."""
    mi, ocfg, ots = make_pair(6, 8, 18, 27, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)  # 2^27 floats = 512 MiB per FFM table
    recs, off = fw.synth_records(6, 0.5, 1.1, 5000, 0.1, 97, 0, 64)
    outs = []
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("FWGPU_PLACEMENT", raising=False)
        else:
            monkeypatch.setenv("FWGPU_PLACEMENT", env)
        re = fw.Regressor(mi)
        tries, lo, hi = re.placement()
        if env == "0":
            assert tries == 1 and lo == 0.0
        else:
            assert tries >= 1 and lo > 0.0 and hi >= lo
        b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        outs.append((b.predictions().copy(), re.table_checksum(capi.TABLE_FFM_W), re.table_checksum(capi.TABLE_FFM_ACC), re.table_checksum(capi.TABLE_LR)))
        b.close()
        re.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]
    small = fw.Regressor(make_pair(4, 4, 14, 14, fw.Optimizer.AdagradLUT)[0])
    assert small.placement() == (1, 0.0, 0.0)
    small.close()


def _chunked_path_parity(n_feat, opt, F=8, k=4, bits=16, ffm_bits=16):
    """The reference takes an example of any size (its gradient cache moves to the heap beyond 170 393 floats, block_ffm.rs:294-312).  The fused
    kernel stages at most 4096 entries; larger examples go through the synchronous pipeline chunk by chunk (regressor.cpp learn_one_chunked).
    Against the oracle: predict, three learn calls on three different huge examples -- rows repeated across chunks, rows that overlap rows of
    other chunks, LR hashes repeated across chunks included -- then an ordinary small example, per-call predictions and the final tables."""
    mi, ocfg, _ = make_pair(F, k, bits, ffm_bits, opt, lr=0.01, ffm_lr=0.01)
    om = fwo.Model(ocfg)
    re = fw.Regressor(mi)
    rng = np.random.default_rng(n_feat)
    R = F * k

    def example(n):
        fld = np.sort(rng.integers(0, F, size=n))
        h = (rng.integers(0, (1 << ffm_bits) // k, size=n) * k).astype(np.int64)  # (k = 4 or 16: a power of two, the row grid)
        h[n // 2] = h[7]                    # the same row 3000+ entries apart (another chunk) ...
        h[n - 5] = h[11] + k                # ... and a row that overlaps a row of an early chunk
        v = rng.uniform(0.002, 0.02, size=n)
        ffm = [(int(h[i]), float(v[i]), int(fld[i]) * k) for i in range(n)]
        lh = rng.integers(0, 1 << bits, size=n)
        lh[n - 3] = lh[2]                   # a repeated LR hash across chunks
        lr = [(int(lh[i]), float(v[i]), int(fld[i])) for i in range(n)] + [(11650396 & ((1 << bits) - 1), 1.0, F)]
        return lr, ffm

    worst = 0.0
    for step in range(5):
        lr, ffm = example(n_feat if step < 4 else 40)
        label = float(step % 2)
        fb = fw.lr_and_ffm_vec(lr, ffm, label=label)
        if step == 0:
            p_gpu, p_cpu = re.predict(fb), om.predict(fwo.lr_entries(lr), fwo.ffm_entries(ffm))
        else:
            p_gpu, p_cpu = re.learn(fb, None, True), om.learn(fwo.lr_entries(lr), fwo.ffm_entries(ffm), label, 1.0, True)
        worst = max(worst, abs(p_gpu - p_cpu))
    assert worst < 2e-5, worst
    for which, ref in ((capi.TABLE_FFM_W, om.ffm_weights), (capi.TABLE_FFM_ACC, om.ffm_acc), (capi.TABLE_LR, om.lr_table)):
        got, ref = re.table_read(which), np.asarray(ref).reshape(-1)
        bad = np.abs(got - ref[:len(got)]) > 2e-5 + 1e-4 * np.abs(ref[:len(got)])
        assert int(bad.sum()) <= 3, (which, int(bad.sum()), float(np.abs(got - ref[:len(got)]).max()))
    # the same through a BATCH that holds one oversize example among ordinary ones (walked example by example, in order)
    fbs, want = [], []
    for step in range(3):
        lr, ffm = example(n_feat if step == 1 else 30)
        fbs.append(fw.lr_and_ffm_vec(lr, ffm, label=float(step % 2)))
        want.append(om.learn(fwo.lr_entries(lr), fwo.ffm_entries(ffm), float(step % 2), 1.0, True))
    b = re.batch(fbs)
    re.learn_batch(b, capi.MODE_HOGWILD, True)
    assert np.abs(b.predictions() - np.array(want, dtype=np.float32)).max() < 2e-5
    b.close()
    re.close()


# ------------------------------------------------------------------ examples beyond what a workgroup stages (block_ffm.rs:294-312)
@pytest.mark.parametrize("n_feat,opt", [(6000, fw.Optimizer.AdagradLUT), (20000, fw.Optimizer.AdagradFlex), (20000, fw.Optimizer.SGD)])
def test_examples_beyond_4096_features_take_the_chunked_path(n_feat, opt):
    _chunked_path_parity(n_feat, opt)


def test_batch_with_an_example_below_4096_entries_that_does_not_fit_the_lds():
    """3 500 features at 30 fields x k = 8 are fewer than the 4096 entries a workgroup indexes but more than the LDS holds (own slots alone: 112 KB).  A batch
    that holds such an example used to come back with FWGPU_ERR_RANGE from the launch; the batch's host copy is now decided by the kernel's own LDS check, and the example
    takes the chunked path inside the batch like any oversize one."""
    _chunked_path_parity(3500, fw.Optimizer.AdagradLUT, F=30, k=8, bits=16, ffm_bits=20)


def test_chunked_path_where_the_phase_kernels_keep_t_in_the_split_record():
    """30 fields x k = 16: the staged T (57.6 KB) would leave one workgroup per CU, so the phase kernels write / read the field sums in the example's split
    record directly (kernels.hip launch_example_phase, KernelParams::t_global).  An oversize example's chunks hold different fields each: the update of
    a chunk must see the WHOLE example's sums, the fields that are empty in the chunk included."""
    _chunked_path_parity(6000, fw.Optimizer.AdagradLUT, F=30, k=16, bits=18, ffm_bits=22)


def test_oversize_example_with_a_deep_head_takes_the_chunked_path():
    """6 000 features + a 2 x 256 ReLU head.  The chunks' partial records add up (per-combo LR sums, field sums, corrections, counts),
    MID forms the head's input from the total once, the head runs once on it (one example = the reference's per-example rule, block_neural.rs:252-340),
    every chunk's update takes the per-slot gradients.  Against the oracle's per-example head: a predict, three learn calls on huge examples, an ordinary
    example through the fused kernel; per-call predictions and the final NN / FFM / LR tables."""
    F, k, bits, ffm_bits, n_feat = 8, 4, 16, 16, 6000
    layers = [(256, "relu", "hu"), (256, "relu", "hu")]
    mi, ocfg, _ = make_pair(F, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, lr=0.01, ffm_lr=0.01)
    mi.nn_layers = [dict(width=w, activation=a, init=i) for w, a, i in layers]
    mi.nn_topology = "one"
    mi.nn_learning_rate, mi.nn_power_t, mi.nn_init_acc_gradient = 0.01, 0.45, 1.0
    om = fwo.Model(ocfg, nn=fwo.make_nn_config(layers, "one", 0.01, 0.45, 1.0))
    re = fw.Regressor(mi)
    L = len(layers)
    re.table_write(capi.TABLE_NN_W, np.concatenate([om.nn_weights(l).copy() for l in range(L + 1)]))  # (Hu draws are implementation-defined: same ones)
    rng = np.random.default_rng(4242)

    def example(n):
        fld = np.sort(rng.integers(0, F, size=n))
        h = (rng.integers(0, (1 << ffm_bits) // k, size=n) * k).astype(np.int64)
        h[n // 2] = h[7]
        h[n - 5] = h[11] + k
        v = rng.uniform(0.002, 0.02, size=n)
        ffm = [(int(h[i]), float(v[i]), int(fld[i]) * k) for i in range(n)]
        lh = rng.integers(0, 1 << bits, size=n)
        lh[n - 3] = lh[2]
        lr = [(int(lh[i]), float(v[i]), int(fld[i])) for i in range(n)] + [(11650396 & ((1 << bits) - 1), 1.0, F)]
        return lr, ffm

    worst = 0.0
    for step in range(5):
        lr, ffm = example(n_feat if step < 4 else 40)
        label = float(step % 2)
        fb = fw.lr_and_ffm_vec(lr, ffm, label=label)
        if step == 0:
            p_gpu, p_cpu = re.predict(fb), om.predict(fwo.lr_entries(lr), fwo.ffm_entries(ffm))
        else:
            p_gpu, p_cpu = re.learn(fb, None, True), om.learn(fwo.lr_entries(lr), fwo.ffm_entries(ffm), label, 1.0, True)
        worst = max(worst, abs(p_gpu - p_cpu))
    assert worst < 5e-5, worst
    w1 = np.concatenate([om.nn_weights(l) for l in range(L + 1)])
    a1 = np.concatenate([om.nn_acc(l) for l in range(L + 1)])
    for which, ref in ((capi.TABLE_NN_W, w1), (capi.TABLE_NN_ACC, a1), (capi.TABLE_FFM_W, om.ffm_weights), (capi.TABLE_FFM_ACC, om.ffm_acc),
                       (capi.TABLE_LR, om.lr_table)):
        got, ref = re.table_read(which), np.asarray(ref).reshape(-1)
        bad = np.abs(got - ref[:len(got)]) > 3e-5 + 1e-4 * np.abs(ref[:len(got)])
        assert int(bad.sum()) <= 3, (which, int(bad.sum()), float(np.abs(got - ref[:len(got)]).max()))
    re.close()


def test_oversize_records_through_record_batches_and_the_trainer():
    """Records that translate to more entries than a workgroup stages (two namespaces of 3000 features each: 6006 FFM features, 6007 LR entries)
    between ordinary ones: a raw-record batch with such a record is translated on the host and walked in order, and so is the trainer's
    micro-batch that holds it -- predictions against the oracle's run over the same records."""
    F, k, bits, ffm_bits = 8, 4, 16, 16
    mi, ocfg, ots = make_pair(F, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, lr=0.01, ffm_lr=0.01)
    rng = np.random.default_rng(77)

    def record(n_big, label):
        slots, tail = [], []
        pos = 3 + F
        for ns in range(F):
            n = n_big if ns < 2 else 1
            if n == 1:
                slots.append(int(rng.integers(0, 1 << 31)))
            else:
                slots.append(0x80000000 | (pos << 16) | (pos + 2 * n))
                for _ in range(n):
                    tail += [int(rng.integers(0, 1 << 31)), int(np.float32(rng.uniform(0.002, 0.02)).view(np.uint32))]
                pos += 2 * n
        rec = [3 + F + len(tail), label, int(np.float32(1.0).view(np.uint32))] + slots + tail
        return np.array(rec, dtype=np.uint32)

    recs_l = [record(1, 1), record(3000, 0), record(1, 1), record(3000, 1), record(1, 0)]
    recs = np.concatenate(recs_l)
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs_l])]).astype(np.uint64)
    om = fwo.Model(ocfg)
    _, p_ref = om.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
    fbt = fw.FeatureBufferTranslator(mi)
    re = fw.Regressor(mi)
    b = re.record_batch(fbt, recs, off)
    re.learn_batch(b, capi.MODE_HOGWILD, True)
    assert np.abs(b.predictions() - p_ref).max() < 2e-5, (b.predictions(), p_ref)
    b.close()
    assert np.abs(re.table_read(capi.TABLE_FFM_W) - om.ffm_weights).max() < 2e-5
    re.close()
    # the trainer (hogwild.rs:51-60): one worker's worth of concurrency, so that the result is the sequential one
    re = fw.Regressor(mi)
    re.set_max_in_flight(1)
    tr = fw.HogwildTrainer(re, mi, micro_batch=4)
    tr.digest_records(recs, off)
    tr.block_until_workers_finished()
    assert tr.examples_seen() == 5
    assert np.abs(re.table_read(capi.TABLE_FFM_W) - om.ffm_weights).max() < 2e-5
    tr.close()
    re.close()
