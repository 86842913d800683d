"""The oracle's row-sparse update rule (fwo_learn_sparse: the multi-GPU "sparse" mode's restatement) pinned against the
oracle's own reference-following learn (fwo_learn, itself pinned to the reference's known-answer tests in test_oracle_kat.py):
where the two rules must coincide, they do, bit for bit."""
import numpy as np

import fwumious_wabbit_amd as fw
from helpers import make_pair, record_labels
from oracle import fwo


def _tables(m):
    return [np.array(m.lr_table), np.array(m.ffm_weights), np.array(m.ffm_acc)]


def test_one_example_without_repeated_or_overlapping_rows_is_the_reference_step():
    """a batch of ONE example whose rows are all distinct and disjoint: one summed-gradient step per row IS the per-occurrence
    update of block_ffm.rs:265-288 / block_lr.rs:135-150"""
    n_ns, k = 8, 4
    R = n_ns * k
    for opt in (fw.Optimizer.AdagradLUT, fw.Optimizer.AdagradFlex, fw.Optimizer.SGD):
        mi, ocfg, ots = make_pair(n_ns, k, 16, 18, opt, lr=0.05, ffm_lr=0.05)
        recs, off = fw.synth_records(n_ns, 0.0, 1.1, 3000, 0.1, 91, 0, 400)
        a, b = fwo.Model(ocfg), fwo.Model(ocfg)
        for m in (a, b):  # the same warm state on both sides
            m.learn_minibatch(ots, recs[:int(off[100])], off[:101])
        checked = 0
        for i in range(100, 400):
            rec = recs[int(off[i]):int(off[i + 1])]
            lr, ffm, label, imp = ots.translate(rec)
            fh = np.sort(ffm["hash"].astype(np.int64))
            if len(np.unique(lr["hash"])) != len(lr) or (len(fh) > 1 and np.diff(fh).min() < R):
                continue
            pa = a.learn(lr, ffm, label, imp)
            pb = b.learn_sparse(ots, rec, np.array([0, len(rec)], dtype=np.uint64))[0]
            assert pa == pb
            for x, y in zip(_tables(a), _tables(b)):
                assert np.array_equal(x, y), (opt, i)
            checked += 1
            if checked == 40:
                break
        assert checked >= 20


def test_partition_of_the_batch_changes_only_the_rounding():
    """the ranks' partition decides the ORDER the occurrence gradients are added in, nothing else"""
    mi, ocfg, ots = make_pair(10, 4, 14, 14, fw.Optimizer.AdagradFlex, lr=0.05, ffm_lr=0.05)
    recs, off = fw.synth_records(10, 1.0, 1.1, 2000, 0.1, 93, 0, 300)
    outs = []
    for parts in ([300], [150, 300], [64, 128, 200, 300]):
        m = fwo.Model(ocfg)
        p = m.learn_sparse(ots, recs, off, parts)
        outs.append((p, _tables(m)))
    for p, tabs in outs[1:]:
        assert np.array_equal(p, outs[0][0])  # predictions come from the batch-start weights: identical
        for x, y in zip(tabs, outs[0][1]):
            assert (np.abs(x - y) <= 1e-6 + 2e-6 * np.abs(y)).all()
            assert np.count_nonzero(x != y) < x.size  # (and mostly the same bits)


def test_repeated_row_takes_one_step_with_the_sum():
    """two examples sharing every row: the LR accumulator grows by (g1*v + g2*v)^2 once, not by two squares"""
    mi, ocfg, ots = make_pair(4, 0, 12, 12, fw.Optimizer.AdagradFlex, lr=0.1)
    recs, off = fw.synth_records(4, 0.0, 1.1, 50, 0.0, 95, 0, 1)
    rec = recs[:int(off[1])]
    two = np.concatenate([rec, rec])
    off2 = np.array([0, len(rec), 2 * len(rec)], dtype=np.uint64)
    m = fwo.Model(ocfg)
    lr, _, label, imp = ots.translate(rec)
    p = m.learn_sparse(ots, two, off2)
    assert p[0] == p[1]
    g = np.float32(-(np.float32(label) - p[0]) * np.float32(imp))
    tab = np.array(m.lr_table).reshape(-1, 2)
    for e in lr:
        G = np.float32(np.float32(g * e["value"]) + np.float32(g * e["value"]))
        assert tab[e["hash"], 1] == np.float32(np.float32(1.0) + G * G)  # init_acc 1.0 + (summed gradient)^2


def test_window_emulation_with_one_example_per_window_is_the_sequential_reference():
    """oracle/fw_oracle.c fwo_learn_window_emulation (round 6: an emulation of the device's concurrent mode, analysis only) must BE the sequential reference when a window holds one
    example and nothing is written back from the window's start -- bit for bit; with 64 examples per window and last-writer-wins weights it must differ, learn, and stay finite."""
    import numpy as np
    import fwumious_wabbit_amd as fw
    from helpers import logloss, make_pair, record_labels
    from oracle import fwo
    mi, ocfg, ots = make_pair(10, 4, 16, 16, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 1.0, 1.1, 3000, 0.1, 5, 0, 4000)
    y = record_labels(recs, off)
    a, b, c = fwo.Model(ocfg), fwo.Model(ocfg), fwo.Model(ocfg)
    _, p1 = a.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
    p2 = b.learn_window_emulation(ots, recs, off, 1, 0, want_preds=True)
    assert np.array_equal(p1, p2) and np.array_equal(a.ffm_weights, b.ffm_weights) and np.array_equal(a.ffm_acc, b.ffm_acc) and np.array_equal(a.lr_table, b.lr_table)
    p3 = c.learn_window_emulation(ots, recs, off, 64, 1, want_preds=True)
    assert np.all(np.isfinite(p3)) and not np.array_equal(p1, p3)
    ll = logloss(p3, y)
    assert ll[-1000:].mean() < ll[:1000].mean()
