"""What a HOGWILD launch does to the steps of FFM rows that MANY concurrent examples hold -- the guard of the FFM row store
policy (kernels.hip "store policy", fwgpu_debug_set_option 5 / 6), in the spirit of the hot LR entry's conservation test.

The reference's hogwild threads share one table in cache-coherent memory (hogwild.rs:89-103, multithread_helpers.rs:11-23):
a step can be overwritten by a concurrent writer of the same float, but it cannot stay invisible to the other threads.  On the
device the eight XCDs' L2s are not coherent with each other, so a store policy that keeps a popular row dirty in one L2 would
let every XCD step a private copy for a whole launch and keep one of the eight at the end.  This file MEASURES how much of the
steps on such rows arrives, per policy, and asserts floors for the shipped one.

Construction (everything recomputable from the launch's own predictions).  30 fields, k = 8 as in BASELINE config C, ~200
features per example.  One field holds the "hot" feature: one of H row hashes, value 1.0, so every hot row sits in 1/H of the
examples.  Every other field holds 7 "partner" features with value 2^-20 on weights preset to 2^11: their products are ordinary
numbers (2^-9), but a partner's own step (~1e-9) is far below the ulp of its weight (2.4e-4), so the partners never move and
the hot feature's gradient cache is the SAME constant in every example:  G = v_hot * sum_j w_j v_j = 7 * 2^-9  for every float
of the row outside the hot field's own slot (block_ffm.rs:219-261).  With g_e = -(label - p_e) from the launch's predictions
(block_loss_functions.rs:141):
  * SGD (optimizer.rs:36-38): every example e holding row r steps each such float by  -lr * g_e * G, so without losses
    w_final - w_0 = -lr * G * sum_e g_e.          surviving fraction S_w  = (w_final - w_0) / that.
  * AdaGrad (optimizer.rs:147-149): acc += (g_e G)^2. surviving fraction S_acc = (acc_final - acc_0) / sum_e (g_e G)^2.
Labels are all 1, so every g_e has the same sign and the sums do not cancel.

What to expect.  Hogwild loses steps BY DESIGN: ~512 examples are in flight, a row in a third of them is being read-modified-
written by ~170 workgroups at any moment, and the update of a row kept from the gather writes w_gather - step (DESIGN.md 4.1:
that damping is what the concurrent mode's quality rests on).  So the fractions are far below 1 under EVERY policy, the
write-through one included; what a sound policy must not do is lose an order of magnitude MORE than write-through does, or
lose more the longer the launch is.
"""
import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi

pytestmark = pytest.mark.gpu

F, K, M_PARTNERS, H_HOT = 30, 8, 7, 3
R = F * K
W_PARTNER, V_PARTNER = 2048.0, 2.0 ** -20
G_CONST = M_PARTNERS * W_PARTNER * V_PARTNER  # exact in f32: 7 * 2^-9
W_HOT0 = 0.01
HOT_BASE = 1 << 12  # hot row hashes: HOT_BASE * (1 + q), q < H_HOT -- more than a row apart, multiples of 8


def _hot_hashes():
    return [HOT_BASE * (1 + q) for q in range(H_HOT)]


def crafted_records(n, hot_field, ffm_bits, seed):
    """n records (parser.rs:57-74): the hot field's slot holds a single hash (value 1.0), every other field a run of M_PARTNERS
    (hash, value) pairs with value 2^-20 whose FFM rows stay clear of the hot rows.  All labels 1, importance 1."""
    rng = np.random.default_rng(seed)
    L = 3 + F + 2 * M_PARTNERS * (F - 1)
    recs = np.zeros((n, L), dtype=np.uint32)
    recs[:, 0] = L
    recs[:, 1] = 1
    recs[:, 2] = np.float32(1.0).view(np.uint32)
    which = rng.integers(0, H_HOT, size=n)
    hot = np.array(_hot_hashes(), dtype=np.uint32)[which]
    mask = (1 << ffm_bits) - 1
    ph = rng.integers(0, 1 << 31, size=(n, F - 1, M_PARTNERS), dtype=np.int64).astype(np.uint32)
    # keep the partners' rows (R floats from hash & mask) away from every hot row: the low bits decide, so move offenders far away
    start = ph & np.uint32(mask & ~7)
    for hh in _hot_hashes():
        near = (start.astype(np.int64) > hh - 2 * R) & (start.astype(np.int64) < hh + 2 * R)
        ph = np.where(near, ph ^ np.uint32(1 << (ffm_bits - 1)), ph)
    vbits = np.float32(V_PARTNER).view(np.uint32)
    col = 3 + F
    pf = 0
    for f in range(F):
        if f == hot_field:
            recs[:, 3 + f] = hot
            continue
        recs[:, 3 + f] = 0x80000000 | (col << 16) | (col + 2 * M_PARTNERS)
        recs[:, col:col + 2 * M_PARTNERS:2] = ph[:, pf, :]
        recs[:, col + 1:col + 2 * M_PARTNERS:2] = vbits
        col += 2 * M_PARTNERS
        pf += 1
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    return recs.reshape(-1), off, which


def make_model(opt, ffm_bits, ffm_lr):
    return fw.ModelInstance(
        learning_rate=1e-6, ffm_learning_rate=ffm_lr, power_t=0.38, ffm_power_t=0.38, init_acc_gradient=1.0, ffm_init_acc_gradient=1.0,
        bit_precision=18, ffm_bit_precision=ffm_bits, ffm_k=K, add_constant_feature=True, optimizer=opt,
        feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(F)],
        ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(F)])


_RECORDS = {}


def _records(n, hot_field, ffm_bits, seed):
    key = (n, hot_field, ffm_bits, seed)
    if key not in _RECORDS:
        _RECORDS.clear()  # (one stream at a time: the 65 536-example ones are 115 MB each)
        _RECORDS[key] = crafted_records(n, hot_field, ffm_bits, seed)
    return _RECORDS[key]


class Rig:
    """one regressor per (optimizer, table size), reset between launches: the partners never move, so only the hot rows and the
    accumulator table have to be put back"""

    def __init__(self, opt, ffm_bits):
        self.opt, self.ffm_bits, self.lr = opt, ffm_bits, 2.0 ** -11
        self.mi = make_model(opt, ffm_bits, self.lr)
        self.re = fw.Regressor(self.mi)
        self.re.set_whole_line_updates(3)  # the update path of config C's tables (chains, rows kept from the gather), on any table size
        self.re.table_fill(capi.TABLE_FFM_W, W_PARTNER)
        self.acc0 = float(self.re.table_read(capi.TABLE_FFM_ACC, 0, 1)[0])
        self.fbt = fw.FeatureBufferTranslator(self.mi)

    def close(self):
        self.re.close()

    def run(self, n, hot_field, policy, flush_every, mode=capi.MODE_HOGWILD, seed=5, lds_keep=-1, acc_start=None):
        """one launch of n crafted examples; per hot row (mean, min, max surviving fraction over the row's floats, share of examples).
        `acc_start`: the accumulators' value before the launch (default: the optimizer's initial value) -- above the hot-row threshold of
        policies 3 / 4 the rig's rows are hot from the first example on"""
        re = self.re
        re.set_store_policy(policy, flush_every)
        re.set_lds_keep(lds_keep)  # (-1: the shipped choice; 0: rows beyond the register-kept ones are re-read by the update)
        acc0 = self.acc0 if acc_start is None else float(acc_start)
        re.table_fill(capi.TABLE_FFM_ACC, acc0)
        for hh in _hot_hashes():
            re.table_write(capi.TABLE_FFM_W, np.full(R, W_HOT0, dtype=np.float32), hh)
        recs, off, which = _records(n, hot_field, self.ffm_bits, seed)
        b = re.record_batch(self.fbt, recs, off)
        lr_acc0 = [float(re.table_read(capi.TABLE_LR, 2 * (hh & ((1 << 18) - 1)), 2)[1]) for hh in _hot_hashes()]  # (the hot feature's LR entries: {w, acc} pairs, 18-bit table)
        re.learn_batch(b, mode, True)
        p = b.predictions().astype(np.float64)
        b.close()
        g = -(1.0 - p)  # block_loss_functions.rs:141, label 1, importance 1
        assert np.all(np.isfinite(p)) and np.all(np.abs(g) > 1e-4), "the crafted stream must not saturate the sigmoid"
        sel = np.ones(R, dtype=bool)
        sel[hot_field * K:(hot_field + 1) * K] = False  # the hot field's own slot gets no gradient (a lone feature has no intra-field partner)
        out = []
        for q, hh in enumerate(_hot_hashes()):
            ge = g[which == q]
            if self.opt == fw.Optimizer.SGD:
                w = re.table_read(capi.TABLE_FFM_W, hh, R).astype(np.float64)
                assert np.allclose(w[~sel], W_HOT0, rtol=0, atol=1e-9)
                frac = (w[sel] - W_HOT0) / (-self.lr * G_CONST * ge.sum())
            else:
                a = re.table_read(capi.TABLE_FFM_ACC, hh, R).astype(np.float64)
                frac = (a[sel] - acc0) / ((ge * G_CONST) ** 2).sum()
            # the hot feature's LR entry (value 1, block_lr.rs:143-147): its accumulator must have grown by sum g_e^2 if every example's g^2 was counted
            lr_acc1 = float(re.table_read(capi.TABLE_LR, 2 * (hh & ((1 << 18) - 1)), 2)[1])
            lr_frac = (lr_acc1 - lr_acc0[q]) / float((ge ** 2).sum()) if self.opt != fw.Optimizer.SGD else float("nan")
            out.append((float(np.mean(frac)), float(np.min(frac)), float(np.max(frac)), float(len(ge)) / n, lr_frac))
        # the partners never moved: spot-check a row
        assert np.all(re.table_read(capi.TABLE_FFM_W, 20000, 64) == W_PARTNER)
        return out


def test_in_order_launch_applies_every_step_under_every_policy():
    """SEQUENTIAL launches are exact whatever the store policy: all steps arrive (fraction 1 up to f32 summation order)."""
    for opt in (fw.Optimizer.SGD, fw.Optimizer.AdagradLUT):
        rig = Rig(opt, 18)
        for policy in (0, 1, 2, 3, 4):
            for mean, lo, hi, share, _lr in rig.run(600, 29, policy, 4, mode=capi.MODE_SEQUENTIAL, seed=3):
                assert abs(lo - 1.0) < 2e-3 and abs(hi - 1.0) < 2e-3, (opt, policy, lo, hi)
        rig.close()


# (policy, write-back interval): (-1, -1) = what the build ships (round 5: policy 3 = policy 1 with thinned accumulator stores on hot register-kept rows; policy 1: weight rows write-back, accumulators write-through; one buffer_wbl2 per
# workgroup every 128 examples); the asserts are about it, the others are measured next to it and printed
SHIPPED = (-1, -1)
MEASURED = [(0, 0), (1, 0), (2, 0), (2, 64), (3, 0), (4, 0)]  # (3: round 5, thinned accumulator stores on hot rows; 4: round 6, thinned atomic adds)
REREAD = "-1,-1 L0"

# Measured on MI355X (profiles/r04c_conservation.txt, r04b_conservation.txt; rows in a third of all examples, ~170 concurrent holders):
#   weights, SGD      write-through 0.014-0.036 (row kept from the gather) / 0.22-0.46 (row re-read in the update); write-back 0.010-0.027 / 0.083-0.114,
#                     whatever the write-back interval down to 8 examples: with the row dirty in eight L2s at once the last write-back wins, an eighth
#                     of the within-XCD survival (tools/l2probe: S2), until the interval falls below the row's hit interval per XCD (interval 1: 0.16-0.27)
#   accumulators      write-through 0.40-0.49 on an 18-bit table, 0.22-0.27 on a 28-bit one (longer read-modify-write window); write-back 0.10-0.12
# The floors sit a third to a half below the smallest value measured for the shipped policy.
# (FLOOR_W_KEPT = half of what the shipped build measures -- 0.0107-0.0111 at the smallest, profiles/r05_conservation.txt, r04c_conservation.txt; the re-read row 0.085-0.11)
FLOOR_W_KEPT, FLOOR_W_REREAD, FLOOR_ACC = 0.0055, 0.05, 0.12


@pytest.mark.parametrize("ffm_bits", [18, 28])
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_hot_ffm_rows_keep_a_bounded_share_of_their_steps(opt, ffm_bits, capsys):
    o = fw.Optimizer.SGD if opt == "sgd" else fw.Optimizer.AdagradLUT
    table = {}
    rig = Rig(o, ffm_bits)
    sizes = (2048, 16384, 65536)
    for n in sizes:
        for hot_field in (0, 29):  # first feature of wave 0 / last feature of the last wave: one is a row kept from the gather, the other re-read in the update
            for pol in [SHIPPED] + MEASURED:
                fr = rig.run(n, hot_field, pol[0], pol[1])
                table[(n, hot_field, pol)] = float(np.mean([f[0] for f in fr]))
            # the shipped policy with no row parked in LDS: the last feature of the last wave is then a row the update RE-READS
            table[(n, hot_field, REREAD)] = float(np.mean([f[0] for f in rig.run(n, hot_field, -1, -1, lds_keep=0)]))
    rig.close()
    with capsys.disabled():
        print(f"\nsurviving fraction of hot FFM rows' steps, {opt}, {ffm_bits}-bit table (mean over {H_HOT} rows, each in 1/{H_HOT} of the examples)")
        print("  launch  hot_field " + " ".join(f"{str(p):>9}" for p in [SHIPPED] + MEASURED + [REREAD]))
        for n in sizes:
            for hot_field in (0, 29):
                print(f"  {n:6d}  {hot_field:9d} " + " ".join(f"{table[(n, hot_field, p)]:9.4f}" for p in [SHIPPED] + MEASURED + [REREAD]))
    for n in sizes:
        if opt == "sgd":  # with no row parked in LDS one of the two positions is a row kept from the gather, the other a row re-read in the update (which is which is the kernel's choice)
            assert max(table[(n, 0, REREAD)], table[(n, 29, REREAD)]) >= FLOOR_W_REREAD, (n, table[(n, 0, REREAD)], table[(n, 29, REREAD)])
        for hot_field in (0, 29):
            shipped, wt = table[(n, hot_field, SHIPPED)], table[(n, hot_field, (0, 0))]
            if opt == "sgd":
                assert shipped >= FLOOR_W_KEPT, (n, hot_field, shipped)
                # eight XCDs' L2s hold the row: the write-back policy may keep an eighth of what device-scope write-through stores keep, not less
                assert shipped >= 0.8 / 8 * wt, (n, hot_field, shipped, wt)
            else:
                assert shipped >= FLOOR_ACC, (n, hot_field, shipped)
                # the accumulators must count what write-through stores count: a private accumulator makes the steps of exactly the most contended rows too large
                # (round 5's shipped policy thins the stores of hot register-kept rows -- one example in eight stores eight times its g^2: measured 0.83-0.88 of
                # write-through on the rig's rows in 16 384- and 65 536-example launches, 0.77-0.86 in the 2048-example one, whose ~230 hits per row are the noisier)
                assert shipped >= (0.8 if n > 2048 else 0.7) * wt, (n, hot_field, shipped, wt)
    # ... and nothing may get worse with the LENGTH of the launch (a row that stays private to an XCD until the launch ends would)
    for hot_field in (0, 29):
        assert table[(65536, hot_field, SHIPPED)] >= 0.5 * table[(2048, hot_field, SHIPPED)], table


@pytest.mark.parametrize("ffm_bits", [18, 28])
def test_thinned_atomic_adds_count_every_gradient_on_hot_rows(ffm_bits, capsys):
    """Store policy 4 (round 6): on a row that is hot -- its accumulators beyond the threshold -- one example in eight ADDS eight times its g^2 with
    device-scope float atomics and nobody stores.  An add cannot lose a race, so what reaches memory is an unbiased estimate of the TRUE sum of g^2 over
    all concurrent examples, which is what the reference's hogwild threads count (`acc += g * g` on coherent memory, optimizer.rs:147-149) -- where the
    store policies keep 0.15-0.45 of it (the table of the test above).  The rows start hot (accumulators at 2.0: beyond the default threshold of 0.5)."""
    rig = Rig(fw.Optimizer.AdagradLUT, ffm_bits)
    table, lr_table = {}, {}
    sizes = (2048, 16384, 65536)
    for n in sizes:
        for hot_field in (0, 29):
            for pol in (3, 4):
                fr = rig.run(n, hot_field, pol, 0, acc_start=2.0)
                table[(n, hot_field, pol)] = float(np.mean([f[0] for f in fr]))
                lr_table[(n, hot_field, pol)] = float(np.mean([f[4] for f in fr]))
            table[(n, hot_field, "4 L0")] = float(np.mean([f[0] for f in rig.run(n, hot_field, 4, 0, acc_start=2.0, lds_keep=0)]))
    rig.close()
    with capsys.disabled():
        print(f"\nshare of the true sum of g^2 that reaches a HOT row's accumulators, adagrad, {ffm_bits}-bit table (rows hot from the start)")
        print("  launch  hot_field  policy 3  policy 4  policy 4, nothing parked in LDS")
        for n in sizes:
            for hot_field in (0, 29):
                print(f"  {n:6d}  {hot_field:9d} {table[(n, hot_field, 3)]:9.4f} {table[(n, hot_field, 4)]:9.4f} {table[(n, hot_field, '4 L0')]:9.4f}")
    with capsys.disabled():
        print("  the hot feature's LR entry (8-byte {w, acc} pair): share of sum g^2 its accumulator received -- policy 3: every holder stores the pair; policy 4: the weight alone is")
        print("  stored and one example in eight adds eight times its g^2 once the entry is hot (accumulator > 32: after ~130 of its hits)")
        for n in sizes:
            print(f"  {n:6d}  " + "  ".join(f"field {hf}: policy 3 {lr_table[(n, hf, 3)]:.4f}, policy 4 {lr_table[(n, hf, 4)]:.4f}" for hf in (0, 29)))
    for n in sizes[1:]:  # (the entry is hot from its ~130th hit on: 2 % of a 16 384-example launch's hits come before)
        for hot_field in (0, 29):
            assert lr_table[(n, hot_field, 4)] >= 0.9, (n, hot_field, lr_table[(n, hot_field, 4)])
    for n in sizes:
        tol = 0.1 if n > 2048 else 0.3  # (one example in eight is drawn: ~85 draws per row in the 2048-example launch)
        for hot_field in (0, 29):
            for key in (4, "4 L0"):
                assert abs(table[(n, hot_field, key)] - 1.0) <= tol, (n, hot_field, key, table[(n, hot_field, key)])
