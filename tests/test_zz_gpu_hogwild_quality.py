"""Statistical GPU tests of the CONCURRENT (hogwild) modes -- this file sorts after every exact-parity file on purpose: a
tolerance miss here must never hide an exact test behind `pytest -x` (GPUTEST_r02: one such miss left 40 tests unreached).

Concurrent training is a different, stale-gradient algorithm -- as is the reference's own hogwild mode (hogwild.rs:89-103,
multithread_helpers.rs:11-23) -- so every concurrent path is compared with the SEQUENTIAL ORACLE on the final hold-out
log-loss of a short stream (benchmark/calc_loss.py:5-25), never with another GPU path.

Every scenario is a function returning (gpu hold-out loss, sequential oracle's hold-out loss); the tests assert on the gap and
`scripts/holdout_spread.py` runs the same functions repeatedly to measure the spread the tolerance rests on.
"""
import os

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import logloss, make_pair, record_labels
from oracle import fwo

pytestmark = [pytest.mark.gpu, pytest.mark.statistical]

# The comparison has to be able to FAIL.  For every scenario:
#   * the tolerance is 1.3 x the largest |gpu hogwild - sequential oracle| measured over 8 runs of the shipped build
#     (scripts/holdout_spread.py -> profiles/r04_holdout_spread.txt; config A: profiles/r04_config_a_in_flight.txt), rounded up to 0.0005;
#   * the test itself checks that this tolerance is at most a THIRD of the scenario's learnable gap -- ln 2 (the loss of the untrained
#     model, which predicts 0.5) minus the sequential oracle's hold-out loss: a run that learns only two thirds of what the reference
#     learns fails.  Streams were lengthened until that holds (round 3's 12-24 k-example streams allowed half the gap).
# Stream families: Zipf(1.1) ids with a teacher FFM (synth_records), the same at Zipf(1.3) with another teacher and 5 % of the labels
# flipped (noisy: the loss floor is well above zero), and the reference's own example data (examples/ffm, config A).
TOL = {
    "short_fused": 0.0070, "short_sync": 0.0055, "hogwild_96k": 0.0075, "config_b": 0.0140, "two_chunk_wl1": 0.0145, "two_chunk_wl2": 0.0186,
    "trainer": 0.0050, "zipf13_noise": 0.0100, "zipf13_noise_k8_win": 0.0160, "config_a_hogwild": 0.0550,
}
LN2 = 0.6931

_ORACLE_CACHE = {}


def _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key=None):
    """hold-out / training loss of the sequential reference algorithm (regressor.rs:356-379; --holdout_after, main.rs:238-241)"""
    if key is not None and key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    om = fwo.Model(ocfg)
    _, p = om.run_stream(ots, recs, off, holdout_after=n_train + 1, nthreads=1)
    y = record_labels(recs, off)
    out = float(logloss(p[n_train:], y[n_train:]).mean()), float(logloss(p[:n_train], y[:n_train]).mean())
    if key is not None:
        _ORACLE_CACHE[key] = out
    return out


def _gpu_holdout(re, fbt, recs, off, y, n_train):
    hb = re.record_batch(fbt, recs[int(off[n_train]):], off[n_train:] - off[n_train])
    re.learn_batch(hb, capi.MODE_HOGWILD, False)
    out = float(logloss(hb.predictions(), y[n_train:]).mean())
    hb.close()
    return out


def _micro_batches(re, fbt, recs, off, n_train, mb, sync=False, hot_lr=None):
    if hot_lr is not None:
        re.set_hot_lr_entry(hot_lr)
    sp = re.split_buffers(mb, 64) if sync else None
    for s0 in range(0, n_train, mb):
        e = min(n_train, s0 + mb)
        b = re.record_batch(fbt, recs[int(off[s0]):int(off[e])], off[s0:e + 1] - off[s0])
        if sync:
            re.learn_batch_sync(b, sp, capi.MODE_HOGWILD)
        else:
            re.learn_batch(b, capi.MODE_HOGWILD, True)
        b.close()
    if sp is not None:
        sp.close()


def _flip_labels(recs, off, frac, seed):
    """label noise: a fraction of the records gets the other label (record word 1, parser.rs:57-60)"""
    rng = np.random.default_rng(seed)
    recs = recs.copy()
    idx = off[:-1].astype(np.int64) + 1
    f = rng.random(len(idx)) < frac
    recs[idx[f]] = 1 - recs[idx[f]]
    return recs


def scenario_short_launches(path, hot_lr=None):
    """2048-example launches of a 10-field model: a workgroup sees 2-3 examples per launch, so everything a workgroup keeps
    pending only reaches the table when it leaves.  `path`: "fused" (the hogwild kernel) or "sync" (the concurrent form of the
    synchronous micro-batch).  65 536 training examples in 32 launches."""
    n_train, n_hold, mb = 65536, 8192, 2048
    mi, ocfg, ots = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 20240611, 0, n_train + n_hold)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="short")
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    _micro_batches(re, fbt, recs, off, n_train, mb, sync=(path == "sync"), hot_lr=hot_lr)
    gpu_hold = _gpu_holdout(re, fbt, recs, off, y, n_train)
    re.close()
    return gpu_hold, ref_hold


def scenario_hogwild_96k(hot_lr=None):
    n_train, n_hold = 98304, 8192
    mi, ocfg, ots = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 20240611, 0, n_train + n_hold)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="96k")
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    _micro_batches(re, fbt, recs, off, n_train, 8192, hot_lr=hot_lr)
    gpu_hold = _gpu_holdout(re, fbt, recs, off, y, n_train)
    re.close()
    return gpu_hold, ref_hold


def scenario_config_b(hot_lr=None):
    """BASELINE configs[1]: 10 fields, k = 4, 22-bit FFM and LR tables, micro-batch 4096, seed 20240611"""
    n_train, n_hold = 10 * 4096, 8192
    mi, ocfg, ots = make_pair(10, 4, 22, 22, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 20240611, 0, n_train + n_hold)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="B")
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    _micro_batches(re, fbt, recs, off, n_train, 4096, hot_lr=hot_lr)
    gpu_hold = _gpu_holdout(re, fbt, recs, off, y, n_train)
    re.close()
    return gpu_hold, ref_hold


def scenario_two_chunk_rows(whole_lines, hot_lr=None):
    """k = 16 at 30 fields (R = 480: the v2 kernel's two-chunk instantiation) with chained duplicate rows, with and without
    whole-line accesses.  256 examples in flight: on a stream of 24 000 examples the gap to the sequential result is
    0.008-0.010 then (0.015-0.020 with the 512 the device would hold: the same for the generic kernel, measured side by side)."""
    n_train, n_hold = 24000, 3000
    mi, ocfg, ots = make_pair(30, 16, 18, 20, fw.Optimizer.AdagradLUT, lr=0.025, ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38)
    recs, off = fw.synth_records(30, 1.0, 1.1, 50000, 0.1, 20240613, 0, n_train + n_hold)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="k16")
    re = fw.Regressor(mi)
    re.set_whole_line_updates(whole_lines)
    re.set_max_in_flight(256)
    fbt = fw.FeatureBufferTranslator(mi)
    _micro_batches(re, fbt, recs, off, n_train, 3000, hot_lr=hot_lr)
    gpu_hold = _gpu_holdout(re, fbt, recs, off, y, n_train)
    re.close()
    return gpu_hold, ref_hold


def scenario_trainer(hot_lr=None):
    """HogwildTrainer::digest_example / block_until_workers_finished (hogwild.rs:51-60) over a record stream, a mix of the
    single-record and the bulk entry points.  256 examples in flight."""
    n_train, n_hold = 48000, 6000
    mi, ocfg, ots = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 77, 0, n_train + n_hold)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="trainer")
    re = fw.Regressor(mi)
    if hot_lr is not None:
        re.set_hot_lr_entry(hot_lr)
    re.set_max_in_flight(256)
    tr = fw.HogwildTrainer(re, mi, micro_batch=1024)
    for i in range(10):
        tr.digest_example(recs[int(off[i]):int(off[i + 1])])
    tr.digest_records(recs[int(off[10]):int(off[n_train])], off[10:n_train + 1] - off[10])
    tr.block_until_workers_finished()
    assert tr.examples_seen() == n_train
    gpu_hold = _gpu_holdout(re, fw.FeatureBufferTranslator(mi), recs, off, y, n_train)
    tr.close()
    re.close()
    return gpu_hold, ref_hold


def scenario_zipf13_noise(k8_win, hot_lr=None):
    """The second stream family: ids drawn Zipf(1.3) (a heavier head: the hot rows are hotter), another teacher (seed 4242), 5 % of
    the labels flipped.  `k8_win`: 20 fields, k = 8, ~40 features per example, weighted features, bench.py's hyper-parameters, and the
    update path of config C's large tables forced onto the 20-bit table (rows kept from the gather, duplicate-row chains, the shipped
    store policy) -- the kernel the headline number comes from, here against the sequential oracle."""
    n_train, n_hold = (int(os.environ.get("Z13_K8_TRAIN", 98304)) if k8_win else 65536), 8192
    if k8_win:
        mi, ocfg, ots = make_pair(20, 8, 20, 20, fw.Optimizer.AdagradLUT, lr=0.025, ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38)
        recs, off = fw.synth_records(20, 1.0, 1.3, 100000, 0.1, 4242, 0, n_train + n_hold)
    else:
        mi, ocfg, ots = make_pair(10, 4, 20, 20, fw.Optimizer.AdagradLUT)
        recs, off = fw.synth_records(10, 0.0, 1.3, 100000, 0.0, 4242, 0, n_train + n_hold)
    recs = _flip_labels(recs, off, 0.05, 99)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key=("z13", k8_win, n_train))
    re = fw.Regressor(mi)
    if k8_win:
        re.set_whole_line_updates(3)
    fbt = fw.FeatureBufferTranslator(mi)
    _micro_batches(re, fbt, recs, off, n_train, 8192, hot_lr=hot_lr)
    gpu_hold = _gpu_holdout(re, fbt, recs, off, y, n_train)
    re.close()
    return gpu_hold, ref_hold


def scenario_config_a_hogwild(hot_lr=None, in_flight=16):
    """BASELINE configs[0]'s model and data (examples/ffm: `--ffm_k 10 -l 0.1 -b 25 --adaptive --power_t 0.0 --noconstant`, the 30 000
    generated training lines, tests/golden/ffm_example) trained CONCURRENTLY: the first 25 000 lines through the trainer with 16 examples
    in flight -- the reference's own default thread count (main.rs:189-194); the data has 500 rows in all, every example holds two of them --
    the last 5 000 as hold-out, against the sequential oracle on the same lines."""
    import gzip
    import os
    from fwumious_wabbit_amd.feed import VowpalParser, VwNamespaceMap
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ffm_example")
    vw = VwNamespaceMap(gzip.open(os.path.join(data, "vw_namespace_map.csv.gz"), "rt").read())
    text = gzip.open(os.path.join(data, "train.vw.gz"), "rb").read()
    recs, off, used, rc = VowpalParser(vw).parse_buffer(text)
    assert rc == capi.OK and used == len(text) and len(off) == 30001
    n_train = 25000
    a, b = vw.lookup("A")[0], vw.lookup("B")[0]
    nd = fw.NamespaceDescriptor
    mi = fw.ModelInstance(learning_rate=0.1, ffm_learning_rate=0.1, power_t=0.0, ffm_power_t=0.0, bit_precision=25, ffm_k=10,
                          ffm_bit_precision=18, add_constant_feature=False, init_acc_gradient=1.0, ffm_init_acc_gradient=1.0,
                          optimizer=fw.Optimizer.AdagradLUT,
                          feature_combo_descs=[fw.FeatureComboDesc([nd(a)]), fw.FeatureComboDesc([nd(b)]), fw.FeatureComboDesc([nd(a), nd(b)])],
                          ffm_fields=[[nd(a)], [nd(b)]])
    ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=0.1, ffm_learning_rate=0.1, power_t=0.0, ffm_power_t=0.0,
                           init_acc_gradient=1.0, ffm_init_acc_gradient=1.0, bit_precision=25, num_combos=3, ffm_k=10,
                           ffm_bit_precision=18, ffm_num_fields=2)
    ots = fwo.TranslatorSpec([([(a, False)], 1.0), ([(b, False)], 1.0), ([(a, False), (b, False)], 1.0)], [[(a, False)], [(b, False)]], False, 25, 10, 18)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="A")
    re = fw.Regressor(mi)
    if hot_lr is not None:
        re.set_hot_lr_entry(hot_lr)
    re.set_max_in_flight(in_flight)
    tr = fw.HogwildTrainer(re, mi, micro_batch=1024)
    tr.digest_records(recs[:int(off[n_train])], off[:n_train + 1])
    tr.block_until_workers_finished()
    gpu_hold = _gpu_holdout(re, fw.FeatureBufferTranslator(mi), recs, off, y, n_train)
    tr.close()
    re.close()
    return gpu_hold, ref_hold


SCENARIOS = {
    "short_fused": lambda **kw: scenario_short_launches("fused", **kw),
    "short_sync": lambda **kw: scenario_short_launches("sync", **kw),
    "hogwild_96k": scenario_hogwild_96k,
    "config_b": scenario_config_b,
    "two_chunk_wl1": lambda **kw: scenario_two_chunk_rows(1, **kw),
    "two_chunk_wl2": lambda **kw: scenario_two_chunk_rows(2, **kw),
    "trainer": scenario_trainer,
    "zipf13_noise": lambda **kw: scenario_zipf13_noise(False, **kw),
    "zipf13_noise_k8_win": lambda **kw: scenario_zipf13_noise(True, **kw),
    "config_a_hogwild": scenario_config_a_hogwild,
}


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_concurrent_training_reaches_the_sequential_oracles_holdout_loss(name):
    gpu_hold, ref_hold = SCENARIOS[name]()
    gap = LN2 - ref_hold  # what the reference learns on this stream
    print(f"hold-out [{name}]: gpu hogwild {gpu_hold:.4f}, sequential oracle {ref_hold:.4f}, learnable gap {gap:.4f}, tolerance {TOL[name]:.4f}")
    assert TOL[name] <= gap / 3 + 1e-9, (name, TOL[name], gap)  # the test can fail: a third of the gap at most
    assert abs(gpu_hold - ref_hold) < TOL[name], (name, gpu_hold, ref_hold)


def test_deep_head_hogwild_learns():
    mi, ocfg, ots = make_pair(8, 4, 16, 16, fw.Optimizer.AdagradLUT, lr=0.025, ffm_lr=0.025, power_t=0.38, ffm_power_t=0.38)
    mi.nn_layers = [dict(width=16, activation="relu", init="hu")]
    recs, off = fw.synth_records(8, 1.0, 1.1, 2000, 0.1, 27, 0, 60000)
    y = record_labels(recs, off)
    re = fw.Regressor(mi)
    b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re.learn_batch(b, capi.MODE_HOGWILD, True)
    p = b.predictions()
    assert np.all(np.isfinite(p))
    ll = logloss(p, y)
    assert ll[-10000:].mean() < ll[:10000].mean() - 0.01 and ll[-10000:].mean() < 0.69
    b.close()
    re.close()
