"""Statistical GPU tests of the CONCURRENT (hogwild) modes -- this file sorts after every exact-parity file on purpose: a
tolerance miss here must never hide an exact test behind `pytest -x` (GPUTEST_r02: one such miss left 40 tests unreached).

Concurrent training is a different, stale-gradient algorithm -- as is the reference's own hogwild mode (hogwild.rs:89-103,
multithread_helpers.rs:11-23) -- so every concurrent path is compared with the SEQUENTIAL ORACLE on the final hold-out
log-loss of a short stream (benchmark/calc_loss.py:5-25), never with another GPU path.

Every scenario is a function returning (gpu hold-out loss, sequential oracle's hold-out loss); the tests assert on the gap and
`scripts/holdout_spread.py` runs the same functions repeatedly to measure the spread the tolerance rests on.
"""
import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import _capi as capi
from helpers import logloss, make_pair, record_labels
from oracle import fwo

pytestmark = [pytest.mark.gpu, pytest.mark.statistical]

# 12-40k-example streams, loss still falling fast: the whole learnable gap of these streams is 0.04-0.09.  Spread of
# |gpu hogwild - sequential oracle| on the round-3 build (scripts/holdout_spread.py, 8 runs per scenario and hot-LR route):
# profiles/r03d_holdout_spread.txt -- hogwild_24k <= 0.0117, config_b <= 0.0097, two_chunk <= 0.0145, trainer <= 0.0083.
HOLDOUT_TOL = 0.02
# The two 2048-example-launch scenarios train on only 16 k examples in eight launches: the first launches run on fresh accumulators
# with 768 examples (fused) or the whole 2048-example batch (synchronous) in flight, and their gap to the sequential result is the
# widest of all: 0.0116 .. 0.0195 over 46 runs of the final builds (profiles/r03d_holdout_spread.txt and its two predecessors in
# gpurun history).  GPUTEST_r02's regression read 0.049 on the same scenario.
SCENARIO_TOL = {"short_fused": 0.03, "short_sync": 0.03}

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


def scenario_short_launches(path, hot_lr=None):
    """2048-example launches of a 10-field model: a workgroup sees 2-3 examples per launch, so everything a workgroup keeps
    pending only reaches the table when it leaves.  `path`: "fused" (the hogwild kernel) or "sync" (the concurrent form of the
    synchronous micro-batch)."""
    n_train, n_hold, mb = 16384, 4096, 2048
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


def scenario_hogwild_24k(hot_lr=None):
    n_train, n_hold = 24000, 4000
    mi, ocfg, ots = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 20240611, 0, n_train + n_hold)
    y = record_labels(recs, off)
    ref_hold, _ = _holdout_loss_oracle(ocfg, ots, recs, off, n_train, key="24k")
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    _micro_batches(re, fbt, recs, off, n_train, 2048, hot_lr=hot_lr)
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
    single-record and the bulk entry points.  256 examples in flight: the gap to the sequential oracle is 0.005 .. 0.008 then;
    with the ~500 the device would hold for these tiny examples its tail comes too close to the tolerance."""
    n_train, n_hold = 12000, 2000
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


SCENARIOS = {
    "short_fused": lambda **kw: scenario_short_launches("fused", **kw),
    "short_sync": lambda **kw: scenario_short_launches("sync", **kw),
    "hogwild_24k": scenario_hogwild_24k,
    "config_b": scenario_config_b,
    "two_chunk_wl1": lambda **kw: scenario_two_chunk_rows(1, **kw),
    "two_chunk_wl2": lambda **kw: scenario_two_chunk_rows(2, **kw),
    "trainer": scenario_trainer,
}


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_concurrent_training_reaches_the_sequential_oracles_holdout_loss(name):
    gpu_hold, ref_hold = SCENARIOS[name]()
    print(f"hold-out [{name}]: gpu hogwild {gpu_hold:.4f}, sequential oracle {ref_hold:.4f}")
    assert gpu_hold < 0.6931  # it learned something
    assert abs(gpu_hold - ref_hold) < SCENARIO_TOL.get(name, HOLDOUT_TOL), (name, gpu_hold, ref_hold)


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
