"""The library's PROCESS-PER-RANK multi-GPU path (fwgpu_dist_init -> fwgpu_dist_learn_sharded / fwgpu_dist_learn_sparse /
fwgpu_dist_gather_tables / fwgpu_dist_all_reduce_sum, dist.cpp) with N = 2 and 4 ranks, each its own process, on ONE GPU: the
seven RCCL entry points come from the shared-memory stand-in of tests/fake_rccl (FWGPU_RCCL_LIBRARY), so the code that only
N > 1 exercises -- shape / count exchange, padding to the largest rank, owner ranges, in-place table gathers, the straddling
tail -- runs against the oracle (fwo_learn_minibatch / fwo_learn_sparse) without a multi-GPU node.  What this does NOT test is
RCCL itself or xGMI."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from helpers import logloss, make_pair, record_labels
from oracle import fwo

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfwgpu_fakerccl.so")
WORKER = os.path.join(ROOT, "tests", "dist_rank_worker.py")


@pytest.fixture(scope="module", autouse=True)
def _fake_rccl_built():
    # normally built by __graft_entry__.build(); a checkout without it builds it here (hipcc is on every box of this image)
    if not os.path.exists(FAKE):
        subprocess.run(["make", "-C", os.path.dirname(FAKE)], check=True)
    assert os.path.exists(FAKE)


def _run_job(tmp_path, mode, n_ranks, cfg, recs, off, parts, allreduce=0, **extra):
    n_ns, k, bits, ffm_bits, opt, lr = cfg
    job = str(tmp_path / f"job_{mode}_{n_ranks}.npz")
    np.savez(job, n_ranks=n_ranks, mode=mode, n_ns=n_ns, k=k, bits=bits, ffm_bits=ffm_bits, optimizer=int(opt), lr=lr, recs=recs, off=off,
             parts=np.asarray(parts, dtype=np.int64), id_file=str(tmp_path / f"id_{mode}_{n_ranks}"), allreduce=allreduce, **extra)
    env = dict(os.environ, FWGPU_RCCL_LIBRARY=FAKE, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, job, str(r), str(tmp_path / f"out_{mode}_{n_ranks}_{r}.npz")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n_ranks)]
    outs = []
    for r, p in enumerate(procs):
        try:
            log, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {r} failed:\n{log[-3000:]}"
    for r in range(n_ranks):
        outs.append(np.load(str(tmp_path / f"out_{mode}_{n_ranks}_{r}.npz")))
    return outs


def _close(a, b):
    # (AdagradLUT bucket edges: see tests/test_gpu_dist.py)
    bad = np.abs(a - b) > 3e-5 + 1e-5 * np.abs(b)
    return int(bad.sum()) <= max(3, a.size // 10000) and float(np.abs(a - b).max()) < 5e-3


def _check_failed_steps(outs, n_steps):
    """the deliberately failed steps in front of a `*_fail` job (tests/dist_rank_worker.py): every rank came back from every one of them --
    the culprit with its own error, the others with FWGPU_ERR_PEER -- and, as the caller's oracle comparison then shows, nothing was applied"""
    PEER = 8
    n = len(outs)
    codes = np.stack([o["codes"] for o in outs])  # [rank, failed step]
    assert codes.shape == (n, n_steps), codes
    assert codes[n - 1, 0] not in (0, PEER) and np.all(codes[:n - 1, 0] == PEER), codes  # step A: the last rank's malformed record
    if n_steps > 1:
        assert np.all(codes[:, 1] == PEER), codes                                        # step B: micro-batch sizes differ: nobody is "the" culprit


@pytest.mark.parametrize("n_ranks,fail", [(2, False), (4, False), (2, True)])
def test_process_per_rank_sharded_step_matches_the_oracle(tmp_path, n_ranks, fail):
    assert os.path.exists(FAKE), "tests/fake_rccl is not built (__graft_entry__.build)"
    n_ns, k, bits, ffm_bits = 10, 4, 14, 14
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    steps, gb = 4, 96
    R = n_ns * k
    recs0, off0 = fw.synth_records(n_ns, 1.0, 1.1, 3000, 0.1, 181, 0, 3 * steps * gb)
    fbt0 = fw.FeatureBufferTranslator(mi)
    # examples without rows that straddle an ownership boundary (the documented deviation, measured in test_gpu_dist.py)
    bounds = [j * (1 << ffm_bits) // 4 for j in range(1, 4)]
    keep = []
    for i in range(len(off0) - 1):
        h = np.asarray(fbt0.translate(recs0[int(off0[i]):int(off0[i + 1])]).ffm_buffer)["hash"].astype(np.int64)
        if not any(((h < b) & (h + R > b)).any() for b in bounds):
            keep.append(i)
        if len(keep) == steps * gb:
            break
    recs = np.concatenate([recs0[int(off0[i]):int(off0[i + 1])] for i in keep])
    off = np.concatenate([[0], np.cumsum([int(off0[i + 1] - off0[i]) for i in keep])]).astype(np.uint64)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg)
    p_ref = np.concatenate([om.learn_minibatch(ots, recs[int(off[s * gb]):int(off[(s + 1) * gb])], off[s * gb:(s + 1) * gb + 1] - off[s * gb])
                            for s in range(steps)])
    ref_tabs = [np.asarray(om.lr_table), np.asarray(om.ffm_weights), np.asarray(om.ffm_acc)]
    parts = [[gb // n_ranks] * n_ranks for _ in range(steps)]
    outs = _run_job(tmp_path, "sharded_fail" if fail else "sharded", n_ranks, (n_ns, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, 0.05), recs, off, parts, allreduce=1)
    if fail:
        _check_failed_steps(outs, 2)
    per = gb // n_ranks
    preds = np.zeros(steps * gb, dtype=np.float32)
    for r, o in enumerate(outs):
        for s in range(steps):
            preds[s * gb + r * per: s * gb + (r + 1) * per] = o["preds"][s * per:(s + 1) * per]
    assert np.abs(logloss(preds, y) - logloss(p_ref, y)).max() < 1e-4
    for r, o in enumerate(outs):  # after gather_tables every rank holds the whole model
        for name, ref in zip(("lr", "ffm_w", "ffm_acc"), ref_tabs):
            assert _close(o[name], ref[:len(o[name])]), (r, name, float(np.abs(o[name] - ref[:len(o[name])]).max()))
            assert np.array_equal(o[name], outs[0][name]), (r, name)
        # owner ranges partition the tables
        lo, hi = int(o["ranges"][0]), int(o["ranges"][1])
        assert lo == r * ((1 << ffm_bits) // n_ranks) and (hi == 0xFFFFFFFF if r == n_ranks - 1 else hi == (r + 1) * ((1 << ffm_bits) // n_ranks))
        # all-reduce (sum) through the same communicator: every rank held the same accumulator table
        assert np.array_equal(o["allreduce"], o["ffm_acc"] * np.float32(n_ranks))


@pytest.mark.parametrize("n_ranks,opt,fail", [(2, fw.Optimizer.AdagradLUT, False), (4, fw.Optimizer.AdagradLUT, False), (3, fw.Optimizer.AdagradFlex, False),
                                              (3, fw.Optimizer.AdagradLUT, True)])
def test_process_per_rank_sparse_step_matches_the_oracle(tmp_path, n_ranks, opt, fail):
    assert os.path.exists(FAKE), "tests/fake_rccl is not built (__graft_entry__.build)"
    n_ns, k, bits, ffm_bits = 12, 4, 15, 15
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, opt, lr=0.05, ffm_lr=0.05)
    steps, gb = 4, 200
    recs, off = fw.synth_records(n_ns, 1.0, 1.1, 4000, 0.1, 183, 0, steps * gb)
    y = record_labels(recs, off)
    # uneven micro-batches (the all-gather pads to the largest rank), one rank EMPTY in one step
    base = {2: [120, 80], 3: [90, 70, 40], 4: [70, 60, 50, 20]}[n_ranks]
    parts = [list(base) for _ in range(steps)]
    parts[2] = [0] + [base[0] + base[1]] + base[2:]
    om = fwo.Model(ocfg)
    p_ref = np.concatenate([om.learn_sparse(ots, recs[int(off[s * gb]):int(off[(s + 1) * gb])], off[s * gb:(s + 1) * gb + 1] - off[s * gb],
                                            np.cumsum(parts[s])) for s in range(steps)])
    ref_tabs = [np.asarray(om.lr_table), np.asarray(om.ffm_weights), np.asarray(om.ffm_acc)]
    outs = _run_job(tmp_path, "sparse_fail" if fail else "sparse", n_ranks, (n_ns, k, bits, ffm_bits, opt, 0.05), recs, off, parts)
    if fail:
        _check_failed_steps(outs, 1)
    preds = np.zeros(steps * gb, dtype=np.float32)
    taken = [0] * n_ranks
    for s in range(steps):
        a = s * gb
        for r in range(n_ranks):
            preds[a:a + parts[s][r]] = outs[r]["preds"][taken[r]:taken[r] + parts[s][r]]
            taken[r] += parts[s][r]
            a += parts[s][r]
    assert np.abs(logloss(preds, y) - logloss(p_ref, y)).max() < 1e-5
    for r, o in enumerate(outs):
        for name, ref in zip(("lr", "ffm_w", "ffm_acc"), ref_tabs):
            assert _close(o[name], ref[:len(o[name])]), (r, name, float(np.abs(o[name] - ref[:len(o[name])]).max()))
            assert np.array_equal(o[name], outs[0][name]), (r, name)  # replicas stay bit-identical


def _no_straddle_stream(mi, n_ns, k, ffm_bits, n, seed):
    """examples without rows that straddle an ownership boundary of 4 ranks (the sharded modes' documented deviation)"""
    R = n_ns * k
    recs0, off0 = fw.synth_records(n_ns, 1.0, 1.1, 3000, 0.1, seed, 0, 3 * n)
    fbt0 = fw.FeatureBufferTranslator(mi)
    bounds = [j * (1 << ffm_bits) // 4 for j in range(1, 4)]
    keep = []
    for i in range(len(off0) - 1):
        h = np.asarray(fbt0.translate(recs0[int(off0[i]):int(off0[i + 1])]).ffm_buffer)["hash"].astype(np.int64)
        if not any(((h < b) & (h + R > b)).any() for b in bounds):
            keep.append(i)
        if len(keep) == n:
            break
    recs = np.concatenate([recs0[int(off0[i]):int(off0[i + 1])] for i in keep])
    off = np.concatenate([[0], np.cumsum([int(off0[i + 1] - off0[i]) for i in keep])]).astype(np.uint64)
    return recs, off


@pytest.mark.parametrize("mode,n_ranks", [("peer_seq", 2), ("peer_seq", 4), ("owner_seq", 2), ("owner_seq", 4)])
def test_process_per_rank_peer_sharded_in_order_is_the_sequential_reference(tmp_path, mode, n_ranks):
    """fwgpu_dist_peer_attach (IPC handles of every rank's tables through the job's all-gather, hipIpcOpenMemHandle) +
    fwgpu_dist_learn_peer with the ranks taking turns (fwgpu_dist_barrier) and each in example order: the job is the sequential
    reference algorithm over the ranks' micro-batches in rank order -- per-example parity with the oracle and the gathered tables,
    with every rank a PROCESS of its own that reaches the other processes' tables through mapped memory.
    `owner_seq`: the same job with OWNER-SIDE APPLY (fwgpu_dist_owner_attach + fwgpu_dist_learn_owner, one example per collective step): weight
    rows fetched from the owner process's tables, gradient rows pushed into a ring in the owner process's memory, the owner applying them."""
    n_ns, k, bits, ffm_bits = 10, 4, 14, 14
    mi, ocfg, ots = make_pair(n_ns, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    steps = 3
    parts = [{2: [70, 50], 4: [40, 30, 30, 20]}[n_ranks] for _ in range(steps)]
    recs, off = _no_straddle_stream(mi, n_ns, k, ffm_bits, steps * 120, 191)
    om = fwo.Model(ocfg)
    _, p_ref = om.run_stream(ots, recs, off, holdout_after=0, nthreads=1)
    outs = _run_job(tmp_path, mode, n_ranks, (n_ns, k, bits, ffm_bits, fw.Optimizer.AdagradLUT, 0.05), recs, off, parts)
    preds = np.zeros(steps * 120, dtype=np.float32)
    taken = [0] * n_ranks
    for s in range(steps):
        a = s * 120
        for r in range(n_ranks):
            preds[a:a + parts[s][r]] = outs[r]["preds"][taken[r]:taken[r] + parts[s][r]]
            taken[r] += parts[s][r]
            a += parts[s][r]
    assert np.abs(preds - p_ref).max() < 1e-5
    ref_tabs = [np.asarray(om.lr_table), np.asarray(om.ffm_weights), np.asarray(om.ffm_acc)]
    for r, o in enumerate(outs):
        for name, ref in zip(("lr", "ffm_w", "ffm_acc"), ref_tabs):
            assert _close(o[name], ref[:len(o[name])]), (r, name)
            assert np.array_equal(o[name], outs[0][name]), (r, name)


@pytest.mark.statistical
def test_process_per_rank_peer_sharded_hogwild_reaches_the_oracles_holdout_loss(tmp_path):
    """the concurrent form: two processes, each running the fused hogwild kernel on its half of every step at its own pace, rows
    reached in their owners' (the other process's) tables; the gathered model's hold-out loss against the sequential oracle's"""
    n_train, n_hold = 24000, 4000
    mi, ocfg, ots = make_pair(10, 4, 18, 18, fw.Optimizer.AdagradLUT)
    recs, off = fw.synth_records(10, 0.0, 1.1, 100000, 0.0, 20240611, 0, n_train + n_hold)
    y = record_labels(recs, off)
    om = fwo.Model(ocfg)
    _, p = om.run_stream(ots, recs, off, holdout_after=n_train + 1, nthreads=1)
    ref_hold = float(logloss(p[n_train:], y[n_train:]).mean())
    step = 2000
    parts = [[step // 2, step // 2] for _ in range(n_train // step)]
    outs = _run_job(tmp_path, "peer", 2, (10, 4, 18, 18, fw.Optimizer.AdagradLUT, 0.1), recs[: int(off[n_train])], off[: n_train + 1], parts)
    assert np.array_equal(outs[0]["ffm_w"], outs[1]["ffm_w"])  # gather_tables: both processes hold the whole model
    re = fw.Regressor(mi)
    from fwumious_wabbit_amd import _capi as capi
    re.table_write(capi.TABLE_LR, outs[0]["lr"])
    re.table_write(capi.TABLE_FFM_W, outs[0]["ffm_w"])
    re.table_write(capi.TABLE_FFM_ACC, outs[0]["ffm_acc"])
    fbt = fw.FeatureBufferTranslator(mi)
    hb = re.record_batch(fbt, recs[int(off[n_train]):], off[n_train:] - off[n_train])
    re.learn_batch(hb, capi.MODE_HOGWILD, False)
    gpu_hold = float(logloss(hb.predictions(), y[n_train:]).mean())
    hb.close()
    re.close()
    print(f"process-per-rank peer-sharded hogwild: hold-out {gpu_hold:.4f}, sequential oracle {ref_hold:.4f}")
    assert gpu_hold < 0.6931 and abs(gpu_hold - ref_hold) < 0.02, (gpu_hold, ref_hold)


def test_a_rank_that_is_gone_ends_the_others_step_through_the_timeout(tmp_path):
    """dist.cpp wait_stream()'s FWGPU_DIST_TIMEOUT_MS path -- read-backs behind a collective go through pinned memory owned by the rank, the
    caller's prediction buffer is filled after the polled wait, a timed-out rank is left without a communicator.  Three ranks; the last one exits after
    the first step; the two others must come back from their next step with FWGPU_ERR_PEER within seconds and be refused (FWGPU_ERR_INVALID) afterwards."""
    n_ns, k, bits, ffm_bits = 6, 4, 12, 12
    n_ranks = 3
    recs, off = fw.synth_records(n_ns, 1.0, 1.1, 500, 0.1, 77, 0, 96)
    job = str(tmp_path / "job_timeout.npz")
    np.savez(job, n_ranks=n_ranks, mode="sparse_timeout", n_ns=n_ns, k=k, bits=bits, ffm_bits=ffm_bits, optimizer=int(fw.Optimizer.AdagradLUT), lr=0.05, recs=recs,
             off=off, parts=np.asarray([[32, 32, 32]], dtype=np.int64), id_file=str(tmp_path / "id_timeout"), allreduce=0)
    env = dict(os.environ, FWGPU_RCCL_LIBRARY=FAKE, HSA_ENABLE_IPC_MODE_LEGACY="0", FWGPU_DIST_TIMEOUT_MS="1500", FWGPU_FAKERCCL_ASYNC="1")
    procs = [subprocess.Popen([sys.executable, WORKER, job, str(r), str(tmp_path / f"out_timeout_{r}.npz")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(n_ranks)]
    for r, p in enumerate(procs):
        try:
            log, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {r} failed:\n{log[-3000:]}"
    for r in range(n_ranks - 1):
        o = np.load(str(tmp_path / f"out_timeout_{r}.npz"))
        assert list(o["codes"]) == [8, 1], (r, o["codes"])  # FWGPU_ERR_PEER, then FWGPU_ERR_INVALID (no communicator left)
        assert float(o["seconds"]) < 60.0, float(o["seconds"])


@pytest.mark.timeout(300)
def test_streaming_owner_apply_with_a_rank_that_is_gone_gives_the_step_up(tmp_path):
    """the streaming step waits INSIDE the kernel for its peers -- a peer that is gone would hold the GPU for ever.  Two ranks; the second exits after the first
    step; the first one's next step must come back with FWGPU_ERR_PEER within seconds of FWGPU_DIST_TIMEOUT_MS (the host sets the abort word every wait loop of the kernel
    looks at) and its streaming state is void afterwards (FWGPU_ERR_PEER again, at once)."""
    n_ns, k, bits, ffm_bits = 6, 4, 16, 16
    n_ranks = 2
    recs, off = fw.synth_records(n_ns, 0.5, 0.0, 10_000_000, 0.3, 79, 0, 800)
    job = str(tmp_path / "job_stream_timeout.npz")
    np.savez(job, n_ranks=n_ranks, mode="owner_stream_timeout", n_ns=n_ns, k=k, bits=bits, ffm_bits=ffm_bits, optimizer=int(fw.Optimizer.SGD), lr=0.01, recs=recs,
             off=off, parts=np.asarray([[400, 400]], dtype=np.int64), id_file=str(tmp_path / "id_stream_timeout"), allreduce=0, no_constant=1, log2_rows=7, log2_lr=7)
    env = dict(os.environ, FWGPU_RCCL_LIBRARY=FAKE, HSA_ENABLE_IPC_MODE_LEGACY="0", FWGPU_DIST_TIMEOUT_MS="3000")
    procs = [subprocess.Popen([sys.executable, WORKER, job, str(r), str(tmp_path / f"out_stream_timeout_{r}.npz")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(n_ranks)]
    for r, p in enumerate(procs):
        try:
            log, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {r} failed:\n{log[-3000:]}"
    o = np.load(str(tmp_path / "out_stream_timeout_0.npz"))
    assert list(o["codes"]) == [8, 8], o["codes"]  # FWGPU_ERR_PEER: the step was given up; then: the streaming state is void
    assert 2.0 < float(o["seconds"]) < 60.0, float(o["seconds"])


@pytest.mark.timeout(420)
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_process_per_rank_streaming_owner_apply_delivers_every_gradient(tmp_path, n_ranks):
    """The STREAMING owner-side apply with one process per rank (fwgpu_dist_owner_stream_attach / fwgpu_dist_learn_owner_stream): the regions and the
    flow-control words of every rank reached through IPC mappings, circular regions of 128 slots for ~4000 gradient rows per source and step, two steps, NO
    collective inside a step.  SGD steps add up, no constant feature, ids drawn uniformly from 10^7: every LR entry must hold -lr * sum g v with g = p - y from the
    ranks' own predictions (block_lr.rs:143-147) -- exactly, where one example holds the entry -- and every rank must have gathered the same model."""
    n_ns, k, bits, ffm_bits, lr = 6, 4, 20, 20, 0.01
    cfg = (n_ns, k, bits, ffm_bits, fw.Optimizer.SGD, lr)
    per, steps = 1200 // n_ranks, 2
    recs, off = fw.synth_records(n_ns, 0.5, 0.0, 10_000_000, 0.3, 78, 0, steps * per * n_ranks)
    parts = [[per] * n_ranks for _ in range(steps)]
    a = _run_job(tmp_path, "owner_stream", n_ranks, cfg, recs, off, parts, no_constant=1, log2_rows=7, log2_lr=7)
    for r in range(1, n_ranks):  # every rank gathered the same model
        assert np.array_equal(a[r]["ffm_w"], a[0]["ffm_w"]) and np.array_equal(a[r]["lr"], a[0]["lr"])
    # predictions back in stream order: the worker of rank r holds [step 0 share, step 1 share]
    p = np.zeros(steps * per * n_ranks)
    for r in range(n_ranks):
        pr = a[r]["preds"]
        for s_ in range(steps):
            p[(s_ * n_ranks + r) * per:(s_ * n_ranks + r + 1) * per] = pr[s_ * per:(s_ + 1) * per]
    mi, _, _ = make_pair(n_ns, k, bits, ffm_bits, fw.Optimizer.SGD, lr=lr, ffm_lr=lr)
    mi.add_constant_feature = False
    fbt = fw.FeatureBufferTranslator(mi)
    y = recs[off[:-1].astype(np.int64) + 1].astype(np.float64)
    want, hits = np.zeros(1 << bits), np.zeros(1 << bits, dtype=np.int64)
    for e in range(len(y)):
        lrb = np.asarray(fbt.translate(recs[int(off[e]):int(off[e + 1])]).lr_buffer)
        np.add.at(want, lrb["hash"].astype(np.int64), -lr * (p[e] - y[e]) * lrb["value"].astype(np.float64))
        np.add.at(hits, lrb["hash"].astype(np.int64), 1)
    got = a[0]["lr"]
    got = (got[0::2] if got.size == 2 << bits else got).astype(np.float64)
    bad = np.abs(got - want) > 2e-6 + 2e-4 * np.abs(want)
    assert not np.any(bad & (hits <= 1)), ("an LR entry one example holds is not -lr * g * v", int((bad & (hits <= 1)).sum()), float(np.abs(got - want).max()))
    assert int(bad.sum()) <= int(np.count_nonzero(hits > 1))
    mi2, _, _ = make_pair(n_ns, k, bits, ffm_bits, fw.Optimizer.SGD, lr=lr, ffm_lr=lr)
    mi2.add_constant_feature = False
    init = fw.Regressor(mi2)
    w0 = init.table_read(fw.capi.TABLE_FFM_W)
    init.close()
    assert np.count_nonzero(a[0]["ffm_w"] != w0[:a[0]["ffm_w"].size]) > 20 * len(y)  # the rows did move
