"""world_size 2 and 4 gloo tests of the multi-GPU steps' PROTOCOLS (fwumious_wabbit_amd/csrc/dist.cpp: owner-sharded step,
row-sparse gradient buckets), on CPU.

The library has no CPU compute path, so what runs here is a miniature of the step with the oracle's arithmetic in numpy (LR block
+ AdagradLUT, block_lr.rs:28-47 / 135-150, optimizer.rs:101-156) and the real collectives over torch.distributed: all-gather of
the examples, per-owner partial sums, reduce-scatter to the home ranks, sigmoid / gradient there, all-gather of the gradients,
owner-side updates in example order, gather of the owned ranges.  It pins the algebra the GPU path implements with RCCL:
N ranks == one learner running synchronous micro-batches of N*B examples, bit for bit for this model (every sum keeps its order)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fwo

BITS, B, STEPS, NNZ = 10, 16, 5, 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]



def _make_lut(lr=0.1, power_t=0.5, init_acc=1.0):  # optimizer.rs:121-144
    x = np.arange(2048, dtype=np.uint32)
    a = (x << 20).view(np.float32).astype(np.float32) + np.float32(init_acc)
    b = ((x + 1) << 20).astype(np.uint32).view(np.float32).astype(np.float32) + np.float32(init_acc)
    with np.errstate(all="ignore"):
        v = np.float32(lr) * (np.power(a, np.float32(-power_t)) + np.power(b, np.float32(-power_t))) * np.float32(0.5)
    v = v.astype(np.float32)
    v[~np.isfinite(v)] = np.float32(lr)
    return v


def _examples(step, n):
    rng = np.random.default_rng(1000 + step)
    idx = rng.integers(0, 1 << BITS, size=(n, NNZ)).astype(np.int64)
    idx[:, 0] = 7  # a hot entry every example touches (the constant feature)
    val = rng.choice([1.0, 0.5, 2.0], size=(n, NNZ)).astype(np.float32)
    y = (rng.random(n) < 0.4).astype(np.float32)
    return idx, val, y


def _sigmoid_grad(wsum, y):  # block_loss_functions.rs:105-153
    wsum = np.float32(wsum)
    p = np.float32(1.0) / (np.float32(1.0) + np.exp(-wsum, dtype=np.float32))
    return p, np.float32(-(y - p))


def _apply(w, acc, lut, idx, val, g, lo, hi):
    """owner-side updates of one example, entries in buffer order (block_lr.rs:135-150)"""
    for h, v in zip(idx, val):
        if lo <= h < hi:
            grad = np.float32(g * v)
            acc[h] = np.float32(acc[h] + grad * grad)
            key = int(np.float32(acc[h]).view(np.uint32)) >> 20
            w[h] = np.float32(w[h] - grad * lut[key])


def _single_learner(n_ranks):
    lut = _make_lut()
    w = np.zeros(1 << BITS, dtype=np.float32)
    acc = np.zeros(1 << BITS, dtype=np.float32)
    preds = []
    for s in range(STEPS):
        idx, val, y = _examples(s, n_ranks * B)
        gs = []
        for e in range(len(y)):  # all examples against the weights of the batch start
            wsum = np.float32(0.0)
            for h, v in zip(idx[e], val[e]):
                wsum = np.float32(wsum + w[h] * v)
            p, g = _sigmoid_grad(wsum, y[e])
            preds.append(p)
            gs.append(g)
        for e in range(len(y)):
            _apply(w, acc, lut, idx[e], val[e], gs[e], 0, 1 << BITS)
    return w, acc, np.array(preds, dtype=np.float32)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = _make_lut()
    per = (1 << BITS) // world
    lo, hi = per * rank, per * (rank + 1)
    w = np.zeros(1 << BITS, dtype=np.float32)   # full-size allocation, only [lo, hi) is kept current (as in dist.cpp)
    acc = np.zeros(1 << BITS, dtype=np.float32)
    preds = []
    for s in range(STEPS):
        idx_all, val_all, y_all = _examples(s, world * B)
        mine = slice(rank * B, (rank + 1) * B)
        # X1: all-gather of the examples (every rank brings its own B)
        gi = [torch.zeros(B, NNZ, dtype=torch.int64) for _ in range(world)]
        gv = [torch.zeros(B, NNZ, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(gi, torch.from_numpy(idx_all[mine].copy()))
        dist.all_gather(gv, torch.from_numpy(val_all[mine].copy()))
        idx = torch.cat(gi).numpy()
        val = torch.cat(gv).numpy()
        assert np.array_equal(idx, idx_all) and np.array_equal(val, val_all)
        # P1: partial sums over the owned entries, entries in buffer order.  One slot per entry keeps the order of the
        # final sum independent of N (the GPU path sums per owner first: 1e-7 relative, see the GPU tests)
        part = np.zeros((world * B, NNZ), dtype=np.float32)
        for e in range(world * B):
            for j, (h, v) in enumerate(zip(idx[e], val[e])):
                if lo <= h < hi:
                    part[e, j] = np.float32(w[h] * v)
        # X2: reduce-scatter (sum) to the home ranks
        recv = torch.zeros(B, NNZ, dtype=torch.float32)
        dist.reduce_scatter(recv, [torch.from_numpy(part[r * B:(r + 1) * B].copy()) for r in range(world)], op=dist.ReduceOp.SUM)
        # P2: logit, prediction, general gradient of the own examples
        g_own = np.zeros(B, dtype=np.float32)
        for e in range(B):
            wsum = np.float32(0.0)
            for j in range(NNZ):
                wsum = np.float32(wsum + recv[e, j].item())
            p, g_own[e] = _sigmoid_grad(wsum, y_all[rank * B + e])
            preds.append(p)
        # X3: all-gather of the gradients
        gg = [torch.zeros(B, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(gg, torch.from_numpy(g_own))
        g_all = torch.cat(gg).numpy()
        # P3: owner-side updates, example order
        for e in range(world * B):
            _apply(w, acc, lut, idx[e], val[e], g_all[e], lo, hi)
    # gather_tables: every rank's owned range into every rank's tables
    for tab in (w, acc):
        parts = [torch.zeros(per, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(tab[lo:hi].copy()))
        tab[:] = torch.cat(parts).numpy()
    out[rank] = (w.copy(), acc.copy(), np.array(preds, dtype=np.float32))
    dist.destroy_process_group()


def _run(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def test_sharded_protocol_world2_and_world4_equal_one_learner():
    for world in (2, 4):
        w_ref, acc_ref, p_ref = _single_learner(world)
        res = _run(world)
        for r, (w, acc, preds) in enumerate(res):
            assert np.array_equal(w, w_ref) and np.array_equal(acc, acc_ref), (world, r)
            # rank r predicted examples [r*B, (r+1)*B) of every step
            mine = np.concatenate([p_ref[s * world * B + r * B: s * world * B + (r + 1) * B] for s in range(STEPS)])
            assert np.array_equal(preds, mine), (world, r)
        assert np.count_nonzero(w_ref) > 100  # it did learn something


# ------------------------------------------------------------------ owner-side apply (dist.cpp fwgpu_dist_learn_owner)
def _owner_worker(rank, world, port, out):
    """the owner-side-apply step's protocol: FETCH the weights of the own examples' entries from their owners, score, PUSH one (hash, gradient)
    per occurrence to the entry's owner -- positions counted by the source, counts exchanged with the step's one collective -- and APPLY what
    arrived, source after source, in push order.  N ranks == one learner running synchronous steps of N*B examples, bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = _make_lut()
    per = (1 << BITS) // world
    lo, hi = per * rank, per * (rank + 1)
    w = np.zeros(1 << BITS, dtype=np.float32)
    acc = np.zeros(1 << BITS, dtype=np.float32)
    preds = []
    cap = B * NNZ
    for s in range(STEPS):
        idx_all, val_all, y_all = _examples(s, world * B)
        idx, val, y = idx_all[rank * B:(rank + 1) * B], val_all[rank * B:(rank + 1) * B], y_all[rank * B:(rank + 1) * B]
        # fetch: the owners' current ranges (the GPU path reads single rows over the link; what matters here is WHOSE copy is read)
        parts = [torch.zeros(per, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(w[lo:hi].copy()))
        w_seen = torch.cat(parts).numpy()
        # score the own examples, push one gradient per occurrence to its owner, in buffer order
        push_h = np.zeros((world, cap), dtype=np.int64)
        push_g = np.zeros((world, cap), dtype=np.float32)
        cnt = np.zeros(world, dtype=np.int64)
        for e in range(B):
            wsum = np.float32(0.0)
            for h, v in zip(idx[e], val[e]):
                wsum = np.float32(wsum + w_seen[h] * v)
            p, g = _sigmoid_grad(wsum, y[e])
            preds.append(p)
            for h, v in zip(idx[e], val[e]):
                o = int(h) // per
                push_h[o, cnt[o]] = h
                push_g[o, cnt[o]] = np.float32(g * v)
                cnt[o] += 1
        # the step's collective: counts and rings (the GPU path writes the rings straight into the owner's memory and gathers only the counts)
        all_cnt = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(all_cnt, torch.from_numpy(cnt.copy()))
        all_h = [torch.zeros(world, cap, dtype=torch.int64) for _ in range(world)]
        all_g = [torch.zeros(world, cap, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(all_h, torch.from_numpy(push_h.copy()))
        dist.all_gather(all_g, torch.from_numpy(push_g.copy()))
        # apply: source after source, push order
        for src in range(world):
            n = int(all_cnt[src][rank])
            hh, gg = all_h[src][rank].numpy(), all_g[src][rank].numpy()
            for j in range(n):
                h, grad = int(hh[j]), np.float32(gg[j])
                assert lo <= h < hi
                acc[h] = np.float32(acc[h] + grad * grad)
                w[h] = np.float32(w[h] - grad * lut[int(np.float32(acc[h]).view(np.uint32)) >> 20])
    for tab in (w, acc):
        parts = [torch.zeros(per, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(tab[lo:hi].copy()))
        tab[:] = torch.cat(parts).numpy()
    out[rank] = (w.copy(), acc.copy(), np.array(preds, dtype=np.float32))
    dist.destroy_process_group()


def test_owner_side_apply_protocol_world2_and_world4_equal_one_learner():
    for world in (2, 4):
        w_ref, acc_ref, p_ref = _single_learner(world)
        port = _free_port()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_owner_worker, args=(world, port, out), nprocs=world, join=True)
        for r in range(world):
            w, acc, preds = out[r]
            assert np.array_equal(w, w_ref) and np.array_equal(acc, acc_ref), (world, r)
            mine = np.concatenate([p_ref[s * world * B + r * B: s * world * B + (r + 1) * B] for s in range(STEPS)])
            assert np.array_equal(preds, mine), (world, r)


# ------------------------------------------------------------------ row-sparse gradient buckets (sparse.hip / dist.cpp)
def _buckets(idx, val, g):
    """one rank's micro-batch -> deduplicated (key, gradient) buckets: occurrences sorted by (key, example, entry), summed in
    order inside 64-element blocks of the sorted list (sparse_reduce_lr_kernel)"""
    occ = sorted((int(idx[e, j]), e, j) for e in range(idx.shape[0]) for j in range(idx.shape[1]))
    keys, vals = [], []
    for b0 in range(0, len(occ), 64):
        cur, acc = None, np.float32(0.0)
        for h, e, j in occ[b0:b0 + 64]:
            if h != cur:
                if cur is not None:
                    keys.append(cur)
                    vals.append(acc)
                cur, acc = h, np.float32(0.0)
            acc = np.float32(acc + np.float32(g[e] * val[e, j]))
        keys.append(cur)
        vals.append(acc)
    return np.array(keys, dtype=np.int64), np.array(vals, dtype=np.float32)


def _apply_buckets(w, acc, lut, bucket_lists):
    """all ranks' buckets, merged by (key, rank, index): one optimizer step per key with the summed gradient (sparse_apply_lr_kernel)"""
    merged = sorted((int(k), r, i, v) for r, (ks, vs) in enumerate(bucket_lists) for i, (k, v) in enumerate(zip(ks, vs)))
    i = 0
    while i < len(merged):
        h, G = merged[i][0], np.float32(0.0)
        while i < len(merged) and merged[i][0] == h:
            G = np.float32(G + merged[i][3])
            i += 1
        if G != 0.0:
            acc[h] = np.float32(acc[h] + G * G)
            key = int(np.float32(acc[h]).view(np.uint32)) >> 20
            w[h] = np.float32(w[h] - G * lut[key])


def _forward(w, idx, val, y):
    ps, gs = [], []
    for e in range(len(y)):
        wsum = np.float32(0.0)
        for h, v in zip(idx[e], val[e]):
            wsum = np.float32(wsum + w[h] * v)
        p, g = _sigmoid_grad(wsum, y[e])
        ps.append(p)
        gs.append(g)
    return np.array(ps, dtype=np.float32), np.array(gs, dtype=np.float32)


def _single_sparse_learner(n_ranks):
    lut = _make_lut()
    w = np.zeros(1 << BITS, dtype=np.float32)
    acc = np.zeros(1 << BITS, dtype=np.float32)
    preds = []
    for s in range(STEPS):
        idx, val, y = _examples(s, n_ranks * B)
        p, g = _forward(w, idx, val, y)
        preds.append(p)
        _apply_buckets(w, acc, lut, [_buckets(idx[r * B:(r + 1) * B], val[r * B:(r + 1) * B], g[r * B:(r + 1) * B]) for r in range(n_ranks)])
    return w, acc, np.concatenate(preds)


def _sparse_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = _make_lut()
    w = np.zeros(1 << BITS, dtype=np.float32)  # a full replica on every rank
    acc = np.zeros(1 << BITS, dtype=np.float32)
    preds = []
    for s in range(STEPS):
        idx_all, val_all, y_all = _examples(s, world * B)
        mine = slice(rank * B, (rank + 1) * B)
        p, g = _forward(w, idx_all[mine], val_all[mine], y_all[mine])  # own examples only: no record exchange in this mode
        preds.append(p)
        keys, vals = _buckets(idx_all[mine], val_all[mine], g)
        # X1: bucket counts, then the buckets padded to the largest (ncclAllGather needs equal counts)
        cnt = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(cnt, torch.tensor([len(keys)], dtype=torch.int64))
        stride = int(max(c.item() for c in cnt))
        pk, pv = np.zeros(stride, dtype=np.int64), np.zeros(stride, dtype=np.float32)
        pk[:len(keys)], pv[:len(keys)] = keys, vals
        gk = [torch.zeros(stride, dtype=torch.int64) for _ in range(world)]
        gv = [torch.zeros(stride, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(gk, torch.from_numpy(pk))
        dist.all_gather(gv, torch.from_numpy(pv))
        lists = [(gk[r].numpy()[:int(cnt[r].item())], gv[r].numpy()[:int(cnt[r].item())]) for r in range(world)]
        _apply_buckets(w, acc, lut, lists)  # every rank applies every rank's buckets, in the same order
    out[rank] = (w.copy(), acc.copy(), np.concatenate(preds))
    dist.destroy_process_group()


def test_sparse_bucket_protocol_world2_and_world4_equal_one_learner():
    """replicas + all-gathered deduplicated gradients == one learner taking one summed-gradient step per key and global batch,
    bit for bit, and the replicas never drift apart (no table exchange at all)"""
    for world in (2, 4):
        w_ref, acc_ref, p_ref = _single_sparse_learner(world)
        port = _free_port()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_sparse_worker, args=(world, port, out), nprocs=world, join=True)
        for r in range(world):
            w, acc, preds = out[r]
            assert np.array_equal(w, w_ref) and np.array_equal(acc, acc_ref), (world, r)
            mine = np.concatenate([p_ref[s * world * B + r * B: s * world * B + (r + 1) * B] for s in range(STEPS)])
            assert np.array_equal(preds, mine), (world, r)
        assert np.count_nonzero(w_ref) > 100


def test_dist_symbols_are_exported_without_loading_rccl():
    """the library resolves librccl only inside fwgpu_dist_init / fwgpu_dist_unique_id: importing it on a box without a GPU
    (this one) must not need RCCL, and the multi-GPU entry points must be there"""
    import ctypes as C

    from fwumious_wabbit_amd import _capi as capi
    L = capi.lib()
    for name in ("fwgpu_dist_unique_id", "fwgpu_dist_init", "fwgpu_dist_learn_sharded", "fwgpu_dist_gather_tables",
                 "fwgpu_dist_all_reduce_sum", "fwgpu_dist_group_create", "fwgpu_dist_group_learn_sharded",
                 "fwgpu_dist_learn_sparse", "fwgpu_dist_learn_sparse_batch", "fwgpu_dist_group_learn_sparse",
                 "fwgpu_learn_batch_sync", "fwgpu_split_create"):
        assert hasattr(L, name)
    with open("/proc/self/maps") as f:
        assert "librccl" not in f.read() or "torch" in open("/proc/self/maps").read()
    assert L.fwgpu_dist_init(None, None, 0, 1, C.byref(C.c_void_p())) != 0  # refused loudly, no crash
