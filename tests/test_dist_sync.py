"""world_size-2 gloo test of the replica delta all-reduce (CPU tensors; the GPU path uses the same code over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fwumious_wabbit_amd.dist_sync import DeltaAllReduce


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out, combine="sum"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1234)
    base_w = torch.randn(1000, generator=g)          # identical initial tables on every rank
    base_a = torch.rand(777, generator=g)
    w, a = base_w.clone(), base_a.clone()
    sync = DeltaAllReduce([w, a], bucket_elems=256, combine=combine)  # several buckets, last one ragged
    c = 1.0 / world if combine == "mean" else 1.0
    expect_w, expect_a = base_w.clone(), base_a.clone()
    for step in range(3):
        # each rank applies its own sparse "training" updates
        gr = torch.Generator().manual_seed(100 * step + rank)
        idx = torch.randint(0, 1000, (50,), generator=gr)
        w[idx] -= 0.01 * (rank + 1)
        a[idx % 777] += 0.5
        # what the agreed model is after the exchange: everyone's updates on top of each other (sum) or their mean
        for r in range(world):
            g2 = torch.Generator().manual_seed(100 * step + r)
            i2 = torch.randint(0, 1000, (50,), generator=g2)
            dw = torch.zeros(1000)
            dw[i2] -= 0.01 * (r + 1)          # same semantics as the in-place op above (last write per index)
            da = torch.zeros(777)
            da[i2 % 777] += 0.5
            expect_w += c * dw
            expect_a += c * da
        sync.sync()
        assert torch.allclose(w, expect_w, atol=1e-6) and torch.allclose(a, expect_a, atol=1e-6)
        assert torch.equal(w, sync.snapshots[0]) and torch.equal(a, sync.snapshots[1])
    # replicas are bit-identical after a sync
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    assert all(torch.equal(gathered[0], x) for x in gathered)
    out[rank] = float(w.sum())
    dist.destroy_process_group()


def _worker_overlap(rank, world, port, out, combine="sum"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base = torch.linspace(-1, 1, 900)
    t = base.clone()
    sync = DeltaAllReduce([t], bucket_elems=256, overlap=True, combine=combine)
    # the bookkeeping invariant: table == agreed snapshot + (all landed deltas of others already inside it) + local tail
    total = base.clone()          # what a single sequential learner applying everyone's updates would hold
    local_unsynced = torch.zeros(900)
    landed_own = torch.zeros(900)
    for step in range(4):
        upd = torch.zeros(900)
        upd[(step * 37 + rank * 101) % 900] = 0.25 * (rank + 1)
        upd[(step * 11 + 5) % 900] += -0.5            # an index BOTH ranks touch
        t += upd                                      # "training" between sync points
        sync.step()                                   # lands the previous exchange, starts the next one
        more = torch.zeros(900)
        more[(step * 7 + rank) % 900] = 0.125         # updates made while the all-reduce is in flight
        t += more
        for r in range(world):
            u2 = torch.zeros(900)
            u2[(step * 37 + r * 101) % 900] = 0.25 * (r + 1)
            u2[(step * 11 + 5) % 900] += -0.5
            m2 = torch.zeros(900)
            m2[(step * 7 + r) % 900] = 0.125
            total += u2 + m2
    sync.finish()      # land the last exchange
    sync.step()        # exchange the remaining local tails ...
    sync.finish()      # ... and land them: now every replica holds everything
    if combine == "mean":  # every update ends up in exactly one exchange and enters the agreed model with weight 1/world
        total = base + (total - base) / world
    assert torch.allclose(t, total, atol=1e-5), float((t - total).abs().max())
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    assert all(torch.allclose(gathered[0], x, atol=1e-6) for x in gathered)
    assert torch.allclose(t, sync.snapshots[0], atol=1e-6)
    out[rank] = sync.n_syncs
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("combine", ["mean", "sum"])
def test_overlapped_delta_allreduce_world2_gloo(combine):
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker_overlap, args=(world, port, out, combine), nprocs=world, join=True)
        assert out[0] == out[1] == 5


@pytest.mark.parametrize("combine", ["mean", "sum"])
def test_delta_allreduce_world2_gloo(combine):
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out, combine), nprocs=world, join=True)
        assert len(out) == world and abs(out[0] - out[1]) == 0.0


def test_delta_allreduce_single_rank_is_identity():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        t = torch.arange(10, dtype=torch.float32)
        s = DeltaAllReduce([t], bucket_elems=4)
        t += 1.5
        want = t.clone()
        s.sync()
        assert torch.equal(t, want) and s.n_syncs == 1 and s.bytes_per_sync() == 40
    finally:
        dist.destroy_process_group()
