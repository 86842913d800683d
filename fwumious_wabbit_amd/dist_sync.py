"""Data-parallel replica synchronisation over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

Every rank trains its own replica on its own shard of the stream.  A sync makes every replica apply what the OTHER
ranks changed since the last agreed snapshot `s0`:

    blocking :  d = t - s0 ; D = all_reduce(c*d) ; s0 <- s0 + D ; t <- s0
    overlapped: start():  d = t - s0 ; launch all_reduce(D <- c*d) asynchronously; training continues on t
                finish(): wait ; t <- t + (D - d) ; s0 <- s0 + D

c = 1/world_size (combine="mean", the default): the agreed model moves by the MEAN of the replicas' deltas (local SGD /
model averaging, weights and AdaGrad accumulators alike).  c = 1 (combine="sum") applies everyone's updates on top of
each other; measured on the BASELINE config-C stream (scripts/replica_sim.py, 48 steps, exchange every 32): hold-out log-loss
0.643 (1 replica) -> mean: 0.626 / 0.617 / 0.614 at 2 / 4 / 8 replicas; sum: 0.675 / 3.8 / 18.3 (diverged) -- every replica
steps with its own small accumulators, so N summed steps on a hot weight are N over-sized steps.

With `u` = the local updates made while the all-reduce was in flight, `t = s0 + d + u` before finish() and
`t = (s0 + D) + u` after it, so the next local delta is `(D - d) + u`... measured against the new snapshot it is exactly `u`:
nothing is lost or applied twice.  Replicas agree up to their not-yet-synchronised local updates -- the multi-GPU analogue of
hogwild.rs staleness.  The tensors are zero-copy views of the library's tables (Regressor.table_as_torch); torch is only plumbing.

xGMI note: a ring all-reduce of the 4.3 GB of tables is bound by one link per hop; large buckets (256 MiB) amortise
launch and ring latency, and the overlapped mode hides the transfer behind `sync_every` steps of training.
"""
import torch
import torch.distributed as dist


class DeltaAllReduce:
    def __init__(self, tables, bucket_elems=1 << 26, group=None, overlap=False, combine="mean", dist_rank=None):
        """tables: list of 1-D float32 tensors (views of the regressor's tables); bucket_elems: all-reduce bucket size in
        elements (2^26 floats = 256 MiB); overlap: keep a delta buffer per table and run the all-reduce asynchronously.
        dist_rank: a fwumious_wabbit_amd.dist.DistRank -- the all-reduce then goes through the library's own RCCL
        communicator (fwgpu_dist_all_reduce_sum, on a side stream) instead of torch.distributed."""
        self.tables = list(tables)
        self.dist_rank = dist_rank
        self._world = dist_rank.n if dist_rank is not None else None
        self._side = torch.cuda.Stream() if dist_rank is not None else None
        self.snapshots = [t.clone() for t in self.tables]
        self.bucket = int(bucket_elems)
        self.group = group
        self.overlap = bool(overlap)
        assert combine in ("mean", "sum")
        self.combine = combine
        self.n_syncs = 0
        self._local = [torch.empty_like(t) for t in self.tables] if overlap else None   # d (kept)
        self._summed = [torch.empty_like(t) for t in self.tables] if overlap else None  # D (all-reduced in place)
        self._works = None

    def _scale(self):
        world = self._world if self._world is not None else dist.get_world_size(self.group)
        return 1.0 / world if self.combine == "mean" else 1.0

    def _all_reduce(self, x, async_op=False):
        if self.dist_rank is not None and x.is_cuda:  # the library's RCCL communicator, on the side stream
            self.dist_rank.all_reduce_sum(x.data_ptr(), x.numel(), self._side.cuda_stream)
            return None
        return dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    # ---- blocking
    def sync(self):
        if self._works is not None:
            self.finish()
        for t, s0 in zip(self.tables, self.snapshots):
            for a in range(0, t.numel(), self.bucket):
                b = min(a + self.bucket, t.numel())
                d = t[a:b] - s0[a:b]
                if self._scale() != 1.0:
                    d *= self._scale()
                if self.dist_rank is not None and d.is_cuda:
                    self._side.wait_stream(torch.cuda.current_stream())
                    self._all_reduce(d)
                    torch.cuda.current_stream().wait_stream(self._side)
                else:
                    self._all_reduce(d)
                s0[a:b] += d
                t[a:b] = s0[a:b]
        self.n_syncs += 1

    # ---- overlapped
    def start(self):
        assert self.overlap, "construct with overlap=True"
        if self._works is not None:
            self.finish()
        works = []
        for t, s0, d, D in zip(self.tables, self.snapshots, self._local, self._summed):
            if t.is_cuda:  # one fused pass in the library instead of two torch passes
                from . import _capi as capi

                capi.check(capi.lib().fwgpu_delta_start(t.data_ptr(), s0.data_ptr(), d.data_ptr(), D.data_ptr(), t.numel(),
                                                        self._scale(), torch.cuda.current_stream(t.device).cuda_stream))
            else:
                torch.sub(t, s0, out=d)
                torch.mul(d, self._scale(), out=D)
            if self.dist_rank is not None and t.is_cuda:
                self._side.wait_stream(torch.cuda.current_stream(t.device))  # D is ready when the side stream starts
            for a in range(0, t.numel(), self.bucket):
                b = min(a + self.bucket, t.numel())
                w = self._all_reduce(D[a:b], async_op=True)
                if w is not None:
                    works.append(w)
        self._works = works

    def in_flight(self):
        return self._works is not None

    def finish(self):
        if self._works is None:
            return
        for w in self._works:
            w.wait()
        if self.dist_rank is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._works = None
        for t, s0, d, D in zip(self.tables, self.snapshots, self._local, self._summed):
            if t.is_cuda:
                from . import _capi as capi

                capi.check(capi.lib().fwgpu_delta_finish(t.data_ptr(), s0.data_ptr(), d.data_ptr(), D.data_ptr(), t.numel(),
                                                         torch.cuda.current_stream(t.device).cuda_stream))
            else:
                s0 += D          # new agreed snapshot
                D -= d           # what the other ranks did
                t += D           # local table keeps its own newer updates
        self.n_syncs += 1

    def step(self):
        """One call per sync point: land the previous exchange (if any) and start the next one."""
        if self.overlap:
            self.finish()
            self.start()
        else:
            self.sync()

    def bytes_per_sync(self):
        return sum(4 * t.numel() for t in self.tables)
