"""Data-parallel replica synchronisation over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

Every rank trains its own replica on its own shard of the stream.  `DeltaAllReduce.sync()` makes all replicas
identical again by summing what each one changed since the last sync:

    table <- snapshot + sum_over_ranks(table_r - snapshot)

i.e. every replica ends up having applied every rank's updates (weights AND AdaGrad accumulators), with a staleness
of `steps_between_syncs x batch x world_size` examples -- the multi-GPU analogue of hogwild.rs.  The tensors are
zero-copy views of the library's tables (Regressor.table_as_torch); torch is only the plumbing here.
"""
import torch
import torch.distributed as dist


class DeltaAllReduce:
    def __init__(self, tables, bucket_elems=1 << 26, group=None):
        """tables: list of 1-D float32 tensors (views of the regressor's tables); bucket_elems: all-reduce bucket
        size in elements (2^26 floats = 256 MiB: large buckets amortise RCCL launch/ring latency over xGMI)."""
        self.tables = list(tables)
        self.snapshots = [t.clone() for t in self.tables]
        self.bucket = int(bucket_elems)
        self.group = group
        self.n_syncs = 0

    def sync(self):
        for t, s0 in zip(self.tables, self.snapshots):
            for a in range(0, t.numel(), self.bucket):
                b = min(a + self.bucket, t.numel())
                d = t[a:b] - s0[a:b]
                dist.all_reduce(d, op=dist.ReduceOp.SUM, group=self.group)
                s0[a:b] += d
                t[a:b] = s0[a:b]
        self.n_syncs += 1

    def bytes_per_sync(self):
        return sum(4 * t.numel() for t in self.tables)
