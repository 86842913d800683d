"""Multi-GPU inside libfwgpu (include/fwgpu.h, fwgpu_dist_*): owner-sharded synchronous steps over RCCL, one process per GPU
(DistRank), or with every rank inside this process (DistGroup: tests, single-box emulation of an N-GPU job)."""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import check, ptr


def unique_id() -> bytes:
    """ncclGetUniqueId: call on ONE rank, hand the 128 bytes to all ranks (e.g. torch.distributed.broadcast_object_list)"""
    buf = (C.c_uint8 * 128)()
    check(capi.lib().fwgpu_dist_unique_id(buf, 128))
    return bytes(buf)


def _recs(records, rec_off):
    return np.ascontiguousarray(records, dtype=np.uint32), np.ascontiguousarray(rec_off, dtype=np.uint64)


class DistRank:
    """one rank of an RCCL job (fwgpu_dist_init); every rank builds the same model first"""

    def __init__(self, regressor, uid: bytes, rank: int, n_ranks: int):
        self.h = C.c_void_p()
        self.regressor = regressor
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(capi.lib().fwgpu_dist_init(regressor.h, buf, rank, n_ranks, C.byref(self.h)))
        self.rank, self.n = rank, n_ranks

    def set_mode(self, mode):
        check(capi.lib().fwgpu_dist_set_mode(self.h, mode))

    def comm_count(self) -> int:
        """ranks of the job as the communicator counts them (ncclCommCount)"""
        n = C.c_int32(0)
        check(capi.lib().fwgpu_dist_comm_count(self.h, C.byref(n)))
        return n.value

    def ranges(self):
        v = [C.c_uint32() for _ in range(4)]
        check(capi.lib().fwgpu_dist_ranges(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def learn_sharded(self, translator, records, rec_off) -> np.ndarray:
        records, rec_off = _recs(records, rec_off)
        n = len(rec_off) - 1
        out = np.zeros(n, dtype=np.float32)
        check(capi.lib().fwgpu_dist_learn_sharded(self.h, C.byref(translator.c), ptr(records), rec_off.ctypes.data_as(C.c_void_p), n, ptr(out)))
        return out

    def learn_sharded_batch(self, translator, batch):
        """the same step with this rank's records already in HBM (Regressor.record_batch); predictions: batch.predictions()"""
        check(capi.lib().fwgpu_dist_learn_sharded_batch(self.h, C.byref(translator.c), batch.h))

    def learn_sparse(self, translator, records, rec_off) -> np.ndarray:
        """row-sparse gradient buckets: this rank's micro-batch against its full replica, every rank applies all ranks' row
        gradients (one optimizer step per row and global batch)"""
        records, rec_off = _recs(records, rec_off)
        n = len(rec_off) - 1
        out = np.zeros(n, dtype=np.float32)
        check(capi.lib().fwgpu_dist_learn_sparse(self.h, C.byref(translator.c), ptr(records), rec_off.ctypes.data_as(C.c_void_p), n, ptr(out)))
        return out

    def learn_sparse_batch(self, translator, batch):
        check(capi.lib().fwgpu_dist_learn_sparse_batch(self.h, C.byref(translator.c), batch.h))

    def owner_stream_attach(self, log2_rows=0, log2_lr=0):
        """collective, once: tables + the streaming regions of every rank (fwgpu_dist_owner_stream_attach)"""
        check(capi.lib().fwgpu_dist_owner_stream_attach(self.h, int(log2_rows), int(log2_lr)))

    def learn_owner_stream(self, translator, records, rec_off, update=True, consumer_workgroups=0) -> np.ndarray:
        """one COLLECTIVE step of the streaming owner-side apply: this rank's records (may be none) -> their predictions"""
        records, rec_off = _recs(records, rec_off)
        n = len(rec_off) - 1
        out = np.zeros(max(n, 1), dtype=np.float32)
        check(capi.lib().fwgpu_dist_learn_owner_stream(self.h, C.byref(translator.c), ptr(records) if n else None, rec_off.ctypes.data_as(C.c_void_p), n, ptr(out),
                                                       1 if update else 0, int(consumer_workgroups)))
        return out[:n]

    def sparse_last_rows(self):
        """(FFM bucket rows, LR bucket entries) this rank sent in its last sparse step"""
        a, b = C.c_uint32(), C.c_uint32()
        check(capi.lib().fwgpu_dist_sparse_last_rows(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def peer_attach(self):
        """collective, once: map every other rank's tables (IPC handles through the job's all-gather)"""
        check(capi.lib().fwgpu_dist_peer_attach(self.h))

    def learn_peer(self, translator, records, rec_off, update=True) -> np.ndarray:
        """peer-sharded hogwild: this rank's records through the fused kernel, rows reached in their owners' tables (not a collective)"""
        records, rec_off = _recs(records, rec_off)
        n = len(rec_off) - 1
        out = np.zeros(max(n, 1), dtype=np.float32)
        check(capi.lib().fwgpu_dist_learn_peer(self.h, C.byref(translator.c), ptr(records), rec_off.ctypes.data_as(C.c_void_p), n, ptr(out), 1 if update else 0))
        return out[:n]

    def owner_attach(self, max_rows, max_lr):
        """owner-side apply: maps tables and gradient rings (sized for steps of up to max_rows gradient rows / max_lr LR gradients per rank); collective"""
        check(capi.lib().fwgpu_dist_owner_attach(self.h, int(max_rows), int(max_lr)))

    def learn_owner(self, translator, records, rec_off, update=True) -> np.ndarray:
        """one COLLECTIVE owner-side-apply step: this rank's records (possibly none) fetch their rows from the owners and push gradient rows to them;
        this rank then applies what the others pushed here"""
        records, rec_off = _recs(records, rec_off)
        n = len(rec_off) - 1
        out = np.zeros(max(n, 1), dtype=np.float32)
        check(capi.lib().fwgpu_dist_learn_owner(self.h, C.byref(translator.c), ptr(records) if n else None, rec_off.ctypes.data_as(C.c_void_p), n, ptr(out), 1 if update else 0))
        return out[:n]

    def learn_peer_batch(self, translator, batch, update=True, stream=None):
        """the same step with the records already in HBM (Regressor.record_batch); predictions: batch.predictions()"""
        check(capi.lib().fwgpu_dist_learn_peer_batch(self.h, C.byref(translator.c), batch.h, 1 if update else 0, stream))

    def barrier(self):
        check(capi.lib().fwgpu_dist_barrier(self.h))

    def gather_tables(self):
        check(capi.lib().fwgpu_dist_gather_tables(self.h))

    def all_reduce_sum(self, device_ptr: int, count: int, stream=None):
        check(capi.lib().fwgpu_dist_all_reduce_sum(self.h, C.c_void_p(device_ptr), count, stream))

    def close(self):
        if self.h:
            capi.lib().fwgpu_dist_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DistGroup:
    """all ranks of a job inside this process (fwgpu_dist_group_*): regressors[j] is rank j's model"""

    def __init__(self, regressors):
        self.regressors = list(regressors)
        self.n = len(self.regressors)
        arr = (C.c_void_p * self.n)(*[r.h for r in self.regressors])
        self.h = C.c_void_p()
        check(capi.lib().fwgpu_dist_group_create(arr, self.n, C.byref(self.h)))

    def set_mode(self, mode):
        check(capi.lib().fwgpu_dist_group_set_mode(self.h, mode))

    def learn_sharded(self, translator, records_per_rank, rec_off_per_rank):
        """one step: rank j brings records_per_rank[j] (the same number of records on every rank) -> its predictions"""
        rr = [_recs(a, b) for a, b in zip(records_per_rank, rec_off_per_rank)]
        n = len(rr[0][1]) - 1
        outs = [np.zeros(n, dtype=np.float32) for _ in rr]
        recp = (C.c_void_p * self.n)(*[a.ctypes.data for a, _ in rr])
        offp = (C.c_void_p * self.n)(*[b.ctypes.data for _, b in rr])
        outp = (C.c_void_p * self.n)(*[o.ctypes.data for o in outs])
        check(capi.lib().fwgpu_dist_group_learn_sharded(self.h, C.byref(translator.c), recp, offp, n, outp))
        return outs

    def learn_sparse(self, translator, records_per_rank, rec_off_per_rank):
        """one row-sparse step: rank j brings records_per_rank[j] (any number) -> its predictions"""
        rr = [_recs(a, b) for a, b in zip(records_per_rank, rec_off_per_rank)]
        ns = np.array([len(b) - 1 for _, b in rr], dtype=np.uint32)
        outs = [np.zeros(max(int(n), 1), dtype=np.float32) for n in ns]
        recp = (C.c_void_p * self.n)(*[a.ctypes.data for a, _ in rr])
        offp = (C.c_void_p * self.n)(*[b.ctypes.data for _, b in rr])
        outp = (C.c_void_p * self.n)(*[o.ctypes.data for o in outs])
        check(capi.lib().fwgpu_dist_group_learn_sparse(self.h, C.byref(translator.c), recp, offp, ns.ctypes.data_as(C.c_void_p), outp))
        return [o[:int(n)] for o, n in zip(outs, ns)]

    def learn_peer(self, translator, records_per_rank, rec_off_per_rank, update=True):
        """one peer-sharded hogwild step: rank j runs the fused kernel on records_per_rank[j] (any number), every row reached in its
        owner's tables (fwgpu_dist_group_learn_peer) -> its predictions"""
        rr = [_recs(a, b) for a, b in zip(records_per_rank, rec_off_per_rank)]
        ns = np.array([len(b) - 1 for _, b in rr], dtype=np.uint32)
        outs = [np.zeros(max(int(n), 1), dtype=np.float32) for n in ns]
        recp = (C.c_void_p * self.n)(*[a.ctypes.data for a, _ in rr])
        offp = (C.c_void_p * self.n)(*[b.ctypes.data for _, b in rr])
        outp = (C.c_void_p * self.n)(*[o.ctypes.data for o in outs])
        check(capi.lib().fwgpu_dist_group_learn_peer(self.h, C.byref(translator.c), recp, offp, ns.ctypes.data_as(C.c_void_p), outp, 1 if update else 0))
        return [o[:int(n)] for o, n in zip(outs, ns)]

    def learn_owner(self, translator, records_per_rank, rec_off_per_rank, update=True):
        """one owner-side-apply step: rank j's records fetch their rows from the owners and push one gradient row per occurrence to them, every
        owner applies what it received on its own tables (fwgpu_dist_group_learn_owner) -> predictions per rank"""
        rr = [_recs(a, b) for a, b in zip(records_per_rank, rec_off_per_rank)]
        ns = np.array([len(b) - 1 for _, b in rr], dtype=np.uint32)
        outs = [np.zeros(max(int(n), 1), dtype=np.float32) for n in ns]
        recp = (C.c_void_p * self.n)(*[a.ctypes.data for a, _ in rr])
        offp = (C.c_void_p * self.n)(*[b.ctypes.data for _, b in rr])
        outp = (C.c_void_p * self.n)(*[o.ctypes.data for o in outs])
        check(capi.lib().fwgpu_dist_group_learn_owner(self.h, C.byref(translator.c), recp, offp, ns.ctypes.data_as(C.c_void_p), outp, 1 if update else 0))
        return [o[:int(n)] for o, n in zip(outs, ns)]

    def learn_owner_stream(self, translator, records_per_rank=None, rec_off_per_rank=None, batches=None, update=True, log2_rows=0, log2_lr=0, consumer_workgroups=0):
        """one step of the STREAMING owner-side apply (fwgpu_dist_group_learn_owner_stream): the owners drain their regions while the sources' kernels
        fill them.  Either host records per rank (-> predictions per rank) or `batches`: one device-resident record batch per rank (predictions land there)."""
        if batches is not None:
            bp = (C.c_void_p * self.n)(*[b.h for b in batches])
            check(capi.lib().fwgpu_dist_group_learn_owner_stream(self.h, C.byref(translator.c), None, None, None, bp, None, 1 if update else 0, int(log2_rows), int(log2_lr),
                                                                 int(consumer_workgroups)))
            return None
        rr = [_recs(a, b) for a, b in zip(records_per_rank, rec_off_per_rank)]
        ns = np.array([len(b) - 1 for _, b in rr], dtype=np.uint32)
        outs = [np.zeros(max(int(n), 1), dtype=np.float32) for n in ns]
        recp = (C.c_void_p * self.n)(*[a.ctypes.data for a, _ in rr])
        offp = (C.c_void_p * self.n)(*[b.ctypes.data for _, b in rr])
        outp = (C.c_void_p * self.n)(*[o.ctypes.data for o in outs])
        check(capi.lib().fwgpu_dist_group_learn_owner_stream(self.h, C.byref(translator.c), recp, offp, ns.ctypes.data_as(C.c_void_p), None, outp, 1 if update else 0,
                                                             int(log2_rows), int(log2_lr), int(consumer_workgroups)))
        return [o[:int(n)] for o, n in zip(outs, ns)]

    def gather_tables(self):
        check(capi.lib().fwgpu_dist_group_gather_tables(self.h))

    def close(self):
        if self.h:
            capi.lib().fwgpu_dist_group_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
