"""One RANK of a process-per-rank job of the library's multi-GPU path (fwgpu_dist_init + fwgpu_dist_learn_sharded / _sparse,
dist.cpp), started N times by tests/test_gpu_dist_procs.py on ONE GPU with FWGPU_RCCL_LIBRARY pointing at the shared-memory
stand-in of tests/fake_rccl.  Reads the job description (an .npz written by the test), runs its share, writes its predictions
and final tables.
usage: python dist_rank_worker.py <job.npz> <rank> <out.npz>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fwumious_wabbit_amd as fw  # noqa: E402
from fwumious_wabbit_amd import _capi as capi  # noqa: E402
from fwumious_wabbit_amd.dist import DistRank, unique_id  # noqa: E402
from helpers import make_pair  # noqa: E402


def main():
    job = np.load(sys.argv[1], allow_pickle=False)
    rank = int(sys.argv[2])
    n_ranks = int(job["n_ranks"])
    mode = str(job["mode"])
    n_ns, k, bits, ffm_bits, opt = (int(job[x]) for x in ("n_ns", "k", "bits", "ffm_bits", "optimizer"))
    mi, _, _ = make_pair(n_ns, k, bits, ffm_bits, opt, lr=float(job["lr"]), ffm_lr=float(job["lr"]))
    if "no_constant" in job.files and int(job["no_constant"]):
        mi.add_constant_feature = False
    recs, off = job["recs"], job["off"]
    parts = job["parts"]  # [steps, n_ranks] records per rank and step
    id_file = str(job["id_file"])
    if rank == 0:  # ncclGetUniqueId on one rank, handed to the others by the launcher's own means (here: a file)
        uid = unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise SystemExit("unique id never arrived")
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    d = DistRank(re, uid, rank, n_ranks)
    if mode in ("sharded", "sharded_fail", "peer_seq"):
        d.set_mode(capi.MODE_SEQUENTIAL)
    if mode in ("peer", "peer_seq"):
        d.peer_attach()
    if mode == "owner_seq":
        d.set_mode(capi.MODE_SEQUENTIAL)
        d.owner_attach(4096, 4096)
    if mode == "owner_sync":   # the step-synchronous form, hogwild: every rank's micro-batch pushed, counts exchanged, owners apply
        d.owner_attach(65536, 65536)
    if mode == "owner_stream":  # the streaming form with regions far smaller than a step
        d.owner_stream_attach(int(job["log2_rows"]), int(job["log2_lr"]))
    preds = []
    pos = 0
    codes = []
    if mode == "sparse_timeout":
        # A rank that is GONE (not failed-and-reporting: gone) in the middle of a job: after one good step the last rank exits without a word; the others'
        # next step finds nobody at the first collective.  With FWGPU_DIST_TIMEOUT_MS set (and FWGPU_FAKERCCL_ASYNC=1, so that the collective sits on
        # the stream like RCCL's would) the polled wait gives up, aborts the communicator and returns FWGPU_ERR_PEER -- terminal for the rank: its next
        # collective call is refused with FWGPU_ERR_INVALID (no communicator), it does not hang and does not touch freed memory.
        a = int(parts[0, :rank].sum())
        b = a + int(parts[0, rank])
        sub, so = recs[int(off[a]):int(off[b])], off[a:b + 1] - off[a]
        d.learn_sparse(fbt, sub, so)
        if rank == n_ranks - 1:
            os._exit(0)
        t0 = time.time()
        for _ in range(2):
            try:
                d.learn_sparse(fbt, sub, so)
                codes.append(0)
            except capi.FwgpuError as e:
                codes.append(e.code)
        np.savez(sys.argv[3], codes=np.asarray(codes, dtype=np.int64), seconds=np.float64(time.time() - t0))
        re.close()  # (the rank object is left to the process exit: its communicator is gone)
        return
    if mode == "owner_stream_timeout":
        # The streaming owner-side apply with a peer that is GONE: after one good step the last rank exits without a word.  The others' next step launches a kernel whose
        # consumers wait for that rank's final positions (and whose producers wait for slots it will never free): with FWGPU_DIST_TIMEOUT_MS set the host's polled wait
        # sets the abort word, every wait loop of the kernel leaves, and the call returns FWGPU_ERR_PEER; the rank's streaming state is void afterwards (FWGPU_ERR_PEER again).
        d.owner_stream_attach(int(job["log2_rows"]), int(job["log2_lr"]))
        a = int(parts[0, :rank].sum())
        b = a + int(parts[0, rank])
        sub, so = recs[int(off[a]):int(off[b])], off[a:b + 1] - off[a]
        d.learn_owner_stream(fbt, sub, so, consumer_workgroups=5 * n_ranks)
        d.barrier()
        if rank == n_ranks - 1:
            os._exit(0)
        t0 = time.time()
        for _ in range(2):
            try:
                d.learn_owner_stream(fbt, sub, so, consumer_workgroups=5 * n_ranks)
                codes.append(0)
            except capi.FwgpuError as e:
                codes.append(e.code)
        np.savez(sys.argv[3], codes=np.asarray(codes, dtype=np.int64), seconds=np.float64(time.time() - t0))
        os._exit(0)  # (the rank's mappings of the gone peer's memory are left to the process exit)
    if mode in ("sharded_fail", "sparse_fail"):
        # Failure model of the collective steps (dist.cpp): a rank whose LOCAL preparation fails still joins the shape exchange with a
        # poisoned shape; every rank returns from the step -- the culprit with its own error, the others with FWGPU_ERR_PEER -- nothing is
        # exchanged or applied, nobody hangs, and the job goes on.  Step A: the last rank passes a record whose slot word points behind the
        # record; step B (sharded only): rank 0 passes one example fewer than the others; then the ordinary steps below must still work.
        learn = d.learn_sharded if mode == "sharded_fail" else d.learn_sparse
        a, b = 0, int(parts[0, rank])
        a += int(parts[0, :rank].sum())
        b += int(parts[0, :rank].sum())
        sub, so = recs[int(off[a]):int(off[b])].copy(), off[a:b + 1] - off[a]
        bad = sub.copy()
        if rank == n_ranks - 1:
            bad[3] = 0x80000000 | (0x3ff0 << 16) | 0xfff0  # namespace 0: features at [0x3ff0, 0xfff0): far behind the record's end
        try:
            learn(fbt, bad, so)
            codes.append(0)
        except capi.FwgpuError as e:
            codes.append(e.code)
        if mode == "sharded_fail":
            n_here = len(so) - 1 - (1 if rank == 0 else 0)
            try:
                learn(fbt, sub[:int(so[n_here])], so[:n_here + 1])
                codes.append(0)
            except capi.FwgpuError as e:
                codes.append(e.code)
        mode = mode.split("_")[0]
    for s in range(parts.shape[0]):
        a = pos + int(parts[s, :rank].sum())
        b = a + int(parts[s, rank])
        sub, so = recs[int(off[a]):int(off[b])], off[a:b + 1] - off[a]
        if mode == "sharded":
            if s % 2 == 0:
                preds.append(d.learn_sharded(fbt, sub, so))
            else:  # the device-resident form of the same call
                ba = re.record_batch(fbt, sub, so)
                d.learn_sharded_batch(fbt, ba)
                preds.append(ba.predictions().copy())
                ba.close()
        elif mode == "sparse":
            preds.append(d.learn_sparse(fbt, sub, so))
        elif mode == "peer":  # hogwild across the ranks: everybody at its own pace
            preds.append(d.learn_peer(fbt, sub, so))
        elif mode == "owner_sync":
            preds.append(d.learn_owner(fbt, sub, so))
        elif mode == "owner_stream":
            preds.append(d.learn_owner_stream(fbt, sub, so, consumer_workgroups=5 * n_ranks))
        elif mode == "owner_seq":
            # owner-side apply, one example per COLLECTIVE step, the ranks taking turns in rank order: the sequential reference.  Every rank calls
            # every step; the ranks whose turn it is not pass no record.
            mine = []
            for turn in range(n_ranks):
                for e in range(int(parts[s, turn])):
                    if turn == rank:
                        mine.append(d.learn_owner(fbt, sub[int(so[e]):int(so[e + 1])], so[e:e + 2] - so[e])[0])
                    else:
                        d.learn_owner(fbt, sub[:0], so[:1])
            preds.append(np.asarray(mine, dtype=np.float32))
        elif mode == "peer_seq":  # rank after rank, each in example order: the sequential reference over the ranks' micro-batches
            for turn in range(n_ranks):
                if turn == rank:
                    preds.append(d.learn_peer(fbt, sub, so))
                d.barrier()
        else:
            raise SystemExit("unknown mode " + mode)
        pos += int(parts[s].sum())
    ranges = np.array(d.ranges(), dtype=np.uint64)
    if mode in ("sharded", "peer", "peer_seq", "owner_seq", "owner_sync", "owner_stream"):
        d.gather_tables()
    tabs = [np.asarray(re.table_read(tt)) for tt in (capi.TABLE_LR, capi.TABLE_FFM_W, capi.TABLE_FFM_ACC)]
    # the replica mode's exchange: all-reduce (sum) of a device buffer through the same communicator -- here the rank's FFM
    # accumulator table scaled by nothing: after gather_tables every rank holds the same table, so the sum is n_ranks x the table
    ar = np.zeros(0, np.float32)
    if int(job["allreduce"]):
        d.all_reduce_sum(re.table_device_ptr(capi.TABLE_FFM_ACC), re.table_len(capi.TABLE_FFM_ACC))
        ar = np.asarray(re.table_read(capi.TABLE_FFM_ACC))
    np.savez(sys.argv[3], preds=np.concatenate(preds) if preds else np.zeros(0, np.float32), lr=tabs[0], ffm_w=tabs[1], ffm_acc=tabs[2],
             ranges=ranges, allreduce=ar, codes=np.asarray(codes, dtype=np.int64))
    d.close()
    re.close()


if __name__ == "__main__":
    main()
