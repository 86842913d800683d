// Multi-GPU inside the library (BASELINE configs[3]): one process per GPU over RCCL / xGMI, or -- for tests and
// single-box emulation -- several ranks inside one process.
//
// Mode "sharded" = owner-sharded tables, synchronous micro-batches (what replaces hogwild.rs:24-103 across GPUs):
//   every rank OWNS a contiguous range of the FFM table and of the LR table (rows are owned by their start address) and keeps
//   only that range current.  One step takes a micro-batch of B records from every rank:
//     X1  all-gather the records (a record is ~1.7 KB at config C) and the batch shape
//     P1  FWD over all N*B examples, owned rows only: partial field sums T / self-pair corrections / LR sums -> split records
//     X2  reduce-scatter (sum) of the split records: every rank gets the complete records of its own B examples
//     P2  MID on the own examples: logit, prediction, general gradient
//     X3  all-gather of the gradients and of the completed records
//     P3  UPD over all N*B examples, owned rows only: AdaGrad per occurrence, exactly as on one GPU
//   The result is the synchronous micro-batch of fwgpu_learn_batch_sync with batch N*B, whatever N is (sums of field sums are
//   taken in another order: 1e-7 relative).  Rows that straddle an ownership boundary (R floats out of table/N) see the next
//   owner's copy of their tail diverge: 6e-6 of the rows at config C with 8 ranks.
//
// Mode "sparse" = full replicas, row-sparse gradient buckets (north_star's wording; sparse.hip has the kernels and the rule):
//     P1  FWD + MID over the rank's OWN micro-batch against its replica; every entry is listed as an occurrence (row, slot)
//     P2  radix sort + segment reduction: one bucket row per row hash (per 64 sorted occurrences)
//     X1  all-gather of the bucket counts, then of {keys, gradient rows} padded to the largest rank
//     P3  merge (sort by hash, rank, index) + apply: one optimizer step per row with the summed gradient, same order everywhere
//   Replicas that start identical stay bit-identical; no table ever crosses the links.
//
// RCCL is resolved at run time (dlopen librccl.so.1): a process that never calls fwgpu_dist_init does not load it, and a
// Python process that already holds torch's copy reuses it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "fwgpu_internal.h"

using namespace fwgpu;

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;  // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;              // optional
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::once_flag once;
    bool ok = false;
    bool load() {
        std::call_once(once, [this] {
            // FWGPU_RCCL_LIBRARY names the library that provides the seven entry points below, e.g. another RCCL build -- or the
            // shared-memory stand-in of tests/fake_rccl, which lets the process-per-rank path run with N > 1 on ONE GPU
            // (RTLD_LOCAL: its nccl* symbols must not shadow a real RCCL the process also holds)
            if (const char *over = std::getenv("FWGPU_RCCL_LIBRARY")) {
                if (over[0]) lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
                if (!lib) return;
            }
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                if (lib) break;
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            }
            if (!lib) return;
#define SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name))
            SYM(GetUniqueId, "ncclGetUniqueId");
            SYM(CommInitRank, "ncclCommInitRank");
            SYM(CommDestroy, "ncclCommDestroy");
            SYM(CommAbort, "ncclCommAbort");
            SYM(CommGetAsyncError, "ncclCommGetAsyncError");
            SYM(CommCount, "ncclCommCount");
            SYM(AllReduce, "ncclAllReduce");
            SYM(AllGather, "ncclAllGather");
            SYM(ReduceScatter, "ncclReduceScatter");
            SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
            ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && AllGather && ReduceScatter;
        });
        return ok;
    }
};
RcclApi g_rccl;

#define FWGPU_NCCL(call)                                                                                                  \
    do {                                                                                                                  \
        ncclResult_t e_ = (call);                                                                                         \
        if (e_ != ncclSuccess)                                                                                            \
            return fail(FWGPU_ERR_DEVICE, std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "RCCL error")); \
    } while (0)

}  // namespace

struct fwgpu_dist;
// Failure model of the collective steps (one rank per process).
//  * A failure BEFORE the step's first exchange (bad records, a batch of another regressor, an example the kernels cannot stage) is
//    what a job actually meets: the failing rank still takes part in the first collective -- the all-gather of the batch shapes -- with
//    a poisoned shape (kShapePoison in word 3), every rank sees it, and every rank returns from the step without having exchanged or applied
//    anything: the failing rank with its own error, the others with FWGPU_ERR_PEER.  The job stays usable.
//  * A failure AFTER the exchange has begun (a HIP or RCCL error mid-step) leaves this rank unable to finish its collectives.  It aborts its
//    communicator (ncclCommAbort) so that its own stream does not hang; that abort is LOCAL -- the peers' collective kernels keep waiting
//    for this rank.  They get out through wait_stream(): every wait behind a collective polls the stream and the communicator's
//    asynchronous error state (ncclCommGetAsyncError, where the library has it) and, after FWGPU_DIST_TIMEOUT_MS (default 0 = wait for
//    ever, as RCCL itself does), aborts the own communicator and returns FWGPU_ERR_PEER.  A job that wants to survive a dead rank sets it.
//    That FWGPU_ERR_PEER is TERMINAL for the rank (no communicator left), unlike the pre-exchange one.  Out of memory while the step's buffers are
//    sized (step_upload / sparse_local run AFTER the shape exchange) is a failure of this second kind: the rank aborts, its peers need the time-out.
constexpr uint32_t kShapePoison = 0xffffffffu;
static int abort_on_failure(fwgpu_dist *d, int rc);

// One rank of the job.
struct fwgpu_dist {
    fwgpu_regressor *r = nullptr;
    int rank = 0, n = 1;
    ncclComm_t comm = nullptr;       // NULL: member of an in-process group
    hipStream_t stream = nullptr;
    SplitRanges rg;
    int mode = FWGPU_MODE_HOGWILD;   // FWGPU_MODE_SEQUENTIAL: every phase on one workgroup, in example order (deterministic: tests)
    // per-step state (grown on demand)
    uint32_t B = 0;                  // records per rank in the current step
    uint64_t wcap = 0;               // words reserved per rank in the gathered batch
    fwgpu_batch *gb = nullptr;       // gathered record batch, N*B examples
    uint32_t gb_ncap = 0;
    uint64_t gb_wcap = 0;
    fwgpu_split *sp = nullptr;       // split records of the N*B examples
    uint32_t sp_n = 0, sp_ffm = 0;
    float *d_own = nullptr;          // [B * split_len] the completed records of the own examples
    size_t own_cap = 0;
    uint32_t *d_shape = nullptr;     // [n * 4] per rank: max_lr, max_ffm, max_rec, n_records (all-gathered)
    const fwgpu_translator_config *tr = nullptr;
    fwgpu_batch *src = nullptr;      // this step's own records, when they are already in HBM
    // row-sparse gradient mode (sparse.hip): one side per table
    struct SparseSide {
        uint32_t occ_cap = 0;        // occurrence slots (= bucket rows in the worst case) the buffers hold
        uint32_t width = 1;          // floats per bucket row (R, or 1 for LR)
        unsigned long long *key = nullptr, *key_sorted = nullptr;
        uint2 *desc = nullptr;
        uint32_t *flags = nullptr, *pos = nullptr, *bk_key = nullptr;
        float *bk_rows = nullptr;
        uint64_t all_cap = 0;        // gathered bucket rows of all ranks
        uint32_t *all_key = nullptr;
        float *all_rows = nullptr;
        unsigned long long *m_key = nullptr, *m_key_sorted = nullptr;
        uint32_t stride = 0;         // this step: bucket rows reserved per rank in the gathered arrays
        void release() {
            for (void *q : {(void *)key, (void *)key_sorted, (void *)desc, (void *)flags, (void *)pos, (void *)bk_key, (void *)bk_rows,
                            (void *)all_key, (void *)all_rows, (void *)m_key, (void *)m_key_sorted})
                if (q) (void)hipFree(q);
            *this = SparseSide();
        }
    } sf, sl;
    void *sp_tmp = nullptr;          // rocPRIM scratch
    size_t sp_tmp_bytes = 0;
    uint32_t *d_counts = nullptr;    // [2 + 2 * n]: own {ffm, lr} bucket rows, then every rank's
    fwgpu_batch *ob = nullptr;       // own records of the step (host-record entry point)
    uint32_t ob_ncap = 0;
    uint64_t ob_wcap = 0;
    fwgpu_batch *cur = nullptr;      // the batch this sparse step runs on
    uint32_t occ_max_ffm = 0, occ_max_lr = 0;
    hipEvent_t ev_peer_step = nullptr;  // recorded behind a peer-sharded step launched on a stream of the caller's (fwgpu_dist_barrier waits for it)
    int begin_rc = FWGPU_OK;         // result of the current step's local preparation (failure model: dist.cpp top)
    bool begin_failed = false;
    hipEvent_t ev_prev = nullptr;    // group step: "the previous rank's local phase is done" (device-side ordering of the ranks)
    // owner-side apply (fwgpu_dist_*_owner): as OWNER, one region per source rank and step parity for the gradient rows pushed to this rank
    // (one allocation: keys | rows | LR entries); as SOURCE, position counters and the ring descriptor its kernel reads
    unsigned char *own_rings = nullptr;
    size_t own_rings_bytes = 0;
    uint32_t ring_cap_ffm = 0, ring_cap_lr = 0, ring_parity = 0;
    uint32_t *d_push_cnt = nullptr;   // [2n + 1] (+ [n * (2n + 1)] behind it: every rank's counts after the step's all-gather)
    PushRings *d_push = nullptr;
    unsigned char *peer_rings[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // the owners' ring allocations as mapped here
    bool rings_attached = false;
    // streaming form of the owner-side apply (fwgpu_dist_*_owner_stream): ONE allocation for both roles -- as owner, a circular region per source (tag
    // words | gradient rows | LR words) and the final positions its consumer polls; as source, per owner the slots' free generations, the LR credits and
    // the position counters (layout: StreamGeom)
    unsigned char *st_mem = nullptr;
    size_t st_bytes = 0;
    uint32_t st_lg_ffm = 0, st_lg_lr = 0, st_n = 0;
    uint32_t st_pos_ffm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_pos_lr[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // as OWNER: positions of source s's rings consumed so far
    bool st_probed = false;            // in-process group: the concurrency probe of the ranks sharing a device has passed
    uint32_t st_share = 1;             // ranks of the job on this rank's device (their kernels must be resident together: each takes an equal share of the device)
    uint32_t st_step = 0;              // streaming steps since the last reset (tags the final positions: a consumer only believes its own step's)
    uint32_t *st_abort = nullptr;      // pinned host word every wait loop of a streaming launch looks at (kernels.hip PushRings::abort): set by stream_finish when a step
                                       // outlives its deadline -- written by a plain host store, so that giving a step up needs no queue of a device whose CUs the step holds
    uint32_t st_next_check = 0;        // process-per-rank form: the step at which the ranks next agree on "is any region close to the 32-bit position wrap"
    bool st_void = false;              // a step was given up: positions, tags and free generations no longer agree between the ranks (terminal for the streaming form of this rank)
    unsigned char *st_peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // every rank's st_mem as reachable from here
    PeerShards *d_peers = nullptr;   // peer-sharded step: the owners' table bases, on this rank's device
    bool peers_attached = false;     // process-per-rank peer mode: the other ranks' tables are mapped (hipIpcOpenMemHandle)
    float *lr_shard = nullptr;       // ... this rank's OWNED range of the LR table in an allocation of its own while the mode is on: what the
    uint64_t lr_shard_lo = 0, lr_shard_n = 0;  // peers map (hipIpcOpenMemHandle hangs on allocations of 2 GiB and more: a `-b 28` LR table is 2 GiB)
    std::vector<void *> ipc_open;    // ... what to close again
    hipEvent_t dbg_ev[3] = {nullptr, nullptr, nullptr};  // debug (scripts/group_bisect.sh): recorded behind FWD / MID / the FFM reduction of sparse_local
    uint32_t last_rows[2] = {0, 0};  // bucket rows {ffm, lr} this rank sent in its last sparse step
    // shapes / counts read back behind a collective land HERE (pinned, owned by the rank, alive until fwgpu_dist_free): a copy into pageable memory
    // blocks the host until the stream drains -- with a dead peer it would never reach wait_stream()'s polling -- and a copy into a function-local
    // buffer could land after the function has given up (wait_stream's time-out path)
    uint32_t *h_pin = nullptr;
    size_t h_pin_words = 0;
    int pinned(size_t words, uint32_t **out) {
        if (words > h_pin_words) {
            if (h_pin) (void)hipHostFree(h_pin);
            h_pin = nullptr;
            h_pin_words = 0;
            const size_t cap = std::max<size_t>(words, 1024);
            if (hipHostMalloc((void **)&h_pin, cap * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return FWGPU_ERR_OOM;
            h_pin_words = cap;
        }
        *out = h_pin;
        return FWGPU_OK;
    }
    ~fwgpu_dist() {
        sf.release();
        sl.release();
        if (h_pin) (void)hipHostFree(h_pin);
        if (sp_tmp) (void)hipFree(sp_tmp);
        if (d_counts) (void)hipFree(d_counts);
        if (ob) fwgpu_batch_free(ob);
        if (gb) fwgpu_batch_free(gb);
        if (sp) fwgpu_split_free(sp);
        if (d_own) (void)hipFree(d_own);
        for (void *q : ipc_open) (void)hipIpcCloseMemHandle(q);
        if (lr_shard) (void)hipFree(lr_shard);
        if (st_mem) (void)hipFree(st_mem);
        if (st_abort) (void)hipHostFree(st_abort);
        if (d_peers) (void)hipFree(d_peers);
        if (own_rings) (void)hipFree(own_rings);
        if (d_push_cnt) (void)hipFree(d_push_cnt);
        if (d_push) (void)hipFree(d_push);
        if (d_shape) (void)hipFree(d_shape);
        if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct fwgpu_dist_group {
    std::vector<std::unique_ptr<fwgpu_dist>> ranks;
};

namespace {

void set_ranges(fwgpu_dist *d) {
    const fwgpu_regressor *r = d->r;
    const uint64_t ffm_span = r->cfg.ffm_k ? (1ull << r->cfg.ffm_bit_precision) : 0, lr_span = r->lr_len;
    // contiguous, equal ranges; row starts are multiples of next_pow2(k) <= 2^ffm_bits / n for every sane n
    const uint64_t fper = ffm_span / d->n, lper = lr_span / d->n;
    d->rg.ffm_lo = (uint32_t)(fper * d->rank);
    d->rg.ffm_hi = d->rank == d->n - 1 ? 0xffffffffu : (uint32_t)(fper * (d->rank + 1));
    d->rg.lr_lo = (uint32_t)(lper * d->rank);
    d->rg.lr_hi = d->rank == d->n - 1 ? 0xffffffffu : (uint32_t)(lper * (d->rank + 1));
}

// ---- step, rank-local parts.  S0: take this rank's records, size the buffers.
int check_shardable(const fwgpu_dist *d) {
    const fwgpu_regressor *r = d->r;
    if (r->cfg.ffm_k && (((1ull << r->cfg.ffm_bit_precision) / d->n) % 64 != 0 || (1ull << r->cfg.ffm_bit_precision) % d->n != 0))
        return fail(FWGPU_ERR_INVALID, "sharded step: the FFM table does not split into n ranges of whole 256 B blocks");
    // the LR table is gathered as n equal ranges too (fwgpu_dist_gather_tables): a rank count that does not divide 2^bits would leave
    // the last rank's tail owned but never gathered
    if (r->cfg.wiring != FWGPU_WIRING_FFM_ONLY && r->lr_len % (uint64_t)d->n != 0)
        return fail(FWGPU_ERR_INVALID, "sharded step: the LR table does not split into n equal ranges (n must divide 2^bit_precision)");
    return FWGPU_OK;
}

int step_begin(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
               uint32_t shape[4]) {
    if (!t || (n && (!records || !rec_off))) return fail(FWGPU_ERR_INVALID, "dist step: NULL argument");
    if (int rc0 = check_shardable(d)) return rc0;
    FWGPU_HIP(hipSetDevice(d->r->device));
    RecordStats st;
    int rc = count_records(t, records, rec_off, n, &st);
    if (rc) return rc;
    shape[0] = st.max_lr;
    shape[1] = st.max_ffm;
    shape[2] = st.max_rec;
    shape[3] = n;
    d->tr = t;
    d->B = n;
    d->src = nullptr;
    return FWGPU_OK;
}
// the rank's micro-batch is already a record batch in HBM (fwgpu_record_batch_create): its shape is known, nothing crosses PCIe
int step_begin_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b, uint32_t shape[4]) {
    if (!t || !b || !b->records) return fail(FWGPU_ERR_INVALID, "dist step: a record batch is needed");
    if (b->owner != d->r) return fail(FWGPU_ERR_INVALID, "dist step: batch belongs to another regressor");
    if (int rc0 = check_shardable(d)) return rc0;
    FWGPU_HIP(hipSetDevice(d->r->device));
    shape[0] = b->max_lr;
    shape[1] = b->max_ffm;
    shape[2] = b->max_rec;
    shape[3] = b->n;
    d->tr = t;
    d->B = b->n;
    d->src = b;
    return FWGPU_OK;
}

// after the shapes of all ranks are known: (re)allocate the gathered batch and the split buffers, upload the own records into
// their slot of the gathered batch
int step_upload(fwgpu_dist *d, const uint32_t *records, const uint64_t *rec_off, const uint32_t *shapes /*[n*4]*/) {
    uint32_t max_lr = 0, max_ffm = 0, max_rec = 0;
    for (int j = 0; j < d->n; j++) {
        if (shapes[4 * j + 3] != d->B) return fail(FWGPU_ERR_INVALID, "sharded step: every rank must bring the same number of records");
        max_lr = std::max(max_lr, shapes[4 * j]);
        max_ffm = std::max(max_ffm, shapes[4 * j + 1]);
        max_rec = std::max(max_rec, shapes[4 * j + 2]);
    }
    const uint32_t B = d->B, NB = B * (uint32_t)d->n;
    const uint64_t wcap = ((uint64_t)B * max_rec + 63) & ~63ull;  // words per rank (upper bound: every record of maximal length)
    d->wcap = wcap;
    if (!d->gb || d->gb_ncap < NB || d->gb_wcap < wcap * d->n) {
        if (d->gb) fwgpu_batch_free(d->gb);
        d->gb = nullptr;
        int rc = record_batch_alloc(d->r, d->tr, NB, wcap * d->n, &d->gb);
        if (rc) return rc;
        d->gb_ncap = NB;
        d->gb_wcap = wcap * d->n;
    }
    if (!d->sp || d->sp_n < NB || d->sp_ffm < max_ffm) {
        if (d->sp) fwgpu_split_free(d->sp);
        d->sp = nullptr;
        int rc = fwgpu_split_create(d->r, NB, std::max<uint32_t>(max_ffm, 16), &d->sp);
        if (rc) return rc;
        d->sp_n = NB;
        d->sp_ffm = max_ffm;
    }
    const size_t own_floats = (size_t)B * d->sp->split_len;
    if (d->own_cap < own_floats) {
        if (d->d_own) (void)hipFree(d->d_own);
        d->d_own = nullptr;
        FWGPU_HIP(hipMalloc((void **)&d->d_own, own_floats * 4));
        d->own_cap = own_floats;
    }
    fwgpu_batch *gb = d->gb;
    gb->n = NB;
    gb->max_lr = d->r->cfg.wiring == FWGPU_WIRING_FFM_ONLY ? 0 : max_lr;
    gb->max_ffm = max_ffm;
    gb->max_rec = max_rec;
    gb->aligned4 = true;
    gb->rec_self_len = true;
    if (d->src && B) {  // device to device: the words as they are, the offsets shifted onto the gathered buffer
        if (d->src->n_words > wcap) return fail(FWGPU_ERR_RANGE, "sharded step: records longer than announced");
        FWGPU_HIP(hipMemcpyAsync(gb->records + wcap * d->rank, d->src->records, d->src->n_words * 4, hipMemcpyDeviceToDevice, d->stream));
        FWGPU_HIP(launch_offset_copy(gb->rec_off + (size_t)B * d->rank, d->src->rec_off, B, wcap * d->rank, d->stream));
        return FWGPU_OK;
    }
    // own records into slot `rank`: words at rank * wcap, offsets rebased onto the gathered buffer
    const uint64_t words = B ? rec_off[B] - rec_off[0] : 0;
    if (words > wcap) return fail(FWGPU_ERR_RANGE, "sharded step: records longer than announced");
    std::vector<uint64_t> off(B);
    for (uint32_t i = 0; i < B; i++) off[i] = rec_off[i] - rec_off[0] + wcap * d->rank;
    if (B) {
        FWGPU_HIP(hipMemcpyAsync(gb->records + wcap * d->rank, records + rec_off[0], words * 4, hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipMemcpyAsync(gb->rec_off + (size_t)B * d->rank, off.data(), (size_t)B * 8, hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));  // `off` lives on this stack frame
    }
    return FWGPU_OK;
}

int phase_fwd(fwgpu_dist *d) {
    SplitRanges rg = d->rg;
    rg.home_lo = d->B * d->rank;
    rg.home_hi = d->B * (d->rank + 1);
    return split_forward(d->r, d->gb, d->sp, d->mode, rg, d->stream);
}
int phase_mid(fwgpu_dist *d) {
    // the completed records of the own examples sit in d_own; MID reads them through a split view of their own
    fwgpu_split view = *d->sp;
    view.d_split = d->d_own;
    view.d_g = d->sp->d_g + (size_t)d->B * d->rank;  // own slice of the gathered gradient array
    return split_mid(d->r, &view, 0, d->B, d->gb->pred + (size_t)d->B * d->rank, false, d->stream);
}
int phase_upd(fwgpu_dist *d) { return split_update(d->r, d->gb, d->sp, d->mode, d->rg, false, d->stream); }

int wait_stream_fwd(fwgpu_dist *d);
int finish(fwgpu_dist *d, float *preds) {
    if (int rc = wait_stream_fwd(d)) return rc;  // polled where a time-out is set (RCCL ranks); the caller's pageable buffer is filled after the wait
    if (preds && d->B) FWGPU_HIP(hipMemcpy(preds, d->gb->pred + (size_t)d->B * d->rank, (size_t)d->B * 4, hipMemcpyDeviceToHost));
    return FWGPU_OK;
}

// ------------------------------------------------------------------ row-sparse gradient buckets (sparse.hip)
int ceil_log2(uint64_t v) {
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

template <typename T>
int grow(T *&q, size_t count) {
    if (q) (void)hipFree(q);
    q = nullptr;
    FWGPU_HIP(hipMalloc((void **)&q, std::max<size_t>(count, 1) * sizeof(T)));
    return FWGPU_OK;
}

int sparse_side_reserve(fwgpu_dist::SparseSide &s, uint32_t occ_cap, uint32_t width) {
    if (s.occ_cap >= occ_cap && s.width == width) return FWGPU_OK;
    s.release();
    s.width = width;
    int rc;
    if ((rc = grow(s.key, occ_cap)) || (rc = grow(s.key_sorted, occ_cap)) || (rc = grow(s.desc, occ_cap)) ||
        (rc = grow(s.flags, (size_t)occ_cap + 1)) || (rc = grow(s.pos, (size_t)occ_cap + 1)) || (rc = grow(s.bk_key, occ_cap)) ||
        (rc = grow(s.bk_rows, (size_t)occ_cap * width)))
        return rc;
    s.occ_cap = occ_cap;
    return FWGPU_OK;
}

int sparse_side_reserve_all(fwgpu_dist::SparseSide &s, uint64_t rows) {
    if (s.all_cap >= rows) return FWGPU_OK;
    for (void *q : {(void *)s.all_key, (void *)s.all_rows, (void *)s.m_key, (void *)s.m_key_sorted})
        if (q) (void)hipFree(q);
    s.all_key = nullptr;
    s.all_rows = nullptr;
    s.m_key = s.m_key_sorted = nullptr;
    s.all_cap = 0;
    int rc;
    if ((rc = grow(s.all_key, rows)) || (rc = grow(s.all_rows, rows * s.width)) || (rc = grow(s.m_key, rows)) || (rc = grow(s.m_key_sorted, rows)))
        return rc;
    s.all_cap = rows;
    return FWGPU_OK;
}

// S0: the step's records (host records are uploaded into the rank's own record batch; a device batch is used as it is)
int sparse_begin(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                 fwgpu_batch *dev_batch, uint32_t shape[4]) {
    if (d->r->nn.n_layers) return fail(FWGPU_ERR_INVALID, "sparse step: models with a deep head are not covered");
    if (d->r->cfg.ffm_k > 64) return fail(FWGPU_ERR_INVALID, "sparse step: ffm_k > 64 is not covered");
    FWGPU_HIP(hipSetDevice(d->r->device));
    d->tr = t;
    if (dev_batch) {
        if (!t || !dev_batch->records) return fail(FWGPU_ERR_INVALID, "sparse step: a record batch is needed");
        if (dev_batch->owner != d->r) return fail(FWGPU_ERR_INVALID, "sparse step: batch belongs to another regressor");
        d->cur = dev_batch;
    } else {
        if (!t || (n && (!records || !rec_off))) return fail(FWGPU_ERR_INVALID, "sparse step: NULL argument");
        RecordStats st;
        int rc = count_records(t, records, rec_off, n, &st);
        if (rc) return rc;
        const uint64_t words = n ? rec_off[n] - rec_off[0] : 0;
        if (!d->ob || d->ob_ncap < n || d->ob_wcap < words) {
            if (d->ob) fwgpu_batch_free(d->ob);
            d->ob = nullptr;
            rc = record_batch_alloc(d->r, t, std::max<uint32_t>(n, 1), std::max<uint64_t>(words, 64), &d->ob);
            if (rc) return rc;
            d->ob_ncap = std::max<uint32_t>(n, 1);
            d->ob_wcap = std::max<uint64_t>(words, 64);
        }
        if (n) {
            rc = record_batch_upload(d->ob, t, records, rec_off, n, d->stream, &st);
            if (rc) return rc;
        } else {
            d->ob->n = 0;
            d->ob->max_lr = d->ob->max_ffm = d->ob->max_rec = 0;
        }
        d->cur = d->ob;
    }
    {
        static const bool dbg_sync_upload = std::getenv("FWGPU_DBG_SYNC_UPLOAD") != nullptr;  // debug: the upload is complete before anything else is enqueued
        if (dbg_sync_upload) FWGPU_HIP(hipStreamSynchronize(d->stream));
    }
    d->B = d->cur->n;
    shape[0] = d->cur->max_lr;
    shape[1] = d->cur->max_ffm;
    shape[2] = d->cur->max_rec;
    shape[3] = d->cur->n;
    return FWGPU_OK;
}

// S1: buffers for the largest batch shape of the job, then FWD, MID and the local segment reduction; the bucket-row counts
// {ffm, lr} end up in d_counts[0..1]
int sparse_local(fwgpu_dist *d, const uint32_t *shapes /*[n*4]*/) {
    fwgpu_regressor *r = d->r;
    uint32_t max_lr = 0, max_ffm = 0, max_n = 0;
    for (int j = 0; j < d->n; j++) {
        max_lr = std::max(max_lr, shapes[4 * j]);
        max_ffm = std::max(max_ffm, shapes[4 * j + 1]);
        max_n = std::max(max_n, shapes[4 * j + 3]);
    }
    const bool has_lr = r->cfg.wiring != FWGPU_WIRING_FFM_ONLY, has_ffm = r->cfg.ffm_k != 0;
    const uint32_t F = has_ffm ? r->cfg.ffm_num_fields : 0, R = F * r->cfg.ffm_k;
    const uint64_t sf_cap = has_ffm ? (uint64_t)max_n * std::max<uint32_t>(4, (max_ffm + 3) & ~3u) : 0;
    const uint64_t sl_cap = has_lr ? (uint64_t)max_n * std::max<uint32_t>(4, (max_lr + 3) & ~3u) : 0;
    if (sf_cap >= 0xffffffffull || sl_cap >= 0xffffffffull) return fail(FWGPU_ERR_RANGE, "sparse step: micro-batch too large for 32-bit occurrence slots");
    int rc;
    if (has_ffm && (rc = sparse_side_reserve(d->sf, (uint32_t)sf_cap, R))) return rc;
    if (has_lr && (rc = sparse_side_reserve(d->sl, (uint32_t)sl_cap, 1))) return rc;
    const size_t tmp = sparse_tmp_bytes((uint32_t)std::max<uint64_t>(std::max(sf_cap, sl_cap) * (uint64_t)d->n, 1));
    if (d->sp_tmp_bytes < tmp) {
        if (d->sp_tmp) (void)hipFree(d->sp_tmp);
        d->sp_tmp = nullptr;
        FWGPU_HIP(hipMalloc(&d->sp_tmp, tmp));
        d->sp_tmp_bytes = tmp;
    }
    if (!d->d_counts) {
        FWGPU_HIP(hipMalloc((void **)&d->d_counts, (size_t)(2 + 2 * d->n) * 4));
    }
    FWGPU_HIP(hipMemsetAsync(d->d_counts, 0, (size_t)(2 + 2 * d->n) * 4, d->stream));
    const uint32_t n = d->B;
    if (!n) return FWGPU_OK;
    if (!d->sp || d->sp_n < n || d->sp_ffm < max_ffm) {
        if (d->sp) fwgpu_split_free(d->sp);
        d->sp = nullptr;
        rc = fwgpu_split_create(r, std::max(n, max_n), std::max<uint32_t>(max_ffm, 16), &d->sp);
        if (rc) return rc;
        d->sp_n = std::max(n, max_n);
        d->sp_ffm = max_ffm;
    }
    {
        static const bool dbg_addr = std::getenv("FWGPU_DBG_ADDR") != nullptr;  // debug: where every per-rank buffer of the step lies
        if (dbg_addr) {
            auto pr = [&](const char *name, const void *q, size_t bytes) {
                std::fprintf(stderr, "[addr] rank %d %-14s %p .. %p (%zu B)\n", d->rank, name, q, (const char *)q + bytes, bytes);
            };
            pr("sf.key", d->sf.key, (size_t)d->sf.occ_cap * 8);
            pr("sf.key_sorted", d->sf.key_sorted, (size_t)d->sf.occ_cap * 8);
            pr("sf.desc", d->sf.desc, (size_t)d->sf.occ_cap * 8);
            pr("sf.flags", d->sf.flags, ((size_t)d->sf.occ_cap + 1) * 4);
            pr("sf.pos", d->sf.pos, ((size_t)d->sf.occ_cap + 1) * 4);
            pr("sf.bk_key", d->sf.bk_key, (size_t)d->sf.occ_cap * 4);
            pr("sf.bk_rows", d->sf.bk_rows, (size_t)d->sf.occ_cap * d->sf.width * 4);
            pr("sl.key", d->sl.key, (size_t)d->sl.occ_cap * 8);
            pr("sl.key_sorted", d->sl.key_sorted, (size_t)d->sl.occ_cap * 8);
            pr("sl.desc", d->sl.desc, (size_t)d->sl.occ_cap * 8);
            pr("sl.flags", d->sl.flags, ((size_t)d->sl.occ_cap + 1) * 4);
            pr("sl.pos", d->sl.pos, ((size_t)d->sl.occ_cap + 1) * 4);
            pr("sl.bk_key", d->sl.bk_key, (size_t)d->sl.occ_cap * 4);
            pr("sl.bk_rows", d->sl.bk_rows, (size_t)d->sl.occ_cap * 4);
            pr("sp_tmp", d->sp_tmp, d->sp_tmp_bytes);
            pr("d_counts", d->d_counts, (size_t)(2 + 2 * d->n) * 4);
            pr("sp.d_split", d->sp->d_split, (size_t)d->sp->n_cap * d->sp->split_len * 4);
            pr("sp.d_selfw", d->sp->d_selfw, (size_t)d->sp->n_cap * d->sp->selfw_stride * 4);
            pr("sp.d_g", d->sp->d_g, (size_t)d->sp->n_cap * 8);
            pr("cur.records", d->cur->records, (size_t)d->cur->n_words * 4);
            pr("cur.pred", d->cur->pred, (size_t)d->cur->n * 4);
            pr("ffm_w", r->d_ffm_w, (size_t)r->ffm_len * 4);
            pr("ffm_acc", r->d_ffm_acc, (size_t)r->ffm_len * 4);
            pr("lr", r->d_lr, (size_t)r->lr_len * 8);
            std::fprintf(stderr, "[addr] rank %d n %u max_ffm %u max_lr %u sf_cap %llu sl_cap %llu\n", d->rank, n, max_ffm, max_lr,
                         (unsigned long long)sf_cap, (unsigned long long)sl_cap);
        }
    }
    OccBuffers occ;
    occ.ffm_key = has_ffm ? d->sf.key : nullptr;
    occ.ffm_desc = d->sf.desc;
    occ.lr_key = has_lr ? d->sl.key : nullptr;
    occ.lr_desc = d->sl.desc;
    SplitRanges rg;  // full replica: every row is this rank's, every example of the launch is its own
    rc = split_forward(r, d->cur, d->sp, FWGPU_MODE_HOGWILD, rg, d->stream, &occ, &d->occ_max_ffm, &d->occ_max_lr);
    if (rc) return rc;
    if (d->dbg_ev[0]) FWGPU_HIP(hipEventRecord(d->dbg_ev[0], d->stream));
    rc = split_mid(r, d->sp, 0, n, d->cur->pred, false, d->stream);
    if (rc) return rc;
    if (d->dbg_ev[1]) FWGPU_HIP(hipEventRecord(d->dbg_ev[1], d->stream));
    if (has_ffm) {
        SparseReduceArgs a{};
        a.keys = d->sf.key;
        a.keys_sorted = d->sf.key_sorted;
        a.n = n * d->occ_max_ffm;
        a.key_bits = 32 + ceil_log2(r->ffm_len);
        a.desc = d->sf.desc;
        a.max_entries = d->occ_max_ffm;
        a.flags = d->sf.flags;
        a.pos = d->sf.pos;
        a.tmp = d->sp_tmp;
        a.tmp_bytes = d->sp_tmp_bytes;
        a.R = R;
        a.k = r->cfg.ffm_k;
        a.split = d->sp->d_split;
        a.split_len = d->sp->split_len;
        a.selfw = d->sp->d_selfw;
        a.selfw_stride = d->sp->selfw_stride;
        a.gbuf = d->sp->d_g;
        a.bk_key = d->sf.bk_key;
        a.bk_rows = d->sf.bk_rows;
        a.d_count = d->d_counts;
        if (a.n > d->sf.occ_cap) return fail(FWGPU_ERR_RANGE, "sparse step: occurrence buffers too small");
        FWGPU_HIP(sparse_reduce(a, d->stream));
    }
    if (d->dbg_ev[2]) FWGPU_HIP(hipEventRecord(d->dbg_ev[2], d->stream));
    if (has_lr) {
        SparseReduceArgs a{};
        a.keys = d->sl.key;
        a.keys_sorted = d->sl.key_sorted;
        a.n = n * d->occ_max_lr;
        a.key_bits = 32 + ceil_log2(r->lr_len);
        a.desc = d->sl.desc;
        a.max_entries = d->occ_max_lr;
        a.flags = d->sl.flags;
        a.pos = d->sl.pos;
        a.tmp = d->sp_tmp;
        a.tmp_bytes = d->sp_tmp_bytes;
        a.R = 0;
        a.gbuf = d->sp->d_g;
        a.bk_key = d->sl.bk_key;
        a.bk_rows = d->sl.bk_rows;
        a.d_count = d->d_counts + 1;
        if (a.n > d->sl.occ_cap) return fail(FWGPU_ERR_RANGE, "sparse step: occurrence buffers too small");
        FWGPU_HIP(sparse_reduce(a, d->stream));
    }
    return FWGPU_OK;
}

// S3: all ranks' buckets are in all_key / all_rows (stride rows per rank, counts on the device): merge and apply
int sparse_apply_side(fwgpu_dist *d, fwgpu_dist::SparseSide &s, bool ffm, const uint32_t *all_key, const float *all_rows,
                      const uint32_t *d_counts, uint32_t n_ranks, uint32_t stride) {
    fwgpu_regressor *r = d->r;
    if (!stride) return FWGPU_OK;
    SparseApplyArgs a{};
    a.all_key = all_key;
    a.all_rows = all_rows;
    a.counts = d_counts;
    a.n_ranks = n_ranks;
    a.stride = stride;
    a.keys = s.m_key;
    a.keys_sorted = s.m_key_sorted;
    a.key_bits = 32 + ceil_log2(ffm ? r->ffm_len : r->lr_len);
    a.tmp = d->sp_tmp;
    a.tmp_bytes = d->sp_tmp_bytes;
    if (ffm) {
        a.R = s.width;
        a.k4 = r->cfg.ffm_k % 4 == 0;
        a.w = r->d_ffm_w;
        a.acc = r->d_ffm_acc;
        a.rate = r->cfg.ffm_learning_rate;
        a.minus_power_t = -r->cfg.ffm_power_t;
        a.lut = r->d_lut_ffm;
    } else {
        a.R = 0;
        a.w = r->d_lr;
        a.rate = r->cfg.learning_rate;
        a.minus_power_t = -r->cfg.power_t;
        a.lut = r->d_lut_lr;
    }
    FWGPU_HIP(sparse_apply(a, r->cfg.optimizer, d->stream));
    return FWGPU_OK;
}

int make_rank(fwgpu_regressor *r, int rank, int n, fwgpu_dist **out) {
    if (!r || !out || n < 1 || rank < 0 || rank >= n) return fail(FWGPU_ERR_INVALID, "dist: bad rank / size");
    if (r->nn.n_layers) return fail(FWGPU_ERR_INVALID, "dist: the sharded mode does not cover models with a deep head yet");
    FWGPU_HIP(hipSetDevice(r->device));
    std::unique_ptr<fwgpu_dist> d(new fwgpu_dist());
    d->r = r;
    d->rank = rank;
    d->n = n;
    FWGPU_HIP(hipDeviceSynchronize());  // (the rank's stream does not wait for the NULL stream: table initialisation must be done)
    {
        static const char *sf = std::getenv("FWGPU_DBG_STREAM_FLAGS");  // debug (scripts/group_bisect.sh): "default" = a blocking stream
        FWGPU_HIP(hipStreamCreateWithFlags(&d->stream, sf && sf[0] == 'd' ? hipStreamDefault : hipStreamNonBlocking));
    }
    FWGPU_HIP(hipMalloc((void **)&d->d_shape, (size_t)n * 4 * sizeof(uint32_t)));
    set_ranges(d.get());
    *out = d.release();
    return FWGPU_OK;
}

}  // namespace

extern "C" {

int fwgpu_dist_barrier(fwgpu_dist *d);

// ------------------------------------------------------------------ RCCL: one rank per process
int fwgpu_dist_unique_id(uint8_t *id, uint64_t cap) {
    if (!id || cap < sizeof(ncclUniqueId)) return fail(FWGPU_ERR_INVALID, "unique id buffer must hold 128 bytes");
    if (!g_rccl.load()) return fail(FWGPU_ERR_DEVICE, "librccl.so.1 is not available");
    ncclUniqueId u;
    FWGPU_NCCL(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return FWGPU_OK;
}

int fwgpu_dist_init(fwgpu_regressor *r, const uint8_t *unique_id, int rank, int n_ranks, fwgpu_dist **out) {
    if (!unique_id) return fail(FWGPU_ERR_INVALID, "NULL unique id");
    if (!g_rccl.load()) return fail(FWGPU_ERR_DEVICE, "librccl.so.1 is not available");
    fwgpu_dist *d = nullptr;
    int rc = make_rank(r, rank, n_ranks, &d);
    if (rc) return rc;
    ncclUniqueId u;
    std::memcpy(&u, unique_id, sizeof(u));
    ncclResult_t e = g_rccl.CommInitRank(&d->comm, n_ranks, u, rank);
    if (e != ncclSuccess) {
        delete d;
        return fail(FWGPU_ERR_DEVICE, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "failed"));
    }
    *out = d;
    return FWGPU_OK;
}

static int abort_on_failure(fwgpu_dist *d, int rc) {
    if (rc != FWGPU_OK && rc != FWGPU_ERR_PEER && d && !d->begin_failed && d->comm && d->n > 1 && g_rccl.CommAbort) {
        const std::string msg = fwgpu_last_error();
        (void)g_rccl.CommAbort(d->comm);
        d->comm = nullptr;
        set_error(msg + " (this rank's communicator was aborted: it cannot continue; peers leave their collective through FWGPU_DIST_TIMEOUT_MS)");
    }
    return rc;
}

// Wait for the rank's stream behind a collective.  Without a time-out this is hipStreamSynchronize; with FWGPU_DIST_TIMEOUT_MS it polls, so that
// a rank whose peer has died leaves the collective with an error instead of waiting for ever.
static int wait_stream(fwgpu_dist *d) {
    static const long timeout_ms = [] {
        const char *e = std::getenv("FWGPU_DIST_TIMEOUT_MS");
        return e ? std::atol(e) : 0L;
    }();
    if (timeout_ms <= 0 || !d->comm || d->n <= 1) {
        FWGPU_HIP(hipStreamSynchronize(d->stream));
        return FWGPU_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(d->stream);
        if (q == hipSuccess) return FWGPU_OK;
        if (q != hipErrorNotReady) return fail(FWGPU_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        ncclResult_t ae = ncclSuccess;
        const bool async_err = g_rccl.CommGetAsyncError && g_rccl.CommGetAsyncError(d->comm, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress;
        const bool late = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms;
        if (async_err || late) {
            if (g_rccl.CommAbort) (void)g_rccl.CommAbort(d->comm);
            d->comm = nullptr;
            // the aborted collective's kernel leaves the stream; what was queued behind it (read-backs into d->h_pin, owned by the rank) drains.
            // Bounded: a stream that does not come back is left to fwgpu_dist_free.  TERMINAL for this rank: it has no communicator any more,
            // every later collective call of it returns FWGPU_ERR_INVALID; only the pre-exchange failures (poisoned shape) let a job go on.
            for (int i = 0; i < 40000 && hipStreamQuery(d->stream) == hipErrorNotReady; i++) std::this_thread::sleep_for(std::chrono::microseconds(50));
            return fail(FWGPU_ERR_PEER, async_err ? "a collective reported an asynchronous RCCL error: communicator aborted, this rank cannot take part in further collective steps"
                                                  : "a collective did not complete within FWGPU_DIST_TIMEOUT_MS (a peer rank is gone?): communicator aborted, this rank cannot take part in further collective steps");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

}  // extern "C"
namespace {
int wait_stream_fwd(fwgpu_dist *d) { return wait_stream(d); }
}  // namespace
extern "C" {

// after the all-gather of the batch shapes: did any rank poison its shape (a failure before the exchange)?
static int check_shapes(fwgpu_dist *d, const std::vector<uint32_t> &shapes, bool same_n) {
    for (int j = 0; j < d->n; j++)
        if (shapes[4 * (size_t)j + 3] == kShapePoison) {
            if (j == d->rank) return d->begin_rc;  // (this rank's own error; its message is in fwgpu_last_error())
            return fail(FWGPU_ERR_PEER, "rank " + std::to_string(j) + " reported a failure before the step's exchange: nothing was exchanged or applied");
        }
    if (same_n)
        for (int j = 0; j < d->n; j++)
            if (shapes[4 * (size_t)j + 3] != shapes[3])
                return fail(FWGPU_ERR_PEER, "sharded step: the ranks passed different micro-batch sizes (rank 0: " + std::to_string(shapes[3]) + ", rank " +
                                                std::to_string(j) + ": " + std::to_string(shapes[4 * (size_t)j + 3]) + "): nothing was exchanged or applied");
    return FWGPU_OK;
}

int fwgpu_dist_free(fwgpu_dist *d) {
    delete d;
    return FWGPU_OK;
}

int fwgpu_dist_set_mode(fwgpu_dist *d, int mode) {
    if (!d || (mode != FWGPU_MODE_SEQUENTIAL && mode != FWGPU_MODE_HOGWILD)) return fail(FWGPU_ERR_INVALID, "dist_set_mode: bad argument");
    d->mode = mode;
    return FWGPU_OK;
}

// ranks of the job AS THE COMMUNICATOR COUNTS THEM (ncclCommCount), not as the launcher's environment says; 0 for a member of an in-process group
int fwgpu_dist_comm_count(const fwgpu_dist *d, int *count) {
    if (!d || !count) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *count = 0;
    if (!d->comm) return FWGPU_OK;
    if (!g_rccl.CommCount) {
        *count = d->n;
        return FWGPU_OK;
    }
    FWGPU_NCCL(g_rccl.CommCount(d->comm, count));
    return FWGPU_OK;
}

int fwgpu_dist_rank(const fwgpu_dist *d, int *rank, int *n_ranks) {
    if (!d) return fail(FWGPU_ERR_INVALID, "NULL dist");
    if (rank) *rank = d->rank;
    if (n_ranks) *n_ranks = d->n;
    return FWGPU_OK;
}

// what this rank owns: FFM rows starting in [ffm_lo, ffm_hi), LR entries [lr_lo, lr_hi)  (hi == 0xffffffff: to the end)
int fwgpu_dist_ranges(const fwgpu_dist *d, uint32_t *ffm_lo, uint32_t *ffm_hi, uint32_t *lr_lo, uint32_t *lr_hi) {
    if (!d) return fail(FWGPU_ERR_INVALID, "NULL dist");
    if (ffm_lo) *ffm_lo = d->rg.ffm_lo;
    if (ffm_hi) *ffm_hi = d->rg.ffm_hi;
    if (lr_lo) *lr_lo = d->rg.lr_lo;
    if (lr_hi) *lr_hi = d->rg.lr_hi;
    return FWGPU_OK;
}

// One sharded step: this rank's micro-batch of n records in, their n predictions out (host buffer, may be NULL).  Collective:
// every rank of the job calls it with the same n.
static int rccl_step(fwgpu_dist *d, const uint32_t *records, const uint64_t *rec_off, uint32_t n, uint32_t shape[4], float *preds,
                     float *d_preds);
// a step's local preparation has run: on failure the rank still joins the shape exchange, with a poisoned shape (failure model above)
static void begin_result(fwgpu_dist *d, int rc, uint32_t shape[4]) {
    d->begin_rc = rc;
    d->begin_failed = rc != FWGPU_OK;
    if (rc) {
        shape[0] = shape[1] = shape[2] = 0;
        shape[3] = kShapePoison;
    }
}

int fwgpu_dist_learn_sharded(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off,
                             uint32_t n, float *preds) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    uint32_t shape[4];
    begin_result(d, step_begin(d, t, records, rec_off, n, shape), shape);
    return abort_on_failure(d, rccl_step(d, records, rec_off, n, shape, preds, nullptr));
}

// The same step with the rank's micro-batch already in HBM (a record batch of this regressor); the predictions land in the
// batch (fwgpu_batch_predictions).  Nothing crosses PCIe; the call returns when the step is enqueued and complete.
int fwgpu_dist_learn_sharded_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    uint32_t shape[4];
    begin_result(d, step_begin_batch(d, t, b, shape), shape);
    return abort_on_failure(d, rccl_step(d, nullptr, nullptr, b ? b->n : 0, shape, nullptr, b ? b->pred : nullptr));
}

static int rccl_step(fwgpu_dist *d, const uint32_t *records, const uint64_t *rec_off, uint32_t n, uint32_t shape[4], float *preds,
                     float *d_preds) {
    int rc;
    // X1a: batch shapes
    FWGPU_HIP(hipMemcpyAsync(d->d_shape + 4 * d->rank, shape, 4 * sizeof(uint32_t), hipMemcpyHostToDevice, d->stream));
    FWGPU_NCCL(g_rccl.AllGather(d->d_shape + 4 * d->rank, d->d_shape, 4, ncclUint32, d->comm, d->stream));
    std::vector<uint32_t> shapes((size_t)d->n * 4);
    uint32_t *pin_shapes = nullptr;
    if ((rc = d->pinned(shapes.size(), &pin_shapes))) return fail(rc, "dist step: no pinned host memory for the read-back");
    FWGPU_HIP(hipMemcpyAsync(pin_shapes, d->d_shape, shapes.size() * 4, hipMemcpyDeviceToHost, d->stream));
    if ((rc = wait_stream(d))) return rc;
    std::memcpy(shapes.data(), pin_shapes, shapes.size() * 4);
    if ((rc = check_shapes(d, shapes, true))) return rc;
    rc = step_upload(d, records, rec_off, shapes.data());
    if (rc) return rc;
    if (n == 0) return FWGPU_OK;
    fwgpu_batch *gb = d->gb;
    const uint32_t B = d->B;
    const size_t SL = d->sp->split_len;
    // X1b: records and their offsets (in place: every rank's slot of the gathered batch)
    FWGPU_NCCL(g_rccl.AllGather(gb->records + d->wcap * d->rank, gb->records, d->wcap, ncclUint32, d->comm, d->stream));
    FWGPU_NCCL(g_rccl.AllGather(gb->rec_off + (size_t)B * d->rank, gb->rec_off, B, ncclUint64, d->comm, d->stream));
    rc = phase_fwd(d);
    if (rc) return rc;
    // X2: every rank receives the sum of all ranks' partial records of ITS examples
    FWGPU_NCCL(g_rccl.ReduceScatter(d->sp->d_split, d->d_own, (size_t)B * SL, ncclFloat, ncclSum, d->comm, d->stream));
    rc = phase_mid(d);
    if (rc) return rc;
    // X3: gradients and completed records to everyone
    FWGPU_NCCL(g_rccl.AllGather(d->sp->d_g + (size_t)B * d->rank, d->sp->d_g, B, ncclFloat, d->comm, d->stream));
    FWGPU_NCCL(g_rccl.AllGather(d->d_own, d->sp->d_split, (size_t)B * SL, ncclFloat, d->comm, d->stream));
    rc = phase_upd(d);
    if (rc) return rc;
    if (d_preds) FWGPU_HIP(hipMemcpyAsync(d_preds, d->gb->pred + (size_t)B * d->rank, (size_t)B * 4, hipMemcpyDeviceToDevice, d->stream));
    return finish(d, preds);
}

// Every rank's owned range into every rank's tables (before saving the model, or before predicting on one GPU).
int fwgpu_dist_gather_tables(fwgpu_dist *d) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (int rc0 = check_shardable(d)) return rc0;
    if (d->peers_attached) {  // peer-sharded ranks run at their own pace and write into each other's tables: everybody is done first
        if (int rcb = fwgpu_dist_barrier(d)) return rcb;
        FWGPU_HIP(hipSetDevice(d->r->device));  // ... and the owned LR range comes back from its shard allocation
        FWGPU_HIP(hipMemcpy(d->r->d_lr + 2 * d->lr_shard_lo, d->lr_shard, d->lr_shard_n * 8, hipMemcpyDeviceToDevice));
    }
    fwgpu_regressor *r = d->r;
    FWGPU_HIP(hipSetDevice(r->device));
    if (r->cfg.ffm_k) {
        const uint64_t per = (1ull << r->cfg.ffm_bit_precision) / d->n;
        FWGPU_NCCL(g_rccl.AllGather(r->d_ffm_w + per * d->rank, r->d_ffm_w, per, ncclFloat, d->comm, d->stream));
        FWGPU_NCCL(g_rccl.AllGather(r->d_ffm_acc + per * d->rank, r->d_ffm_acc, per, ncclFloat, d->comm, d->stream));
        // the spill-over tail behind 2^ffm_bits belongs to the last rank
        const uint64_t tail = r->ffm_len - (1ull << r->cfg.ffm_bit_precision);
        if (tail && d->n > 1) {
            // (broadcast from the last rank through an all-reduce of a masked copy would need scratch; the tail is R floats:
            // a gather of one block more from the last rank is the cheap route)
            // every rank contributes its tail; the last rank's is kept
            float *tmp = nullptr;
            FWGPU_HIP(hipMalloc((void **)&tmp, tail * 4 * d->n * 2));
            FWGPU_NCCL(g_rccl.AllGather(r->d_ffm_w + (1ull << r->cfg.ffm_bit_precision), tmp, tail, ncclFloat, d->comm, d->stream));
            FWGPU_NCCL(g_rccl.AllGather(r->d_ffm_acc + (1ull << r->cfg.ffm_bit_precision), tmp + tail * d->n, tail, ncclFloat, d->comm, d->stream));
            FWGPU_HIP(hipMemcpyAsync(r->d_ffm_w + (1ull << r->cfg.ffm_bit_precision), tmp + tail * (d->n - 1), tail * 4, hipMemcpyDeviceToDevice, d->stream));
            FWGPU_HIP(hipMemcpyAsync(r->d_ffm_acc + (1ull << r->cfg.ffm_bit_precision), tmp + tail * d->n + tail * (d->n - 1), tail * 4, hipMemcpyDeviceToDevice, d->stream));
            FWGPU_HIP(hipStreamSynchronize(d->stream));
            (void)hipFree(tmp);
        }
    }
    const uint64_t lper = r->lr_len / d->n;  // entries of 2 floats ({w, acc}; SGD keeps the same stride on the device)
    FWGPU_NCCL(g_rccl.AllGather(r->d_lr + 2 * lper * d->rank, r->d_lr, 2 * lper, ncclFloat, d->comm, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));
    return FWGPU_OK;
}

// ------------------------------------------------------------------ peer-sharded hogwild, one PROCESS per rank
// The process-per-rank form of fwgpu_dist_group_learn_peer: every rank exports IPC handles of its three tables
// (hipIpcGetMemHandle), the handles travel through the job's own all-gather, every rank maps the other ranks' tables
// (hipIpcOpenMemHandle: the owner's memory over xGMI, or the same device's memory when several ranks share a GPU) and from then on
// runs the fused hogwild kernel on its own micro-batches with every row reached in its owner's allocation.  No collective per step.
// fwgpu_dist_barrier orders the ranks where the caller needs it (before fwgpu_dist_gather_tables, before a hold-out pass; the tests
// use it to run rank after rank = the sequential reference).
int fwgpu_dist_barrier(fwgpu_dist *d) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    FWGPU_HIP(hipSetDevice(d->r->device));
    // (a peer-sharded step may have been launched on a stream of the caller's: the barrier covers it through the event recorded behind it)
    if (d->ev_peer_step) FWGPU_HIP(hipStreamWaitEvent(d->stream, d->ev_peer_step, 0));
    FWGPU_HIP(hipStreamSynchronize(d->stream));  // this rank's launches are done ...
    FWGPU_HIP(hipMemsetAsync(d->d_shape, 0, 4, d->stream));
    FWGPU_NCCL(g_rccl.AllReduce(d->d_shape, d->d_shape, 1, ncclFloat, ncclSum, d->comm, d->stream));
    return wait_stream(d);  // ... and so are everybody else's
}

int fwgpu_dist_peer_attach(fwgpu_dist *d) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (d->peers_attached) return FWGPU_OK;
    fwgpu_regressor *r = d->r;
    const int N = d->n;
    if (N > 8 || (N & (N - 1))) return fail(FWGPU_ERR_INVALID, "peer-sharded step: 1, 2, 4 or 8 ranks");
    if (r->nn.n_layers) return fail(FWGPU_ERR_INVALID, "peer-sharded step: models with a deep head are not covered");
    int lg = 0;
    while ((1 << lg) < N) lg++;
    if ((r->cfg.ffm_k && (int)r->cfg.ffm_bit_precision < lg) || (int)r->cfg.bit_precision < lg)
        return fail(FWGPU_ERR_INVALID, "peer-sharded step: fewer table entries than ranks");
    // hipIpcOpenMemHandle hangs on an allocation of 2^31 bytes or more on this runtime (measured on the 2 GiB LR table of `-b 28`, which is why the
    // LR table travels as per-rank shards below); an FFM table of that size -- ffm_bit_precision >= 29 -- would meet the same hang in the peers
    if (r->cfg.ffm_k && r->ffm_len * 4ull >= (1ull << 31))
        return fail(FWGPU_ERR_RANGE, "peer-sharded step: FFM tables of 2 GiB or more cannot be mapped by the peers (hipIpcOpenMemHandle); use ffm_bit_precision <= 28");
    FWGPU_HIP(hipSetDevice(r->device));
    struct Handles {
        hipIpcMemHandle_t w, acc, lr;
        uint32_t has_ffm, pad[3];
    };
    static_assert(sizeof(Handles) % 4 == 0, "all-gathered as 32-bit words");
    static const bool dbg = std::getenv("FWGPU_DBG_PEER") != nullptr;
#define PEER_DBG(...) do { if (dbg) { std::fprintf(stderr, "[peer %d] ", d->rank); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)
    Handles mine{};
    mine.has_ffm = r->cfg.ffm_k ? 1u : 0u;
    if (r->cfg.ffm_k) {
        FWGPU_HIP(hipIpcGetMemHandle(&mine.w, r->d_ffm_w));
        FWGPU_HIP(hipIpcGetMemHandle(&mine.acc, r->d_ffm_acc));
    }
    PEER_DBG("ffm handles exported");
    // The LR table's owned range moves into an allocation of its own for as long as the mode is on (copied back by
    // fwgpu_dist_gather_tables): 1/N of the table, so that what peers have to map stays below 2 GiB -- hipIpcOpenMemHandle hangs on an
    // allocation of exactly 2^31 bytes (ROCm 7.x, measured: a `-b 28` LR table), the two 1.07 GB FFM tables open fine.
    d->lr_shard_n = r->lr_len / (uint64_t)N;
    d->lr_shard_lo = d->lr_shard_n * (uint64_t)d->rank;
    FWGPU_HIP(hipMalloc((void **)&d->lr_shard, d->lr_shard_n * 8));
    FWGPU_HIP(hipMemcpy(d->lr_shard, r->d_lr + 2 * d->lr_shard_lo, d->lr_shard_n * 8, hipMemcpyDeviceToDevice));
    FWGPU_HIP(hipIpcGetMemHandle(&mine.lr, d->lr_shard));
    PEER_DBG("lr handle exported (%llu bytes)", (unsigned long long)r->lr_len * 8);
    Handles *d_all = nullptr;
    FWGPU_HIP(hipMalloc((void **)&d_all, sizeof(Handles) * (size_t)N));
    struct FreeOnExit {  // (every early return below releases the staging buffer)
        void *q;
        ~FreeOnExit() { if (q) (void)hipFree(q); }
    } free_d_all{d_all};
    FWGPU_HIP(hipMemcpyAsync(d_all + d->rank, &mine, sizeof(Handles), hipMemcpyHostToDevice, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));
    ncclResult_t e = N > 1 ? g_rccl.AllGather(d_all + d->rank, d_all, sizeof(Handles) / 4, ncclUint32, d->comm, d->stream) : ncclSuccess;
    std::vector<Handles> all((size_t)N);
    PEER_DBG("handles all-gathered");
    hipError_t he = hipStreamSynchronize(d->stream);
    if (he == hipSuccess) he = hipMemcpy(all.data(), d_all, sizeof(Handles) * (size_t)N, hipMemcpyDeviceToHost);
    if (e != ncclSuccess) return fail(FWGPU_ERR_DEVICE, "peer attach: all-gather of the IPC handles failed");
    if (he != hipSuccess) return fail(FWGPU_ERR_DEVICE, std::string("peer attach: ") + hipGetErrorString(he));
    PeerShards ps{};
    ps.n = (uint32_t)N;
    ps.shift_ffm = r->cfg.ffm_k ? r->cfg.ffm_bit_precision - lg : 31;
    ps.shift_lr = r->cfg.bit_precision - lg;
    for (int j = 0; j < N; j++) {
        // (an owner's LR base is its shard allocation moved back by its range start: the kernel indexes it like the whole table, and the
        // owner lookup only ever sends entries of that range there)
        if (j == d->rank) {
            ps.ffm_w[j] = r->d_ffm_w;
            ps.ffm_acc[j] = r->d_ffm_acc;
            ps.lr[j] = d->lr_shard - 2 * d->lr_shard_lo;
            continue;
        }
        auto open = [&](const hipIpcMemHandle_t &h, float **out) -> int {
            void *q = nullptr;
            PEER_DBG("opening a handle of rank %d ...", j);
            FWGPU_HIP(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
            PEER_DBG("... opened at %p", q);
            d->ipc_open.push_back(q);
            *out = static_cast<float *>(q);
            return FWGPU_OK;
        };
        int rc;
        if (all[j].has_ffm && ((rc = open(all[j].w, &ps.ffm_w[j])) || (rc = open(all[j].acc, &ps.ffm_acc[j])))) return rc;
        if ((rc = open(all[j].lr, &ps.lr[j]))) return rc;
        ps.lr[j] -= 2 * d->lr_shard_n * (uint64_t)j;
    }
    if (!d->d_peers) FWGPU_HIP(hipMalloc((void **)&d->d_peers, sizeof(PeerShards)));
    FWGPU_HIP(hipMemcpy(d->d_peers, &ps, sizeof(PeerShards), hipMemcpyHostToDevice));
    d->peers_attached = true;
    return fwgpu_dist_barrier(d);  // nobody launches before everybody has mapped everybody
}

// One peer-sharded step of THIS rank: its n records through the fused kernel (mode = fwgpu_dist_set_mode), rows reached in their owners'
// tables.  Not a collective: ranks run at their own pace (hogwild across GPUs).  predictions: host buffer, may be NULL.
int fwgpu_dist_learn_peer(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                          float *preds, int update) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (!d->peers_attached) return fail(FWGPU_ERR_INVALID, "fwgpu_dist_peer_attach first");
    uint32_t shape[4];
    int rc = sparse_begin(d, t, records, rec_off, n, nullptr, shape);  // (uploads the rank's records into its own batch)
    if (rc) return rc;
    FWGPU_HIP(hipSetDevice(d->r->device));
    if ((rc = run_batch_peer(d->r, d->cur, d->mode, update, d->d_peers, d->stream))) return rc;
    if (preds && d->B) FWGPU_HIP(hipMemcpyAsync(preds, d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));
    return FWGPU_OK;
}

// the same step with the rank's micro-batch already resident in HBM (a record batch of this rank's regressor); predictions land in the
// batch; asynchronous on the rank's stream unless `hip_stream` is given (then launched there)
int fwgpu_dist_learn_peer_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b, int update, void *hip_stream) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (!d->peers_attached) return fail(FWGPU_ERR_INVALID, "fwgpu_dist_peer_attach first");
    if (!t || !b || !b->records) return fail(FWGPU_ERR_INVALID, "peer step: a record batch is needed");
    if (b->owner != d->r) return fail(FWGPU_ERR_INVALID, "peer step: batch belongs to another regressor");
    FWGPU_HIP(hipSetDevice(d->r->device));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : d->stream;
    int rc = run_batch_peer(d->r, b, d->mode, update, d->d_peers, st);
    if (rc) return rc;
    if (st != d->stream) {  // fwgpu_dist_barrier (and through it gather_tables) must cover this launch too
        if (!d->ev_peer_step) FWGPU_HIP(hipEventCreateWithFlags(&d->ev_peer_step, hipEventDisableTiming));
        FWGPU_HIP(hipEventRecord(d->ev_peer_step, st));
    }
    return FWGPU_OK;
}

// Dense all-reduce of a device float buffer over the job (replica mode: table deltas; deep head: dense gradient sums)
int fwgpu_dist_all_reduce_sum(fwgpu_dist *d, float *device_buf, uint64_t count, void *hip_stream) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : d->stream;
    FWGPU_NCCL(g_rccl.AllReduce(device_buf, device_buf, count, ncclFloat, ncclSum, d->comm, s));
    if (!hip_stream) FWGPU_HIP(hipStreamSynchronize(s));
    return FWGPU_OK;
}

// ------------------------------------------------------------------ row-sparse gradient buckets over RCCL
// One step: this rank's micro-batch (any size, ranks may differ) is scored against the replica as it is; the deduplicated row
// gradients of all ranks are all-gathered and every rank applies all of them in the same order -> replicas stay bit-identical.
static int rccl_sparse_step(fwgpu_dist *d, uint32_t shape[4], float *preds) {
    int rc;
    const int N = d->n;
    FWGPU_HIP(hipMemcpyAsync(d->d_shape + 4 * d->rank, shape, 4 * sizeof(uint32_t), hipMemcpyHostToDevice, d->stream));
    if (N > 1) FWGPU_NCCL(g_rccl.AllGather(d->d_shape + 4 * d->rank, d->d_shape, 4, ncclUint32, d->comm, d->stream));
    std::vector<uint32_t> shapes((size_t)N * 4);
    uint32_t *pin_shapes = nullptr;
    if ((rc = d->pinned(shapes.size(), &pin_shapes))) return fail(rc, "dist step: no pinned host memory for the read-back");
    FWGPU_HIP(hipMemcpyAsync(pin_shapes, d->d_shape, shapes.size() * 4, hipMemcpyDeviceToHost, d->stream));
    if ((rc = wait_stream(d))) return rc;
    std::memcpy(shapes.data(), pin_shapes, shapes.size() * 4);
    if ((rc = check_shapes(d, shapes, false))) return rc;
    rc = sparse_local(d, shapes.data());
    if (rc) return rc;
    // bucket-row counts of every rank
    uint32_t *all_counts = d->d_counts + 2;
    if (N > 1)
        FWGPU_NCCL(g_rccl.AllGather(d->d_counts, all_counts, 2, ncclUint32, d->comm, d->stream));
    else
        FWGPU_HIP(hipMemcpyAsync(all_counts, d->d_counts, 8, hipMemcpyDeviceToDevice, d->stream));
    std::vector<uint32_t> counts((size_t)2 * N);
    uint32_t *pin_counts = nullptr;
    if ((rc = d->pinned(counts.size(), &pin_counts))) return fail(rc, "dist step: no pinned host memory for the read-back");
    FWGPU_HIP(hipMemcpyAsync(pin_counts, all_counts, counts.size() * 4, hipMemcpyDeviceToHost, d->stream));
    if ((rc = wait_stream(d))) return rc;
    std::memcpy(counts.data(), pin_counts, counts.size() * 4);
    d->last_rows[0] = counts[2 * d->rank];
    d->last_rows[1] = counts[2 * d->rank + 1];
    for (int side = 0; side < 2; side++) {
        fwgpu_dist::SparseSide &s = side == 0 ? d->sf : d->sl;
        if (!s.occ_cap) continue;
        uint32_t stride = 0;
        for (int j = 0; j < N; j++) stride = std::max(stride, counts[2 * j + side]);
        if (!stride) continue;
        if (stride > s.occ_cap) return fail(FWGPU_ERR_RANGE, "sparse step: a rank sent more bucket rows than the buffers hold");
        // per-rank counts of this side, contiguous on the device: reuse the flags array (free after the local reduction)
        std::vector<uint32_t> side_counts(N);
        for (int j = 0; j < N; j++) side_counts[j] = counts[2 * j + side];
        FWGPU_HIP(hipMemcpyAsync(s.flags, side_counts.data(), (size_t)N * 4, hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));  // side_counts is a local
        const uint32_t *keys = s.bk_key;
        const float *rows = s.bk_rows;
        if (N > 1) {
            rc = sparse_side_reserve_all(s, (uint64_t)stride * N);
            if (rc) return rc;
            FWGPU_NCCL(g_rccl.AllGather(s.bk_key, s.all_key, stride, ncclUint32, d->comm, d->stream));
            FWGPU_NCCL(g_rccl.AllGather(s.bk_rows, s.all_rows, (size_t)stride * s.width, ncclFloat, d->comm, d->stream));
            keys = s.all_key;
            rows = s.all_rows;
        } else {
            rc = sparse_side_reserve_all(s, stride);  // (the merged key lists only; keys and rows are used in place)
            if (rc) return rc;
        }
        rc = sparse_apply_side(d, s, side == 0, keys, rows, s.flags, (uint32_t)N, stride);
        if (rc) return rc;
    }
    if ((rc = wait_stream(d))) return rc;  // (polled; the caller's buffer is pageable: its copy comes after, not in front of the wait)
    if (preds && d->B) FWGPU_HIP(hipMemcpy(preds, d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost));
    return FWGPU_OK;
}

int fwgpu_dist_learn_sparse(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off,
                            uint32_t n, float *preds) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    uint32_t shape[4];
    begin_result(d, sparse_begin(d, t, records, rec_off, n, nullptr, shape), shape);
    return abort_on_failure(d, rccl_sparse_step(d, shape, preds));
}

// what the rank's last sparse step put on the wire: bucket rows of R floats (+ 4-byte key) and LR buckets (key + float)
int fwgpu_dist_sparse_last_rows(const fwgpu_dist *d, uint32_t *ffm_rows, uint32_t *lr_rows) {
    if (!d) return fail(FWGPU_ERR_INVALID, "NULL dist");
    if (ffm_rows) *ffm_rows = d->last_rows[0];
    if (lr_rows) *lr_rows = d->last_rows[1];
    return FWGPU_OK;
}

// the rank's micro-batch already in HBM (fwgpu_record_batch_create); predictions land in the batch
int fwgpu_dist_learn_sparse_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    uint32_t shape[4];
    begin_result(d, sparse_begin(d, t, nullptr, nullptr, 0, b, shape), shape);
    return abort_on_failure(d, rccl_sparse_step(d, shape, nullptr));
}

// ------------------------------------------------------------------ in-process group: the same step, collectives by copies
// n regressors (one per rank; usually all on one device, which is how the one-GPU box emulates N GPUs and how the tests run).
int fwgpu_dist_group_create(fwgpu_regressor *const *regs, int n, fwgpu_dist_group **out) {
    if (!regs || !out || n < 1) return fail(FWGPU_ERR_INVALID, "dist group: bad argument");
    std::unique_ptr<fwgpu_dist_group> g(new fwgpu_dist_group());
    for (int i = 0; i < n; i++) {
        fwgpu_dist *d = nullptr;
        int rc = make_rank(regs[i], i, n, &d);
        if (rc) return rc;
        g->ranks.emplace_back(d);
    }
    *out = g.release();
    return FWGPU_OK;
}

int fwgpu_dist_group_set_mode(fwgpu_dist_group *g, int mode) {
    if (!g) return fail(FWGPU_ERR_INVALID, "NULL group");
    for (auto &d : g->ranks) {
        int rc = fwgpu_dist_set_mode(d.get(), mode);
        if (rc) return rc;
    }
    return FWGPU_OK;
}

int fwgpu_dist_group_free(fwgpu_dist_group *g) {
    delete g;
    return FWGPU_OK;
}

int fwgpu_dist_group_learn_sharded(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                   const uint64_t *const *rec_off, uint32_t n, float *const *preds) {
    if (!g || !records || !rec_off) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const int N = (int)g->ranks.size();
    std::vector<uint32_t> shapes((size_t)N * 4);
    for (int j = 0; j < N; j++) {
        int rc = step_begin(g->ranks[j].get(), t, records[j], rec_off[j], n, &shapes[4 * j]);
        if (rc) return rc;
    }
    for (int j = 0; j < N; j++) {
        int rc = step_upload(g->ranks[j].get(), records[j], rec_off[j], shapes.data());
        if (rc) return rc;
    }
    if (n == 0) return FWGPU_OK;
    const uint32_t B = n;
    const size_t SL = g->ranks[0]->sp->split_len;
    auto sync_all = [&]() -> int {
        for (auto &d : g->ranks) FWGPU_HIP(hipStreamSynchronize(d->stream));
        return FWGPU_OK;
    };
    // X1: all-gather of records and offsets = copy slot j of rank j into slot j of everyone else
    for (int j = 0; j < N; j++) {
        fwgpu_dist *src = g->ranks[j].get();
        for (int i = 0; i < N; i++) {
            if (i == j) continue;
            fwgpu_dist *dst = g->ranks[i].get();
            FWGPU_HIP(hipMemcpyAsync(dst->gb->records + dst->wcap * j, src->gb->records + src->wcap * j, src->wcap * 4, hipMemcpyDeviceToDevice, dst->stream));
            FWGPU_HIP(hipMemcpyAsync(dst->gb->rec_off + (size_t)B * j, src->gb->rec_off + (size_t)B * j, (size_t)B * 8, hipMemcpyDeviceToDevice, dst->stream));
        }
    }
    // The ranks' phase kernels run ONE AFTER THE OTHER on the device (rank j's stream waits for an event behind rank j-1's phase): phase kernels of
    // two hardware queues that overlap are not reproducible on this build (see fwgpu_dist_group_learn_sparse below).
    // FWGPU_GROUP_CONCURRENT=local removes the ordering (debug).
    static const char *cc_env = std::getenv("FWGPU_GROUP_CONCURRENT");
    const bool ordered = !(cc_env && (cc_env[0] == 'l' || (cc_env[0] == 'a' && cc_env[1] == 'l')));
    auto chain = [&](int j) -> int {
        if (!ordered || j == 0) return FWGPU_OK;
        fwgpu_dist *dj = g->ranks[j].get();
        if (!dj->ev_prev) FWGPU_HIP(hipEventCreateWithFlags(&dj->ev_prev, hipEventDisableTiming));
        FWGPU_HIP(hipEventRecord(dj->ev_prev, g->ranks[j - 1]->stream));
        FWGPU_HIP(hipStreamWaitEvent(dj->stream, dj->ev_prev, 0));
        return FWGPU_OK;
    };
    int rc = sync_all();
    if (rc) return rc;
    for (int j = 0; j < N; j++) {
        if ((rc = chain(j))) return rc;
        rc = phase_fwd(g->ranks[j].get());
        if (rc) return rc;
    }
    rc = sync_all();
    if (rc) return rc;
    // X2: reduce-scatter: rank i's own records = sum over ranks j (in rank order) of j's partial records of i's examples
    for (int i = 0; i < N; i++) {
        fwgpu_dist *dst = g->ranks[i].get();
        for (int j = 0; j < N; j++) {
            const float *part = g->ranks[j]->sp->d_split + (size_t)B * i * SL;
            if (j == 0)
                FWGPU_HIP(hipMemcpyAsync(dst->d_own, part, (size_t)B * SL * 4, hipMemcpyDeviceToDevice, dst->stream));
            else
                FWGPU_HIP(launch_add(dst->d_own, part, (uint64_t)B * SL, dst->stream));
        }
    }
    rc = sync_all();
    if (rc) return rc;
    for (int j = 0; j < N; j++) {
        if ((rc = chain(j))) return rc;
        rc = phase_mid(g->ranks[j].get());
        if (rc) return rc;
    }
    rc = sync_all();
    if (rc) return rc;
    // X3: all-gather of gradients and completed records
    for (int j = 0; j < N; j++) {
        fwgpu_dist *src = g->ranks[j].get();
        for (int i = 0; i < N; i++) {
            fwgpu_dist *dst = g->ranks[i].get();
            if (i != j)
                FWGPU_HIP(hipMemcpyAsync(dst->sp->d_g + (size_t)B * j, src->sp->d_g + (size_t)B * j, (size_t)B * 4, hipMemcpyDeviceToDevice, dst->stream));
            FWGPU_HIP(hipMemcpyAsync(dst->sp->d_split + (size_t)B * j * SL, src->d_own, (size_t)B * SL * 4, hipMemcpyDeviceToDevice, dst->stream));
        }
    }
    rc = sync_all();
    if (rc) return rc;
    for (int j = 0; j < N; j++) {
        if ((rc = chain(j))) return rc;
        rc = phase_upd(g->ranks[j].get());
        if (rc) return rc;
    }
    for (int j = 0; j < N; j++) {
        rc = finish(g->ranks[j].get(), preds ? preds[j] : nullptr);
        if (rc) return rc;
    }
    return FWGPU_OK;
}

// Row-sparse step of an in-process group: records[j] / rec_off[j] / n[j] are rank j's micro-batch
int fwgpu_dist_group_learn_sparse(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                  const uint64_t *const *rec_off, const uint32_t *n, float *const *preds) {
    if (!g || !records || !rec_off || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const int N = (int)g->ranks.size();
    std::vector<uint32_t> shapes((size_t)N * 4);
    int rc;
    for (int j = 0; j < N; j++)
        if ((rc = sparse_begin(g->ranks[j].get(), t, records[j], rec_off[j], n[j], nullptr, &shapes[4 * j]))) return rc;
    // The ranks' local phases run ONE AFTER THE OTHER ON THE DEVICE: rank j's stream waits (event, no host synchronisation) for rank
    // j-1's local phase.  With the FWD (+ MID) kernels of two ranks overlapping on different hardware queues the step is not reproducible
    // (scripts/group_repro.py, profiles/r03_group_concurrency.txt): single WORKGROUPS of a phase kernel come up with wrong
    // loop-invariant state -- their examples read wrong, occasionally an address built from it faults.  It is not this library's memory
    // accesses (buffers disjoint, keys / scans valid, LDS canary untouched, arguments intact) and not the size of the argument block, as a
    // first pass concluded: it follows the register allocation the KernelParams layout happens to select, and it goes away -- 34 of 34 runs
    // on every failing layout -- when the phase kernels keep their spilled scalar registers in scratch memory instead of VGPR lanes
    // (`make PHASE_SGPR_SPILLS=scratch`, 0.55x the phase kernels' speed).  The default build keeps the fast spills and this ordering, which
    // is exact on every build.  FWGPU_GROUP_CONCURRENT=local removes the ordering (debug), FWGPU_DBG_GROUP_CHAIN = fwd | mid | red moves the
    // wait to an earlier point of rank j-1's phase.  The RCCL path has one rank per process and one queue.
    static const char *dbg_chain = std::getenv("FWGPU_DBG_GROUP_CHAIN");
    static const char *cc_env = std::getenv("FWGPU_GROUP_CONCURRENT");
    const bool unordered = cc_env && (cc_env[0] == 'l' || (cc_env[0] == 'a' && cc_env[1] == 'l'));
    const int chain_at = dbg_chain ? (dbg_chain[0] == 'f' ? 0 : dbg_chain[0] == 'm' ? 1 : dbg_chain[0] == 'r' ? 2 : 3) : (unordered ? -1 : 3);
    for (int j = 0; j < N; j++) {
        fwgpu_dist *dj = g->ranks[j].get();
        if (chain_at >= 0 && chain_at < 3)
            for (int e = 0; e < 3; e++)
                if (!dj->dbg_ev[e]) FWGPU_HIP(hipEventCreateWithFlags(&dj->dbg_ev[e], hipEventDisableTiming));
        if (chain_at == 3 && j > 0) {
            if (!dj->ev_prev) FWGPU_HIP(hipEventCreateWithFlags(&dj->ev_prev, hipEventDisableTiming));
            FWGPU_HIP(hipEventRecord(dj->ev_prev, g->ranks[j - 1]->stream));
            FWGPU_HIP(hipStreamWaitEvent(dj->stream, dj->ev_prev, 0));
        } else if (chain_at >= 0 && j > 0) {
            FWGPU_HIP(hipStreamWaitEvent(dj->stream, g->ranks[j - 1]->dbg_ev[chain_at], 0));
        }
        if ((rc = sparse_local(dj, shapes.data()))) return rc;
    }
    std::vector<uint32_t> counts((size_t)2 * N);
    for (int j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        FWGPU_HIP(hipMemcpyAsync(&counts[2 * j], d->d_counts, 8, hipMemcpyDeviceToHost, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
        d->last_rows[0] = counts[2 * j];
        d->last_rows[1] = counts[2 * j + 1];
    }
    {
        // debug (scripts/group_bisect.sh): what did every rank's local phase leave behind?  FWD's occurrence keys must be
        // (hash << 32 | slot) with slot == their index (or ~0), the sorted keys ascending, pos non-decreasing and <= the bucket count
        static const bool dbg_verify = std::getenv("FWGPU_DBG_VERIFY") != nullptr;
        if (dbg_verify)
            for (int j = 0; j < N; j++) {
                fwgpu_dist *d = g->ranks[j].get();
                for (int side = 0; side < 2; side++) {
                    fwgpu_dist::SparseSide &sd = side == 0 ? d->sf : d->sl;
                    const uint32_t me = side == 0 ? d->occ_max_ffm : d->occ_max_lr;
                    const size_t nk = (size_t)d->B * me;
                    if (!sd.occ_cap || !nk) continue;
                    std::vector<unsigned long long> k(nk), ks(nk);
                    std::vector<uint32_t> pos(nk + 1);
                    FWGPU_HIP(hipMemcpy(k.data(), sd.key, nk * 8, hipMemcpyDeviceToHost));
                    FWGPU_HIP(hipMemcpy(ks.data(), sd.key_sorted, nk * 8, hipMemcpyDeviceToHost));
                    FWGPU_HIP(hipMemcpy(pos.data(), sd.pos, (nk + 1) * 4, hipMemcpyDeviceToHost));
                    size_t bad_key = 0, bad_sorted = 0, bad_pos = 0, valid = 0, valid_sorted = 0;
                    const uint64_t tab = side == 0 ? d->r->ffm_len : d->r->lr_len;
                    for (size_t i = 0; i < nk; i++) {
                        if (k[i] == ~0ull) continue;
                        valid++;
                        if ((uint32_t)k[i] != (uint32_t)i || (k[i] >> 32) >= tab) bad_key++;
                    }
                    for (size_t i = 0; i < nk; i++) {
                        if (ks[i] != ~0ull) valid_sorted++;
                        if (i && ks[i] < ks[i - 1]) bad_sorted++;
                    }
                    for (size_t i = 1; i <= nk; i++)
                        if (pos[i] < pos[i - 1] || pos[i] > counts[2 * j + side]) bad_pos++;
                    std::fprintf(stderr, "[verify] rank %d side %d keys %zu valid %zu bad %zu | sorted valid %zu out-of-order %zu | pos bad %zu count %u\n", j, side,
                                 nk, valid, bad_key, valid_sorted, bad_sorted, bad_pos, counts[2 * j + side]);
                }
                // order-independent checksums of what each stage produced: FWD (split records, own slots), MID (gradients), reduction (bucket rows)
                unsigned long long *d_cs = nullptr, cs[5] = {0, 0, 0, 0, 0};
                FWGPU_HIP(hipMalloc((void **)&d_cs, 5 * 8));
                FWGPU_HIP(hipMemset(d_cs, 0, 5 * 8));
                FWGPU_HIP(launch_checksum(d->sp->d_split, (uint64_t)d->B * d->sp->split_len, d_cs + 0, d->stream));
                FWGPU_HIP(launch_checksum(d->sp->d_g, d->B, d_cs + 1, d->stream));
                if (d->sf.occ_cap) FWGPU_HIP(launch_checksum(d->sf.bk_rows, (uint64_t)counts[2 * j] * d->sf.width, d_cs + 2, d->stream));
                if (d->sl.occ_cap) FWGPU_HIP(launch_checksum(d->sl.bk_rows, counts[2 * j + 1], d_cs + 3, d->stream));
                FWGPU_HIP(launch_checksum(d->cur->pred, d->B, d_cs + 4, d->stream));
                FWGPU_HIP(hipStreamSynchronize(d->stream));
                FWGPU_HIP(hipMemcpy(cs, d_cs, 5 * 8, hipMemcpyDeviceToHost));
                (void)hipFree(d_cs);
                if (j == 0) std::fprintf(stderr, "[canary] changed LDS canary words so far: %u\n", dbg_canary_read());
                std::fprintf(stderr, "[stage] rank %d split %016llx g %016llx ffm_rows %016llx lr_rows %016llx pred %016llx\n", j, cs[0], cs[1], cs[2], cs[3], cs[4]);
            }
    }
    for (int side = 0; side < 2; side++) {
        uint32_t stride = 0;
        for (int j = 0; j < N; j++) stride = std::max(stride, counts[2 * j + side]);
        if (!stride) continue;
        std::vector<uint32_t> side_counts(N);
        for (int j = 0; j < N; j++) side_counts[j] = counts[2 * j + side];
        for (int i = 0; i < N; i++) {
            fwgpu_dist *dst = g->ranks[i].get();
            fwgpu_dist::SparseSide &s = side == 0 ? dst->sf : dst->sl;
            if (!s.occ_cap) continue;
            if ((rc = sparse_side_reserve_all(s, (uint64_t)stride * N))) return rc;
            FWGPU_HIP(hipMemcpyAsync(s.flags, side_counts.data(), (size_t)N * 4, hipMemcpyHostToDevice, dst->stream));
            for (int j = 0; j < N; j++) {  // all-gather = rank j's bucket into slot j of everyone
                fwgpu_dist *src = g->ranks[j].get();
                const fwgpu_dist::SparseSide &q = side == 0 ? src->sf : src->sl;
                const uint32_t c = side_counts[j];
                if (!c) continue;
                FWGPU_HIP(hipMemcpyAsync(s.all_key + (size_t)stride * j, q.bk_key, (size_t)c * 4, hipMemcpyDeviceToDevice, dst->stream));
                FWGPU_HIP(hipMemcpyAsync(s.all_rows + (size_t)stride * j * s.width, q.bk_rows, (size_t)c * s.width * 4, hipMemcpyDeviceToDevice, dst->stream));
            }
            FWGPU_HIP(hipStreamSynchronize(dst->stream));  // side_counts is a local
        }
        for (int i = 0; i < N; i++) {
            fwgpu_dist *dst = g->ranks[i].get();
            fwgpu_dist::SparseSide &s = side == 0 ? dst->sf : dst->sl;
            if (!s.occ_cap) continue;
            if ((rc = sparse_apply_side(dst, s, side == 0, s.all_key, s.all_rows, s.flags, (uint32_t)N, stride))) return rc;
            // the ranks of an in-process group share one GPU: their phases run one after the other (the group exists for tests and
            // single-box emulation; with concurrent rank streams the step was observed to be non-reproducible on some boxes)
            {
                const char *cc = std::getenv("FWGPU_GROUP_CONCURRENT");
                if (!(cc && cc[0] == 'a')) FWGPU_HIP(hipStreamSynchronize(dst->stream));
            }
        }
    }
    for (int j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        if (preds && preds[j] && d->B) FWGPU_HIP(hipMemcpyAsync(preds[j], d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
    }
    return FWGPU_OK;
}

// Peer-sharded hogwild step (SURVEY 8e "owner-sharded tables", without the synchronous batch): the tables are sharded by owner --
// rank s's allocation is the authoritative copy of the FFM rows that START in [s, s+1) * 2^ffm_bits / N and of the LR entries in
// the same share of 2^bits -- and every rank runs the FUSED hogwild kernel on its own micro-batch, reaching each row in its owner's
// memory: a plain pointer when the owner lives on the same device (the one-GPU emulation, the tests), a peer-mapped pointer over
// xGMI when it lives on another GPU of the process (hipDeviceEnablePeerAccess below).  No collective, no barrier between the
// ranks: this is hogwild.rs:89-103 with GPUs in the place of threads and xGMI in the place of the cache-coherent bus -- the
// reference's update rule per occurrence, staleness = the examples in flight on all GPUs.
// What crosses a link per example at config C with N GPUs: (N-1)/N of the row traffic, 200 rows x 960 B x (1 read in the gather +
// read w, read acc, write w, write acc in the update) = 0.96 MB x 7/8 = 0.84 MB at N = 8, spread over 7 links: 0.12 MB per link
// and example in each direction pair -> ~1.2 M examples/s per GPU at 153 GB/s per link: a CAPACITY mode with hogwild semantics
// (no batch-size limit, no divergence cliff), slower than replicas (0.26 KB per example) by the same arithmetic that makes the
// sparse mode slow.  DESIGN.md 7 has the table.
// FWGPU_MODE_SEQUENTIAL: rank after rank, each on one workgroup, in example order: the deterministic form the tests compare with
// the oracle (the sequential reference algorithm over the ranks' micro-batches in rank order).
// Rows that start below an ownership boundary and reach across it live wholly in their owner's copy (the documented deviation
// of the sharded mode: 6e-6 of the rows at config C with 8 ranks).
int fwgpu_dist_group_learn_peer(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                const uint64_t *const *rec_off, const uint32_t *n, float *const *preds, int update) {
    if (!g || !t || !records || !rec_off || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const int N = (int)g->ranks.size();
    if (N > 8 || (N & (N - 1))) return fail(FWGPU_ERR_INVALID, "peer-sharded step: 1, 2, 4 or 8 ranks");
    fwgpu_regressor *r0 = g->ranks[0]->r;
    if (r0->nn.n_layers) return fail(FWGPU_ERR_INVALID, "peer-sharded step: models with a deep head are not covered");
    int lg = 0;
    while ((1 << lg) < N) lg++;
    if ((r0->cfg.ffm_k && (int)r0->cfg.ffm_bit_precision < lg) || (int)r0->cfg.bit_precision < lg)
        return fail(FWGPU_ERR_INVALID, "peer-sharded step: fewer table entries than ranks");
    PeerShards ps{};
    ps.n = (uint32_t)N;
    ps.shift_ffm = r0->cfg.ffm_k ? r0->cfg.ffm_bit_precision - lg : 31;
    ps.shift_lr = r0->cfg.bit_precision - lg;
    for (int j = 0; j < N; j++) {
        fwgpu_regressor *r = g->ranks[j]->r;
        ps.ffm_w[j] = r->d_ffm_w;
        ps.ffm_acc[j] = r->d_ffm_acc;
        ps.lr[j] = r->d_lr;
    }
    // owners on other devices: map their memory (xGMI peer access; a no-op for same-device groups)
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            const int di = g->ranks[i]->r->device, dj = g->ranks[j]->r->device;
            if (di == dj) continue;
            FWGPU_HIP(hipSetDevice(di));
            hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(FWGPU_ERR_DEVICE, "hipDeviceEnablePeerAccess failed");
            (void)hipGetLastError();
        }
    uint32_t shape[4];
    int rc;
    const int mode = g->ranks[0]->mode;
    for (int j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        if ((rc = sparse_begin(d, t, records[j], rec_off[j], n[j], nullptr, shape))) return rc;  // (uploads the rank's records into its own batch)
        FWGPU_HIP(hipSetDevice(d->r->device));
        if (!d->d_peers) FWGPU_HIP(hipMalloc((void **)&d->d_peers, sizeof(PeerShards)));
        FWGPU_HIP(hipMemcpyAsync(d->d_peers, &ps, sizeof(PeerShards), hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));  // (ps is a local)
        if ((rc = run_batch_peer(d->r, d->cur, mode, update, d->d_peers, d->stream))) return rc;
        if (mode == FWGPU_MODE_SEQUENTIAL) FWGPU_HIP(hipStreamSynchronize(d->stream));  // rank after rank
    }
    for (int j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        FWGPU_HIP(hipSetDevice(d->r->device));
        if (preds && preds[j] && d->B) FWGPU_HIP(hipMemcpyAsync(preds[j], d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
    }
    return FWGPU_OK;
}

// every rank's owned range into every rank's tables
// ------------------------------------------------------------------ owner-side apply: sharded hogwild without a read-modify-write on the links
// SURVEY 8e's decomposition, the step VERDICT r03 asked for.  Tables are sharded by owner as in the peer-sharded mode.  A rank's fused kernel
// FETCHES the weight rows of its examples from their owners (peer loads), computes prediction and general gradient, and -- instead of stepping
// the rows in the owner's memory (five row passes over the link: fwgpu_dist_*_peer) -- PUSHES one gradient row per occurrence into a ring in
// the owner's memory.  The owner then runs the optimizer on its own tables (owner_apply_kernel): accumulators never cross a link, no
// read-modify-write spans two GPUs, and what an example puts on the links falls from 0.84 MB to 0.34 MB at config C (fetch w + push g).
// Ring layout in owner o's allocation, for step parity q and source s (capacities in rows / LR entries):
//   keys  [2][N][cap_ffm] u32 | rows [2][N][cap_ffm * R] f32 | lr [2][N][cap_lr] {hash, gradient}
// Positions come from counters in the SOURCE's memory (no atomic ever crosses a link); the counts travel with the step's one collective
// (process-per-rank form: an all-gather that also tells every owner that all sources' kernels are done).  Two parities: a source may push step
// t + 2 only after its all-gather of step t + 1 has completed, which every owner joins after its apply of step t -- no further barrier.
// Semantics: per-occurrence AdaGrad with gradients taken from the weights the example's forward pass read (hogwild.rs's staleness: one step);
// in order -- one example per step, ranks taking turns -- it IS the sequential reference (tests/test_gpu_dist.py, test_gpu_dist_procs.py).
namespace {
struct RingGeom {
    uint32_t N, R, cap_ffm, cap_lr;
    size_t off_rows, off_lr, bytes;
};
RingGeom ring_geom(uint32_t N, uint32_t R, uint32_t cap_ffm, uint32_t cap_lr) {
    RingGeom g{N, R, cap_ffm, cap_lr, 0, 0, 0};
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    g.off_rows = up((size_t)2 * N * cap_ffm * 4);
    g.off_lr = g.off_rows + up((size_t)2 * N * cap_ffm * R * 4);
    g.bytes = g.off_lr + up((size_t)2 * N * cap_lr * 8);
    return g;
}
uint32_t *ring_keys(unsigned char *base, const RingGeom &g, uint32_t q, uint32_t s) { return reinterpret_cast<uint32_t *>(base) + ((size_t)q * g.N + s) * g.cap_ffm; }
float *ring_rows(unsigned char *base, const RingGeom &g, uint32_t q, uint32_t s) { return reinterpret_cast<float *>(base + g.off_rows) + ((size_t)q * g.N + s) * g.cap_ffm * g.R; }
uint2 *ring_lr(unsigned char *base, const RingGeom &g, uint32_t q, uint32_t s) { return reinterpret_cast<uint2 *>(base + g.off_lr) + ((size_t)q * g.N + s) * g.cap_lr; }

int rings_reserve(fwgpu_dist *d, uint32_t N, uint32_t cap_ffm, uint32_t cap_lr) {
    const uint32_t R = d->r->cfg.ffm_k ? d->r->cfg.ffm_k * d->r->cfg.ffm_num_fields : 0;
    FWGPU_HIP(hipSetDevice(d->r->device));
    if (!d->d_push_cnt) {
        FWGPU_HIP(hipMalloc((void **)&d->d_push_cnt, (size_t)(N + 1) * (2 * N + 1) * 4));
        FWGPU_HIP(hipMalloc((void **)&d->d_push, sizeof(PushRings)));
    }
    if (d->own_rings && d->ring_cap_ffm >= cap_ffm && d->ring_cap_lr >= cap_lr) return FWGPU_OK;
    if (d->rings_attached) return fail(FWGPU_ERR_RANGE, "owner-side apply: the step is larger than the rings fwgpu_dist_owner_attach sized");
    if (d->own_rings) (void)hipFree(d->own_rings);
    d->own_rings = nullptr;
    d->ring_cap_ffm = std::max(cap_ffm, d->ring_cap_ffm);
    d->ring_cap_lr = std::max(cap_lr, d->ring_cap_lr);
    const RingGeom g = ring_geom(N, R, d->ring_cap_ffm, d->ring_cap_lr);
    FWGPU_HIP(hipMalloc((void **)&d->own_rings, std::max<size_t>(g.bytes, 256)));
    d->own_rings_bytes = g.bytes;
    return FWGPU_OK;
}

// the descriptor source `d` (rank s of N) hands to its kernel: every owner's region for (parity, s), as this rank reaches it
int push_descriptor(fwgpu_dist *d, uint32_t N, uint32_t s, unsigned char *const owner_base[8], uint32_t cap_ffm, uint32_t cap_lr) {
    const uint32_t R = d->r->cfg.ffm_k ? d->r->cfg.ffm_k * d->r->cfg.ffm_num_fields : 0;
    const RingGeom g = ring_geom(N, R, cap_ffm, cap_lr);
    PushRings pr{};
    pr.n = N;
    pr.cap_ffm = cap_ffm;
    pr.cap_lr = cap_lr;
    pr.cnt = d->d_push_cnt;
    for (uint32_t o = 0; o < N; o++) {
        pr.ffm_key[o] = ring_keys(owner_base[o], g, d->ring_parity, s);
        pr.ffm_rows[o] = ring_rows(owner_base[o], g, d->ring_parity, s);
        pr.lr_ent[o] = ring_lr(owner_base[o], g, d->ring_parity, s);
    }
    FWGPU_HIP(hipMemsetAsync(d->d_push_cnt, 0, (size_t)(2 * N + 1) * 4, d->stream));
    FWGPU_HIP(hipMemcpyAsync(d->d_push, &pr, sizeof(pr), hipMemcpyHostToDevice, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));  // (pr is a local)
    return FWGPU_OK;
}
}  // namespace

// In-process group form.  records[j] / rec_off[j] / n[j]: rank j's micro-batch.  FWGPU_MODE_SEQUENTIAL (fwgpu_dist_group_set_mode): rank after
// rank, each in example order, pushes in buffer order and in-order applies -- with one example per call that is the sequential reference.
int fwgpu_dist_group_learn_owner(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                 const uint64_t *const *rec_off, const uint32_t *n, float *const *preds, int update) {
    if (!g || !t || !records || !rec_off || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const uint32_t N = (uint32_t)g->ranks.size();
    if (N > 8 || (N & (N - 1))) return fail(FWGPU_ERR_INVALID, "owner-side apply: 1, 2, 4 or 8 ranks");
    fwgpu_regressor *r0 = g->ranks[0]->r;
    if (r0->nn.n_layers) return fail(FWGPU_ERR_INVALID, "owner-side apply: models with a deep head are not covered");
    int lg = 0;
    while ((1u << lg) < N) lg++;
    if ((r0->cfg.ffm_k && (int)r0->cfg.ffm_bit_precision < lg) || (int)r0->cfg.bit_precision < lg)
        return fail(FWGPU_ERR_INVALID, "owner-side apply: fewer table entries than ranks");
    PeerShards ps{};
    ps.n = N;
    ps.shift_ffm = r0->cfg.ffm_k ? r0->cfg.ffm_bit_precision - lg : 31;
    ps.shift_lr = r0->cfg.bit_precision - lg;
    for (uint32_t j = 0; j < N; j++) {
        ps.ffm_w[j] = g->ranks[j]->r->d_ffm_w;
        ps.ffm_acc[j] = g->ranks[j]->r->d_ffm_acc;
        ps.lr[j] = g->ranks[j]->r->d_lr;
    }
    for (uint32_t i = 0; i < N; i++)
        for (uint32_t j = 0; j < N; j++) {
            const int di = g->ranks[i]->r->device, dj = g->ranks[j]->r->device;
            if (di == dj) continue;
            FWGPU_HIP(hipSetDevice(di));
            hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(FWGPU_ERR_DEVICE, "hipDeviceEnablePeerAccess failed");
            (void)hipGetLastError();
        }
    int rc;
    const int mode = g->ranks[0]->mode;
    std::vector<uint32_t> shapes((size_t)N * 4);
    uint32_t need_ffm = 64, need_lr = 64;
    for (uint32_t j = 0; j < N; j++) {
        if ((rc = sparse_begin(g->ranks[j].get(), t, records[j], rec_off[j], n[j], nullptr, &shapes[4 * j]))) return rc;  // (uploads the rank's records)
        const uint64_t nl = (uint64_t)shapes[4 * j + 0] * shapes[4 * j + 3], nf = (uint64_t)shapes[4 * j + 1] * shapes[4 * j + 3];
        if (nl > 0x7fffffffull || nf > 0x7fffffffull) return fail(FWGPU_ERR_RANGE, "owner-side apply: the step's gradient rows do not fit 31-bit ring positions; use smaller steps");
        need_lr = std::max(need_lr, (uint32_t)nl);
        need_ffm = std::max(need_ffm, (uint32_t)nf);
    }
    for (uint32_t j = 0; j < N; j++)
        if ((rc = rings_reserve(g->ranks[j].get(), N, need_ffm, need_lr))) return rc;
    unsigned char *bases[8] = {nullptr};
    uint32_t cap_ffm = 0xffffffffu, cap_lr = 0xffffffffu;
    for (uint32_t o = 0; o < N; o++) {
        bases[o] = g->ranks[o]->own_rings;
        cap_ffm = std::min(cap_ffm, g->ranks[o]->ring_cap_ffm);
        cap_lr = std::min(cap_lr, g->ranks[o]->ring_cap_lr);
    }
    // (one geometry for the whole group: the owners' allocations may differ in size after a regrowth, the regions are laid out for the smallest)
    for (uint32_t o = 0; o < N; o++)
        if (g->ranks[o]->ring_cap_ffm != cap_ffm || g->ranks[o]->ring_cap_lr != cap_lr) {
            g->ranks[o]->ring_cap_ffm = 0;  // force a common size
            g->ranks[o]->ring_cap_lr = 0;
            if (g->ranks[o]->own_rings) (void)hipFree(g->ranks[o]->own_rings);
            g->ranks[o]->own_rings = nullptr;
        }
    for (uint32_t o = 0; o < N; o++) {
        if ((rc = rings_reserve(g->ranks[o].get(), N, std::max(need_ffm, cap_ffm), std::max(need_lr, cap_lr)))) return rc;
        bases[o] = g->ranks[o]->own_rings;
    }
    cap_ffm = g->ranks[0]->ring_cap_ffm;
    cap_lr = g->ranks[0]->ring_cap_lr;
    const uint32_t R = r0->cfg.ffm_k ? r0->cfg.ffm_k * r0->cfg.ffm_num_fields : 0;
    const RingGeom geom = ring_geom(N, R, cap_ffm, cap_lr);
    std::vector<uint32_t> counts((size_t)N * (2 * N + 1), 0);
    for (uint32_t j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        FWGPU_HIP(hipSetDevice(d->r->device));
        d->ring_parity = 0;
        if (!d->d_peers) FWGPU_HIP(hipMalloc((void **)&d->d_peers, sizeof(PeerShards)));
        FWGPU_HIP(hipMemcpyAsync(d->d_peers, &ps, sizeof(PeerShards), hipMemcpyHostToDevice, d->stream));
        if ((rc = push_descriptor(d, N, j, bases, cap_ffm, cap_lr))) return rc;
        if ((rc = run_batch_peer(d->r, d->cur, mode, update, d->d_peers, d->stream, d->d_push))) return rc;
        FWGPU_HIP(hipMemcpyAsync(&counts[(size_t)j * (2 * N + 1)], d->d_push_cnt, (size_t)(2 * N + 1) * 4, hipMemcpyDeviceToHost, d->stream));
        if (mode == FWGPU_MODE_SEQUENTIAL) {  // rank after rank: this rank's gradients are applied before the next rank reads a weight
            FWGPU_HIP(hipStreamSynchronize(d->stream));
            if (counts[(size_t)j * (2 * N + 1) + 2 * N]) return fail(FWGPU_ERR_RANGE, "owner-side apply: a ring overflowed");
            for (uint32_t o = 0; o < N; o++) {
                fwgpu_dist *od = g->ranks[o].get();
                FWGPU_HIP(hipSetDevice(od->r->device));
                FWGPU_HIP(launch_owner_apply(od->r, od->r->d_lr, ring_keys(bases[o], geom, 0, j), ring_rows(bases[o], geom, 0, j), counts[(size_t)j * (2 * N + 1) + o],
                                             ring_lr(bases[o], geom, 0, j), counts[(size_t)j * (2 * N + 1) + N + o], true, od->stream));
                FWGPU_HIP(hipStreamSynchronize(od->stream));
            }
        }
    }
    if (mode != FWGPU_MODE_SEQUENTIAL) {
        for (auto &d : g->ranks) FWGPU_HIP(hipStreamSynchronize(d->stream));
        for (uint32_t j = 0; j < N; j++)
            if (counts[(size_t)j * (2 * N + 1) + 2 * N]) return fail(FWGPU_ERR_RANGE, "owner-side apply: a ring overflowed");
        for (uint32_t o = 0; o < N; o++) {
            fwgpu_dist *od = g->ranks[o].get();
            FWGPU_HIP(hipSetDevice(od->r->device));
            for (uint32_t j = 0; j < N; j++)
                FWGPU_HIP(launch_owner_apply(od->r, od->r->d_lr, ring_keys(bases[o], geom, 0, j), ring_rows(bases[o], geom, 0, j), counts[(size_t)j * (2 * N + 1) + o],
                                             ring_lr(bases[o], geom, 0, j), counts[(size_t)j * (2 * N + 1) + N + o], false, od->stream));
        }
    }
    for (uint32_t j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        FWGPU_HIP(hipSetDevice(d->r->device));
        if (preds && preds[j] && d->B) FWGPU_HIP(hipMemcpyAsync(preds[j], d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
    }
    return FWGPU_OK;
}

// ------------------------------------------------------------------ owner-side apply, STREAMING form
// The step-synchronous form above applies a step's gradients when the step's kernels have ended: every gradient of the step was taken at the step's
// first weights and every one of them lands, which bounds the step to ~1024 global examples.  Here the owners drain their regions WHILE the
// sources fill them: the first workgroups of every rank's example kernel are that rank's consumers (kernels.hip owner_stream_consume), the regions are
// circular, flow control goes through words the owner stores in the SOURCE's memory, and a source's last producer workgroup stores the regions' final
// positions of the step into the owners' memory -- staleness = the examples in flight, like hogwild.rs:89-103, whatever the step's size; no collective,
// no host round trip and no second stream inside a step.  What it needs is what hogwild needs: the ranks' kernels RUN AT THE SAME TIME (one process per
// GPU: always; an in-process group sharing one device: one hardware queue per rank, i.e. up to four ranks).
// The streaming form needs the kernels of ranks that share a device to RUN AT THE SAME TIME.  Streams of one process map onto a handful of hardware queues
// (ROCclr: GPU_MAX_HW_QUEUES, default 4, handed out round robin), and two streams on one queue run their kernels one after the other -- with four
// in-process ranks plus the default stream that is a rank waiting for a peer whose kernel sits behind its own.  The library asks for eight queues before
// the runtime initialises (no effect when the variable is set already, or when HIP was initialised before the library was loaded -- and measured: none either
// once torch has been imported into the process, whichever came first).  So the group form PROBES before its first streaming step (stream_concurrent below) and
// refuses loudly instead of hanging; one process per rank has its own queues and needs none of this.
__attribute__((constructor)) static void fwgpu_ask_for_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

namespace {
struct StreamGeom {
    uint32_t N, R, lg_ffm, lg_lr;
    size_t off_tag, off_rows, off_lr, off_fin, off_free, off_credit, off_cnt, off_done, off_own, off_push, bytes;
};
StreamGeom stream_geom(uint32_t N, uint32_t R, uint32_t lg_ffm, uint32_t lg_lr) {
    StreamGeom g{N, R, lg_ffm, lg_lr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t cf = (size_t)1 << lg_ffm, cl = (size_t)1 << lg_lr;
    size_t o = 0;
    g.off_tag = o; o = up(o + (size_t)N * cf * 8);             // owner role, per source
    g.off_rows = o; o = up(o + (size_t)N * cf * R * 4);
    g.off_lr = o; o = up(o + (size_t)N * cl * 8);
    g.off_fin = o; o = up(o + (size_t)2 * 2 * N * 8);           // two halves, by step parity (see stream_launch)
    g.off_free = o; o = up(o + (size_t)N * cf * 4);             // source role, per owner
    g.off_credit = o; o = up(o + (size_t)N * cl * 4);           // (per owner: the LR slots' free generations)
    g.off_cnt = o; o = up(o + (size_t)2 * N * 4);
    g.off_done = o; o = up(o + 4);
    g.off_own = o; o = up(o + sizeof(OwnerStream));             // the two descriptors the kernel reads
    g.off_push = o; o = up(o + sizeof(PushRings));
    g.bytes = o;
    return g;
}
int stream_reserve(fwgpu_dist *d, uint32_t N, uint32_t lg_ffm, uint32_t lg_lr, bool *fresh, bool allow_fine = true) {
    const uint32_t R = d->r->cfg.ffm_k ? d->r->cfg.ffm_k * d->r->cfg.ffm_num_fields : 0;
    const StreamGeom g = stream_geom(N, R, lg_ffm, lg_lr);
    FWGPU_HIP(hipSetDevice(d->r->device));
    *fresh = false;
    if (!d->st_mem || d->st_bytes != g.bytes || d->st_lg_ffm != lg_ffm || d->st_lg_lr != lg_lr || d->st_n != N) {
        if (d->st_mem) (void)hipFree(d->st_mem);
        d->st_mem = nullptr;
        // The regions are polled and written by OTHER ranks' kernels while this rank's kernel runs: fine-grained device memory where the runtime grants it
        // (coarse-grained memory promises cross-agent visibility at kernel boundaries only; every access of the protocol is system-scope -- sc0 sc1 -- on top).
        // FWGPU_STREAM_FINEGRAINED=0: plain hipMalloc (what every round-5 test ran on; ranks that share one device need no more).
        static const bool fine = [] { const char *e = std::getenv("FWGPU_STREAM_FINEGRAINED"); return !(e && e[0] == '0'); }();
        bool got = false;
        if (fine && allow_fine) {
            got = hipExtMallocWithFlags((void **)&d->st_mem, g.bytes, hipDeviceMallocFinegrained) == hipSuccess;
            if (!got) {
                (void)hipGetLastError();
                d->st_mem = nullptr;
            }
        }
        if (!got && hipMalloc((void **)&d->st_mem, g.bytes) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FWGPU_ERR_OOM, "owner-side apply, streaming form: no memory for the regions (smaller log2 capacities?)");
        }
        if (!d->st_abort) {
            if (hipHostMalloc((void **)&d->st_abort, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                return fail(FWGPU_ERR_OOM, "owner-side apply, streaming form: no pinned host memory for the abort word");
            }
            *d->st_abort = 0u;
        }
        d->st_bytes = g.bytes;
        d->st_lg_ffm = lg_ffm;
        d->st_lg_lr = lg_lr;
        d->st_n = N;
        *fresh = true;
    }
    return FWGPU_OK;
}
// everything back to position 0 (tags, free generations, credits, counters, final positions): first use, and when positions come close to the 32-bit
// wrap (generations are compared modulo: a fresh start at a step boundary is simpler than proving the wrap)
int stream_reset(fwgpu_dist *d) {
    const uint32_t R = d->r->cfg.ffm_k ? d->r->cfg.ffm_k * d->r->cfg.ffm_num_fields : 0;
    const StreamGeom g = stream_geom(d->st_n, R, d->st_lg_ffm, d->st_lg_lr);
    FWGPU_HIP(hipSetDevice(d->r->device));
    FWGPU_HIP(hipMemsetAsync(d->st_mem + g.off_tag, 0, g.off_rows - g.off_tag, d->stream));
    FWGPU_HIP(hipMemsetAsync(d->st_mem + g.off_lr, 0, g.bytes - g.off_lr, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));
    for (int s = 0; s < 8; s++) d->st_pos_ffm[s] = d->st_pos_lr[s] = 0;
    d->st_step = 0;
    d->st_next_check = 0;
    return FWGPU_OK;
}
// do the kernels of these ranks (which share device `dev`) run at the same time?  (kernels.hip rendezvous_kernel)
int stream_concurrent(const std::vector<fwgpu_dist *> &ranks, int dev, bool *ok) {
    FWGPU_HIP(hipSetDevice(dev));
    uint32_t *d = nullptr;
    FWGPU_HIP(hipMalloc((void **)&d, 8));
    FWGPU_HIP(hipMemset(d, 0, 8));
    for (fwgpu_dist *r : ranks) FWGPU_HIP(hipStreamSynchronize(r->stream));
    for (fwgpu_dist *r : ranks) FWGPU_HIP(launch_rendezvous(d, (uint32_t)ranks.size(), d + 1, r->stream));
    for (fwgpu_dist *r : ranks) FWGPU_HIP(hipStreamSynchronize(r->stream));
    uint32_t h[2] = {0, 0};
    FWGPU_HIP(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    *ok = h[1] == (uint32_t)ranks.size();
    return FWGPU_OK;
}
uint32_t stream_consumer_wgs(uint32_t N, uint32_t wish, uint32_t waves_per_wg) {
    // consumer waves: n for the LR regions + the row regions' stripes.  0 = a share of the grid, chosen at launch (kernels.hip launch_persistent)
    if (!wish) return 0xffffffffu;
    uint32_t G = wish;
    while ((uint64_t)G * waves_per_wg < 2ull * N) G++;
    return G;
}
// One rank's launch of a streaming step: the descriptors of both roles go to the rank's own allocation, then the example kernel with the consumers on top.
// base[j] = rank j's allocation as reachable from this rank.
int stream_launch(fwgpu_dist *d, const StreamGeom &geo, unsigned char *const base[8], float *lr_base, int update, uint32_t consumer_wgs) {
    if (d->st_void) return fail(FWGPU_ERR_PEER, "owner-side apply, streaming form: an earlier step of this rank was given up (time-out): the regions' state is void, the rank cannot take part in further streaming steps");
    const uint32_t N = geo.N, R = geo.R, me = (uint32_t)d->rank;
    const size_t cf = (size_t)1 << geo.lg_ffm, cl = (size_t)1 << geo.lg_lr;
    FWGPU_HIP(hipSetDevice(d->r->device));
    OwnerStream os{};
    os.n = N;
    os.R = R;
    os.log2cap_ffm = geo.lg_ffm;
    os.log2cap_lr = geo.lg_lr;
    os.step = d->st_step;
    for (uint32_t s = 0; s < N; s++) {
        os.ffm_tag[s] = reinterpret_cast<const unsigned long long *>(d->st_mem + geo.off_tag) + (size_t)s * cf;
        os.ffm_rows[s] = reinterpret_cast<const float *>(d->st_mem + geo.off_rows) + (size_t)s * cf * R;
        os.lr_word[s] = reinterpret_cast<const unsigned long long *>(d->st_mem + geo.off_lr) + (size_t)s * cl;
        os.ffm_free[s] = reinterpret_cast<uint32_t *>(base[s] + geo.off_free) + (size_t)me * cf;
        os.lr_free[s] = reinterpret_cast<uint32_t *>(base[s] + geo.off_credit) + (size_t)me * cl;
        os.start_ffm[s] = d->st_pos_ffm[s];
        os.start_lr[s] = d->st_pos_lr[s];
    }
    // Final positions are kept per step PARITY: with no collective in a step, source A may be a whole step ahead of owner B (its producers need only room in
    // B's regions) and store step t + 1's positions while B's consumers of step t still run -- into the other half.  It cannot be two steps ahead: its kernel
    // t + 1 ends only when its own consumers have B's positions of step t + 1, i.e. after B's kernel t.
    const size_t fin_half = (size_t)(d->st_step & 1u) * 2 * N;
    os.fin = reinterpret_cast<const unsigned long long *>(d->st_mem + geo.off_fin) + fin_half;
    os.w = d->r->d_ffm_w;
    os.acc = d->r->d_ffm_acc;
    os.lr = lr_base;
    os.ffm_rate = d->r->cfg.ffm_learning_rate;
    os.ffm_mpt = -d->r->cfg.ffm_power_t;
    os.lr_rate = d->r->cfg.learning_rate;
    os.lr_mpt = -d->r->cfg.power_t;
    os.lut_ffm = d->r->d_lut_ffm;
    os.lut_lr = d->r->d_lut_lr;
    os.abort = d->st_abort;
    PushRings pr{};
    pr.n = N;
    pr.stream = 1;
    pr.abort = d->st_abort;
    pr.log2cap_ffm = geo.lg_ffm;
    pr.log2cap_lr = geo.lg_lr;
    pr.cnt = reinterpret_cast<uint32_t *>(d->st_mem + geo.off_cnt);
    pr.consumers = update ? consumer_wgs : 0;  // (a read-only step pushes nothing: nobody has anything to drain)
    pr.step = d->st_step;
    pr.src = me;
    pr.done = reinterpret_cast<uint32_t *>(d->st_mem + geo.off_done);
    pr.own = reinterpret_cast<const OwnerStream *>(d->st_mem + geo.off_own);
    for (uint32_t o = 0; o < N; o++) {
        pr.ffm_tag[o] = reinterpret_cast<unsigned long long *>(base[o] + geo.off_tag) + (size_t)me * cf;
        pr.ffm_rows[o] = reinterpret_cast<float *>(base[o] + geo.off_rows) + (size_t)me * cf * R;
        pr.lr_word[o] = reinterpret_cast<unsigned long long *>(base[o] + geo.off_lr) + (size_t)me * cl;
        pr.ffm_free[o] = reinterpret_cast<const uint32_t *>(d->st_mem + geo.off_free) + (size_t)o * cf;
        pr.lr_free[o] = reinterpret_cast<const uint32_t *>(d->st_mem + geo.off_credit) + (size_t)o * cl;
        pr.fin_remote[o] = reinterpret_cast<unsigned long long *>(base[o] + geo.off_fin) + fin_half;
    }
    FWGPU_HIP(hipMemcpyAsync(d->st_mem + geo.off_own, &os, sizeof(os), hipMemcpyHostToDevice, d->stream));
    FWGPU_HIP(hipMemcpyAsync(d->st_mem + geo.off_push, &pr, sizeof(pr), hipMemcpyHostToDevice, d->stream));
    FWGPU_HIP(hipMemsetAsync(d->st_mem + geo.off_done, 0, 4, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));  // (os / pr are locals)
    // (a stripe's stride -- consumer waves per source -- stays below a quarter of a region's capacity: kernels.hip owner_stream_consume takes two positions of a
    // stripe per round and the producers' flow control must never wait for a row such a round holds)
    const uint32_t max_waves = (uint32_t)std::min<uint64_t>(0x7fffffffull, ((uint64_t)1 << geo.lg_ffm) / 4 * N + N);
    return run_batch_peer(d->r, d->cur, FWGPU_MODE_HOGWILD, update, d->d_peers, d->stream, reinterpret_cast<const PushRings *>(d->st_mem + geo.off_push), pr.consumers,
                          d->st_share, max_waves, N);
}
// the launch has ended: where this rank's regions (as owner) stand now = where the next step's consumers start
// A streaming launch waits INSIDE the kernel for its peers (consumers for tags and final positions, producers for free slots): a peer that never launches --
// gone, or returned early with an error of its own -- would hold this rank's GPU for ever.  The wait is therefore polled against a deadline
// (FWGPU_DIST_TIMEOUT_MS where set, ten minutes otherwise); past it the host sets the abort word, every wait loop of the kernel leaves, the step is reported
// as failed and the rank's streaming state as void.
int stream_wait(fwgpu_dist *d) {
    static const long timeout_ms = [] {
        const char *e = std::getenv("FWGPU_DIST_TIMEOUT_MS");
        const long v = e ? std::atol(e) : 0L;
        return v > 0 ? v : 600000L;
    }();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t it = 0;; it++) {
        const hipError_t q = hipStreamQuery(d->stream);
        if (q == hipSuccess) return FWGPU_OK;
        if (q != hipErrorNotReady) return fail(FWGPU_ERR_DEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) {
            __atomic_store_n(d->st_abort, 1u, __ATOMIC_SEQ_CST);
            d->st_void = true;
            for (int i = 0; i < 200000 && hipStreamQuery(d->stream) == hipErrorNotReady; i++) std::this_thread::sleep_for(std::chrono::microseconds(50));
            return fail(FWGPU_ERR_PEER, "owner-side apply, streaming form: the step did not complete within its deadline (FWGPU_DIST_TIMEOUT_MS; a peer rank is gone or never launched): "
                                        "the launch was told to give up, this rank's streaming state is void");
        }
        if (it < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}
int stream_finish(fwgpu_dist *d, const StreamGeom &geo, int update, bool *near_wrap) {
    FWGPU_HIP(hipSetDevice(d->r->device));
    if (int rcw = stream_wait(d)) return rcw;
    if (update) {
        std::vector<unsigned long long> fin((size_t)2 * geo.N);
        FWGPU_HIP(hipMemcpy(fin.data(), d->st_mem + geo.off_fin + (size_t)(d->st_step & 1u) * 2 * geo.N * 8, fin.size() * 8, hipMemcpyDeviceToHost));
        for (uint32_t s = 0; s < geo.N; s++) {
            if ((uint32_t)(fin[s] >> 32) != d->st_step || (uint32_t)(fin[geo.N + s] >> 32) != d->st_step)
                return fail(FWGPU_ERR_DEVICE, "owner-side apply, streaming form: a source's final positions of the step never arrived");
            d->st_pos_ffm[s] = (uint32_t)fin[s];
            d->st_pos_lr[s] = (uint32_t)fin[geo.N + s];
            *near_wrap |= d->st_pos_ffm[s] > 0x60000000u || d->st_pos_lr[s] > 0x60000000u;
        }
    }
    return FWGPU_OK;
}
}  // namespace

// In-process group form of the streaming step.  batches[j] (may be NULL): rank j's micro-batch already in HBM (a record batch of its regressor);
// otherwise records[j] / rec_off[j] / n[j].  HOGWILD only (the in-order reference is fwgpu_dist_group_learn_owner in FWGPU_MODE_SEQUENTIAL).
// log2_rows / log2_lr: capacity of one (owner, source) region in gradient rows / LR gradients (0: 2^15 / 2^16); consumer_workgroups: workgroups of a
// rank's launch that drain its regions (0: 32).
int fwgpu_dist_group_learn_owner_stream(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records, const uint64_t *const *rec_off,
                                        const uint32_t *n, fwgpu_batch *const *batches, float *const *preds, int update, uint32_t log2_rows, uint32_t log2_lr,
                                        uint32_t consumer_workgroups) {
    if (!g || !t || (!batches && (!records || !rec_off || !n))) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const uint32_t N = (uint32_t)g->ranks.size();
    if (N > 8 || (N & (N - 1))) return fail(FWGPU_ERR_INVALID, "owner-side apply: 1, 2, 4 or 8 ranks");
    fwgpu_regressor *r0 = g->ranks[0]->r;
    if (r0->nn.n_layers) return fail(FWGPU_ERR_INVALID, "owner-side apply: models with a deep head are not covered");
    if (r0->cfg.bit_precision > 30) return fail(FWGPU_ERR_INVALID, "owner-side apply, streaming form: LR hashes of more than 30 bits are not covered (the region's words keep 2 bits)");
    {   // the ranks' kernels must run at the same time: ranks that share a device need a hardware queue each
        uint32_t same = 0;
        for (uint32_t j = 0; j < N; j++) same += g->ranks[j]->r->device == r0->device ? 1u : 0u;
        for (uint32_t j = 0; j < N; j++) {
            uint32_t sh = 0;
            for (uint32_t i = 0; i < N; i++) sh += g->ranks[i]->r->device == g->ranks[j]->r->device ? 1u : 0u;
            g->ranks[j]->st_share = sh;
        }
        if (same > 4) return fail(FWGPU_ERR_INVALID, "owner-side apply, streaming form: more than four in-process ranks on one device (their kernels would queue behind one another); one process per rank there");
        if (!g->ranks[0]->st_probed) {  // once per group: can the ranks that share a device run their kernels at the same time?
            for (uint32_t j = 0; j < N; j++) {
                std::vector<fwgpu_dist *> on_dev;
                for (uint32_t i = 0; i < N; i++)
                    if (g->ranks[i]->r->device == g->ranks[j]->r->device) on_dev.push_back(g->ranks[i].get());
                if (on_dev.size() < 2 || on_dev[0] != g->ranks[j].get()) continue;  // (each device once)
                bool ok = false;
                int rcp = stream_concurrent(on_dev, g->ranks[j]->r->device, &ok);
                if (rcp) return rcp;
                if (!ok)
                    return fail(FWGPU_ERR_DEVICE, "owner-side apply, streaming form: the kernels of the " + std::to_string(on_dev.size()) + " in-process ranks on device " +
                                                      std::to_string(g->ranks[j]->r->device) + " do not run at the same time (their streams share hardware queues: GPU_MAX_HW_QUEUES, "
                                                      "default 4 and fixed once the HIP runtime -- or torch -- is up); fewer ranks per device, or one process per rank");
            }
            for (uint32_t j = 0; j < N; j++) g->ranks[j]->st_probed = true;
        }
    }
    int lg = 0;
    while ((1u << lg) < N) lg++;
    if ((r0->cfg.ffm_k && (int)r0->cfg.ffm_bit_precision < lg) || (int)r0->cfg.bit_precision < lg)
        return fail(FWGPU_ERR_INVALID, "owner-side apply: fewer table entries than ranks");
    const uint32_t lgf = log2_rows ? log2_rows : 15, lgl = log2_lr ? log2_lr : 16;
    if (lgf < 6 || lgf > 24 || lgl < 6 || lgl > 24) return fail(FWGPU_ERR_INVALID, "owner-side apply, streaming form: log2 capacities of 6 .. 24");
    const uint32_t R = r0->cfg.ffm_k ? r0->cfg.ffm_k * r0->cfg.ffm_num_fields : 0;
    PeerShards ps{};
    ps.n = N;
    ps.shift_ffm = r0->cfg.ffm_k ? r0->cfg.ffm_bit_precision - lg : 31;
    ps.shift_lr = r0->cfg.bit_precision - lg;
    for (uint32_t j = 0; j < N; j++) {
        ps.ffm_w[j] = g->ranks[j]->r->d_ffm_w;
        ps.ffm_acc[j] = g->ranks[j]->r->d_ffm_acc;
        ps.lr[j] = g->ranks[j]->r->d_lr;
    }
    for (uint32_t i = 0; i < N; i++)
        for (uint32_t j = 0; j < N; j++) {
            const int di = g->ranks[i]->r->device, dj = g->ranks[j]->r->device;
            if (di == dj) continue;
            FWGPU_HIP(hipSetDevice(di));
            hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(FWGPU_ERR_DEVICE, "hipDeviceEnablePeerAccess failed");
            (void)hipGetLastError();
        }
    int rc;
    bool any_fresh = false;
    for (uint32_t j = 0; j < N; j++) {
        bool fresh = false;
        if ((rc = stream_reserve(g->ranks[j].get(), N, lgf, lgl, &fresh))) return rc;
        any_fresh |= fresh;
    }
    if (any_fresh)
        for (uint32_t j = 0; j < N; j++)
            if ((rc = stream_reset(g->ranks[j].get()))) return rc;
    const StreamGeom geo = stream_geom(N, R, lgf, lgl);
    unsigned char *bases[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (uint32_t j = 0; j < N; j++) bases[j] = g->ranks[j]->st_mem;
    std::vector<uint32_t> shapes((size_t)N * 4);
    // Every rank's preparation first: an input error of ANY rank (unusable records, a batch of another regressor) returns here, before one step counter has
    // moved and before one kernel has been launched -- the group's ranks stay at the same step (their step tags, the final positions' parity halves and
    // the regions' positions all hang on it: a rank one step ahead of its peers would wait for final positions that never come).
    for (uint32_t j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        fwgpu_batch *bj = batches ? batches[j] : nullptr;
        if ((rc = sparse_begin(d, t, bj ? nullptr : records[j], bj ? nullptr : rec_off[j], bj ? bj->n : n[j], bj, &shapes[4 * j]))) return rc;
        FWGPU_HIP(hipSetDevice(d->r->device));
        if (!d->d_peers) FWGPU_HIP(hipMalloc((void **)&d->d_peers, sizeof(PeerShards)));
        FWGPU_HIP(hipMemcpyAsync(d->d_peers, &ps, sizeof(PeerShards), hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));  // (ps is a local)
    }
    for (uint32_t j = 0; j < N; j++) g->ranks[j]->st_step++;
    const uint32_t waves = std::max<uint32_t>(1u, r0->launch.threads / 64u);
    const uint32_t CW = stream_consumer_wgs(N, consumer_workgroups, waves);
    // From here on every rank MUST launch: the peers' kernels wait for its final positions and for its consumers.  A rank whose launch fails (the launch shape
    // does not fit, no memory) takes part with an EMPTY batch instead -- its producers publish the final positions at once, its consumers serve the peers --
    // and its error is what the call returns once every rank's kernel has ended.
    int first_rc = FWGPU_OK;
    std::string first_msg;
    for (uint32_t j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        rc = stream_launch(d, geo, bases, d->r->d_lr, update, CW);
        if (rc) {
            if (!first_rc) {
                first_rc = rc;
                first_msg = fwgpu_last_error();
            }
            uint32_t sh2[4];
            const uint64_t zero_off[1] = {0};
            int rc2 = sparse_begin(d, t, nullptr, zero_off, 0, nullptr, sh2);
            if (!rc2) rc2 = stream_launch(d, geo, bases, d->r->d_lr, update, CW);
            if (rc2) {  // not even an empty launch: the peers are told to give up (they would wait for this rank for ever)
                for (uint32_t i = 0; i < N; i++) {
                    if (g->ranks[i]->st_abort) __atomic_store_n(g->ranks[i]->st_abort, 1u, __ATOMIC_SEQ_CST);
                    g->ranks[i]->st_void = true;
                }
            }
        }
    }
    bool near_wrap = false;
    for (uint32_t j = 0; j < N; j++)
        if ((rc = stream_finish(g->ranks[j].get(), geo, update, &near_wrap)) && !first_rc) {
            first_rc = rc;
            first_msg = fwgpu_last_error();
        }
    if (first_rc) {
        set_error(first_msg + " (rank-level failure inside a streaming step: the other ranks' kernels were served by an empty launch)");
        return first_rc;
    }
    if (near_wrap)
        for (uint32_t j = 0; j < N; j++)
            if ((rc = stream_reset(g->ranks[j].get()))) return rc;
    for (uint32_t j = 0; j < N; j++) {
        fwgpu_dist *d = g->ranks[j].get();
        FWGPU_HIP(hipSetDevice(d->r->device));
        if (preds && preds[j] && d->B) FWGPU_HIP(hipMemcpy(preds[j], d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost));
    }
    return FWGPU_OK;
}

// Process-per-rank form.  fwgpu_dist_owner_attach: peer attach (tables + LR shards) plus this rank's rings -- sized for steps of up to
// max_rows gradient rows and max_lr LR gradients per source -- exported and mapped like the tables.  fwgpu_dist_learn_owner: one COLLECTIVE step
// (every rank calls it; n may be 0): push, all-gather of the counts, apply of what the sources pushed here.
int fwgpu_dist_owner_attach(fwgpu_dist *d, uint32_t max_rows, uint32_t max_lr) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (d->rings_attached) return FWGPU_OK;
    int rc = fwgpu_dist_peer_attach(d);
    if (rc) return rc;
    const uint32_t N = (uint32_t)d->n;
    if ((rc = rings_reserve(d, N, std::max<uint32_t>(max_rows, 64), std::max<uint32_t>(max_lr, 64)))) return rc;
    if (d->own_rings_bytes >= (1ull << 31)) return fail(FWGPU_ERR_RANGE, "owner-side apply: rings of 2 GiB or more cannot be mapped by the peers; use smaller steps");
    FWGPU_HIP(hipSetDevice(d->r->device));
    hipIpcMemHandle_t mine{};
    FWGPU_HIP(hipIpcGetMemHandle(&mine, d->own_rings));
    static_assert(sizeof(hipIpcMemHandle_t) % 4 == 0, "all-gathered as 32-bit words");
    hipIpcMemHandle_t *d_all = nullptr;
    FWGPU_HIP(hipMalloc((void **)&d_all, sizeof(hipIpcMemHandle_t) * (size_t)N));
    struct FreeOnExit {
        void *q;
        ~FreeOnExit() { if (q) (void)hipFree(q); }
    } guard{d_all};
    FWGPU_HIP(hipMemcpyAsync(d_all + d->rank, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));
    if (N > 1) FWGPU_NCCL(g_rccl.AllGather(d_all + d->rank, d_all, sizeof(mine) / 4, ncclUint32, d->comm, d->stream));
    if ((rc = wait_stream(d))) return rc;
    std::vector<hipIpcMemHandle_t> all((size_t)N);
    FWGPU_HIP(hipMemcpy(all.data(), d_all, sizeof(mine) * (size_t)N, hipMemcpyDeviceToHost));
    for (uint32_t j = 0; j < N; j++) {
        if ((int)j == d->rank) {
            d->peer_rings[j] = d->own_rings;
            continue;
        }
        void *q = nullptr;
        FWGPU_HIP(hipIpcOpenMemHandle(&q, all[j], hipIpcMemLazyEnablePeerAccess));
        d->ipc_open.push_back(q);
        d->peer_rings[j] = static_cast<unsigned char *>(q);
    }
    d->rings_attached = true;
    d->ring_parity = 0;
    return fwgpu_dist_barrier(d);
}

int fwgpu_dist_learn_owner(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                           float *preds, int update) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (!d->rings_attached) return fail(FWGPU_ERR_INVALID, "fwgpu_dist_owner_attach first");
    const uint32_t N = (uint32_t)d->n;
    uint32_t shape[4];
    begin_result(d, sparse_begin(d, t, records, rec_off, n, nullptr, shape), shape);
    if (!d->begin_failed && ((uint64_t)shape[1] * shape[3] > d->ring_cap_ffm || (uint64_t)shape[0] * shape[3] > d->ring_cap_lr))
        begin_result(d, fail(FWGPU_ERR_RANGE, "owner-side apply: the step is larger than the rings fwgpu_dist_owner_attach sized"), shape);
    FWGPU_HIP(hipSetDevice(d->r->device));
    const uint32_t W = 2 * N + 1;
    int rc;
    if (!d->begin_failed) {
        if ((rc = push_descriptor(d, N, (uint32_t)d->rank, d->peer_rings, d->ring_cap_ffm, d->ring_cap_lr))) return abort_on_failure(d, rc);
        if ((rc = run_batch_peer(d->r, d->cur, d->mode, update, d->d_peers, d->stream, d->d_push))) return abort_on_failure(d, rc);
    } else {
        const std::vector<uint32_t> poison(W, kShapePoison);
        FWGPU_HIP(hipMemcpyAsync(d->d_push_cnt, poison.data(), W * 4, hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
    }
    // the step's one collective: every rank's counts to every rank.  Its completion on this rank also says that every source's kernel has finished.
    uint32_t *all = d->d_push_cnt + W;
    if (N > 1) {
        FWGPU_HIP(hipMemcpyAsync(all + (size_t)d->rank * W, d->d_push_cnt, W * 4, hipMemcpyDeviceToDevice, d->stream));
        rc = g_rccl.AllGather(all + (size_t)d->rank * W, all, W, ncclUint32, d->comm, d->stream) == ncclSuccess ? FWGPU_OK : fail(FWGPU_ERR_DEVICE, "owner-side apply: all-gather of the counts failed");
        if (rc) return abort_on_failure(d, rc);
    } else {
        FWGPU_HIP(hipMemcpyAsync(all, d->d_push_cnt, W * 4, hipMemcpyDeviceToDevice, d->stream));
    }
    std::vector<uint32_t> counts((size_t)N * W);
    uint32_t *pin_counts = nullptr;
    if ((rc = d->pinned(counts.size(), &pin_counts))) return fail(rc, "dist step: no pinned host memory for the read-back");
    FWGPU_HIP(hipMemcpyAsync(pin_counts, all, counts.size() * 4, hipMemcpyDeviceToHost, d->stream));
    if ((rc = wait_stream(d))) return rc;
    std::memcpy(counts.data(), pin_counts, counts.size() * 4);
    for (uint32_t j = 0; j < N; j++)
        if (counts[(size_t)j * W] == kShapePoison) {
            d->ring_parity ^= 1;  // (every rank flips: the ranks stay in step)
            if ((int)j == d->rank) return d->begin_rc;
            return fail(FWGPU_ERR_PEER, "rank " + std::to_string(j) + " reported a failure before the step's exchange: nothing was applied");
        }
    for (uint32_t j = 0; j < N; j++)
        if (counts[(size_t)j * W + 2 * N]) return abort_on_failure(d, fail(FWGPU_ERR_RANGE, "owner-side apply: a ring overflowed"));
    const uint32_t R = d->r->cfg.ffm_k ? d->r->cfg.ffm_k * d->r->cfg.ffm_num_fields : 0;
    const RingGeom geom = ring_geom(N, R, d->ring_cap_ffm, d->ring_cap_lr);
    float *lr_base = d->lr_shard ? d->lr_shard - 2 * d->lr_shard_lo : d->r->d_lr;
    for (uint32_t s = 0; s < N; s++)
        FWGPU_HIP(launch_owner_apply(d->r, lr_base, ring_keys(d->own_rings, geom, d->ring_parity, s), ring_rows(d->own_rings, geom, d->ring_parity, s),
                                     counts[(size_t)s * W + d->rank], ring_lr(d->own_rings, geom, d->ring_parity, s), counts[(size_t)s * W + N + d->rank],
                                     d->mode == FWGPU_MODE_SEQUENTIAL, d->stream));
    d->ring_parity ^= 1;
    if (d->mode == FWGPU_MODE_SEQUENTIAL && N > 1) {
        // the in-order form is the sequential reference: step t + 1's fetches (any rank's) must see step t's applies (every owner's).  A rank waits
        // only for its OWN apply below; what orders the other owners' applies in front of its next push kernel is this one-word all-reduce queued
        // behind the apply on every rank (the concurrent form tolerates the one-step staleness and does not pay for it)
        FWGPU_HIP(hipMemsetAsync(d->d_shape, 0, 4, d->stream));
        rc = g_rccl.AllReduce(d->d_shape, d->d_shape, 1, ncclFloat, ncclSum, d->comm, d->stream) == ncclSuccess ? FWGPU_OK : fail(FWGPU_ERR_DEVICE, "owner-side apply: closing all-reduce failed");
        if (rc) return abort_on_failure(d, rc);
    }
    if ((rc = wait_stream(d))) return rc;  // (polled; the caller's buffer is pageable: its copy comes after, not in front of the wait)
    if (preds && d->B) FWGPU_HIP(hipMemcpy(preds, d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost));
    return FWGPU_OK;
}

// Process-per-rank form of the STREAMING step.  fwgpu_dist_owner_stream_attach (collective, once): tables and LR shards as fwgpu_dist_peer_attach maps
// them, plus every rank's streaming allocation (one IPC handle per rank: its regions as owner AND the words the owners store for it as source).
// fwgpu_dist_learn_owner_stream: one step of THIS rank -- its kernel (consumers + producers) and nothing else: no collective.  Every rank must call it the
// same number of times (a rank's consumers leave a step when every source's final positions OF THAT STEP have arrived; n may be 0).  A rank whose
// local preparation fails takes part with an empty batch and returns its own error; the peers are not told (there is no exchange to tell them through).
// Every ~10^9 gradient rows the ranks start their positions over: that one step ends with two barriers.
int fwgpu_dist_owner_stream_attach(fwgpu_dist *d, uint32_t log2_rows, uint32_t log2_lr) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (d->r->cfg.bit_precision > 30) return fail(FWGPU_ERR_INVALID, "owner-side apply, streaming form: LR hashes of more than 30 bits are not covered (the region's words keep 2 bits)");
    const uint32_t N = (uint32_t)d->n;
    const uint32_t lgf = log2_rows ? log2_rows : 15, lgl = log2_lr ? log2_lr : 16;
    if (lgf < 6 || lgf > 24 || lgl < 6 || lgl > 24) return fail(FWGPU_ERR_INVALID, "owner-side apply, streaming form: log2 capacities of 6 .. 24");
    int rc = fwgpu_dist_peer_attach(d);
    if (rc) return rc;
    bool fresh = false;
    if ((rc = stream_reserve(d, N, lgf, lgl, &fresh))) return rc;
    if (d->st_bytes >= (1ull << 31)) return fail(FWGPU_ERR_RANGE, "owner-side apply, streaming form: regions of 2 GiB or more cannot be mapped by the peers; smaller log2 capacities");
    if ((rc = stream_reset(d))) return rc;
    FWGPU_HIP(hipSetDevice(d->r->device));
    hipIpcMemHandle_t mine{};
    if (hipIpcGetMemHandle(&mine, d->st_mem) != hipSuccess) {
        // (a runtime that does not export fine-grained allocations: the regions as plain device memory, as in round 5)
        (void)hipGetLastError();
        (void)hipFree(d->st_mem);
        d->st_mem = nullptr;
        if ((rc = stream_reserve(d, N, lgf, lgl, &fresh, /*allow_fine=*/false))) return rc;
        if ((rc = stream_reset(d))) return rc;
        FWGPU_HIP(hipIpcGetMemHandle(&mine, d->st_mem));
    }
    hipIpcMemHandle_t *d_all = nullptr;
    FWGPU_HIP(hipMalloc((void **)&d_all, sizeof(hipIpcMemHandle_t) * (size_t)N));
    struct FreeOnExit {
        void *q;
        ~FreeOnExit() { if (q) (void)hipFree(q); }
    } guard{d_all};
    FWGPU_HIP(hipMemcpyAsync(d_all + d->rank, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
    FWGPU_HIP(hipStreamSynchronize(d->stream));
    if (N > 1) FWGPU_NCCL(g_rccl.AllGather(d_all + d->rank, d_all, sizeof(mine) / 4, ncclUint32, d->comm, d->stream));
    if ((rc = wait_stream(d))) return rc;
    std::vector<hipIpcMemHandle_t> all((size_t)N);
    FWGPU_HIP(hipMemcpy(all.data(), d_all, sizeof(mine) * (size_t)N, hipMemcpyDeviceToHost));
    {   // which ranks share this rank's GPU?  (their kernels must be resident together: each takes an equal share of the device)
        char bus[64] = {0};
        FWGPU_HIP(hipDeviceGetPCIBusId(bus, sizeof(bus), d->r->device));
        uint32_t hsh = 2166136261u;
        for (const char *c = bus; *c; ++c) hsh = (hsh ^ (uint32_t)(unsigned char)*c) * 16777619u;
        FWGPU_HIP(hipMemcpyAsync(d->d_shape + 4 * d->rank, &hsh, 4, hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
        if (N > 1) FWGPU_NCCL(g_rccl.AllGather(d->d_shape + 4 * d->rank, d->d_shape, 4, ncclUint32, d->comm, d->stream));
        if ((rc = wait_stream(d))) return rc;
        std::vector<uint32_t> ids((size_t)N * 4);
        FWGPU_HIP(hipMemcpy(ids.data(), d->d_shape, ids.size() * 4, hipMemcpyDeviceToHost));
        d->st_share = 0;
        for (uint32_t j = 0; j < N; j++) d->st_share += ids[4 * (size_t)j] == hsh ? 1u : 0u;
    }
    for (uint32_t j = 0; j < N; j++) {
        if ((int)j == d->rank) {
            d->st_peer[j] = d->st_mem;
            continue;
        }
        void *q = nullptr;
        FWGPU_HIP(hipIpcOpenMemHandle(&q, all[j], hipIpcMemLazyEnablePeerAccess));
        d->ipc_open.push_back(q);
        d->st_peer[j] = static_cast<unsigned char *>(q);
    }
    return fwgpu_dist_barrier(d);
}

int fwgpu_dist_learn_owner_stream(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n, float *preds,
                                  int update, uint32_t consumer_workgroups) {
    if (!d || !d->comm) return fail(FWGPU_ERR_INVALID, "not an RCCL rank (fwgpu_dist_init)");
    if (!d->st_mem || !d->st_peer[d->rank]) return fail(FWGPU_ERR_INVALID, "fwgpu_dist_owner_stream_attach first");
    if (d->mode != FWGPU_MODE_HOGWILD) return fail(FWGPU_ERR_INVALID, "owner-side apply, streaming form: FWGPU_MODE_HOGWILD only (the in-order reference is fwgpu_dist_learn_owner)");
    const uint32_t N = (uint32_t)d->n;
    const uint32_t R = d->r->cfg.ffm_k ? d->r->cfg.ffm_k * d->r->cfg.ffm_num_fields : 0;
    const StreamGeom geo = stream_geom(N, R, d->st_lg_ffm, d->st_lg_lr);
    uint32_t shape[4];
    int own_rc = sparse_begin(d, t, records, rec_off, n, nullptr, shape);
    if (own_rc) {  // this rank's records are unusable: an empty batch keeps the job's steps aligned (its consumers must serve the peers all the same)
        const std::string msg = fwgpu_last_error();
        uint32_t sh2[4];
        const uint64_t zero_off[1] = {0};
        int rc2 = sparse_begin(d, t, nullptr, zero_off, 0, nullptr, sh2);
        if (rc2) return rc2;
        set_error(msg + " (the rank took part in the step with an empty batch)");
    }
    FWGPU_HIP(hipSetDevice(d->r->device));
    float *lr_base = d->lr_shard ? d->lr_shard - 2 * d->lr_shard_lo : d->r->d_lr;
    d->st_step++;
    const uint32_t waves = std::max<uint32_t>(1u, d->r->launch.threads / 64u);
    int rc = stream_launch(d, geo, d->st_peer, lr_base, update, stream_consumer_wgs(N, consumer_workgroups, waves));
    if (rc && !d->st_void) {  // the launch shape does not fit, no memory, ...: the peers' kernels wait for this rank's final positions and consumers all the same -- an empty launch serves them
        own_rc = rc;
        const std::string msg = fwgpu_last_error();
        uint32_t sh2[4];
        const uint64_t zero_off[1] = {0};
        int rc2 = sparse_begin(d, t, nullptr, zero_off, 0, nullptr, sh2);
        if (!rc2) rc2 = stream_launch(d, geo, d->st_peer, lr_base, update, stream_consumer_wgs(N, consumer_workgroups, waves));
        if (rc2) {
            d->st_void = true;
            return rc2;
        }
        set_error(msg + " (the rank took part in the step with an empty batch)");
        rc = FWGPU_OK;
    }
    if (rc) return rc;
    bool near_wrap = false;
    if ((rc = stream_finish(d, geo, update, &near_wrap))) return rc;
    if (N > 1 && d->st_step >= d->st_next_check) {
        // Does ANY rank come close to the positions' 32-bit wrap?  Two words, all-reduced: the flag, and this step's bound on the positions a region can advance by (examples x
        // the longest example's entries, summed over the ranks: an upper bound for every (owner, source) region).  The NEXT check is due before 2^28 more positions can have
        // passed at that pace -- the same step on every rank, since it is computed from the all-reduced value (round 5 checked every 1024 steps whatever the step's size:
        // steps of 65 536 examples on two ranks advance a region by 6.5 M positions each and would have wrapped in between).
        const uint64_t own_bound = (uint64_t)std::max(shape[0], shape[1]) * (uint64_t)shape[3];
        float v2[2] = {near_wrap ? 1.0f : 0.0f, (float)own_bound};
        FWGPU_HIP(hipMemcpyAsync(d->d_shape, v2, 8, hipMemcpyHostToDevice, d->stream));
        FWGPU_HIP(hipStreamSynchronize(d->stream));
        FWGPU_NCCL(g_rccl.AllReduce(d->d_shape, d->d_shape, 2, ncclFloat, ncclSum, d->comm, d->stream));
        if ((rc = wait_stream(d))) return rc;
        FWGPU_HIP(hipMemcpy(v2, d->d_shape, 8, hipMemcpyDeviceToHost));
        if (v2[0] > 0.0f) {
            if ((rc = fwgpu_dist_barrier(d))) return rc;
            if ((rc = stream_reset(d))) return rc;
            if ((rc = fwgpu_dist_barrier(d))) return rc;
        }
        const double per_step = std::max(1.0, (double)v2[1]);
        const double period = std::min(1024.0, std::max(1.0, std::floor(268435456.0 / per_step)));
        d->st_next_check = d->st_step + (uint32_t)period;
    } else if (N == 1 && near_wrap) {
        if ((rc = stream_reset(d))) return rc;
    }
    if (preds && d->B) FWGPU_HIP(hipMemcpy(preds, d->cur->pred, (size_t)d->B * 4, hipMemcpyDeviceToHost));
    return own_rc;
}

int fwgpu_dist_group_gather_tables(fwgpu_dist_group *g) {
    if (!g) return fail(FWGPU_ERR_INVALID, "NULL group");
    const int N = (int)g->ranks.size();
    for (int j = 0; j < N; j++) {
        fwgpu_regressor *src = g->ranks[j]->r;
        for (int i = 0; i < N; i++) {
            if (i == j) continue;
            fwgpu_regressor *dst = g->ranks[i]->r;
            if (src->cfg.ffm_k) {
                const uint64_t span = 1ull << src->cfg.ffm_bit_precision, per = span / N;
                const uint64_t lo = per * j, cnt = j == N - 1 ? src->ffm_len - lo : per;
                FWGPU_HIP(hipMemcpy(dst->d_ffm_w + lo, src->d_ffm_w + lo, cnt * 4, hipMemcpyDeviceToDevice));
                FWGPU_HIP(hipMemcpy(dst->d_ffm_acc + lo, src->d_ffm_acc + lo, cnt * 4, hipMemcpyDeviceToDevice));
            }
            const uint64_t lper = src->lr_len / N, llo = lper * j, lcnt = j == N - 1 ? src->lr_len - llo : lper;
            FWGPU_HIP(hipMemcpy(dst->d_lr + 2 * llo, src->d_lr + 2 * llo, lcnt * 8, hipMemcpyDeviceToDevice));
        }
    }
    return FWGPU_OK;
}

}  // extern "C"
