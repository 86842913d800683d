// TEST INFRASTRUCTURE, not product: an in-tree stand-in for the seven RCCL entry points libfwgpu resolves at run time
// (fwumious_wabbit_amd/csrc/dist.cpp, struct RcclApi), so that the library's process-per-rank path -- shape / count exchange,
// padding to the largest rank, owner ranges, in-place gathers of the tables -- runs under pytest with 2-4 PROCESSES on one GPU.
// Selected with FWGPU_RCCL_LIBRARY=<this .so> before the first fwgpu_dist_* call; nothing in the product links or loads it otherwise.
//
// Transport: a POSIX shared-memory segment named by the "unique id"; every collective is host-synchronous --
//   stream sync, device -> segment slot of this rank, barrier, every rank combines the slots it needs (sums in rank order: the
//   same bits on every rank), host -> device, barrier.  Large messages go through in slot-sized chunks.
// Semantics follow nccl.h: AllGather(send, recv, sendcount), ReduceScatter(send, recv, recvcount), AllReduce(send, recv, count),
// in-place forms included.
// FWGPU_FAKERCCL_ASYNC=1: ncclAllGather behaves like the real thing towards the CALLER -- it returns at once and leaves a kernel on the stream that
// ends when the exchange has happened (a helper thread does the copies and the barriers); with a dead peer that kernel stays on the stream until
// ncclCommAbort, which is what the library's polled waits (FWGPU_DIST_TIMEOUT_MS, dist.cpp wait_stream) exist for and what a host-synchronous
// stand-in cannot exercise.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct FakeComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

namespace {
constexpr size_t kSlot = 16u << 20;  // bytes per rank and chunk
struct Header {
    std::atomic<uint32_t> magic, arrived, generation, attached, aborted;
    uint32_t nranks;
};
}  // namespace

struct FakeComm {
    int rank = 0, n = 1;
    int device = 0;
    std::atomic<int> pending{0};          // asynchronous collectives whose helper thread is still running
    std::atomic<int> async_error{0};      // ncclResult_t of the last failed asynchronous collective (ncclCommGetAsyncError)
    unsigned *flags = nullptr;            // host-mapped completion flags of the asynchronous collectives (ring)
    unsigned flag_next = 0;
    hipStream_t helper_stream = nullptr;
    char name[128] = {0};
    Header *hdr = nullptr;
    unsigned char *slots = nullptr;
    size_t bytes = 0;
    bool barrier() {
        const uint32_t gen = hdr->generation.load(std::memory_order_acquire);
        if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)n) {
            hdr->arrived.store(0, std::memory_order_relaxed);
            hdr->generation.store(gen + 1, std::memory_order_release);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (hdr->generation.load(std::memory_order_acquire) == gen) {
            if (hdr->aborted.load(std::memory_order_acquire)) return false;  // a rank called ncclCommAbort
            std::this_thread::yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;  // a rank died: fail, do not hang the test
        }
        return true;
    }
    unsigned char *slot(int r) { return slots + (size_t)r * kSlot; }
};

static size_t esize(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

template <typename T>
static void sum_into(std::vector<unsigned char> &acc, const unsigned char *src, size_t n, bool first) {
    T *a = reinterpret_cast<T *>(acc.data());
    const T *b = reinterpret_cast<const T *>(src);
    for (size_t i = 0; i < n; i++) a[i] = first ? b[i] : (T)(a[i] + b[i]);
}
static ncclResult_t sum_any(std::vector<unsigned char> &acc, const unsigned char *src, size_t n, ncclDataType_t t, bool first) {
    switch (t) {
    case ncclFloat32: sum_into<float>(acc, src, n, first); return ncclSuccess;
    case ncclFloat64: sum_into<double>(acc, src, n, first); return ncclSuccess;
    case ncclInt32: sum_into<int32_t>(acc, src, n, first); return ncclSuccess;
    case ncclUint32: sum_into<uint32_t>(acc, src, n, first); return ncclSuccess;
    case ncclInt64: sum_into<int64_t>(acc, src, n, first); return ncclSuccess;
    case ncclUint64: sum_into<uint64_t>(acc, src, n, first); return ncclSuccess;
    default: return ncclInvalidArgument;
    }
}

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: shared memory / barrier failure (did a rank die?)";
    case ncclInvalidArgument: return "fake rccl: unsupported datatype or operation";
    default: return "fake rccl: internal error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static std::atomic<unsigned> ctr{0};
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "/fwgpu_fakerccl_%d_%u_%ld", (int)getpid(), ctr++,
             (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    FakeComm *c = new FakeComm();
    c->rank = rank;
    c->n = nranks;
    memcpy(c->name, id.internal, sizeof(c->name));
    c->name[sizeof(c->name) - 1] = 0;
    c->bytes = 4096 + (size_t)nranks * kSlot;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
    } else {
        for (int i = 0; i < 60000 && fd < 0; i++) {  // rank 0 creates the segment
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->bytes)) {
                close(fd);
                fd = -1;
            }
            if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (fd < 0) return ncclSystemError;
    }
    void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    c->hdr = reinterpret_cast<Header *>(m);
    c->slots = reinterpret_cast<unsigned char *>(m) + 4096;
    if (rank == 0) {
        c->hdr->nranks = (uint32_t)nranks;
        c->hdr->arrived.store(0);
        c->hdr->generation.store(0);
        c->hdr->attached.store(0);
        c->hdr->aborted.store(0);
        c->hdr->magic.store(0x46574743u, std::memory_order_release);
    } else {
        for (int i = 0; i < 60000 && c->hdr->magic.load(std::memory_order_acquire) != 0x46574743u; i++)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (c->hdr->magic.load() != 0x46574743u) return ncclSystemError;
    }
    c->hdr->attached.fetch_add(1);
    if (!c->barrier()) return ncclSystemError;
    (void)hipGetDevice(&c->device);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    while (c->pending.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(100));  // helper threads use *c
    if (c->flags) (void)hipHostFree(c->flags);
    if (c->helper_stream) (void)hipStreamDestroy(c->helper_stream);
    const uint32_t left = c->hdr->attached.fetch_sub(1) - 1;
    munmap(reinterpret_cast<void *>(c->hdr), c->bytes);
    if (left == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) {
    if (!c || !count) return ncclInvalidArgument;
    *count = c->n;
    return ncclSuccess;
}

// a failing rank tears the job down: its peers' pending and later collectives return an error instead of waiting
ncclResult_t ncclCommAbort(ncclComm_t c) {
    if (!c) return ncclSuccess;
    c->hdr->aborted.store(1, std::memory_order_release);
    return ncclCommDestroy(c);
}

#define HIPOK(x) do { if ((x) != hipSuccess) return ncclUnhandledCudaError; } while (0)

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *e) {
    if (!c || !e) return ncclInvalidArgument;
    *e = (ncclResult_t)c->async_error.load(std::memory_order_acquire);
    return ncclSuccess;
}

// the kernel an asynchronous collective leaves on the caller's stream: it ends when the helper thread has finished (or given up on) the exchange
__global__ void fake_rccl_wait_kernel(const unsigned *flag) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0) __builtin_amdgcn_s_sleep(64);
}

static ncclResult_t all_gather_body(const void *send, void *recv, size_t total, ncclComm_t c, hipStream_t copy_stream) {
    for (size_t off = 0; off < total || (total == 0 && off == 0); off += kSlot) {
        const size_t m = total - off < kSlot ? total - off : kSlot;
        if (m) {
            HIPOK(hipMemcpyAsync(c->slot(c->rank), (const char *)send + off, m, hipMemcpyDeviceToHost, copy_stream));
            HIPOK(hipStreamSynchronize(copy_stream));
        }
        if (!c->barrier()) return ncclSystemError;
        for (int r = 0; r < c->n && m; r++) HIPOK(hipMemcpyAsync((char *)recv + (size_t)r * total + off, c->slot(r), m, hipMemcpyHostToDevice, copy_stream));
        HIPOK(hipStreamSynchronize(copy_stream));
        if (!c->barrier()) return ncclSystemError;
        if (total == 0) break;
    }
    return ncclSuccess;
}

static ncclResult_t all_gather_async(const void *send, void *recv, size_t total, ncclComm_t c, hipStream_t s) {
    if (!c->flags) {
        HIPOK(hipHostMalloc((void **)&c->flags, 1024 * sizeof(unsigned), hipHostMallocMapped));
        memset(c->flags, 0, 1024 * sizeof(unsigned));
        HIPOK(hipStreamCreateWithFlags(&c->helper_stream, hipStreamNonBlocking));
    }
    unsigned *flag = c->flags + (c->flag_next++ & 1023u);
    __atomic_store_n(flag, 0u, __ATOMIC_RELEASE);
    unsigned *dflag = nullptr;
    HIPOK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    hipEvent_t ready;
    HIPOK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIPOK(hipEventRecord(ready, s));  // the send buffer is complete when this fires
    hipLaunchKernelGGL(fake_rccl_wait_kernel, dim3(1), dim3(1), 0, s, dflag);
    HIPOK(hipGetLastError());
    c->pending.fetch_add(1, std::memory_order_acq_rel);
    std::thread([=]() {
        (void)hipSetDevice(c->device);
        ncclResult_t e = hipEventSynchronize(ready) == hipSuccess ? all_gather_body(send, recv, total, c, c->helper_stream) : ncclUnhandledCudaError;
        (void)hipEventDestroy(ready);
        if (e != ncclSuccess) c->async_error.store((int)e, std::memory_order_release);
        __atomic_store_n(flag, 1u, __ATOMIC_RELEASE);  // the stream goes on -- with the gathered data, or (failure / abort) without it
        c->pending.fetch_sub(1, std::memory_order_acq_rel);
    }).detach();
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    const size_t es = esize(t), total = count * es;
    static const bool async = [] { const char *e = getenv("FWGPU_FAKERCCL_ASYNC"); return e && e[0] == '1'; }();
    if (async) return all_gather_async(send, recv, total, c, s);
    HIPOK(hipStreamSynchronize(s));
    for (size_t off = 0; off < total || (total == 0 && off == 0); off += kSlot) {
        const size_t m = total - off < kSlot ? total - off : kSlot;
        if (m) HIPOK(hipMemcpy(c->slot(c->rank), (const char *)send + off, m, hipMemcpyDeviceToHost));
        if (!c->barrier()) return ncclSystemError;
        for (int r = 0; r < c->n && m; r++) HIPOK(hipMemcpy((char *)recv + (size_t)r * total + off, c->slot(r), m, hipMemcpyHostToDevice));
        if (!c->barrier()) return ncclSystemError;
        if (total == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (op != ncclSum) return ncclInvalidArgument;
    const size_t es = esize(t), total = count * es;
    HIPOK(hipStreamSynchronize(s));
    std::vector<unsigned char> acc(kSlot);
    for (size_t off = 0; off < total; off += kSlot) {
        const size_t m = total - off < kSlot ? total - off : kSlot;
        HIPOK(hipMemcpy(c->slot(c->rank), (const char *)send + off, m, hipMemcpyDeviceToHost));
        if (!c->barrier()) return ncclSystemError;
        for (int r = 0; r < c->n; r++) {
            ncclResult_t e = sum_any(acc, c->slot(r), m / es, t, r == 0);
            if (e != ncclSuccess) return e;
        }
        HIPOK(hipMemcpy((char *)recv + off, acc.data(), m, hipMemcpyHostToDevice));
        if (!c->barrier()) return ncclSystemError;
    }
    return ncclSuccess;
}

// send: n * recvcount elements; rank d receives the sum over ranks of send[d * recvcount ...]
ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (op != ncclSum) return ncclInvalidArgument;
    const size_t es = esize(t), total = recvcount * es;
    const size_t sub = (kSlot / (size_t)c->n) & ~(size_t)63;  // a rank's slot holds one sub-slot per destination
    HIPOK(hipStreamSynchronize(s));
    std::vector<unsigned char> acc(sub);
    for (size_t off = 0; off < total; off += sub) {
        const size_t m = total - off < sub ? total - off : sub;
        for (int d = 0; d < c->n; d++)
            HIPOK(hipMemcpy(c->slot(c->rank) + (size_t)d * sub, (const char *)send + (size_t)d * total + off, m, hipMemcpyDeviceToHost));
        if (!c->barrier()) return ncclSystemError;
        for (int r = 0; r < c->n; r++) {
            ncclResult_t e = sum_any(acc, c->slot(r) + (size_t)c->rank * sub, m / es, t, r == 0);
            if (e != ncclSuccess) return e;
        }
        HIPOK(hipMemcpy((char *)recv + off, acc.data(), m, hipMemcpyHostToDevice));
        if (!c->barrier()) return ncclSystemError;
    }
    return ncclSuccess;
}

}  // extern "C"
