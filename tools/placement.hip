// Microbenchmark 3 (round 2): does it matter WHERE in device memory a table lands?
// scripts/placement_probe.py showed that, inside one process, one regressor out of four runs the same learn launches 9 % faster than
// the others, reproducibly -- with identical relative offsets between its tables.  This tool allocates K buffers of the table size
// and times the same random-row patterns on each of them (and on pairs): random 1 KiB whole-line read-modify-write (the FFM update),
// random 960 B row reads (the gather) and random 8-byte read-modify-writes (the LR block).
// Build: hipcc --offload-arch=gfx950 -O3 tools/placement.hip -o tools/placement ; run on the GPU box: tools/placement [K] [GiB per buffer]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// one wave per row: whole 1 KiB (8 lines) read-modify-write on one or two buffers, device-scope accesses (sc1) like the learn kernel
template <bool TWO>
__global__ void rmw_rows(float *a, float *b, uint32_t lines, uint32_t nrows, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t line = mix(r * 2654435761u + seed) % (lines - 8);
        const size_t off = (size_t)line * 32 + lane * 4;  // floats
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(a + (size_t)line * 32, 0, 1024, 0x00020000);
        u4 x = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, 0, 16);
        u4 y = {0, 0, 0, 0};
        __amdgpu_buffer_rsrc_t rb = ra;
        if (TWO) {
            rb = __builtin_amdgcn_make_buffer_rsrc(b + (size_t)line * 32, 0, 1024, 0x00020000);
            y = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16, 0, 16);
        }
        x.x += 1;
        y.y += x.x;
        __builtin_amdgcn_raw_buffer_store_b128(x, ra, lane * 16, 0, 16);
        if (TWO) __builtin_amdgcn_raw_buffer_store_b128(y, rb, lane * 16, 0, 16);
        (void)off;
    }
}

// the update pattern with the tables striped over several allocations: row r uses pair (r mod npairs)
struct PairSet {
    float *a[8];
    float *b[8];
    int n;
};
__global__ void rmw_rows_multi(PairSet ps, uint32_t lines, uint32_t nrows, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t hsh = mix(r * 2654435761u + seed);
        const uint32_t line = hsh % (lines - 8);
        const int q = (int)((hsh >> 7) % (uint32_t)ps.n);
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(ps.a[q] + (size_t)line * 32, 0, 1024, 0x00020000);
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(ps.b[q] + (size_t)line * 32, 0, 1024, 0x00020000);
        u4 x = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, 0, 16);
        u4 y = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16, 0, 16);
        x.x += 1;
        y.y += x.x;
        __builtin_amdgcn_raw_buffer_store_b128(x, ra, lane * 16, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(y, rb, lane * 16, 0, 16);
    }
}
__global__ void read_rows_multi(PairSet ps, uint32_t lines, uint32_t nrows, uint32_t seed, float *sink) {
    const uint32_t lane = threadIdx.x & 63;
    float acc = 0.f;
    for (uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t hsh = mix(r * 2654435761u + seed);
        const uint32_t start = (hsh % (lines - 8)) * 32 + (mix(r + seed) & 3) * 8;
        const int q = (int)((hsh >> 7) % (uint32_t)ps.n);
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(ps.a[q] + start, 0, 960, 0x00020000);
        u4 x = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, 0, 16);
        acc += __uint_as_float(x.x);
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void read_rows(const float *a, uint32_t lines, uint32_t nrows, uint32_t seed, float *sink) {
    const uint32_t lane = threadIdx.x & 63;
    float acc = 0.f;
    for (uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t start = (mix(r * 2654435761u + seed) % (lines - 8)) * 32 + (mix(r + seed) & 3) * 8;  // 32 B aligned rows
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a) + start, 0, 960, 0x00020000);
        u4 x = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, 0, 16);
        acc += __uint_as_float(x.x);
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void rmw_pairs(unsigned long long *t, uint32_t entries, uint32_t n, uint32_t seed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned long long *p = t + (mix(i * 2654435761u + seed) % entries);
        unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename F>
static float time_ms(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    const double gib = argc > 2 ? atof(argv[2]) : 1.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) + (4u << 20);  // the FFM table of config C is 1 GiB + the spill-over tail
    const uint32_t lines = (uint32_t)(bytes / 128);
    std::vector<float *> buf(K);
    for (int i = 0; i < K; i++) {
        CK(hipMalloc((void **)&buf[i], bytes));
        CK(hipMemset(buf[i], 0, bytes));
    }
    float *sink;
    CK(hipMalloc((void **)&sink, 64));
    const uint32_t nrows = 3300000;  // rows per 16 384-example launch
    const dim3 grid(256 * 12), block(256);
    printf("K = %d buffers of %.2f GiB; per buffer: RMW of %u random 1 KiB windows, read of %u random 960 B rows, %u random 8 B RMWs\n", K,
           bytes / double(1 << 30), nrows, nrows, nrows);
    for (int rep = 0; rep < (argc > 3 ? 0 : 2); rep++)
        for (int i = 0; i < K; i++) {
            const float t_rmw = time_ms([&] { hipLaunchKernelGGL(rmw_rows<false>, grid, block, 0, 0, buf[i], buf[i], lines, nrows, 17u); }, 5);
            const float t_rd = time_ms([&] { hipLaunchKernelGGL(read_rows, grid, block, 0, 0, buf[i], lines, nrows, 29u, sink); }, 5);
            const float t_lr = time_ms([&] { hipLaunchKernelGGL(rmw_pairs, grid, block, 0, 0, (unsigned long long *)buf[i], (uint32_t)(bytes / 8), nrows, 31u); }, 5);
            printf("pass %d buffer %d at %p: window RMW %.3f ms (%.2f TB/s)  row read %.3f ms (%.2f TB/s)  8 B RMW %.3f ms\n", rep, i, (void *)buf[i], t_rmw,
                   2.0 * nrows * 1024 / t_rmw / 1e9, t_rd, nrows * 960.0 / t_rd / 1e9, t_lr);
        }
    if (argc > 3 && argv[3][0] == 'r') {  // regions: classify the buffers, then stripe the tables over 1 / 2 / 3 / 4 regions
        std::vector<int> rep, cls(K, -1);
        std::vector<std::vector<int>> members;
        float fast = 1e9f, slow = 0.f;
        for (int j = 0; j < K; j++) {
            for (size_t c = 0; c < rep.size() && cls[j] < 0; c++) {
                const float t = time_ms([&] { hipLaunchKernelGGL(rmw_rows<true>, grid, block, 0, 0, buf[rep[c]], buf[j], lines, 600000u, 17u); }, 2);
                if (t > 0.415f) cls[j] = (int)c;
                fast = t < fast ? t : fast;
                slow = t > slow ? t : slow;
            }
            if (cls[j] < 0) {
                cls[j] = (int)rep.size();
                rep.push_back(j);
                members.emplace_back();
            }
            members[cls[j]].push_back(j);
        }
        printf("regions found among %d buffers: %zu (pair probe %.3f - %.3f ms); classes: ", K, rep.size(), fast, slow);
        for (int j = 0; j < K; j++) printf("%d", cls[j]);
        printf("\n");
        std::vector<int> big;  // regions with at least 2 members
        for (size_t c = 0; c < members.size(); c++)
            if (members[c].size() >= 2) big.push_back((int)c);
        printf("regions with >= 2 buffers: %zu\n", big.size());
        auto run = [&](const char *name, std::vector<std::pair<int, int>> pairs) {
            PairSet ps;
            ps.n = (int)pairs.size();
            for (int q = 0; q < ps.n; q++) {
                ps.a[q] = buf[pairs[q].first];
                ps.b[q] = buf[pairs[q].second];
            }
            const float t = time_ms([&] { hipLaunchKernelGGL(rmw_rows_multi, grid, block, 0, 0, ps, lines, nrows, 17u); }, 4);
            const float tr = time_ms([&] { hipLaunchKernelGGL(read_rows_multi, grid, block, 0, 0, ps, lines, nrows, 29u, sink); }, 4);
            printf("%-64s two-table RMW %.3f ms (%.2f TB/s)   row reads of the first table %.3f ms (%.2f TB/s)\n", name, t, 4.0 * nrows * 1024 / t / 1e9, tr,
                   nrows * 960.0 / tr / 1e9);
        };
        if (big.size() >= 2) {
            const int A0 = members[big[0]][0], A1 = members[big[0]][1], B0 = members[big[1]][0], B1 = members[big[1]][1];
            run("same region (w = A0, acc = A1)", {{A0, A1}});
            run("two regions (w = A0, acc = B0)", {{A0, B0}});
            run("two regions, balanced ((A0,B0), (B1,A1))", {{A0, B0}, {B1, A1}});
            if (big.size() >= 3) {
                const int C0 = members[big[2]][0], C1 = members[big[2]][1];
                run("three regions ((A0,B0), (B1,C0), (C1,A1))", {{A0, B0}, {B1, C0}, {C1, A1}});
                if (big.size() >= 4) {
                    const int D0 = members[big[3]][0], D1 = members[big[3]][1];
                    run("four regions ((A0,B0), (B1,C0), (C1,D0), (D1,A1))", {{A0, B0}, {B1, C0}, {C1, D0}, {D1, A1}});
                    run("four regions, w in A+C, acc in B+D ((A0,B0), (C0,D0))", {{A0, B0}, {C0, D0}});
                }
            }
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'p') {  // pads: both tables in ONE allocation, the second `pad` bytes behind the end of the first
        float *blk;
        const size_t maxpad = 3ull << 30;
        CK(hipMalloc((void **)&blk, 2 * bytes + maxpad));
        CK(hipMemset(blk, 0, 2 * bytes + maxpad));
        const size_t pads[] = {0, 128, 256, 4096, 4096 + 128, 37 * 128, 65536, 65536 + 128 * 3, 1 << 20, (1 << 20) + 128 * 5, 2 << 20, (2 << 20) + 4096 + 128,
                               16 << 20, (16 << 20) + 128 * 77, 1ull << 30, (1ull << 30) + 128 * 1237, (1ull << 30) + (1 << 20), 0x12345680ull, 0x7654300ull,
                               3 * 4096, 5 * 65536, 7ull << 20, 11ull << 20, 513ull << 20, 1500ull << 20, 2047ull << 20};
        for (size_t pad : pads) {
            float *b2 = (float *)((char *)blk + bytes + pad);
            const float t = time_ms([&] { hipLaunchKernelGGL(rmw_rows<true>, grid, block, 0, 0, blk, b2, lines, 600000u, 17u); }, 3);
            printf("pad %12zu (0x%zx): %.3f ms\n", pad, pad, t);
        }
        const float t1 = time_ms([&] { hipLaunchKernelGGL(rmw_rows<true>, grid, block, 0, 0, blk, buf[0], lines, 600000u, 17u); }, 3);
        const float t2 = time_ms([&] { hipLaunchKernelGGL(rmw_rows<true>, grid, block, 0, 0, blk, buf[K - 1], lines, 600000u, 17u); }, 3);
        printf("block vs buffer 0: %.3f ms, vs buffer %d: %.3f ms\n", t1, K - 1, t2);
        return 0;
    }
    if (argc > 3) {  // scan: every buffer against a few reference buffers only
        for (int ref : {0, K / 3, (2 * K) / 3, K - 1}) {
            printf("vs buffer %2d:", ref);
            for (int j = 0; j < K; j++) {
                if (j == ref) {
                    printf("   -- ");
                    continue;
                }
                const float t = time_ms([&] { hipLaunchKernelGGL(rmw_rows<true>, grid, block, 0, 0, buf[ref], buf[j], lines, 600000u, 17u); }, 2);
                printf(" %.3f", t);
            }
            printf("\n");
        }
        return 0;
    }
    printf("pairs (w = buffer i, acc = buffer j), window RMW on both, ms:\n");
    for (int i = 0; i < K; i++) {
        printf("  w=%d:", i);
        for (int j = 0; j < K; j++) {
            if (i == j) {
                printf("    --  ");
                continue;
            }
            const float t = time_ms([&] { hipLaunchKernelGGL(rmw_rows<true>, grid, block, 0, 0, buf[i], buf[j], lines, nrows, 17u); }, 3);
            printf("  %.3f", t);
        }
        printf("\n");
    }
    return 0;
}
