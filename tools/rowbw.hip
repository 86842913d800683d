// Microbenchmark: what HBM bandwidth can MI355X sustain on the access pattern of the FFM learn path --
// random rows of R floats (960 B at config C) out of a 1 GiB table, 16 B/lane buffer loads/stores?
//   mode 0: read rows (plain loads)          mode 1: read rows (sc1 loads)
//   mode 2: read w + read acc + write w + write acc (sc1), i.e. the update phase
//   mode 3: streaming read of the whole table (reference point)
// Build: hipcc --offload-arch=gfx950 -O3 tools/rowbw.hip -o tools/rowbw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <int AUX, int U>
__global__ void read_rows(const float *tab, const uint32_t *rows, uint32_t nrows, uint32_t R, float *sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    f4 acc = {0, 0, 0, 0};
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            const uint32_t h = __builtin_amdgcn_readfirstlane(rows[r]);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(tab + h, R * 4), lane * 16, 0, AUX);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __builtin_bit_cast(f4, v[u]);
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int U, int LAUX = 16, int SAUX = 16, bool DO_LOAD = true, bool DO_STORE = true>
__global__ void rmw_rows(float *w, float *a, const uint32_t *rows, uint32_t nrows, uint32_t R) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 vw[U], va[U];
        uint32_t h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            h[u] = __builtin_amdgcn_readfirstlane(rows[r]);
            vw[u] = va[u] = u4{1, 2, 3, (unsigned)i};
            if (DO_LOAD) {
                vw[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(w + h[u], R * 4), lane * 16, 0, LAUX);
                va[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(a + h[u], R * 4), lane * 16, 0, LAUX);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 x = __builtin_bit_cast(f4, vw[u]), y = __builtin_bit_cast(f4, va[u]);
            y += x * x;
            x -= y * 1e-9f;
            if (DO_STORE) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x), rsrc(w + h[u], R * 4), lane * 16, 0, SAUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), rsrc(a + h[u], R * 4), lane * 16, 0, SAUX);
            } else if (x.x == 123.456f) {
                a[0] = y.x;
            }
        }
    }
}

// update phase with L2 float atomics: acc += g^2 (returning), w += -upd (no return); 4 B per lane
template <int U>
__global__ void atomic_rows(float *w, float *a, const uint32_t *rows, uint32_t nrows, uint32_t R) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            const uint32_t h = __builtin_amdgcn_readfirstlane(rows[r]);
            for (uint32_t e = lane; e < R; e += 64) {
                const float old = __hip_atomic_fetch_add(a + h + e, 1e-6f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(w + h + e, -1e-9f * old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- interleaved layout experiment: ONE table, per 8-float block [8 x w][8 x acc] (64 B), so a feature's row is 2R
// contiguous floats (64 B aligned): the forward pass reads the w halves (32 of every 64 B), the update reads the acc halves
// and writes whole 64 B blocks.
template <int AUX, int U>
__global__ void read_rows_il(const float *tab, const uint32_t *rows, uint32_t nrows, uint32_t R, float *sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t voff = (lane >> 1) * 64 + (lane & 1) * 16;  // w half of block lane/2
    f4 acc = {0, 0, 0, 0};
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            const uint32_t h = __builtin_amdgcn_readfirstlane(rows[r]);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(tab + 2 * (size_t)h, R * 8), voff, 0, AUX);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __builtin_bit_cast(f4, v[u]);
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}
template <int U, bool LOAD_W>
__global__ void rmw_rows_il(float *tab, const uint32_t *rows, uint32_t nrows, uint32_t R) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t voff = (lane >> 1) * 64 + (lane & 1) * 16;
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 vw[U], va[U];
        uint32_t h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            h[u] = __builtin_amdgcn_readfirstlane(rows[r]);
            vw[u] = u4{1, 2, 3, (unsigned)i};
            if (LOAD_W) vw[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(tab + 2 * (size_t)h[u], R * 8), voff, 0, 16);
            va[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(tab + 2 * (size_t)h[u], R * 8), voff + 32, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 x = __builtin_bit_cast(f4, vw[u]), y = __builtin_bit_cast(f4, va[u]);
            y += x * x;
            x -= y * 1e-9f;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x), rsrc(tab + 2 * (size_t)h[u], R * 8), voff, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), rsrc(tab + 2 * (size_t)h[u], R * 8), voff + 32, 0, 16);
        }
    }
}
// same access pattern on the current layout (two tables), w kept from the forward pass: read acc, write w and acc
template <int U>
__global__ void upd_rows_sep(float *w, float *a, const uint32_t *rows, uint32_t nrows, uint32_t R) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 va[U];
        uint32_t h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            h[u] = __builtin_amdgcn_readfirstlane(rows[r]);
            va[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(a + h[u], R * 4), lane * 16, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 x = {1.0f, 2.0f, 3.0f, (float)i}, y = __builtin_bit_cast(f4, va[u]);
            y += x * x;
            x -= y * 1e-9f;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x), rsrc(w + h[u], R * 4), lane * 16, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), rsrc(a + h[u], R * 4), lane * 16, 0, 16);
        }
    }
}

__global__ void stream_read(const f4 *tab, size_t n, float *sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += tab[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <typename F>
static float time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const uint32_t R = argc > 1 ? atoi(argv[1]) : 240;
    const size_t tab_floats = (1ull << 28) + R;
    const uint32_t nrows = 3276800;  // 16384 examples x 200 rows
    float *w, *a, *sink;
    uint32_t *rows;
    CK(hipMalloc(&w, (tab_floats + 64) * 4));
    CK(hipMalloc(&a, (tab_floats + 64) * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&rows, nrows * 4));
    CK(hipMemset(w, 0, (tab_floats + 64) * 4));
    CK(hipMemset(a, 0, (tab_floats + 64) * 4));
    const uint32_t align_mask = argc > 3 ? (uint32_t)atoi(argv[3]) - 1 : 7u;  // row start granularity in floats
    printf("row starts aligned to %u floats\n", align_mask + 1);
    std::vector<uint32_t> h(nrows);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        x = (uint32_t)(s >> 20) & ((1u << 28) - 1) & ~align_mask;
    }
    CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
    const double row_bytes = (double)nrows * R * 4;
    printf("R=%u floats (%u B rows), %u random rows per launch, table 1 GiB x2\n", R, R * 4, nrows);
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const f4 *)w, tab_floats / 4, sink); });
        printf("stream read 1 GiB                         : %7.3f ms  %7.1f GB/s\n", ms, tab_floats * 4 / ms / 1e6);
    }
    if (argc > 2) {  // layout experiment only: rowbw 240 il
        float *il;
        CK(hipMalloc(&il, (2 * tab_floats + 256) * 4));
        CK(hipMemset(il, 0, (2 * tab_floats + 256) * 4));
        for (int wpc : {16, 32}) {
            const int blocks = 256 * wpc / 4;
#define T(name, mult, ...)                                                                                   \
    {                                                                                                        \
        float ms = time_ms([&] { __VA_ARGS__; });                                                            \
        printf("%-44s waves/CU=%2d : %7.3f ms  %7.1f GB/s (%d x row bytes)\n", name, wpc, ms, mult * row_bytes / ms / 1e6, mult); \
    }
            T("separate: forward read w", 1, hipLaunchKernelGGL((read_rows<16, 4>), dim3(blocks), dim3(256), 0, 0, w, rows, nrows, R, sink));
            T("interleaved: forward read w halves", 1, hipLaunchKernelGGL((read_rows_il<16, 4>), dim3(blocks), dim3(256), 0, 0, il, rows, nrows, R, sink));
            T("separate: read acc, write w + acc", 3, hipLaunchKernelGGL((upd_rows_sep<4>), dim3(blocks), dim3(256), 0, 0, w, a, rows, nrows, R));
            T("interleaved: read acc, write 64 B blocks", 3, hipLaunchKernelGGL((rmw_rows_il<4, false>), dim3(blocks), dim3(256), 0, 0, il, rows, nrows, R));
            T("separate: read w + acc, write both", 4, hipLaunchKernelGGL((rmw_rows<4>), dim3(blocks), dim3(256), 0, 0, w, a, rows, nrows, R));
            T("interleaved: read + write whole blocks", 4, hipLaunchKernelGGL((rmw_rows_il<4, true>), dim3(blocks), dim3(256), 0, 0, il, rows, nrows, R));
        }
        return 0;
    }
    for (int wpc : {8, 16, 24, 32}) {  // waves per CU
        const int blocks = 256 * wpc / 4;  // 256-thread blocks
#define RUN_READ(AUX, U, name)                                                                                        \
    {                                                                                                                 \
        float ms = time_ms([&] { hipLaunchKernelGGL((read_rows<AUX, U>), dim3(blocks), dim3(256), 0, 0, w, rows, nrows, R, sink); }); \
        printf("%-22s waves/CU=%2d U=%d     : %7.3f ms  %7.1f GB/s\n", name, wpc, U, ms, row_bytes / ms / 1e6);       \
    }
        RUN_READ(0, 1, "read rows plain");
        RUN_READ(0, 4, "read rows plain");
        RUN_READ(0, 8, "read rows plain");
        RUN_READ(16, 1, "read rows sc1");
        RUN_READ(16, 4, "read rows sc1");
        RUN_READ(16, 8, "read rows sc1");
#define RUN_RMW(U)                                                                                                    \
    {                                                                                                                 \
        float ms = time_ms([&] { hipLaunchKernelGGL((rmw_rows<U>), dim3(blocks), dim3(256), 0, 0, w, a, rows, nrows, R); });  \
        printf("%-22s waves/CU=%2d U=%d     : %7.3f ms  %7.1f GB/s (4 x row bytes)\n", "rmw w+acc sc1", wpc, U, ms,   \
               4 * row_bytes / ms / 1e6);                                                                             \
    }
        {
            float ms = time_ms([&] { hipLaunchKernelGGL((atomic_rows<2>), dim3(blocks), dim3(256), 0, 0, w, a, rows, nrows, R); });
            printf("%-22s waves/CU=%2d U=2     : %7.3f ms  %7.1f GB/s (4 x row bytes)\n", "atomic acc+w", wpc, ms, 4 * row_bytes / ms / 1e6);
        }
#define RUN_VAR(LA, SA, DL, DS, name, mult)                                                                           \
    {                                                                                                                 \
        float ms = time_ms([&] { hipLaunchKernelGGL((rmw_rows<2, LA, SA, DL, DS>), dim3(blocks), dim3(256), 0, 0, w, a, rows, nrows, R); }); \
        printf("%-22s waves/CU=%2d U=2     : %7.3f ms  %7.1f GB/s (%d x row bytes)\n", name, wpc, ms, mult * row_bytes / ms / 1e6, mult); \
    }
        if (wpc == 16) {
            RUN_VAR(16, 0, true, true, "rmw ld sc1 st plain", 4);
            RUN_VAR(16, 2, true, true, "rmw ld sc1 st nt", 4);
            RUN_VAR(16, 17, true, true, "rmw ld sc1 st sc0sc1", 4);
            RUN_VAR(0, 0, true, true, "rmw ld plain st plain", 4);
            RUN_VAR(16, 16, true, false, "load w+acc only sc1", 2);
            RUN_VAR(16, 16, false, true, "store w+acc only sc1", 2);
            RUN_VAR(16, 0, false, true, "store w+acc only plain", 2);
        }
        RUN_RMW(1);
        RUN_RMW(2);
        RUN_RMW(4);
    }
    return 0;
}
