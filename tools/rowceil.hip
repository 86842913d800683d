// Microbenchmark 2 (round 2): what is the HBM ceiling of the FFM update pattern on MI355X, and what sets it?
// rowbw.hip (round 1) measured: random 960 B rows read at ~6.4 TB/s, written at ~3.15 TB/s, read+write of w and acc at
// ~4.2 TB/s.  A streaming fill of the same tables runs at 5-6.4 TB/s (ffm_init_kernel / fill_lr_kernel in the round-1
// traces), so the write rate is a property of the PATTERN.  This tool separates the candidates:
//   streaming  : read / write / copy / in-place RMW / two-table RMW (the update pattern with perfect locality)
//   random rows: read, write (one table / two tables), RMW (one / two tables), copy (read row i, write row j),
//                for row lengths 240 and 256 floats and row starts aligned to 32 / 64 / 128 B,
//                and "separate kernels" (all reads, then all writes) against the fused RMW.
// Build: hipcc --offload-arch=gfx950 -O3 tools/rowceil.hip -o tools/rowceil ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// ---- streaming kernels (grid-stride, 16 B per lane)
__global__ void s_read(const f4 *a, size_t n, float *sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}
__global__ void s_write(f4 *a, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = f4{v, v, v, v};
}
__global__ void s_copy(const f4 *a, f4 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void s_rmw1(f4 *a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] * 1.0001f;
}
__global__ void s_rmw2(f4 *a, f4 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 x = a[i], y = b[i];
        y += x * x;
        x -= y * 1e-9f;
        a[i] = x;
        b[i] = y;
    }
}

// ---- random rows.  MODE bits: 1 load w, 2 load a, 4 store w, 8 store a, 16 = stores go to the rows of list 2 (copy)
template <int MODE, int U, int LAUX, int SAUX>
__global__ void rows_k(float *w, float *a, const uint32_t *rows, const uint32_t *rows2, uint32_t nrows, uint32_t R, float *sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    f4 keep = {0, 0, 0, 0};
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 vw[U], va[U];
        uint32_t h[U], h2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            h[u] = __builtin_amdgcn_readfirstlane(rows[r]);
            h2[u] = (MODE & 16) ? __builtin_amdgcn_readfirstlane(rows2[r]) : h[u];
            vw[u] = va[u] = u4{1, 2, 3, (unsigned)i};
            if (MODE & 1) vw[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(w + h[u], R * 4), lane * 16, 0, LAUX);
            if (MODE & 2) va[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(a + h[u], R * 4), lane * 16, 0, LAUX);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 x = __builtin_bit_cast(f4, vw[u]), y = __builtin_bit_cast(f4, va[u]);
            y += x * x;
            x -= y * 1e-9f;
            if (MODE & 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x), rsrc(w + h2[u], R * 4), lane * 16, 0, SAUX);
            if (MODE & 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), rsrc(a + h2[u], R * 4), lane * 16, 0, SAUX);
            if (!(MODE & 12)) keep += x + y;
        }
    }
    if (keep.x + keep.y + keep.z + keep.w == 123.456f) sink[0] = keep.x;
}


// ---- full-line ("window") RMW: the row [4h, 4h+4R) is read and written back as the whole 128 B lines it touches
// (1024 B window from the line-aligned start, plus one more line for the rows that straddle 9 lines).  Bytes outside
// the row are written back unchanged.  Measures what partial-line writes cost.
template <int U, int MODE>  // MODE bit 1: load w, 2: load a (always stores both tables)
__global__ void rows_win(float *w, float *a, const uint32_t *rows, uint32_t nrows, uint32_t R) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 vw[U], va[U], tw[U], ta[U];
        uint32_t s[U], nb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            const uint32_t h = __builtin_amdgcn_readfirstlane(rows[r]);
            s[u] = (h * 4u) & ~127u;                                  // byte offset of the first line
            nb[u] = (((h * 4u + R * 4u + 127u) & ~127u) - s[u]);      // bytes of whole lines covered
            vw[u] = va[u] = tw[u] = ta[u] = u4{1, 2, 3, (unsigned)i};
            __amdgpu_buffer_rsrc_t rw = rsrc((char *)w + s[u], nb[u]), ra = rsrc((char *)a + s[u], nb[u]);
            if (MODE & 1) vw[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, 0, 16);
            if (MODE & 2) va[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, 0, 16);
            if (nb[u] > 1024) {  // wave-uniform
                if (MODE & 1) tw[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, 1024 + lane * 16, 0, 16);
                if (MODE & 2) ta[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, 1024 + lane * 16, 0, 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            __amdgpu_buffer_rsrc_t rw = rsrc((char *)w + s[u], nb[u]), ra = rsrc((char *)a + s[u], nb[u]);
            f4 x = __builtin_bit_cast(f4, vw[u]), y = __builtin_bit_cast(f4, va[u]);
            y += x * x;
            x -= y * 1e-9f;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x), rw, lane * 16, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), ra, lane * 16, 0, 16);
            if (nb[u] > 1024) {
                f4 x2 = __builtin_bit_cast(f4, tw[u]), y2 = __builtin_bit_cast(f4, ta[u]);
                y2 += x2 * x2;
                x2 -= y2 * 1e-9f;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x2), rw, 1024 + lane * 16, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y2), ra, 1024 + lane * 16, 0, 16);
            }
        }
    }
}

// ---- line-interleaved layout experiment: ONE table, 128 B lines alternate w / acc (line 2j = w line j, 2j+1 = acc line j),
// so a row's w AND acc lines form one contiguous span of 2 x (whole lines of the row).  MODE 0: forward read (w lines only),
// MODE 1: update = read the whole span, write the whole span (whole lines).
template <int U, int MODE>
__global__ void rows_il128(float *tab, const uint32_t *rows, uint32_t nrows, uint32_t R, float *sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    f4 keep = {0, 0, 0, 0};
    // lane -> byte offset inside the interleaved span: line (lane / 8) of the w (or acc) array, 16 B piece lane % 8
    const uint32_t off_w = (lane >> 3) * 256 + (lane & 7) * 16, off_a = off_w + 128;
    for (uint32_t i = wave * U; i < nrows; i += nwaves * U) {
        u4 vw[U], va[U];
        uint32_t s[U], nb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = i + u < nrows ? i + u : i;
            const uint32_t h = __builtin_amdgcn_readfirstlane(rows[r]);
            s[u] = ((h * 4u) & ~127u) * 2u;                                   // byte offset of the span in the interleaved table
            nb[u] = ((((h * 4u) & 127u) + R * 4u + 127u) & ~127u) * 2u;       // bytes of the span (<= 2304)
            if (nb[u] > 2048u) nb[u] = 2048u;                                 // (9-line rows: tail ignored in this experiment)
            __amdgpu_buffer_rsrc_t rs = rsrc((char *)tab + s[u], nb[u]);
            vw[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off_w, 0, 16);
            va[u] = u4{1, 2, 3, 4};
            if (MODE == 1) va[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off_a, 0, 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 x = __builtin_bit_cast(f4, vw[u]), y = __builtin_bit_cast(f4, va[u]);
            y += x * x;
            x -= y * 1e-9f;
            if (MODE == 1) {
                __amdgpu_buffer_rsrc_t rs = rsrc((char *)tab + s[u], nb[u]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, x), rs, off_w, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), rs, off_a, 0, 16);
            } else {
                keep += x;
            }
        }
    }
    if (keep.x + keep.y + keep.z + keep.w == 123.456f) sink[0] = keep.x;
}

template <typename F>
static float time_ms(F launch, int reps = 4) {
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
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

static void gen_rows(std::vector<uint32_t> &h, uint64_t seed, uint32_t align_floats, uint32_t span_bits) {
    uint64_t s = seed;
    for (auto &x : h) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        x = (uint32_t)(s >> 20) & ((1u << span_bits) - 1) & ~(align_floats - 1);
    }
}

int main(int argc, char **argv) {
    const size_t tab_floats = (1ull << 28) + 1024;
    const uint32_t nrows = 3276800;  // 16384 examples x 200 rows
    float *w, *a, *sink;
    uint32_t *rows, *rows2;
    CK(hipMalloc(&w, tab_floats * 4));
    CK(hipMalloc(&a, tab_floats * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&rows, nrows * 4));
    CK(hipMalloc(&rows2, nrows * 4));
    CK(hipMemset(w, 0, tab_floats * 4));
    CK(hipMemset(a, 0, tab_floats * 4));
    const double GB = 1e9, tab_bytes = (double)(1ull << 28) * 4;
    const size_t n4 = (1ull << 28) / 4;
    printf("== streaming, 1 GiB tables, 2048 x 256 threads, 16 B/lane (bytes moved / time)\n");
#define S(name, mult, ...)                                                                              \
    {                                                                                                   \
        float ms = time_ms([&] { __VA_ARGS__; });                                                       \
        printf("%-46s : %7.3f ms  %7.1f GB/s\n", name, ms, mult * tab_bytes / ms / 1e6);                 \
    }
    S("stream read", 1, hipLaunchKernelGGL(s_read, dim3(2048), dim3(256), 0, 0, (const f4 *)w, n4, sink));
    S("stream write", 1, hipLaunchKernelGGL(s_write, dim3(2048), dim3(256), 0, 0, (f4 *)w, n4, 0.0f));
    S("stream copy w->a (1R+1W)", 2, hipLaunchKernelGGL(s_copy, dim3(2048), dim3(256), 0, 0, (const f4 *)w, (f4 *)a, n4));
    S("stream rmw in place, one table (1R+1W)", 2, hipLaunchKernelGGL(s_rmw1, dim3(2048), dim3(256), 0, 0, (f4 *)a, n4));
    S("stream rmw w+acc (2R+2W) = update pattern", 4, hipLaunchKernelGGL(s_rmw2, dim3(2048), dim3(256), 0, 0, (f4 *)w, (f4 *)a, n4));
    CK(hipMemset(w, 0, tab_floats * 4));
    CK(hipMemset(a, 0, tab_floats * 4));

    struct Geo { uint32_t R, align; };
    const Geo geos[] = {{240, 8}, {240, 16}, {240, 32}, {256, 32}};
    for (const Geo &g : geos) {
        const uint32_t R = g.R;
        std::vector<uint32_t> h(nrows), h2(nrows);
        gen_rows(h, 88172645463325252ull, g.align, 28);
        gen_rows(h2, 0x9E3779B97F4A7C15ull, g.align, 28);
        CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(rows2, h2.data(), nrows * 4, hipMemcpyHostToDevice));
        const double row_bytes = (double)nrows * R * 4;
        printf("== random rows: R=%u floats (%u B), starts aligned to %u B, %u rows per launch (%.2f GB per table pass)\n", R, R * 4,
               g.align * 4, nrows, row_bytes / GB);
        for (int wpc : {16, 32}) {
            const int blocks = 256 * wpc / 4;
#define T(name, mult, MODE, U, LA, SA)                                                                                            \
    {                                                                                                                             \
        float ms = time_ms([&] { hipLaunchKernelGGL((rows_k<MODE, U, LA, SA>), dim3(blocks), dim3(256), 0, 0, w, a, rows, rows2, nrows, R, sink); }); \
        printf("%-46s waves/CU=%2d U=%d : %7.3f ms  %7.1f GB/s (%d x row bytes)\n", name, wpc, U, ms, mult * row_bytes / ms / 1e6, mult);     \
    }
            T("read w (sc1)", 1, 1, 4, 16, 16);
            T("read w + acc (sc1)", 2, 3, 4, 16, 16);
            T("write w (sc1)", 1, 4, 4, 16, 16);
            T("write w (plain)", 1, 4, 4, 16, 0);
            T("write w + acc (sc1)", 2, 12, 4, 16, 16);
            T("rmw one table: read acc, write acc (sc1)", 2, 10, 4, 16, 16);
            T("copy: read w rows i, write acc rows j (sc1)", 2, 1 | 8 | 16, 4, 16, 16);
            T("read acc, write w + acc (resident-w update)", 3, 14, 4, 16, 16);
            T("rmw w+acc (2R+2W) (sc1)", 4, 15, 4, 16, 16);
            T("rmw w+acc (2R+2W) (sc1) U=8", 4, 15, 8, 16, 16);
            T("rmw w+acc (2R+2W) (plain ld/st)", 4, 15, 4, 0, 0);
#define W(name, mult, U, MODE)                                                                                         \
    {                                                                                                                  \
        float ms = time_ms([&] { hipLaunchKernelGGL((rows_win<U, MODE>), dim3(blocks), dim3(256), 0, 0, w, a, rows, nrows, R); }); \
        printf("%-46s waves/CU=%2d U=%d : %7.3f ms  %7.1f GB/s (%d x ROW bytes; whole lines moved)\n", name, wpc, U, ms, mult * row_bytes / ms / 1e6, mult); \
    }
            W("WINDOW rmw w+acc: whole 128 B lines (sc1)", 4, 4, 3);
            W("WINDOW rmw w+acc: whole 128 B lines U=2", 4, 2, 3);
            W("WINDOW read acc, write w+acc whole lines", 3, 4, 2);
        }
        // "separate kernels": all reads of w+acc, then all writes of w+acc -- the no-overlap reference for the fused rmw
        {
            const int blocks = 256 * 32 / 4;
            float ms = time_ms([&] {
                hipLaunchKernelGGL((rows_k<3, 4, 16, 16>), dim3(blocks), dim3(256), 0, 0, w, a, rows, rows2, nrows, R, sink);
                hipLaunchKernelGGL((rows_k<12, 4, 16, 16>), dim3(blocks), dim3(256), 0, 0, w, a, rows, rows2, nrows, R, sink);
            });
            printf("%-46s waves/CU=32 U=4 : %7.3f ms  %7.1f GB/s (4 x row bytes)\n", "read-all kernel, then write-all kernel", ms,
                   4 * row_bytes / ms / 1e6);
        }
    }
    {   // line-interleaved w/acc layout (one 2 GiB table)
        float *il;
        CK(hipMalloc(&il, tab_floats * 8 + 8192));
        CK(hipMemset(il, 0, tab_floats * 8 + 8192));
        const uint32_t R = 240;
        std::vector<uint32_t> h(nrows);
        gen_rows(h, 88172645463325252ull, 8, 28);
        CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
        const double row_bytes = (double)nrows * R * 4;
        printf("== line-interleaved layout (w line, acc line, w line, ...), R=240, starts aligned to 32 B\n");
        for (int wpc : {16, 32}) {
            const int blocks = 256 * wpc / 4;
#define IL(name, mult, U, MODE)                                                                                        \
    {                                                                                                                  \
        float ms = time_ms([&] { hipLaunchKernelGGL((rows_il128<U, MODE>), dim3(blocks), dim3(256), 0, 0, il, rows, nrows, R, sink); }); \
        printf("%-46s waves/CU=%2d U=%d : %7.3f ms  %7.1f GB/s (%d x ROW bytes)\n", name, wpc, U, ms, mult * row_bytes / ms / 1e6, mult); \
    }
            IL("IL128 forward: read w lines only", 1, 4, 0);
            IL("IL128 update: read + write whole span", 4, 4, 1);
            IL("IL128 update: read + write whole span U=2", 4, 2, 1);
        }
        CK(hipFree(il));
    }
    // locality: the same number of rows drawn from a smaller span of the table (MALL = 256 MiB)
    {
        const uint32_t R = 240;
        for (uint32_t span_bits : {24u, 26u, 28u}) {
            std::vector<uint32_t> h(nrows);
            gen_rows(h, 88172645463325252ull, 8, span_bits);
            CK(hipMemcpy(rows, h.data(), nrows * 4, hipMemcpyHostToDevice));
            const double row_bytes = (double)nrows * R * 4;
            const int wpc = 32, blocks = 256 * wpc / 4;
            printf("== span 2^%u floats per table (%.0f MiB x2)\n", span_bits, (double)(1ull << span_bits) * 4 / 1048576.0);
            T("read w (sc1)", 1, 1, 4, 16, 16);
            T("write w + acc (sc1)", 2, 12, 4, 16, 16);
            T("rmw w+acc (2R+2W) (sc1)", 4, 15, 4, 16, 16);
        }
    }
    return 0;
}
