// Row-sparse gradient buckets (gfx950): the multi-GPU mode in which every GPU keeps a full replica of the tables and the ranks
// exchange, per micro-batch, the DEDUPLICATED gradients of the rows their examples touched (north_star: "RCCL all-reduce of
// sparse gradient buckets"; hogwild.rs:24-103 is the reference's shared-table trainer this replaces across GPUs).
//
// One step, every rank on its own micro-batch (weights frozen for the whole global batch):
//   FWD     (kernels.hip, phase 1)  field sums / LR sums into the split records + one OCCURRENCE per entry:
//                                   key = (row hash << 32 | slot), slot = example * max_entries + entry, and {value, field}
//   MID     (kernels.hip)           logit, prediction, general gradient g per example
//   sort    (rocPRIM radix sort)    occurrences of a row become adjacent, in (example, entry) order
//   reduce  (here)                  one wave per 64 sorted occurrences: the gradient rows g*v*(T[f] - own slot) of equal hashes are
//                                   summed into one bucket row -> at most one extra bucket row per 64 occurrences of a hot row
//   exchange (dist.cpp)             all-gather of {bucket keys, bucket rows}, every rank then holds all ranks' buckets
//   merge   (radix sort)            bucket rows of all ranks by (hash, rank, index)
//   apply   (here)                  per row: G = sum of its bucket rows in that order, ONE optimizer step with G (optimizer.rs);
//                                   rows may overlap (block_ffm.rs:92-94 -- they start at hash & mask and are R long), so rows are
//                                   grouped by block = hash / R: a wave applies the rows of one block in ascending hash order,
//                                   all even blocks first, then all odd ones (rows of two different even blocks never intersect).
// Everything after FWD is a deterministic function of the global batch: replicas that start identical stay bit-identical, with no
// table exchange at all.  What changes against the reference is the update rule: one AdaGrad step per row and micro-batch with the
// summed gradient instead of one step per occurrence (the oracle's fwo_learn_sparse restates exactly this rule).
//
// All of this is HBM-bound row traffic (960 B rows at config C): coalesced lane-per-float accesses, no LDS, no MFMA.
#include "fwgpu_device.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace fwgpu {

namespace {

constexpr unsigned long long kNoKey = ~0ull;

__device__ __forceinline__ float ld_dev(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flags[i] = 1 where a bucket row starts: first valid key of a 64-block of the sorted list, or a new hash.  flags[n] = 0 (so that
// the exclusive scan's element n is the number of bucket rows).
__global__ void sparse_flags_kernel(const unsigned long long *keys, uint32_t n, uint32_t *flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t f = 0;
    if (i < n) {
        const unsigned long long k = keys[i];
        if (k != kNoKey) f = (i & 63u) == 0 || (uint32_t)(keys[i - 1] >> 32) != (uint32_t)(k >> 32);
    }
    flags[i] = f;
}

// FFM: one wave per 64 sorted occurrences.  Lane t fetches occurrence t's description once; the wave then walks the block,
// four occurrences' row loads in flight at a time (they are independent of where the runs end), and adds every run of equal
// hashes in order into one bucket row.
__global__ __launch_bounds__(256) void sparse_reduce_ffm_kernel(const unsigned long long *keys, uint32_t n, const uint32_t *pos,
                                                                const uint2 *desc, uint32_t max_ffm, const float *split,
                                                                uint32_t split_len, const float *selfw, uint32_t selfw_stride,
                                                                const float *gbuf, uint32_t R, uint32_t k, uint32_t *bk_key,
                                                                float *bk_rows) {
    const uint32_t lane = threadIdx.x & 63, base = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (base >= n) return;
    const uint32_t i = base + lane;
    const unsigned long long key = i < n ? keys[i] : kNoKey;
    const bool valid = key != kNoKey;
    const int cnt = __popcll(__ballot(valid));  // valid keys sort first: lanes [0, cnt)
    if (!cnt) return;
    const uint32_t m_h = (uint32_t)(key >> 32);
    uint32_t m_ex = 0, m_ii = 0, m_f = 0, m_out = 0;
    float m_v = 0.0f, m_g = 0.0f;
    if (valid) {
        const uint32_t slot = (uint32_t)key;
        m_ex = slot / max_ffm;
        m_ii = slot - m_ex * max_ffm;
        const uint2 d = desc[slot];
        m_v = __uint_as_float(d.x);
        m_f = d.y;
        m_g = gbuf[m_ex];
        m_out = pos[i];  // (meaningful on run heads)
    }
    for (uint32_t c0 = 0; c0 < R; c0 += 256) {
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t cur_h = __shfl(m_h, 0, 64), cur_out = __shfl(m_out, 0, 64);
        auto flush = [&]() {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = c0 + u * 64 + lane;
                if (e < R) bk_rows[(size_t)cur_out * R + e] = sum[u];
                sum[u] = 0.0f;
            }
            if (lane == 0 && c0 == 0) bk_key[cur_out] = cur_h;
        };
        for (int t0 = 0; t0 < cnt; t0 += 4) {
            float tv[4][4], sw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = min(t0 + q, cnt - 1);
                const uint32_t ex = __shfl(m_ex, t, 64), ii = __shfl(m_ii, t, 64), f = __shfl(m_f, t, 64);
                const float *trow = split + (size_t)ex * split_len + f * R;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t e = c0 + u * 64 + lane;
                    tv[q][u] = e < R ? trow[e] : 0.0f;
                }
                // (the own-slot correction touches the k floats of field f only)
                const uint32_t es = f * k + (lane < k ? lane : 0);
                sw[q] = (lane < k && es >= c0 && es < c0 + 256) ? selfw[(size_t)ex * selfw_stride + ii * k + lane] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = t0 + q;
                if (t >= cnt) break;
                const uint32_t h = __shfl(m_h, t, 64), f = __shfl(m_f, t, 64);
                const float v = __shfl(m_v, t, 64), g = __shfl(m_g, t, 64);
                if (h != cur_h) {
                    flush();
                    cur_h = h;
                    cur_out = __shfl(m_out, t, 64);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t e = c0 + u * 64 + lane;
                    const uint32_t z = e / k;
                    const float swx = __shfl(sw[q], (e - z * k) & 63u, 64);  // (taken by all lanes: a lane exchange under a lane mask reads zeros)
                    if (e < R) {
                        float x = tv[q][u];
                        if (z == f) x = __fsub_rn(x, __fmul_rn(swx, v));  // block_ffm.rs:238
                        const float G = __fmul_rn(v, x);                  // block_ffm.rs:239, 249
                        sum[u] = __fadd_rn(sum[u], __fmul_rn(g, G));      // block_ffm.rs:278, summed over the row's occurrences
                    }
                }
            }
        }
        flush();
    }
}

// The same for rows of 16-byte aligned float4s (k % 4 == 0: every row starts on a 16 B boundary and a lane's four floats lie in one
// field slot): one 16 B load per lane and occurrence instead of four 4 B loads.  Same arithmetic, same order.
__global__ __launch_bounds__(256) void sparse_reduce_ffm4_kernel(const unsigned long long *keys, uint32_t n, const uint32_t *pos,
                                                                 const uint2 *desc, uint32_t max_ffm, const float *split,
                                                                 uint32_t split_len, const float *selfw, uint32_t selfw_stride,
                                                                 const float *gbuf, uint32_t R, uint32_t k, uint32_t *bk_key,
                                                                 float *bk_rows) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63, base = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (base >= n) return;
    const uint32_t i = base + lane;
    const unsigned long long key = i < n ? keys[i] : kNoKey;
    const bool valid = key != kNoKey;
    const int cnt = __popcll(__ballot(valid));
    if (!cnt) return;
    const uint32_t m_h = (uint32_t)(key >> 32);
    uint32_t m_ex = 0, m_ii = 0, m_f = 0, m_out = 0;
    float m_v = 0.0f, m_g = 0.0f;
    if (valid) {
        const uint32_t slot = (uint32_t)key;
        m_ex = slot / max_ffm;
        m_ii = slot - m_ex * max_ffm;
        const uint2 d = desc[slot];
        m_v = __uint_as_float(d.x);
        m_f = d.y;
        m_g = gbuf[m_ex];
        m_out = pos[i];
    }
    for (uint32_t c0 = 0; c0 < R; c0 += 256) {
        const uint32_t e = c0 + lane * 4;  // this lane's four floats [e, e + 4)
        const bool on = e < R;
        const uint32_t z = e / k, kk = e - z * k;
        f4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t cur_h = __shfl(m_h, 0, 64), cur_out = __shfl(m_out, 0, 64);
        auto flush = [&]() {
            if (on) *reinterpret_cast<f4 *>(bk_rows + (size_t)cur_out * R + e) = sum;
            sum = f4{0.0f, 0.0f, 0.0f, 0.0f};
            if (lane == 0 && c0 == 0) bk_key[cur_out] = cur_h;
        };
        for (int t0 = 0; t0 < cnt; t0 += 4) {
            f4 tv[4];
            float sw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = min(t0 + q, cnt - 1);
                const uint32_t ex = __shfl(m_ex, t, 64), ii = __shfl(m_ii, t, 64), f = __shfl(m_f, t, 64);
                const float *trow = split + (size_t)ex * split_len + f * R;
                tv[q] = on ? *reinterpret_cast<const f4 *>(trow + e) : f4{0.0f, 0.0f, 0.0f, 0.0f};
                const uint32_t es = f * k + (lane < k ? lane : 0);
                sw[q] = (lane < k && es >= c0 && es < c0 + 256) ? selfw[(size_t)ex * selfw_stride + ii * k + lane] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = t0 + q;
                if (t >= cnt) break;
                const uint32_t h = __shfl(m_h, t, 64), f = __shfl(m_f, t, 64);
                const float v = __shfl(m_v, t, 64), g = __shfl(m_g, t, 64);
                if (h != cur_h) {
                    flush();
                    cur_h = h;
                    cur_out = __shfl(m_out, t, 64);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float swx = __shfl(sw[q], (kk + j) & 63u, 64);  // (taken by all lanes)
                    float x = tv[q][j];
                    if (z == f) x = __fsub_rn(x, __fmul_rn(swx, v));  // block_ffm.rs:238
                    const float G = __fmul_rn(v, x);                  // block_ffm.rs:239, 249
                    sum[j] = __fadd_rn(sum[j], __fmul_rn(g, G));      // block_ffm.rs:278
                }
            }
        }
        flush();
    }
}

// LR: one wave per 64 sorted occurrences (gradient = g * value, block_lr.rs:135-150).  Every lane fetches its own occurrence;
// the run heads then add their run in order (lane broadcasts), so the sums do not depend on how the work is spread.
__global__ __launch_bounds__(256) void sparse_reduce_lr_kernel(const unsigned long long *keys, uint32_t n, const uint32_t *pos,
                                                               const uint2 *desc, uint32_t max_lr, const float *gbuf,
                                                               uint32_t *bk_key, float *bk_val) {
    const uint32_t lane = threadIdx.x & 63, base = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (base >= n) return;
    const uint32_t i = base + lane;
    const unsigned long long key = i < n ? keys[i] : kNoKey;
    const bool valid = key != kNoKey;
    const uint32_t h = (uint32_t)(key >> 32), slot = (uint32_t)key;
    const float grad = valid ? __fmul_rn(gbuf[slot / max_lr], __uint_as_float(desc[slot].x)) : 0.0f;
    const uint32_t hprev = __shfl_up(h, 1, 64);
    const bool head = valid && (lane == 0 || hprev != h);
    float sum = 0.0f;
    const unsigned long long vmask = __ballot(valid);
    const int cnt = __popcll(vmask);  // valid keys sort first: lanes [0, cnt)
    for (int t = 0; t < cnt; ++t) {
        const uint32_t ht = __shfl(h, t, 64);
        const float gt = __shfl(grad, t, 64);
        if (head && ht == h) sum = __fadd_rn(sum, gt);  // (sorted: the lanes with this hash are exactly the run that starts here)
    }
    if (head) {
        const uint32_t out = pos[i];
        bk_key[out] = h;
        bk_val[out] = sum;
    }
}

// merged key list of the gathered buckets: index = rank * stride + u
__global__ void sparse_merge_keys_kernel(const uint32_t *all_key, const uint32_t *counts, uint32_t n_ranks, uint32_t stride,
                                         unsigned long long *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ranks * stride) return;
    const uint32_t r = i / stride, u = i - r * stride;
    keys[i] = u < counts[r] ? (((unsigned long long)all_key[i] << 32) | i) : kNoKey;
}

// FFM apply: one wave per element of the merged sorted list; the wave of a block's first element applies the whole block.
template <int OPT>
__global__ __launch_bounds__(256) void sparse_apply_ffm_kernel(const unsigned long long *keys, uint32_t n, const float *rows,
                                                               uint32_t R, uint32_t parity, float *ffm_w, float *ffm_acc,
                                                               float rate, float minus_power_t, const float *lut) {
    const uint32_t lane = threadIdx.x & 63, i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const unsigned long long ki = keys[i];
    if (ki == kNoKey) return;
    const uint32_t blk = (uint32_t)(ki >> 32) / R;
    if ((blk & 1u) != parity) return;
    if (i > 0 && (uint32_t)(keys[i - 1] >> 32) / R == blk) return;  // not the block's first element
    uint32_t j = i;
    while (j < n) {
        const unsigned long long kj = keys[j];
        if (kj == kNoKey) break;
        const uint32_t h = (uint32_t)(kj >> 32);
        if (h / R != blk) break;
        uint32_t j2 = j + 1;
        while (j2 < n && (uint32_t)(keys[j2] >> 32) == h) ++j2;
        for (uint32_t c0 = 0; c0 < R; c0 += 256) {  // four 64-float chunks of the row in flight
            float G[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w[4], acc[4];
            for (uint32_t t = j; t < j2; ++t) {
                const float *row = rows + (size_t)(uint32_t)keys[t] * R;
                float x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t e = c0 + u * 64 + lane;
                    x[u] = e < R ? row[e] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) G[u] = __fadd_rn(G[u], x[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = c0 + u * 64 + lane;
                const bool on = e < R && G[u] != 0.0f;  // (a zero gradient changes nothing in any of the optimizers)
                acc[u] = (on && OPT != FWGPU_OPT_SGD) ? ld_dev(ffm_acc + h + e) : 0.0f;
                w[u] = on ? ld_dev(ffm_w + h + e) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t e = c0 + u * 64 + lane;
                if (e < R && G[u] != 0.0f) {
                    const float upd = opt_step<OPT>(G[u], acc[u], rate, minus_power_t, lut);
                    st_dev(ffm_w + h + e, w[u] - upd);  // block_ffm.rs:282
                    if (OPT != FWGPU_OPT_SGD) st_dev(ffm_acc + h + e, acc[u]);
                }
            }
        }
        // the next row of the block may overlap this one: its lanes must read what these stores wrote.  Both sides are
        // device-scope (sc1) accesses that meet in this XCD's L2; the stores only have to be acknowledged first.  (A release
        // fence here would write the whole L2 back per row: measured 2.8 ms per launch.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        j = j2;
    }
}

// The same with 16 B accesses (k % 4 == 0): device-scope (sc1) buffer loads / stores of float4, as the example kernels use.
// One wave per 64 consecutive elements of the merged list: the lanes find the block heads of this parity among them, the wave then
// applies those blocks one after the other (a launch with one wave per ELEMENT spends most of its time starting waves that exit).
template <int OPT>
__global__ __launch_bounds__(256) void sparse_apply_ffm4_kernel(const unsigned long long *keys, uint32_t n, const float *rows,
                                                                uint32_t R, uint32_t parity, float *ffm_w, float *ffm_acc,
                                                                float rate, float minus_power_t, const float *lut) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63, base = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
    if (base >= n) return;
    const uint32_t i = base + lane;
    bool head = false;
    if (i < n) {
        const unsigned long long ki = keys[i];
        if (ki != kNoKey) {
            const uint32_t blk = (uint32_t)(ki >> 32) / R;
            head = (blk & 1u) == parity && (i == 0 || (uint32_t)(keys[i - 1] >> 32) / R != blk);
        }
    }
    unsigned long long heads = __ballot(head);
    while (heads) {
        const uint32_t first = (uint32_t)__builtin_ctzll(heads);
        heads &= heads - 1;
        uint32_t j = base + first;
        const uint32_t blk = (uint32_t)(keys[j] >> 32) / R;
        while (j < n) {
            const unsigned long long kj = keys[j];
            if (kj == kNoKey) break;
            const uint32_t h = (uint32_t)(kj >> 32);
            if (h / R != blk) break;
            uint32_t j2 = j + 1;
            while (j2 < n && (uint32_t)(keys[j2] >> 32) == h) ++j2;
            __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(ffm_w + h, 0, (int)(R * 4), 0x00020000);
            __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(ffm_acc + h, 0, (int)(R * 4), 0x00020000);
            for (uint32_t c0 = 0; c0 < R; c0 += 256) {
                const uint32_t e = c0 + lane * 4;
                const bool on = e < R;
                // (the table loads do not depend on the bucket rows: issued first, they overlap them)
                f4 w = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(e * 4), 0, 16));
                f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                if (OPT != FWGPU_OPT_SGD) acc = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(e * 4), 0, 16));
                f4 G = {0.0f, 0.0f, 0.0f, 0.0f};
                for (uint32_t t = j; t < j2; ++t) {
                    const float *row = rows + (size_t)(uint32_t)keys[t] * R;
                    const f4 x = on ? *reinterpret_cast<const f4 *>(row + e) : f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) G[q] = __fadd_rn(G[q], x[q]);
                }
                // (lanes beyond the row read zeros and their stores are dropped: raw buffer of 4R bytes)
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (G[q] != 0.0f) {  // (a zero gradient changes nothing in any of the optimizers)
                        float a = acc[q];
                        const float upd = opt_step<OPT>(G[q], a, rate, minus_power_t, lut);
                        acc[q] = a;
                        w[q] = w[q] - upd;  // block_ffm.rs:282
                        any = true;
                    }
                }
                if (on && any) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, w), rw, (int)(e * 4), 0, 16);
                    if (OPT != FWGPU_OPT_SGD) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc), ra, (int)(e * 4), 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next row of the block may overlap this one (see the scalar kernel)
            j = j2;
        }
    }
}

// LR apply: one thread per element; the thread of a hash's first element sums the run and takes the step.
template <int OPT>
__global__ void sparse_apply_lr_kernel(const unsigned long long *keys, uint32_t n, const float *vals, float *lr, float rate,
                                       float minus_power_t, const float *lut) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long ki = keys[i];
    if (ki == kNoKey) return;
    const uint32_t h = (uint32_t)(ki >> 32);
    if (i > 0 && (uint32_t)(keys[i - 1] >> 32) == h) return;
    float G = 0.0f;
    for (uint32_t t = i; t < n; ++t) {
        const unsigned long long kt = keys[t];
        if ((uint32_t)(kt >> 32) != h || kt == kNoKey) break;
        G = __fadd_rn(G, vals[(uint32_t)kt]);
    }
    if (G == 0.0f) return;
    float2 *p = reinterpret_cast<float2 *>(lr) + h;
    float2 wa = *p;
    wa.x -= opt_step<OPT>(G, wa.y, rate, minus_power_t, lut);  // block_lr.rs:141-148
    *p = wa;
}

hipError_t sort_keys(void *tmp, size_t tmp_bytes, const unsigned long long *in, unsigned long long *out, uint32_t n, int end_bit,
                     hipStream_t stream) {
    size_t need = tmp_bytes;
    return rocprim::radix_sort_keys(tmp, need, in, out, (size_t)n, 0u, (unsigned)end_bit, stream);
}

}  // namespace

size_t sparse_tmp_bytes(uint32_t n_max) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_keys(nullptr, a, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n_max, 0u, 64u, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n_max + 1, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return (std::max(a, b) + 255) & ~(size_t)255;
}

// sorted occurrence keys -> bucket rows.  flags/pos: n + 1 elements; *d_count receives the number of bucket rows (device).
hipError_t sparse_reduce(const SparseReduceArgs &a, hipStream_t stream) {
    if (!a.n) return hipMemsetAsync(a.d_count, 0, 4, stream);
    hipError_t e = sort_keys(a.tmp, a.tmp_bytes, a.keys, a.keys_sorted, a.n, a.key_bits, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sparse_flags_kernel, dim3((a.n + 256) / 256), dim3(256), 0, stream, a.keys_sorted, a.n, a.flags);
    size_t need = a.tmp_bytes;
    e = rocprim::exclusive_scan(a.tmp, need, a.flags, a.pos, 0u, (size_t)a.n + 1, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(a.d_count, a.pos + a.n, 4, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    const uint32_t blocks64 = (a.n + 63) / 64;
    if (a.R) {
        if (a.k % 4 == 0 && a.split_len % 4 == 0)
            hipLaunchKernelGGL(sparse_reduce_ffm4_kernel, dim3((blocks64 + 3) / 4), dim3(256), 0, stream, a.keys_sorted, a.n, a.pos, a.desc,
                               a.max_entries, a.split, a.split_len, a.selfw, a.selfw_stride, a.gbuf, a.R, a.k, a.bk_key, a.bk_rows);
        else
            hipLaunchKernelGGL(sparse_reduce_ffm_kernel, dim3((blocks64 + 3) / 4), dim3(256), 0, stream, a.keys_sorted, a.n, a.pos, a.desc,
                               a.max_entries, a.split, a.split_len, a.selfw, a.selfw_stride, a.gbuf, a.R, a.k, a.bk_key, a.bk_rows);
    } else {
        hipLaunchKernelGGL(sparse_reduce_lr_kernel, dim3((blocks64 + 3) / 4), dim3(256), 0, stream, a.keys_sorted, a.n, a.pos, a.desc,
                           a.max_entries, a.gbuf, a.bk_key, a.bk_rows);
    }
    return hipGetLastError();
}

template <int OPT>
static hipError_t sparse_apply_t(const SparseApplyArgs &a, hipStream_t stream) {
    const uint32_t n = a.n_ranks * a.stride;
    if (!n) return hipSuccess;
    // (one rank: its bucket list is sorted by (hash, index) as it stands)
    hipLaunchKernelGGL(sparse_merge_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a.all_key, a.counts, a.n_ranks, a.stride,
                       a.n_ranks == 1 ? a.keys_sorted : a.keys);
    if (a.n_ranks > 1) {
        hipError_t e = sort_keys(a.tmp, a.tmp_bytes, a.keys, a.keys_sorted, n, a.key_bits, stream);
        if (e != hipSuccess) return e;
    }
    if (a.R) {
        for (uint32_t parity = 0; parity < 2; ++parity)
            if (a.k4)
                hipLaunchKernelGGL(sparse_apply_ffm4_kernel<OPT>, dim3(((n + 63) / 64 + 3) / 4), dim3(256), 0, stream, a.keys_sorted, n, a.all_rows, a.R, parity,
                                   a.w, a.acc, a.rate, a.minus_power_t, a.lut);
            else
                hipLaunchKernelGGL(sparse_apply_ffm_kernel<OPT>, dim3((n + 3) / 4), dim3(256), 0, stream, a.keys_sorted, n, a.all_rows, a.R, parity,
                                   a.w, a.acc, a.rate, a.minus_power_t, a.lut);
    } else {
        hipLaunchKernelGGL(sparse_apply_lr_kernel<OPT>, dim3((n + 255) / 256), dim3(256), 0, stream, a.keys_sorted, n, a.all_rows, a.w, a.rate,
                           a.minus_power_t, a.lut);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ table placement probe (regressor.cpp: place_ffm_tables)
// Random 1 KiB whole-line read-modify-write on two buffers at once -- the FFM update's access pattern on (w, acc).  Values are
// written back unchanged.  tools/placement.hip measured that the same pattern runs 2.04 ms or 2.45 ms depending only on WHICH two
// allocations are paired (profiles/r02_placement.txt): device memory falls into groups, and two tables of one group contend.
namespace {
__device__ __forceinline__ uint32_t probe_mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void pair_probe_kernel(float *a, float *b, uint32_t lines, uint32_t nrows, uint32_t seed) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t line = probe_mix(r * 2654435761u + seed) % (lines - 8);
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(a + (size_t)line * 32, 0, 1024, 0x00020000);
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + (size_t)line * 32, 0, 1024, 0x00020000);
        const u4 x = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, 0, 16);
        u4 y = x;
        if (b) y = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(x, ra, lane * 16, 0, 16);
        if (b) __builtin_amdgcn_raw_buffer_store_b128(y, rb, lane * 16, 0, 16);
    }
}
}  // namespace

// milliseconds of the pair pattern over `nrows` random windows of two buffers of `bytes` each (best of `reps`); b == NULL: a alone
hipError_t pair_probe_ms(float *a, float *b, size_t bytes, uint32_t nrows, int reps, float *ms_out) {
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&e1);
    if (e != hipSuccess) {
        (void)hipEventDestroy(e0);
        return e;
    }
    const uint32_t lines = (uint32_t)std::min<size_t>(bytes / 128, 0xffffffffu);
    float best = 1e30f;
    for (int i = 0; i <= reps; i++) {  // (the first one warms up)
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(pair_probe_kernel, dim3(256 * 12), dim3(256), 0, 0, a, b, lines, nrows, 17u + (uint32_t)i);
        (void)hipEventRecord(e1, 0);
        e = hipEventSynchronize(e1);
        if (e != hipSuccess) break;
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (i > 0) best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = best;
    return e;
}

hipError_t sparse_apply(const SparseApplyArgs &a, int optimizer, hipStream_t stream) {
    switch (optimizer) {
    case FWGPU_OPT_SGD: return sparse_apply_t<FWGPU_OPT_SGD>(a, stream);
    case FWGPU_OPT_ADAGRAD_FLEX: return sparse_apply_t<FWGPU_OPT_ADAGRAD_FLEX>(a, stream);
    default: return sparse_apply_t<FWGPU_OPT_ADAGRAD_LUT>(a, stream);
    }
}

}  // namespace fwgpu
