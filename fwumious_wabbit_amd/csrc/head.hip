// Mini-batched deep head on the matrix cores (BASELINE config E: FFM k=16 + 2 x 256 ReLU, topology "one").
//
// The reference's head (block_neural.rs:196-340, block_relu.rs:38-112, regressor.rs:191-323) is per example: sgemv forward,
// then every dense weight takes an AdaGrad step with that one example's gradient -- 20 B x 194 k weights = 3.9 MB of L2 traffic
// per example, which is what holds the exact mode (kernels.hip nn_forward / nn_backward) at 0.8 M examples/s.  This file is the
// explicit mini-batch mode that sits NEXT to it (never instead of it): inside a synchronous micro-batch (regressor.cpp
// "split pipeline") the dense weights are frozen, the batch goes through the layers as GEMMs on
// v_mfma_f32_32x32x2_f32 (exact f32, 157 TFLOP/s peak), the weight gradients are SUMMED over the batch and every dense weight
// takes ONE optimizer step with the sum.  oracle/fw_oracle.c fwo_learn_minibatch is the CPU statement of the same mode.
//
//   x [n, X]  --W1-->  z1 -relu-> h1 [n, w1]  --W2-->  z2 -relu-> h2 [n, w2] ...   logit = w_f . [h_last | x] + b_f   (topology one)
//   g = -(y - p) * importance ;  d h_last = g * w_f[:wl] ;  dx = g * w_f[wl:]  (+ the path through the layers)
//   dW_l = dz_l^T . in_l  (K = n),  d in_l = dz_l . W_l ;  AdaGrad: acc += G^2 ; w -= G * lut[bits(acc) >> 20]   per weight, once
#include "fwgpu_internal.h"

namespace fwgpu {

typedef float f16v __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------- GEMM
// C[M, N] (ldc) = op(A) . op(B)  [+ C]  with an epilogue.  f32 in, f32 accumulate, v_mfma_f32_32x32x2_f32.
// Workgroup = 256 threads = 4 waves, 64 x 64 tile of C, wave (wr, wc) owns the 32 x 32 sub-tile; K in steps of 16 through LDS.
//   A(m, k) = TA ? A[k * lda + m] : A[m * lda + k]        B(k, n) = TB ? B[n * ldb + k] : B[k * ldb + n]
// EPI 0: C = acc          EPI 1: C = relu?(acc + bias[n]), mask[m, n] = acc + bias >= 0 (block_relu.rs:38-54)
// EPI 2: C = acc * mul[m, n] (ReLU backward, block_relu.rs:105-110)          EPI 3: C += acc
constexpr int kTM = 64, kTN = 64, kTK = 16, kPad = 4;

template <bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(256) head_gemm(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
                                                 int N, int K, int lda, int ldb, int ldc, const float *__restrict__ bias,
                                                 float *__restrict__ aux, int relu) {
    __shared__ float As[kTK][kTM + kPad];  // k-major: lane i reads As[k][i] (consecutive lanes, consecutive banks)
    __shared__ float Bs[kTK][kTN + kPad];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTN;
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += kTK) {
        // stage: 64 x 16 elements of each operand, 4 per thread, the thread index running along the contiguous dimension
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            int m, k;
            if (TA) {  // m contiguous in memory
                m = e & 63;
                k = e >> 6;
            } else {  // k contiguous
                k = e & 15;
                m = e >> 4;
            }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.0f;
            if (gm < M && gk < K) v = TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk];
            As[k][m] = v;
            int n, kb;
            if (TB) {  // k contiguous
                kb = e & 15;
                n = e >> 4;
            } else {  // n contiguous
                n = e & 63;
                kb = e >> 6;
            }
            const int gn = n0 + n, gkb = k0 + kb;
            float u = 0.0f;
            if (gn < N && gkb < K) u = TB ? B[(size_t)gn * ldb + gkb] : B[(size_t)gkb * ldb + gn];
            Bs[kb][n] = u;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTK; kk += 2) {
            // 32x32x2: lane l feeds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
            const float a = As[kk + (lane >> 5)][wr * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wc * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = n0 + wc * 32 + (lane & 31);
    if (col < N) {
        const float bj = (EPI == 1 && bias) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) {
                const size_t ix = (size_t)row * ldc + col;
                float v = acc[r];
                if (EPI == 1) {
                    v += bj;
                    const bool neg = relu && v < 0.0f;
                    aux[ix] = neg ? 0.0f : 1.0f;
                    v = neg ? 0.0f : v;
                    C[ix] = v;
                } else if (EPI == 2) {
                    C[ix] = v * aux[ix];
                } else if (EPI == 3) {
                    C[ix] += v;
                } else {
                    C[ix] = v;
                }
            }
        }
    }
}

// The same product for the shapes the head really has (a batch of ~1024 rows against 256 ... 496 columns): the 64 x 64 tiles above make 32-64 workgroups
// of them, i.e. most of the 256 CUs idle (2-3 TFLOP/s).  Here a workgroup owns ONE 32 x 32 tile of C and its eight waves split K: every wave feeds the
// matrix core straight from global memory (the operands, a few MB, live in L2: no LDS staging, no barrier in the K loop), the loads of the next three groups of eight k
// in flight while the current eight are multiplied; the partial tiles meet in LDS and are summed in wave order (deterministic).
//   k of MFMA t of group u: 8 u + 4 (lane >> 5) + t -- so that a lane's four k are CONTIGUOUS: one 16-byte load where k is the fast dimension.
// Needs K % 8 == 0, lda / ldb % 4 == 0 and 16-byte aligned operands where k is their fast dimension (head_step checks; otherwise head_gemm above).
static bool g_force_tiled = false;  // (fwgpu_debug_head_gemm: tests run both kernels on one shape)
constexpr int kSplitK = 8;  // waves of a workgroup = shares of K
template <bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(64 * kSplitK) head_gemm_splitk(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M,
                                                        int N, int K, int lda, int ldb, int ldc, const float *__restrict__ bias,
                                                        float *__restrict__ aux, int relu) {
    __shared__ float part[kSplitK][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, ij = lane & 31;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int gm = m0 + ij, gn = n0 + ij;
    const bool mok = gm < M, nok = gn < N;
    const int U = K >> 3, u0 = (U * wave) / kSplitK, u1 = (U * (wave + 1)) / kSplitK;
    f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    typedef float f4v __attribute__((ext_vector_type(4)));
    auto load_a = [&](int u) -> f4v {
        f4v v = {0, 0, 0, 0};
        const int kb = 8 * u + 4 * half;
        if (mok) {
            if (TA) {  // A[k * lda + m]: four rows of the k-major operand, each coalesced across the 32 lanes of a half
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = A[(size_t)(kb + t) * lda + gm];
            } else {
                v = *reinterpret_cast<const f4v *>(A + (size_t)gm * lda + kb);
            }
        }
        return v;
    };
    auto load_b = [&](int u) -> f4v {
        f4v v = {0, 0, 0, 0};
        const int kb = 8 * u + 4 * half;
        if (nok) {
            if (TB) {
                v = *reinterpret_cast<const f4v *>(B + (size_t)gn * ldb + kb);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = B[(size_t)(kb + t) * ldb + gn];
            }
        }
        return v;
    };
    // three groups of eight k in flight ahead of the one being multiplied (rotating registers: no indexed arrays)
    const f4v z4 = {0, 0, 0, 0};
    f4v a0 = u0 < u1 ? load_a(u0) : z4, b0 = u0 < u1 ? load_b(u0) : z4;
    f4v a1 = u0 + 1 < u1 ? load_a(u0 + 1) : z4, b1 = u0 + 1 < u1 ? load_b(u0 + 1) : z4;
    f4v a2 = u0 + 2 < u1 ? load_a(u0 + 2) : z4, b2 = u0 + 2 < u1 ? load_b(u0 + 2) : z4;
    for (int u = u0; u < u1; ++u) {
        f4v a3 = z4, b3 = z4;
        if (u + 3 < u1) {
            a3 = load_a(u + 3);
            b3 = load_b(u + 3);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
        a0 = a1; b0 = b1;
        a1 = a2; b1 = b2;
        a2 = a3; b2 = b3;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    // wave w finishes registers (16 / kSplitK) w ... of the tile: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = n0 + ij;
    if (col < N) {
        const float bj = (EPI == 1 && bias) ? bias[col] : 0.0f;
        constexpr int RPW = 16 / kSplitK;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = RPW * wave + q;
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < M) {
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < kSplitK; ++w) v += part[w][r][lane];  // wave order: deterministic
                const size_t ix = (size_t)row * ldc + col;
                if (EPI == 1) {
                    v += bj;
                    const bool neg = relu && v < 0.0f;
                    aux[ix] = neg ? 0.0f : 1.0f;
                    C[ix] = neg ? 0.0f : v;
                } else if (EPI == 2) {
                    C[ix] = v * aux[ix];
                } else if (EPI == 3) {
                    C[ix] += v;
                } else {
                    C[ix] = v;
                }
            }
        }
    }
}

template <bool TA, bool TB, int EPI>
static hipError_t gemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, const float *bias,
                       float *aux, int relu, hipStream_t s) {
    if (M <= 0 || N <= 0) return hipSuccess;
    static const bool env_tiled = std::getenv("FWGPU_HEAD_GEMM_TILED") != nullptr;  // (A/B runs: the LDS-tiled 64 x 64 kernel for every shape)
    const bool old_only = env_tiled || g_force_tiled;
    const bool a_vec = TA || (lda % 4 == 0 && ((uintptr_t)A & 15u) == 0), b_vec = !TB || (ldb % 4 == 0 && ((uintptr_t)B & 15u) == 0);
    // (the split-K kernel is for the small products of a micro-batch -- every 32 x 32 tile of C a workgroup of its own; a predict-only slab of tens of
    // thousands of rows fills the device with 64 x 64 tiles and re-uses each staged operand 64 times)
    static const int splitk_max_m = [] { const char *e = std::getenv("FWGPU_HEAD_GEMM_SPLITK_MAX_M"); return e ? std::atoi(e) : 4096; }();
    if (!old_only && M <= splitk_max_m && K >= 8 * kSplitK && K % 8 == 0 && a_vec && b_vec) {
        dim3 grid((N + 31) / 32, (M + 31) / 32);
        hipLaunchKernelGGL((head_gemm_splitk<TA, TB, EPI>), grid, dim3(64 * kSplitK), 0, s, A, B, C, M, N, K, lda, ldb, ldc, bias, aux, relu);
        return hipGetLastError();
    }
    dim3 grid((N + kTN - 1) / kTN, (M + kTM - 1) / kTM);
    hipLaunchKernelGGL((head_gemm<TA, TB, EPI>), grid, dim3(256), 0, s, A, B, C, M, N, K, lda, ldb, ldc, bias, aux, relu);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------- final neuron
// One wave per example: logit = w_f . [h_last | x] + b_f (regressor.rs:307-319), sigmoid / log-loss gradient
// (block_loss_functions.rs:105-153); then the final neuron's input gradients from the FROZEN w_f: d h_last (times the last
// ReLU mask) and the direct part of dx.
__global__ void __launch_bounds__(256) head_final_kernel(const float *__restrict__ h_last, const float *__restrict__ mask_last, const float *__restrict__ x,
                                                         const float *__restrict__ wf, const float *__restrict__ yi, float *__restrict__ pred,
                                                         float *__restrict__ gvec, float *__restrict__ dz_last, float *__restrict__ dx, int n, int wl,
                                                         int X, int topo_one, int update) {
    const int lane = threadIdx.x & 63;
    const int ex = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ex >= n) return;
    const int fin = wl + (topo_one ? X : 0);
    float dot = 0.0f;
    for (int i = lane; i < wl; i += 64) dot += wf[i] * h_last[(size_t)ex * wl + i];
    if (topo_one)
        for (int i = lane; i < X; i += 64) dot += wf[wl + i] * x[(size_t)ex * X + i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    const float z = wf[fin] + dot;
    const float label = yi[2 * (size_t)ex], imp = yi[2 * (size_t)ex + 1];
    float p, g;
    if (isnan(z)) {
        p = 1.0f / (1.0f + expf(-0.0f));
        g = 0.0f;
    } else if (z < -50.0f) {
        p = 1.0f / (1.0f + expf(50.0f));
        g = 0.0f;
    } else if (z > 50.0f) {
        p = 1.0f / (1.0f + expf(-50.0f));
        g = 0.0f;
    } else {
        p = 1.0f / (1.0f + expf(-z));
        g = -(label - p) * imp;
    }
    if (!update || imp == 0.0f) g = 0.0f;  // regressor.rs:366: importance 0 is a prediction, nothing is learned
    if (lane == 0) {
        pred[ex] = p;
        gvec[ex] = g;
    }
    if (!update) return;  // (a predict-only batch has no gradient buffers)
    for (int i = lane; i < wl; i += 64) dz_last[(size_t)ex * wl + i] = g * wf[i] * mask_last[(size_t)ex * wl + i];
    for (int i = lane; i < X; i += 64) dx[(size_t)ex * X + i] = topo_one ? g * wf[wl + i] : 0.0f;
}

// out[c] = sum_e scale[e] * Mat[e, c]  (scale == NULL: plain column sums): the bias gradients and the final neuron's weight
// gradients.  Workgroup = 64 columns x 16 row groups (1024 threads: 256 ... 496 columns make only 4 ... 8 workgroups, so the rows have to be spread inside
// them); every group walks its sixteenth of the examples in order, the partial sums are added in a fixed order: deterministic, coalesced (64
// consecutive columns per wave load).
constexpr int kColsumGroups = 16;
__global__ void __launch_bounds__(64 * kColsumGroups) head_colsum_kernel(const float *__restrict__ mat, const float *__restrict__ scale, float *__restrict__ out,
                                                                         int n, int cols, int ld) {
    __shared__ float part[kColsumGroups][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int per = (n + kColsumGroups - 1) / kColsumGroups, e0 = grp * per, e1 = min(n, e0 + per);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (c < cols) {
        int e = e0;
        for (; e + 4 <= e1; e += 4) {
            a0 += (scale ? scale[e] : 1.0f) * mat[(size_t)e * ld + c];
            a1 += (scale ? scale[e + 1] : 1.0f) * mat[(size_t)(e + 1) * ld + c];
            a2 += (scale ? scale[e + 2] : 1.0f) * mat[(size_t)(e + 2) * ld + c];
            a3 += (scale ? scale[e + 3] : 1.0f) * mat[(size_t)(e + 3) * ld + c];
        }
        for (; e < e1; ++e) a0 += (scale ? scale[e] : 1.0f) * mat[(size_t)e * ld + c];
    }
    part[grp][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && c < cols) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < kColsumGroups; ++g) t += part[g][lane];
        out[c] = t;
    }
}
__global__ void __launch_bounds__(256) head_sum_kernel(const float *__restrict__ v, float *__restrict__ out, int n) {
    __shared__ float part[256];
    float acc = 0.0f;
    for (int e = threadIdx.x; e < n; e += 256) acc += v[e];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}

// One optimizer step per dense weight with the batch's summed gradient (optimizer.rs:15-162)
__global__ void head_apply_kernel(float *__restrict__ w, float *__restrict__ acc, const float *__restrict__ dw, unsigned long long n, int optimizer,
                                  float rate, float minus_power_t, const float *__restrict__ lut) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = dw[i];
        if (g == 0.0f) continue;
        float upd;
        if (optimizer == FWGPU_OPT_SGD) {
            upd = g * rate;
        } else {
            const float na = __fadd_rn(acc[i], __fmul_rn(g, g));
            acc[i] = na;
            if (optimizer == FWGPU_OPT_ADAGRAD_FLEX) {
                upd = __fmul_rn(__fmul_rn(g, rate), powf(na, minus_power_t));
                if (isnan(upd) || isinf(upd)) upd = 0.0f;
            } else {
                upd = __fmul_rn(g, lut[(__float_as_uint(na) >> (31 - kLutBits)) & (uint32_t)(kLutSize - 1)]);  // (masked: see opt_step)
            }
        }
        w[i] -= upd;
    }
}

// ---------------------------------------------------------------------------------------------------------------- driver
struct HeadScratch {
    uint32_t n_cap = 0;
    float *h[kNnMaxLayers] = {}, *m[kNnMaxLayers] = {}, *dz[kNnMaxLayers] = {};
    float *gvec = nullptr, *dw = nullptr;
};
static HeadScratch &scratch_of(fwgpu_regressor *r) {
    if (!r->head_scratch) r->head_scratch = new HeadScratch();
    return *static_cast<HeadScratch *>(r->head_scratch);
}
void head_scratch_free(fwgpu_regressor *r) {
    HeadScratch *hs = static_cast<HeadScratch *>(r->head_scratch);
    if (!hs) return;
    for (uint32_t l = 0; l < kNnMaxLayers; l++)
        for (float *q : {hs->h[l], hs->m[l], hs->dz[l]})
            if (q) (void)hipFree(q);
    if (hs->gvec) (void)hipFree(hs->gvec);
    if (hs->dw) (void)hipFree(hs->dw);
    delete hs;
    r->head_scratch = nullptr;
}

int head_step(fwgpu_regressor *r, fwgpu_split *sp, uint32_t first, uint32_t n, float *d_pred, bool update, hipStream_t stream) {
    const DevNN &nn = r->nn;
    const uint32_t L = nn.n_layers, X = nn.X;
    if (!L) return fail(FWGPU_ERR_INVALID, "head_step: the model has no deep head");
    if (!n) return FWGPU_OK;
    HeadScratch &hs = scratch_of(r);
    if (hs.n_cap < n) {
        for (uint32_t l = 0; l < kNnMaxLayers; l++)
            for (float **q : {&hs.h[l], &hs.m[l], &hs.dz[l]}) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
        if (hs.gvec) (void)hipFree(hs.gvec);
        hs.gvec = nullptr;
        for (uint32_t l = 0; l < L; l++)
            for (float **q : {&hs.h[l], &hs.m[l], &hs.dz[l]}) FWGPU_HIP(hipMalloc((void **)q, (size_t)n * nn.out[l] * sizeof(float)));
        FWGPU_HIP(hipMalloc((void **)&hs.gvec, (size_t)n * sizeof(float)));
        hs.n_cap = n;
    }
    if (!hs.dw) FWGPU_HIP(hipMalloc((void **)&hs.dw, (size_t)r->nn_len * sizeof(float)));
    const float *x = sp->d_x + (size_t)first * X;
    const float *yi = sp->d_g + 2 * (size_t)first;
    float *dx = sp->d_dx + (size_t)first * X;
    // ---- forward through the hidden layers: z = in . W^T + b, ReLU (or identity) with its mask
    const float *in = x;
    uint32_t in_w = X;
    for (uint32_t l = 0; l < L; l++) {
        const float *W = nn.w + nn.off[l];
        FWGPU_HIP((gemm<false, true, 1>(in, W, hs.h[l], (int)n, (int)nn.out[l], (int)in_w, (int)in_w, (int)nn.in[l], (int)nn.out[l],
                                       W + (size_t)nn.in[l] * nn.out[l], hs.m[l], (int)nn.relu[l], stream)));
        in = hs.h[l];
        in_w = nn.out[l];
    }
    // ---- final neuron, sigmoid, gradient; input gradients of the final neuron
    const uint32_t wl = nn.out[L - 1];
    const float *wf = nn.w + nn.off[L];
    hipLaunchKernelGGL(head_final_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, hs.h[L - 1], hs.m[L - 1], x, wf, yi, d_pred, hs.gvec,
                       hs.dz[L - 1], dx, (int)n, (int)wl, (int)X, nn.topology == 1 ? 1 : 0, update ? 1 : 0);
    FWGPU_HIP(hipGetLastError());
    if (!update) return FWGPU_OK;
    float *dW = hs.dw;
    // final neuron's weight gradients: sum_e g_e * [h_last | x]_e, bias: sum_e g_e
    {
        float *dwf = dW + nn.off[L];
        hipLaunchKernelGGL(head_colsum_kernel, dim3((wl + 63) / 64), dim3(64 * kColsumGroups), 0, stream, hs.h[L - 1], hs.gvec, dwf, (int)n, (int)wl, (int)wl);
        if (nn.topology == 1)
            hipLaunchKernelGGL(head_colsum_kernel, dim3((X + 63) / 64), dim3(64 * kColsumGroups), 0, stream, x, hs.gvec, dwf + wl, (int)n, (int)X, (int)X);
        hipLaunchKernelGGL(head_sum_kernel, dim3(1), dim3(256), 0, stream, hs.gvec, dwf + nn.in[L], (int)n);
        FWGPU_HIP(hipGetLastError());
    }
    // ---- backward through the hidden layers (block_neural.rs:252-340 in matrix form, frozen weights)
    for (int l = (int)L - 1; l >= 0; --l) {
        const float *W = nn.w + nn.off[l];
        const uint32_t out = nn.out[l], inw = nn.in[l];
        const float *lin = l == 0 ? x : hs.h[l - 1];
        // dW_l[j, i] = sum_e dz[e, j] * in[e, i]
        FWGPU_HIP((gemm<true, false, 0>(hs.dz[l], lin, dW + nn.off[l], (int)out, (int)inw, (int)n, (int)out, (int)inw, (int)inw, nullptr, nullptr, 0, stream)));
        hipLaunchKernelGGL(head_colsum_kernel, dim3((out + 63) / 64), dim3(64 * kColsumGroups), 0, stream, hs.dz[l], (const float *)nullptr,
                           dW + nn.off[l] + (size_t)inw * out, (int)n, (int)out, (int)out);
        FWGPU_HIP(hipGetLastError());
        if (l > 0) {  // d in = dz . W, then through the previous layer's ReLU mask
            FWGPU_HIP((gemm<false, false, 2>(hs.dz[l], W, hs.dz[l - 1], (int)n, (int)inw, (int)out, (int)out, (int)inw, (int)inw, nullptr, hs.m[l - 1], 0, stream)));
        } else {  // BlockCopy sums the two branches into dx (block_misc.rs:456-475)
            FWGPU_HIP((gemm<false, false, 3>(hs.dz[0], W, dx, (int)n, (int)X, (int)out, (int)out, (int)inw, (int)X, nullptr, nullptr, 0, stream)));
        }
    }
    // ---- one optimizer step per dense weight
    hipLaunchKernelGGL(head_apply_kernel, dim3(512), dim3(256), 0, stream, nn.w, nn.acc, dW, (unsigned long long)r->nn_len, r->cfg.optimizer, nn.rate,
                       nn.minus_power_t, nn.lut);
    FWGPU_HIP(hipGetLastError());
    return FWGPU_OK;
}

}  // namespace fwgpu

extern "C" int fwgpu_debug_head_gemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, int ta, int tb, int epilogue,
                                     const float *bias, float *aux, int relu, int tiled, void *stream) {
    using namespace fwgpu;
    hipStream_t st = (hipStream_t)stream;
    g_force_tiled = tiled != 0;
    hipError_t e = hipErrorInvalidValue;
    const int combo = (ta ? 2 : 0) | (tb ? 1 : 0);
#define FW_HG(TA, TB, EPI) e = gemm<TA, TB, EPI>(A, B, C, M, N, K, lda, ldb, ldc, bias, aux, relu, st)
    if (combo == 1 && epilogue == 1) FW_HG(false, true, 1);
    else if (combo == 2 && epilogue == 0) FW_HG(true, false, 0);
    else if (combo == 0 && epilogue == 2) FW_HG(false, false, 2);
    else if (combo == 0 && epilogue == 3) FW_HG(false, false, 3);
#undef FW_HG
    g_force_tiled = false;
    if (e != hipSuccess) return fail(FWGPU_ERR_INVALID, "head_gemm: not one of the head's products ((ta, tb, epilogue) = (0,1,1), (1,0,0), (0,0,2), (0,0,3)) or a launch error");
    return FWGPU_OK;
}
