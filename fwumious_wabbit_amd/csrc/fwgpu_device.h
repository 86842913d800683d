// Device helpers shared by the HIP translation units (kernels.hip, sparse.hip).
#pragma once
#include "fwgpu_internal.h"

namespace fwgpu {

// ---- optimizer steps (optimizer.rs).  acc chain is kept free of FMA contraction so that, given the same
// gradient, acc (and hence the integer LUT key) is bit-identical to the reference.
template <int OPT>
__device__ __forceinline__ float opt_step(float grad, float &acc, float rate, float minus_power_t, const float *lut) {
    if (OPT == FWGPU_OPT_SGD) {
        return grad * rate;  // optimizer.rs:36-38
    } else if (OPT == FWGPU_OPT_ADAGRAD_FLEX) {  // optimizer.rs:76-88
        float na = __fadd_rn(acc, __fmul_rn(grad, grad));
        acc = na;
        float u = __fmul_rn(__fmul_rn(grad, rate), powf(na, minus_power_t));
        return (isnan(u) || isinf(u)) ? 0.0f : u;
    } else {  // optimizer.rs:147-156
        float na = __fadd_rn(acc, __fmul_rn(grad, grad));
        acc = na;
        // (a diverged model can put a NaN with the sign bit set here, whose key would index past the 2048-entry table: the
        // reference's bounds check panics at that point, optimizer.rs:152; this stays inside the table and goes on with NaNs)
        uint32_t key = (__float_as_uint(na) >> (31 - kLutBits)) & (uint32_t)(kLutSize - 1);
        return __fmul_rn(grad, lut[key]);
    }
}

}  // namespace fwgpu
