// Mini-batched deep head on the matrix cores (BASELINE config E) -- see head_step below.
#include "fwgpu_internal.h"

namespace fwgpu {

int head_step(fwgpu_regressor *r, fwgpu_split *sp, uint32_t first, uint32_t n, float *d_pred, bool update, hipStream_t stream) {
    (void)r; (void)sp; (void)first; (void)n; (void)d_pred; (void)update; (void)stream;
    return fail(FWGPU_ERR_INVALID, "mini-batched deep head: not built yet");
}

}  // namespace fwgpu
