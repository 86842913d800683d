/*
 * fw_ffi.h -- the reference's own serving FFI, exported by libfwgpu.so under the reference's names and signatures
 * (/root/reference/src/lib.rs:50-53, 150-236).  A program that links the reference's libfw (Java/JNI, C, Python ctypes)
 * links this library instead and keeps its calls: the predictions come from the MI355X example kernel.
 *
 *   lib.rs:150-185  new_fw_predictor_prototype(command)   command = the fw command line; `-i/--initial_regressor FILE` is
 *                                                         loaded as an immutable regressor (persistence.rs:127-174);
 *                                                         `--device N` (ours) picks the GPU.  NULL + fwgpu_last_error() on
 *                                                         failure (the reference panics).
 *   lib.rs:187-205  clone_lite(prototype)                 cheap per-thread copy sharing the weights
 *   lib.rs:207-212  fw_predict(ptr, vw_text)              -> prediction, or -1.0 for EOF / a line that does not parse
 *   lib.rs:224-232  fw_setup_cache(ptr, context_text)     -> 0.0 (or -1.0); remembers the request's context part
 *   lib.rs:214-222  fw_predict_with_cache(ptr, text)      -> prediction for context + text
 *   lib.rs:234-236  free_predictor(ptr)
 */
#ifndef FW_FFI_H
#define FW_FFI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct FfiPredictor FfiPredictor;

FfiPredictor *new_fw_predictor_prototype(const char *command);
FfiPredictor *clone_lite(FfiPredictor *prototype);
float fw_predict(FfiPredictor *ptr, const char *input_buffer);
float fw_predict_with_cache(FfiPredictor *ptr, const char *input_buffer);
float fw_setup_cache(FfiPredictor *ptr, const char *input_buffer);
void free_predictor(FfiPredictor *ptr);

/* Not in the reference: all candidates of one request in ONE device launch.  inputs[i] is what fw_predict
 * (with_cache == 0) or fw_predict_with_cache (with_cache != 0) would be given, out[i] what it would return. */
int fwgpu_predictor_predict_batch(FfiPredictor *ptr, const char *const *inputs, uint32_t n, int with_cache, float *out);

#ifdef __cplusplus
}
#endif
#endif
