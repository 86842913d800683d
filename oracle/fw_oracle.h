/*
 * fw_oracle.h -- CPU restatement of the fwumious_wabbit LR+FFM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C, single-file,
 * f32-everywhere restatement of the reference algorithm, following the reference's
 * loop and summation order.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (fwumious_wabbit_amd/)
 * never links or calls anything in oracle/.
 *
 * Pinning: the reference is Rust and cannot be built here (no cargo/rustc), so the
 * oracle is pinned against the reference's own known-answer tests, transcribed as
 * data in tests/golden/ (see tests/test_oracle_kat.py):
 *   optimizer.rs:170-226, regressor.rs:556-884, block_ffm.rs:1238-2037,
 *   persistence.rs:250-643, feature_buffer.rs:374-797, block_misc.rs:907-969,
 *   parser.rs:474-1183 (murmur3 hashes).
 * Parity UNPINNED parts (no reference test observes them): merand48 weight init
 * (third-party crate merand48 0.1.0, restated from VW's published LCG).
 *
 * All citations are file:line in /root/reference/src/.
 */
#ifndef FW_ORACLE_H
#define FW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* feature_buffer.rs:10-15 */
typedef struct {
    uint32_t hash;
    float value;
    uint32_t combo_index;
} fwo_lr_entry;

/* feature_buffer.rs:17-22 */
typedef struct {
    uint32_t hash;
    float value;
    uint32_t contra_field_index; /* field * ffm_k */
} fwo_ffm_entry;

/* model_instance.rs:24-28 */
enum { FWO_OPT_SGD = 100, FWO_OPT_ADAGRAD_FLEX = 200, FWO_OPT_ADAGRAD_LUT = 300 };

/* Graph wiring.  REGRESSOR = regressor.rs:173-330 for the LR(+FFM) graph:
 * [LR, FFM, Triangle, Sigmoid]; sigmoid input = LR slots then triangle rows
 * (graph.rs:251-285).  FFM_ONLY = the block tests' wiring
 * (block_ffm.rs:1253-1254: new_ffm_block -> new_logloss_block): the sigmoid sums
 * the raw F*F FFM outputs, no LR, no Triangle. */
enum { FWO_WIRING_REGRESSOR = 0, FWO_WIRING_FFM_ONLY = 1 };

typedef struct {
    int32_t optimizer;
    float learning_rate, power_t, init_acc_gradient; /* block_lr.rs:53-67 */
    uint32_t bit_precision;
    uint32_t num_combos;   /* feature_combo_descs.len() (+1 if constant), block_lr.rs:53-56 */
    uint32_t ffm_k, ffm_bit_precision, ffm_num_fields; /* block_ffm.rs:70-94 */
    float ffm_learning_rate, ffm_power_t, ffm_init_acc_gradient;
    float ffm_init_center, ffm_init_width, ffm_init_zero_band; /* block_ffm.rs:793-822 */
    int32_t wiring;
} fwo_config;

typedef struct fwo_model fwo_model;

/* ---- optimizer primitives (optimizer.rs) ---- */
#define FWO_LUT_BITS 11
#define FWO_LUT_SIZE (1 << FWO_LUT_BITS)
void fwo_lut_init(float *lut, float learning_rate, float power_t, float init_acc); /* optimizer.rs:121-144 */
float fwo_step_sgd(float lr, float g);                                             /* optimizer.rs:36-38 */
float fwo_step_flex(float lr, float minus_power_t, float g, float *acc);           /* optimizer.rs:76-88 */
float fwo_step_lut(const float *lut, float g, float *acc);                         /* optimizer.rs:147-156 */

/* ---- hashing ---- */
uint32_t fwo_murmur3_32(const uint8_t *data, size_t len, uint32_t seed); /* fasthash 0.4 murmur3::hash32_with_seed */
float fwo_merand48(uint64_t seed);                                       /* merand48 0.1.0 / VW rand48.cc */

/* ---- model ---- */
fwo_model *fwo_create(const fwo_config *cfg);
void fwo_free(fwo_model *m);
void fwo_init_weights(fwo_model *m);             /* regressor.rs:352: allocate_and_init_weights */
void fwo_ffm_fill(fwo_model *m, float w);        /* tests' ffm_init (block_ffm.rs:1228-1235) */
/* raw views (LR: interleaved {w,acc} pairs, 2 floats/entry; FFM: separate arrays) */
float *fwo_lr_table(fwo_model *m, uint64_t *n_entries);
float *fwo_ffm_weights(fwo_model *m, uint64_t *len);
float *fwo_ffm_acc(fwo_model *m, uint64_t *len);
const float *fwo_lut_lr(fwo_model *m);
const float *fwo_lut_ffm(fwo_model *m);

/* regressor.rs:356-379 (learn; routes to predict when !update || importance==0) */
float fwo_learn(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm, uint32_t n_ffm,
                float label, float importance, int update);
/* regressor.rs:381-395 (predict; inference numerics block_ffm.rs:316-440) */
float fwo_predict(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm, uint32_t n_ffm);
/* block_helpers.rs:162-174 slearn2: forward_backward chain without the predict fast path */
float fwo_forward_backward(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm,
                           uint32_t n_ffm, float label, float importance, int update);

/* ---- deep head: BlockCopy (block_misc.rs:435-519), BlockNeuronLayer (block_neural.rs:196-340),
 * BlockRELU (block_relu.rs:79-112), final single neuron InitType::One (regressor.rs:312-319).
 * topology 1 = "one" (default, model_instance.rs:42): x = [LR slots, triangle]; h = relu(W x + b) ... ;
 *                logit = w_f . [h_last, x] + b_f ;  topology 2 = "two": logit = w_f . h_last + b_f (no copy/join).
 * Hidden-layer init Hu / Xavier draws from rand_xoshiro + rand_distr in the reference (block_neural.rs:385-406):
 * PARITY UNPINNED (third-party RNG/ziggurat, no reference test observes a value); the oracle uses its own
 * deterministic N(0, sqrt(2/in)) / U(+-sqrt(6/(in*out))) stand-in and tests load identical weights on both sides. */
#define FWO_NN_MAX_LAYERS 8
enum { FWO_NN_INIT_XAVIER = 0, FWO_NN_INIT_HU = 1, FWO_NN_INIT_ONE = 2, FWO_NN_INIT_ZERO = 3 };
typedef struct {
    uint32_t n_layers;                  /* hidden layers (the final 1-neuron layer is implied) */
    uint32_t width[FWO_NN_MAX_LAYERS];
    uint32_t relu[FWO_NN_MAX_LAYERS];   /* activation: 1 relu, 0 none */
    uint32_t init[FWO_NN_MAX_LAYERS];
    uint32_t topology;                  /* 1 = "one", 2 = "two" */
    float nn_learning_rate, nn_power_t, nn_init_acc_gradient;
} fwo_nn_config;
int fwo_set_nn(fwo_model *m, const fwo_nn_config *nn);  /* call before fwo_init_weights */
/* layer 0..n_layers-1 = hidden, n_layers = final neuron; weights are [(in+1)*out]: W[j*in+i], biases at in*out+j */
float *fwo_nn_weights(fwo_model *m, uint32_t layer, uint64_t *len);
float *fwo_nn_acc(fwo_model *m, uint32_t layer, uint64_t *len);
/* one BlockNeuronLayer forward_backward with upstream gradient `out_grad` (block_neural.rs:252-340), for its KATs:
 * writes outputs to y, replaces x by the input gradients when update != 0 */
void fwo_neuron_layer_fb(int optimizer, float lr, float power_t, float init_acc, float *w, float *acc, uint32_t n_in,
                         uint32_t n_out, float *x, float *y, const float *out_grad, int update);

/* ---- block_misc.rs:742-884 Triangle (exposed for its KAT) ---- */
void fwo_triangle_forward(const float *in, uint32_t width, float *out);
void fwo_triangle_backward(const float *gout, uint32_t width, float *gin);

/* ---- translation: record -> feature buffer (feature_buffer.rs:138-338) ---- */
typedef struct {
    uint32_t n_combos;           /* feature_combo_descs.len() */
    const uint32_t *combo_off;   /* n_combos+1 offsets into combo_ns */
    const uint32_t *combo_ns;    /* namespace_index of each member */
    const uint8_t *combo_ns_f32; /* 1 if NamespaceFormat::F32 */
    const float *combo_weight;   /* n_combos */
    int32_t add_constant_feature;
    uint32_t bit_precision;
    uint32_t ffm_k, ffm_bit_precision;
    uint32_t n_fields;
    const uint32_t *field_off;   /* n_fields+1 offsets into field_ns */
    const uint32_t *field_ns;
    const uint8_t *field_ns_f32;
} fwo_translator;

uint32_t fwo_lr_hash_mask(uint32_t bit_precision);             /* feature_buffer.rs:140 */
uint32_t fwo_ffm_hash_mask(uint32_t ffm_bits, uint32_t ffm_k); /* feature_buffer.rs:141-148 */

/* Translates one record.  Returns 0 on success, -1 if an output buffer is too small.
 * label / importance decoded per feature_buffer.rs:187-189. */
int fwo_translate(const fwo_translator *t, const uint32_t *record, fwo_lr_entry *lr_out, uint32_t lr_cap,
                  uint32_t *n_lr, fwo_ffm_entry *ffm_out, uint32_t ffm_cap, uint32_t *n_ffm, float *label,
                  float *importance);

/* ---- stream runner: main.rs:213-270 (single thread) / hogwild.rs:24-103 (threads>1) ----
 * records back to back, rec_off[i] = u32 offset of record i (n+1 entries).
 * Examples numbered from 1; examples with number >= holdout_after (if holdout_after>0) are
 * predicted with update=false (main.rs:238-241).  preds may be NULL.  With nthreads>1 the
 * training part runs hogwild (racy, unordered, predictions 0.0 like main.rs:242-243), then
 * the holdout tail is predicted single-threaded.  Returns seconds spent in the training part. */
double fwo_run_stream(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off,
                      uint64_t n, uint64_t holdout_after, int nthreads, float *preds);

/* read-only pass (update = false) over n examples on nthreads threads; same predictions as the hold-out tail of fwo_run_stream */
void fwo_predict_stream(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n,
                        int nthreads, float *preds);

/* ---- synchronous micro-batch: the checker of the library's fwgpu_learn_batch_sync / sharded multi-GPU step ----
 * NOT a mode of the reference (which updates after every example): it is what N hogwild threads do when they all read the
 * weights at the same moment.  All n examples are scored with the weights as they are (forward passes of
 * block_lr.rs:28-47 / block_ffm.rs:163-261 and, with a deep head, block_neural.rs:196-222); then, per example in order, the
 * FFM and LR updates (block_ffm.rs:265-288, block_lr.rs:135-150) with the gradients of that frozen forward pass.  Deep
 * head: the dense gradients are SUMMED over the batch and every dense weight takes ONE optimizer step with the sum
 * (mini-batch AdaGrad), input gradients come from the frozen weights.  preds: the n predictions. */
void fwo_learn_minibatch(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n,
                         float *preds);
/* Row-sparse gradient buckets: update rule of the library's multi-GPU "sparse" mode (one optimizer step per table row and batch,
 * gradient summed over the row's occurrences; parts = the ranks' micro-batches, examples [part_end[p-1], part_end[p])).
 * Models without a deep head. */
/* An emulation of the device's concurrent mode for analysis (windows of examples scored against the window's first tables; flags bit 0: FFM weights written back as
 * start value - step, last writer wins; bit 1: accumulators likewise): NOT a reference code path, see fw_oracle.c */
void fwo_learn_window_emulation(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n, uint32_t window,
                                uint32_t flags, float *preds);
void fwo_learn_sparse(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n,
                      const uint64_t *part_end, uint32_t n_parts, float *preds);

#ifdef __cplusplus
}
#endif
#endif
