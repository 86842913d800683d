/*
 * fwgpu.h -- C ABI of the MI355X-native LR+FFM online learner.
 *
 * This library is a drop-in for ONE path of outbrain-inc/fwumious_wabbit: what sits behind
 * `Regressor` (src/regressor.rs:142-147) and `HogwildTrainer` (src/hogwild.rs:13-61), i.e.
 * block_lr / block_ffm / block_misc::Triangle / block_loss_functions / optimizer / hogwild.
 * Everything is plain pointers and sizes; no C++/torch types cross this boundary.
 *
 * Conventions (cf. the reference's existing predict-only C ABI, src/lib.rs:151-243):
 *  - every call returns an int status (FWGPU_OK == 0); fwgpu_last_error() gives the message of the
 *    last failure on the calling thread.  Nothing unwinds across the boundary.
 *  - the caller owns every input buffer; buffers are borrowed for the duration of the call only
 *    (fwgpu_digest_records copies the records, like main.rs:243 `Vec::from(buffer)`).
 *  - the library owns weights, optimizer state and all device scratch.
 *  - one handle == one device context.  learn/update calls on one handle are single-producer
 *    (regressor.rs:362-365: learn() is not thread-safe).
 *  - `stream` arguments are `hipStream_t` passed as void* (NULL = the null stream), so a caller that
 *    already runs on a stream (e.g. torch.cuda.current_stream().cuda_stream) can time and order the
 *    work with its own events.
 *  - there is NO CPU fallback: if no HIP device is usable fwgpu_create fails with FWGPU_ERR_DEVICE.
 */
#ifndef FWGPU_H
#define FWGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FWGPU_ABI_VERSION 1

enum {
    FWGPU_OK = 0,
    FWGPU_ERR_INVALID = 1, /* bad argument / config (the reference exits 1: main.rs:44-47) */
    FWGPU_ERR_DEVICE = 2,  /* HIP error or no device */
    FWGPU_ERR_OOM = 3,
    FWGPU_ERR_RANGE = 4,   /* buffer too small / index out of range */
    FWGPU_ERR_FORMAT = 5,  /* malformed record, weight blob, cache or model file */
    FWGPU_ERR_PARSE = 6,   /* the text parser rejected a line; fwgpu_last_error() holds the reference's message */
    FWGPU_ERR_IO = 7,      /* file could not be opened / read / written */
    FWGPU_ERR_PEER = 8,    /* multi-GPU step: ANOTHER rank of the job failed (or is gone); this rank exchanged and applied nothing in the step */
    /* not errors: commands the text parser hands back instead of a record (parser.rs:31-57) */
    FWGPU_PARSE_FLUSH = 100,
    FWGPU_PARSE_HOGWILD_LOAD = 101
};

/* model_instance.rs:24-28 `enum Optimizer` */
enum { FWGPU_OPT_SGD = 100, FWGPU_OPT_ADAGRAD_FLEX = 200, FWGPU_OPT_ADAGRAD_LUT = 300 };

/* Graph wiring.  REGRESSOR: regressor.rs:173-330 LR -> [FFM -> Triangle -> Join] -> Sigmoid.
 * FFM_ONLY: the wiring of the reference's FFM block tests (block_ffm.rs:1253-1254: FFM -> Sigmoid). */
enum { FWGPU_WIRING_REGRESSOR = 0, FWGPU_WIRING_FFM_ONLY = 1 };

/* How a batch is executed on the device.
 * SEQUENTIAL: one workgroup walks the batch in order; example i sees every update of examples <i.
 *             This is the reference's single-thread semantics (main.rs:213-270) and the parity mode.
 * HOGWILD:    all workgroups run examples concurrently against the shared tables with unsynchronised
 *             read-modify-write, the device analogue of hogwild.rs:24-103 (non-deterministic). */
enum { FWGPU_MODE_SEQUENTIAL = 0, FWGPU_MODE_HOGWILD = 1 };

/* feature_buffer.rs:10-15 `HashAndValue` */
typedef struct fwgpu_lr_entry {
    uint32_t hash;
    float value;
    uint32_t combo_index;
} fwgpu_lr_entry;

/* feature_buffer.rs:17-22 `HashAndValueAndSeq` */
typedef struct fwgpu_ffm_entry {
    uint32_t hash;
    float value;
    uint32_t contra_field_index; /* field * ffm_k */
} fwgpu_ffm_entry;

/* The ModelInstance fields the blocks on this path read (model_instance.rs:47-97;
 * block_lr.rs:53-67, block_ffm.rs:70-94, 793-822). */
typedef struct fwgpu_config {
    int32_t optimizer;        /* FWGPU_OPT_* */
    float learning_rate;      /* -l */
    float power_t;            /* --power_t */
    float init_acc_gradient;  /* --init_acc_gradient */
    uint32_t bit_precision;   /* -b : LR table has 2^b entries (block_lr.rs:67) */
    uint32_t num_combos;      /* feature_combo_descs.len() + (add_constant_feature ? 1 : 0) (block_lr.rs:53-56) */
    uint32_t ffm_k;           /* --ffm_k, 0 = no FFM block */
    uint32_t ffm_bit_precision;
    uint32_t ffm_num_fields;  /* ffm_fields.len() */
    float ffm_learning_rate, ffm_power_t, ffm_init_acc_gradient;
    float ffm_init_center, ffm_init_width, ffm_init_zero_band;
    int32_t wiring;           /* FWGPU_WIRING_* */
    int32_t device;           /* HIP device ordinal */
} fwgpu_config;

typedef struct fwgpu_regressor fwgpu_regressor; /* opaque, like lib.rs:50-53 FfiPredictor */
typedef struct fwgpu_batch fwgpu_batch;         /* opaque device-resident micro-batch */
typedef struct fwgpu_trainer fwgpu_trainer;     /* opaque record-stream trainer (HogwildTrainer role) */

const char *fwgpu_last_error(void);
int fwgpu_abi_version(void);

/* ---------------------------------------------------------------- Regressor
 * fwgpu_create        <= Regressor::new_without_weights(&ModelInstance)   regressor.rs:173-330
 *                        (tables are allocated here, zero-filled)
 * fwgpu_init_weights  <= Regressor::allocate_and_init_weights             regressor.rs:352-354
 * fwgpu_learn         <= Regressor::learn(&FeatureBuffer,&mut PortBuffer,update) -> f32   regressor.rs:356-379
 * fwgpu_predict       <= Regressor::predict(&FeatureBuffer,&mut PortBuffer) -> f32        regressor.rs:381-395
 * fwgpu_free          <= drop(Regressor)
 * The FeatureBuffer (feature_buffer.rs:24-31) is passed as its four members. ffm entries must be
 * ordered by field, as FeatureBufferTranslator produces them (block_ffm.rs:165-183 relies on it). */
int fwgpu_create(const fwgpu_config *cfg, fwgpu_regressor **out);
int fwgpu_free(fwgpu_regressor *r);
int fwgpu_init_weights(fwgpu_regressor *r);
int fwgpu_learn(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm,
                uint32_t n_ffm, float label, float importance, int update, float *prediction);
int fwgpu_predict(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm,
                  uint32_t n_ffm, float *prediction);

/* ---------------------------------------------------------------- serving context cache
 * fwgpu_setup_cache        <= Regressor::setup_cache(&FeatureBuffer, &mut Vec<BlockCache>, should_create)  regressor.rs:409-423
 *                             (*cache == NULL creates it; otherwise the existing cache is refilled)
 * fwgpu_predict_with_cache <= Regressor::predict_with_cache(&FeatureBuffer, &mut PortBuffer, &[BlockCache]) -> f32
 *                             regressor.rs:397-407.  fb is the FeatureBuffer of context + candidate, as the reference's
 *                             caller passes it (lib.rs:88-108).
 * The cache is BlockFFM's (block_ffm.rs:442-782): the context features' field sums and self-pair corrections, computed once
 * on the device, plus `features_present` (regressor.rs:25-38: hash + contra_field_index).  A candidate then gathers only the
 * FFM rows of the features that are not present in the cache.  BlockLR's cache never holds anything in the reference
 * (block_lr.rs:236-239 skips every masked hash), so LR entries are all read, as there.  Models with a deep head: refused.
 * fwgpu_block_cache_filter: the FFM entries forward_with_cache still gathers (what an entry batch used with
 * fwgpu_batch_set_cache: must hold); predict-only launches of that batch then start every example from the cache. */
typedef struct fwgpu_block_cache fwgpu_block_cache;
int fwgpu_setup_cache(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm,
                      uint32_t n_ffm, fwgpu_block_cache **cache);
int fwgpu_predict_with_cache(fwgpu_regressor *r, const fwgpu_block_cache *cache, const fwgpu_lr_entry *lr, uint32_t n_lr,
                             const fwgpu_ffm_entry *ffm, uint32_t n_ffm, float *prediction);
int fwgpu_block_cache_filter(const fwgpu_block_cache *cache, const fwgpu_ffm_entry *ffm, uint32_t n_ffm, fwgpu_ffm_entry *out,
                       uint32_t *n_out);
int fwgpu_block_cache_free(fwgpu_block_cache *cache);
/* Record batches with the cache: the context's own record marks its namespace slots as covered; the record of a request
 * (context + candidate, parser.rs:195-211) then goes to the device whole and the example kernel's translation leaves the
 * covered slots' FFM features out.  That equals fwgpu_block_cache_filter on the request's translation whenever
 * fwgpu_block_cache_record_ok returns 1 (the request has not named a covered namespace again, and none of its own features
 * has the hash and field of a cached one); a request for which it returns 0 takes the entry route. */
struct fwgpu_translator_config;
int fwgpu_block_cache_cover_record(fwgpu_block_cache *cache, const struct fwgpu_translator_config *t, const uint32_t *record, uint32_t len);
int fwgpu_block_cache_record_ok(const fwgpu_block_cache *cache, const struct fwgpu_translator_config *t, const uint32_t *record, uint32_t len);

/* ---------------------------------------------------------------- deep head (BASELINE config E)
 * fwgpu_set_nn <= the `--nn_layers / --nn N:width:W / --nn N:activation:relu / --nn N:init:hu / --nn_topology` part of
 * Regressor::new_without_weights (regressor.rs:191-320): BlockCopy -> [BlockNeuronLayer -> BlockRELU]* -> Join ->
 * single neuron (InitType::One) in front of the sigmoid.  Call after fwgpu_create and before fwgpu_init_weights.
 * Per-example semantics are the reference's (block_neural.rs:252-340: every dense weight with a non-zero upstream
 * gradient takes an AdaGrad step, neuron by neuron).
 * INIT STREAM -- PARITY UNPINNED: hidden-layer init Hu / Xavier draws from the library's own deterministic generator.  The
 * reference draws from Xoshiro256PlusPlus::seed_from_u64(..) + rand_distr::Normal (block_neural.rs:385-406; third-party crates,
 * no reference test observes a value), so a model INITIALISED here has the reference's distribution but not its numbers: it is
 * not the model `fw` would initialise from the same command line.  Cross-checks load identical weights on both sides
 * (fwgpu_table_write(FWGPU_TABLE_NN_W) or a model file); InitType::One / Zero are exact.
 * Mini-batch mode: inside fwgpu_learn_batch_sync a deep head trains mini-batched on the matrix cores (dense weights frozen per
 * batch, gradients summed, ONE optimizer step per weight and batch) -- a different algorithm from the per-example one above,
 * never chosen implicitly. */
#define FWGPU_NN_MAX_LAYERS 8
enum { FWGPU_NN_INIT_XAVIER = 0, FWGPU_NN_INIT_HU = 1, FWGPU_NN_INIT_ONE = 2, FWGPU_NN_INIT_ZERO = 3 };
typedef struct fwgpu_nn_config {
    uint32_t n_layers;                     /* hidden layers */
    uint32_t width[FWGPU_NN_MAX_LAYERS];   /* --nn N:width:W (default 20) */
    uint32_t relu[FWGPU_NN_MAX_LAYERS];    /* --nn N:activation:relu -> 1, none -> 0 */
    uint32_t init[FWGPU_NN_MAX_LAYERS];    /* --nn N:init:... (default hu) */
    uint32_t topology;                     /* 1 = "one" (default, model_instance.rs:42), 2 = "two" */
    float nn_learning_rate, nn_power_t, nn_init_acc_gradient; /* model_instance.rs:426-428 */
} fwgpu_nn_config;
int fwgpu_set_nn(fwgpu_regressor *r, const fwgpu_nn_config *nn);

/* ---------------------------------------------------------------- weight (de)serialisation
 * fwgpu_serialized_len / fwgpu_write_weights / fwgpu_read_weights
 *      <= Regressor::write_weights_to_buf / overwrite_weights_from_buf    regressor.rs:426-469
 * Blob = u64 LE total element count (sum of get_serialized_len), then per block, in order:
 *   LR : 2^b x {f32 w, f32 acc}   (SGD: {f32 w})              block_lr.rs:257-275, block_helpers.rs:17-28
 *   FFM: len x f32 w, then len x f32 acc (SGD: no acc part)   block_ffm.rs:835-863
 *   NN : per dense layer, weights then optimizer state        block_neural.rs:430-448
 * fwgpu_serialized_len returns the byte size of that blob. */
int fwgpu_serialized_len(fwgpu_regressor *r, uint64_t *n_bytes);
int fwgpu_write_weights(fwgpu_regressor *r, uint8_t *buf, uint64_t cap, uint64_t *written);
int fwgpu_read_weights(fwgpu_regressor *r, const uint8_t *buf, uint64_t len);

/* Raw table access (tests, weight patching).  which: 0 = LR table as interleaved {w,acc} floats
 * (2*2^b floats), 1 = FFM weights, 2 = FFM optimizer state (each 2^ffm_bits + F*k floats,
 * block_ffm.rs:92-94).  Offsets and counts are in floats. */
enum { FWGPU_TABLE_LR = 0, FWGPU_TABLE_FFM_W = 1, FWGPU_TABLE_FFM_ACC = 2,
       FWGPU_TABLE_NN_W = 3, FWGPU_TABLE_NN_ACC = 4 /* all dense layers back to back: hidden layers, then the final neuron */ };
int fwgpu_table_len(fwgpu_regressor *r, int which, uint64_t *n_floats);
int fwgpu_table_read(fwgpu_regressor *r, int which, uint64_t offset, uint64_t count, float *host_out);
int fwgpu_table_write(fwgpu_regressor *r, int which, uint64_t offset, uint64_t count, const float *host_in);
int fwgpu_table_fill(fwgpu_regressor *r, int which, float value);
/* Order-independent checksum of a table computed on the device: sum over i of
 * mix64(i ^ bits(table[i])) (wrapping u64). */
int fwgpu_table_checksum(fwgpu_regressor *r, int which, uint64_t *checksum);
/* Device address of a table (for zero-copy interop, e.g. wrapping it for an RCCL all-reduce). */
int fwgpu_table_device_ptr(fwgpu_regressor *r, int which, void **dev_ptr);

/* ---------------------------------------------------------------- micro-batches
 * A batch is a CSR of FeatureBuffers resident in HBM: example i owns lr[lr_off[i]..lr_off[i+1]) and
 * ffm[ffm_off[i]..ffm_off[i+1]).  label/importance per example (feature_buffer.rs:187-189).
 * fwgpu_batch_create copies host arrays to the device (synchronously, on the null stream). */
int fwgpu_batch_create(fwgpu_regressor *r, const fwgpu_lr_entry *lr, const uint32_t *lr_off,
                       const fwgpu_ffm_entry *ffm, const uint32_t *ffm_off, const float *label,
                       const float *importance, uint32_t n_examples, fwgpu_batch **out);
int fwgpu_batch_free(fwgpu_batch *b);
int fwgpu_batch_size(const fwgpu_batch *b, uint32_t *n_examples, uint64_t *n_lr, uint64_t *n_ffm);
/* Synchronous micro-batch: every example of the batch is scored with the weights of the batch start, then all updates are
 * applied (FWGPU_MODE_SEQUENTIAL: in example order; FWGPU_MODE_HOGWILD: concurrently).  This is the step the owner-sharded
 * multi-GPU mode (fwgpu_dist_*) runs across GPUs, and the mode in which a deep head trains mini-batched on the matrix cores.
 * fwgpu_split = the device buffers of one such batch (n_examples x {field sums, own slots, gradients}). */
typedef struct fwgpu_split fwgpu_split;
int fwgpu_split_create(fwgpu_regressor *r, uint32_t n_examples, uint32_t max_ffm_per_example, fwgpu_split **out);
int fwgpu_split_free(fwgpu_split *sp);
int fwgpu_learn_batch_sync(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, void *hip_stream);
/* predict-only launches of this batch start every example's field sums from this context cache (NULL detaches it); a record
 * batch needs a cache that went through fwgpu_block_cache_cover_record */
int fwgpu_batch_set_cache(fwgpu_batch *b, const fwgpu_block_cache *cache);
/* Enqueue one pass over the batch on `stream`: for every example, Regressor::learn(fb, update)
 * (update=0: Regressor::predict).  Predictions land in the batch's device buffer. Asynchronous. */
int fwgpu_learn_batch(fwgpu_regressor *r, fwgpu_batch *b, int mode, int update, void *stream);
/* Copy the batch's predictions to the host (synchronises `stream`). */
int fwgpu_batch_predictions(fwgpu_batch *b, float *host_out, uint32_t n, void *stream);
/* Device pointer of the batch's prediction buffer (n_examples floats). */
int fwgpu_batch_predictions_device(fwgpu_batch *b, void **dev_ptr);

/* ---------------------------------------------------------------- record translation
 * FeatureBufferTranslator (feature_buffer.rs:33-44, 138-338) for primitive namespaces.
 * A translator describes which namespaces feed which LR combo / FFM field; namespaces are
 * identified by their index in the record (vwmap.rs:115-147). */
typedef struct fwgpu_translator_config {
    uint32_t n_combos;           /* feature_combo_descs.len() */
    const uint32_t *combo_off;   /* n_combos+1 offsets into combo_ns */
    const uint32_t *combo_ns;    /* namespace_index of each combo member */
    const uint8_t *combo_ns_f32; /* 1 if the namespace is NamespaceFormat::F32 */
    const float *combo_weight;   /* per combo */
    int32_t add_constant_feature;
    uint32_t n_fields;           /* ffm_fields.len() */
    const uint32_t *field_off;   /* n_fields+1 offsets into field_ns */
    const uint32_t *field_ns;
    const uint8_t *field_ns_f32;
    /* the ModelInstance fields FeatureBufferTranslator::new reads for its masks (feature_buffer.rs:138-148) */
    uint32_t bit_precision;
    uint32_t ffm_k;              /* 0: no ffm_buffer is produced */
    uint32_t ffm_bit_precision;
} fwgpu_translator_config;

/* lr_hash_mask / ffm_hash_mask as FeatureBufferTranslator::new computes them (feature_buffer.rs:138-148) */
uint32_t fwgpu_lr_hash_mask(uint32_t bit_precision);
uint32_t fwgpu_ffm_hash_mask(uint32_t ffm_bit_precision, uint32_t ffm_k);

/* Translate one record (parser.rs:57-74 layout) into caller-provided entry buffers (pure host code).
 * <= FeatureBufferTranslator::translate(record, example_number)   feature_buffer.rs:174-338 */
int fwgpu_translate(const fwgpu_translator_config *t, const uint32_t *record, uint32_t record_len, fwgpu_lr_entry *lr_out, uint32_t lr_cap, uint32_t *n_lr,
                    fwgpu_ffm_entry *ffm_out, uint32_t ffm_cap, uint32_t *n_ffm, float *label, float *importance);
/* Translate n back-to-back records (rec_off[i] = u32 offset of record i, n+1 entries) into a
 * device-resident batch. */
int fwgpu_batch_from_records(fwgpu_regressor *r, const fwgpu_translator_config *t, const uint32_t *records,
                             const uint64_t *rec_off, uint32_t n, fwgpu_batch **out);

/* Raw-record batch: the n records are copied to HBM as they are (4 x record_len bytes per example) and
 * FeatureBufferTranslator::translate (feature_buffer.rs:174-338) runs INSIDE the example kernel's stage phase, so the host
 * only validates the slot words.  Same results as fwgpu_batch_from_records, bit for bit. */
int fwgpu_record_batch_create(fwgpu_regressor *r, const fwgpu_translator_config *t, const uint32_t *records,
                              const uint64_t *rec_off, uint32_t n, fwgpu_batch **out);

/* ---------------------------------------------------------------- multi-GPU (replaces hogwild.rs:24-103 across GPUs)
 * One process per GPU over RCCL / xGMI.  fwgpu_dist_unique_id on one rank, the 128 bytes to every rank by any side channel
 * (the Rust caller of hogwild.rs:51-60 would use its own launcher's), then fwgpu_dist_init on every rank with its regressor
 * (every rank creates and initialises the SAME model: weight init is deterministic).
 *
 * fwgpu_dist_learn_sharded: the owner-sharded synchronous step.  Every rank owns a contiguous range of the FFM table and of
 * the LR table (fwgpu_dist_ranges) and brings a micro-batch of n records; the records are all-gathered, every rank gathers
 * the rows it owns for ALL examples, the partial field sums are reduce-scattered to the examples' home ranks, which compute
 * prediction and gradient; gradients and field sums are all-gathered and every rank applies the AdaGrad updates of the rows it
 * owns.  Semantics: fwgpu_learn_batch_sync with a micro-batch of n_ranks * n examples (every example sees the weights of the
 * step's start; per-occurrence AdaGrad on every row, as on one GPU).  Collective call: same n on every rank.
 * fwgpu_dist_gather_tables: every rank's owned range into every rank's tables (before saving or single-GPU prediction).
 * fwgpu_dist_all_reduce_sum: plain RCCL all-reduce of a device float buffer (replica mode: table deltas).
 * The fwgpu_dist_group_* functions run the same step with all ranks inside ONE process (collectives become device copies):
 * the tests and the single-box emulation of an N-GPU job use them. */
typedef struct fwgpu_dist fwgpu_dist;
typedef struct fwgpu_dist_group fwgpu_dist_group;
int fwgpu_dist_unique_id(uint8_t *id, uint64_t cap /* >= 128 */);
int fwgpu_dist_init(fwgpu_regressor *r, const uint8_t *unique_id, int rank, int n_ranks, fwgpu_dist **out);
int fwgpu_dist_free(fwgpu_dist *d);
/* FWGPU_MODE_HOGWILD (default): the examples of a phase run concurrently; FWGPU_MODE_SEQUENTIAL: in example order on one
 * workgroup (deterministic, for parity checks) */
int fwgpu_dist_set_mode(fwgpu_dist *d, int mode);
int fwgpu_dist_group_set_mode(fwgpu_dist_group *g, int mode);
/* Owner-side apply: sharded hogwild whose read-modify-writes never cross a link (what hogwild.rs:89-103 does with threads on one shared table,
 * with GPUs and owner-sharded tables): a rank FETCHES its examples' weight rows from their owners, and PUSHES one gradient row per occurrence
 * into a ring in the owner's memory; the owner runs the optimizer on its own tables.  fwgpu_dist_owner_attach maps tables and rings (rings sized
 * for steps of up to max_rows gradient rows / max_lr LR gradients per rank); fwgpu_dist_learn_owner is ONE COLLECTIVE STEP -- every rank calls
 * it, n may be 0 -- of push, count exchange, apply.  FWGPU_MODE_SEQUENTIAL with one example per step, ranks taking turns, is the sequential
 * reference.  The in-process group form takes every rank's micro-batch in one call. */
int fwgpu_dist_owner_attach(fwgpu_dist *d, uint32_t max_rows, uint32_t max_lr);
int fwgpu_dist_learn_owner(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                           float *predictions, int update);
int fwgpu_dist_group_learn_owner(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                 const uint64_t *const *rec_off, const uint32_t *n, float *const *predictions, int update);
/* STREAMING form of the owner-side apply: the owners drain their regions WHILE the sources' kernels fill them (a persistent consumer grid per owner,
 * circular regions, flow control through words the owner stores in the source's memory: no read-modify-write and no poll crosses a link), so that the
 * staleness of a gradient is the examples in flight -- as with hogwild.rs:89-103's threads -- whatever the step's size; the step-synchronous form above is
 * bounded to ~1024 global examples per step by stability.  HOGWILD only.  batches (or NULL): rank j's micro-batch already in HBM (a record batch of
 * its regressor; predictions land in the batch), else records / rec_off / n.  log2_rows / log2_lr: capacity of one (owner, source) region in gradient
 * rows / LR gradients (0: 2^15 / 2^16); consumer_workgroups: size of an owner's consumer grid (0: 48). */
int fwgpu_dist_group_learn_owner_stream(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records, const uint64_t *const *rec_off,
                                        const uint32_t *n, fwgpu_batch *const *batches, float *const *predictions, int update, uint32_t log2_rows,
                                        uint32_t log2_lr, uint32_t consumer_workgroups);
/* ... and its process-per-rank form: fwgpu_dist_owner_stream_attach (collective, once) maps tables and the streaming regions of every rank;
 * fwgpu_dist_learn_owner_stream is ONE COLLECTIVE STEP (every rank calls it; n may be 0) with two small collectives (shapes, final positions). */
int fwgpu_dist_owner_stream_attach(fwgpu_dist *d, uint32_t log2_rows, uint32_t log2_lr);
int fwgpu_dist_learn_owner_stream(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                                  float *predictions, int update, uint32_t consumer_workgroups);
int fwgpu_dist_rank(const fwgpu_dist *d, int *rank, int *n_ranks);
/* ranks of the job as the RCCL communicator itself counts them (ncclCommCount); 0 for a member of an in-process group */
int fwgpu_dist_comm_count(const fwgpu_dist *d, int *count);
int fwgpu_dist_ranges(const fwgpu_dist *d, uint32_t *ffm_lo, uint32_t *ffm_hi, uint32_t *lr_lo, uint32_t *lr_hi);
int fwgpu_dist_learn_sharded(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off,
                             uint32_t n, float *predictions);
/* the same step with the rank's micro-batch already resident in HBM (a record batch): predictions land in the batch */
int fwgpu_dist_learn_sharded_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b);
int fwgpu_dist_gather_tables(fwgpu_dist *d);
int fwgpu_dist_all_reduce_sum(fwgpu_dist *d, float *device_buf, uint64_t count, void *hip_stream);
int fwgpu_dist_group_create(fwgpu_regressor *const *regressors, int n_ranks, fwgpu_dist_group **out);
int fwgpu_dist_group_free(fwgpu_dist_group *g);
int fwgpu_dist_group_learn_sharded(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                   const uint64_t *const *rec_off, uint32_t n, float *const *predictions);
int fwgpu_dist_group_gather_tables(fwgpu_dist_group *g);
/* Row-sparse gradient buckets (north_star: "RCCL all-reduce of sparse gradient buckets"; replaces hogwild.rs:24-103's shared
 * table ACROSS GPUs): every rank keeps a full replica and scores its own micro-batch (any size) against it; the gradients of the
 * touched rows are summed per row on the rank (segment reduction over the sorted occurrence list), all-gathered as
 * {row hash, R-float gradient row} buckets, and every rank applies all buckets in the same order: ONE optimizer step per row and
 * global batch with the summed gradient.  Replicas that start identical stay bit-identical; there is no table exchange.
 * The update rule differs from the reference's (one step per occurrence): the oracle's fwo_learn_sparse restates it.
 * Models without a deep head.  Predictions: host buffer (may be NULL) / the batch's own. */
int fwgpu_dist_learn_sparse(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off,
                            uint32_t n, float *predictions);
int fwgpu_dist_learn_sparse_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b);
/* bucket rows the rank sent in its last sparse step: FFM rows (4-byte key + R floats each) and LR entries (key + float) */
int fwgpu_dist_sparse_last_rows(const fwgpu_dist *d, uint32_t *ffm_rows, uint32_t *lr_rows);
/* in-process group: rank j brings n[j] records */
int fwgpu_dist_group_learn_sparse(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                  const uint64_t *const *rec_off, const uint32_t *n, float *const *predictions);
/* Peer-sharded hogwild step (replaces hogwild.rs:24-103's shared table ACROSS GPUs with the reference's own update rule): the tables
 * are sharded by owner (rank s owns the FFM rows that START in its 1/N of 2^ffm_bits and the LR entries in its 1/N of 2^bits), every
 * rank runs the fused hogwild kernel on its own n[j] records and reaches each row IN ITS OWNER'S MEMORY -- a plain pointer on the
 * same device, a peer-mapped pointer over xGMI between the GPUs of one process.  No collective and no barrier between ranks:
 * per-occurrence AdaGrad steps, staleness = the examples in flight on all GPUs.  1, 2, 4 or 8 ranks; models without a deep head.
 * fwgpu_dist_group_set_mode(FWGPU_MODE_SEQUENTIAL): rank after rank, in example order -- the sequential reference algorithm over
 * the ranks' micro-batches in rank order (the deterministic form).  update = 0: predict only.  fwgpu_dist_group_gather_tables
 * afterwards gives every rank the whole model. */
int fwgpu_dist_group_learn_peer(fwgpu_dist_group *g, const fwgpu_translator_config *t, const uint32_t *const *records,
                                const uint64_t *const *rec_off, const uint32_t *n, float *const *predictions, int update);
/* The same mode with one PROCESS per rank (hogwild.rs:24-103 across the GPUs of a node, one process per GPU as RCCL jobs are launched):
 * fwgpu_dist_peer_attach -- collective, once -- exports this rank's tables as IPC handles (hipIpcGetMemHandle), all-gathers the handles
 * through the job's communicator and maps every other rank's tables (hipIpcOpenMemHandle: a peer GPU's memory over xGMI, or the same
 * device's when ranks share a GPU); fwgpu_dist_learn_peer then runs THIS rank's n records through the fused kernel, each row reached
 * in its owner's allocation -- not a collective, ranks run at their own pace; fwgpu_dist_barrier (collective) is where the caller
 * orders them: before fwgpu_dist_gather_tables, before a hold-out pass, or between ranks for a deterministic rank-after-rank run.
 * While the mode is on, a rank's OWNED range of the LR table lives in an allocation of its own (what the peers map: 1/N of the table --
 * hipIpcOpenMemHandle hangs on a 2 GiB allocation on ROCm 7.2); fwgpu_dist_gather_tables copies it back before it assembles the
 * tables, so fwgpu_predict / fwgpu_save on the rank's regressor see the trained LR weights only after that call. */
int fwgpu_dist_peer_attach(fwgpu_dist *d);
int fwgpu_dist_learn_peer(fwgpu_dist *d, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                          float *predictions, int update);
/* ... with the rank's micro-batch already in HBM (a record batch of its regressor); launched on hip_stream, or on the rank's own stream */
int fwgpu_dist_learn_peer_batch(fwgpu_dist *d, const fwgpu_translator_config *t, fwgpu_batch *b, int update, void *hip_stream);
int fwgpu_dist_barrier(fwgpu_dist *d);


/* ---------------------------------------------------------------- HogwildTrainer replacement
 * fwgpu_trainer_create  <= HogwildTrainer::new(regressor, &model_instance, num_workers)  hogwild.rs:24-49
 * fwgpu_digest_records  <= HogwildTrainer::digest_example(Vec<u32>)                      hogwild.rs:51-53
 *                          (n records per call; records are copied)
 * fwgpu_finish          <= HogwildTrainer::block_until_workers_finished                  hogwild.rs:55-60
 *                          (flushes the partial micro-batch and waits; the trainer stays usable)
 * Records are translated on the host, packed into micro-batches of `micro_batch` examples and run in
 * HOGWILD mode on an internal stream, double-buffered.  Like the reference, no predictions are
 * produced for hogwild-trained examples (main.rs:242-243). */
int fwgpu_trainer_create(fwgpu_regressor *r, const fwgpu_translator_config *t, uint32_t micro_batch,
                         fwgpu_trainer **out);
int fwgpu_digest_records(fwgpu_trainer *tr, const uint32_t *records, const uint64_t *rec_off, uint32_t n);
int fwgpu_finish(fwgpu_trainer *tr);
int fwgpu_trainer_free(fwgpu_trainer *tr);
int fwgpu_trainer_examples_seen(const fwgpu_trainer *tr, uint64_t *n);
/* The hold-out / test-only protocol of the reference's example loop (main.rs:184-185, 238-241, `--holdout_after N`, `-t`):
 * examples are numbered from 1 in the order they are digested; those numbered >= holdout_after (0 = none), or all of them
 * when testonly != 0, are predicted with update = false and never learned.  Their predictions are kept in stream order and
 * can be fetched after fwgpu_finish (out == NULL: count only).  Learned examples have no prediction here, like the
 * reference's hogwild branch (it prints 0.0, main.rs:242-243). */
int fwgpu_trainer_set_holdout(fwgpu_trainer *tr, uint64_t holdout_after, int testonly);
int fwgpu_trainer_predictions(fwgpu_trainer *tr, float *out, uint64_t cap, uint64_t *n);

/* ---------------------------------------------------------------- launch tuning (optional)
 * threads: workgroup size (multiple of 64, <=1024); workgroups_per_cu: persistent grid = CUs*this.
 * 0 keeps the default.  Does not change results in SEQUENTIAL mode. */
int fwgpu_set_launch(fwgpu_regressor *r, uint32_t threads, uint32_t workgroups_per_cu);
/* HOGWILD launches process as many examples concurrently as the device holds workgroups (768 on MI355X at config C).  On a
 * small or tiny-example data set that much staleness keeps the model from learning (hogwild.rs runs 16 threads): n caps
 * the examples in flight, n = 16 is the reference's default degree of concurrency, n = 1 is its single-thread loop (one
 * workgroup walks the stream in order: identical to FWGPU_MODE_SEQUENTIAL), 0 = no cap. */
int fwgpu_set_max_in_flight(fwgpu_regressor *r, uint32_t n_examples);

/* ---------------------------------------------------------------- multi-GPU replica bookkeeping (device pointers)
 * Data-parallel replicas exchange what each changed since the last agreed snapshot (replaces hogwild.rs's shared
 * memory across GPUs; the all-reduce itself is RCCL via torch.distributed, see fwumious_wabbit_amd/dist_sync.py):
 *   fwgpu_delta_start : local_delta = table - snapshot ; summed_delta = scale * local_delta
 *                       (summed_delta is then all-reduced in place; scale = 1/world_size makes the sum the MEAN of the
 *                        replicas' deltas, which is what keeps N >= 4 replicas from diverging: scripts/replica_sim.py)
 *   fwgpu_delta_finish: snapshot += summed_delta ; table += summed_delta - local_delta
 * One fused pass each over n_floats elements, enqueued on `stream`. */
int fwgpu_delta_start(const void *table, const void *snapshot, void *local_delta, void *summed_delta, uint64_t n_floats,
                      float scale, void *stream);
int fwgpu_delta_finish(void *table, void *snapshot, const void *local_delta, const void *summed_delta, uint64_t n_floats,
                       void *stream);

/* ---------------------------------------------------------------- diagnostics
 * Cross-XCD visibility probe for the access pattern the HOGWILD mode relies on: one workgroup publishes a
 * 1 KiB payload `iters` times (sc1 or plain stores), 15 workgroups spread over the XCDs re-read it (sc1 or
 * plain loads) after a device-scope flag and count words older than the flag.  With use_sc1=1 the count must
 * be 0; with use_sc1=0 it shows the hazard the kernels avoid. */
/* Per-phase shader-clock accounting of the example kernel (debug): enable=1 allocates/zeroes 8 device counters that
 * every workgroup's thread 0 adds to: [0] stage entries, [1] field boundaries + overlap scan, [2] row gather,
 * [3] dot + LR forward + sigmoid, [4] LR update, [5] FFM update, [6] wait for the slowest wave, [7] examples.
 * out16 (may be NULL) receives the counters accumulated so far.  The stamps are compiled into -DFW_TICKS builds of kernels.hip only
 * (scripts/build_variant.sh ticks -DFW_TICKS; scripts/perf_probe.py): the shipped kernels add nothing, the counters stay 0. */
int fwgpu_debug_phase_ticks(fwgpu_regressor *r, int enable, uint64_t *out16);
/* How the FFM accumulator table was placed relative to the weight table (fwgpu_create tries a few candidate allocations for
 * tables beyond the Infinity Cache and times the update's access pattern on each pair; FWGPU_PLACEMENT=0 disables it):
 * candidates tried (1: no search), fastest and slowest pair probe in milliseconds (0 when nothing was timed). */
int fwgpu_debug_placement(const fwgpu_regressor *r, int *tries, float *ms_fastest, float *ms_slowest);
/* 0 = automatic kernel choice, 1 = force the generic kernel (v1), 2 = register-resident rows (v2) where applicable. */
int fwgpu_debug_set_kernel_version(fwgpu_regressor *r, int version);
/* Tuning switches for experiments.  option 1: value 1 = read the AdaGrad LUT from global memory instead of an LDS copy.
 * option 2: the update path of large models (k % 4 == 0, rows of at most 256 floats): 0 = float-granular row updates, rows repeated
 *   inside an example serialised on one wave (the round-1 path); 1 = automatic (default): tables beyond the Infinity Cache take the
 *   chained path (repeated rows applied by their first occurrence's wave, from registers), with whole-128-byte-line accesses only
 *   when the accumulator table could not be placed away from the weight table; 2 = chained path with whole-line accesses, always;
 *   3 = chained path with float-granular accesses, always (what large tables run when the placement search succeeded, on any table).
 * option 3: value 1 = no duplicate-row chains (A/B runs).
 * option 4: HOGWILD launches step the constant feature's LR entry (in every example, feature_buffer.rs:270-276) with atomics: the
 *   step is taken at the accumulator the example's forward pass read + its own g^2, `acc += g^2` and `w -= step` are added to the
 *   table by float atomics.  value 1 (default): both per example; value n > 1: a workgroup's WEIGHT deltas stay pending in LDS for
 *   n of its examples; 0 = plain per-example read-modify-writes, which serialise on that one entry and overwrite each other.
 *   SEQUENTIAL launches never use it.
 * option 5: store policy of the FFM row stores in HOGWILD launches of that update path: 0 = both tables device-scope write-through,
 *   1 = weight rows write-back through the XCD's L2, 2 = both tables write-back, 3 (the default since round 5) = 1 with thinned accumulator stores on hot
 *   register-kept rows (one example in eight stores eight times its g^2: DESIGN 4.2), 4 = 3 with the thinned store replaced by a thinned device-scope
 *   atomic add (one example in eight adds eight times its g^2, on every hot row of the example: nothing is lost to a concurrent writer); -1 = the build's
 *   default.  option 9: policies 3 / 4 call a row hot once its accumulators have grown by more than value / 1024 (-1 = default: 0.5); option 10: one example in
 *   2^value touches a hot row's accumulators (0..6; -1 = default: 3).  option 6: with
 *   policy 1 / 2 a workgroup writes its XCD's dirty L2 lines back every `value` of its examples (0 = only when the launch ends; -1 = the
 *   build's default: 128) -- the bound on how long a popular row can stay private to one XCD (DESIGN.md 4.2,
 *   tests/test_gpu_conservation.py).  SEQUENTIAL launches are exact under every policy.
 * option 13: rows the large-table kernel keeps from the gather (20 in registers + 3 in LDS per wave, written back as w_gather - step): 1 (default, -1) = kept; 0 = none, every row is
 *   re-read by the update (14 % slower; without the damping of rows many concurrent examples hold -- on BASELINE configs[2]'s stream the reference's own hold-out curve).
 * option 12: store policy 4 also on hot LR entries of that kernel (the weight stored alone, the accumulator by thinned atomic adds): 1 (default, -1) / 0.
 * option 11: HOGWILD launches of a model with a deep head and rows of 257..512 floats (BASELINE config E: k = 16 at 30 fields): 1 (default, -1) = the head runs
 *   as a phase of the large-table kernel, two 512-thread workgroups per CU, where they fit; 0 = always on the generic kernel (one 1024-thread workgroup per CU).
 *   SEQUENTIAL launches take the generic kernel (the parity mode) unless value 2 forces them onto the large-table kernel too (tests of that kernel's head phase).
 * option 7: value 0 = the updating launches do not prefetch the next example's record (A/B runs; default 1).
 * option 8: rows per wave, beyond the 20 kept in registers, whose gather-time weights are parked in LDS for the update phase instead of
 *   being re-read (config-C-shaped rows on the chained path): 0..3, -1 (default) = as many as still let two workgroups share a CU.
 * (The update path of option 2 = 1 / 2 keeps the first 20 rows of every wave's share of an example from the gather and writes
 *   them back as w_gather - step in HOGWILD launches: what the concurrent mode's hold-out loss rests on) */
int fwgpu_debug_set_option(fwgpu_regressor *r, int option, int value);
/* an f32 as serde_json / ryu prints it in the embedded JSON documents ("0.1", "1.0", "1e-7"); NUL-terminated */
int fwgpu_debug_format_f32(float v, char *buf, uint32_t cap);
int fwgpu_debug_coherence_probe(int device, int use_sc1, uint32_t iters, uint32_t *stale_words, uint32_t *timeouts);
/* One product of the mini-batched head (head.hip) on device pointers, for tests against a plain f32 reference:
 * C[M, N] = op(A) . op(B) with A(m, k) = ta ? A[k * lda + m] : A[m * lda + k], B(k, n) = tb ? B[n * ldb + k] : B[k * ldb + n]; epilogue 0: C = acc,
 * 1: C = relu(acc + bias[n]) with the 0 / 1 mask in aux, 2: C = acc * aux[m, n], 3: C += acc.  tiled != 0 forces the LDS-tiled 64 x 64 kernel;
 * otherwise the split-K kernel runs where K % 8 == 0 and the operands allow 16-byte loads.  The three (ta, tb) pairs the head uses: (0, 1), (1, 0), (0, 0). */
int fwgpu_debug_head_gemm(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc, int ta, int tb, int epilogue,
                          const float *bias, float *aux, int relu, int tiled, void *stream);

/* ---------------------------------------------------------------- feed path: namespace map, VW text parser, input cache
 * (SURVEY.md 8 f1/f3).  Host-side code; none of it needs a device.
 *
 * vw_namespace_map.csv (vwmap.rs:106-151): "vwname,verbose[,f32]" per line, namespace_index = line number,
 * optional "_namespace_skip_prefix,N".  The JSON form is serde_json::to_vec_pretty(vw_source) as embedded in cache and
 * model files (persistence.rs:36-53); fwgpu_vwmap_to_json reproduces it byte for byte (buf == NULL: size query). */
typedef struct fwgpu_vwmap fwgpu_vwmap;
int fwgpu_vwmap_from_csv(const char *csv, uint64_t len, fwgpu_vwmap **out);
int fwgpu_vwmap_from_json(const char *json, uint64_t len, fwgpu_vwmap **out);
void fwgpu_vwmap_free(fwgpu_vwmap *vw);
uint32_t fwgpu_vwmap_num_namespaces(const fwgpu_vwmap *vw); /* max namespace_index + 1 (vwmap.rs:83-88) */
uint32_t fwgpu_vwmap_num_entries(const fwgpu_vwmap *vw);
int fwgpu_vwmap_to_json(const fwgpu_vwmap *vw, char *buf, uint64_t cap, uint64_t *len);
/* name -> namespace_index / format; verbose != 0 looks the verbose name up (map_verbose_to_namespace_descriptor) */
int fwgpu_vwmap_lookup(const fwgpu_vwmap *vw, const char *name, uint64_t len, int verbose, uint32_t *index,
                       uint32_t *is_f32);

/* VowpalParser (parser.rs:24-461).  fwgpu_parser_parse_line = next_vowpal_to_size on one line as read_until(b'\n')
 * delivers it (the newline, when present, is part of `line`); the record (parser.rs:57-74) is copied to `out`
 * (out == NULL: only *n_words).  len == 0 -> end of stream, *n_words = 0.  Returns FWGPU_OK, FWGPU_PARSE_FLUSH,
 * FWGPU_PARSE_HOGWILD_LOAD (file name via fwgpu_parser_command_argument) or FWGPU_ERR_PARSE with the reference's
 * message in fwgpu_last_error().  fwgpu_parser_parse_with_prefix = next_vowpal_with_cache (parser.rs:195-211): the
 * cached context bytes followed by the request's bytes.  fwgpu_parser_parse_buffer parses many lines into records laid
 * out back to back (rec_off: n_records + 1 word offsets), stopping at max_records, a full buffer, or the first line
 * that is not an example (*consumed = bytes fully parsed; the return value says why it stopped). */
typedef struct fwgpu_parser fwgpu_parser;
int fwgpu_parser_create(const fwgpu_vwmap *vw, fwgpu_parser **out);
void fwgpu_parser_free(fwgpu_parser *p);
int fwgpu_parser_parse_line(fwgpu_parser *p, const char *line, uint64_t len, uint32_t *out, uint32_t cap, uint32_t *n_words);
int fwgpu_parser_parse_with_prefix(fwgpu_parser *p, const char *prefix, uint64_t prefix_len, const char *line, uint64_t len,
                                   uint32_t *out, uint32_t cap, uint32_t *n_words);
/* The same result as fwgpu_parser_parse_with_prefix(prefix, line) for every line, without scanning the context again per
 * request (lib.rs:88-108 scans context + candidate every time): the context is scanned once up to its last token boundary,
 * requests resume there.  Contexts that cannot be resumed (fwgpu_parse_prefix_resumable == 0) take the concatenating route. */
typedef struct fwgpu_parse_prefix fwgpu_parse_prefix;
int fwgpu_parse_prefix_create(fwgpu_parser *p, const char *prefix, uint64_t len, fwgpu_parse_prefix **out);
void fwgpu_parse_prefix_free(fwgpu_parse_prefix *px);
int fwgpu_parse_prefix_resumable(const fwgpu_parse_prefix *px);
int fwgpu_parser_parse_after_prefix(fwgpu_parser *p, const fwgpu_parse_prefix *px, const char *line, uint64_t len, uint32_t *out,
                                    uint32_t cap, uint32_t *n_words);
/* Candidate-only records.  When the scanned part of the context is all of it (fwgpu_parse_prefix_is_record: its output buffer
 * equals the context's own record), fwgpu_parser_parse_candidate returns what the request ADDED to that record (*is_delta = 1):
 * header, a slot word per namespace the request filled (every other slot NO_FEATURES = "as in the context"), the request's
 * feature words.  A request that continues a namespace the context began comes back merged (*is_delta = 0). */
int fwgpu_parse_prefix_is_record(const fwgpu_parse_prefix *px, const uint32_t *record, uint32_t len);
int fwgpu_parser_parse_candidate(fwgpu_parser *p, const fwgpu_parse_prefix *px, const char *line, uint64_t len, uint32_t *out,
                                 uint32_t cap, uint32_t *n_words, int *is_delta);
const char *fwgpu_parser_command_argument(const fwgpu_parser *p);
int fwgpu_parser_parse_buffer(fwgpu_parser *p, const char *text, uint64_t len, uint32_t *words, uint64_t words_cap,
                              uint64_t *rec_off, uint64_t max_records, uint64_t *n_records, uint64_t *n_words,
                              uint64_t *consumed);

/* RecordCache (cache.rs:54-232).  fwgpu_cache_open(input, vw): if "<input>.fwcache" exists and its header ("FWCA",
 * version 11, vw_source equal to `vw`) verifies, the cache is opened for reading; otherwise "<input>.fwcache.writing" is
 * created for writing and renamed by fwgpu_cache_write_finish (cache.rs:70-131, 146-152).  Inputs whose name ends in
 * "gz" use an LZ4 frame stream (cache.rs:73).  fwgpu_cache_next_records = bulk get_next_record: whole records into
 * `words`, rec_off[0..n_records] word offsets; *n_records == 0 at end of file. */
typedef struct fwgpu_cache fwgpu_cache;
int fwgpu_cache_open(const char *input_filename, const fwgpu_vwmap *vw, fwgpu_cache **out);
int fwgpu_cache_is_reading(const fwgpu_cache *c);
int fwgpu_cache_is_writing(const fwgpu_cache *c);
int fwgpu_cache_push_records(fwgpu_cache *c, const uint32_t *words, uint64_t n_words);
int fwgpu_cache_write_finish(fwgpu_cache *c);
int fwgpu_cache_next_records(fwgpu_cache *c, uint32_t *words, uint64_t words_cap, uint64_t *rec_off, uint64_t max_records,
                             uint64_t *n_records, uint64_t *n_words);
void fwgpu_cache_free(fwgpu_cache *c);
/* The example loop over a cache file (main.rs:213-270: get_next_record -> digest_example) in native code: up to max_records
 * (0 = to the end of the file) are read, copied to pinned memory and learned; a reader thread overlaps the file with the
 * device.  Call fwgpu_finish afterwards, as after fwgpu_digest_records. */
int fwgpu_trainer_digest_cache(fwgpu_trainer *tr, fwgpu_cache *cache, uint64_t max_records, uint64_t *n_digested);
/* The same loop over VW TEXT (no cache, or the cache-writing first pass of `-c`): `text` is cut at line breaks into one
 * slice per host thread (threads == 0: 8), each parsed by a clone of `parser`; the records are learned in the original
 * order and appended to `cache` when it is non-NULL and open for writing.  Stops at the first line that is not an
 * example: the return value says why (FWGPU_PARSE_FLUSH, FWGPU_PARSE_HOGWILD_LOAD, FWGPU_ERR_PARSE), *consumed = bytes
 * fully digested, *n_examples = examples learned. */
int fwgpu_parser_clone(const fwgpu_parser *src, fwgpu_parser **out);
/* create_buffered_input (buffer_handler.rs:8-37): ".vw" plain, ".gz" gzip (several members allowed, MultiGzDecoder), ".zst"
 * zstd (libzstd.so.1 resolved at run time); any other extension is refused with the reference's message.
 * fwgpu_input_read fills `buf` with decompressed bytes, *n == 0 at the end.  fwgpu_trainer_digest_file = the example loop
 * over such a file: 64 MiB windows cut at line breaks -> fwgpu_trainer_digest_text. */
typedef struct fwgpu_input fwgpu_input;
int fwgpu_input_open(const char *filename, fwgpu_input **out);
int fwgpu_input_read(fwgpu_input *in, char *buf, uint64_t cap, uint64_t *n);
void fwgpu_input_close(fwgpu_input *in);
int fwgpu_trainer_digest_file(fwgpu_trainer *tr, fwgpu_parser *parser, fwgpu_cache *cache, const char *filename,
                              uint32_t threads, uint64_t *n_examples);
int fwgpu_trainer_digest_text(fwgpu_trainer *tr, fwgpu_parser *parser, fwgpu_cache *cache, const char *text, uint64_t len,
                              uint32_t threads, uint64_t *n_examples, uint64_t *consumed);

/* ---------------------------------------------------------------- model files
 * File = "FWRE", u32 version 6, u64 + JSON(vw_source), u64 + JSON(ModelInstance), weights blob (persistence.rs:17-97,
 * regressor.rs:426-469).  ModelInstance (model_instance.rs:47-97) crosses the boundary as an opaque handle made from /
 * rendered to that JSON; fwgpu_mi_to_json reproduces serde_json::to_vec_pretty (nn layer keys sorted: the reference's
 * HashMap order is arbitrary).  fwgpu_mi_configs fills the structs fwgpu_create / fwgpu_set_nn / the translator take
 * (any of the three may be NULL); their array pointers stay valid while the handle lives.  Features of ModelInstance this
 * path does not implement (transformed namespaces, nn topologies four/five, layernorm, dropout, maxnorm) are
 * rejected there, never ignored. */
typedef struct fwgpu_model_instance fwgpu_model_instance;
int fwgpu_mi_from_json(const char *json, uint64_t len, fwgpu_model_instance **out);
void fwgpu_mi_free(fwgpu_model_instance *mi);
int fwgpu_mi_to_json(const fwgpu_model_instance *mi, char *buf, uint64_t cap, uint64_t *len);
int fwgpu_mi_configs(fwgpu_model_instance *mi, int device, fwgpu_config *cfg, fwgpu_translator_config *tr,
                     fwgpu_nn_config *nn);
int fwgpu_mi_set_inference(fwgpu_model_instance *mi, int dequantize_weights); /* main.rs:141-145 */
/* save_regressor_to_filename (persistence.rs:73-89); quantize_weights: FFM weights as f16 buckets (block_ffm.rs:835-848) */
int fwgpu_model_save(const char *path, const fwgpu_vwmap *vw, const fwgpu_model_instance *mi, fwgpu_regressor *r,
                     int quantize_weights);
/* load_regressor_without_weights (persistence.rs:91-125): header only, no device */
int fwgpu_model_read_header(const char *path, fwgpu_vwmap **vw, fwgpu_model_instance **mi);
/* new_regressor_from_filename (persistence.rs:127-174).  *r == NULL: a regressor is created on `device` (immutable != 0:
 * optimizer forced to SGD, only the weights kept).  *r != NULL: hogwild_load (persistence.rs:176-187), the weights
 * are loaded into the existing regressor.  vw / mi may be NULL. */
int fwgpu_model_load(const char *path, int device, int immutable, fwgpu_vwmap **vw, fwgpu_model_instance **mi,
                     fwgpu_regressor **r);
/* --convert_inference_regressor (main.rs:136-148): training file -> inference file; host only */
int fwgpu_model_convert_inference(const char *in_path, const char *out_path, int quantize_weights);
/* quantization.rs:42-98: out = 8-byte header {f32 increment, f32 min} + n f16 bucket numbers (8 + 2n bytes) */
int fwgpu_quantize_ffm_weights(const float *weights, uint64_t n, uint8_t *out, uint64_t cap);
int fwgpu_dequantize_ffm_weights(const uint8_t *in, uint64_t n, float *weights);

/* ---------------------------------------------------------------- synthetic record streams
 * Generates records in the parser's output format (parser.rs:57-74) for the BASELINE.json configs:
 * n_namespaces namespaces == fields; per namespace 1+Poisson(mean_extra) features (exactly 1 when
 * mean_extra==0); ids ~ Zipf(zipf_s) over ids_per_ns; feature hash = murmur3(decimal id,
 * seed=murmur3(namespace name)) & MASK31 as the parser does (parser.rs:82-87, 382-385); a feature has
 * value 1.0 with probability 1-p_weighted else U(0.5,2); label from a fixed random teacher.
 * Two-call protocol: call with records==NULL to get the sizes, then with buffers. */
typedef struct fwgpu_synth_config {
    uint32_t n_namespaces;
    float mean_extra;
    double zipf_s;
    uint32_t ids_per_ns;
    float p_weighted;
    uint64_t seed;
} fwgpu_synth_config;
int fwgpu_synth_records(const fwgpu_synth_config *cfg, uint64_t first_example, uint32_t n, uint32_t *records,
                        uint64_t records_cap, uint64_t *rec_off, uint64_t *n_words);
uint32_t fwgpu_murmur3_32(const uint8_t *data, size_t len, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
