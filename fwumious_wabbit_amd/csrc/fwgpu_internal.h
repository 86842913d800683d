// Internal declarations shared by the host side (regressor.cpp, translate.cpp, trainer.cpp) and the
// HIP kernels (kernels.hip).  Not part of the C ABI (include/fwgpu.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "fwgpu.h"

namespace fwgpu {

constexpr int kLutBits = 11;  // optimizer.rs:98 FASTMATH_LR_LUT_BITS
constexpr int kLutSize = 1 << kLutBits;
constexpr uint32_t kFfmContraBufLen = 41472;  // regressor.rs:23 FFM_CONTRA_BUF_LEN

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
#define FWGPU_HIP(call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return ::fwgpu::fail(FWGPU_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

// Device copy of a FeatureBufferTranslator (feature_buffer.rs:33-44): which namespaces feed which LR combo / FFM field.
struct DevTranslator {
    const uint32_t *combo_off;   // [n_combos+1] into combo_ns
    const uint32_t *combo_ns;    // namespace index of each combo member
    const uint8_t *combo_f32;    // member namespace is NamespaceFormat::F32
    const float *combo_w;        // [n_combos] combo weight
    const uint32_t *pair_ns;     // flattened (field, namespace) pairs in field order
    const uint8_t *pair_f32;
    const uint8_t *pair_field;   // field index of each pair
    uint32_t n_combos, n_pairs, add_const, lr_mask, ffm_mask;
    uint32_t n_members;          // combo_off[n_combos]
};

// Deep head: BlockCopy -> [BlockNeuronLayer -> BlockRELU]* -> Join -> single neuron (regressor.rs:191-320)
constexpr int kNnMaxLayers = 8;
struct DevNN {
    uint32_t n_layers;  // hidden layers; 0 = no deep head
    uint32_t topology;  // 1 = "one" (final neuron sees [h_last, x]), 2 = "two" (h_last only)
    uint32_t in[kNnMaxLayers + 1], out[kNnMaxLayers + 1], off[kNnMaxLayers + 1];  // per layer incl. the final neuron;
                                                                                   // off = float offset into w / acc
    uint32_t relu[kNnMaxLayers];
    uint32_t X;          // length of x = LR slots + triangle
    uint32_t sum_width;  // sum of the hidden widths
    uint32_t max_in;     // largest layer input (LDS scratch)
    uint32_t max_out;    // widest layer (LDS scratch: the active-neuron list of the dense backward)
    float rate, minus_power_t;
    float *w, *acc;      // all layers back to back: per layer W[j*in+i] then the biases (block_neural.rs:86-88)
    const float *lut;
};

// peer-sharded tables: the owners' table bases (device pointers valid on the launching device: same device, or peer-mapped)
struct PeerShards {
    uint32_t n, shift_ffm, shift_lr, pad_;
    float *ffm_w[8], *ffm_acc[8], *lr[8];
};

// owner-side apply (dist.cpp fwgpu_dist_*_owner): where a SOURCE rank's kernel puts the gradient rows of its examples -- one region per owner, in
// the OWNER's memory (same device, or peer / IPC mapped), written by this source only; positions come from counters in the source's own memory
struct PushRings {
    uint32_t n, cap_ffm, cap_lr, pad_;
    uint32_t *cnt;  // [2n + 1] source-local: rows pushed to owner o, LR entries pushed to owner o (at n + o), then an overflow flag
    uint32_t *ffm_key[8];  // per owner: row hash of every pushed gradient row ...
    float *ffm_rows[8];    // ... and its R floats (block_ffm.rs:278: general gradient x gradient cache)
    uint2 *lr_ent[8];      // per owner: {hash, gradient bits} of every pushed LR entry (block_lr.rs:140-147)
    // ---- streaming form (fwgpu_dist_*_owner_stream): the regions are CIRCULAR and the owners drain them while this kernel runs.  Positions (cnt) run on
    // from step to step; position q lives in slot q & (cap - 1) for generation q >> log2cap.  A gradient row is complete when its tag word -- hash |
    // (generation + 1) << 32, stored AFTER the row's floats have been acknowledged -- says so; a slot may be written for generation g when the owner has
    // said so in THIS rank's memory (ffm_free: per-slot generation; LR: a consumed-prefix credit): no read-modify-write and no poll ever crosses a link.
    uint32_t stream, log2cap_ffm, log2cap_lr, pad2_;
    unsigned long long *ffm_tag[8];  // per owner: [cap_ffm] tag words of this source's ring in the owner's memory
    unsigned long long *lr_word[8];  // per owner: [cap_lr] {hash (30 bits) | ((generation % 3) + 1) << 30, gradient bits << 32}
    const uint32_t *ffm_free[8];     // source-local, per owner: [cap_ffm] the generation slot q may be written for (the owner's consumer stores it)
    const uint32_t *abort;           // a word the HOST sets when a step outlives its deadline (a peer rank is gone, or never launched): every wait loop of the streaming
                                     // form -- producers waiting for a free slot, consumers waiting for a tag or a final position -- looks at it and leaves (dist.cpp stream_finish)
    const uint32_t *lr_free[8];      // source-local, per owner: [cap_lr] the generation LR slot q may be written for
    // The rank's consumers run INSIDE its example kernel: the first `consumers` workgroups of the launch drain this rank's regions as owner (`own`), the
    // others are the producers.  A producer workgroup that finds no more examples counts itself in `done`; the last one stores every region's final
    // position -- tagged with the step -- into the OWNERS' memory (fin_remote[o] + this rank's index, then + n for the LR region): the owners' consumers
    // leave when they have reached it.  No host round trip, no collective, no second stream.
    uint32_t consumers, step, src, pad3_;
    uint32_t *done;                          // source-local counter of finished producer workgroups (zeroed before the launch)
    unsigned long long *fin_remote[8];       // per owner: its array of final positions [2n]: {step << 32 | position}
    const struct OwnerStream *own;           // device memory: this rank's regions as owner
};

// the owner's side of the streaming form: what its consumer kernel drains (kernels.hip owner_stream_kernel)
struct OwnerStream {
    uint32_t n, R, log2cap_ffm, log2cap_lr;
    const unsigned long long *ffm_tag[8];  // per source: this owner's ring of that source (owner-local memory, written by the source)
    const float *ffm_rows[8];
    const unsigned long long *lr_word[8];
    uint32_t *ffm_free[8];                  // per source: slot -> generation, in the SOURCE's memory
    uint32_t *lr_credit[8];                 // (unused)
    uint32_t *lr_free[8];                   // per source: LR slot -> generation, in the SOURCE's memory
    uint32_t start_ffm[8], start_lr[8];     // first position of this step, per source
    const unsigned long long *fin;          // owner-local [2n]: {step << 32 | final position}, stored by source s's last producer workgroup (FFM regions, then LR)
    uint32_t step, pad_;                    // the step this launch belongs to (fin words of other steps are not this launch's)
    const uint32_t *abort;                  // as PushRings::abort
    float *w, *acc, *lr;
    float ffm_rate, ffm_mpt, lr_rate, lr_mpt;
    const float *lut_ffm, *lut_lr;
};

// Everything the example kernel needs, passed by value.
struct KernelParams {
    // ---- tables (HBM) ----
    float *ffm_w;          // [2^ffm_bits + F*k]            block_ffm.rs:40
    float *ffm_acc;        // same length                    block_ffm.rs:41
    float *lr;             // [2^b] x {w, acc} interleaved   block_lr.rs:20
    const float *lut_lr;   // [2048]                         optimizer.rs:101-103
    const float *lut_ffm;  // [2048]
    // ---- batch (HBM, CSR) ----
    const uint32_t *ffm_hash;
    const float *ffm_val;
    const uint8_t *ffm_fld;  // field index (contra_field_index / k)
    const uint32_t *ffm_off;  // [n+1]
    const uint32_t *lr_hash;
    const float *lr_val;
    const uint32_t *lr_off;  // [n+1]
    const uint16_t *lr_combo;  // combo slot of each LR entry (only read when the deep head is on)
    const float *label;
    const float *importance;
    float *pred;
    uint32_t n_examples;
    // ---- or: raw records (parser.rs:57-74), translated on the device in the stage phase (records != NULL)
    const uint32_t *records;
    const uint64_t *rec_off;  // [n+1] u32-word offsets
    uint32_t max_rec;         // longest record of the batch, in words (LDS capacity)
    int32_t rec_self_len;     // record lengths come from the records' word 0 (gathered batches have gaps between ranks)
    DevTranslator tr;
    DevNN nn;
    uint32_t num_combos;
    // ---- model ----
    uint32_t F, k, R;          // fields, ffm_k, R = F*k (row length in floats)
    uint32_t max_ffm, max_lr;  // LDS capacities for one example's entries
    int32_t has_lr;            // wiring REGRESSOR: LR block participates
    int32_t update;
    int32_t aligned4;          // every FFM row of the batch starts on a 16-byte boundary
    int32_t window;            // the v2 kernel's update takes the window path (kernels.hip update_rows_win: duplicate-row chains, no resident rows)
    int32_t line_pass;         // ... and reads / writes back the WHOLE 128 B lines a row touches (0: float-granular accesses)
    uint32_t k_log2;           // log2(k) when k is a power of two, else 0xff
    int32_t no_chain;          // debug option 3: duplicate rows are serialised in phase B instead of chained (A/B runs)
    uint32_t hot_lr_hash;      // hogwild launches: the LR entry every example holds (the constant feature's), see hot_lr_flush (kernels.hip)
    uint32_t hot_lr_every;     // its pending steps reach the table every this many examples of a workgroup (0: off, plain read-modify-writes)
    int32_t chain;             // rows of the same hash inside one example are chained to the first and applied from registers (set with window)
    float lr_rate, lr_minus_power_t;    // SGD / AdagradFlex parameters of the LR block
    float ffm_rate, ffm_minus_power_t;  // ... of the FFM block
    int32_t lut_global;                 // 1: AdaGrad LUT read from global memory (through L1) instead of an LDS copy
    int32_t t_global;                   // phase kernels (FWD / UPD of the synchronous pipeline): the field sums T are written to / read from the example's split record in HBM directly, not staged in LDS (launch_example_phase)
    int32_t lut_lds_forced;             // ... unless the launch runs a kernel that keeps it in LDS as a compile-time fact (resolve_row_mode)
    int32_t store_policy;               // hogwild launches of the v2 window kernel: how FFM row stores reach memory (kernels.hip "store policy"): 0 = both tables
                                        // device-scope write-through, 1 = weights write-back through the XCD's L2, 2 = both tables write-back
    float acc_hot_theta;                // policy 3: a kept row whose accumulators exceed this is "hot": its accumulator row is stored for one example in
    int32_t no_kept_rows;               // the large-table kernel without rows kept from the gather (fwgpu_debug_set_option 13 / FWGPU_KEPT_ROWS=0): every row re-read by the update
    float lr_hot_theta;                 // store policy 4 on the LR block: an entry whose accumulator exceeds this is hot (its initial value + the FFM rows' theta)
    int32_t lr_thin;                    // ... on (1) / off (0): FWGPU_LR_THIN, fwgpu_debug_set_option 12
    int32_t nn_v2;                      // deep head as a phase of the v2 kernel's two-chunk instantiation (config E's concurrent launches; prepare_launch decides)
    int32_t thin_reread;                // policy 3 also on the re-read rows of the two-chunk instantiations (update_rows_win): FWGPU_THIN_REREAD (default 1)
    uint32_t acc_sample_log2;           // 2^acc_sample_log2 only, with that many times the example's g^2 (an unbiased, thinned write-through: kernels.hip "store policy")
    uint32_t wb_flush_every;            // policies 1 / 2: a workgroup writes its XCD's dirty L2 lines back (buffer_wbl2 sc1) every this many of its examples (0: never)
    int32_t tr_lds;                     // record batches on the v2 kernel: the translator's tables are read from an LDS copy (kernels.hip TrLds)
    uint32_t lds_keep, lds_keep_words;  // config-C-shaped kernel: rows per wave BEYOND the register-kept ones whose gather-time w stays in LDS for the update (0..FW_LDS_KEEP_MAX), and the region's size (rows x waves x R floats)
    int32_t prefetch;                   // record batches on the v2 kernel: example n+1's record is copied to LDS while example n is in its dot / update phases
    uint32_t *work;                     // next example to process (zeroed before every launch)
    uint32_t host_cus, host_wgs_cap, host_grid_cap;    // host side only: persistent grid = occupancy x CUs (capped), when launched with grid 0
    uint32_t grid_wgs;                                 // workgroups of THIS launch, filled in by launch_persistent: the example kernels read it here where they need it (gridDim.x, an implicit argument, is loaded once in the prologue and then lives in a scalar register across the whole example loop)
    uint32_t host_share;                               // host side only: ranks whose kernels must be RESIDENT TOGETHER on this device (streaming owner-side apply on a shared device): the persistent grid is an equal share of what the device holds (0 / 1: all of it)
    uint32_t host_stream_min_consumer_waves;           // host side only: floor of the automatic consumer count (two waves per source)
    uint32_t host_stream_max_consumer_waves;           // host side only: bound on the consumer waves of a streaming launch (a stripe's stride must stay well below a region's capacity)
    uint32_t host_extra_wgs;                           // host side only: (0xffffffff: a share of the grid chosen at launch, written to PushRings::consumers before the kernel starts)
                                                       // host side only: workgroups on top of the example workgroups (streaming owner-side apply: the consumers; at least one producer workgroup is kept)
    int32_t kernel_version;             // 0 = auto, 1 = force the v1 kernel, 2 = v2 where applicable
    unsigned long long *ticks;          // optional [8] per-phase shader-clock accumulators (debug), else NULL
#ifndef FW_KP_NO_CANARY                 // (debug builds of scripts/kp_size_exp.sh drop the two fields: sizeof(KernelParams) 744 -> 728)
    uint32_t *dbg_canary;               // debug (FWGPU_DBG_LDS_CANARY): device counter of LDS canary words found changed, else NULL
    uint32_t dbg_canary_off;            // ... byte offset of the 1 KiB canary behind the kernel's own LDS layout
#endif
#ifdef FW_KP_PAD                        // debug builds (scripts/kp_size_exp.sh): the struct's size as a variable of the group-concurrency fault
    unsigned char kp_pad[FW_KP_PAD];
#endif
    // ---- serving context cache (regressor.rs:397-423, block_ffm.rs:442-782): the context features' field sums, in T's layout
    const float *ctx_T;                 // [F*R] T[z][f][k] partial sums of the cached features (NULL: none); read-only launches only
    const float *ctx_dcf;               // [F]   their self-pair corrections
    const uint32_t *ctx_rec;            // candidate-only record batches: the context's record (namespaces a candidate's record lacks come from it)
    uint32_t ctx_rec_len;
    const uint32_t *ctx_cover;          // record batches: bit per namespace slot whose FFM features the cache already holds (the stage phase skips them)
    float *emit_T;                      // setup_cache: example 0's T and dcf are written here after the gather
    float *emit_dcf;
#if defined(FW_KP_PAD_POS) && FW_KP_PAD_POS == 1  // debug builds (scripts/kp_pos_exp.sh): 16 bytes HERE
    unsigned char kp_pos_pad[16];
#endif
    // ---- synchronous micro-batch ("split") pipeline: FWD -> [exchange] -> MID -> [head / exchange] -> UPD  (kernels.hip)
    float *split;                       // [n * split_len] per example: T[F*R], dcf[F], LR sums[split_nlr], field counts[F], label, importance
    uint32_t split_len, split_nlr;      // floats per example; LR sums kept: 1 (total) or num_combos (deep head)
    float *split_selfw;                 // [n * selfw_stride] the gather's copy of every entry's own slot (stays on the rank)
    uint32_t selfw_stride;
    float *gbuf;                        // [n] general gradient of every example (MID -> UPD)
    float *xbuf, *dxbuf;                // deep head, mini-batched: x and d logit/d x per example [n * nn.X]
    int32_t concurrent;                 // host side: a HOGWILD launch of more than one workgroup (prepare_launch); with it a whole-line updating launch of the v2 kernel does without Lds::selfw (resolve_row_mode)
    int32_t no_selfw;                   // v2 kernel, read-only launch or concurrent whole-line update: the entries' own slots (Lds::selfw) are not kept (resolve_row_mode)
    int32_t emit_x;                     // read-only launch of a model with a deep head on the v2 kernel: the example's head input x goes to xbuf, {label, importance} to gbuf,
                                        // and the layers run afterwards for the whole batch on the matrix cores (regressor.cpp run_batch, head.hip head_step)
#if defined(FW_KP_PAD_POS) && FW_KP_PAD_POS == 2  // debug builds (scripts/kp_pos_exp.sh): 16 bytes HERE
    unsigned char kp_pos_pad[16];
#endif
    uint32_t own_lo_ffm, own_hi_ffm;    // this rank owns the FFM rows with own_lo <= hash < own_hi (sharded tables)...
    uint32_t own_lo_lr, own_hi_lr;      // ... and these LR entries
    uint32_t home_lo, home_hi;          // examples of the launch whose label / counts this rank contributes (its own shard of the batch)
#if defined(FW_KP_PAD_POS) && FW_KP_PAD_POS == 3  // debug builds (scripts/kp_pos_exp.sh): 16 bytes HERE
    unsigned char kp_pos_pad[16];
#endif
    // ---- peer-sharded tables (dist.cpp fwgpu_dist_group_learn_peer): the tables are sharded by owner, every rank runs the fused hogwild
    // kernel on its own examples and reaches a row IN ITS OWNER'S MEMORY (same device, or a peer GPU's memory mapped over xGMI).
    // owner(row) = row start >> shard_shift; every owner's allocation is indexed like the whole table.
    const struct PeerShards *shards;   // device memory; NULL: the regressor's own tables
    const struct PushRings *push;      // with shards: the update phase pushes gradient rows to their owners instead of stepping the rows itself (owner-side apply)
#if defined(FW_KP_PAD_POS) && FW_KP_PAD_POS == 4  // debug builds (scripts/kp_pos_exp.sh): 16 bytes HERE
    unsigned char kp_pos_pad[16];
#endif
    // ---- row-sparse gradient mode (sparse.hip): the FWD phase lists every entry as an occurrence, slot = example * max_ffm + entry
    unsigned long long *occ_ffm_key;    // [n * max_ffm] (row hash << 32 | slot), ~0 for unused slots and examples that do not update
    uint2 *occ_ffm_desc;                // [n * max_ffm] {value bits, field}
    unsigned long long *occ_lr_key;     // [n * max_lr]
    uint2 *occ_lr_desc;                 // [n * max_lr]  {value bits, 0}
#ifdef FW_KP_TAIL_PAD                   // debug builds: the same, BEHIND the last field (no other field moves: the kernels' code stays what it is)
    unsigned char kp_tail_pad[FW_KP_TAIL_PAD];
#endif
};

struct LaunchConfig {
    uint32_t threads = 512;
    uint32_t workgroups_per_cu = 0;  // 0: as many as LDS allows (capped)
    int32_t kernel_version = 0;      // 0 = auto
    int32_t lut_global = 0;
    int32_t window = 1;              // whole-line FFM row updates: 0 off, 1 auto (tables > Infinity Cache), 2 always (debug option 2)
    int32_t no_chain = 0;            // debug option 3
    uint32_t hot_lr_every = 1;       // debug option 4: hot LR entry route (0 off, 1 atomics per example, n>1 weight deltas pending n examples)
    int32_t store_policy = -1;       // debug option 5: FFM row store policy of hogwild launches (-1: the build's default, kDefaultStorePolicy)
    int32_t kept_rows = -1;          // debug option 13: 0 = no rows kept from the gather in the large-table kernel (-1: FWGPU_KEPT_ROWS or kept)
    int32_t lr_thin = -1;            // debug option 12: store policy 4 also on hot LR entries (-1: FWGPU_LR_THIN or the build's default)
    int32_t nn_v2 = -1;              // debug option 11: deep head of concurrent two-chunk launches on the v2 kernel (-1: FWGPU_NN_V2 or on)
    float acc_hot_theta = -1.0f;     // debug option 9: policies 3 / 4, a row is hot once its accumulators have grown by this much (-1: FWGPU_ACC_HOT_THETA or 0.5)
    int32_t acc_sample_log2 = -1;    // debug option 10: policies 3 / 4, one example in 2^this touches a hot row's accumulators (-1: FWGPU_ACC_SAMPLE_LOG2 or 3)
    int32_t wb_flush_every = -1;     // debug option 6: write-back interval of policies 1 / 2 in examples per workgroup (-1: default, 0: never)
    int32_t lds_keep = -1;           // debug option 8: rows per wave kept in LDS beyond the register-kept ones (-1: as many as leave two workgroups per CU)
    int32_t prefetch = 1;            // debug option 7: next-record prefetch of the v2 kernel (A/B runs)
    bool threads_set = false;
    uint32_t max_in_flight = 0;  // cap on the persistent grid = examples processed concurrently (0: what the device holds)  // fwgpu_set_launch chose the workgroup size: no automatic choice
};

size_t example_kernel_lds_bytes(const KernelParams &p, int optimizer);
bool example_kernel_is_resident(const KernelParams &p, uint32_t threads);  // does a launch of this shape run on the v2 kernel (fw_example_kernel_r)?
void resolve_row_mode(KernelParams &p, uint32_t threads);  // settles KernelParams::window / chain for this launch shape
// Enqueue the example kernel.  grid==1 gives the sequential (in-order) semantics.
hipError_t launch_example_kernel(const KernelParams &p, int optimizer, bool coherent, uint32_t grid,
                                 uint32_t threads, hipStream_t stream);
// split pipeline: phase 1 = FWD (gather the owned rows, write the split records), 3 = UPD (updates of the owned rows from the
// records and the gradients); the MID step (records -> logit -> prediction, general gradient / deep-head input) is its own kernel
hipError_t launch_example_phase(const KernelParams &p, int optimizer, int phase, uint32_t grid, uint32_t threads, hipStream_t stream);
hipError_t launch_split_mid(const KernelParams &p, uint32_t n_examples, hipStream_t stream);
uint32_t dbg_canary_read();  // debug: LDS canary words found changed so far (phase launches under FWGPU_DBG_LDS_CANARY)
uint32_t split_record_len(uint32_t F, uint32_t R, uint32_t nlr);
// row-sparse gradient buckets (sparse.hip)
struct SparseReduceArgs {
    const unsigned long long *keys;   // [n] occurrence keys as the FWD phase wrote them
    unsigned long long *keys_sorted;  // [n]
    uint32_t n;
    int key_bits;                     // 32 + bits of the largest hash
    const uint2 *desc;
    uint32_t max_entries;             // slot = example * max_entries + entry
    uint32_t *flags, *pos;            // [n + 1]
    void *tmp;
    size_t tmp_bytes;
    // FFM rows (R != 0): gradient rows from the split records; LR (R == 0): g * value
    uint32_t R, k;
    const float *split;
    uint32_t split_len;
    const float *selfw;
    uint32_t selfw_stride;
    const float *gbuf;
    uint32_t *bk_key;                 // out: hash of every bucket row
    float *bk_rows;                   // out: [rows * max(R, 1)]
    uint32_t *d_count;                // out: number of bucket rows
};
struct SparseApplyArgs {
    const uint32_t *all_key;          // [n_ranks * stride] bucket keys of every rank
    const float *all_rows;            // [n_ranks * stride * max(R, 1)]
    const uint32_t *counts;           // [n_ranks] device: valid rows per rank
    uint32_t n_ranks, stride;
    unsigned long long *keys, *keys_sorted;  // [n_ranks * stride]
    int key_bits;
    void *tmp;
    size_t tmp_bytes;
    uint32_t R;                       // 0: LR entries ({w, acc} pairs in `w`)
    int32_t k4;                       // FFM: ffm_k % 4 == 0 (rows start on 16 B boundaries, a lane's float4 lies in one field slot)
    float *w, *acc;
    float rate, minus_power_t;
    const float *lut;
};
size_t sparse_tmp_bytes(uint32_t n_max);
hipError_t sparse_reduce(const SparseReduceArgs &a, hipStream_t stream);
hipError_t sparse_apply(const SparseApplyArgs &a, int optimizer, hipStream_t stream);
hipError_t pair_probe_ms(float *a, float *b, size_t bytes, uint32_t nrows, int reps, float *ms_out);  // sparse.hip: table placement probe
hipError_t launch_ffm_init(float *w, float *acc, uint64_t len, uint32_t k, float init_width, float init_zero_band,
                           float init_center, float acc0, hipStream_t stream);
hipError_t launch_fill(float *p, uint64_t n, float v, hipStream_t stream);
hipError_t launch_add(float *dst, const float *src, uint64_t n, hipStream_t stream);
hipError_t launch_offset_copy(uint64_t *dst, const uint64_t *src, uint32_t n, uint64_t add, hipStream_t stream);
hipError_t launch_fill_lr(float *lr, uint64_t n_entries, float w, float acc, hipStream_t stream);
hipError_t launch_checksum(const float *p, uint64_t n, unsigned long long *out, hipStream_t stream);
hipError_t launch_delta_start(const float *t, const float *s0, float *d, float *D, uint64_t n, float scale, hipStream_t stream);
hipError_t launch_delta_finish(float *t, float *s0, const float *d, const float *D, uint64_t n, hipStream_t stream);
hipError_t launch_coherence_probe(unsigned *scratch, int use_sc1, unsigned iters, unsigned blocks, hipStream_t stream);

}  // namespace fwgpu

struct fwgpu_block_cache;
namespace fwgpu { struct HostBatch; }
struct fwgpu_batch {
    fwgpu_regressor *owner = nullptr;
    const fwgpu_block_cache *cache = nullptr;  // context cache the next predict-only launch of this batch starts from
    float *emit_T = nullptr, *emit_dcf = nullptr;  // setup_cache launch: where example 0's field sums go
    uint32_t n = 0;
    uint64_t n_lr = 0, n_ffm = 0;
    uint32_t max_lr = 0, max_ffm = 0;
    bool aligned4 = true;
    // one device allocation, carved
    void *dev = nullptr;
    size_t dev_bytes = 0;
    uint32_t *ffm_hash = nullptr;
    float *ffm_val = nullptr;
    uint8_t *ffm_fld = nullptr;
    uint32_t *ffm_off = nullptr;
    uint32_t *lr_hash = nullptr;
    float *lr_val = nullptr;
    uint32_t *lr_off = nullptr;
    uint16_t *lr_combo = nullptr;
    float *label = nullptr;
    float *importance = nullptr;
    float *pred = nullptr;
    uint32_t *work = nullptr;  // device counter the workgroups of a launch take their examples from
    // raw-record batches (device-side translation)
    uint32_t *records = nullptr;
    uint64_t *rec_off = nullptr;
    uint32_t max_rec = 0;
    bool rec_self_len = false;
    bool delta_records = false;  // serving: records hold the candidate's namespaces only, the rest come from the context cache's record
    uint64_t n_words = 0;
    uint64_t words_cap = 0;
    uint32_t n_cap = 0;
    // single-request batches (serving.cpp): records / offsets / predictions live in host memory mapped into the device's
    // address space (the kernel reads the request and writes the prediction over PCIe: no copy calls), and every launch takes
    // the next counter of a zeroed ring instead of a memset of `work`
    void *host_block = nullptr;
    ptrdiff_t host_delta = 0;  // host address of a mapped array = its device address + host_delta
    uint32_t *h_records = nullptr;
    uint64_t *h_rec_off = nullptr;
    float *h_pred = nullptr;
    uint32_t *work_ring = nullptr;
    uint32_t work_next = 0;
    void *tr_dev = nullptr;  // device blob holding the DevTranslator arrays
    fwgpu::DevTranslator tr{};
    std::unique_ptr<fwgpu::HostBatch> host_copy;  // entry batches with an example of more than 4096 FFM features / LR entries: walked example by example (regressor.cpp)
};

struct fwgpu_regressor {
    fwgpu_config cfg{};
    int device = 0;
    int num_cus = 256;
    size_t lds_per_cu = 160 * 1024;
    uint64_t lr_len = 0;   // entries
    uint64_t ffm_len = 0;  // floats
    float *d_lr = nullptr, *d_ffm_w = nullptr, *d_ffm_acc = nullptr;
    int placement_tries = 0;                     // candidate allocations timed for d_ffm_acc (regressor.cpp: place_ffm_acc)
    float placement_ms_lo = 0, placement_ms_hi = 0;  // fastest / slowest pair probe among them
    float placement_ms_single = 0;                    // the same probe on the weight table alone
    bool placement_contended = false;                 // the search ran and every candidate contended with the weight table
    float *d_lut_lr = nullptr, *d_lut_ffm = nullptr;
    uint32_t lr_hash_mask = 0, ffm_hash_mask = 0;
    fwgpu::LaunchConfig launch;
    unsigned long long *d_ticks = nullptr;  // debug phase timing, see fwgpu_debug_phase_ticks
    // deep head
    fwgpu::DevNN nn{};
    fwgpu_nn_config nn_cfg{};
    float *d_nn_w = nullptr, *d_nn_acc = nullptr, *d_lut_nn = nullptr;
    uint64_t nn_len = 0;
    void *head_scratch = nullptr;  // mini-batched head (head.hip): activations, masks, gradients of the current batch
    float *pred_x = nullptr, *pred_yi = nullptr;  // predict-only batches of a model with a deep head: x [n * X] and {label, importance} [2n] of the batch in flight
    uint32_t pred_cap = 0;
    // scratch for single-example calls
    fwgpu_batch *one = nullptr;
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
};

// device buffers of one synchronous micro-batch (regressor.cpp "split pipeline")
struct fwgpu_split {
    fwgpu_regressor *owner = nullptr;
    uint32_t n_cap = 0, split_len = 0, nlr = 1, selfw_stride = 0;
    float *d_split = nullptr, *d_selfw = nullptr, *d_g = nullptr, *d_x = nullptr, *d_dx = nullptr;
};

// Vec<BlockCache> of the reference (regressor.rs:40-50): what setup_cache leaves behind for predict_with_cache
struct fwgpu_block_cache {
    fwgpu_regressor *owner = nullptr;
    float *d_T = nullptr;    // [F*R] + [F] (dcf) in one allocation
    float *d_dcf = nullptr;
    std::vector<uint64_t> present;  // sorted (hash << 32 | contra_field_index) of the cached FFM features (features_present)
    // record batches (fwgpu_block_cache_cover_record): the context's record decides which namespace slots are covered
    uint32_t *d_cover = nullptr;           // device copy of `cover`, followed by the context's record
    uint32_t *d_ctx_rec = nullptr;         // (inside d_cover's allocation)
    std::vector<uint32_t> ctx_rec;         // the context's record
    size_t ctx_rec_cap = 0;                // words of the allocation behind d_ctx_rec
    std::vector<uint32_t> cover;           // bit per namespace slot
    std::vector<uint32_t> ctx_slots;       // the context record's slot words (a request that rewrites a covered slot is not covered)
    std::vector<uint64_t> present_bits;    // 4096-bit filter in front of `present`
};

namespace fwgpu {
// Host worker threads kept between calls (serving requests, trainer chunks): starting a few dozen std::threads costs as much as
// the work a call gives them.  run(n, fn) calls fn(0) on the caller and fn(1 .. n-1) on the workers; one run at a time.
class Workers {
public:
    ~Workers() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    template <class F>
    void run(unsigned n, F &&fn) {
        if (n <= 1) {
            fn(0u);
            return;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            while (threads_.size() + 1 < n) {
                const unsigned id = (unsigned)threads_.size() + 1;
                threads_.emplace_back([this, id] { loop(id); });
            }
            job_ = [&fn](unsigned k) { fn(k); };
            active_ = n;
            remaining_ = n - 1;
            generation_++;
        }
        cv_.notify_all();
        fn(0u);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return remaining_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(unsigned id) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)> job;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || (generation_ != seen && id < active_); });
                if (stop_) return;
                seen = generation_;
                job = job_;
            }
            job(id);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--remaining_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::function<void(unsigned)> job_;
    uint64_t generation_ = 0;
    unsigned active_ = 0, remaining_ = 0;
    bool stop_ = false;
};


}  // namespace fwgpu

namespace fwgpu {
// host helpers implemented in regressor.cpp / translate.cpp
struct HostBatch {  // SoA staging of a CSR batch on the host
    std::vector<uint32_t> ffm_hash, ffm_off, lr_hash, lr_off;
    std::vector<float> ffm_val, lr_val, label, importance;
    std::vector<uint8_t> ffm_fld;
    std::vector<uint16_t> lr_combo;
    uint32_t max_lr = 0, max_ffm = 0;
    bool aligned4 = true;
    void clear();
    uint32_t size() const { return (uint32_t)label.size(); }
};
int batch_alloc(fwgpu_regressor *r, uint32_t n, uint64_t n_lr, uint64_t n_ffm, fwgpu_batch **out, bool host_mapped = false);
int batch_upload(fwgpu_batch *b, const HostBatch &hb, hipStream_t stream);
void keep_host_copy_if_oversize(fwgpu_batch *b, HostBatch &&hb);
int record_batch_host_copy_if_oversize(fwgpu_regressor *r, const fwgpu_translator_config *t, fwgpu_batch *b, const uint32_t *records,
                                       const uint64_t *rec_off, uint32_t n);
int append_example(const fwgpu_regressor *r, HostBatch &hb, const fwgpu_lr_entry *lr, uint32_t n_lr,
                   const fwgpu_ffm_entry *ffm, uint32_t n_ffm, float label, float importance);
int translate_record(const fwgpu_translator_config *t, const uint32_t *rec, uint32_t rec_len,
                     std::vector<fwgpu_lr_entry> &lr, std::vector<fwgpu_ffm_entry> &ffm, float *label,
                     float *importance);
int check_translator(const fwgpu_regressor *r, const fwgpu_translator_config *t);
// validates one record against the translator and counts the entries its translation produces
int count_record(const fwgpu_translator_config *t, const uint32_t *rec, uint32_t rec_len, uint32_t *n_lr, uint32_t *n_ffm);
int block_cache_record_ok(const fwgpu_block_cache *c, const fwgpu_translator_config *t, const uint32_t *record, uint32_t len, bool delta);
// the same for a candidate-only record on top of its context's record (ctx may be NULL)
int count_record(const fwgpu_translator_config *t, const uint32_t *rec, uint32_t rec_len, const uint32_t *ctx, uint32_t ctx_len,
                 uint32_t *n_lr, uint32_t *n_ffm);
// device-resident raw-record batch (translation happens inside the example kernel)
int record_batch_alloc(fwgpu_regressor *r, const fwgpu_translator_config *t, uint32_t n_cap, uint64_t words_cap,
                       fwgpu_batch **out, bool host_mapped = false);
// next launch of a host-mapped batch: points b->work at a fresh zero counter (re-zeroing the ring when it wraps)
int mapped_batch_next_counter(fwgpu_batch *b, hipStream_t stream);
constexpr uint32_t kWorkRing = 1024;
struct RecordStats {  // what the host learns about a slice of records without translating them
    uint32_t max_lr = 0, max_ffm = 0, max_rec = 0;
    uint64_t tot_lr = 0, tot_ffm = 0;
    void merge(const RecordStats &o) {
        max_lr = max_lr > o.max_lr ? max_lr : o.max_lr;
        max_ffm = max_ffm > o.max_ffm ? max_ffm : o.max_ffm;
        max_rec = max_rec > o.max_rec ? max_rec : o.max_rec;
        tot_lr += o.tot_lr;
        tot_ffm += o.tot_ffm;
    }
};
int count_records(const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                  RecordStats *st);
// stats == NULL: validate and count here
int record_batch_upload(fwgpu_batch *b, const fwgpu_translator_config *t, const uint32_t *records,
                        const uint64_t *rec_off, uint32_t n, hipStream_t stream, const RecordStats *stats = nullptr);
uint32_t lr_hash_mask(uint32_t bit_precision);
uint32_t ffm_hash_mask(uint32_t ffm_bits, uint32_t ffm_k);
void lut_init(float *lut, float learning_rate, float power_t, float init_acc);
KernelParams make_params(const fwgpu_regressor *r, const fwgpu_batch *b, int update);
// the fused learn / predict launch of `b` with every row and LR entry reached in its owner's tables (generic kernel); d_shards: device copy
int run_batch_peer(fwgpu_regressor *r, fwgpu_batch *b, int mode, int update, const PeerShards *d_shards, hipStream_t stream,
                   const PushRings *d_push = nullptr, uint32_t stream_consumers = 0, uint32_t device_share = 1, uint32_t stream_max_consumer_waves = 0, uint32_t n_ranks = 0);
// owner-side apply: the optimizer steps of `n_rows` pushed gradient rows / `n_lr` pushed LR gradients on this regressor's own tables
hipError_t launch_rendezvous(uint32_t *counter, uint32_t n, uint32_t *met, hipStream_t stream);
hipError_t launch_owner_apply(const fwgpu_regressor *r, float *lr_base, const uint32_t *keys, const float *rows, uint32_t n_rows, const uint2 *lr_ent,
                              uint32_t n_lr, bool in_order, hipStream_t stream);
struct SplitRanges {  // what a rank owns (sharded tables) and which examples of the launch are its own
    uint32_t ffm_lo = 0, ffm_hi = 0xffffffffu, lr_lo = 0, lr_hi = 0xffffffffu, home_lo = 0, home_hi = 0xffffffffu;
};
struct OccBuffers {  // where the FWD phase lists the entries of every example (row-sparse gradient mode)
    unsigned long long *ffm_key = nullptr, *lr_key = nullptr;
    uint2 *ffm_desc = nullptr, *lr_desc = nullptr;
};
int split_forward(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, const SplitRanges &rg, hipStream_t stream,
                  const OccBuffers *occ = nullptr, uint32_t *max_ffm = nullptr, uint32_t *max_lr = nullptr);
int split_mid(fwgpu_regressor *r, fwgpu_split *sp, uint32_t first, uint32_t n, float *d_pred, bool head, hipStream_t stream);
int split_update(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, const SplitRanges &rg, bool head, hipStream_t stream);
// mini-batched deep head on the records' x (head.hip): forward, sigmoid, backward, one AdaGrad step per dense weight
int head_step(fwgpu_regressor *r, fwgpu_split *sp, uint32_t first, uint32_t n, float *d_pred, bool update, hipStream_t stream);
void head_scratch_free(fwgpu_regressor *r);
uint32_t pick_grid(const fwgpu_regressor *r, const KernelParams &p, int mode, uint32_t threads);
}  // namespace fwgpu
