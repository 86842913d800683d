// HIP kernels (gfx950 / CDNA4) for the LR+FFM learn/predict path.
//
// One workgroup owns one example at a time (persistent grid, stride gridDim.x).  Per example:
//   stage   : the example's FeatureBuffer (feature_buffer.rs:24-31) is copied HBM -> LDS;
//   gather  : each wave takes whole fields; for every feature of the field it loads the complete weight
//             row w[hash .. hash+R) (R = F*k floats, one coalesced 16 B/lane buffer load, block_ffm.rs:163-217)
//             and accumulates the field sum S[f][z][k] in registers, then writes it TRANSPOSED into LDS:
//             T[z][f][k] = S[f][z][k], so that T[f][.] is exactly the "contra" row a feature of field f
//             multiplies with (block_ffm.rs:219-261);
//   dot     : sum_{f,z,k} S[f][z][k]*S[z][f][k] = sum_e T[e]*T[perm(e)] from LDS, minus the self-pair
//             correction sum_i v_i^2 |w_i[f_i]|^2 (block_ffm.rs:316-440, 1107-1201), wave64 shuffle + LDS
//             reduction; LR forward = random 8-byte gathers (block_lr.rs:28-47);
//   sigmoid : block_loss_functions.rs:105-153 (clip +-50, NaN -> 0.5, general gradient);
//   update  : per feature row, read-modify-write of w and acc with the optimizer step fused
//             (block_ffm.rs:265-288, optimizer.rs); LR entries likewise (block_lr.rs:135-150).
//
// Semantics that are kept exactly as in the reference inside ONE example:
//   * duplicate LR hashes apply sequentially in buffer order (regressor.rs:629-655 pins this);
//   * FFM rows of different features may overlap (rows start at hash&mask but are R long,
//     block_ffm.rs:92-94); overlapping features are applied in buffer order, each seeing the previous
//     one's accumulator, and all gradients are computed from the pre-update weights.
// Across examples: grid==1 walks the batch in order (sequential = the reference's single thread); a larger
// grid runs examples concurrently with unsynchronised read-modify-write (hogwild.rs semantics).
//
// Coherence: the per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another
// CU's stores.  All table accesses of an updating launch therefore use device-scope (sc1) loads and stores
// (buffer_* ... sc1 / agent-scope relaxed atomics for the 8-byte LR entries).  Read-only launches use
// plain cached loads.
#include "fwgpu_internal.h"
#include "fwgpu_device.h"
#include <cstdlib>
#include <cstddef>
#include <algorithm>

// Store policy of the v2 kernel's FFM row traffic in HOGWILD launches (template argument POL of fw_example_kernel_r, chosen per launch from
// KernelParams::store_policy; fwgpu_debug_set_option(r, 5, policy) / FWGPU_STORE_POLICY select it at run time, tests/test_gpu_conservation.py
// measures what each policy does to the steps of rows that many concurrent examples hold):
//   0  both tables device-scope write-through (buffer_store ... sc1): every 64 B request goes to the memory side and is acknowledged from there;
//   1  WEIGHT rows write-back through the XCD's L2, accumulators write-through (round 3's shipped build: +9.5 % examples/s);
//   2  BOTH tables write-back (the fastest: 0.555-0.56 of the HBM peak, profiles/r04b_policy_ab.txt).
//   3  (round 5's shipped policy) policy 1 with THINNED accumulator stores on hot rows.  What skew costs under policy 1 is the write-through of the accumulator
//      lines that many concurrent examples hold (profiles/r05_skew_x_store_policy.txt: uniform ids 0.639 of the peak under policies 1 and 2 alike; Zipf 1.3:
//      0.547 against 0.659; L2 tag stalls 6.4x, profiles/r05_skew_pmc_counters.txt).  A kept row whose accumulators exceed KernelParams::acc_hot_theta stores
//      its accumulator row for one example in 2^acc_sample_log2 only (a hash of the example's ticket and the row's slot decides), with 2^acc_sample_log2 times
//      the example's g^2: the expectation of what reaches memory is what write-through sends, one coherent copy (policy 2's trouble is eight private ones),
//      an m-th of the requests on exactly the lines that queue.  The STEP of every example still uses acc_read + its own g^2.
//   4  (shipped since round 6) policy 3 with the thinned store replaced by a thinned ATOMIC ADD: the one example in m = 2^acc_sample_log2 whose turn it is adds m x its g^2 to the
//      accumulator row with fire-and-forget device-scope float atomics (buffer_atomic_add_f32 ... sc1, four per lane), everybody else issues nothing.  A thinned
//      STORE writes acc_read + m g^2: when it loses its race against another example's store, m examples' worth of g^2 are gone (tests/test_gpu_conservation.py:
//      hot rows kept 0.15-0.29 of their true sum under policies 0, 1 and 3 alike).  An add cannot lose: what reaches memory is an unbiased estimate of the TRUE sum
//      of g^2 over all concurrent examples -- what the reference's hogwild threads count (hogwild.rs:89-103: `acc += g*g` on coherent memory, optimizer.rs:147-149)
//      -- at an m-th of the requests.  It applies to every row of a wave (kept in registers, parked in LDS, re-read), since there is no race to lose.
// A write-back line is visible to the other seven XCDs when it leaves this XCD's L2.  Cold lines leave within microseconds (an XCD's 4 MB L2 turns
// over every ~16 us at this kernel's write rate); a line that is re-touched before it is evicted -- the head rows of a Zipf field -- would stay
// dirty for the whole launch, each XCD stepping a private copy.  KernelParams::wb_flush_every bounds that window: every that many examples a
// workgroup issues ONE `buffer_wbl2 sc1` (write back all dirty lines of this XCD's L2; lines stay valid), staggered over the workgroups, so
// that an XCD's L2 is written back every few microseconds whatever the rows' popularity.  All loads stay device-scope (sc1: L1 bypassed, served by
// the L2 or, for lines another XCD has written through, by the memory side).
// In-order launches (one workgroup = one XCD) are exact under every policy; the launch's end writes everything back.
#ifndef FW_DEFAULT_STORE_POLICY
#define FW_DEFAULT_STORE_POLICY 4
#endif
#ifndef FW_DEFAULT_WB_FLUSH_EVERY
#define FW_DEFAULT_WB_FLUSH_EVERY 128
#endif
namespace fwgpu {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// aux (cache policy) bits of the raw buffer builtins on gfx940+: bit0 sc0, bit1 nt, bit4 sc1.
constexpr int kAuxPlain = 0;
constexpr int kAuxSc1 = 16;
constexpr int kAuxSys = 17;  // sc0 sc1: system scope -- rows in a PEER GPU's memory (peer-sharded tables over xGMI)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    // raw buffer (stride 0), num_records in bytes; out-of-range lanes load 0 and their stores are dropped.
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    typedef f4 type;
    template <int AUX>
    static __device__ __forceinline__ f4 load(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
        u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX);
        return __builtin_bit_cast(f4, v);
    }
    template <int AUX>
    static __device__ __forceinline__ void store(f4 v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)byte_off, 0, AUX);
    }
    static __device__ __forceinline__ f4 lds_load(const float *p) { return *reinterpret_cast<const f4 *>(p); }
    static __device__ __forceinline__ void lds_store(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
    static __device__ __forceinline__ f4 zero() { return f4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ float get(const f4 &v, int i) { return v[i]; }
    static __device__ __forceinline__ void set(f4 &v, int i, float x) { v[i] = x; }
};
template <>
struct Vec<1> {
    typedef float type;
    template <int AUX>
    static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
        unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, AUX);
        return __uint_as_float(v);
    }
    template <int AUX>
    static __device__ __forceinline__ void store(float v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, AUX);
    }
    static __device__ __forceinline__ float lds_load(const float *p) { return *p; }
    static __device__ __forceinline__ void lds_store(float *p, float v) { *p = v; }
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ float get(const float &v, int) { return v; }
    static __device__ __forceinline__ void set(float &v, int, float x) { v = x; }
};

__device__ __forceinline__ float logistic(float t) { return 1.0f / (1.0f + expf(-t)); }  // block_loss_functions.rs:15-17

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <bool COH, bool SYS = false>
__device__ __forceinline__ float2 lr_load(const float *lr, uint32_t h) {
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(lr) + h;
    unsigned long long v;
    if (COH && SYS)
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (COH)
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        v = *p;
    return float2{__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32))};
}
template <bool COH, bool SYS = false>
__device__ __forceinline__ void lr_store(float *lr, uint32_t h, float2 wa) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(lr) + h;
    unsigned long long v = (unsigned long long)__float_as_uint(wa.x) | ((unsigned long long)__float_as_uint(wa.y) << 32);
    if (COH && SYS)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (COH)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// Table bases of a row / an LR entry: the regressor's own tables, or -- peer-sharded tables -- the owner's (KernelParams::shards).
// `h` is wave-uniform at the row sites (a scalar branch and a scalar load from the kernel arguments).
// (The owners' bases live in device memory, not in the kernel arguments: an array indexed at run time inside the by-value KernelParams
// made the compiler copy the whole struct to scratch -- 1-2 KB per lane in every kernel, -30 % on the generic kernel.)
// SH is a template argument of the generic kernel (its own instantiations, launched by run_batch_peer only): a run-time test per row
// cost the deep-head launch 17 %.
template <bool SH>
__device__ __forceinline__ float *ffm_w_base(const KernelParams &p, uint32_t h) { return SH ? p.shards->ffm_w[h >> p.shards->shift_ffm] : p.ffm_w; }
template <bool SH>
__device__ __forceinline__ float *ffm_acc_base(const KernelParams &p, uint32_t h) { return SH ? p.shards->ffm_acc[h >> p.shards->shift_ffm] : p.ffm_acc; }
template <bool SH>
__device__ __forceinline__ float *lr_base(const KernelParams &p, uint32_t h) { return SH ? p.shards->lr[h >> p.shards->shift_lr] : p.lr; }

// LDS carve-up (all offsets 16-byte aligned).
struct Lds {
    float *T;        // F*R
    float *selfw;    // max_ffm*k : w_i[f_i*k ..] as read in the gather phase (pre-update)
    float *lut;      // 2048 (AdagradLUT only)
    uint32_t *e_hash;  // max_ffm
    float *e_val;
    uint32_t *e_fld;  // field (low 8 bits) | kRowHasChain | kRowChained | kRowDep
    uint32_t *l_hash;  // max_lr
    float *l_val;
    uint32_t *fstart, *fend;  // F
    float *red;               // 3*16
    float *dcf;               // F: per-field self-pair correction
    uint32_t *set_ffm;        // open-addressing set: FFM row block keys (overlap pre-filter), or -- chains -- first entry index of every row hash
    uint32_t *set_lr;         // open-addressing set: first entry index of every LR hash
    uint32_t *l_flag;         // per LR entry: kRowChained / kRowHasChain (duplicate LR hashes, block_lr.rs:135-150 order)
    uint32_t *set_blk;        // open-addressing set of FFM row block keys of first occurrences (chains: set_ffm then holds hashes)
    uint32_t *rec;            // raw record staged for device-side translation (max_rec words)
    uint32_t *tcnt;           // per (field,namespace) pair and per combo: entry count, then exclusive offset
    uint32_t *l_combo;        // combo slot of each LR entry (deep head only)
    float *nn;                // deep-head scratch: x[X], xg[X], h[sum_width], m[sum_width], l_prod[max_lr]
    uint32_t *ctr;            // 8 counters, then 3 floats: the hot LR entry's acc snapshot, pending weight delta, pending acc delta; [16..] see kCtr*
    float *keep;              // v2 kernel: [wave][lds_keep][R] gather-time w of the rows a wave keeps in LDS (beyond its register-kept ones)
    uint32_t *rec_next;       // v2 kernel, record batches: the NEXT example's record, copied from HBM while this example is in its dot / update phases
};
// ctr[] slots of the v2 kernel's prefetch and write-back bookkeeping
constexpr int kCtrNext = 16;      // the next example's ticket, published to all threads by the post-gather barrier
constexpr int kCtrPfLen = 17;     // words of the next example's record that sit in rec_next (0: not prefetched)
constexpr int kCtrWbCount = 18;   // examples of this workgroup since its last buffer_wbl2
constexpr int kCtrWbEvery = 19;   // KernelParams::wb_flush_every, or 0 when this launch never writes back (in-order launches, policy 0)
constexpr int kCtrLabel = 20;     // v2 kernel: the example's label and importance (float bits), left here by the stage phase for the loss: two vector registers less through the gather
constexpr int kCtrImp = 21;

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// LDS floats the deep head needs: x and its gradient, activations and masks of every hidden layer, the final neuron's
// input gradient, and one product per LR entry
__host__ __device__ inline uint32_t nn_lds_floats(const KernelParams &p) {
    if (!p.nn.n_layers) return 0;
    if (p.emit_x) return p.max_lr + 16;  // (the v2 kernel only forms the head's input: one product per LR entry)
    // (fg takes a layer's input gradient only for the first layer -- X floats; the later layers' land in place -- and the LR products of the forward pass
    // are dead long before the unwinding writes it: one region for both)
    return 2 * p.nn.X + 2 * p.nn.sum_width + (p.nn.X > p.max_lr ? p.nn.X : p.max_lr) + p.nn.max_out + 1 + 16;
}

// size of the open-addressing sets: power of two >= 2*n (load factor <= 0.5)
__host__ __device__ inline uint32_t set_size(uint32_t n) {
    uint32_t s = 16;
    while (s < 2 * n) s <<= 1;
    return s;
}
__host__ __device__ inline uint32_t log2u(uint32_t pow2) {
    uint32_t l = 0;
    while ((1u << l) < pow2) l++;
    return l;
}
constexpr uint32_t kSetEmpty = 0xffffffffu;
// e_fld flag bits of an FFM entry (set in the stage phase of updating launches)
constexpr uint32_t kRowDep = 0x80000000u;       // overlaps an earlier row of another hash: applied in buffer order, after all others
constexpr uint32_t kRowChained = 0x40000000u;   // same hash as an earlier row: applied by that row's owner, from registers
constexpr uint32_t kRowHasChain = 0x20000000u;  // later rows of the same hash are chained to this one
constexpr uint32_t kFldMask = 0xffu;
// returns true if `key` was already present
__device__ __forceinline__ bool set_insert(uint32_t *tab, uint32_t mask, uint32_t shift, uint32_t key) {
    uint32_t slot = (key * 2654435761u) >> shift;
    for (;;) {
        const uint32_t old = atomicCAS(&tab[slot], kSetEmpty, key);
        if (old == kSetEmpty) return false;
        if (old == key) return true;
        slot = (slot + 1) & mask;
    }
}
// Index-valued variant: the table holds entry INDICES, the key of a slot is keys[tab[slot]].  Inserting entry i under
// keys[i] leaves the SMALLEST index of every key in its slot, i.e. the first occurrence in buffer order.
__device__ __forceinline__ void first_insert(uint32_t *tab, uint32_t mask, uint32_t shift, const uint32_t *keys, uint32_t i) {
    const uint32_t key = keys[i];
    uint32_t slot = (key * 2654435761u) >> shift;
    for (;;) {
        const uint32_t old = atomicCAS(&tab[slot], kSetEmpty, i);
        if (old == kSetEmpty) return;
        if (keys[old] == key) {  // (the slot may change meanwhile, but only to another index of the same key)
            atomicMin(&tab[slot], i);
            return;
        }
        slot = (slot + 1) & mask;
    }
}
// first occurrence of a key that is known to be present
__device__ __forceinline__ uint32_t first_find(const uint32_t *tab, uint32_t mask, uint32_t shift, const uint32_t *keys, uint32_t key) {
    uint32_t slot = (key * 2654435761u) >> shift;
    for (;;) {
        const uint32_t v = tab[slot];
        if (keys[v] == key) return v;
        slot = (slot + 1) & mask;
    }
}
__device__ __forceinline__ bool set_contains(const uint32_t *tab, uint32_t mask, uint32_t shift, uint32_t key) {
    uint32_t slot = (key * 2654435761u) >> shift;
    for (;;) {
        const uint32_t v = tab[slot];
        if (v == kSetEmpty) return false;
        if (v == key) return true;
        slot = (slot + 1) & mask;
    }
}

// The hot LR entry of a hogwild launch (the constant feature's: it is in every example, feature_buffer.rs:270-276).  Every
// workgroup's read-modify-write of that one 8-byte entry goes to the memory side and they serialise there (measured: 60-80 ns
// per write, which capped a 10-field model at 16 M examples/s), and of the ~10 that overlap at any moment only the last write
// survives.  So in hogwild launches this one entry is stepped with fire-and-forget atomics instead of load + store:
//   * the step is taken at (the accumulator this example's FORWARD pass read) + g^2 -- the forward pass loads the entry's
//     {w, acc} pair anyway, so the accumulator is as fresh as the weight the prediction was made with: stale by at most the
//     examples in flight, which is hogwild's own staleness (hogwild.rs:89-103: a thread reads, steps, writes back);
//     `acc += g^2` then goes to the table as an atomic add, so every example's g^2 arrives whatever the interleaving.
//     (Round 2 stepped on a per-WORKGROUP snapshot of the accumulator refreshed every 32 of the workgroup's examples: with short
//     launches every workgroup took all its steps at the launch-start accumulator, i.e. at the largest step size, and the entry
//     overshot -- GPUTEST_r02.  The step SIZE must follow the global accumulator; only the weight delta tolerates batching.)
//   * the weight delta is added with an atomic as well; `hot_lr_every` (ctr[13]) > 1 keeps a workgroup's deltas pending in LDS
//     for that many of its examples first (its own forward passes see them), 1 sends every step at once.
// Nothing is lost: the entry's accumulator ends a launch at acc0 + the sum of all examples' g^2, which a test checks.
// In-order launches (one workgroup, the bit-exact mode) and the phases of the synchronous pipeline do not use it.
// (The switch and the entry's hash sit in LDS next to the state, ctr[13] / ctr[12], so that nothing of this stays live in
// scalar registers across the example loop: the v2 kernel has none to spare.  State: hot[0] = the accumulator the forward pass
// read, hot[1] = pending weight delta.)
__device__ __forceinline__ float *hot_lr_state(const Lds &s) { return reinterpret_cast<float *>(s.ctr + 8); }
__device__ __forceinline__ bool hot_lr_is(const Lds &s, uint32_t h) { return s.ctr[13] != 0 && h == s.ctr[12]; }
template <bool COH>
__device__ __forceinline__ void hot_lr_init(const KernelParams &p, const Lds &s, bool fused) {  // thread 0, before the example loop
    const bool on = fused && COH && p.hot_lr_every != 0 && p.has_lr && p.update && p.grid_wgs > 1;
    s.ctr[13] = on ? p.hot_lr_every : 0;
    s.ctr[12] = p.hot_lr_hash;
    s.ctr[7] = 0;
    float *hot = hot_lr_state(s);
    hot[0] = hot[1] = hot[2] = 0.0f;
}
// The pending weight delta goes to the table (taken out of LDS by exchange: a step another thread adds meanwhile stays pending).
// Called by the thread that has just stepped the entry, every ctr[13] examples, and by thread 0 after the example loop.
template <bool SH = false>
__device__ __forceinline__ void hot_lr_flush(const KernelParams &p, const Lds &s) {
    float *hot = hot_lr_state(s);
    const float dw = atomicExch(hot + 1, 0.0f);
    s.ctr[7] = 0;
    if (dw == 0.0f) return;
    __hip_atomic_fetch_add(lr_base<SH>(p, s.ctr[12]) + 2 * (size_t)s.ctr[12], dw, __ATOMIC_RELAXED, SH ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
}
// The weight of LR entry `h` as the forward pass sees it (block_lr.rs:36-45); for the hot entry: + this workgroup's pending
// delta, and the accumulator that came with it is kept for the update phase.
template <bool COH, bool SH = false>
__device__ __forceinline__ float lr_forward_weight(const KernelParams &p, const Lds &s, uint32_t h) {
    const float2 wa = lr_load<COH, SH>(lr_base<SH>(p, h), h);
    if (COH && hot_lr_is(s, h)) {
        float *hot = hot_lr_state(s);
        hot[0] = wa.y;
        return wa.x + hot[1];
    }
    return wa.x;
}

// the same, returning the whole {w, acc} pair: the v2 kernel keeps a thread's first entry from the forward pass for its update (lr_update `kept`)
template <bool COH, bool SH = false>
__device__ __forceinline__ float2 lr_forward_pair(const KernelParams &p, const Lds &s, uint32_t h) {
    float2 wa = lr_load<COH, SH>(lr_base<SH>(p, h), h);
    if (COH && hot_lr_is(s, h)) {
        float *hot = hot_lr_state(s);
        hot[0] = wa.y;
        wa.x += hot[1];
    }
    return wa;
}

__host__ __device__ inline size_t lds_layout(uint32_t F, uint32_t k, uint32_t max_ffm, uint32_t max_lr, uint32_t n_luts,
                                             uint32_t max_rec, uint32_t tr_items, uint32_t nn_floats, bool chain, size_t *off /*[25]*/,
                                             uint32_t pf_words = 0, uint32_t tr_words = 0, uint32_t keep_words = 0, bool t_in_lds = true, bool selfw_on = true) {
    size_t o = 0;
    size_t R = (size_t)F * k;
    const size_t t_bytes = t_in_lds ? 4 * F * R : 0;  // (phase kernels keep T in the split record: KernelParams::t_global)
    off[0] = o; o = align16(o + t_bytes);
    off[1] = o; o = align16(o + (selfw_on ? 4 * (size_t)max_ffm * k : 0));
    off[2] = o; o = align16(o + 4 * (size_t)kLutSize * n_luts);
    off[3] = o; o = align16(o + 4 * (size_t)max_ffm);
    off[4] = o; o = align16(o + 4 * (size_t)max_ffm);
    off[5] = o; o = align16(o + 4 * (size_t)max_ffm);
    off[6] = o; o = align16(o + 4 * (size_t)max_lr);
    off[7] = o; o = align16(o + 4 * (size_t)max_lr);
    off[8] = o; o = align16(o + 4 * (size_t)F);
    off[9] = o; o = align16(o + 4 * (size_t)F);
    off[10] = o; o = align16(o + 4 * 3 * 16);
    off[11] = o; o = align16(o + 4 * 32);  // ctr[8] + hot LR entry state (hot_lr_*) + the v2 kernel's prefetch / write-back slots (kCtr*)
    off[12] = o; o = align16(o + 4 * (size_t)F);
    // The record copy and the two hash sets are only alive during the stage phase, T only from the end of the stage phase on:
    // when they fit they live INSIDE T's region (config C: 6.4 KB of 28.8 KB), which is what lets a third workgroup fit a CU.
    {
        const size_t a13 = 0, a14 = align16(a13 + 4 * (size_t)set_size(max_ffm)), a15 = align16(a14 + 4 * (size_t)set_size(max_lr)),
                     a20 = align16(a15 + 4 * (size_t)max_rec), aend = align16(a20 + (chain ? 4 * (size_t)set_size(max_ffm) : 0));
        if (aend <= t_bytes) {
            off[13] = a13;
            off[14] = a14;
            off[15] = a15;
            off[20] = a20;
        } else {
            off[13] = o; o = align16(o + 4 * (size_t)set_size(max_ffm));
            off[14] = o; o = align16(o + 4 * (size_t)set_size(max_lr));
            off[15] = o; o = align16(o + 4 * (size_t)max_rec);
            off[20] = o; o = align16(o + (chain ? 4 * (size_t)set_size(max_ffm) : 0));
        }
    }
    off[16] = o; o = align16(o + 4 * (size_t)tr_items);
    off[21] = o; o = align16(o + 4 * (size_t)max_lr);
    off[17] = o; o = align16(o + (nn_floats ? 4 * (size_t)max_lr : 0));
    off[18] = o; o = align16(o + 4 * (size_t)nn_floats);
    off[22] = o; o = align16(o + 4 * (size_t)pf_words);  // rec_next (outside T's region: it is written while T is alive)
    off[23] = o; o = align16(o + 4 * (size_t)tr_words);  // v2 kernel: the translator's tables (TrLds)
    off[24] = o; o = align16(o + 4 * (size_t)keep_words);  // v2 kernel: gather-time w of the rows kept in LDS (Lds::keep)
    return o;
}

__device__ __forceinline__ bool k_nonzero(uint32_t k) { return k != 0; }

// ------------------------------------------------------------------ stage phase (shared by both example kernels)
struct SetGeom {
    uint32_t setf_n, setl_n, setf_shift, setl_shift, blk_shift;
};
// 2^blk_shift >= the longest span two rows can conflict over: rows that conflict then have block keys
// (hash >> blk_shift) differing by <= 1.  Plain rows conflict when they share a float (span R); whole-line updates
// (KernelParams::window) conflict when they share a 128 B line (span <= R + 31 floats rounded up to whole lines).
__device__ __forceinline__ uint32_t conflict_blk_shift(const KernelParams &p) {
    const uint32_t span = p.window ? ((p.R + 31u + 31u) & ~31u) : p.R;
    uint32_t sh = 0;
    while ((1u << sh) < span) sh++;
    return sh;
}
struct StageOut {
    uint32_t nf, nl;
    float label, imp;
    bool do_update;
};

// record slot decoding, feature_reader! (feature_buffer.rs:47-108) on the LDS copy of the record
// `ctx` (or NULL): the serving context's record, which a candidate-only record inherits every namespace from that it does
// not hold itself (fwgpu_parser_parse_after_prefix, delta form)
__device__ __forceinline__ uint32_t slot_count(const uint32_t *rec, const uint32_t *ctx, uint32_t ns) {
    uint32_t w = rec[3 + ns];
    if (ctx && w == 0x80000000u) w = ctx[3 + ns];
    if (!(w & 0x80000000u)) return 1;                      // single feature, value 1.0 (parser.rs:62-66)
    return ((w & 0xffffu) - ((w >> 16) & 0x3fffu)) >> 1;   // NO_FEATURES = 0x80000000 -> 0
}
__device__ __forceinline__ void slot_get(const uint32_t *rec, const uint32_t *ctx, uint32_t ns, bool is_f32, uint32_t q, uint32_t &hash,
                                         float &val) {
    uint32_t w = rec[3 + ns];
    if (ctx && w == 0x80000000u) {
        rec = ctx;
        w = rec[3 + ns];
    }
    if (!(w & 0x80000000u)) {
        hash = w;
        val = 1.0f;
    } else {
        const uint32_t st = (w >> 16) & 0x3fffu;
        hash = rec[st + 2 * q];
        val = is_f32 ? 1.0f : __uint_as_float(rec[st + 2 * q + 1]);  // feature_buffer.rs:88-104
    }
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
        // (`lane >= d` as arithmetic -- all ones when d - 1 - lane is negative: as six compares the lane masks of a scan are twelve scalar registers,
        // computed ahead of the scan, in the stage phase where the example kernels' scalar pressure peaks)
        // (through an asm statement: written in C++ the compiler turns it back into the compare)
        int m;
        asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(d - 1 - lane));
        v += t & (uint32_t)m;
    }
    return v;
}

// Brings example `ex` into LDS: either copies its pre-translated entries, or translates its raw record
// (FeatureBufferTranslator::translate, feature_buffer.rs:178-338, bit-exact hashing) on the spot; then finds the field
// boundaries and runs the O(1) pre-filters for overlapping FFM rows / duplicate LR hashes (exact scan only if they hit).
// debug: shader-clock stamps inside the stage phase (slots 8.. of fwgpu_debug_phase_ticks)
struct StageTicker {
    unsigned long long *out;
    unsigned long long last;
    __device__ __forceinline__ void stamp(int slot) {
        if (out) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            atomicAdd(out + slot, now - last);
            last = now;
        }
    }
};

// The translator's tables (which namespaces feed which LR combo / FFM field) as the stage phase reads them: straight from HBM (TrGlobal), or
// from a packed copy in LDS that the v2 kernel's prologue makes (TrLds).  Through TrGlobal every lookup is a global load followed by an
// s_waitcnt vmcnt(0) -- half a dozen dependent round trips per stage phase, and each of those waits also waits for the acknowledgement of
// every row store the previous example's update phase has in flight; through TrLds the stage phase of a prefetched record touches no memory.
struct TrGlobal {
    const DevTranslator &t;
    __device__ __forceinline__ uint32_t pair_ns(uint32_t j) const { return t.pair_ns[j]; }
    __device__ __forceinline__ bool pair_f32(uint32_t j) const { return t.pair_f32[j] != 0; }
    __device__ __forceinline__ uint32_t pair_field(uint32_t j) const { return t.pair_field[j]; }
    __device__ __forceinline__ uint32_t combo_off(uint32_t c) const { return t.combo_off[c]; }
    __device__ __forceinline__ uint32_t combo_ns(uint32_t m) const { return t.combo_ns[m]; }
    __device__ __forceinline__ bool combo_f32(uint32_t m) const { return t.combo_f32[m] != 0; }
    __device__ __forceinline__ float combo_w(uint32_t c) const { return t.combo_w[c]; }
};
struct TrLds {
    const uint32_t *pair;  // [n_pairs]      namespace | field << 16 | f32 << 31
    const uint32_t *coff;  // [n_combos + 1] first member of every combo
    const uint32_t *cm;    // [n_members]    namespace | f32 << 31
    const float *cw;       // [n_combos]     combo weight
    __device__ __forceinline__ uint32_t pair_ns(uint32_t j) const { return pair[j] & 0xffffu; }
    __device__ __forceinline__ bool pair_f32(uint32_t j) const { return (pair[j] >> 31) != 0; }
    __device__ __forceinline__ uint32_t pair_field(uint32_t j) const { return (pair[j] >> 16) & 0xffu; }
    __device__ __forceinline__ uint32_t combo_off(uint32_t c) const { return coff[c]; }
    __device__ __forceinline__ uint32_t combo_ns(uint32_t m) const { return cm[m] & 0x7fffffffu; }
    __device__ __forceinline__ bool combo_f32(uint32_t m) const { return (cm[m] >> 31) != 0; }
    __device__ __forceinline__ float combo_w(uint32_t c) const { return cw[c]; }
};
__host__ __device__ inline uint32_t tr_lds_words(const DevTranslator &t) { return t.n_pairs + (t.n_combos + 1) + t.n_members + t.n_combos; }

// Must be called by every thread of the workgroup (it contains barriers).
template <bool CTX = true, class TR = TrGlobal>  // CTX: the launch may carry a serving context cache (read-only launches only)
__device__ __forceinline__ StageOut stage_example(const KernelParams &p, const Lds &s, const SetGeom &g, uint32_t ex,
                                                  int tid, int bd, const TR &tr, unsigned long long *tick_out = nullptr, uint32_t pf_len = 0) {
    StageTicker tk{tick_out, tick_out ? __builtin_amdgcn_s_memtime() : 0ull};
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t F = p.F, R = p.R;
    StageOut o;
    const uint32_t *grec = nullptr;
    uint32_t rec_len = 0, fo = 0, lo = 0;
    if (p.records && pf_len) {
        // (v2 kernel) the record is in LDS already: copied there during the previous example's dot phase, so that this stage phase starts
        // without a memory round trip -- and without waiting for the previous example's row stores to be acknowledged, which the first
        // vmcnt wait of a stage phase that loads from HBM does
        rec_len = pf_len;
        o.label = (float)s.rec_next[1];
        o.imp = __uint_as_float(s.rec_next[2]);
        o.nf = o.nl = 0;
    } else if (p.records) {
        const uint64_t r0 = p.rec_off[ex];
        grec = p.records + r0;
        // gathered batches (sharded multi-GPU step) have gaps between the ranks' records: the record carries its own
        // length in word 0 (parser.rs:57-60)
        rec_len = p.rec_self_len ? grec[0] : (uint32_t)(p.rec_off[ex + 1] - r0);
        o.label = (float)grec[1];              // feature_buffer.rs:187
        o.imp = __uint_as_float(grec[2]);      // feature_buffer.rs:188-189
        o.nf = o.nl = 0;
    } else {
        fo = p.ffm_off[ex];
        o.nf = p.ffm_off[ex + 1] - fo;
        lo = p.lr_off[ex];
        o.nl = p.lr_off[ex + 1] - lo;
        o.label = p.label[ex];
        o.imp = p.importance[ex];
    }
    o.do_update = p.update && (o.imp != 0.0f);  // regressor.rs:366
    const bool do_update = o.do_update;

    for (uint32_t i = tid; i < F; i += bd) {
        s.fstart[i] = 0;
        s.fend[i] = 0;
    }
    if (tid == 0) {
        // (the zero is made opaque: left alone the compiler keeps a four-register zero vector alive across the whole example loop for this one
        // ds_write_b128, spills it in the 128-register kernel and reloads it from scratch HERE -- behind an s_waitcnt vmcnt(0) that drains
        // wave 0's row stores at the top of every stage phase)
        uint32_t zero = 0;
        asm volatile("" : "+v"(zero));
        s.ctr[0] = zero;  // next field to gather (v1)
        s.ctr[1] = zero;  // some FFM rows of this example overlap an earlier row (exact)
        s.ctr[2] = zero;  // pre-filter: rows MAY overlap
        s.ctr[3] = zero;  // duplicate LR hashes in this example
        s.ctr[14] = zero;  // deep head: LR entries not grouped by combo slot (nn_forward)
    }
    if (do_update) {
        for (uint32_t i = tid; i < g.setf_n; i += bd) {
            s.set_ffm[i] = kSetEmpty;
            if (p.chain) s.set_blk[i] = kSetEmpty;
        }
        for (uint32_t i = tid; i < g.setl_n; i += bd) s.set_lr[i] = kSetEmpty;
        for (uint32_t i = tid; i < p.max_lr; i += bd) s.l_flag[i] = 0;
    }
    if (!p.records) {
        for (uint32_t i = tid; i < o.nf; i += bd) {
            s.e_hash[i] = p.ffm_hash[fo + i];
            s.e_val[i] = p.ffm_val[fo + i];
            s.e_fld[i] = p.ffm_fld[fo + i];
        }
        for (uint32_t i = tid; i < o.nl; i += bd) {
            s.l_hash[i] = p.lr_hash[lo + i];
            s.l_val[i] = p.lr_val[lo + i];
            if (p.nn.n_layers) s.l_combo[i] = p.lr_combo[lo + i];
        }
        __syncthreads();
    } else {
        const DevTranslator &t = p.tr;
        const uint32_t NP = t.n_pairs, NC = t.n_combos;
        if (pf_len)
            for (uint32_t i = tid; i < rec_len; i += bd) s.rec[i] = s.rec_next[i];
        else
            for (uint32_t i = tid; i < rec_len; i += bd) s.rec[i] = grec[i];
        const uint32_t *ctx_rec = nullptr;
        if (CTX && p.ctx_rec) {  // candidate-only records: the context's record sits behind the candidate's in LDS
            ctx_rec = s.rec + rec_len;
            for (uint32_t i = tid; i < p.ctx_rec_len; i += bd) s.rec[rec_len + i] = p.ctx_rec[i];
        }
        __syncthreads();
        tk.stamp(8);
        // Two steps.  (1) wave 0 counts the features of every (field, namespace) pair, wave 1 the entries of every LR
        // combo, each with a wave64 prefix scan -> start offsets in LDS.  (2) after a barrier EVERY thread emits one
        // output entry: it finds its pair / combo by binary search in the offsets and decodes its own feature(s), so the
        // emit is one step deep instead of "features per namespace" steps on two waves.
        uint32_t *ffm_base = s.tcnt;            // [NP + 1]
        uint32_t *lr_base = s.tcnt + NP + 1;    // [NC + 1]
        if (wave == 0) {
            uint32_t carry = 0;
            for (uint32_t b0 = 0; b0 < NP; b0 += 64) {
                const uint32_t j = b0 + lane;
                bool on = j < NP && k_nonzero(p.k);
                if (CTX && p.ctx_cover && on) {  // features the context cache holds are not gathered again (block_ffm.rs:548, 600)
                    const uint32_t ns = tr.pair_ns(j);
                    on = !((p.ctx_cover[ns >> 5] >> (ns & 31)) & 1u);
                }
                const uint32_t cnt = on ? slot_count(s.rec, ctx_rec, tr.pair_ns(j)) : 0;
                const uint32_t inc = wave_scan_incl(cnt, lane);
                if (j < NP) ffm_base[j] = carry + inc - cnt;
                carry += (uint32_t)__shfl((int)inc, 63, 64);
            }
            if (lane == 0) {
                ffm_base[NP] = carry;
                s.ctr[4] = carry;
            }
        }
        if (wave == (bd > 64 ? 1 : 0)) {
            uint32_t carry = 0;
            for (uint32_t b0 = 0; b0 < NC; b0 += 64) {
                const uint32_t c = b0 + lane;
                const bool on = c < NC && p.has_lr;
                const uint32_t m0 = on ? tr.combo_off(c) : 0, m1 = on ? tr.combo_off(c + 1) : 0;
                uint32_t total = on ? 1 : 0;
                for (uint32_t m = m0; m < m1; ++m) total *= slot_count(s.rec, ctx_rec, tr.combo_ns(m));
                const uint32_t inc = wave_scan_incl(total, lane);
                if (c < NC) lr_base[c] = carry + inc - total;
                carry += (uint32_t)__shfl((int)inc, 63, 64);
            }
            if (lane == 0) {
                lr_base[NC] = carry;
                const bool add_const = p.has_lr && t.add_const;
                if (add_const) {  // feature_buffer.rs:270-276
                    s.l_hash[carry] = 11650396u & t.lr_mask;
                    s.l_val[carry] = 1.0f;
                    if (p.nn.n_layers) s.l_combo[carry] = NC;
                }
                s.ctr[5] = carry + (add_const ? 1u : 0u);
            }
        }
        __syncthreads();
        // ffm_buffer, ordered by field (feature_buffer.rs:314-335): entry e belongs to the last pair whose start <= e
        const uint32_t nf_all = s.ctr[4];
        for (uint32_t e = tid; e < nf_all; e += bd) {
            uint32_t lo_ = 0, hi_ = NP;  // invariant: ffm_base[lo_] <= e < ffm_base[hi_]
            while (hi_ - lo_ > 1) {
                const uint32_t mid = (lo_ + hi_) >> 1;
                if (ffm_base[mid] <= e) lo_ = mid;
                else hi_ = mid;
            }
            const uint32_t j = lo_;
            uint32_t h;
            float v;
            slot_get(s.rec, ctx_rec, tr.pair_ns(j), tr.pair_f32(j), e - ffm_base[j], h, v);
            s.e_hash[e] = h & t.ffm_mask;
            s.e_val[e] = v;
            s.e_fld[e] = tr.pair_field(j);
        }
        // lr_buffer (feature_buffer.rs:194-276): hash = (h_prev * 16777619) ^ h_next, values multiply.  Threads are
        // taken from the top of the workgroup so that the waves busy with the ffm_buffer above are not the same ones.
        const uint32_t nl_combo = lr_base[NC];
        for (uint32_t e = (uint32_t)(bd - 1 - tid); e < nl_combo; e += bd) {
            uint32_t lo_ = 0, hi_ = NC;
            while (hi_ - lo_ > 1) {
                const uint32_t mid = (lo_ + hi_) >> 1;
                if (lr_base[mid] <= e) lo_ = mid;
                else hi_ = mid;
            }
            const uint32_t c = lo_;
            const uint32_t m0 = tr.combo_off(c), m1 = tr.combo_off(c + 1);
            // digits of the entry's index within the combo, first namespace most significant (the reference's loop nesting)
            uint32_t rem = e - lr_base[c], div = lr_base[c + 1] - lr_base[c];
            uint32_t hash = 0;
            float val = 1.0f;
            for (uint32_t m = m0; m < m1; ++m) {
                const uint32_t cm = slot_count(s.rec, ctx_rec, tr.combo_ns(m));
                div /= cm;
                const uint32_t q = rem / div;
                rem -= q * div;
                uint32_t h;
                float v;
                slot_get(s.rec, ctx_rec, tr.combo_ns(m), tr.combo_f32(m), q, h, v);
                if (m == m0) {
                    hash = h;
                    val = v;
                } else {
                    hash = (hash * 16777619u) ^ h;  // feature_buffer.rs:242-251 (wrapping)
                    val = val * v;
                }
            }
            s.l_hash[e] = hash & t.lr_mask;
            s.l_val[e] = val * tr.combo_w(c);
            if (p.nn.n_layers) s.l_combo[e] = c;
        }
        __syncthreads();
        tk.stamp(9);
        o.nf = s.ctr[4];
        o.nl = s.ctr[5];
    }
    const uint32_t nf = o.nf, nl = o.nl;
    // field boundaries; O(1) pre-filters: FFM rows that may overlap (block keys equal or adjacent), duplicate LR hashes
    for (uint32_t i = tid; i < nf; i += bd) {
        const uint32_t f = s.e_fld[i];
        if (i == 0 || s.e_fld[i - 1] != f) s.fstart[f] = i;
        if (i == nf - 1 || s.e_fld[i + 1] != f) s.fend[f] = i + 1;
        if (do_update) {
            if (p.chain) {  // set_ffm: first entry of every row hash
                first_insert(s.set_ffm, g.setf_n - 1, g.setf_shift, s.e_hash, i);
            } else if (set_insert(s.set_ffm, g.setf_n - 1, g.setf_shift, s.e_hash[i] >> g.blk_shift)) {
                s.ctr[2] = 1;
            }
        }
    }
    if (do_update && p.has_lr)
        for (uint32_t i = tid; i < nl; i += bd)
            first_insert(s.set_lr, g.setl_n - 1, g.setl_shift, s.l_hash, i);
    __syncthreads();
    tk.stamp(10);
    if (do_update && p.has_lr)  // duplicate LR hashes: later occurrences are chained to the first (lr_update)
        for (uint32_t i = tid; i < nl; i += bd) {
            const uint32_t own = first_find(s.set_lr, g.setl_n - 1, g.setl_shift, s.l_hash, s.l_hash[i]);
            if (own != i) {
                atomicOr(&s.l_flag[i], kRowChained);
                atomicOr(&s.l_flag[own], kRowHasChain);
            }
        }
    if (do_update && p.chain) {
        // Rows of the SAME hash (a feature drawn twice, or two features colliding: 98 % of config C's examples have some)
        // are chained to the first one: its owner applies them in buffer order from registers (update_rows_win).  Only
        // first occurrences enter the overlap pre-filter, so duplicates alone never trigger the exact scan.
        for (uint32_t i = tid; i < nf; i += bd) {
            const uint32_t h = s.e_hash[i];
            const uint32_t own = first_find(s.set_ffm, g.setf_n - 1, g.setf_shift, s.e_hash, h);
            if (own != i) {
                atomicOr(&s.e_fld[i], kRowChained);
                atomicOr(&s.e_fld[own], kRowHasChain);
            } else if (set_insert(s.set_blk, g.setf_n - 1, g.setf_shift, h >> g.blk_shift)) {
                s.ctr[2] = 1;
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < nf; i += bd)
            if (!(s.e_fld[i] & kRowChained) && set_contains(s.set_blk, g.setf_n - 1, g.setf_shift, (s.e_hash[i] >> g.blk_shift) + 1))
                s.ctr[2] = 1;
    } else if (do_update) {
        for (uint32_t i = tid; i < nf; i += bd)
            if (set_contains(s.set_ffm, g.setf_n - 1, g.setf_shift, (s.e_hash[i] >> g.blk_shift) + 1)) s.ctr[2] = 1;
    }
    __syncthreads();
    tk.stamp(11);
    // (the hash sets and the record copy, which may share T's LDS region, are dead from here on)
    // T columns and self-pair corrections of empty fields are zero (block_ffm.rs:168-180).  No barrier is needed before
    // the gather: it writes the columns of the non-empty fields only, and a barrier follows it.
    if (p.k) {
        const uint32_t per = R / (p.k % 4 == 0 ? 4 : 1), vecw = p.k % 4 == 0 ? 4 : 1;
        // (UPD phase with T in the split record: the record IS the example's T -- for a chunk of an oversize example, the whole example's -- and is left alone)
        if (!(p.t_global && p.update))
        for (uint32_t idx = tid; idx < F * per; idx += bd) {
            const uint32_t f = idx / per, q = idx - f * per;
            if (s.fstart[f] == s.fend[f]) {  // (with a context cache: the cached features' sums stand alone)
                const uint32_t ee = q * vecw, zz = ee / p.k;
                for (uint32_t j = 0; j < vecw; ++j) {
                    const uint32_t ix = zz * R + f * p.k + (ee - zz * p.k) + j;
                    s.T[ix] = (CTX && p.ctx_T) ? p.ctx_T[ix] : 0.0f;
                }
            }
        }
        for (uint32_t f = tid; f < F; f += bd)
            if (s.fstart[f] == s.fend[f]) s.dcf[f] = (CTX && p.ctx_dcf) ? p.ctx_dcf[f] : 0.0f;
    }
    if (do_update && s.ctr[2]) {
        // Rare: some rows may overlap.  Exact scan: does an EARLIER feature's row [h_j, h_j+R) overlap mine?
        // (rows are R long but start on a next_pow2(k) grid: block_ffm.rs:92-94, feature_buffer.rs:141-148)
        uint32_t my_dep[4] = {0, 0, 0, 0};  // supports nf <= 4*bd (checked on the host)
        int slot = 0;
        for (uint32_t i = tid; i < nf; i += bd, ++slot) {
            const uint32_t h = s.e_hash[i];
            uint32_t d = 0;
            for (uint32_t j = 0; j < i; ++j) {
                const uint32_t hj = s.e_hash[j];
                if (p.chain && hj == h) continue;  // same row: chained, not a dependency
                if (p.window) {  // whole-line updates: do the 128 B line spans [h & ~31, round_up(h + R, 32)) intersect?
                    const uint32_t a0 = h & ~31u, a1 = (h + R + 31u) & ~31u, b0 = hj & ~31u, b1 = (hj + R + 31u) & ~31u;
                    d |= (a0 < b1 && b0 < a1) ? 1u : 0u;
                } else {
                    const uint32_t diff = h > hj ? h - hj : hj - h;
                    d |= (diff < R) ? 1u : 0u;
                }
            }
            my_dep[slot & 3] = d;
        }
        __syncthreads();
        slot = 0;
        uint32_t any = 0;
        for (uint32_t i = tid; i < nf; i += bd, ++slot) {
            if (my_dep[slot & 3]) {
                // (a row that overlaps an earlier row of another hash leaves its chain: every later row of its hash overlaps
                // that row too, so the whole tail of the chain is applied in order in phase B)
                s.e_fld[i] = (s.e_fld[i] & ~kRowChained) | kRowDep;
                any = 1;
            }
        }
        if (any) s.ctr[1] = 1;
        __syncthreads();
    }
    return o;
}

#ifndef FW_PHASE_TU  // (host functions live in ONE of the two translation units of this file: see the Makefile)
size_t example_kernel_lds_bytes(const KernelParams &p, int optimizer) {
    size_t off[25];
    // (lut_lds_forced: the v2 kernel's single-chunk instantiations ALWAYS keep the AdaGrad LUT in LDS -- kLdsLut -- whatever option 1 says)
    return lds_layout(p.F, p.k, p.max_ffm, p.max_lr, (optimizer == FWGPU_OPT_ADAGRAD_LUT && p.update && (!p.lut_global || p.lut_lds_forced)) ? 1 : 0,
                      p.records ? p.max_rec : 0, p.records ? p.tr.n_pairs + p.tr.n_combos + 2 : 0, nn_lds_floats(p), p.chain != 0, off,
                      (p.records && p.prefetch) ? p.max_rec : 0, (p.records && p.tr_lds) ? tr_lds_words(p.tr) : 0, p.lds_keep_words, !p.t_global, !p.no_selfw);
}
#endif


// ------------------------------------------------------------------ deep head (a18), per-example reference semantics
struct NnBuf {
    float *x;    // [X]  inputs: LR slots then triangle (the Join span); BlockCopy output 1 (kept as values)
    float *xg;   // [X]  d logit / d x : through the layers + (topology "one") the final neuron's direct part
    float *h;    // [sum_width] post-activation outputs of the hidden layers
    float *m;    // [sum_width] ReLU 0/1 masks, then the layers' output gradients
    float *fg;   // [max(X, max_lr)] the final neuron's output gradient, then the first layer's input gradient (nn_backward) ...
    float *prod; // ... and, before that, w*v of every LR entry (nn_forward): the same region
    uint32_t *act;  // [max_out + 1] indices of a layer's neurons with a nonzero output gradient, then their count (nn_layer_backward_vec)
};
__device__ __forceinline__ NnBuf nn_buf(const KernelParams &p, const Lds &s) {
    NnBuf b;
    b.x = s.nn;
    b.xg = b.x + p.nn.X;
    b.h = b.xg + p.nn.X;
    b.m = b.h + p.nn.sum_width;
    b.fg = b.m + p.nn.sum_width;
    b.prod = b.fg;
    b.act = reinterpret_cast<uint32_t *>(b.fg + (p.nn.X > p.max_lr ? p.nn.X : p.max_lr));
    return b;
}
template <bool COH>
__device__ __forceinline__ float nn_ld(const float *p) {
    if (COH) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
template <bool COH>
__device__ __forceinline__ void nn_st(float *p, float v) {
    if (COH)
        __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
__device__ __forceinline__ uint32_t tri_index(uint32_t a, uint32_t b) {  // row-major lower triangle, block_misc.rs:871-882
    const uint32_t i = a > b ? a : b, j = a > b ? b : a;
    return i * (i + 1) / 2 + j;
}

// x = [per-combo LR sums, triangle of the FFM pair outputs]; then the layers; returns the logit (same value in all threads).
// block_lr.rs:36-45, block_misc.rs:864-883, block_neural.rs:196-222, block_relu.rs:38-54, regressor.rs:307-319
template <int VEC, bool COH>
__device__ __forceinline__ float nn_forward(const KernelParams &p, const Lds &s, uint32_t nl, int tid, int bd) {
    const int lane = tid & 63, wave = tid >> 6, nw = bd >> 6;
    const NnBuf b = nn_buf(p, s);
    const DevNN &n = p.nn;
    const uint32_t F = p.F, k = p.k, R = p.R, C = p.num_combos;
#ifdef FW_TICKS  // sub-phase stamps of the head (slots 8..14 of fwgpu_debug_phase_ticks; -DFW_TICKS builds only)
    unsigned long long nn_last = __builtin_amdgcn_s_memtime();
#define FW_NN_TICK(slot)                                                   \
    if (tid == 0 && p.ticks) {                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
        atomicAdd(p.ticks + (slot), now_ - nn_last);                       \
        nn_last = now_;                                                    \
    }
#else
#define FW_NN_TICK(slot)
#endif
    for (uint32_t i = tid; i < nl; i += bd) {
        b.prod[i] = lr_forward_weight<COH>(p, s, s.l_hash[i]) * s.l_val[i];
        if (i + 1 < nl && s.l_combo[i] > s.l_combo[i + 1]) s.ctr[14] = 1;  // (zeroed by the stage phase)
    }
    __syncthreads();
    // Each combo slot sums its entries in buffer order.  The translator emits the entries combo by combo (feature_buffer.rs:194-267),
    // so slot c's entries are one run, found by bisection; a batch of entries in any other order takes the full scan (the scan
    // by 31 threads over 200 entries each was 30 us of an example's 210).
    const bool by_combo = s.ctr[14] == 0;
    for (uint32_t c = tid; c < C; c += bd) {
        float acc = 0.0f;
        if (by_combo) {
            uint32_t lo_ = 0, hi_ = nl;  // first entry with l_combo >= c
            while (lo_ < hi_) {
                const uint32_t mid = (lo_ + hi_) >> 1;
                if (s.l_combo[mid] < c) lo_ = mid + 1;
                else hi_ = mid;
            }
            for (uint32_t i = lo_; i < nl && s.l_combo[i] == c; ++i) acc += b.prod[i];
        } else {
            for (uint32_t i = 0; i < nl; ++i)
                if (s.l_combo[i] == c) acc += b.prod[i];
        }
        b.x[c] = acc;
    }
    const uint32_t T = F * (F + 1) / 2;
    for (uint32_t t = tid; t < T; t += bd) {
        uint32_t i = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        const uint32_t j = t - i * (i + 1) / 2;
        float dot = 0.0f;
        if (i != j) {
            for (uint32_t kk = 0; kk < k; ++kk) dot += s.T[i * R + j * k + kk] * s.T[j * R + i * k + kk];
        } else if (p.no_selfw) {
            // (launches that keep no LDS copy of the entries' own slots -- the v2 kernel's concurrent two-chunk launches: the diagonal as split_mid_kernel and
            // the batched predict-only head form it: exactly 0 for a field of at most one feature, 0.5 (|field sum|^2 - sum of the features' own squares) otherwise)
            for (uint32_t kk = 0; kk < k; ++kk) dot += s.T[i * R + i * k + kk] * s.T[i * R + i * k + kk];
            dot = (s.fend[i] - s.fstart[i]) <= 1u ? 0.0f : 0.5f * (dot - s.dcf[i]);
        } else {
            // Diagonal in the reference's own form (block_ffm.rs:231-243): per feature, sum_k w * v * (contra - w * v).
            // A field with one feature gives EXACTLY 0 this way; "0.5 * (dot - dcf)" leaves a 1e-10 residue there and the
            // first AdaGrad step of a dense weight (nn_init_acc_gradient = 0) blows any nonzero input up to O(lr).
            for (uint32_t e = s.fstart[i]; e < s.fend[i]; ++e) {
                const float v = s.e_val[e];
                float corr = 0.0f;
                for (uint32_t kk = 0; kk < k; ++kk) {
                    const float w = s.selfw[e * k + kk];
                    corr += w * (v * (s.T[i * R + i * k + kk] - w * v));
                }
                dot += corr * 0.5f;
            }
        }
        b.x[C + t] = dot;  // triangle (block_misc.rs:864-883): 2 * (0.5 * dot) off the diagonal
    }
    __syncthreads();
    FW_NN_TICK(8);
    const float *in_vec = b.x;
    uint32_t hoff = 0;
    for (uint32_t l = 0; l < n.n_layers; ++l) {
        const uint32_t in = n.in[l], out = n.out[l];
        const float *W = n.w + n.off[l];
        // a wave computes JU neurons per pass so that JU * ceil(in / 64) weight loads are in flight per lane
#ifndef FW_NN_FJU
#define FW_NN_FJU 8
#endif
        constexpr int JU = FW_NN_FJU;
        // hogwild launches: the same dot products from 16-byte device-scope loads (4-byte ones run at a third of the rate);
        // the order of the sum differs, which only the in-order mode promises
        const bool vec16 = COH && p.grid_wgs > 1 && (in & 3u) == 0 && (n.off[l] & 3u) == 0 && ((uintptr_t)in_vec & 15u) == 0;
        for (uint32_t j0 = wave * JU; j0 < out; j0 += nw * JU) {
            float dot[JU], bias[JU];
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                dot[u] = 0.0f;
                // (issued in front of the weights: loaded behind the reduction, the bias was a second round trip per pass)
                bias[u] = j0 + u < out ? nn_ld<COH>(W + (size_t)in * out + j0 + u) : 0.0f;
            }
            if (vec16) {
                const __amdgpu_buffer_rsrc_t rw = make_rsrc(W, in * out * 4);
                for (uint32_t q = lane; q < (in >> 2); q += 64) {
                    f4 w[JU];
#pragma unroll
                    for (int u = 0; u < JU; ++u)
                        w[u] = j0 + u < out ? Vec<4>::load<kAuxSc1>(rw, ((j0 + u) * in + 4 * q) * 4) : Vec<4>::zero();
                    const f4 xv = Vec<4>::lds_load(in_vec + 4 * q);
#pragma unroll
                    for (int u = 0; u < JU; ++u) dot[u] += w[u][0] * xv[0] + w[u][1] * xv[1] + w[u][2] * xv[2] + w[u][3] * xv[3];
                }
            } else
            for (uint32_t i = lane; i < in; i += 64) {
                float w[JU];
#pragma unroll
                for (int u = 0; u < JU; ++u) w[u] = j0 + u < out ? nn_ld<COH>(W + (size_t)(j0 + u) * in + i) : 0.0f;
                const float xi = in_vec[i];
#pragma unroll
                for (int u = 0; u < JU; ++u) dot[u] += w[u] * xi;
            }
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const uint32_t j = j0 + u;
                const float d = wave_sum(dot[u]);
                if (j < out) {
                    const float pre = bias[u] + d;
                    if (n.relu[l]) {
                        b.h[hoff + j] = pre < 0.0f ? 0.0f : pre;
                        b.m[hoff + j] = pre < 0.0f ? 0.0f : 1.0f;
                    } else {
                        b.h[hoff + j] = pre;
                        b.m[hoff + j] = 1.0f;
                    }
                }
            }
        }
        __syncthreads();
        FW_NN_TICK(9 + (l ? 1 : 0));
        in_vec = b.h + hoff;
        hoff += out;
    }
    // final neuron over [h_last, x] (topology "one") or h_last (topology "two")
    const uint32_t L = n.n_layers, fin = n.in[L], wl = n.out[L - 1];
    const float *Wf = n.w + n.off[L];
    float dot = 0.0f;
    for (uint32_t i = tid; i < fin; i += bd) dot += nn_ld<COH>(Wf + i) * (i < wl ? in_vec[i] : b.x[i - wl]);
    dot = wave_sum(dot);
    if (lane == 0) s.red[wave] = dot;
    __syncthreads();
    float z = 0.0f;
    for (int w = 0; w < nw; ++w) z += s.red[w];
    z = nn_ld<COH>(Wf + fin) + z;
    __syncthreads();
    FW_NN_TICK(11);
    return z;
}

// One BlockNeuronLayer backward (block_neural.rs:252-340): neuron by neuron (j), every input i: AdaGrad step on
// W[j][i] with gradient og[j]*in[i]; in_grad[i] += W_old[j][i]*og[j].  Thread i owns column i, so the j order of the
// reference is kept per weight and no two threads touch the same weight.  in_vals and in_grad may alias.
template <int OPT, bool COH>
__device__ __forceinline__ void nn_layer_backward(const DevNN &n, uint32_t l, const float *og, const float *in_a,
                                                  uint32_t split, const float *in_b, float *grad_a, float *grad_b,
                                                  int tid, int bd) {
    const uint32_t in = n.in[l], out = n.out[l];
    float *W = n.w + n.off[l], *A = n.acc + n.off[l];
    for (uint32_t i = tid; i < in; i += bd) {
        const float xi = i < split ? in_a[i] : in_b[i - split];
        float oe = 0.0f;
#ifndef FW_NN_BJU
#define FW_NN_BJU 16
#endif
        constexpr int JU = FW_NN_BJU;  // weights (and accumulators) of JU neurons in flight per thread
        for (uint32_t j0 = 0; j0 < out; j0 += JU) {
            float w[JU], a[JU], gg[JU];
            // (a neuron's output gradient is the same LDS word for every thread: "is it zero" is decided on its bits in a scalar register -- +-0 shifted left
            // is 0, a NaN is not, as in the float compare -- and kept as one bit of one scalar register: JU neurons in flight are then scalar branches on ONE
            // register, not JU 64-bit lane masks)
            uint32_t live = 0;  // bit u: neuron j0 + u takes a step
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const uint32_t j = j0 + u;
                gg[u] = j < out ? og[j] : 0.0f;
                const uint32_t gbits = __builtin_amdgcn_readfirstlane(__float_as_uint(gg[u])) << 1;
                w[u] = 0.0f;
                a[u] = 0.0f;
                if (gbits != 0u) {  // block_neural.rs:275-277
                    live |= 1u << u;
                    const size_t ix = (size_t)j * in + i;
                    w[u] = nn_ld<COH>(W + ix);
                    if (OPT != FWGPU_OPT_SGD) a[u] = nn_ld<COH>(A + ix);
                }
            }
            uint32_t j0s = j0;
            asm volatile("" : "+s"(j0s), "+s"(live));  // (tested afresh -- the first loop's test results, kept for here, are lane masks again -- and the neuron indices formed afresh instead of living in JU scalars from the first loop on)
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                if (!((live >> u) & 1u)) continue;
                const size_t ix = (size_t)(j0s + u) * in + i;
                const float upd = opt_step<OPT>(gg[u] * xi, a[u], n.rate, n.minus_power_t, n.lut);
                oe += w[u] * gg[u];
                nn_st<COH>(W + ix, w[u] - upd);
                if (OPT != FWGPU_OPT_SGD) nn_st<COH>(A + ix, a[u]);
            }
        }
        if (i < split)
            grad_a[i] = oe;
        else
            grad_b[i - split] = oe;
    }
    for (uint32_t j = tid; j < out; j += bd) {  // bias terms (block_neural.rs:296-306)
        const float gg = og[j];
        if (gg != 0.0f) {
            const size_t ix = (size_t)in * out + j;
            const float w = nn_ld<COH>(W + ix);
            float acc = OPT == FWGPU_OPT_SGD ? 0.0f : nn_ld<COH>(A + ix);
            const float upd = opt_step<OPT>(gg, acc, n.rate, n.minus_power_t, n.lut);
            nn_st<COH>(W + ix, w - upd);
            if (OPT != FWGPU_OPT_SGD) nn_st<COH>(A + ix, acc);
        }
    }
}

// The same layer backward for HOGWILD launches, where no order is promised: 16-byte device-scope accesses instead of 4-byte
// ones (a scalar sc1 store is a fabric write of its own, about 6x the time per byte of a 16-byte one: at config E the head
// issued 388 k of them per example, against 13 k for the whole FFM update), and every thread of the workgroup busy: thread
// (q, grp) owns input columns 4q .. 4q+3 for the neurons of group grp; the groups' shares of the input gradient meet in LDS.
// Needs 16-byte aligned rows (in % 4 == 0, layer offset % 4 == 0) and in / 4 <= workgroup size; otherwise the caller keeps
// nn_layer_backward.  Contains barriers: called by every thread.
#ifndef FW_NN_VJU
#define FW_NN_VJU 4
#endif
template <int OPT>
__device__ __forceinline__ void nn_layer_backward_vec(const DevNN &n, uint32_t l, const float *og, const float *in_a,
                                                      uint32_t split, const float *in_b, float *grad_a, float *grad_b,
                                                      uint32_t *act, int tid, int bd) {
    const uint32_t in = n.in[l], out = n.out[l], nq = in >> 2;
    float *W = n.w + n.off[l], *A = n.acc + n.off[l];
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(W, in * out * 4), ra = make_rsrc(A, in * out * 4);
    // The neurons that take a step: those with a nonzero output gradient (block_neural.rs:275-277) -- behind a ReLU about half of a layer.  Their indices
    // are compacted into LDS first, so that the groups below share the ACTIVE neurons evenly and a pass never holds slots of dead ones (round 4 walked
    // all `out` neurons in passes of FW_NN_VJU and skipped the dead ones inside a pass: every pass still paid its memory round trip).
    if (tid < 64) {
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < out; base += 64) {
            const uint32_t j = base + (uint32_t)tid;
            const bool on = j < out && og[j] != 0.0f;
            const unsigned long long m = __ballot(on);
            if (on) act[cnt + (uint32_t)__popcll(m & ((1ull << tid) - 1ull))] = j;
            cnt += (uint32_t)__popcll(m);
        }
        if (tid == 0) act[out] = cnt;
    }
    // bias terms (block_neural.rs:296-306): their loads are issued here, in front of the weight passes, and stepped behind them
    const bool has_bias = (uint32_t)tid < out && og[(uint32_t)tid < out ? tid : 0] != 0.0f;
    float bw = 0.0f, bacc = 0.0f;
    if (has_bias) {
        const size_t ix = (size_t)in * out + tid;
        bw = nn_ld<true>(W + ix);
        if (OPT != FWGPU_OPT_SGD) bacc = nn_ld<true>(A + ix);
    }
    __syncthreads();
    const uint32_t n_act = act[out];
    uint32_t G = (uint32_t)bd / nq;
    G = G > n_act ? (n_act ? n_act : 1u) : G;
    const uint32_t q = (uint32_t)tid % nq, grp = (uint32_t)tid / nq;
    const bool active = grp < G;
    const uint32_t jn = (n_act + G - 1) / G, jlo = grp * jn, jhi = jlo + jn < n_act ? jlo + jn : n_act;
    f4 xi = Vec<4>::zero(), oe = Vec<4>::zero();
    if (active) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = 4 * q + c;
            xi[c] = i < split ? in_a[i] : in_b[i - split];
        }
        constexpr int JU = FW_NN_VJU;  // neurons in flight per thread: 2 x JU x 16 bytes (8 in flight, and loading the next batch
                                       // while this one is stepped, both measured slower: registers)
        for (uint32_t a0 = jlo; a0 < jhi; a0 += JU) {
            f4 w[JU], a[JU];
            float gg[JU];
            uint32_t bo[JU];
            // (a slot beyond the thread's share works on an offset outside the layer: the buffer loads return 0 there and the stores are dropped, its gradient
            // is 0 -- every slot runs the same instructions, no lane mask per slot is kept from the loads to the stores)
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const bool on = a0 + u < jhi;
                const uint32_t j = on ? act[a0 + u] : 0u;
                gg[u] = on ? og[j] : 0.0f;
                bo[u] = on ? (j * in + 4 * q) * 4 : 0xfffffff0u;
                w[u] = Vec<4>::load<kAuxSc1>(rw, bo[u]);
                a[u] = OPT != FWGPU_OPT_SGD ? Vec<4>::load<kAuxSc1>(ra, bo[u]) : Vec<4>::zero();
            }
#pragma unroll
            for (int u = 0; u < JU; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float acc = a[u][c];
                    const float upd = opt_step<OPT>(gg[u] * xi[c], acc, n.rate, n.minus_power_t, n.lut);
                    oe[c] += w[u][c] * gg[u];
                    w[u][c] = w[u][c] - upd;
                    a[u][c] = acc;
                }
                Vec<4>::store<kAuxSc1>(w[u], rw, bo[u]);
                if (OPT != FWGPU_OPT_SGD) Vec<4>::store<kAuxSc1>(a[u], ra, bo[u]);
            }
        }
    }
    __syncthreads();  // every thread has read its inputs: the gradients may overwrite them (in_vals and in_grad alias)
    if (grp == 0) {   // (nq <= workgroup size: group 0 always exists; with no active neuron its sums are the zeros the caller needs)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = 4 * q + c;
            if (i < split) grad_a[i] = oe[c];
            else grad_b[i - split] = oe[c];
        }
    }
    __syncthreads();
    if (active && grp != 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i = 4 * q + c;
            atomicAdd(i < split ? &grad_a[i] : &grad_b[i - split], oe[c]);
        }
    }
    if (has_bias) {
        const size_t ix = (size_t)in * out + tid;
        const float gg = og[tid];
        const float upd = opt_step<OPT>(gg, bacc, n.rate, n.minus_power_t, n.lut);
        nn_st<true>(W + ix, bw - upd);
        if (OPT != FWGPU_OPT_SGD) nn_st<true>(A + ix, bacc);
    }
}

// one layer's backward: the 16-byte form in hogwild launches where the layer's shape allows it (uniform across the workgroup)
template <int OPT, bool COH>
__device__ __forceinline__ void nn_layer_backward_any(const DevNN &n, uint32_t l, const float *og, const float *in_a,
                                                      uint32_t split, const float *in_b, float *grad_a, float *grad_b,
                                                      uint32_t *act, int tid, int bd, bool concurrent) {
    const uint32_t in = n.in[l];
    if (COH && concurrent && in >= 4 && (in & 3u) == 0 && (n.off[l] & 3u) == 0 && (in >> 2) <= (uint32_t)bd && n.out[l] <= (uint32_t)bd && bd >= 64)
        nn_layer_backward_vec<OPT>(n, l, og, in_a, split, in_b, grad_a, grad_b, act, tid, bd);
    else
        nn_layer_backward<OPT, COH>(n, l, og, in_a, split, in_b, grad_a, grad_b, tid, bd);
}

// The unwinding of the head: final neuron, then [ReLU mask, layer] from last to first, BlockCopy summing both branches.
// Leaves d logit/d x in xg[] (LR slot gradients first, then the triangle gradients).
template <int OPT, bool COH>
__device__ __forceinline__ void nn_backward(const KernelParams &p, const Lds &s, float g, int tid, int bd) {
    const NnBuf b = nn_buf(p, s);
    const DevNN &n = p.nn;
    const uint32_t L = n.n_layers, X = n.X, wl = n.out[L - 1];
    uint32_t hoff_last = n.sum_width - wl;
#ifdef FW_TICKS
    unsigned long long nn_last = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0) b.fg[0] = g;  // og of the single final neuron
    __syncthreads();
    // final neuron: inputs [h_last | x], input gradients -> [h_last (in place) | xg]
    nn_layer_backward_any<OPT, COH>(n, L, b.fg, b.h + hoff_last, wl, b.x, b.h + hoff_last, b.xg, b.act, tid, bd, p.grid_wgs > 1);
    if (n.topology != 1)
        for (uint32_t i = tid; i < X; i += bd) b.xg[i] = 0.0f;
    __syncthreads();
    FW_NN_TICK(12);
    uint32_t hoff = hoff_last;
    for (int l = (int)L - 1; l >= 0; --l) {
        const uint32_t out = n.out[l];
        // BlockRELU backward (block_relu.rs:105-110): mask * upstream gradient -> this layer's output gradient
        for (uint32_t j = tid; j < out; j += bd) b.m[hoff + j] = b.m[hoff + j] * b.h[hoff + j];
        __syncthreads();
        if (l > 0) {
            const uint32_t pin = n.out[l - 1];
            nn_layer_backward_any<OPT, COH>(n, l, b.m + hoff, b.h + hoff - pin, pin, b.h + hoff - pin, b.h + hoff - pin,
                                            b.h + hoff - pin, b.act, tid, bd, p.grid_wgs > 1);
            hoff -= pin;
        } else {
            // first layer: inputs x; its input gradient is ADDED to the copy branch (BlockCopy, block_misc.rs:456-475)
            nn_layer_backward_any<OPT, COH>(n, 0, b.m + hoff, b.x, X, b.x, b.fg, b.fg, b.act, tid, bd, p.grid_wgs > 1);
            __syncthreads();
            for (uint32_t i = tid; i < X; i += bd) b.xg[i] = b.fg[i] + b.xg[i];
        }
        __syncthreads();
        FW_NN_TICK(13 + (l ? 0 : 1));
    }
}

// LR update (block_lr.rs:135-150).  Entries with the same hash are applied by the thread owning the FIRST occurrence, in
// buffer order, on one register copy of {w, acc}: duplicates chain exactly like the reference's loop (regressor.rs:629-655
// pins this).  The entry is read again here rather than kept from the forward pass: keeping it saved no time and widened
// the hogwild read-modify-write window of hot entries (constant feature) by two phases.
// `use_kept` / `kept`: the {w, acc} pair of entry `tid` as this thread's forward pass read it (v2 kernel): that entry is stepped without a second load -- the
// load's round trip was 16 % of an example's time (profiles/r04_phase_ticks.txt) for 0.4 % of its bytes.  In order it is the same pair a reload would
// return (nothing writes the entry between an example's forward pass and its update); concurrently the entry's read-modify-write window grows from the
// update's round trip to the dot + sigmoid phases, ~6 of an example's ~100 us.
template <int OPT, bool COH, bool SH = false>
__device__ __forceinline__ void lr_update(const KernelParams &p, const Lds &s, uint32_t nl, float g, const float *gx,
                                          const float *lut_lr, int tid, int bd, uint32_t lo = 0, uint32_t hi = 0xffffffffu, bool use_kept = false,
                                          float2 kept = float2{0.0f, 0.0f}, uint32_t thin_seed = 0xffffffffu) {
    for (uint32_t t = tid; t < nl; t += bd) {
        const uint32_t fl = s.l_flag[t];
        const uint32_t h = s.l_hash[t];
        if (COH && hot_lr_is(s, h)) {  // (every entry of that hash steps from the accumulator the forward pass read; chains are for the table route)
            float *entry = lr_base<SH>(p, h) + 2 * (size_t)h;
            float *hot = hot_lr_state(s);
            const float grad = (gx ? gx[s.l_combo[t]] : g) * s.l_val[t];
            float acc = hot[0];
            const float upd = opt_step<OPT>(grad, acc, p.lr_rate, p.lr_minus_power_t, lut_lr);  // (acc += g^2 first: optimizer.rs:76-77, 147-149)
            if (OPT != FWGPU_OPT_SGD) __hip_atomic_fetch_add(entry + 1, grad * grad, __ATOMIC_RELAXED, SH ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
            if (s.ctr[13] == 1) {
                __hip_atomic_fetch_add(entry, -upd, __ATOMIC_RELAXED, SH ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
            } else {
                atomicAdd(hot + 1, -upd);
                if (!(fl & kRowChained)) {  // (one entry of that hash per example is not chained)
                    const uint32_t n = s.ctr[7] + 1;
                    s.ctr[7] = n;
                    if (n >= s.ctr[13]) hot_lr_flush<SH>(p, s);
                }
            }
            continue;
        }
        if (fl & kRowChained) continue;
        if (h < lo || h >= hi) continue;  // sharded tables: another rank's entry
        float2 wa = kept;  // (by value: a pointer to the caller's pair would put it in scratch memory)
        if (!(use_kept && t == (uint32_t)tid)) wa = lr_load<COH, SH>(lr_base<SH>(p, h), h);
        const float acc_read = wa.y;
        {
            const float grad = (gx ? gx[s.l_combo[t]] : g) * s.l_val[t];
            wa.x -= opt_step<OPT>(grad, wa.y, p.lr_rate, p.lr_minus_power_t, lut_lr);
        }
        if (fl & kRowHasChain)
            for (uint32_t j = t + 1; j < nl; ++j)
                if (s.l_hash[j] == h) {
                    const float grad = (gx ? gx[s.l_combo[j]] : g) * s.l_val[j];
                    wa.x -= opt_step<OPT>(grad, wa.y, p.lr_rate, p.lr_minus_power_t, lut_lr);
                }
        // Store policy 4 on the LR block (thin_seed given: concurrent launches of the large-table kernel, KernelParams::lr_thin): an entry that is HOT -- its accumulator beyond
        // lr_hot_theta: an id that thousands of examples have stepped -- is read-modify-written by every example in flight that holds it, and an 8-byte store of {w, acc} that
        // loses its race loses that example's g^2 (the FFM rows' story, tests/test_gpu_conservation.py).  So the weight alone is stored (4 bytes), and one example in m ADDS m
        // times what it added to the accumulator with a fire-and-forget float atomic: every g^2 is counted in expectation, as the reference's threads count it (optimizer.rs:147-149).
        if (COH && !SH && OPT != FWGPU_OPT_SGD && thin_seed != 0xffffffffu && acc_read > p.lr_hot_theta) {
            float *entry = lr_base<SH>(p, h) + 2 * (size_t)h;
            __hip_atomic_store(reinterpret_cast<unsigned *>(entry), __float_as_uint(wa.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t m = 1u << p.acc_sample_log2;
            if (((((thin_seed * 2654435761u) ^ (h * 40503u)) >> 9) & (m - 1u)) == 0u)
                __hip_atomic_fetch_add(entry + 1, (float)m * (wa.y - acc_read), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        lr_store<COH, SH>(lr_base<SH>(p, h), h, wa);
    }
}

// Update of one FFM feature row by one wave (block_ffm.rs:269-286), U rows at a time for memory-level
// parallelism.  idx[u] == 0xffffffff marks an unused slot.
// Chunks of a row (64 lanes x VEC floats each) whose loads are issued together: rows of up to FW_UPD_CG chunks -- 512 floats at 16 bytes per lane, i.e. config
// E's 480 -- take ONE memory round trip per batch of U rows instead of one per chunk (round 5: the generic kernel walked the chunks one after the other,
// 12.5 round trips per wave and example at config E where 6.25 do).
#ifndef FW_UPD_CG
#define FW_UPD_CG 2
#endif
template <int VEC, int OPT, int AUX, int U, bool SH = false, int CG = FW_UPD_CG>
__device__ __forceinline__ void update_rows(const KernelParams &p, const Lds &s, const uint32_t (&idx)[U], float g,
                                            int lane, const float *gpair = nullptr, uint32_t nf = 0) {
    typedef typename Vec<VEC>::type V;
    const uint32_t R = p.R, k = p.k;
    const uint32_t nchunk = (R + 64 * VEC - 1) / (64 * VEC);
    for (uint32_t c0 = 0; c0 < nchunk; c0 += CG) {
        V wv[U][CG], av[U][CG];
        __amdgpu_buffer_rsrc_t rw[U], ra[U];
        uint32_t fld[U], hsh[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fld[u] = hsh[u] = 0;
            const bool on = idx[u] != 0xffffffffu;
            if (on) {
                const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[idx[u]]);
                hsh[u] = h;
                fld[u] = __builtin_amdgcn_readfirstlane(s.e_fld[idx[u]]);
                rw[u] = make_rsrc(ffm_w_base<SH>(p, h) + h, R * 4);
                if (OPT != FWGPU_OPT_SGD) ra[u] = make_rsrc(ffm_acc_base<SH>(p, h) + h, R * 4);
            }
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) {
                wv[u][cc] = Vec<VEC>::zero();
                av[u][cc] = Vec<VEC>::zero();
                if (on && c0 + cc < nchunk) {  // (lanes beyond the row read 0: the descriptor ends at the row's end)
                    const uint32_t e0 = ((c0 + cc) * 64 + lane) * VEC;
                    wv[u][cc] = Vec<VEC>::template load<AUX>(rw[u], e0 * 4);
                    if (OPT != FWGPU_OPT_SGD) av[u][cc] = Vec<VEC>::template load<AUX>(ra[u], e0 * 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx[u] == 0xffffffffu) continue;
#pragma unroll
            for (int cc = 0; cc < CG; ++cc) {
                if (c0 + cc >= nchunk) continue;
                const uint32_t e0 = ((c0 + cc) * 64 + lane) * VEC;
                const bool inb = e0 < R;
                const uint32_t z = inb ? e0 / k : 0;
                // one occurrence of the row (entry i of field f): gradient from the pre-update weights (T, selfw), optimizer step on
                // the running register copy
                auto apply = [&](uint32_t i, uint32_t f) {
                    const float v = s.e_val[i];
                    V tv = Vec<VEC>::zero(), sw = Vec<VEC>::zero();
                    const bool self = inb && (z == f);
                    if (inb) tv = Vec<VEC>::lds_load(s.T + f * R + e0);
                    if (self) sw = Vec<VEC>::lds_load(s.selfw + i * k + (e0 - z * k));
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float t = Vec<VEC>::get(tv, j);
                        if (self) t = __fsub_rn(t, __fmul_rn(Vec<VEC>::get(sw, j), v));  // contra - w*v  block_ffm.rs:238
                        const float G = __fmul_rn(v, t);             // gradient cache      block_ffm.rs:239, 249
                        // general gradient of output (f, z): uniform g, or -- deep head -- the mirrored triangle gradient
                        // (block_misc.rs:822-832)
                        const float gz = gpair ? gpair[tri_index(f, VEC == 1 ? (inb ? e0 / k : 0) : z)] : g;
                        const float grad = __fmul_rn(gz, G);          // block_ffm.rs:278
                        float acc = Vec<VEC>::get(av[u][cc], j);
                        const float upd = opt_step<OPT>(grad, acc, p.ffm_rate, p.ffm_minus_power_t, s.lut);
                        Vec<VEC>::set(av[u][cc], j, acc);
                        Vec<VEC>::set(wv[u][cc], j, Vec<VEC>::get(wv[u][cc], j) - upd);  // block_ffm.rs:282
                    }
                };
                apply(idx[u], fld[u] & kFldMask);
                if (fld[u] & kRowHasChain) {  // later entries of the same hash, in buffer order (see update_rows_win)
                    for (uint32_t base = idx[u] + 1; base < nf; base += 64) {
                        const uint32_t j = base + lane;
                        unsigned long long m = __ballot(j < nf && s.e_hash[j] == hsh[u] && (s.e_fld[j] & kRowChained));
                        while (m) {
                            const uint32_t b = (uint32_t)__builtin_ctzll(m);
                            m &= m - 1;
                            const uint32_t jj = base + b;
                            apply(jj, __builtin_amdgcn_readfirstlane(s.e_fld[jj]) & kFldMask);
                        }
                    }
                }
                Vec<VEC>::template store<AUX>(wv[u][cc], rw[u], e0 * 4);
                if (OPT != FWGPU_OPT_SGD) Vec<VEC>::template store<AUX>(av[u][cc], ra[u], e0 * 4);
            }
        }
    }
}


// ---- whole-line ("window") update of FFM rows.
// A row is R floats from byte 4*hash of the table: 32 B aligned at k = 8, so its first and last 128 B line are usually
// partial.  tools/rowceil.hip (profiles/r02_rowceil.txt): a partial-line write costs as much as 4-5 whole lines at the
// memory side (fill + write-back), which is what held random 960 B row writes at 3.2-3.5 TB/s while 1024 B whole-line
// rows write at 6.8 TB/s.  Here the wave reads and writes back the WHOLE lines the row touches: window = the 128 B
// aligned span [4h & ~127, round_up(4h + 4R, 128)), 16 B per lane, 64 lanes per 1 KiB chunk.  Lanes outside the row
// carry the neighbouring weights through unchanged.  In the in-order mode this is exact; between concurrent examples it
// widens hogwild's unsynchronised read-modify-write from "the same float" to "the same 128 B line" for the (at most two)
// edge lines of a row.  Rows whose WINDOWS intersect inside one example are serialised like overlapping rows.
// KernelParams::line_pass selects it per launch: the whole lines pay (-6.5 %) while w and acc contend for one region of the
// device memory; with acc placed away from w (regressor.cpp place_ffm_acc) the float-granular accesses of the same code path
// (sb = 0, nb = 4R: exactly the row) are as fast, and the race stays at the float.  The duplicate-row chains below are what
// this path keeps in both cases.
template <int OPT, int AUX, int U, int NCH, int AUX_SW = AUX, int AUX_SA = AUX>
__device__ __forceinline__ void update_rows_win(const KernelParams &p, const Lds &s, const uint32_t (&idx)[U], float g,
                                                int lane, uint32_t nf, const float *gpair = nullptr, uint32_t thin_seed = 0xffffffffu) {
    const uint32_t R = p.R, k = p.k, ksh = p.k_log2;
    // NCH 1 KiB chunks per row, all loaded before anything is computed (one memory round trip per batch of U rows).
    // A k = 8 row that starts 96 B into a line spans 9 lines: its second chunk is the one extra line (8 lanes).
    f4 wv[U][NCH], av[U][NCH];
    uint32_t hh[U], sb[U], nb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        hh[u] = sb[u] = nb[u] = 0;
        if (idx[u] != 0xffffffffu) {
            hh[u] = __builtin_amdgcn_readfirstlane(s.e_hash[idx[u]]);
            sb[u] = (hh[u] * 4u) & 127u;                          // row start within its first line, bytes
            nb[u] = (sb[u] + R * 4u + 127u) & ~127u;              // whole lines covered, bytes
            if (!p.line_pass || nb[u] > (uint32_t)NCH * 1024u) {  // (wave-uniform) float-granular launch, or more lines than NCH chunks hold:
                sb[u] = 0;                                        // this row keeps float-granular accesses
                nb[u] = R * 4u;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            wv[u][c] = f4{0.f, 0.f, 0.f, 0.f};
            av[u][c] = wv[u][c];
            if (nb[u] > (uint32_t)c * 1024u) {                    // wave-uniform
                const uint32_t fl = hh[u] - (sb[u] >> 2);         // float index of the window start
                wv[u][c] = Vec<4>::template load<AUX>(make_rsrc(p.ffm_w + fl, nb[u]), c * 1024 + lane * 16);
                if (OPT != FWGPU_OPT_SGD)
                    av[u][c] = Vec<4>::template load<AUX>(make_rsrc(p.ffm_acc + fl, nb[u]), c * 1024 + lane * 16);
            }
        }
    }
    // Every row of the batch is stepped before any row is stored.  vmcnt counts in issue order: a row's AdaGrad-table lookups (through L1 in the two-chunk launches) and the
    // next chunk's would otherwise wait behind the acknowledgement of the write-through stores just issued -- one drain per chunk; with U rows in flight that is what kept
    // the batch from paying (round 6).  The stepped values replace the loaded ones in place (lanes outside the row keep theirs bit for bit); what the accumulator side of a
    // chunk does -- store the new row, store nothing, add a delta -- is one bit each of two scalar registers.
    uint32_t st_mask = 0, add_mask = 0;  // bit u * NCH + c: the chunk's accumulators are stored / added (store policy 4, a hot row on the example's turn: av then holds the delta)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (nb[u] == 0) continue;
        const uint32_t fbits = __builtin_amdgcn_readfirstlane(s.e_fld[idx[u]]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (nb[u] <= (uint32_t)c * 1024u) continue;
            const int e = c * 256 + lane * 4 - (int)(sb[u] >> 2);    // this lane's first element of the row
            const bool inb = e >= 0 && e < (int)R;
            const uint32_t z = inb ? (uint32_t)e >> ksh : 0xfffffffeu;
            f4 wn = wv[u][c], an = av[u][c];
            // one occurrence of the row (entry i): gradient from the pre-update weights (T and selfw were taken in the
            // gather), AdaGrad step on the running (wn, an)
            auto apply = [&](uint32_t i, uint32_t f) {
                const float v = s.e_val[i];
                const bool self = z == f;
                f4 tv = f4{0.f, 0.f, 0.f, 0.f}, sw = tv;
                if (inb) tv = Vec<4>::lds_load(s.T + f * R + e);
                // (the entry's own slot: the LDS copy the gather left, or -- concurrent launches, resolve_row_mode -- this lane's part of the row as just re-read, before any step of this example)
                if (self) sw = (NCH > 1 && p.no_selfw) ? wv[u][c] : Vec<4>::lds_load(s.selfw + i * k + (e - (int)(z << ksh)));
                const float gz = (gpair && inb) ? gpair[tri_index(f, z)] : g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = tv[j];
                    if (self) t = __fsub_rn(t, __fmul_rn(sw[j], v));     // contra - w*v  block_ffm.rs:238
                    const float G = __fmul_rn(v, t);                     // block_ffm.rs:239, 249
                    const float grad = __fmul_rn(gz, G);                 // block_ffm.rs:278
                    float acc = an[j];
                    const float upd = opt_step<OPT>(grad, acc, p.ffm_rate, p.ffm_minus_power_t, s.lut);
                    an[j] = acc;
                    wn[j] = wn[j] - upd;                                 // block_ffm.rs:282
                }
            };
            apply(idx[u], fbits & kFldMask);
            if (fbits & kRowHasChain) {
                // later entries with the same hash, in buffer order: exactly the reference's sequence of updates on this row
                // (block_ffm.rs:269-286 walks the buffer; each occurrence sees the previous one's w and acc)
                for (uint32_t base = idx[u] + 1; base < nf; base += 64) {
                    const uint32_t j = base + lane;
                    unsigned long long m = __ballot(j < nf && s.e_hash[j] == hh[u] && (s.e_fld[j] & kRowChained));
                    while (m) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        const uint32_t jj = base + b;
                        apply(jj, __builtin_amdgcn_readfirstlane(s.e_fld[jj]) & kFldMask);
                    }
                }
            }
            if (!inb) {  // neighbouring weights: written back bit for bit
                wn = wv[u][c];
                an = av[u][c];
            }
            // Store policies 3 / 4 on re-read rows (thin_seed given: two-chunk rows under policy 3, every re-read row under policy 4): a row whose accumulators exceed
            // acc_hot_theta is touched by one example in m only.  Policy 3: that example stores av + m (an - av) (unbiased; the step above was taken with the true running
            // accumulator); policy 4: it ADDS m (an - av) -- what it and its chained duplicates added, m-fold -- and nobody stores (lanes outside the row add 0).
            bool st = OPT != FWGPU_OPT_SGD, ad = false;
            if (OPT != FWGPU_OPT_SGD && thin_seed != 0xffffffffu) {
                const f4 a0 = av[u][c];
                const bool hot = __ballot(inb && (a0[0] > p.acc_hot_theta || a0[1] > p.acc_hot_theta || a0[2] > p.acc_hot_theta || a0[3] > p.acc_hot_theta)) != 0ull;
                if (hot) {
                    const uint32_t m = 1u << p.acc_sample_log2;
                    const uint32_t draw = ((thin_seed * 2654435761u) ^ (idx[u] * 40503u + (uint32_t)c * 9973u)) >> 9;
                    const bool turn = (draw & (m - 1u)) == 0u;
                    if (p.store_policy == 4) {
                        st = false;
                        ad = turn;
                        const float mf = (float)m;
#pragma unroll
                        for (int j = 0; j < 4; ++j) an[j] = mf * (an[j] - a0[j]);
                    } else {
                        st = turn;
                        const float ms = (float)(m - 1u);
#pragma unroll
                        for (int j = 0; j < 4; ++j) an[j] = an[j] + ms * (an[j] - a0[j]);
                    }
                }
            }
            wv[u][c] = wn;
            av[u][c] = an;
            st_mask |= (st ? 1u : 0u) << (u * NCH + c);
            add_mask |= (ad ? 1u : 0u) << (u * NCH + c);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (nb[u] == 0) continue;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (nb[u] <= (uint32_t)c * 1024u) continue;
            const uint32_t fl = hh[u] - (sb[u] >> 2);
            Vec<4>::template store<AUX_SW>(wv[u][c], make_rsrc(p.ffm_w + fl, nb[u]), c * 1024 + lane * 16);
            if (OPT != FWGPU_OPT_SGD) {
                Vec<4>::template store<AUX_SA>(av[u][c], make_rsrc(p.ffm_acc + fl, ((st_mask >> (u * NCH + c)) & 1u) ? nb[u] : 0u), c * 1024 + lane * 16);
                if ((add_mask >> (u * NCH + c)) & 1u) {  // (wave-uniform; fire-and-forget)
                    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.ffm_acc + fl, nb[u]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(av[u][c][j], ra, c * 1024 + lane * 16 + 4 * j, 0, kAuxSc1);
                }
            }
        }
    }
}

// split record of one example (floats): T[F*R] | dcf[F] | LR sums[nlr] | feature count per field[F] | label | importance | pad
__host__ __device__ inline uint32_t split_len_of(uint32_t F, uint32_t R, uint32_t nlr) { return (F * R + F + nlr + F + 2 + 3) & ~3u; }
#ifndef FW_PHASE_TU
uint32_t split_record_len(uint32_t F, uint32_t R, uint32_t nlr) { return split_len_of(F, R, nlr); }
#endif

// LDS carve-up of the example kernels (lds_layout), the overlap pre-filter's geometry and the LDS copy of the translator's tables
__device__ __forceinline__ void bind_lds(const KernelParams &p, unsigned char *smem, bool use_lut, Lds &s, SetGeom &geom, TrLds &trl) {
    size_t off[25];
    lds_layout(p.F, p.k, p.max_ffm, p.max_lr, use_lut ? 1 : 0, p.records ? p.max_rec : 0,
               p.records ? p.tr.n_pairs + p.tr.n_combos + 2 : 0, nn_lds_floats(p), p.chain != 0, off, (p.records && p.prefetch) ? p.max_rec : 0,
               (p.records && p.tr_lds) ? tr_lds_words(p.tr) : 0, p.lds_keep_words, !p.t_global, !p.no_selfw);
    s.T = reinterpret_cast<float *>(smem + off[0]);
    s.selfw = reinterpret_cast<float *>(smem + off[1]);
    s.lut = reinterpret_cast<float *>(smem + off[2]);
    s.e_hash = reinterpret_cast<uint32_t *>(smem + off[3]);
    s.e_val = reinterpret_cast<float *>(smem + off[4]);
    s.e_fld = reinterpret_cast<uint32_t *>(smem + off[5]);
    s.l_hash = reinterpret_cast<uint32_t *>(smem + off[6]);
    s.l_val = reinterpret_cast<float *>(smem + off[7]);
    s.fstart = reinterpret_cast<uint32_t *>(smem + off[8]);
    s.fend = reinterpret_cast<uint32_t *>(smem + off[9]);
    s.red = reinterpret_cast<float *>(smem + off[10]);
    s.ctr = reinterpret_cast<uint32_t *>(smem + off[11]);
    s.dcf = reinterpret_cast<float *>(smem + off[12]);
    s.set_ffm = reinterpret_cast<uint32_t *>(smem + off[13]);
    s.set_lr = reinterpret_cast<uint32_t *>(smem + off[14]);
    s.rec = reinterpret_cast<uint32_t *>(smem + off[15]);
    s.l_flag = reinterpret_cast<uint32_t *>(smem + off[21]);
    s.set_blk = reinterpret_cast<uint32_t *>(smem + off[20]);
    s.tcnt = reinterpret_cast<uint32_t *>(smem + off[16]);
    s.l_combo = reinterpret_cast<uint32_t *>(smem + off[17]);
    s.nn = reinterpret_cast<float *>(smem + off[18]);
    s.rec_next = reinterpret_cast<uint32_t *>(smem + off[22]);
    s.keep = reinterpret_cast<float *>(smem + off[24]);
    geom.setf_n = set_size(p.max_ffm);
    geom.setl_n = set_size(p.max_lr);
    geom.setf_shift = 32 - log2u(geom.setf_n);
    geom.setl_shift = 32 - log2u(geom.setl_n);
    geom.blk_shift = conflict_blk_shift(p);

    if (!use_lut) s.lut = const_cast<float *>(p.lut_ffm);  // LUT read through L1 (the sc1 row traffic bypasses L1)
    uint32_t *tw = reinterpret_cast<uint32_t *>(smem + off[23]);
    trl.pair = tw;
    trl.coff = tw + p.tr.n_pairs;
    trl.cm = tw + p.tr.n_pairs + p.tr.n_combos + 1;
    trl.cw = reinterpret_cast<const float *>(tw + p.tr.n_pairs + p.tr.n_combos + 1 + p.tr.n_members);
}

// The carve-up alone, for the views the example kernels take per phase (stage / gather + dot / head unwinding / table update / tail / epilogue): what a phase
// does not use of it is dead code, what it uses is formed where the phase starts instead of living in scalar registers from an earlier phase on.
__device__ __forceinline__ Lds lds_view(const KernelParams &p, unsigned char *smem, bool use_lut) {
    Lds s;
    SetGeom geom_unused;
    TrLds trl_unused;
    bind_lds(p, smem, use_lut, s, geom_unused, trl_unused);
    return s;
}

// The kernels take their parameters by value (one argument block, scalar loads).  Left alone, the compiler loads every field once in the
// prologue of the persistent kernel and keeps ~180 scalars alive across the example loop -- 200 of them spilled to VGPR lanes, which in turn
// pushed 8 vector registers of the config-C learn kernel to scratch.  kp_fresh() hands out the argument block's address through an empty
// asm statement, so loads cannot move above the point of the call: called at the top of every example, the fields are (re)loaded where they
// are used (scalar cache hits) and the pressure goes away: 200 -> 85 spilled scalars, 8 -> 1 spilled vector registers, 36 -> 8 B/lane of scratch.
__device__ __forceinline__ const KernelParams &kp_fresh() {
    const __attribute__((address_space(4))) char *k = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; argument block address handed out" : "+s"(k));
    return *(const KernelParams *)k;  // (KernelParams is the kernels' first and only explicit argument: offset 0 of the segment)
}
// ------------------------------------------------------------------ owner-side apply, streaming form: the consumer side
// The first PushRings::consumers workgroups of a streaming launch drain this rank's circular regions as OWNER while the other workgroups -- and the other
// ranks' kernels -- fill them (staleness = the examples in flight, not the step: hogwild.rs:89-103 -- nobody waits for a step to end).  An eighth of the
// consumer waves take the LR regions (blocks of 64 positions, a lane per word), the others stripes of the row regions: source s = w % n, positions
// start + j, start + j + J, ...; a slot is handed back (lr_free / ffm_free, in the source's memory) as soon as its word / gradient row is in registers.  A wave leaves a region when the region's final position for THIS step is known (fin: stored by the
// source's last producer workgroup) and reached.
template <int OPT>
__device__ __forceinline__ void owner_stream_consume(const OwnerStream &os, uint32_t cw, uint32_t CW, uint32_t lane) {
    const uint32_t N = os.n;
    // (the host's way out of a step whose peer never arrives: dist.cpp stream_finish sets the word when the step outlives its deadline)
    // (the word lives in pinned host memory -- a write to it needs no queue of the device, whose CUs a waiting step holds -- so it is looked at once per 1024 polls of a wait loop)
    uint32_t polls = 0;
    auto aborted = [&]() -> bool { return (++polls & 1023u) == 0u && __hip_atomic_load(os.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u; };
    auto fin_of = [&](uint32_t idx, uint32_t &fin) -> bool {  // final position of ring idx (FFM: s, LR: n + s), once known
        const unsigned long long v = __hip_atomic_load(os.fin + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        fin = (uint32_t)v;
        return (uint32_t)(v >> 32) == os.step;
    };
    // consumer waves [0, NL): the LR regions.  Wave w: source w % n, blocks of 64 positions b = w / n, b + BL, ... of that source's region (BL waves per
    // source); a lane takes one word, steps its entry and hands the slot back.  NL = n x max(1, CW / (8 n)).
    // (at most 16 waves per source: 64 words a round each is ~300 M words/s, beyond what the producers send, and every waiting lane polls memory; and never more
    // blocks under way than HALF a region holds: the words' 2-bit generation tags tell a generation from its two neighbours, not from the third)
    uint32_t BL = CW / (8u * N);
    BL = BL > 16u ? 16u : BL;
    const uint32_t half_blocks = (1u << os.log2cap_lr) / 128u;
    BL = BL > half_blocks ? half_blocks : BL;
    BL = BL < 1u ? 1u : BL;
    const uint32_t NL = N * BL;
    if (cw < NL) {
        const uint32_t s = cw % N, lg = os.log2cap_lr, mask = (1u << lg) - 1u, gmask = lg ? (0xffffffffu >> lg) : 0xffffffffu;
        for (uint32_t base = os.start_lr[s] + (cw / N) * 64u;; base += BL * 64u) {
            const uint32_t pq = base + lane;
            unsigned long long word = 0;
            bool have = false, past = false;
            uint32_t fin = 0;
            for (;;) {  // this lane's position: produced, or never going to be
                word = __hip_atomic_load(os.lr_word[s] + (pq & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                have = (((uint32_t)word >> 30) & 3u) == ((pq >> lg) % 3u) + 1u;
                if (have) break;
                if ((fin_of(N + s, fin) && (int32_t)(pq - fin) >= 0) || aborted()) {
                    past = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(64);  // (~1 us: a waiting lane's poll is a memory request)
            }
            if (have) {
                __hip_atomic_store(os.lr_free[s] + (pq & mask), ((pq >> lg) + 1u) & gmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the word is in a register)
                const uint32_t h = (uint32_t)word & 0x3fffffffu;
                const float grad = __uint_as_float((uint32_t)(word >> 32));
                float2 wa = lr_load<true>(os.lr, h);
                wa.x -= opt_step<OPT>(grad, wa.y, os.lr_rate, os.lr_mpt, os.lut_lr);  // block_lr.rs:145-147
                lr_store<true>(os.lr, h, wa);
            }
            if (__ballot(past)) break;  // (positions are consecutive: every later block of this wave lies beyond the end as well)
        }
        return;
    }
    const uint32_t W = CW - NL, gw = cw - NL;
    const uint32_t s = gw % N, j = gw / N, J = W / N;
    if (j >= J) return;
    const uint32_t lg = os.log2cap_ffm, mask = (1u << lg) - 1u, gmask = lg ? (0xffffffffu >> lg) : 0xffffffffu, R = os.R;
    // KR positions of the stripe per round: their tag words are polled together (lane u loads tag u: one round trip), the rows of those that are there are
    // loaded together (one more), then stepped.  A round waits until each of its positions is either there or beyond the region's final position.
#ifndef FW_STREAM_KR
#define FW_STREAM_KR 2
#endif
    constexpr int KR = FW_STREAM_KR;
    for (uint32_t p0 = os.start_ffm[s] + j;; p0 += KR * J) {
        uint32_t hh[KR];
        bool on[KR];
        uint32_t n_past = 0;
        for (;;) {
            unsigned long long wv = 0;
            if (lane < (uint32_t)KR) {
                const uint32_t pq = p0 + lane * J;
                wv = __hip_atomic_load(os.ffm_tag[s] + (pq & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            uint32_t fin = 0, known = 0;
            bool all = true;
            n_past = 0;
#pragma unroll
            for (int u = 0; u < KR; ++u) {
                const uint32_t pq = p0 + (uint32_t)u * J, gen = (pq >> lg) & gmask;
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wv >> 32), u), lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wv, u);
                on[u] = hi == ((gen + 1u) & gmask);
                hh[u] = lo;
                if (!on[u]) {
                    if (!known) {  // (one look at the final position per poll)
                        uint32_t k_ = 0, f_ = 0;
                        if (lane == 0) k_ = fin_of(s, f_) ? 1u : 0u;
                        known = __builtin_amdgcn_readfirstlane(k_) ? 1u : 2u;
                        fin = __builtin_amdgcn_readfirstlane(f_);
                    }
                    if (known == 1u && (int32_t)(pq - fin) >= 0) n_past++;
                    else all = false;
                }
            }
            if (all) break;
            if (__builtin_amdgcn_readfirstlane(aborted() ? 1u : 0u)) {  // (the step is void: leave with what is there)
                n_past = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        // (positions of a stripe are consumed in order: once one is beyond the end, the later ones are too)
        f4 gv[KR][2], wv2[KR][2], av[KR][2];
        bool vec[KR];
#pragma unroll
        for (int u = 0; u < KR; ++u) {
            vec[u] = on[u] && ((R | hh[u]) & 3u) == 0;
            const uint32_t slot = (p0 + (uint32_t)u * J) & mask;
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(os.ffm_rows[s] + (size_t)slot * R, vec[u] ? R * 4 : 0), rw = make_rsrc(os.w + hh[u], vec[u] ? R * 4 : 0),
                                         ra = make_rsrc(os.acc + hh[u], vec[u] ? R * 4 : 0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {  // (rows of up to 512 floats: both chunks' loads before anything is stepped; a slot that is not there has zero-length descriptors)
                const uint32_t e0 = (c * 64 + lane) * 4;
                gv[u][c] = Vec<4>::load<kAuxSys>(rg, e0 * 4);
                wv2[u][c] = Vec<4>::load<kAuxSc1>(rw, e0 * 4);
                av[u][c] = OPT == FWGPU_OPT_SGD ? Vec<4>::zero() : Vec<4>::load<kAuxSc1>(ra, e0 * 4);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < KR; ++u) {
            if (!on[u]) continue;
            const uint32_t pq = p0 + (uint32_t)u * J, slot = pq & mask, gen = (pq >> lg) & gmask, h = hh[u];
            if (vec[u]) {
                if (R <= 512 && lane == 0)  // the gradient row is in registers: its slot may be written for the next generation
                    __hip_atomic_store(os.ffm_free[s] + slot, (gen + 1u) & gmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(os.ffm_rows[s] + (size_t)slot * R, R * 4), rw = make_rsrc(os.w + h, R * 4), ra = make_rsrc(os.acc + h, R * 4);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint32_t e0 = (c * 64 + lane) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float a = av[u][c][q];
                        wv2[u][c][q] = wv2[u][c][q] - opt_step<OPT>(gv[u][c][q], a, os.ffm_rate, os.ffm_mpt, os.lut_ffm);  // block_ffm.rs:279-282
                        av[u][c][q] = a;
                    }
                    Vec<4>::store<kAuxSc1>(wv2[u][c], rw, e0 * 4);
                    if (OPT != FWGPU_OPT_SGD) Vec<4>::store<kAuxSc1>(av[u][c], ra, e0 * 4);
                }
                for (uint32_t e0 = 512 + lane * 4; e0 < R; e0 += 256) {  // longer rows: the rest chunk by chunk
                    const f4 g2 = Vec<4>::load<kAuxSys>(rg, e0 * 4);
                    f4 w2 = Vec<4>::load<kAuxSc1>(rw, e0 * 4), a2 = OPT == FWGPU_OPT_SGD ? Vec<4>::zero() : Vec<4>::load<kAuxSc1>(ra, e0 * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float a = a2[q];
                        w2[q] = w2[q] - opt_step<OPT>(g2[q], a, os.ffm_rate, os.ffm_mpt, os.lut_ffm);
                        a2[q] = a;
                    }
                    Vec<4>::store<kAuxSc1>(w2, rw, e0 * 4);
                    if (OPT != FWGPU_OPT_SGD) Vec<4>::store<kAuxSc1>(a2, ra, e0 * 4);
                }
                if (R > 512) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(os.ffm_free[s] + slot, (gen + 1u) & gmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else {
                for (uint32_t e = lane; e < R; e += 64) {
                    const float grad = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(os.ffm_rows[s] + (size_t)slot * R + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                    float a = OPT == FWGPU_OPT_SGD ? 0.0f : __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(os.acc + h + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    float wv = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(os.w + h + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    wv -= opt_step<OPT>(grad, a, os.ffm_rate, os.ffm_mpt, os.lut_ffm);
                    __hip_atomic_store(reinterpret_cast<unsigned *>(os.w + h + e), __float_as_uint(wv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (OPT != FWGPU_OPT_SGD) __hip_atomic_store(reinterpret_cast<unsigned *>(os.acc + h + e), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(os.ffm_free[s] + slot, (gen + 1u) & gmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (n_past) break;
    }
}

// PH = 0: the fused learn / predict step.  PH = 1 (FWD) and PH = 3 (UPD): the two table-touching halves of the synchronous
// micro-batch pipeline -- every example of the batch sees the weights of the batch start, updates are applied afterwards:
//   FWD : stage, gather the rows this rank OWNS (all of them on one GPU), write T / dcf / LR sums to the example's split record
//   (exchange: records summed over the ranks; MID kernel: logit, prediction, general gradient; or the mini-batched deep head)
//   UPD : stage, T and the entries' own slots back from the records, AdaGrad on the owned rows and LR entries
template <int VEC, int OPT, bool COH, int PH = 0, bool NN = true, bool SH = false>
__global__ void __launch_bounds__(1024) fw_example_kernel(const KernelParams /* read through kp_fresh() */) {
    const KernelParams &p = kp_fresh();
    typedef typename Vec<VEC>::type V;
    // (peer-sharded tables: a row may live in another GPU's memory -- system-scope accesses there, device scope otherwise)
    constexpr int AUX = COH ? (SH ? kAuxSys : kAuxSc1) : kAuxPlain;
    // a serving context cache (ctx_*, emit_*) only ever comes with read-only launches of the whole kernel, never with the phases of a split step
    constexpr bool kCtx = !COH && PH == 0;
#ifndef FW_V1_UG
#define FW_V1_UG 8
#endif
#ifndef FW_V1_UU
#define FW_V1_UU 2
#endif
    constexpr int UG = FW_V1_UG;  // feature rows in flight per wave in the gather phase
    constexpr int UU = FW_V1_UU;  // feature rows in flight per wave in the update phase (x2 tables)
    extern __shared__ __align__(16) unsigned char smem[];
    const bool use_lut = (OPT == FWGPU_OPT_ADAGRAD_LUT) && p.update && !p.lut_global;
    Lds s;
    SetGeom geom;
    TrLds trl_unused;
    bind_lds(p, smem, use_lut, s, geom, trl_unused);

    const int tid = threadIdx.x, bd = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = bd >> 6;  // (the prologue's; every phase of the example loop derives its own)
    (void)lane;

    if (use_lut)
        for (int i = tid; i < kLutSize; i += bd) s.lut[i] = p.lut_ffm[i];

    // debug phase timing (thread 0 of every workgroup; all stamps sit right after a barrier or at a phase end): -DFW_TICKS builds only, like the v2
    // kernel's (scripts/perf_probe.py builds its own library) -- the eight sums and the stamp are 18 vector registers and a lane mask alive across the
    // whole example loop, in a kernel that runs at its 128-register limit with the deep head
#ifdef FW_TICKS
    unsigned long long tk_last = 0, tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool timing = p.ticks != nullptr && tid == 0;
#define FW_TICK(slot)                                           \
    if (timing) {                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tk[slot] += now_ - tk_last;                             \
        tk_last = now_;                                         \
    }
    if (timing) tk_last = __builtin_amdgcn_s_memtime();
#else
#define FW_TICK(slot)
#endif

    // Examples are handed out by a device counter, not by a static stride: a workgroup that becomes resident late (another
    // kernel -- RCCL during a replica exchange -- holds its slot, or the grid was over-subscribed) finds the work already
    // done instead of running its whole share alone after everyone else has finished.  One workgroup gets 0, 1, 2, ...
    // (the in-order mode).  The next ticket is requested one example ahead, so its latency is never waited for.
    if (SH && PH == 0 && p.push && p.push->stream && blockIdx.x < p.push->consumers) {
        // streaming owner-side apply: this workgroup is one of the rank's CONSUMERS for the whole launch (owner_stream_consume)
        owner_stream_consume<OPT>(*p.push->own, blockIdx.x * (uint32_t)nw + (uint32_t)wave, p.push->consumers * (uint32_t)nw, (uint32_t)lane);
        return;
    }
    if (tid == 0) {
        s.ctr[6] = atomicAdd(p.work, 1u);
        hot_lr_init<COH>(p, s, PH == 0);
        s.ctr[kCtrWbCount] = blockIdx.x & 15u;
    }
#ifndef FW_KP_NO_CANARY
    if (p.dbg_canary)  // debug: 256 words behind this kernel's own LDS layout; nobody may write there
        for (uint32_t i = tid; i < 256; i += bd) reinterpret_cast<uint32_t *>(smem + p.dbg_canary_off)[i] = 0xC0FFEE00u + i;
#endif
    for (;;) {
        // Previous example's LDS reads are done.  A workgroup-scope barrier does not drain vmcnt on this target,
        // so in the in-order (single workgroup) mode every wave first waits for its own table stores to be
        // acknowledged: the next example must read what this one wrote.  Concurrent (hogwild) grids skip the
        // wait and let the stores drain under the next example's gather.
        // The stage phase and the rest of the example each take their own view of the argument block and of the LDS carve-up (see fw_example_kernel_r):
        // what only the stage phase needs dies at its end instead of living in scalar registers through the gather and the update.
        StageOut so;
        uint32_t ex;
        {
            const KernelParams &p = kp_fresh();  // (shadows the prologue's: this example's loads start here)
            // (the thread index is made opaque once per example, so that the lane masks derived from it are recomputed where they are used instead of
            // living in scalar-register pairs across the whole example loop: see fw_example_kernel_r)
            int tid_now = threadIdx.x, bd_now = blockDim.x;
            uint32_t grid_now = p.grid_wgs;
            asm volatile("; thread index, workgroup and grid size handed out" : "+v"(tid_now), "+s"(bd_now), "+s"(grid_now));
            const int tid = tid_now, bd = bd_now;
            Lds s;
            SetGeom geom;
            TrLds trl_unused;
            bind_lds(p, smem, (OPT == FWGPU_OPT_ADAGRAD_LUT) && p.update && !p.lut_global, s, geom, trl_unused);
            if (grid_now == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ex = s.ctr[6];
            if (ex >= p.n_examples) break;
            FW_TICK(6);
#ifdef FW_TICKS
            if (timing) tk[7] += 1;
#endif
            // Phase kernels: T lives in the example's split record (FWD writes the field sums there as they are finished, UPD reads them from there):
            // without the F * R floats of LDS two to four workgroups share a CU instead of one, i.e. as many examples of a micro-batch run concurrently
            if (PH != 0 && p.t_global) s.T = p.split + (size_t)ex * p.split_len;
            so = stage_example<kCtx>(p, s, geom, ex, tid, bd, TrGlobal{p.tr});
        }
        const KernelParams &p = kp_fresh();
        int tid_now = threadIdx.x, bd_now = blockDim.x;
        asm volatile("; thread index and workgroup size handed out" : "+v"(tid_now), "+s"(bd_now));
        const int tid = tid_now, lane = tid & 63, wave = tid >> 6, bd = bd_now, nw = bd >> 6;
        Lds s = lds_view(p, smem, (OPT == FWGPU_OPT_ADAGRAD_LUT) && p.update && !p.lut_global);
        if (PH != 0 && p.t_global) s.T = p.split + (size_t)ex * p.split_len;
        const uint32_t F = p.F, k = p.k, R = p.R;  // (shadow the prologue's, like everything else the example works with)
        const uint32_t nchunk = R ? (R + 64 * VEC - 1) / (64 * VEC) : 0;
        uint32_t next_ticket = 0;  // (every thread is past its read of ctr[6]: the stage phase has barriers)
        if (tid == 0) next_ticket = atomicAdd(p.work, 1u);
        const uint32_t nf = so.nf, nl = so.nl;
        const float label = so.label, imp = so.imp;
        const bool do_update = so.do_update;
        FW_TICK(1);
        // ---------------- gather: field sums, transposed into LDS
        if (k && PH != 3) {
            // Static field -> wave assignment: wave w takes the fields whose first feature index falls into its share
            // (owner(f) = floor(fstart[f] * n_waves / n_features): contiguous, balanced by features, as in the v2 kernel).
            // (Round 1 handed fields out through an LDS ticket; that loop was only correct as long as the compiler did not
            // jump-thread a lane-conditional branch into it -- a hang inside a persistent kernel.  No atomics, no hazard.)
            for (uint32_t f = 0; f < F; ++f) {
                if (s.fstart[f] == s.fend[f] || (s.fstart[f] * (uint32_t)nw) / nf != (uint32_t)wave) continue;
                const uint32_t fs = s.fstart[f], fe = s.fend[f];
                // sum_{i in field f} v_i^2 |w_i[f*k..]|^2 (block_ffm.rs:418-426).  Reduced per field so that the
                // result does not depend on which wave happened to take the field.
                float dc = 0.0f;
                for (uint32_t c = 0; c < nchunk; ++c) {
                    const uint32_t e0 = (c * 64 + lane) * VEC;
                    const bool inb = e0 < R;
                    const uint32_t z = inb ? e0 / k : 0;
                    const bool self = inb && (z == f);
                    V acc = Vec<VEC>::zero();
                    if (kCtx && p.ctx_T && inb) {  // context cache: the cached features of this field come first (block_ffm.rs:548-556)
#pragma unroll
                        for (int j = 0; j < VEC; ++j) Vec<VEC>::set(acc, j, p.ctx_T[z * R + f * k + (e0 - z * k) + j]);
                    }
                    for (uint32_t i = fs; i < fe; i += UG) {
                        V r[UG];
                        float v[UG];
                        // (which of the UG rows exist: bits of ONE scalar register, tested afresh in the second loop -- as UG booleans every test result
                        // of the first loop was kept for the second as a 64-bit lane mask)
                        uint32_t onm = 0;
#pragma unroll
                        for (int u = 0; u < UG; ++u) {
                            r[u] = Vec<VEC>::zero();
                            v[u] = 0.0f;
                            if (i + u < fe) {
                                const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i + u]);
                                bool on = true;
                                if (PH == 1) on = h >= p.own_lo_ffm && h < p.own_hi_ffm;  // sharded tables: owned rows only
                                v[u] = s.e_val[i + u];
                                if (on) {
                                    onm |= 1u << u;
                                    r[u] = Vec<VEC>::template load<AUX>(make_rsrc(ffm_w_base<SH>(p, h) + h, R * 4), e0 * 4);
                                }
                            }
                        }
                        asm volatile("" : "+s"(onm));
#pragma unroll
                        for (int u = 0; u < UG; ++u) {
                            if ((onm >> u) & 1u) {
                                float ss = 0.0f;
#pragma unroll
                                for (int j = 0; j < VEC; ++j) {
                                    const float w = Vec<VEC>::get(r[u], j);
                                    Vec<VEC>::set(acc, j, __fadd_rn(Vec<VEC>::get(acc, j), __fmul_rn(w, v[u])));  // block_ffm.rs:205
                                    ss += w * w;
                                }
                                if (self) {
                                    dc += ss * v[u] * v[u];
                                    Vec<VEC>::lds_store(s.selfw + (i + u) * k + (e0 - z * k), r[u]);
                                }
                            }
                        }
                    }
                    if (inb) Vec<VEC>::lds_store(s.T + z * R + f * k + (e0 - z * k), acc);
                }
                dc = wave_sum(dc);
                if (kCtx && p.ctx_dcf) dc += p.ctx_dcf[f];
                s.dcf[f] = dc;  // same value from all 64 lanes
            }
        }
        __syncthreads();
        FW_TICK(2);
        if (kCtx && p.emit_T && ex == 0) {  // setup_cache: keep this example's field sums (Regressor::setup_cache, regressor.rs:409-423)
            for (uint32_t i = tid; i < F * R; i += bd) p.emit_T[i] = s.T[i];
            for (uint32_t f = tid; f < F; f += bd) p.emit_dcf[f] = s.dcf[f];
        }
        if (PH == 1) {
            // ---------------- FWD: this rank's share of the example goes to its split record
            float *rec = p.split + (size_t)ex * p.split_len;
            const bool home = ex >= p.home_lo && ex < p.home_hi;
            if (!p.t_global)
                for (uint32_t i = tid; i < F * R; i += bd) rec[i] = s.T[i];
            for (uint32_t f = tid; f < F; f += bd) {
                rec[F * R + f] = s.dcf[f];
                rec[F * R + F + p.split_nlr + f] = home ? (float)(s.fend[f] - s.fstart[f]) : 0.0f;
            }
            if (tid == 0) {
                rec[F * R + 2 * F + p.split_nlr] = home ? label : 0.0f;
                rec[F * R + 2 * F + p.split_nlr + 1] = home ? imp : 0.0f;
            }
            for (uint32_t i = tid; i < nf * k; i += bd) p.split_selfw[(size_t)ex * p.selfw_stride + i] = s.selfw[i];
            // row-sparse gradient mode (sparse.hip): every entry of the example becomes one occurrence, key = (row, slot).
            // Sorting the keys groups the occurrences of a row in (example, entry) order.
            if (p.occ_ffm_key || p.occ_lr_key) {
                const bool upd = imp != 0.0f;  // regressor.rs:366
                if (p.occ_ffm_key)
                for (uint32_t i = tid; i < p.max_ffm; i += bd) {
                    const uint32_t slot = ex * p.max_ffm + i;
                    const bool on = upd && i < nf;
                    p.occ_ffm_key[slot] = on ? (((unsigned long long)s.e_hash[i] << 32) | slot) : ~0ull;
                    if (on) p.occ_ffm_desc[slot] = uint2{__float_as_uint(s.e_val[i]), s.e_fld[i] & kFldMask};
                }
                if (p.occ_lr_key && p.has_lr)
                    for (uint32_t i = tid; i < p.max_lr; i += bd) {
                        const uint32_t slot = ex * p.max_lr + i;
                        const bool on = upd && i < nl;
                        p.occ_lr_key[slot] = on ? (((unsigned long long)s.l_hash[i] << 32) | slot) : ~0ull;
                        if (on) p.occ_lr_desc[slot] = uint2{__float_as_uint(s.l_val[i]), 0u};
                    }
            }
            if (NN && p.split_nlr > 1) {  // deep head: one sum per LR combo slot, entries in buffer order (block_lr.rs:36-45)
                for (uint32_t c = tid; c < p.split_nlr; c += bd) {
                    float acc = 0.0f;
                    for (uint32_t i = 0; i < nl; ++i) {
                        const uint32_t h = s.l_hash[i];
                        if (s.l_combo[i] == c && h >= p.own_lo_lr && h < p.own_hi_lr) acc += lr_load<COH>(p.lr, h).x * s.l_val[i];
                    }
                    rec[F * R + F + c] = acc;
                }
            } else {
                float lrs = 0.0f;
                if (p.has_lr)
                    for (uint32_t i = tid; i < nl; i += bd) {
                        const uint32_t h = s.l_hash[i];
                        if (h >= p.own_lo_lr && h < p.own_hi_lr) lrs += lr_load<COH>(p.lr, h).x * s.l_val[i];
                    }
                lrs = wave_sum(lrs);
                if (lane == 0) s.red[32 + wave] = lrs;
                __syncthreads();
                if (tid == 0) {
                    float t = 0.0f;
                    for (int w = 0; w < nw; ++w) t += s.red[32 + w];
                    rec[F * R + F] = t;
                }
            }
            if (tid == 0) s.ctr[6] = next_ticket;
            continue;
        }
        float g_split = 0.0f;
        if (PH == 3) {
            // ---------------- UPD: the batch-start field sums and own slots come back from the records
            const float *rec = p.split + (size_t)ex * p.split_len;
            if (!p.t_global)
                for (uint32_t i = tid; i < F * R; i += bd) s.T[i] = rec[i];
            for (uint32_t i = tid; i < nf * k; i += bd) s.selfw[i] = p.split_selfw[(size_t)ex * p.selfw_stride + i];
            g_split = p.gbuf[ex];
            __syncthreads();
        }

        // ---------------- all-pairs dot from LDS + LR forward
        float dot = 0.0f;
        if (k && PH == 0) {
            const uint32_t nq = F * R / VEC;
            for (uint32_t q = tid; q < nq; q += bd) {
                const uint32_t e0 = q * VEC;
                const uint32_t a = e0 / R, rem = e0 - a * R;
                const uint32_t b = rem / k, kk = rem - b * k;
                const V x = Vec<VEC>::lds_load(s.T + e0);
                const V y = Vec<VEC>::lds_load(s.T + b * R + a * k + kk);
#pragma unroll
                for (int j = 0; j < VEC; ++j) dot += Vec<VEC>::get(x, j) * Vec<VEC>::get(y, j);
            }
        }
        float pr, g;
        if (PH == 3) {
            pr = 0.0f;
            g = g_split;
        } else {
        float lrs = 0.0f;
        if (p.has_lr)
            for (uint32_t i = tid; i < nl; i += bd) {
                lrs += lr_forward_weight<COH, SH>(p, s, s.l_hash[i]) * s.l_val[i];
            }
        dot = wave_sum(dot);
        lrs = wave_sum(lrs);
        if (lane == 0) {
            s.red[wave] = dot;
            s.red[32 + wave] = lrs;
        }
        __syncthreads();
        float dot_t = 0.0f, dc_t = 0.0f, lr_t = 0.0f;
        for (int w = 0; w < nw; ++w) {
            dot_t += s.red[w];
            lr_t += s.red[32 + w];
        }
        for (uint32_t f = 0; f < F; ++f) dc_t += s.dcf[f];
        // sigmoid input: LR slots first, then the FFM pair sum (graph.rs:251-285 tape order)
        float wsum = 0.0f;
        if (p.has_lr) wsum += lr_t;
        if (k) wsum += 0.5f * (dot_t - dc_t);
        // deep head: the sigmoid sees the final neuron's output instead (regressor.rs:191-323)
        if (NN && p.nn.n_layers) wsum = nn_forward<VEC, COH>(p, s, nl, tid, bd);

        // ---------------- sigmoid / log-loss gradient (block_loss_functions.rs:105-153)
        if (isnan(wsum)) {
            pr = logistic(0.0f);
            g = 0.0f;
        } else if (wsum < -50.0f) {
            pr = logistic(-50.0f);
            g = 0.0f;
        } else if (wsum > 50.0f) {
            pr = logistic(50.0f);
            g = 0.0f;
        } else {
            pr = logistic(wsum);
            g = -(label - pr) * imp;
        }
        if (tid == 0) p.pred[ex] = pr;
        }  // PH != 3
        FW_TICK(3);

        // ---------------- update.  g == 0 leaves every weight and accumulator unchanged in all three
        // optimizers (acc += 0, w -= 0), so the whole phase is skipped.
        const bool head_split = NN && PH == 3 && p.dxbuf != nullptr;  // mini-batched deep head: per-slot gradients come from the head kernels
        // ("does this example update" is formed again from the argument block and the importance, regressor.rs:366: as the stage phase's own boolean it is
        // a lane mask alive from the stage phase through the gather and the head's forward pass)
        const bool do_update_now = kp_fresh().update && imp != 0.0f;
        (void)do_update;
        if (do_update_now && (g != 0.0f || head_split)) {
            // The head's unwinding and the table update each take a view of their own, like the stage phase: the LDS offsets and table addresses they need are
            // formed where they start, not carried in scalar registers from the top of the example through the gather and the head's forward pass.
            // deep head: unwind it first; afterwards every LR slot and every FFM pair has its own general gradient
            if (NN && (head_split || p.nn.n_layers)) {
                const KernelParams &p = kp_fresh();
                int tid_h = threadIdx.x, bd_h = blockDim.x;
                asm volatile("; thread index and workgroup size handed out" : "+v"(tid_h), "+s"(bd_h));
                const int tid = tid_h, bd = bd_h;
                Lds s = lds_view(p, smem, (OPT == FWGPU_OPT_ADAGRAD_LUT) && p.update && !p.lut_global);
                if (head_split) {
                    // the example's slot gradients (X floats, written by the head kernels) come into LDS once: every row chunk of the update looks its pair's
                    // gradient up there instead of in global memory
                    float *xl = nn_buf(p, s).xg;
                    const float *src = p.dxbuf + (size_t)ex * p.nn.X;
                    for (uint32_t i = tid; i < p.nn.X; i += bd) xl[i] = src[i];
                    __syncthreads();
                } else {
                    nn_backward<OPT, COH>(p, s, g, tid, bd);
                }
            }
            const KernelParams &p = kp_fresh();
            int tid_u = threadIdx.x, bd_u = blockDim.x;
            asm volatile("; thread index and workgroup size handed out" : "+v"(tid_u), "+s"(bd_u));
            const int tid = tid_u, lane = tid & 63, wave = tid >> 6, bd = bd_u, nw = bd >> 6;
            Lds s = lds_view(p, smem, (OPT == FWGPU_OPT_ADAGRAD_LUT) && p.update && !p.lut_global);
            if (PH != 0 && p.t_global) s.T = p.split + (size_t)ex * p.split_len;
            const uint32_t F = p.F, k = p.k, R = p.R;
            const uint32_t nchunk = R ? (R + 64 * VEC - 1) / (64 * VEC) : 0;
            const float *lut_lr = p.lut_lr;
            (void)F; (void)k; (void)nchunk; (void)lane; (void)wave; (void)nw; (void)lut_lr;
            const float *gx = nullptr, *gpair = nullptr;
            if (NN && (head_split || p.nn.n_layers)) {
                gx = nn_buf(p, s).xg;
                gpair = gx + p.num_combos;
            }
            const uint32_t olo = PH == 3 ? p.own_lo_ffm : 0u, ohi = PH == 3 ? p.own_hi_ffm : 0xffffffffu;
            if (SH && p.push) {
                // ---------------- owner-side apply: this example's gradients travel to the rows' owners, who run the optimizer on their own tables
                // (dist.cpp fwgpu_dist_*_owner).  Every OCCURRENCE is pushed -- repeated and overlapping rows need no special care here: the owner
                // applies what it receives one after the other.  In-order launches push in buffer order from one wave / one thread, so that the
                // owner's sequence of steps is the reference's (block_ffm.rs:269-286, block_lr.rs:140-150).
                const PushRings &pr = *p.push;
                const bool in_order = p.grid_wgs == 1;
                if (pr.stream) {
                    // ---------------- streaming form: circular regions drained by the owners WHILE this kernel runs (owner_stream_kernel)
                    if (p.has_lr) {
                        const uint32_t lg = pr.log2cap_lr, capl = 1u << lg;
                        for (uint32_t t0 = 0; t0 < nl; t0 += bd) {  // (uniform trip count: the position allocation below is a wave-wide step)
                            const uint32_t t = t0 + (uint32_t)tid;
                            const bool act_ = t < nl;
                            const uint32_t h = act_ ? s.l_hash[t] : 0u, o = act_ ? h >> p.shards->shift_lr : 0xffffffffu;
                            // positions: ONE atomic per wave and owner (the counters are single addresses: a per-word atomic serialises 13 M of them per step)
                            uint32_t pos = 0;
                            for (uint32_t oo = 0; oo < pr.n; ++oo) {
                                const unsigned long long m_ = __ballot(o == oo);
                                if (!m_) continue;
                                uint32_t base_ = 0;
                                if ((uint32_t)lane == (uint32_t)__builtin_ctzll(m_)) base_ = atomicAdd(&pr.cnt[pr.n + oo], (uint32_t)__popcll(m_));
                                base_ = (uint32_t)__shfl((int)base_, (int)__builtin_ctzll(m_), 64);
                                if (o == oo) pos = base_ + (uint32_t)__popcll(m_ & ((1ull << lane) - 1ull));
                            }
                            if (!act_) continue;
                            const float grad = g * s.l_val[t];  // block_lr.rs:143
                            const unsigned long long word = (unsigned long long)(h | ((((pos >> lg) % 3u) + 1u) << 30)) | ((unsigned long long)__float_as_uint(grad) << 32);
                            // flow control: slot q may be written for generation g when the owner has said so (lr_free, in this rank's memory).  The store sits
                            // INSIDE the loop: a lane whose position is free must not wait (at the loop's end) for a lane of its wave whose position is not --
                            // the consumers take a region's blocks in order and might never get to that one.
                            const uint32_t lslot = pos & (capl - 1u), lgen = (pos >> lg) & (lg ? (0xffffffffu >> lg) : 0xffffffffu);
                            uint32_t lr_polls = 0;
                            for (bool sent = false; !sent;) {
                                if ((__hip_atomic_load(pr.lr_free[o] + lslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & (lg ? (0xffffffffu >> lg) : 0xffffffffu)) == lgen) {
                                    __hip_atomic_store(pr.lr_word[o] + lslot, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    sent = true;
                                } else if ((++lr_polls & 1023u) == 0u && __hip_atomic_load(pr.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                                    sent = true;  // (the host gave the step up: PushRings::abort, looked at once per 1024 polls)
                                } else {
                                    __builtin_amdgcn_s_sleep(8);
                                }
                            }
                        }
                    }
                    if (k) {
#ifndef FW_STREAM_PU
#define FW_STREAM_PU 4
#endif
                        constexpr int PU = FW_STREAM_PU;  // gradient rows a wave has under way at once
                        const uint32_t lg = pr.log2cap_ffm, capf = 1u << lg, gmask = lg ? (0xffffffffu >> lg) : 0xffffffffu;
                        // This wave's rows: i = wave, wave + nw, ...  Taken OWNER BY OWNER, so that the wave draws all its positions of an owner's region with one
                        // atomic (the counters are single addresses: one atomic per row serialises 13 M of them per step), then PU rows at a time.
                        for (uint32_t ow = 0; ow < pr.n; ++ow) {
                        uint32_t mine = 0;
                        for (uint32_t i = (uint32_t)wave; i < nf; i += (uint32_t)nw)
                            mine += (__builtin_amdgcn_readfirstlane(s.e_hash[i]) >> p.shards->shift_ffm) == ow ? 1u : 0u;
                        if (!mine) continue;
                        uint32_t next_pos = 0;
                        if (lane == 0) next_pos = atomicAdd(&pr.cnt[ow], mine);
                        next_pos = __builtin_amdgcn_readfirstlane(next_pos);
                        for (uint32_t iw = (uint32_t)wave, left = mine; left;) {
                            uint32_t h_[PU], o_[PU], pos_[PU], row_[PU];
#pragma unroll
                            for (int u = 0; u < PU; ++u) {
                                h_[u] = pos_[u] = 0;
                                o_[u] = ow;
                                row_[u] = 0xffffffffu;
                                if (left) {
                                    while ((__builtin_amdgcn_readfirstlane(s.e_hash[iw]) >> p.shards->shift_ffm) != ow) iw += (uint32_t)nw;  // (`left` rows of this owner remain)
                                    row_[u] = iw;
                                    h_[u] = __builtin_amdgcn_readfirstlane(s.e_hash[iw]);
                                    pos_[u] = next_pos++;
                                    iw += (uint32_t)nw;
                                    left--;
                                }
                            }
                            auto announce = [&](uint32_t mask_) {  // tag words of the rows written so far: their floats are at the owners
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                                for (int u = 0; u < PU; ++u) {
                                    if (!((mask_ >> u) & 1u) || lane != 0) continue;
                                    const uint32_t slot = pos_[u] & (capf - 1u), gen = (pos_[u] >> lg) & gmask;
                                    __hip_atomic_store(pr.ffm_tag[o_[u]] + slot, (unsigned long long)h_[u] | ((unsigned long long)((gen + 1u) & gmask) << 32), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_SYSTEM);
                                }
                            };
                            uint32_t written = 0, row_polls = 0;
#pragma unroll
                            for (int u = 0; u < PU; ++u) {
                                const uint32_t i = row_[u];
                                if (i == 0xffffffffu) continue;
                                const uint32_t slot = pos_[u] & (capf - 1u), gen = (pos_[u] >> lg) & gmask;
                                // the slot's previous generation has been consumed: the owner says so in this rank's own memory.  A wave never WAITS while it
                                // holds written rows it has not announced (the consumers take a region's stripes in order: two waves waiting for each other's
                                // unannounced rows would wait for ever).
                                for (;;) {
                                    const uint32_t fr = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pr.ffm_free[o_[u]] + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                                    if ((fr & gmask) == gen) break;
                                    if ((++row_polls & 1023u) == 0u && __builtin_amdgcn_readfirstlane(__hip_atomic_load(pr.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0u) break;  // (the host gave the step up)
                                    if (written) {
                                        announce(written);
                                        written = 0;
                                    }
                                    __builtin_amdgcn_s_sleep(8);
                                }
                                written |= 1u << u;
                                const uint32_t f = __builtin_amdgcn_readfirstlane(s.e_fld[i]) & kFldMask;
                                const __amdgpu_buffer_rsrc_t rr = make_rsrc(pr.ffm_rows[o_[u]] + (size_t)slot * R, R * 4);
                                const float v = s.e_val[i];
                                for (uint32_t c = 0; c < nchunk; ++c) {
                                    const uint32_t e0 = (c * 64 + lane) * VEC;
                                    if (e0 >= R) continue;
                                    const uint32_t z = e0 / k;
                                    V tv = Vec<VEC>::lds_load(s.T + f * R + e0), sw = Vec<VEC>::zero(), gv = Vec<VEC>::zero();
                                    const bool self = z == f;
                                    if (self) sw = Vec<VEC>::lds_load(s.selfw + i * k + (e0 - z * k));
#pragma unroll
                                    for (int j = 0; j < VEC; ++j) {
                                        float t_ = Vec<VEC>::get(tv, j);
                                        if (self) t_ = __fsub_rn(t_, __fmul_rn(Vec<VEC>::get(sw, j), v));  // contra - w*v   block_ffm.rs:238
                                        Vec<VEC>::set(gv, j, __fmul_rn(g, __fmul_rn(v, t_)));               // block_ffm.rs:239, 278
                                    }
                                    Vec<VEC>::template store<kAuxSys>(gv, rr, e0 * 4);
                                }
                            }
                            if (written) announce(written);
                        }
                        }  // (owner by owner)
                    }
                } else {
                auto push_lr = [&](uint32_t t) {
                    const uint32_t h = s.l_hash[t], o = h >> p.shards->shift_lr;
                    const uint32_t pos = atomicAdd(&pr.cnt[pr.n + o], 1u);
                    if (pos < pr.cap_lr) {
                        const float grad = g * s.l_val[t];  // block_lr.rs:143
                        __hip_atomic_store(reinterpret_cast<unsigned long long *>(pr.lr_ent[o] + pos), (unsigned long long)h | ((unsigned long long)__float_as_uint(grad) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else {
                        pr.cnt[2 * pr.n] = 1;
                    }
                };
                if (p.has_lr) {
                    if (in_order) {
                        if (tid == 0)
                            for (uint32_t t = 0; t < nl; ++t) push_lr(t);
                    } else {
                        for (uint32_t t = tid; t < nl; t += bd) push_lr(t);
                    }
                }
                if (k && (!in_order || wave == 0)) {
                    for (uint32_t i = in_order ? 0u : (uint32_t)wave; i < nf; i += in_order ? 1u : (uint32_t)nw) {
                        const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i]);
                        const uint32_t f = __builtin_amdgcn_readfirstlane(s.e_fld[i]) & kFldMask;
                        const uint32_t o = h >> p.shards->shift_ffm;
                        uint32_t pos = 0;
                        if (lane == 0) pos = atomicAdd(&pr.cnt[o], 1u);
                        pos = __builtin_amdgcn_readfirstlane(pos);
                        if (pos >= pr.cap_ffm) {
                            if (lane == 0) pr.cnt[2 * pr.n] = 1;
                            continue;
                        }
                        if (lane == 0) __hip_atomic_store(pr.ffm_key[o] + pos, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const __amdgpu_buffer_rsrc_t rr = make_rsrc(pr.ffm_rows[o] + (size_t)pos * R, R * 4);
                        const float v = s.e_val[i];
                        for (uint32_t c = 0; c < nchunk; ++c) {
                            const uint32_t e0 = (c * 64 + lane) * VEC;
                            if (e0 >= R) continue;
                            const uint32_t z = e0 / k;
                            V tv = Vec<VEC>::lds_load(s.T + f * R + e0), sw = Vec<VEC>::zero(), gv = Vec<VEC>::zero();
                            const bool self = z == f;
                            if (self) sw = Vec<VEC>::lds_load(s.selfw + i * k + (e0 - z * k));
#pragma unroll
                            for (int j = 0; j < VEC; ++j) {
                                float t_ = Vec<VEC>::get(tv, j);
                                if (self) t_ = __fsub_rn(t_, __fmul_rn(Vec<VEC>::get(sw, j), v));  // contra - w*v   block_ffm.rs:238
                                Vec<VEC>::set(gv, j, __fmul_rn(g, __fmul_rn(v, t_)));               // block_ffm.rs:239, 278
                            }
                            Vec<VEC>::template store<kAuxSys>(gv, rr, e0 * 4);
                        }
                    }
                }
                }  // (step-synchronous regions)
            } else {
            if (p.has_lr) lr_update<OPT, COH, SH>(p, s, nl, g, gx, lut_lr, tid, bd, PH == 3 ? p.own_lo_lr : 0u, PH == 3 ? p.own_hi_lr : 0xffffffffu);
            FW_TICK(4);
            if (k) {
                // phase A: rows with no earlier overlapping row, all waves, UU rows each
                for (uint32_t i0 = wave * UU; i0 < nf; i0 += nw * UU) {
                    uint32_t idx[UU];
#pragma unroll
                    for (int u = 0; u < UU; ++u) {
                        const uint32_t i = i0 + u;
                        idx[u] = (i < nf && !(s.e_fld[i] & (kRowDep | kRowChained)) && s.e_hash[i] >= olo && s.e_hash[i] < ohi) ? i : 0xffffffffu;
                    }
                    update_rows<VEC, OPT, AUX, UU, SH>(p, s, idx, g, lane, gpair, nf);
                }
                // phase B: overlapping rows, strictly in buffer order on one wave
                if (s.ctr[1]) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // phase A stores acknowledged
                    __syncthreads();
                    if (wave == 0) {
                        for (uint32_t i = 0; i < nf; ++i) {
                            if ((s.e_fld[i] & kRowDep) && s.e_hash[i] >= olo && s.e_hash[i] < ohi) {
                                uint32_t idx[1] = {i};
                                update_rows<VEC, OPT, AUX, 1, SH>(p, s, idx, g, lane, gpair, nf);
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                                __builtin_amdgcn_s_waitcnt(0);
                            }
                        }
                    }
                }
            }
            }  // (not the owner-side-apply push)
            FW_TICK(5);
        }
        // (the tail's lane test starts from a fresh thread index: `tid == 0` from the top of the example would be one more mask alive across the whole body)
        int tid_tail = threadIdx.x;
        asm volatile("; thread index handed out" : "+v"(tid_tail));
        if (tid_tail == 0) {  // published by the loop-top barrier
            // (through a view of its own: the counters' offset is formed here from the argument block, instead of its dozen inputs living in scalar
            // registers from the post-stage view down to this line)
            const KernelParams &pt = kp_fresh();
            Lds st = lds_view(pt, smem, (OPT == FWGPU_OPT_ADAGRAD_LUT) && pt.update && !pt.lut_global);
            st.ctr[6] = next_ticket;
        }
    }
    // (the epilogue takes a view of its own as well: through the prologue's, the argument block's address, the counters' LDS offset and the
    // `tid == 0` mask stayed in scalar registers across the whole example loop)
    const KernelParams &pe = kp_fresh();
    int tid_e = threadIdx.x, bd_e = blockDim.x;
    asm volatile("; thread index and workgroup size handed out" : "+v"(tid_e), "+s"(bd_e));
    if (COH && tid_e == 0) {
        Lds se = lds_view(pe, smem, (OPT == FWGPU_OPT_ADAGRAD_LUT) && pe.update && !pe.lut_global);
        if (se.ctr[13]) hot_lr_flush<SH>(pe, se);  // (every thread's steps are in: the loop ends on a barrier)
    }
    if (SH && PH == 0 && pe.push && pe.push->stream) {
        const KernelParams &p = pe;
        const int tid = tid_e;
        // streaming owner-side apply: this producer workgroup is through.  Its pushes have been acknowledged (rows: before their tag words; LR words: waited
        // for here); the LAST producer workgroup of the launch tells every owner where this source's regions end for this step.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const PushRings &pr = *p.push;
            const uint32_t producers = p.grid_wgs - pr.consumers;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (atomicAdd(pr.done, 1u) + 1u == producers) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                for (uint32_t o = 0; o < pr.n; ++o) {
                    const uint32_t ff = __hip_atomic_load(pr.cnt + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t fl = __hip_atomic_load(pr.cnt + pr.n + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(pr.fin_remote[o] + pr.src, ((unsigned long long)pr.step << 32) | ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pr.fin_remote[o] + pr.n + pr.src, ((unsigned long long)pr.step << 32) | fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
#ifndef FW_KP_NO_CANARY
    if (pe.dbg_canary)
        for (uint32_t i = tid_e; i < 256; i += bd_e)
            if (reinterpret_cast<uint32_t *>(smem + pe.dbg_canary_off)[i] != 0xC0FFEE00u + i) atomicAdd(pe.dbg_canary, 1u);
#endif
#ifdef FW_TICKS
    if (timing)
        for (int i = 0; i < 8; ++i) atomicAdd(p.ticks + i, tk[i]);
#endif
#undef FW_TICK
}

// ------------------------------------------------------------------ launch
// grid == 0: persistent launch.  The grid is what the device can keep RESIDENT (occupancy of this very kernel at this
// workgroup size and LDS size, times the CUs): a larger grid would queue workgroups behind the resident ones and leave the
// device half empty for the tail of the launch (measured: +19 % time with 3 workgroups per CU requested and 2 resident).
template <typename K>
static hipError_t launch_persistent(K kern, const KernelParams &p, uint32_t grid, uint32_t threads, size_t lds,
                                    hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (grid == 0) {
        // the query is a pure function of (kernel, workgroup size, LDS size): remember the last answer per kernel
        static thread_local const void *c_kern = nullptr;  // (all example kernels share this instantiation: same signature)
        static thread_local uint32_t c_threads = 0;
        static thread_local size_t c_lds = 0;
        static thread_local int c_per_cu = 0;
        int per_cu = c_per_cu;
        if (c_kern != reinterpret_cast<const void *>(kern) || c_threads != threads || c_lds != lds || c_per_cu == 0) {
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (int)threads, lds);
            if (e != hipSuccess) return e;
            if (per_cu < 1) per_cu = 1;
            c_kern = reinterpret_cast<const void *>(kern);
            c_threads = threads;
            c_lds = lds;
            c_per_cu = per_cu;
        }
        if (p.host_wgs_cap && (uint32_t)per_cu > p.host_wgs_cap) per_cu = (int)p.host_wgs_cap;
        uint64_t g = (uint64_t)per_cu * (p.host_cus ? p.host_cus : 1);
        // kernels of several ranks resident together on this device: an equal share each, less a quarter (what the occupancy query promises for ONE kernel
        // alone is not what four kernels with their own scratch and wave slots get: a rank whose consumers do not become resident stalls everybody)
        if (p.host_share > 1) g = std::max<uint64_t>(2, g * 3 / (4 * p.host_share));
        grid = (uint32_t)(g < p.n_examples ? g : p.n_examples);
        if (p.host_grid_cap && grid > p.host_grid_cap) grid = p.host_grid_cap;
        if (p.host_extra_wgs) {  // streaming owner-side apply: the consumers come on top of the example workgroups, all of them resident together
            uint32_t extra = p.host_extra_wgs;
            if (extra == 0xffffffffu) {
                // A share of the grid.  The consumers do the larger part of an example's memory work (gradient row in, w and acc read-modify-written: 960 B of
                // traffic against the producers' 384 B per 240-float row): five eighths by default (FWGPU_STREAM_CONSUMER_EIGHTHS, A/B runs).
                static const uint32_t eighths = [] { const char *e = std::getenv("FWGPU_STREAM_CONSUMER_EIGHTHS"); const int v = e ? std::atoi(e) : 5; return (uint32_t)(v < 1 ? 1 : v > 7 ? 7 : v); }();
                extra = std::max<uint32_t>(1u, (uint32_t)(g * eighths / 8));
                const uint32_t waves = threads / 64u;
                if (p.host_stream_max_consumer_waves && extra * waves > p.host_stream_max_consumer_waves) extra = std::max<uint32_t>(1u, p.host_stream_max_consumer_waves / waves);
                // ... and never fewer consumer waves than two per source (one for its LR region, one for its row region: dist.cpp stream_consumer_wgs keeps the same floor
                // for an explicit count) -- a tiny grid's share would leave a source's regions undrained
                if (p.host_stream_min_consumer_waves)
                    while (extra * waves < p.host_stream_min_consumer_waves) extra++;
                static thread_local uint32_t slot[64];
                static thread_local uint32_t next = 0;
                uint32_t *v = &slot[next++ & 63u];
                *v = extra;
                e = hipMemcpyAsync(const_cast<char *>(reinterpret_cast<const char *>(p.push)) + offsetof(PushRings, consumers), v, 4, hipMemcpyHostToDevice, stream);
                if (e != hipSuccess) return e;
            }
            const uint32_t room = g > extra ? (uint32_t)g - extra : 1u;
            grid = std::max<uint32_t>(1u, std::min(grid, room)) + extra;
        }
    }
    KernelParams q = p;
    q.grid_wgs = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, q);
    return hipGetLastError();
}

template <int VEC, int OPT, bool COH>
static hipError_t launch_t(const KernelParams &p, uint32_t grid, uint32_t threads, size_t lds, hipStream_t stream) {
    // (models without a deep head run an instantiation without its code: fewer registers, no spills in the row loops)
    if (p.nn.n_layers) return launch_persistent(fw_example_kernel<VEC, OPT, COH, 0, true>, p, grid, threads, lds, stream);
    if (p.shards) return launch_persistent(fw_example_kernel<VEC, OPT, COH, 0, false, true>, p, grid, threads, lds, stream);  // peer-sharded tables
    return launch_persistent(fw_example_kernel<VEC, OPT, COH, 0, false>, p, grid, threads, lds, stream);
}

template <int VEC>
static hipError_t launch_v(const KernelParams &p, int optimizer, bool coherent, uint32_t grid, uint32_t threads,
                           size_t lds, hipStream_t stream) {
    if (!p.update) return launch_t<VEC, FWGPU_OPT_SGD, false>(p, grid, threads, lds, stream);
    // An updating launch is a coherent one (device-scope accesses), in order or not: every caller passes coherent = update.  The <AdaGrad, non-coherent>
    // instantiations this switch used to name were never launched -- and the deep head's spilled 650 .. 1600 vector registers (profiles/r04_kernel_resource_usage.txt).
    if (!coherent) return hipErrorInvalidValue;
    switch (optimizer) {
    case FWGPU_OPT_SGD: return launch_t<VEC, FWGPU_OPT_SGD, true>(p, grid, threads, lds, stream);
    case FWGPU_OPT_ADAGRAD_FLEX: return launch_t<VEC, FWGPU_OPT_ADAGRAD_FLEX, true>(p, grid, threads, lds, stream);
    default: return launch_t<VEC, FWGPU_OPT_ADAGRAD_LUT, true>(p, grid, threads, lds, stream);
    }
}

#ifdef FW_PHASE_TU
// (This part is the file's SECOND translation unit, built with -DFW_PHASE_TU.  `make PHASE_SGPR_SPILLS=scratch` compiles it with
// -mllvm -amdgpu-spill-sgpr-to-vgpr=0: the phase kernels then keep their spilled scalar registers in scratch memory, not in lanes of a vector
// register.  With the spills in VGPR lanes, FWD / MID launches of two ranks that overlap on different hardware queues give single workgroups
// wrong loop-invariant state -- DESIGN.md 7, profiles/r03_group_concurrency.txt.  The default build keeps the VGPR-lane form, 1.8x faster, and
// dist.cpp orders the ranks of an in-process group on the device.)
// ------------------------------------------------------------------ split pipeline: phase launches and the MID kernel
template <int VEC>
static hipError_t launch_phase_v(const KernelParams &p, int optimizer, int phase, uint32_t grid, uint32_t threads, size_t lds,
                                 hipStream_t stream) {
    if (p.nn.n_layers) {
        if (phase == 1) return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_SGD, false, 1, true>, p, grid, threads, lds, stream);
        switch (optimizer) {
        case FWGPU_OPT_SGD: return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_SGD, true, 3, true>, p, grid, threads, lds, stream);
        case FWGPU_OPT_ADAGRAD_FLEX: return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_ADAGRAD_FLEX, true, 3, true>, p, grid, threads, lds, stream);
        default: return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_ADAGRAD_LUT, true, 3, true>, p, grid, threads, lds, stream);
        }
    }
    if (phase == 1) {
        // debug (scripts/group_repro.py): FWD with device-scope (sc1) table loads instead of cached ones
        static const bool dbg_coh = std::getenv("FWGPU_DBG_FWD_COH") != nullptr;
        if (dbg_coh && VEC == 4) return launch_persistent(fw_example_kernel<4, FWGPU_OPT_SGD, true, 1, false>, p, grid, threads, lds, stream);
        return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_SGD, false, 1, false>, p, grid, threads, lds, stream);
    }
    switch (optimizer) {
    case FWGPU_OPT_SGD: return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_SGD, true, 3, false>, p, grid, threads, lds, stream);
    case FWGPU_OPT_ADAGRAD_FLEX: return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_ADAGRAD_FLEX, true, 3, false>, p, grid, threads, lds, stream);
    default: return launch_persistent(fw_example_kernel<VEC, FWGPU_OPT_ADAGRAD_LUT, true, 3, false>, p, grid, threads, lds, stream);
    }
}

static uint32_t *g_dbg_canary = nullptr;
hipError_t launch_example_phase(const KernelParams &p_in, int optimizer, int phase, uint32_t grid, uint32_t threads, hipStream_t stream) {
    if (p_in.n_examples == 0) return hipSuccess;
    KernelParams p = p_in;
    p.window = 0;  // (the generic kernel's update path)
    p.prefetch = p.tr_lds = p.lut_lds_forced = p.no_selfw = 0;  // (v2-only LDS regions / choices: the phases are the generic kernel, whose gather keeps the entries' own slots)
    p.lds_keep = p.lds_keep_words = 0;
    p.update = phase == 3 ? 1 : 0;
    p.chain = p.update && !p.no_chain;
    {
        // T in the split record, not in LDS (FWGPU_PHASE_T_LDS=1: the staged form, A/B runs).  The workgroup size the caller chose was sized for the
        // FUSED kernel's LDS (1024 threads where that lets one workgroup per CU live): the phases get 512-thread workgroups then, two per CU (256- and
        // 384-thread ones measured slower: profiles/r04_phase_t_in_record.txt).
        static const bool staged = std::getenv("FWGPU_PHASE_T_LDS") != nullptr;
        // ... where the staged form leaves room for ONE workgroup per CU only (k = 16 at 30 fields: T = 57.6 KB).  Where two or three fit anyway (config C),
        // the record's latency in the update costs 3 % and nothing is gained.
        p.t_global = 0;
        const size_t lds_staged = example_kernel_lds_bytes(p, optimizer);
        p.t_global = (!staged && p.k != 0 && p.split != nullptr && 2 * lds_staged > 160 * 1024) ? 1 : 0;
        if (p.t_global && threads == 1024 && (uint64_t)p.max_ffm <= 4ull * 512 && grid != 1) threads = 512;
    }
    size_t lds = example_kernel_lds_bytes(p, optimizer);
    static const bool canary = std::getenv("FWGPU_DBG_LDS_CANARY") != nullptr;
    static uint32_t *d_canary = nullptr;
#ifndef FW_KP_NO_CANARY
    if (canary) {
        if (!d_canary && (hipMalloc((void **)&d_canary, 4) != hipSuccess || hipMemset(d_canary, 0, 4) != hipSuccess)) return hipErrorOutOfMemory;
        g_dbg_canary = d_canary;
        p.dbg_canary = d_canary;
        p.dbg_canary_off = (uint32_t)lds;
        lds += 1024;
    }
#else
    (void)canary;
    (void)d_canary;
#endif
    {
        static const char *pad = std::getenv("FWGPU_DBG_LDS_PAD");  // debug: extra dynamic LDS bytes behind the layout (nothing uses them)
        if (pad) lds += (size_t)atoi(pad);
    }
    if (p.k % 4 == 0 && p.aligned4) return launch_phase_v<4>(p, optimizer, phase, grid, threads, lds, stream);
    return launch_phase_v<1>(p, optimizer, phase, grid, threads, lds, stream);
}
uint32_t dbg_canary_read() {
    uint32_t v = 0;
    if (g_dbg_canary) (void)hipMemcpy(&v, g_dbg_canary, 4, hipMemcpyDeviceToHost);
    return v;
}

// MID: one workgroup per example.  From the (summed) split record: the logit exactly as the fused kernel forms it
// (LR sum + 0.5 * (sum_e T[e] * T[perm(e)] - sum_f dcf[f])), sigmoid / log-loss (block_loss_functions.rs:105-153) -> prediction and
// general gradient; or, with a mini-batched deep head, the head's input x = [LR combo sums, triangle of the pair outputs]
// (block_misc.rs:864-883; a field holding at most one feature has a diagonal of exactly 0, as in the reference's own form).
__global__ void __launch_bounds__(256) split_mid_kernel(const KernelParams p, uint32_t n) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t F = p.F, k = p.k, R = p.R, C = p.split_nlr;
    for (uint32_t ex = blockIdx.x; ex < n; ex += gridDim.x) {
        const float *rec = p.split + (size_t)ex * p.split_len;
        const float *T = rec, *dcf = rec + F * R, *lrs = dcf + F, *cnt = lrs + C;
        const float label = cnt[F], imp = cnt[F + 1];
        if (p.xbuf) {
            float *x = p.xbuf + (size_t)ex * p.nn.X;
            for (uint32_t c = tid; c < C; c += 256) x[c] = lrs[c];
            const uint32_t NT = F * (F + 1) / 2;
            for (uint32_t t = tid; t < NT; t += 256) {
                uint32_t i = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                while (i * (i + 1) / 2 > t) --i;
                const uint32_t j = t - i * (i + 1) / 2;
                float dot = 0.0f;
                for (uint32_t kk = 0; kk < k; ++kk) dot += T[i * R + j * k + kk] * T[j * R + i * k + kk];
                if (i == j) dot = cnt[i] <= 1.0f ? 0.0f : 0.5f * (dot - dcf[i]);
                x[C + t] = dot;
            }
            if (tid == 0) {  // the head kernel needs them next to x
                p.gbuf[2 * (size_t)ex] = label;
                p.gbuf[2 * (size_t)ex + 1] = imp;
            }
            continue;
        }
        float dot = 0.0f;
        for (uint32_t e = tid; e < F * R; e += 256) {
            const uint32_t a = e / R, rem = e - a * R, b = rem / k, kk = rem - b * k;
            dot += T[e] * T[b * R + a * k + kk];
        }
        dot = wave_sum(dot);
        __syncthreads();
        if (lane == 0) red[wave] = dot;
        __syncthreads();
        if (tid == 0) {
            float dot_t = 0.0f, dc_t = 0.0f;
            for (int w = 0; w < 4; ++w) dot_t += red[w];
            for (uint32_t f = 0; f < F; ++f) dc_t += dcf[f];
            float wsum = 0.0f;
            if (p.has_lr) wsum += lrs[0];
            if (k) wsum += 0.5f * (dot_t - dc_t);
            float pr, g;
            if (isnan(wsum)) {
                pr = logistic(0.0f);
                g = 0.0f;
            } else if (wsum < -50.0f) {
                pr = logistic(-50.0f);
                g = 0.0f;
            } else if (wsum > 50.0f) {
                pr = logistic(50.0f);
                g = 0.0f;
            } else {
                pr = logistic(wsum);
                g = -(label - pr) * imp;
            }
            p.pred[ex] = pr;
            p.gbuf[ex] = g;  // importance 0 gives g = 0: the UPD phase then skips the example (regressor.rs:366)
        }
    }
}

hipError_t launch_split_mid(const KernelParams &p, uint32_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(split_mid_kernel, dim3(n < 4096 ? n : 4096), dim3(256), 0, stream, p, n);
    return hipGetLastError();
}

#endif  // FW_PHASE_TU

// ------------------------------------------------------------------ v2: static wave ranges, occupancy-tuned
//
// Same math and same LDS stage / dot / sigmoid / LR code as fw_example_kernel, but the FFM row traffic is organised
// for memory-level parallelism:
//   * every wave owns a CONTIGUOUS range of the example's features, cut at field boundaries and balanced by the rule
//     owner(field) = floor(first_feature_index * n_waves / n_features)  (monotone, so ranges are contiguous); the range is
//     uniform per wave and kept in SGPRs;
//   * the wave issues its row loads in deep batches: the first MAXR rows of its range into a statically indexed register
//     array that STAYS resident through dot/sigmoid (their update then loads only the accumulator rows, UA at a time), the
//     rest four at a time; field sums are accumulated in buffer order (bit-identical to the reference);
//   * rows beyond MAXR, and rows that overlap an earlier row of the same example (rare), are re-read in the update
//     (update_rows, UO rows in flight).
// Rounds 1-2: three workgroups per CU at 80 VGPRs with 0-2 kept rows beat two workgroups at 128 VGPRs with 12: occupancy beat residency
// for SPEED.  Round 3: the kept rows decide the QUALITY of the concurrent mode -- a kept row is written back as w_gather - step, which
// overwrites what other examples did to it during this example's lifetime -- and the hold-out loss falls monotonically with MAXR at
// equal examples/s (0.662 / 0.656 / 0.648 / 0.644 / 0.639 for 0 / 2 / 4 / 6 / 8 kept rows per wave at three workgroups per CU, where the
// registers end at 8: profiles/r03_pareto.txt).  With duplicate-row chains, the placed accumulator table and kept rows that skip the
// re-read of w, TWO workgroups per CU at 128 VGPRs (FW_LB_WAVES_WIN = 4) with FW_MAXR_WIN = 14 kept rows and no spill are now both faster (round 4: 20 kept rows on the slimmed kernel, see FW_UG_WIN)
// and better: 4.79 M examples/s at 0.6355 against 4.71 M at 0.641 (8 kept rows, three workgroups).  The small-table path keeps FW_MAXR = 2
// at six waves per SIMD.
// Only for 16 B-aligned single-chunk rows (k % 4 == 0, R <= 256 floats): BASELINE configs B and C.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

#ifndef FW_UG  // rows in flight per wave in the v2 gather
#define FW_UG 4
#endif
#ifndef FW_UA
#define FW_UA 2
#endif
// ... and in the config-C kernel (WIN, single-chunk rows), whose registers go to the rows kept from the gather: 20 kept rows, overflow rows two at a
// time, accumulator rows one at a time = 128 VGPRs with one spilled; measured (profiles/r04_kept_rows_ab.txt, driver shape / 10 M examples):
//   kept rows (UG, UA)   14 (4, 2)      16 (4, 2)      18 (2, 2)      20 (2, 1)      22 (2, 1)      24 (2, 1)
//   examples/s           5.43-5.46 M    5.45-5.49 M    5.45-5.50 M    5.50-5.53 M    5.41-5.43 M    5.28-5.32 M
//   hold-out after 10 M  0.6299         0.6291         0.6276         0.6233-0.6237  0.6227         0.6224
// The kept rows' update as branch-free straight-line code with the NEXT row's accumulator load issued before this row's stores (round 4's last
// kernel change: 5.80-5.87 M examples/s against 5.63-5.69 M on one box, `roofline.frac` 0.561-0.568, same loss; profiles/r04_pipelined_update_ab.txt.
// The same form for the ~5 overflow rows of a wave -- 5 / 8 / 12 static slots with fresh w and acc -- measured 0.6 % SLOWER and is not kept).
#ifndef FW_PIPE_UPD
#define FW_PIPE_UPD 1
#endif
#ifndef FW_ATOM_COALESCED  // store policy 4: the thinned adds in a layout of their own (256 contiguous bytes per instruction), behind a wave-uniform branch
#define FW_ATOM_COALESCED 1
#endif
// Rows of a wave's range BEYOND the FW_MAXR_WIN register-kept ones whose gather-time w is parked in LDS (Lds::keep) instead of being re-read by the
// update phase: at most this many per wave; the host grants as many as leave two workgroups on a CU (regressor.cpp prepare_launch, KernelParams::lds_keep).
// Config C: 55 KB + 3 x 7.7 KB per workgroup -> 3 rows, 23 of ~25 rows per wave written back as w_gather - step; +0.7 % examples/s and 0.002 of hold-out
// loss after 10 M examples (profiles/r04_parked_rows_ab.txt); a fourth slot that stays empty costs 0.15 %.
#ifndef FW_LDS_KEEP_MAX
#define FW_LDS_KEEP_MAX 3
#endif
#ifndef FW_PARK_DIRECT
#define FW_PARK_DIRECT 1  // the parked rows are loaded with LDS-direct loads in the gather's first burst (0: through registers, two at a time, behind the kept rows)
#endif
#ifndef FW_PIPE_DEPTH
#define FW_PIPE_DEPTH 3  // rows the accumulator loads of the pipelined kept-row update run ahead, at most
#endif
#ifndef FW_UG_WIN
#define FW_UG_WIN 2
#endif
#ifndef FW_UA_WIN
#define FW_UA_WIN 1
#endif
#ifndef FW_WIN_NCH  // 1 KiB chunks per row in the whole-line update.  2 would give the k = 8 rows that span 9 lines their ninth line too: measured slower (3.96 vs 3.86 ms, the extra registers spill)
#define FW_WIN_NCH 1
#endif
#ifndef FW_UO
#define FW_UO 1
#endif
#ifndef FW_LB_THREADS
#define FW_LB_THREADS 512
#endif
// 6 waves per SIMD (<= 85 VGPRs): three 512-thread workgroups per CU.  Occupancy beats residency: 2 resident rows per
// wave at 3 workgroups/CU is 10 % faster (training) / 22 % faster (inference) than 12 resident rows at 2 workgroups/CU.
#ifndef FW_LB_WAVES
#define FW_LB_WAVES 6
#endif
// NC = 16-byte chunks per lane and row: 1 for rows of up to 256 floats (config C: 240), 2 for rows of up to 512 floats (k = 16 with 30
// fields: 480).  Two-chunk rows keep T alone at 57.6 KB of LDS, so two workgroups share a CU and the register budget is 128 VGPRs.
// NN: the deep head (a18) as a phase of the two-chunk instantiation -- config E's CONCURRENT launches: nn_forward between the gather and the sigmoid, nn_backward in front
// of the table update, every FFM pair and every LR combo slot stepping with its own general gradient.  Two 512-thread workgroups per CU where the generic kernel
// (which keeps the in-order launches: the parity mode) runs one of 1024 (round 6).
template <int OPT, bool COH, int MAXR, bool WIN, int NC = 1, int POL = FW_DEFAULT_STORE_POLICY, bool NN = false>
#ifndef FW_LB_WAVES_WIN  // the window path (config C's updating launches): FOUR waves per SIMD = two workgroups per CU, 128 registers -- room for
#define FW_LB_WAVES_WIN 4  // 14 (round 4: 20) kept rows per wave; faster AND better than three workgroups with 8 kept rows
#endif
#ifndef FW_NN_THREADS  // workgroup size / waves per SIMD of the instantiation with the deep head as a phase (config E's concurrent launches)
#define FW_NN_THREADS 512
#endif
#ifndef FW_NN_WAVES
#define FW_NN_WAVES 4
#endif
__global__ void __launch_bounds__(NN ? FW_NN_THREADS : FW_LB_THREADS, NC == 1 ? (WIN ? FW_LB_WAVES_WIN : FW_LB_WAVES) : (NN ? FW_NN_WAVES : 4)) fw_example_kernel_r(const KernelParams /* read through kp_fresh() */) {
    const KernelParams &p = kp_fresh();
    static_assert(NC == 1 || MAXR == 0, "resident rows are a single-chunk feature");
    static_assert(!NN || (NC == 2 && WIN && COH), "the head is a phase of the concurrent two-chunk instantiation only");
    typedef f4 V;
    constexpr int VEC = 4;
    constexpr int AUX = COH ? kAuxSc1 : kAuxPlain;
// cache-policy bits of the write-back stores: 2 = nt, the L2's streaming replacement policy (still write-back: a hint about WHICH line leaves first).
// The weight rows take it: a row an example has just written is not read again by that XCD before it is evicted anyway; measured on the 20-kept-rows
// kernel (profiles/r04_w_nt_and_policy2_on_20_kept_rows.txt): 5.58-5.61 M examples/s against 5.48-5.52 M, hold-out after 10 M examples 0.6233 / 0.6237
// against 0.6226 / 0.6248.  (On the accumulators of policy 2 it changes neither their dirty lifetime nor the loss: profiles/r04c_policy_ab_nt.txt.)
#ifndef FW_WB_AUX_W
#define FW_WB_AUX_W 2
#endif
#ifndef FW_WB_AUX_A
#define FW_WB_AUX_A 0
#endif
    // (nt on the device-scope LOADS was measured and rejected: gather -2 %, accumulator loads -4 %, profiles/r04_nt_loads_and_flush256_ab.txt)
    constexpr int AUX_G = AUX, AUX_LA = AUX;
    constexpr int AUX_SW = (COH && POL < 1) ? kAuxSc1 : (COH ? FW_WB_AUX_W : kAuxPlain);  // weight-row stores (store policy: top of this file; 3 = 1 here)
    constexpr int AUX_SA = (COH && (POL < 2 || POL >= 3)) ? kAuxSc1 : (COH ? FW_WB_AUX_A : kAuxPlain);  // accumulator-row stores
    constexpr bool kThin = COH && POL >= 3;  // thinned accumulator traffic on hot rows (store policies 3 and 4)
    constexpr bool kAtom = COH && POL == 4;  // ... as atomic adds of m g^2 (policy 4) instead of stores of acc_read + m g^2 (policy 3)
    constexpr int UA = (WIN && NC == 1) ? FW_UA_WIN : FW_UA;  // accumulator rows in flight per wave in the update phase
    constexpr int UG = (WIN && NC == 1) ? FW_UG_WIN : FW_UG;  // overflow rows in flight per wave in the gather
#ifndef FW_UO_NN
#define FW_UO_NN FW_UO
#endif
#ifndef FW_UO_NC2  // ... of the headless two-chunk instantiations
#define FW_UO_NC2 FW_UO
#endif
    // overflow rows (w + acc) in flight per wave.  (With the deep head, FW_UO_NN: 2, 4 and 5 rows per round trip were measured against 1 in round 6 -- 1.31-1.34 / 1.26 / 1.28-1.30 M
    // examples/s against 1.33-1.39 M: the update's share of an example's lifetime does not shrink with the rows in flight, profiles/r06_configE_v2_head.txt.)
    constexpr int UO = NN ? FW_UO_NN : (NC == 2 ? FW_UO_NC2 : FW_UO);
    extern __shared__ __align__(16) unsigned char smem[];
    // Single-chunk rows (configs B / C): the AdaGrad LUT is ALWAYS the LDS copy, decided at compile time -- s.lut is then an LDS pointer the
    // compiler can see through (ds_read_b32).  As a run-time choice between the LDS copy and the global table the pointer was generic: every
    // lookup a flat_load, and a flat access makes the wave wait for vmcnt(0) AND lgkmcnt(0) -- i.e. for the acknowledgement of every row
    // store issued before it.  (resolve_row_mode sets lut_lds_forced for these launches so that the host sizes the LDS the same way.)
    constexpr bool kLdsLut = (OPT == FWGPU_OPT_ADAGRAD_LUT) && NC == 1;
    const bool use_lut = kLdsLut || ((OPT == FWGPU_OPT_ADAGRAD_LUT) && p.update && !p.lut_global);
    Lds s;
    SetGeom geom;
    TrLds trl;
    bind_lds(p, smem, use_lut, s, geom, trl);

    const int tid = threadIdx.x, bd = blockDim.x;  // (the prologue's; every phase of the example loop derives its own)
    // The example loop forms the thread index from the wave's index -- one scalar register -- and the lane's number in the wave: threadIdx.x itself is a
    // vector register alive across the whole loop, i.e. (128-register instantiation) spilled, and its reload from scratch at the end of every example was
    // an s_waitcnt vmcnt(0) that drained the example's row stores before the next stage phase could start.
    const uint32_t wave_s = __builtin_amdgcn_readfirstlane((uint32_t)threadIdx.x >> 6);
#define FW_TID_FRESH() ((int)(wave_s * 64u + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))))
    constexpr int kPfMax = 2;  // words of the next record a thread carries through the dot phase (records of up to kPfMax * blockDim words are prefetched)

    if (use_lut)
        for (int i = tid; i < kLutSize; i += bd) s.lut[i] = p.lut_ffm[i];

    // the translator's tables, packed into LDS once per workgroup (record batches; visible to all threads behind the loop-top barrier)
    if (p.records) {
        uint32_t *tw = const_cast<uint32_t *>(trl.pair);
        const DevTranslator &t = p.tr;
        const uint32_t NP = t.n_pairs, NCB = t.n_combos, NM = t.n_members;
        for (uint32_t j = tid; j < NP; j += bd) tw[j] = t.pair_ns[j] | ((uint32_t)t.pair_field[j] << 16) | (t.pair_f32[j] ? 0x80000000u : 0u);
        for (uint32_t c = tid; c <= NCB; c += bd) tw[NP + c] = t.combo_off[c];
        for (uint32_t m = tid; m < NM; m += bd) tw[NP + NCB + 1 + m] = t.combo_ns[m] | (t.combo_f32[m] ? 0x80000000u : 0u);
        for (uint32_t c = tid; c < NCB; c += bd) tw[NP + NCB + 1 + NM + c] = __float_as_uint(t.combo_w[c]);
    }
    // Per-phase shader-clock stamps (fwgpu_debug_phase_ticks) exist in -DFW_TICKS builds only (scripts/perf_probe.py builds its own library): in the
    // shipped kernel the stamps' atomics made the compiler drain every outstanding store at the top of every example (an s_waitcnt vmcnt(0) behind
    // the loop-top barrier), which is exactly the wait the record prefetch below is there to remove.
#ifdef FW_TICKS
    unsigned long long tk_last = 0;
    const bool timing = p.ticks != nullptr && tid == 0;
#define FW_TICK(slot)                                                 \
    if (timing) {                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(p.ticks + (slot), now_ - tk_last);                  \
        tk_last = now_;                                               \
    }
    if (timing) tk_last = __builtin_amdgcn_s_memtime();
#else
#define FW_TICK(slot)
#endif

    if (tid == 0) {
        s.ctr[6] = atomicAdd(p.work, 1u);  // examples come from a device counter: see fw_example_kernel
        hot_lr_init<COH>(p, s, true);
        s.ctr[kCtrPfLen] = 0;
        // write-back policies: this workgroup's turn to write its XCD's L2 back comes every wb_flush_every examples, staggered by workgroup
        const uint32_t every = (COH && POL >= 1 && p.grid_wgs > 1) ? p.wb_flush_every : 0u;
        s.ctr[kCtrWbEvery] = every;
        s.ctr[kCtrWbCount] = every ? blockIdx.x % every : 0u;
    }
    for (;;) {
        // The example's stage phase and the rest of it (gather / dot / update) each take their OWN view of the argument block and of the LDS carve-up:
        // the two dozen LDS offsets, the translator's geometry and the launch flags the stage phase works with die at its end instead of living in scalar
        // registers beside the ones the gather and the update need (the kernel's peak of live scalars is inside the stage phase).
        StageOut so;
        uint32_t ex;
        {
            const KernelParams &p = kp_fresh();  // (shadows the prologue's: this example's loads start here)
            // The thread index is made opaque once per example, and everything that depends on it -- the lane's place in a row, and above all the
            // dozen lane MASKS (tid == 0, tid < n, lane < 2 / 4 / ... of the wave scans) -- is derived inside the loop: left loop-invariant, every
            // such mask is a pair of scalar registers that lives across the whole example loop, i.e. is spilled to a VGPR lane in the prologue
            // and read back where it is used (~30 of the kernel's ~100 spilled scalars).
            int tid_now = FW_TID_FRESH(), bd_now = blockDim.x;
            uint32_t grid_now = p.grid_wgs;
            asm volatile("; thread index, workgroup and grid size handed out" : "+v"(tid_now), "+s"(bd_now), "+s"(grid_now));
            const int tid = tid_now, bd = bd_now;
            // ... and so is the LDS carve-up: two dozen offsets that would otherwise live in scalar registers from the prologue on
            Lds s;
            SetGeom geom;
            TrLds trl;
            bind_lds(p, smem, use_lut, s, geom, trl);
            if (grid_now == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in-order mode: see fw_example_kernel
            __syncthreads();
            ex = s.ctr[6];
            if (ex >= p.n_examples) break;
            FW_TICK(6);
            // The next example's ticket is requested HERE and parked in LDS at the end of the stage phase, with the label and the importance the loss needs:
            // requested behind the stage phase, as until round 5, the returned ticket sat in a vector register until the gather's row loads had been
            // issued -- at the kernel's register peak.
            uint32_t next_ticket = 0;
            if (tid == 0) next_ticket = atomicAdd(p.work, 1u);
#ifdef FW_TICKS
            if (timing) atomicAdd(p.ticks + 7, 1ull);
            so = stage_example<!COH>(p, s, geom, ex, tid, bd, trl, timing ? p.ticks : nullptr, s.ctr[kCtrPfLen]);
#else
            so = stage_example<!COH>(p, s, geom, ex, tid, bd, trl, nullptr, s.ctr[kCtrPfLen]);
#endif
            if (tid == 0) {  // (visible to everybody behind the gather's barrier)
                s.ctr[kCtrNext] = next_ticket;
                s.ctr[kCtrLabel] = __float_as_uint(so.label);
                s.ctr[kCtrImp] = __float_as_uint(so.imp);
            }
        }
        const KernelParams &p = kp_fresh();
        int tid_now = FW_TID_FRESH(), bd_now = blockDim.x;
        uint32_t grid_now = p.grid_wgs;
        asm volatile("; thread index, workgroup and grid size handed out" : "+v"(tid_now), "+s"(bd_now), "+s"(grid_now));
        const int tid = tid_now, lane = tid & 63, wave = tid >> 6, bd = bd_now, nw = bd >> 6;
        const uint32_t F = p.F, k = p.k, R = p.R;
        const float *lut_lr = p.lut_lr;  // 201 lookups per example: read through L1 (an LDS copy measured no faster)
        Lds s = lds_view(p, smem, use_lut);
        // this lane's 4 floats of a row's chunk c: elements [e0, e0+4) = slot z, offset kk0
        uint32_t e0c[NC], zc[NC], kkc[NC];
        bool inbc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            e0c[c] = (c * 64 + lane) * VEC;
            inbc[c] = e0c[c] < R;
            zc[c] = inbc[c] ? e0c[c] / k : 0xfffffffeu;
            kkc[c] = inbc[c] ? e0c[c] - zc[c] * k : 0;
        }
        const uint32_t e0 = e0c[0], z = zc[0];  // (chunk 0: what the resident-row code, NC == 1 only, works on)
        const bool inb = inbc[0];
        const uint32_t nf = __builtin_amdgcn_readfirstlane(so.nf), nl = __builtin_amdgcn_readfirstlane(so.nl);
        FW_TICK(1);

        // ---------------- this wave's feature range [lo, hi): the fields f with floor(fstart[f]*nw/nf) == wave
        uint32_t lo = 0xffffffffu, hi = 0;
        for (uint32_t f = lane; f < F; f += 64) {
            const uint32_t a = s.fstart[f], b = s.fend[f];
            if (b > a && (a * (uint32_t)nw) / nf == (uint32_t)wave) {
                lo = a < lo ? a : lo;
                hi = b > hi ? b : hi;
            }
        }
        lo = wave_min_u32(lo);
        hi = wave_max_u32(hi);
        // uniform by construction; saying so keeps the range and everything indexed by it in SGPRs, which removed the
        // kernel's VGPR spills (measured: 4.53 -> 4.29 ms per launch)
        lo = __builtin_amdgcn_readfirstlane(lo);
        hi = __builtin_amdgcn_readfirstlane(hi);
        const uint32_t cnt = hi > lo ? hi - lo : 0;
        if (cnt == 0) lo = 0;

        // ---------------- gather: all row loads up front, rows stay resident
        // Which rows of the range are kept: the first MAXR -- or (-DFW_KEEP_LAST=1, config-C kernel) the LAST MAXR: the overflow rows then come FIRST in
        // buffer order and are gathered first, eight at a time, in the registers the kept rows do not occupy yet -- one round trip for them instead of
        // one per pair behind the kept rows'.  Measured (profiles/r04_keep_last_ab.txt): +1.2 % examples/s (0.576 against 0.569 of the peak) and a
        // hold-out loss 0.004-0.005 HIGHER from 4 M examples on (0.6297-0.6345 against 0.6244-0.6293 after 10 M, three runs each): the kept rows are
        // then gathered ~10 us later, their w_gather - step overwrites less of what concurrent examples did, and that damping is what the loss rests
        // on.  Not the default.  The field sums are accumulated in buffer order either way.
#ifndef FW_KEEP_LAST
#define FW_KEEP_LAST 0
#endif
#ifndef FW_UG_FIRST
#define FW_UG_FIRST 8
#endif
        constexpr bool kKeepLast = FW_KEEP_LAST && FW_PIPE_UPD && WIN && NC == 1 && MAXR > 0;
        const uint32_t nk = cnt < (uint32_t)MAXR ? cnt : (uint32_t)MAXR;  // rows kept
        constexpr bool kLdsKeep = FW_LDS_KEEP_MAX > 0 && !kKeepLast && FW_PIPE_UPD && WIN && NC == 1 && MAXR > 0;
        constexpr int LKM = kLdsKeep ? FW_LDS_KEEP_MAX : 0;
        const uint32_t lk = kLdsKeep ? p.lds_keep : 0u;                                          // rows of this launch kept in LDS per wave
        const uint32_t nk2 = cnt < (uint32_t)MAXR + lk ? cnt : (uint32_t)MAXR + lk;              // rows kept, registers + LDS
        const uint32_t kb = kKeepLast ? hi - nk : lo;                      // the first of them (cnt == 0: lo == hi == 0)
        V rows[MAXR > 0 ? MAXR : 1];
        // The rows parked in LDS go there DIRECTLY (LDS-direct loads: no register in between), issued in the same burst as the register rows' loads instead
        // of two at a time behind them: destination = wave-uniform slot base + lane * 16.  -DFW_PARK_FIRST=1 issues them in FRONT of the register rows' loads.
#ifndef FW_PARK_ASM
#define FW_PARK_ASM 1  // (default since round 6: +0.5-1.2 % in three of three interleaved pairs, profiles/r06_park_asm_ab.txt; 0 = the compiler-issued LDS-direct loads)
#endif
#ifndef FW_PARK_FIRST
#define FW_PARK_FIRST FW_PARK_ASM
#endif
        auto park_rows = [&]() {
            if (kLdsKeep && FW_PARK_DIRECT) {
#pragma unroll
                for (int j = 0; j < LKM; ++j) {
                    if ((uint32_t)(MAXR + j) < nk2) {
                        const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[lo + MAXR + j]);
                        float *dst = s.keep + ((uint32_t)wave * lk + (uint32_t)j) * R;
                        if (inb) {
#if FW_PARK_ASM
                            // (issued behind the compiler's back: it treats an LDS-direct load as a store to ANY LDS address, and waits for vmcnt(0) -- for all
                            // the row loads of the burst -- before every LDS read of the consume steps below.  The parked rows' own readers wait explicitly.
                            // Issued FIRST, the three loads the compiler does not count are the oldest: its vmcnt(n) waits for the register rows stay exact.)
                            const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)dst);
                            const uint32_t voff = e0 * 4u;
                            const float *sbase = p.ffm_w + h;
                            // (m0 is a reserved register the compiler keeps for its own LDS-direct loads -- the record prefetch -- and does not accept as a clobber: saved and restored)
                            uint32_t m0_saved;
                            if (AUX_G == kAuxSc1)
                                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 sc1\n\ts_mov_b32 m0, %0" : "=&s"(m0_saved) : "s"(lds_base), "v"(voff), "s"(sbase) : "memory");
                            else
                                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(m0_saved) : "s"(lds_base), "v"(voff), "s"(sbase) : "memory");
#else
                            const float *src = p.ffm_w + h + e0;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                             (__attribute__((address_space(3))) void *)dst, 16, 0, AUX_G);
#endif
                        }
                    }
                }
            }
        };
        if (!kKeepLast) {
            if (FW_PARK_FIRST) park_rows();
#pragma unroll
            for (int sl = 0; sl < MAXR; ++sl) {
                // (a slot beyond the range loads through a zero-length descriptor: zeros, no memory access -- as a branch around the load every
                // slot's `sl < nk` became a lane mask that lived from here to the end of the burst, 40 scalar registers for 20 rows)
                const bool on = (uint32_t)sl < nk;
                const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[kb + sl]);  // (beyond the range: some word of the LDS, unused)
                rows[sl] = Vec<VEC>::template load<AUX_G>(make_rsrc(p.ffm_w + h, on ? R * 4 : 0u), e0 * 4);
            }
            if (!FW_PARK_FIRST) park_rows();
        }
        {
            V acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = Vec<VEC>::zero();
            float dc = 0.0f;
            uint32_t cur = 0xffffffffu;
            // consume one row (buffer order; ROW = its NC chunks): field switch -> flush the finished field's sum, transposed, into T
#define FW_CONSUME(ROW, IDX)                                                                                  \
    {                                                                                                         \
        const uint32_t f_ = __builtin_amdgcn_readfirstlane(s.e_fld[(IDX)] & kFldMask);                         \
        const float v_ = s.e_val[(IDX)];                                                                      \
        if (f_ != cur) {                                                                                      \
            if (cur != 0xffffffffu) {                                                                         \
                _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                \
                    if (inbc[c]) Vec<VEC>::lds_store(s.T + zc[c] * R + cur * k + kkc[c], acc[c]);              \
                dc = wave_sum(dc);                                                                            \
                if (!COH && p.ctx_dcf) dc += p.ctx_dcf[cur];                                                  \
                s.dcf[cur] = dc;                                                                              \
            }                                                                                                 \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                                  \
                acc[c] = Vec<VEC>::zero();                                                                    \
                if (!COH && p.ctx_T && inbc[c]) /* context cache: the cached features of the field come first */ \
                    acc[c] = *reinterpret_cast<const f4 *>(p.ctx_T + zc[c] * R + f_ * k + kkc[c]);             \
            }                                                                                                 \
            dc = 0.0f;                                                                                        \
            cur = f_;                                                                                         \
        }                                                                                                     \
        _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                                      \
            float ss_ = 0.0f;                                                                                 \
            _Pragma("unroll") for (int j = 0; j < VEC; ++j) {                                                 \
                const float w_ = (ROW)[c][j];                                                                 \
                acc[c][j] = __fadd_rn(acc[c][j], __fmul_rn(w_, v_)); /* block_ffm.rs:205 */                   \
                ss_ += w_ * w_;                                                                               \
            }                                                                                                 \
            if (zc[c] == f_) {                                                                                \
                dc += ss_ * v_ * v_;                                                                          \
                if ((COH && NC == 1) || !p.no_selfw) Vec<VEC>::lds_store(s.selfw + (IDX)*k + kkc[c], (ROW)[c]);             \
            }                                                                                                 \
        }                                                                                                     \
    }
            if (kKeepLast) {
                // overflow rows [lo, kb) first: transient (they are re-read in the update phase)
                constexpr int UGF = FW_UG_FIRST;
                for (uint32_t i = lo; i < kb; i += UGF) {
                    V r[UGF][NC];
#pragma unroll
                    for (int u = 0; u < UGF; ++u) {
                        r[u][0] = Vec<VEC>::zero();
                        if (i + u < kb) {
                            const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i + u]);
                            r[u][0] = Vec<VEC>::template load<AUX_G>(make_rsrc(p.ffm_w + h, R * 4), e0 * 4);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UGF; ++u)
                        if (i + u < kb) FW_CONSUME(r[u], i + u)
                }
#pragma unroll
                for (int sl = 0; sl < MAXR; ++sl) {
                    rows[sl] = Vec<VEC>::zero();
                    if ((uint32_t)sl < nk) {
                        const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[kb + sl]);
                        rows[sl] = Vec<VEC>::template load<AUX_G>(make_rsrc(p.ffm_w + h, R * 4), e0 * 4);
                    }
                }
            }
            // (the count is handed out afresh: compared with the SAME value as in the burst above, every slot's `sl < nk` is computed once, up there,
            // and kept as a 64-bit lane mask until its row is consumed -- two scalar registers per kept row)
            uint32_t nk_c = __builtin_amdgcn_readfirstlane(nk);
            asm volatile("; kept-row count handed out" : "+s"(nk_c));
#pragma unroll
            for (int sl = 0; sl < MAXR; ++sl)
                if ((uint32_t)sl < nk_c) {
                    V one[NC];
                    one[0] = rows[sl];
                    FW_CONSUME(one, kb + sl)
                }
            if (kLdsKeep && FW_PARK_DIRECT) {
                // rows parked in LDS: their loads were issued right behind the register rows', which have all been consumed by now (vmcnt counts in
                // order: nothing else of this wave is in flight)
                if (nk2 > (uint32_t)MAXR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < LKM; ++j)
                    if ((uint32_t)(MAXR + j) < nk2) {
                        V one[NC];
                        one[0] = Vec<VEC>::zero();
                        if (inb) one[0] = Vec<VEC>::lds_load(s.keep + ((uint32_t)wave * lk + (uint32_t)j) * R + e0);
                        FW_CONSUME(one, lo + MAXR + j)
                    }
            }
            // overflow rows of this range: transient (they are re-read in the update phase)
            for (uint32_t i = kKeepLast ? hi : ((kLdsKeep && FW_PARK_DIRECT) ? lo + nk2 + (nk2 < (uint32_t)MAXR ? (uint32_t)MAXR - nk2 : 0u) : lo + MAXR); i < hi; i += UG) {
                V r[UG][NC];
#pragma unroll
                for (int u = 0; u < UG; ++u) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) r[u][c] = Vec<VEC>::zero();
                    if (i + u < hi) {
                        const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i + u]);
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            r[u][c] = Vec<VEC>::template load<AUX_G>(make_rsrc(p.ffm_w + h, R * 4), e0c[c] * 4);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UG; ++u)
                    if (i + u < hi) {
                        FW_CONSUME(r[u], i + u)
                        // the first lds_keep of them stay in LDS for the update phase (as read HERE, like a register-kept row)
                        if (kLdsKeep && !FW_PARK_DIRECT && inb && i + u - (lo + MAXR) < lk)
                            Vec<VEC>::lds_store(s.keep + ((uint32_t)wave * lk + (i + u - (lo + MAXR))) * R + e0, r[u][0]);
                    }
            }
            if (cur != 0xffffffffu) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (inbc[c]) Vec<VEC>::lds_store(s.T + zc[c] * R + cur * k + kkc[c], acc[c]);
                dc = wave_sum(dc);
                if (!COH && p.ctx_dcf) dc += p.ctx_dcf[cur];
                s.dcf[cur] = dc;
            }
#undef FW_CONSUME
        }
        __syncthreads();
        FW_TICK(2);
        // ---------------- the NEXT example's record: HBM -> registers now, -> LDS after the dot phase.  Its stage phase then needs no memory
        // round trip, so the acknowledgements of this example's row stores are waited for under the next stage phase's LDS work, not before it.
        // (LDS-direct loads: no register carries the words through the dot phase -- with 20 rows kept per wave there is none to spare)
        uint32_t pf_len = 0;
        if (p.records && p.prefetch) {
            const uint32_t nt = s.ctr[kCtrNext];
            if (nt < p.n_examples) {
                const uint64_t r0 = p.rec_off[nt];
                const uint32_t *grec = p.records + r0;
                pf_len = p.rec_self_len ? grec[0] : (uint32_t)(p.rec_off[nt + 1] - r0);
                pf_len = __builtin_amdgcn_readfirstlane(pf_len);
                if (pf_len > (uint32_t)(kPfMax * bd)) pf_len = 0;  // (a longer record is fetched by its own stage phase)
#pragma unroll
                for (int j = 0; j < kPfMax; ++j) {
                    // destination = wave-uniform LDS base + lane * 4: word i of the record lands in rec_next[i]
                    const uint32_t i0 = __builtin_amdgcn_readfirstlane((uint32_t)(wave * 64 + j * bd)), i = i0 + (uint32_t)lane;
                    if (i < pf_len)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(grec + i),
                                                         (__attribute__((address_space(3))) void *)(s.rec_next + i0), 4, 0, 0);
                }
            }
        }
        if (!COH && p.emit_T && ex == 0) {  // setup_cache: keep this example's field sums (Regressor::setup_cache, regressor.rs:409-423)
            for (uint32_t i = tid; i < F * R; i += bd) p.emit_T[i] = s.T[i];
            for (uint32_t f = tid; f < F; f += bd) p.emit_dcf[f] = s.dcf[f];
        }

        float wsum_nn = 0.0f;
        if (NN) {  // x = [LR combo sums | triangle] -> layers -> final neuron (block_neural.rs:196-222); its own barriers, and a view of its own
            const KernelParams &p = kp_fresh();
            int tid_h = FW_TID_FRESH(), bd_h = blockDim.x;
            asm volatile("; thread index and workgroup size handed out" : "+v"(tid_h), "+s"(bd_h));
            const Lds s = lds_view(p, smem, use_lut);
            wsum_nn = nn_forward<VEC, COH>(p, s, __builtin_amdgcn_readfirstlane(so.nl), tid_h, bd_h);
        }
        // ---------------- all-pairs dot from LDS + LR forward (identical to fw_example_kernel)
        float dot = 0.0f;
        if (!NN) {
            const uint32_t nq = F * R / VEC;
            for (uint32_t q = tid; q < nq; q += bd) {
                const uint32_t ee = q * VEC;
                const uint32_t a = ee / R, rem = ee - a * R;
                const uint32_t b = rem / k, kk = rem - b * k;
                const V x = Vec<VEC>::lds_load(s.T + ee);
                const V y = Vec<VEC>::lds_load(s.T + b * R + a * k + kk);
#pragma unroll
                for (int j = 0; j < VEC; ++j) dot += x[j] * y[j];
            }
        }
        float lrs = 0.0f;
        float2 lr_kept = float2{0.0f, 0.0f};  // this thread's first LR entry as the forward pass read it (lr_update `kept`)
        const bool emit_x = !COH && p.emit_x;  // (read-only instantiations only)
        if (!NN && p.has_lr)
            for (uint32_t i = tid; i < nl; i += bd) {
                const float2 wa = lr_forward_pair<COH>(p, s, s.l_hash[i]);
                if (i == (uint32_t)tid) lr_kept = wa;
                lrs += wa.x * s.l_val[i];
                if (emit_x) {  // the head's input keeps one sum per combo slot (block_lr.rs:36-45): the products wait in LDS
                    s.nn[i] = wa.x * s.l_val[i];
                    if (i + 1 < nl && s.l_combo[i] > s.l_combo[i + 1]) s.ctr[14] = 1;  // (zeroed by the stage phase)
                }
            }
        dot = wave_sum(dot);
        lrs = wave_sum(lrs);
        if (lane == 0) {
            s.red[wave] = dot;
            s.red[32 + wave] = lrs;
        }
        if (p.records && p.prefetch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next record has landed in rec_next (nothing else of this wave is in flight here)
            if (tid == 0) s.ctr[kCtrPfLen] = pf_len;
        }
        __syncthreads();
        if (emit_x) {
            // ---------------- deep head, read-only launch: x = [per-combo LR sums | triangle of the FFM pair outputs] (block_misc.rs:864-883, the diagonal in
            // the reference's per-feature form, as nn_forward) goes to the batch's x buffer; the layers run afterwards for all examples at once (head.hip)
            float *x = p.xbuf + (size_t)ex * p.nn.X;
            const uint32_t C = p.num_combos;
            const bool by_combo = s.ctr[14] == 0;
            for (uint32_t c = tid; c < C; c += bd) {
                float acc = 0.0f;
                if (by_combo) {
                    uint32_t lo_ = 0, hi_ = nl;  // first entry with l_combo >= c
                    while (lo_ < hi_) {
                        const uint32_t mid = (lo_ + hi_) >> 1;
                        if (s.l_combo[mid] < c) lo_ = mid + 1;
                        else hi_ = mid;
                    }
                    for (uint32_t i = lo_; i < nl && s.l_combo[i] == c; ++i) acc += s.nn[i];
                } else {
                    for (uint32_t i = 0; i < nl; ++i)
                        if (s.l_combo[i] == c) acc += s.nn[i];
                }
                x[c] = acc;
            }
            const uint32_t NT = F * (F + 1) / 2;
            for (uint32_t t = tid; t < NT; t += bd) {
                uint32_t i = (uint32_t)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                while (i * (i + 1) / 2 > t) --i;
                const uint32_t j = t - i * (i + 1) / 2;
                float d = 0.0f;
                for (uint32_t kk = 0; kk < k; ++kk) d += s.T[i * R + j * k + kk] * s.T[j * R + i * k + kk];
                // the diagonal as split_mid_kernel forms it: exactly 0 for a field of at most one feature, 0.5 (|field sum|^2 - sum of the features' own squares) otherwise
                // (the per-feature form of the training forward needs the entries' own slots in LDS, which a read-only launch does not keep)
                if (i == j) d = (s.fend[i] - s.fstart[i]) <= 1u ? 0.0f : 0.5f * (d - s.dcf[i]);
                x[C + t] = d;
            }
            if (tid == 0) {
                p.gbuf[2 * (size_t)ex] = __uint_as_float(s.ctr[kCtrLabel]);
                p.gbuf[2 * (size_t)ex + 1] = __uint_as_float(s.ctr[kCtrImp]);
                s.ctr[6] = s.ctr[kCtrNext];
            }
            continue;
        }
        float dot_t = 0.0f, dc_t = 0.0f, lr_t = 0.0f;
        for (int w = 0; w < nw; ++w) {
            dot_t += s.red[w];
            lr_t += s.red[32 + w];
        }
        for (uint32_t f = 0; f < F; ++f) dc_t += s.dcf[f];
        float wsum = 0.0f;
        if (p.has_lr) wsum += lr_t;
        wsum += 0.5f * (dot_t - dc_t);
        if (NN) wsum = wsum_nn;  // deep head: the sigmoid sees the final neuron's output (regressor.rs:191-323)

        const float label = __uint_as_float(s.ctr[kCtrLabel]), imp = __uint_as_float(s.ctr[kCtrImp]);  // (parked by the stage phase)
        float pr, g;
        if (isnan(wsum)) {
            pr = logistic(0.0f);
            g = 0.0f;
        } else if (wsum < -50.0f) {
            pr = logistic(-50.0f);
            g = 0.0f;
        } else if (wsum > 50.0f) {
            pr = logistic(50.0f);
            g = 0.0f;
        } else {
            pr = logistic(wsum);
            g = -(label - pr) * imp;
        }
        if (tid == 0) p.pred[ex] = pr;
        FW_TICK(3);

        const bool do_update = p.update && imp != 0.0f;  // regressor.rs:366 (as the stage phase's StageOut::do_update)
        if (do_update && g != 0.0f) {
            const bool lr_upd = p.has_lr;
            const float *gx = nullptr, *gpair = nullptr;
            if (NN) {  // deep head: unwind it first; afterwards every LR combo slot and every FFM pair has its own general gradient (block_neural.rs:252-340, block_misc.rs:822-832)
                {
                    const KernelParams &p = kp_fresh();
                    int tid_h = FW_TID_FRESH(), bd_h = blockDim.x;
                    asm volatile("; thread index and workgroup size handed out" : "+v"(tid_h), "+s"(bd_h));
                    const Lds s = lds_view(p, smem, use_lut);
                    nn_backward<OPT, COH>(p, s, g, tid_h, bd_h);
                }
                gx = nn_buf(p, s).xg;
                gpair = gx + p.num_combos;
            }
            // The pair kept from the forward pass saves the update's load round trip: +1.8 % examples/s at config C at the same loss.  Only for
            // examples of at least FW_LR_KEEP_MIN LR entries: on streams of small examples (10-40 entries, ~15 us per example) the longer
            // read-modify-write window of the hot LR entries costs 0.005-0.01 of hold-out loss (profiles/r04_lr_pair_kept_ab.txt; round 2 saw the same).
#ifndef FW_LR_KEEP_MIN
#define FW_LR_KEEP_MIN 128
#endif
            if (lr_upd) lr_update<OPT, COH>(p, s, nl, g, gx, lut_lr, tid, bd, 0u, 0xffffffffu, !NN && nl >= FW_LR_KEEP_MIN, lr_kept,
                                            (WIN && (kAtom || NC == 2) && p.store_policy == 4 && p.lr_thin && p.grid_wgs > 1) ? ex : 0xffffffffu);  // (the large-table path only)
            FW_TICK(4);
            // phase A, resident rows: w comes from registers (read once, in the gather); only acc is loaded.
            // Chained duplicates (WIN): a row that is chained to an earlier one is applied by that row's owner, and an owner WITH a
            // chain applies it from the window path below (which walks the chain): neither is stepped here.
            constexpr uint32_t kResSkip = WIN ? (kRowDep | kRowChained | kRowHasChain) : kRowDep;
            if (FW_PIPE_UPD && WIN && NC == 1 && MAXR > 0) {
                // Every kept slot goes through the same instructions; a slot without a row (beyond the wave's range, or a row the window path
                // applies) gets descriptors of zero length: its loads return 0, its stores are dropped.  With no branch between them the compiler
                // counts the memory operations exactly (s_waitcnt vmcnt(n) instead of vmcnt(0)): waiting for row sl + 1's accumulators does not
                // wait for the acknowledgement of row sl's stores any more.
                // (the range's first index is handed out afresh: as the gather's own `kb`, every slot's kb + sl stayed in a scalar register from the gather to here)
                uint32_t kb_u = __builtin_amdgcn_readfirstlane(kb);
                // (... and so is the per-wave count of rows parked in LDS: with the gather's own `lk` this lane's address in the parked rows' slots is computed
                // once, up there, and kept in a vector register through the dot phase -- spilled in this 128-register kernel, and its reload from scratch HERE
                // is an s_waitcnt vmcnt(0): every parked row's step then waited for the acknowledgement of every store of the rows before it)
                uint32_t lk_u = __builtin_amdgcn_readfirstlane(lk);
                asm volatile("; first kept row and parked-row count handed out" : "+s"(kb_u), "+s"(lk_u));
                auto slot = [&](int sl, uint32_t &h, uint32_t &f, bool &ok) {
                    const uint32_t i = kb_u + (uint32_t)sl;
                    const uint32_t fb = __builtin_amdgcn_readfirstlane(s.e_fld[i]);
                    ok = (uint32_t)sl < nk2 && !(fb & kResSkip);
                    h = ok ? __builtin_amdgcn_readfirstlane(s.e_hash[i]) : 0u;
                    f = ok ? (fb & kFldMask) : 0u;
                };
                // The accumulator loads run AHEAD of the row being stepped by a distance that grows as the kept rows' registers are released:
                // 1 row at slot 0 (20 rows + 2 accumulator rows alive), one more per slot up to FW_PIPE_DEPTH -- the register peak stays where it was.
                // vmcnt counts in issue order, so waiting for row sl's accumulators also waits for every store issued before their load: at distance
                // 1 that is row sl - 2's write-through accumulator store, at distance D row sl - 1 - D's.
                constexpr int NS = MAXR + LKM, DMAX = FW_PIPE_DEPTH;
                V av[NS > 0 ? NS : 1];
                auto issue = [&](int j) {
                    uint32_t h, f;
                    bool ok;
                    slot(j, h, f, ok);
                    av[j] = OPT != FWGPU_OPT_SGD ? Vec<VEC>::template load<AUX_LA>(make_rsrc(p.ffm_acc + h, ok ? R * 4 : 0), e0 * 4) : Vec<VEC>::zero();
                };
                uint32_t add_mask = 0;  // policy 4: the slots whose hot row this example adds its (m-fold) g^2 to
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int d_prev = sl < DMAX ? sl : DMAX, d_now = sl + 1 < DMAX ? sl + 1 : DMAX;
                    const int first = sl == 0 ? 0 : sl - 1 + d_prev + 1, last = sl + d_now;
#pragma unroll
                    for (int j = first; j <= last; ++j)
                        if (j < NS) issue(j);
                    uint32_t h0, f0;
                    bool ok0;
                    slot(sl, h0, f0, ok0);
                    V a_cur = av[sl];
                    // store policy 3: is this a hot row (wave-uniform: any lane's accumulator beyond theta), and is this example the one in m that stores it?
                    float g2_scale = 0.0f;
                    bool acc_store = ok0, acc_add = false, hot_row = false;
                    if (kThin && (kAtom || sl < MAXR) && p.grid_wgs > 1) {  // (policy 3: rows kept in registers only: a thinned STORE on a row parked in LDS -- stepped last, behind the longest window -- loses its race more often, tests/test_gpu_conservation.py; an atomic add has no race to lose)
                        const bool hot = __ballot(a_cur[0] > p.acc_hot_theta || a_cur[1] > p.acc_hot_theta || a_cur[2] > p.acc_hot_theta || a_cur[3] > p.acc_hot_theta) != 0ull;
                        if (hot) {
                            const uint32_t m = 1u << p.acc_sample_log2;
                            const uint32_t draw = ((ex * 2654435761u) ^ ((kb_u + (uint32_t)sl) * 40503u + (uint32_t)wave * 9973u)) >> 9;
                            const bool turn = ok0 && (draw & (m - 1u)) == 0u;
                            acc_store = kAtom ? false : turn;
                            acc_add = kAtom && turn;
                            g2_scale = kAtom ? (float)m : (float)(m - 1u);
                            hot_row = true;
                        }
                    }
                    const float v = s.e_val[kb_u + (uint32_t)sl];
                    V wv;
                    if (sl < MAXR) {
                        wv = rows[sl < MAXR ? sl : 0];
                    } else {  // a row parked in LDS by the gather
                        wv = Vec<VEC>::zero();
                        if (inb && ok0) wv = Vec<VEC>::lds_load(s.keep + ((uint32_t)wave * lk_u + (uint32_t)(sl - MAXR)) * R + e0);
                    }
                    V tv = Vec<VEC>::zero();
                    if (inb) tv = Vec<VEC>::lds_load(s.T + f0 * R + e0);
                    const bool self = (z == f0);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        float t = tv[j];
                        if (self) t = __fsub_rn(t, __fmul_rn(wv[j], v));  // contra - w*v  block_ffm.rs:238
                        const float G = __fmul_rn(v, t);
                        const float grad = __fmul_rn(g, G);
                        float acc = a_cur[j];
                        const float upd = opt_step<OPT>(grad, acc, p.ffm_rate, p.ffm_minus_power_t, s.lut);
                        // (what is STORED for a thinned row: m x this example's g^2 on top of what it read; policy 4, strided form: what is ADDED to a hot row, m x this example's g^2)
                        a_cur[j] = kAtom ? ((!FW_ATOM_COALESCED && hot_row) ? g2_scale * (grad * grad) : acc) : (kThin ? acc + g2_scale * (grad * grad) : acc);
                        wv[j] = wv[j] - upd;  // block_ffm.rs:282
                    }
                    Vec<VEC>::template store<AUX_SW>(wv, make_rsrc(p.ffm_w + h0, ok0 ? R * 4 : 0), e0 * 4);
                    if (OPT != FWGPU_OPT_SGD) Vec<VEC>::template store<AUX_SA>(a_cur, make_rsrc(p.ffm_acc + h0, (kThin ? acc_store : ok0) ? R * 4 : 0), e0 * 4);
                    if (kAtom && OPT != FWGPU_OPT_SGD) {
#if FW_ATOM_COALESCED
                        add_mask |= (acc_add ? 1u : 0u) << sl;  // (the adds themselves: behind the loop, where no row is alive in registers any more)
#else
                        // fire-and-forget (no return value: nothing waits for them); a row that is not hot, or not this example's turn, adds through a zero-length descriptor: dropped
                        const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.ffm_acc + h0, acc_add ? R * 4 : 0);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(a_cur[j], ra, (int)(e0 * 4 + 4 * j), 0, kAuxSc1);
#endif
                    }
                }
                if (kAtom && FW_ATOM_COALESCED && OPT != FWGPU_OPT_SGD) {
                    // Store policy 4's adds: fire-and-forget (no return value: nothing waits for them) and COALESCED -- lane l adds to floats l, 64 + l, 128 + l,
                    // 192 + l of the row, so one instruction covers 256 contiguous bytes = four 64-byte requests of 16 floats each (in the step's own layout, 4
                    // consecutive floats per lane, an instruction touches sixteen 64-byte segments with 4 floats each: four times the requests at the memory
                    // side; measured in round 6, profiles/r06_store_policy4_*.txt).  The gradient is formed again in that layout from what the step used: T and
                    // the entry's own slot as the gather read it, both in LDS.  A plain loop over the few slots whose turn it is: no row is alive in registers here.
                    uint32_t am = __builtin_amdgcn_readfirstlane(add_mask);
                    const uint32_t ksh = p.k_log2;
                    const float mf = (float)(1u << p.acc_sample_log2);
                    while (am) {
                        const uint32_t sl = (uint32_t)__builtin_ctz(am);
                        am &= am - 1u;
                        const uint32_t i = kb_u + sl;
                        const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i]);
                        const uint32_t f = __builtin_amdgcn_readfirstlane(s.e_fld[i]) & kFldMask;
                        const float v = s.e_val[i];
                        const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.ffm_acc + h, R * 4);  // (floats beyond the row: dropped by the descriptor)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t e = (uint32_t)lane + 64u * (uint32_t)j;
                            const uint32_t ec = e < R ? e : R - 1u;
                            float t = s.T[f * R + ec];
                            const float sw = s.selfw[i * k + (ec & (k - 1u))];
                            t = ((ec >> ksh) == f) ? __fsub_rn(t, __fmul_rn(sw, v)) : t;  // contra - w*v  block_ffm.rs:238
                            const float grad = __fmul_rn(g, __fmul_rn(v, t));
                            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(mf * (grad * grad), ra, (int)(e * 4u), 0, kAuxSc1);
                        }
                    }
                }
            } else
#pragma unroll
            for (int g0 = 0; g0 < MAXR; g0 += UA) {
                if ((uint32_t)g0 < nk) {
                    V av[UA];
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        av[u] = Vec<VEC>::zero();
                        if (OPT != FWGPU_OPT_SGD && (uint32_t)(g0 + u) < nk) {
                            const uint32_t i = kb + g0 + u;
                            if (!(s.e_fld[i] & kResSkip)) {
                                const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i]);
                                av[u] = Vec<VEC>::template load<AUX_LA>(make_rsrc(p.ffm_acc + h, R * 4), e0 * 4);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        if (g0 + u < MAXR && (uint32_t)(g0 + u) < nk) {
                            const uint32_t i = kb + g0 + u;
                            const uint32_t fb = s.e_fld[i];
                            if (!(fb & kResSkip)) {
                                const uint32_t f = __builtin_amdgcn_readfirstlane(fb & kFldMask);
                                const uint32_t h = __builtin_amdgcn_readfirstlane(s.e_hash[i]);
                                const float v = s.e_val[i];
                                V wv = rows[(g0 + u) < MAXR ? (g0 + u) : 0];
                                V tv = Vec<VEC>::zero();
                                if (inb) tv = Vec<VEC>::lds_load(s.T + f * R + e0);
                                const bool self = (z == f);
#pragma unroll
                                for (int j = 0; j < VEC; ++j) {
                                    float t = tv[j];
                                    if (self) t = __fsub_rn(t, __fmul_rn(wv[j], v));  // contra - w*v  block_ffm.rs:238
                                    const float G = __fmul_rn(v, t);
                                    const float grad = __fmul_rn(g, G);
                                    float acc = av[u][j];
                                    const float upd = opt_step<OPT>(grad, acc, p.ffm_rate, p.ffm_minus_power_t, s.lut);
                                    av[u][j] = acc;
                                    wv[j] = wv[j] - upd;  // block_ffm.rs:282
                                }
                                Vec<VEC>::template store<AUX_SW>(wv, make_rsrc(p.ffm_w + h, R * 4), e0 * 4);
                                if (OPT != FWGPU_OPT_SGD)
                                    Vec<VEC>::template store<AUX_SA>(av[u], make_rsrc(p.ffm_acc + h, R * 4), e0 * 4);
                            }
                        }
                    }
                }
            }
            // phase A, overflow rows of this range: the v1 route (fresh read of w)
            for (uint32_t i0 = (WIN && MAXR > 0) ? lo : lo + MAXR; i0 < hi; i0 += UO) {
                uint32_t idx[UO];
#pragma unroll
                for (int u = 0; u < UO; ++u) {
                    const uint32_t i = i0 + u;
                    idx[u] = (i < hi && !(s.e_fld[i] & (kRowDep | kRowChained))) ? i : 0xffffffffu;
                    // (the first MAXR rows of the range were stepped from registers, unless they own a chain)
                    if (WIN && MAXR > 0 && i >= kb && i < kb + nk2 && !(s.e_fld[i] & kRowHasChain)) idx[u] = 0xffffffffu;
                }
                if (WIN)
                    update_rows_win<OPT, AUX, UO, (NC > FW_WIN_NCH ? NC : FW_WIN_NCH), AUX_SW, AUX_SA>(p, s, idx, g, lane, nf, gpair, (kThin && (NC > 1 || kAtom) && p.store_policy >= 3 && p.grid_wgs > 1 && p.thin_reread) ? ex : 0xffffffffu);
                else
                    update_rows<VEC, OPT, AUX, UO, false, NC>(p, s, idx, g, lane);
            }
            // phase B: rows overlapping an earlier row of this example, strictly in buffer order on one wave
            if (s.ctr[1]) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (wave == 0) {
                    for (uint32_t i = 0; i < nf; ++i) {
                        if (s.e_fld[i] & kRowDep) {
                            uint32_t idx[1] = {i};
                            if (WIN)
                                update_rows_win<OPT, AUX, 1, (NC > FW_WIN_NCH ? NC : FW_WIN_NCH), AUX_SW, AUX_SA>(p, s, idx, g, lane, nf, gpair);
                            else
                                update_rows<VEC, OPT, AUX, 1, false, NC>(p, s, idx, g, lane);
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                            __builtin_amdgcn_s_waitcnt(0);
                        }
                    }
                }
            }
            FW_TICK(5);
        }
        // (the tail's lane tests start from a fresh thread index: `tid == 0` from the top of the example would be one more mask alive across the whole body)
        int tid_tail = FW_TID_FRESH();
        asm volatile("; thread index handed out" : "+v"(tid_tail));
        if (tid_tail == 0) s.ctr[6] = s.ctr[kCtrNext];  // published by the loop-top barrier
        if (COH && POL >= 1 && tid_tail == bd - 1) {
            // bounded staleness of the write-back policies: this workgroup's turn to write its XCD's dirty L2 lines back (one instruction, not
            // waited for here: it completes under the next example's stage phase)
            const uint32_t every = s.ctr[kCtrWbEvery];
            if (every) {
                uint32_t c = s.ctr[kCtrWbCount] + 1;
                if (c >= every) {
                    c = 0;
                    asm volatile("buffer_wbl2 sc1" ::: "memory");
                }
                s.ctr[kCtrWbCount] = c;
            }
        }
    }
    {
        int tid_end = FW_TID_FRESH();
        asm volatile("; thread index handed out" : "+v"(tid_end));
        if (COH && tid_end == 0) {
            // (a view of its own: through the prologue's, the counters' LDS offset stayed in a scalar register across the whole example loop)
            const KernelParams &p = kp_fresh();
            Lds s = lds_view(p, smem, use_lut);
            if (s.ctr[13]) hot_lr_flush(p, s);  // (every thread's steps are in: the loop ends on a barrier)
        }
    }
#undef FW_TICK
#undef FW_TID_FRESH
}

#ifndef FW_MAXR
#define FW_MAXR 2
#endif
#ifndef FW_MAXR_WIN
#define FW_MAXR_WIN 20
#endif
#ifndef FW_PHASE_TU  // (the fused kernels' launchers and the small utility kernels: first translation unit only)
template <int OPT, bool COH>
static hipError_t launch_r(const KernelParams &p, uint32_t grid, uint32_t threads, size_t lds, hipStream_t stream) {
    if (p.R > 64 * 4) {  // two-chunk rows (k = 16 at config E's 30 fields): no resident rows
        if (COH && p.nn_v2) return launch_persistent(fw_example_kernel_r<OPT, COH, 0, true, 2, FW_DEFAULT_STORE_POLICY, COH>, p, grid, threads, lds, stream);  // ... with the deep head as a phase
        if (p.window) return launch_persistent(fw_example_kernel_r<OPT, COH, 0, true, 2>, p, grid, threads, lds, stream);
        return launch_persistent(fw_example_kernel_r<OPT, COH, 0, false, 2>, p, grid, threads, lds, stream);
    }
    if (p.window) {
        // the config-C kernel: its store policy is a launch parameter (instantiations of their own, so that the shipped policy pays nothing for the others)
        if (COH && p.store_policy == 0) return launch_persistent(fw_example_kernel_r<OPT, COH, FW_MAXR_WIN, true, 1, COH ? 0 : FW_DEFAULT_STORE_POLICY>, p, grid, threads, lds, stream);
        if (COH && p.store_policy == 2) return launch_persistent(fw_example_kernel_r<OPT, COH, FW_MAXR_WIN, true, 1, COH ? 2 : FW_DEFAULT_STORE_POLICY>, p, grid, threads, lds, stream);
        if (COH && p.store_policy == 3) return launch_persistent(fw_example_kernel_r<OPT, COH, FW_MAXR_WIN, true, 1, COH ? 3 : FW_DEFAULT_STORE_POLICY>, p, grid, threads, lds, stream);
        // (no rows kept from the gather -- every row re-read by the update: the concurrent mode without last-writer-wins over an example's lifetime; option 13, policy 4 only)
        if (COH && p.store_policy == 4 && p.no_kept_rows) return launch_persistent(fw_example_kernel_r<OPT, COH, 0, true, 1, COH ? 4 : FW_DEFAULT_STORE_POLICY>, p, grid, threads, lds, stream);
        if (COH && p.store_policy == 4) return launch_persistent(fw_example_kernel_r<OPT, COH, FW_MAXR_WIN, true, 1, COH ? 4 : FW_DEFAULT_STORE_POLICY>, p, grid, threads, lds, stream);
        return launch_persistent(fw_example_kernel_r<OPT, COH, FW_MAXR_WIN, true, 1, COH ? 1 : FW_DEFAULT_STORE_POLICY>, p, grid, threads, lds, stream);
    }
    return launch_persistent(fw_example_kernel_r<OPT, COH, FW_MAXR, false>, p, grid, threads, lds, stream);
}

static hipError_t launch_resident(const KernelParams &p, int optimizer, bool coherent, uint32_t grid, uint32_t threads,
                                  size_t lds, hipStream_t stream) {
    if (!p.update) return launch_r<FWGPU_OPT_SGD, false>(p, grid, threads, lds, stream);
    if (!coherent) return hipErrorInvalidValue;  // (an updating launch is a coherent one: see launch_v)
    switch (optimizer) {
    case FWGPU_OPT_SGD: return launch_r<FWGPU_OPT_SGD, true>(p, grid, threads, lds, stream);
    case FWGPU_OPT_ADAGRAD_FLEX: return launch_r<FWGPU_OPT_ADAGRAD_FLEX, true>(p, grid, threads, lds, stream);
    default: return launch_r<FWGPU_OPT_ADAGRAD_LUT, true>(p, grid, threads, lds, stream);
    }
}

// Does this launch run on the register-resident kernel (v2)?  16 B-aligned single-chunk rows, no deep head.
static bool uses_resident_kernel(const KernelParams &p, uint32_t threads) {
    // rows of up to 256 floats (one 16-byte chunk per lane) or up to 512 (two chunks; a field slot must not straddle the chunks)
    const bool fits = p.R <= 64 * 4 || (p.R <= 64 * 4 * 2 && p.k != 0 && 256 % p.k == 0);
    // a deep head: read-only launches that only form its input (emit_x), and -- nn_v2, chosen by prepare_launch where two workgroups fit a CU -- config E's concurrent updating launches
    const bool head_ok = p.nn.n_layers == 0 || (p.emit_x && !p.update) || (p.nn_v2 && p.update && p.R > 64 * 4);
    return p.k % 4 == 0 && p.aligned4 && fits && p.kernel_version != 1 && head_ok && threads <= ((p.nn_v2 && p.update) ? (FW_NN_THREADS > FW_LB_THREADS ? FW_NN_THREADS : FW_LB_THREADS) : FW_LB_THREADS);
}
bool example_kernel_is_resident(const KernelParams &p, uint32_t threads) { return uses_resident_kernel(p, threads); }
uint32_t nn_v2_threads() { return FW_NN_THREADS; }
// Whole-line updates and duplicate-row chains exist in the v2 kernel's update path only, and only updating launches need them.
void resolve_row_mode(KernelParams &p, uint32_t threads) {
    p.window = (p.window && uses_resident_kernel(p, threads) && p.update && p.k_log2 != 0xffu) ? 1 : 0;
    if (p.store_policy < 0) p.store_policy = FW_DEFAULT_STORE_POLICY;  // (-1: the build's default; the host side does not know the -D flags of a variant build)
    if (p.wb_flush_every == 0xffffffffu) p.wb_flush_every = FW_DEFAULT_WB_FLUSH_EVERY;
    // chained duplicate rows: the generic kernel's update path and the whole-line path apply them from registers; the
    // register-resident variant of the v2 kernel (window off) keeps the old route (duplicates serialised in phase B)
    p.chain = (p.update && !p.no_chain && (p.window || !uses_resident_kernel(p, threads))) ? 1 : 0;
    // the next record is prefetched by the v2 kernel's updating launches (read-only launches have no store drain to hide, and their LDS is what lets
    // three workgroups share a CU)
    p.prefetch = (p.prefetch && p.records && p.update && uses_resident_kernel(p, threads)) ? 1 : 0;
    p.tr_lds = (p.records && uses_resident_kernel(p, threads)) ? 1 : 0;  // the v2 kernel reads the translator's tables from an LDS copy
    // The v2 kernel's single-chunk updating instantiations keep the LUT in LDS as a compile-time fact (kLdsLut): settled HERE, so that the host's
    // LDS size -- the occupancy choice and the 160 KiB check of prepare_launch -- is the size the launch really uses (debug option 1 does not apply to them).
    p.lut_lds_forced = (uses_resident_kernel(p, threads) && p.R <= 64 * 4 && p.update) ? 1 : 0;
    // The entries' own slots (Lds::selfw: every row's w[f_i * k ..] as the gather read it) are read by the update and by the generic kernel's head only.  Read-only launches of the
    // v2 kernel do without them -- and so do its CONCURRENT whole-line updates (update_rows_win): the row the update has just re-read holds the same slot, as fresh as the rest of
    // the row (between concurrent examples either is a value the weight had; in-order launches keep the LDS copy, which is what makes rows that overlap an earlier row of the
    // example exact).  At k = 16 that is 16-19 KB of 89: what lets a SECOND workgroup live on a CU (+27 % examples/s where it fits, profiles/r05_k16_two_wgs_probe.txt).
    static const bool selfw_lds_forced = [] { const char *e = std::getenv("FWGPU_SELFW_LDS"); return e && std::atoi(e) != 0; }();
    // (two-chunk rows only: config C's kernel has the LDS to spare, and the choice costs its update two vector registers it does not have)
    p.no_selfw = (uses_resident_kernel(p, threads) && (!p.update || (p.window && p.concurrent && p.R > 64 * 4 && !selfw_lds_forced))) ? 1 : 0;
    // rows kept in LDS beyond the register-kept ones: the chained-update instantiation of single-chunk rows only (fw_example_kernel_r, FW_LDS_KEEP_MAX)
    if (!(p.lut_lds_forced && p.window) || FW_KEEP_LAST || p.no_kept_rows) p.lds_keep = 0;
    if (p.lds_keep > FW_LDS_KEEP_MAX) p.lds_keep = FW_LDS_KEEP_MAX;
    p.lds_keep_words = p.lds_keep * (threads / 64) * p.R;
}

hipError_t launch_example_kernel(const KernelParams &p_in, int optimizer, bool coherent, uint32_t grid, uint32_t threads,
                                 hipStream_t stream) {
    if (p_in.n_examples == 0 && !p_in.host_extra_wgs) return hipSuccess;
    KernelParams p = p_in;
    const bool v2 = uses_resident_kernel(p, threads);
    resolve_row_mode(p, threads);  // (idempotent: run_batch has normally done it already, to size the LDS)
    (void)v2;
    const size_t lds = example_kernel_lds_bytes(p, optimizer);
    // 16-byte row vectors need k % 4 == 0: then R % 4 == 0 and hash & mask is a multiple of next_pow2(k) >= 4
    // floats (feature_buffer.rs:141-148), so every row starts 16-byte aligned.
    // Entries that did not come through the translator's mask (raw fwgpu_learn calls) may be unaligned.
    if (p.k % 4 == 0 && p.aligned4) {
        // single-chunk rows: the register-resident kernel (v2); p.kernel_version == 1 forces v1 (tests, A/B runs)
        if (v2) return launch_resident(p, optimizer, coherent, grid, threads, lds, stream);
        return launch_v<4>(p, optimizer, coherent, grid, threads, lds, stream);
    }
    return launch_v<1>(p, optimizer, coherent, grid, threads, lds, stream);
}

// ------------------------------------------------------------------ owner-side apply (dist.cpp fwgpu_dist_*_owner)
// The owner's half of the step: one wave per pushed gradient row runs the optimizer on the owner's OWN tables (acc += grad^2; w -= step,
// optimizer.rs), one thread per pushed LR gradient likewise.  in_order (one wave): everything in the order it was pushed -- with one source
// pushing one example at a time that is the reference's sequence of updates.  Concurrently the steps race like any hogwild step, but only
// inside the owner: no read-modify-write ever crosses a link, and the accumulators never leave their GPU.
template <int OPT>
__global__ void __launch_bounds__(256) owner_apply_kernel(float *w, float *acc, float *lr, const uint32_t *keys, const float *rows, uint32_t n_rows,
                                                          const uint2 *lr_ent, uint32_t n_lr, uint32_t R, float ffm_rate, float ffm_mpt, const float *lut_ffm,
                                                          float lr_rate, float lr_mpt, const float *lut_lr, int in_order) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = gw; i < n_rows; i += nw) {
        const uint32_t h = __builtin_amdgcn_readfirstlane(__hip_atomic_load(keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (((R | h) & 3u) == 0) {  // 16-byte rows (k % 4 == 0): one 16 B access per lane and table, like the fused kernel's
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(rows + (size_t)i * R, R * 4), rw = make_rsrc(w + h, R * 4), ra = make_rsrc(acc + h, R * 4);
            for (uint32_t e0 = lane * 4; e0 < R; e0 += 256) {
                const f4 gv = Vec<4>::load<kAuxSys>(rg, e0 * 4);
                f4 wv = Vec<4>::load<kAuxSc1>(rw, e0 * 4), av = OPT == FWGPU_OPT_SGD ? Vec<4>::zero() : Vec<4>::load<kAuxSc1>(ra, e0 * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = av[c];
                    wv[c] = wv[c] - opt_step<OPT>(gv[c], a, ffm_rate, ffm_mpt, lut_ffm);  // block_ffm.rs:279-282
                    av[c] = a;
                }
                Vec<4>::store<kAuxSc1>(wv, rw, e0 * 4);
                if (OPT != FWGPU_OPT_SGD) Vec<4>::store<kAuxSc1>(av, ra, e0 * 4);
            }
        } else
        for (uint32_t e = lane; e < R; e += 64) {
            const float grad = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(rows + (size_t)i * R + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            float a = OPT == FWGPU_OPT_SGD ? 0.0f : __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(acc + h + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            float wv = __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(w + h + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            wv -= opt_step<OPT>(grad, a, ffm_rate, ffm_mpt, lut_ffm);  // block_ffm.rs:279-282
            __hip_atomic_store(reinterpret_cast<unsigned *>(w + h + e), __float_as_uint(wv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (OPT != FWGPU_OPT_SGD) __hip_atomic_store(reinterpret_cast<unsigned *>(acc + h + e), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (in_order) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next row of the ring may be the same row, or overlap it
    }
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (uint32_t j = in_order ? 0u : gt; j < n_lr; j += in_order ? 1u : nt) {
        if (in_order && gt != 0) break;
        const unsigned long long ent = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(lr_ent + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t h = (uint32_t)ent;
        const float grad = __uint_as_float((uint32_t)(ent >> 32));
        float2 wa = lr_load<true>(lr, h);  // ({w, acc} pairs on the device whatever the optimizer)
        wa.x -= opt_step<OPT>(grad, wa.y, lr_rate, lr_mpt, lut_lr);  // block_lr.rs:145-147
        lr_store<true>(lr, h, wa);
    }
}

// Can kernels launched on these streams RUN AT THE SAME TIME?  One single-wave kernel per stream: count yourself, then wait (bounded: tens of milliseconds) until all n have
// counted.  Streams that share a hardware queue run their kernels one after the other: the first one's wait runs out.  (dist.cpp: the streaming
// owner-side apply of an in-process group needs this of the ranks that share a device, and nothing the library can set guarantees it.)
__global__ void rendezvous_kernel(uint32_t *counter, uint32_t n, uint32_t *met) {
    if (threadIdx.x != 0) return;
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (;;) {
        if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= n) {
            __hip_atomic_fetch_add(met, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (__builtin_amdgcn_s_memtime() - t0 > 40000000ull) return;  // (20 ms at the shader clock, 0.4 s should s_memtime count at 100 MHz)
        __builtin_amdgcn_s_sleep(64);
    }
}
hipError_t launch_rendezvous(uint32_t *counter, uint32_t n, uint32_t *met, hipStream_t stream) {
    hipLaunchKernelGGL(rendezvous_kernel, dim3(1), dim3(64), 0, stream, counter, n, met);
    return hipGetLastError();
}

hipError_t launch_owner_apply(const fwgpu_regressor *r, float *lr_base, const uint32_t *keys, const float *rows, uint32_t n_rows, const uint2 *lr_ent,
                              uint32_t n_lr, bool in_order, hipStream_t stream) {
    if (!n_rows && !n_lr) return hipSuccess;
    const uint32_t F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0, R = F * r->cfg.ffm_k;
    const uint32_t blocks = in_order ? 1u : std::min<uint32_t>(4096u, std::max<uint32_t>(1u, (n_rows + 3) / 4 + (n_lr + 255) / 256));
    const dim3 grid(blocks), block(in_order ? 64 : 256);
#define FW_APPLY(OPT)                                                                                                                              \
    hipLaunchKernelGGL(owner_apply_kernel<OPT>, grid, block, 0, stream, r->d_ffm_w, r->d_ffm_acc, lr_base, keys, rows, n_rows, lr_ent, n_lr, R, \
                       r->cfg.ffm_learning_rate, -r->cfg.ffm_power_t, r->d_lut_ffm, r->cfg.learning_rate, -r->cfg.power_t, r->d_lut_lr, in_order ? 1 : 0)
    switch (r->cfg.optimizer) {
    case FWGPU_OPT_SGD: FW_APPLY(FWGPU_OPT_SGD); break;
    case FWGPU_OPT_ADAGRAD_FLEX: FW_APPLY(FWGPU_OPT_ADAGRAD_FLEX); break;
    default: FW_APPLY(FWGPU_OPT_ADAGRAD_LUT); break;
    }
#undef FW_APPLY
    return hipGetLastError();
}

// ------------------------------------------------------------------ init / fill / checksum

// merand48 0.1.0 (VW's LCG), one step from `seed` (block_ffm.rs:801, 811)
__device__ __forceinline__ float merand48(unsigned long long seed) {
    seed = 0xeece66d5deece66dULL * seed + 2147483647ULL;
    const uint32_t t = (uint32_t)((seed >> 25) & 0x7FFFFF) | (127u << 23);
    return __uint_as_float(t) - 1.0f;
}

// block_ffm.rs:784-829
__global__ void ffm_init_kernel(float *w, float *acc, unsigned long long len, float one_over_k_root, float init_width,
                                float init_zero_band, float init_center, float acc0) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        float x;
        if (init_width == 0.0f) {
            x = __fmul_rn(__fsub_rn(__fmul_rn(1.0f, merand48(len + i)), 0.5f), one_over_k_root);
        } else {
            const float zero_half_band_width = __fmul_rn(__fmul_rn(init_width, init_zero_band), 0.5f);
            const float band_width = __fmul_rn(init_width, __fsub_rn(1.0f, init_zero_band));
            x = __fsub_rn(__fmul_rn(merand48(i), band_width), __fmul_rn(band_width, 0.5f));
            if (x > 0.0f)
                x = __fadd_rn(x, zero_half_band_width);
            else
                x = __fsub_rn(x, zero_half_band_width);
            x = __fadd_rn(x, init_center);
        }
        w[i] = x;
        acc[i] = acc0;
    }
}

hipError_t launch_ffm_init(float *w, float *acc, uint64_t len, uint32_t k, float init_width, float init_zero_band,
                           float init_center, float acc0, hipStream_t stream) {
    if (!len) return hipSuccess;
    const float one_over_k_root = 1.0f / sqrtf((float)k) / 50.0f;  // block_ffm.rs:797
    hipLaunchKernelGGL(ffm_init_kernel, dim3(2048), dim3(256), 0, stream, w, acc, (unsigned long long)len,
                       one_over_k_root, init_width, init_zero_band, init_center, acc0);
    return hipGetLastError();
}

__global__ void fill_kernel(float *p, unsigned long long n, float v) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
hipError_t launch_fill(float *p, uint64_t n, float v, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, stream, p, (unsigned long long)n, v);
    return hipGetLastError();
}

__global__ void fill_lr_kernel(float2 *p, unsigned long long n, float w, float acc) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        p[i] = float2{w, acc};
}
hipError_t launch_fill_lr(float *lr, uint64_t n_entries, float w, float acc, hipStream_t stream) {
    if (!n_entries) return hipSuccess;
    hipLaunchKernelGGL(fill_lr_kernel, dim3(2048), dim3(256), 0, stream, reinterpret_cast<float2 *>(lr),
                       (unsigned long long)n_entries, w, acc);
    return hipGetLastError();
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__global__ void checksum_kernel(const float *p, unsigned long long n, unsigned long long *out) {
    unsigned long long acc = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc += mix64((i << 32) ^ (unsigned long long)__float_as_uint(p[i]) ^ (i >> 32));
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
hipError_t launch_checksum(const float *p, uint64_t n, unsigned long long *out, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, stream, p, (unsigned long long)n, out);
    return hipGetLastError();
}


__global__ void offset_copy_kernel(unsigned long long *dst, const unsigned long long *src, unsigned n, unsigned long long add) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i] + add;
}
hipError_t launch_offset_copy(uint64_t *dst, const uint64_t *src, uint32_t n, uint64_t add, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(offset_copy_kernel, dim3(64), dim3(256), 0, stream, reinterpret_cast<unsigned long long *>(dst),
                       reinterpret_cast<const unsigned long long *>(src), n, (unsigned long long)add);
    return hipGetLastError();
}
__global__ void add_kernel(float *dst, const float *src, unsigned long long n) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}
hipError_t launch_add(float *dst, const float *src, uint64_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(add_kernel, dim3(1024), dim3(256), 0, stream, dst, src, (unsigned long long)n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ replica delta bookkeeping (multi-GPU sync)
// start : d = D = t - s0                    (D is then all-reduced in place over RCCL)
// finish: s0 += D ; t += D - d              (others' updates land; local updates made meanwhile stay in t)
__global__ void delta_start_kernel(const float *t, const float *s0, float *d, float *D, unsigned long long n, float scale) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += stride) {
        if (i * 4 + 4 <= n) {
            const f4 x = reinterpret_cast<const f4 *>(t)[i] - reinterpret_cast<const f4 *>(s0)[i];
            reinterpret_cast<f4 *>(d)[i] = x;
            reinterpret_cast<f4 *>(D)[i] = x * scale;
        } else {
            for (unsigned long long j = i * 4; j < n; ++j) {
                d[j] = t[j] - s0[j];
                D[j] = d[j] * scale;
            }
        }
    }
}
__global__ void delta_finish_kernel(float *t, float *s0, const float *d, const float *D, unsigned long long n) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += stride) {
        if (i * 4 + 4 <= n) {
            const f4 Dv = reinterpret_cast<const f4 *>(D)[i];
            const f4 others = Dv - reinterpret_cast<const f4 *>(d)[i];
            reinterpret_cast<f4 *>(s0)[i] += Dv;
            reinterpret_cast<f4 *>(t)[i] += others;
        } else {
            for (unsigned long long j = i * 4; j < n; ++j) {
                s0[j] += D[j];
                t[j] += D[j] - d[j];
            }
        }
    }
}
hipError_t launch_delta_start(const float *t, const float *s0, float *d, float *D, uint64_t n, float scale, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(delta_start_kernel, dim3(2048), dim3(256), 0, stream, t, s0, d, D, (unsigned long long)n, scale);
    return hipGetLastError();
}
hipError_t launch_delta_finish(float *t, float *s0, const float *d, const float *D, uint64_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(delta_finish_kernel, dim3(2048), dim3(256), 0, stream, t, s0, d, D, (unsigned long long)n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ cross-XCD visibility probe (debug / tests)
// Block 0 publishes a 1 KiB payload (value = iteration number) with sc1 or plain stores, drains its stores and
// bumps a device-scope flag.  Every other block (they land on all 8 XCDs) keeps the payload line warm in its
// own L1/L2, waits for the flag, re-reads the payload with sc1 or plain loads and counts words older than the
// flag.  No fences: this is exactly the access pattern of the hogwild-mode table traffic.
__global__ void coherence_probe_kernel(unsigned *payload, unsigned *flag, unsigned *acks, unsigned *stale,
                                       unsigned *timeouts, int use_sc1, unsigned iters) {
    const int lane = threadIdx.x;  // 64 threads
    const unsigned nblk = gridDim.x;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(payload, 256 * 4);
    const unsigned spin_cap = 4000000u;
    if (blockIdx.x == 0) {
        for (unsigned it = 1; it <= iters; ++it) {
            u4 v = {it, it, it, it};
            if (use_sc1)
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, kAuxSc1);
            else
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, kAuxPlain);
            __builtin_amdgcn_s_waitcnt(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(flag, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned b = 1; b < nblk; ++b) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(acks + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > spin_cap) {
                            atomicAdd(timeouts, 1u);
                            return;
                        }
                    }
                }
            }
            __syncthreads();
        }
    } else {
        unsigned bad = 0;
        for (unsigned it = 1; it <= iters; ++it) {
            unsigned spins = 0;
            for (;;) {
                // keep the payload line resident in this CU's L1 / this XCD's L2 while waiting
                u4 w = use_sc1 ? __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, kAuxSc1)
                               : __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, kAuxPlain);
                unsigned f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f >= it) break;
                if (w.x == 0xdeadbeefu) bad += 1000000u;  // keeps the load alive
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_cap) {
                    if (lane == 0) atomicAdd(timeouts, 1u);
                    return;
                }
            }
            u4 v = use_sc1 ? __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, kAuxSc1)
                           : __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, kAuxPlain);
            bad += (v.x < it) + (v.y < it) + (v.z < it) + (v.w < it);
            __builtin_amdgcn_s_waitcnt(0);
            if (lane == 0) __hip_atomic_store(acks + blockIdx.x, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        if (bad) atomicAdd(stale, bad);
    }
}

hipError_t launch_coherence_probe(unsigned *scratch /* >= 256+1+64+2 words, zeroed */, int use_sc1, unsigned iters,
                                  unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(coherence_probe_kernel, dim3(blocks), dim3(64), 0, stream, scratch, scratch + 256,
                       scratch + 320, scratch + 400, scratch + 401, use_sc1, iters);
    return hipGetLastError();
}

#endif  // !FW_PHASE_TU
}  // namespace fwgpu
