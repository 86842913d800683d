// Host side of the C ABI: the `Regressor` object (regressor.rs:142-147) that owns the device tables,
// single-example learn/predict (regressor.rs:356-395), micro-batches, and weight (de)serialisation
// (regressor.rs:426-469).  There is no CPU compute path: every learn/predict runs the HIP kernel.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <memory>

#include "fwgpu_internal.h"

#ifndef FWGPU_LR_THIN_DEFAULT  // store policy 4 also on hot LR entries of the large-table path (kernels.hip lr_update; DESIGN 4.2: shipped in round 6)
#define FWGPU_LR_THIN_DEFAULT 1
#endif
namespace fwgpu {

static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

// optimizer.rs:121-144 (host: runs once per block at create time, then lives in HBM)
void lut_init(float *lut, float learning_rate, float power_t, float init_acc) {
    const float minus_power_t = -power_t;
    for (uint32_t x = 0; x < (uint32_t)kLutSize; x++) {
        uint32_t b0 = x << (31 - kLutBits), b1 = (x + 1) << (31 - kLutBits);
        float f0, f1;
        memcpy(&f0, &b0, 4);
        memcpy(&f1, &b1, 4);
        const float float_x = f0 + init_acc;
        const float float_x_plus_one = f1 + init_acc;
        float val = learning_rate * (powf(float_x, minus_power_t) + powf(float_x_plus_one, minus_power_t)) * 0.5f;
        if (std::isnan(val) || std::isinf(val)) val = learning_rate;
        lut[x] = val;
    }
}

static float initial_acc(int optimizer, float init_acc) {
    // optimizer.rs:40-42 (SGD: no state), 90-92 (Flex: init_acc), 158-161 (LUT: 0.0, folded into the table)
    return optimizer == FWGPU_OPT_ADAGRAD_FLEX ? init_acc : 0.0f;
}

void HostBatch::clear() {
    ffm_hash.clear();
    ffm_val.clear();
    ffm_fld.clear();
    lr_hash.clear();
    lr_val.clear();
    lr_combo.clear();
    label.clear();
    importance.clear();
    ffm_off.assign(1, 0);
    lr_off.assign(1, 0);
    max_lr = max_ffm = 0;
    aligned4 = true;
}

int append_example(const fwgpu_regressor *r, HostBatch &hb, const fwgpu_lr_entry *lr, uint32_t n_lr,
                   const fwgpu_ffm_entry *ffm, uint32_t n_ffm, float label, float importance) {
    if (hb.ffm_off.empty()) hb.clear();
    const uint32_t k = r->cfg.ffm_k, F = r->cfg.ffm_num_fields;
    if (k == 0) n_ffm = 0;  // no FFM block: ffm_buffer is ignored
    uint32_t prev = 0;
    for (uint32_t i = 0; i < n_ffm; i++) {
        const uint32_t cfi = ffm[i].contra_field_index;
        if (cfi % k != 0 || cfi / k >= F) return fail(FWGPU_ERR_RANGE, "ffm entry: contra_field_index is not field*k of a known field");
        if (cfi < prev) return fail(FWGPU_ERR_INVALID, "ffm entries must be ordered by field (block_ffm.rs:165-183)");
        prev = cfi;
        if ((uint64_t)ffm[i].hash + (uint64_t)F * k > r->ffm_len) return fail(FWGPU_ERR_RANGE, "ffm entry: hash outside the weight table");
        if (ffm[i].hash & 3u) hb.aligned4 = false;
        hb.ffm_hash.push_back(ffm[i].hash);
        hb.ffm_val.push_back(ffm[i].value);
        hb.ffm_fld.push_back((uint8_t)(cfi / k));
    }
    if (r->cfg.wiring == FWGPU_WIRING_FFM_ONLY) n_lr = 0;
    for (uint32_t i = 0; i < n_lr; i++) {
        if (lr[i].hash >= r->lr_len) return fail(FWGPU_ERR_RANGE, "lr entry: hash outside the weight table");
        if (lr[i].combo_index >= r->cfg.num_combos) return fail(FWGPU_ERR_RANGE, "lr entry: combo_index >= num_combos");
        hb.lr_hash.push_back(lr[i].hash);
        hb.lr_val.push_back(lr[i].value);
        hb.lr_combo.push_back((uint16_t)lr[i].combo_index);
    }
    hb.ffm_off.push_back((uint32_t)hb.ffm_hash.size());
    hb.lr_off.push_back((uint32_t)hb.lr_hash.size());
    hb.label.push_back(label);
    hb.importance.push_back(importance);
    hb.max_lr = std::max(hb.max_lr, n_lr);
    hb.max_ffm = std::max(hb.max_ffm, n_ffm);
    return FWGPU_OK;
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

int batch_alloc(fwgpu_regressor *r, uint32_t n, uint64_t n_lr, uint64_t n_ffm, fwgpu_batch **out, bool host_mapped) {
    std::unique_ptr<fwgpu_batch> b(new fwgpu_batch());
    b->owner = r;
    b->n = n;
    b->n_lr = n_lr;
    b->n_ffm = n_ffm;
    size_t o = 0, off[11];
    off[0] = o; o = up256(o + 4 * n_ffm);
    off[1] = o; o = up256(o + 4 * n_ffm);
    off[2] = o; o = up256(o + n_ffm);
    off[3] = o; o = up256(o + 4 * ((size_t)n + 1));
    off[4] = o; o = up256(o + 4 * n_lr);
    off[5] = o; o = up256(o + 4 * n_lr);
    off[6] = o; o = up256(o + 4 * ((size_t)n + 1));
    off[7] = o; o = up256(o + 4 * (size_t)n);
    off[8] = o; o = up256(o + 4 * (size_t)n);
    off[9] = o; o = up256(o + 4 * (size_t)n);
    off[10] = o; o = up256(o + 2 * n_lr);
    b->dev_bytes = std::max<size_t>(o, 256);
    FWGPU_HIP(hipSetDevice(r->device));
    unsigned char *base = nullptr;
    if (host_mapped) {
        // single-example batches (fwgpu_learn / fwgpu_predict / predict_with_cache): the entries live in host memory the device
        // has mapped; the kernel's stage phase reads them over PCIe and the prediction comes back the same way -- no copy calls
        FWGPU_HIP(hipHostMalloc(&b->host_block, b->dev_bytes, hipHostMallocMapped | hipHostMallocCoherent));
        void *dv = nullptr;
        FWGPU_HIP(hipHostGetDevicePointer(&dv, b->host_block, 0));
        base = static_cast<unsigned char *>(dv);
        b->host_delta = static_cast<unsigned char *>(b->host_block) - base;
        FWGPU_HIP(hipMalloc((void **)&b->work_ring, kWorkRing * sizeof(uint32_t)));
        FWGPU_HIP(hipMemset(b->work_ring, 0, kWorkRing * sizeof(uint32_t)));
        b->work = b->work_ring;
    } else {
        FWGPU_HIP(hipMalloc(&b->dev, b->dev_bytes));
        FWGPU_HIP(hipMalloc((void **)&b->work, 64));
        base = static_cast<unsigned char *>(b->dev);
    }
    b->ffm_hash = reinterpret_cast<uint32_t *>(base + off[0]);
    b->ffm_val = reinterpret_cast<float *>(base + off[1]);
    b->ffm_fld = reinterpret_cast<uint8_t *>(base + off[2]);
    b->ffm_off = reinterpret_cast<uint32_t *>(base + off[3]);
    b->lr_hash = reinterpret_cast<uint32_t *>(base + off[4]);
    b->lr_val = reinterpret_cast<float *>(base + off[5]);
    b->lr_off = reinterpret_cast<uint32_t *>(base + off[6]);
    b->label = reinterpret_cast<float *>(base + off[7]);
    b->importance = reinterpret_cast<float *>(base + off[8]);
    b->pred = reinterpret_cast<float *>(base + off[9]);
    b->lr_combo = reinterpret_cast<uint16_t *>(base + off[10]);
    *out = b.release();
    return FWGPU_OK;
}

static size_t up256b(size_t x) { return (x + 255) & ~(size_t)255; }

int mapped_batch_next_counter(fwgpu_batch *b, hipStream_t stream) {
    if (!b->work_ring) return fail(FWGPU_ERR_INVALID, "not a host-mapped batch");
    if (b->work_next == kWorkRing) {  // every counter has been used once (a 1-example launch leaves 2 in its counter)
        FWGPU_HIP(hipMemsetAsync(b->work_ring, 0, kWorkRing * sizeof(uint32_t), stream));
        b->work_next = 0;
    }
    b->work = b->work_ring + b->work_next++;
    return FWGPU_OK;
}

int record_batch_alloc(fwgpu_regressor *r, const fwgpu_translator_config *t, uint32_t n_cap, uint64_t words_cap,
                       fwgpu_batch **out, bool host_mapped) {
    int rc = check_translator(r, t);
    if (rc) return rc;
    if (t->n_fields > 255) return fail(FWGPU_ERR_INVALID, "translator: more than 255 fields");
    std::unique_ptr<fwgpu_batch> b(new fwgpu_batch());
    b->owner = r;
    b->n = 0;
    b->n_cap = n_cap;
    b->words_cap = words_cap;
    FWGPU_HIP(hipSetDevice(r->device));
    size_t o = 0;
    const size_t o_rec = o; o = up256b(o + 4 * words_cap);
    const size_t o_off = o; o = up256b(o + 8 * ((size_t)n_cap + 1));
    const size_t o_pred = o; o = up256b(o + 4 * (size_t)n_cap);
    b->dev_bytes = std::max<size_t>(o, 256);
    unsigned char *base = nullptr;
    if (host_mapped) {
        FWGPU_HIP(hipHostMalloc(&b->host_block, b->dev_bytes, hipHostMallocMapped | hipHostMallocCoherent));
        void *dv = nullptr;
        FWGPU_HIP(hipHostGetDevicePointer(&dv, b->host_block, 0));
        base = static_cast<unsigned char *>(dv);
        unsigned char *hb = static_cast<unsigned char *>(b->host_block);
        b->h_records = reinterpret_cast<uint32_t *>(hb + o_rec);
        b->h_rec_off = reinterpret_cast<uint64_t *>(hb + o_off);
        b->h_pred = reinterpret_cast<float *>(hb + o_pred);
        FWGPU_HIP(hipMalloc((void **)&b->work_ring, kWorkRing * sizeof(uint32_t)));
        FWGPU_HIP(hipMemset(b->work_ring, 0, kWorkRing * sizeof(uint32_t)));
        b->work = b->work_ring;
    } else {
        FWGPU_HIP(hipMalloc(&b->dev, b->dev_bytes));
        FWGPU_HIP(hipMalloc((void **)&b->work, 64));
        base = static_cast<unsigned char *>(b->dev);
    }
    b->records = reinterpret_cast<uint32_t *>(base + o_rec);
    b->rec_off = reinterpret_cast<uint64_t *>(base + o_off);
    b->pred = reinterpret_cast<float *>(base + o_pred);
    // translator blob: the (field, namespace) pairs are flattened in field order
    const uint32_t ncm = t->n_combos ? t->combo_off[t->n_combos] : 0;
    const uint32_t npr = (t->ffm_k && t->n_fields) ? t->field_off[t->n_fields] : 0;
    std::vector<uint8_t> pair_field(npr);
    for (uint32_t f = 0; f < t->n_fields && t->ffm_k; f++)
        for (uint32_t m = t->field_off[f]; m < t->field_off[f + 1]; m++) pair_field[m] = (uint8_t)f;
    size_t q = 0;
    const size_t q_coff = q; q = up256b(q + 4 * ((size_t)t->n_combos + 1));
    const size_t q_cns = q; q = up256b(q + 4 * (size_t)ncm);
    const size_t q_cf32 = q; q = up256b(q + ncm);
    const size_t q_cw = q; q = up256b(q + 4 * (size_t)t->n_combos);
    const size_t q_pns = q; q = up256b(q + 4 * (size_t)npr);
    const size_t q_pf32 = q; q = up256b(q + npr);
    const size_t q_pfld = q; q = up256b(q + npr);
    std::vector<unsigned char> blob(std::max<size_t>(q, 256), 0);
    memcpy(blob.data() + q_coff, t->combo_off, 4 * ((size_t)t->n_combos + 1));
    if (ncm) {
        memcpy(blob.data() + q_cns, t->combo_ns, 4 * (size_t)ncm);
        memcpy(blob.data() + q_cf32, t->combo_ns_f32, ncm);
    }
    if (t->n_combos) memcpy(blob.data() + q_cw, t->combo_weight, 4 * (size_t)t->n_combos);
    if (npr) {
        memcpy(blob.data() + q_pns, t->field_ns, 4 * (size_t)npr);
        memcpy(blob.data() + q_pf32, t->field_ns_f32, npr);
        memcpy(blob.data() + q_pfld, pair_field.data(), npr);
    }
    FWGPU_HIP(hipMalloc(&b->tr_dev, blob.size()));
    FWGPU_HIP(hipMemcpy(b->tr_dev, blob.data(), blob.size(), hipMemcpyHostToDevice));
    unsigned char *tb = static_cast<unsigned char *>(b->tr_dev);
    b->tr.combo_off = reinterpret_cast<const uint32_t *>(tb + q_coff);
    b->tr.combo_ns = reinterpret_cast<const uint32_t *>(tb + q_cns);
    b->tr.combo_f32 = reinterpret_cast<const uint8_t *>(tb + q_cf32);
    b->tr.combo_w = reinterpret_cast<const float *>(tb + q_cw);
    b->tr.pair_ns = reinterpret_cast<const uint32_t *>(tb + q_pns);
    b->tr.pair_f32 = reinterpret_cast<const uint8_t *>(tb + q_pf32);
    b->tr.pair_field = reinterpret_cast<const uint8_t *>(tb + q_pfld);
    b->tr.n_combos = t->n_combos;
    b->tr.n_pairs = npr;
    b->tr.n_members = ncm;
    b->tr.add_const = t->add_constant_feature ? 1 : 0;
    b->tr.lr_mask = r->lr_hash_mask;
    b->tr.ffm_mask = r->ffm_hash_mask;
    *out = b.release();
    return FWGPU_OK;
}

int count_records(const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off, uint32_t n,
                  RecordStats *st) {
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t len = (uint32_t)(rec_off[i + 1] - rec_off[i]);
        uint32_t nl, nf;
        int rc = count_record(t, records + rec_off[i], len, &nl, &nf);
        if (rc) return rc;
        st->max_lr = std::max(st->max_lr, nl);
        st->max_ffm = std::max(st->max_ffm, nf);
        st->max_rec = std::max(st->max_rec, len);
        st->tot_lr += nl;
        st->tot_ffm += nf;
    }
    return FWGPU_OK;
}

int record_batch_upload(fwgpu_batch *b, const fwgpu_translator_config *t, const uint32_t *records, const uint64_t *rec_off,
                        uint32_t n, hipStream_t stream, const RecordStats *stats) {
    if (!b->records) return fail(FWGPU_ERR_INVALID, "not a record batch");
    const uint64_t words = n ? rec_off[n] - rec_off[0] : 0;
    if (n > b->n_cap || words > b->words_cap) return fail(FWGPU_ERR_RANGE, "record batch larger than its allocation");
    // validate + count on the host (slot words only, no hashing): sizes the kernel's LDS arrays
    RecordStats st;
    if (stats) {
        st = *stats;
    } else {
        int rc = count_records(t, records, rec_off, n, &st);
        if (rc) return rc;
    }
    const uint32_t max_lr = st.max_lr, max_ffm = st.max_ffm, max_rec = st.max_rec;
    const uint64_t tot_lr = st.tot_lr, tot_ffm = st.tot_ffm;
    if (n) {
        FWGPU_HIP(hipMemcpyAsync(b->records, records + rec_off[0], words * 4, hipMemcpyHostToDevice, stream));
        if (rec_off[0] == 0) {
            FWGPU_HIP(hipMemcpyAsync(b->rec_off, rec_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, stream));
        } else {
            std::vector<uint64_t> rel((size_t)n + 1);
            for (uint32_t i = 0; i <= n; i++) rel[i] = rec_off[i] - rec_off[0];
            FWGPU_HIP(hipMemcpyAsync(b->rec_off, rel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, stream));
            FWGPU_HIP(hipStreamSynchronize(stream));
        }
    }
    b->n = n;
    b->n_lr = tot_lr;
    b->n_ffm = tot_ffm;
    b->n_words = words;
    b->max_lr = b->owner->cfg.wiring == FWGPU_WIRING_FFM_ONLY ? 0 : max_lr;
    b->max_ffm = max_ffm;
    b->max_rec = max_rec;
    b->aligned4 = true;  // hash & ffm_mask clears the low log2(next_pow2(k)) bits (feature_buffer.rs:141-148)
    return FWGPU_OK;
}

int batch_upload(fwgpu_batch *b, const HostBatch &hb, hipStream_t stream) {
    const uint32_t n = hb.size();
    if (n > b->n || hb.ffm_hash.size() > b->n_ffm || hb.lr_hash.size() > b->n_lr)
        return fail(FWGPU_ERR_RANGE, "batch_upload: host batch larger than the device allocation");
#define UP(dst, vec)                                                                                              \
    if (!(vec).empty()) {                                                                                         \
        if (b->host_block)                                                                                        \
            memcpy(reinterpret_cast<unsigned char *>(dst) + b->host_delta, (vec).data(), (vec).size() * sizeof((vec)[0])); \
        else                                                                                                      \
            FWGPU_HIP(hipMemcpyAsync((dst), (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, stream)); \
    }
    UP(b->ffm_hash, hb.ffm_hash);
    UP(b->ffm_val, hb.ffm_val);
    UP(b->ffm_fld, hb.ffm_fld);
    UP(b->ffm_off, hb.ffm_off);
    UP(b->lr_hash, hb.lr_hash);
    UP(b->lr_val, hb.lr_val);
    UP(b->lr_combo, hb.lr_combo);
    UP(b->lr_off, hb.lr_off);
    UP(b->label, hb.label);
    UP(b->importance, hb.importance);
#undef UP
    b->max_lr = hb.max_lr;
    b->max_ffm = hb.max_ffm;
    b->aligned4 = hb.aligned4;
    return FWGPU_OK;
}

KernelParams make_params(const fwgpu_regressor *r, const fwgpu_batch *b, int update) {
    KernelParams p{};
    p.ffm_w = r->d_ffm_w;
    p.ffm_acc = r->d_ffm_acc;
    p.lr = r->d_lr;
    p.lut_lr = r->d_lut_lr;
    p.lut_ffm = r->d_lut_ffm;
    p.ffm_hash = b->ffm_hash;
    p.ffm_val = b->ffm_val;
    p.ffm_fld = b->ffm_fld;
    p.ffm_off = b->ffm_off;
    p.lr_hash = b->lr_hash;
    p.lr_val = b->lr_val;
    p.lr_off = b->lr_off;
    p.lr_combo = b->lr_combo;
    p.nn = r->nn;
    p.num_combos = r->cfg.num_combos;
    p.label = b->label;
    p.importance = b->importance;
    p.pred = b->pred;
    p.n_examples = b->n;
    p.F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0;
    p.k = r->cfg.ffm_k;
    p.R = p.F * p.k;
    p.max_ffm = std::max<uint32_t>(4, (b->max_ffm + 3) & ~3u);
    p.max_lr = std::max<uint32_t>(4, (b->max_lr + 3) & ~3u);
    p.has_lr = r->cfg.wiring == FWGPU_WIRING_REGRESSOR;
    p.update = update ? 1 : 0;
    p.aligned4 = b->aligned4 ? 1 : 0;
    // Whole-line row updates pay where partial-line writes reach HBM, i.e. when w + acc do not fit the 256 MiB Infinity
    // Cache (profiles/r02_rowceil.txt: cache-resident tables write partial lines at full rate).  Small tables also are
    // where concurrent examples would meet on a line most often, so they keep float-granular writes.
    // (launch_example_kernel keeps the flag only where the whole-line path exists.)
    p.window = r->launch.window >= 2 || (r->launch.window == 1 && r->ffm_len * 8ull > (256ull << 20)) ? 1 : 0;  // (3: the path of large tables on a table of any size, tests)
    // Whole-line accesses pay when w and acc contend for one region of the device memory (partial-line writes are what is slow
    // then: -6.5 %); with the accumulator table placed away from the weights they buy nothing (4.64 vs 4.60 M examples/s) and
    // only widen hogwild's race from the float to the line, so they are used when asked for (option 2 = 2) or when the placement
    // search found no partner that does not contend.
    p.line_pass = r->launch.window == 2 || (p.window && r->placement_contended) ? 1 : 0;
    p.no_chain = r->launch.no_chain;
    p.hot_lr_hash = 11650396u & lr_hash_mask(r->cfg.bit_precision);  // feature_buffer.rs:8, 270-276
    p.hot_lr_every = r->launch.hot_lr_every;
    {
        static const char *env = getenv("FWGPU_HOT_LR_EVERY");  // A/B runs: 0 = plain read-modify-writes
        if (env) p.hot_lr_every = (uint32_t)atoi(env);
    }
    p.k_log2 = 0xffu;
    for (uint32_t l = 0; l < 16; l++)
        if ((1u << l) == r->cfg.ffm_k) p.k_log2 = l;
    p.lr_rate = r->cfg.learning_rate;
    p.lr_minus_power_t = -r->cfg.power_t;
    p.ffm_rate = r->cfg.ffm_learning_rate;
    p.ffm_minus_power_t = -r->cfg.ffm_power_t;
    p.ticks = r->d_ticks;
    if (!update && b->cache) {  // predict_with_cache: the field sums start from the cached features' (regressor.rs:397-407)
        p.ctx_T = b->cache->d_T;
        p.ctx_dcf = b->cache->d_dcf;
        if (b->records) p.ctx_cover = b->cache->d_cover;
        if (b->records && b->delta_records) {
            p.ctx_rec = b->cache->d_ctx_rec;
            p.ctx_rec_len = (uint32_t)b->cache->ctx_rec.size();
        }
    }
    if (!update) {
        p.emit_T = b->emit_T;
        p.emit_dcf = b->emit_dcf;
    }
    p.records = b->records;
    p.rec_self_len = b->rec_self_len ? 1 : 0;
    p.rec_off = b->rec_off;
    p.max_rec = (b->max_rec + 3) & ~3u;
    p.tr = b->tr;
    p.kernel_version = r->launch.kernel_version;
    p.lut_global = r->launch.lut_global;
    {   // FFM row store policy of hogwild launches and its write-back interval (kernels.hip, "store policy"); -1 / ~0 = the kernels' build default
        static const char *env_pol = getenv("FWGPU_STORE_POLICY"), *env_wb = getenv("FWGPU_WB_FLUSH_EVERY"), *env_pf = getenv("FWGPU_PREFETCH");
        p.store_policy = r->launch.store_policy >= 0 ? r->launch.store_policy : (env_pol ? atoi(env_pol) : -1);
        static const char *env_th = getenv("FWGPU_ACC_HOT_THETA"), *env_sm = getenv("FWGPU_ACC_SAMPLE_LOG2");  // policy 3's two knobs (A/B runs)
        // "hot" = the row has ACCUMULATED more than theta: measured from the accumulators' initial value (AdagradFlex starts them at ffm_init_acc_gradient,
        // optimizer.rs:90-92; AdagradLUT at 0 with the initial value folded into its table, 158-161), so a row is never hot before it has been stepped
        const float theta = r->launch.acc_hot_theta >= 0.0f ? r->launch.acc_hot_theta : (env_th ? (float)atof(env_th) : 0.5f);
        p.acc_hot_theta = theta + initial_acc(r->cfg.optimizer, r->cfg.ffm_init_acc_gradient);
        const int sm = r->launch.acc_sample_log2 >= 0 ? r->launch.acc_sample_log2 : (env_sm ? atoi(env_sm) : 3);
        p.acc_sample_log2 = (uint32_t)std::min(std::max(sm, 0), 6);  // (1u << it in the kernels: one example in 1 .. 64)
        static const char *env_kr = getenv("FWGPU_KEPT_ROWS");
        p.no_kept_rows = (r->launch.kept_rows >= 0 ? r->launch.kept_rows == 0 : (env_kr && env_kr[0] == '0')) ? 1 : 0;
        static const char *env_lt = getenv("FWGPU_LR_THIN");
        // (an LR entry's g^2 is (g v)^2 ~ 0.1-0.25 per hit where an FFM float's is ~1e-4: the same "stepped by more than a hundred examples" reads 64 x the FFM rows' threshold here)
        static const char *env_ls = getenv("FWGPU_LR_HOT_SCALE");
        p.lr_hot_theta = theta * (env_ls ? (float)atof(env_ls) : 64.0f) + initial_acc(r->cfg.optimizer, r->cfg.init_acc_gradient);
        p.lr_thin = r->launch.lr_thin >= 0 ? r->launch.lr_thin : (env_lt ? atoi(env_lt) : FWGPU_LR_THIN_DEFAULT);
        static const char *env_tr = getenv("FWGPU_THIN_REREAD");
        p.thin_reread = env_tr ? atoi(env_tr) : 1;
        p.wb_flush_every = r->launch.wb_flush_every >= 0 ? (uint32_t)r->launch.wb_flush_every : (env_wb ? (uint32_t)atoi(env_wb) : 0xffffffffu);
        p.prefetch = (r->launch.prefetch && !(env_pf && env_pf[0] == '0')) ? 1 : 0;
        static const char *env_lk = getenv("FWGPU_LDS_KEEP");
        // rows per wave kept in LDS beyond the register-kept ones: the wish (clamped to the kernel's maximum by resolve_row_mode, to what leaves two
        // workgroups on a CU by prepare_launch)
        p.lds_keep = r->launch.lds_keep >= 0 ? (uint32_t)r->launch.lds_keep : (env_lk ? (uint32_t)atoi(env_lk) : 255u);
    }
    p.work = b->work;
    p.host_cus = r->num_cus;
    p.host_wgs_cap = r->launch.workgroups_per_cu;
    p.host_grid_cap = r->launch.max_in_flight;
    return p;
}

uint32_t pick_grid(const fwgpu_regressor *r, const KernelParams &p, int mode, uint32_t threads) {
    (void)r; (void)p; (void)threads;
    // SEQUENTIAL: one workgroup walks the batch.  HOGWILD: 0 = "what the device keeps resident" (launch_persistent)
    return mode == FWGPU_MODE_SEQUENTIAL ? 1u : 0u;
}

// launch shape of a batch: workgroup size, LDS copy of the LUT or not (settles p.window / p.chain / p.lut_global)
uint32_t nn_v2_threads();  // (kernels.hip: the workgroup size the head-as-a-phase instantiation was built for)
static int prepare_launch(fwgpu_regressor *r, fwgpu_batch *b, int mode, int update, KernelParams &p, uint32_t &threads) {
    p = make_params(r, b, update);
    p.concurrent = (mode == FWGPU_MODE_HOGWILD && p.host_grid_cap != 1 && b->n > 1) ? 1 : 0;  // (one workgroup -- in-order semantics -- otherwise)
    threads = r->launch.threads;
    if (mode == FWGPU_MODE_SEQUENTIAL) threads = std::max<uint32_t>(threads, 512);
    if ((uint64_t)p.max_ffm > 4ull * threads) threads = 1024;
    if ((uint64_t)p.max_ffm > 4ull * threads)
        return fail(FWGPU_ERR_RANGE, "an example has more than 4096 FFM features");
    // Small examples (BASELINE configs[1]: 10 features, 11 LR entries): a 512-thread workgroup has eight waves for ten rows, and the rate of such a launch is (examples in
    // flight) / (an example's latency) with MORE than ~512 in flight costing both rate and loss (the hot LR entries' lines).  Two waves per example and two workgroups per CU:
    // 13.8 M examples/s at hold-out 0.59431 against 11.9 M at 0.59419 for the 512-thread shape (profiles/r06_configB_time_to_loss_table.txt).  FWGPU_SMALL_SHAPE=0: the old shape.
    {
        static const char *env_ss = getenv("FWGPU_SMALL_SHAPE");
        // (not the generic kernel forced by kernel_version 1: the peer-sharded and streaming launches size their consumer workgroups by the regressor's own workgroup size)
        if (!(env_ss && env_ss[0] == '0') && p.concurrent && update && !r->launch.threads_set && r->launch.kernel_version != 1 && !p.nn.n_layers && p.max_ffm <= 32 && p.max_lr <= 64) {
            threads = 128;
            if (!p.host_grid_cap) p.host_grid_cap = 2 * r->num_cus;
        }
    }
    // Config E's concurrent launches: the deep head as a phase of the v2 kernel's two-chunk instantiation (kernels.hip fw_example_kernel_r<..., NN = true>) where TWO of its
    // 512-thread workgroups fit a CU -- the AdaGrad LUT read through L1, no record prefetch, no LDS copy of the entries' own slots -- instead of one 1024-thread workgroup
    // of the generic kernel.  In-order launches (the parity mode) and shapes that do not fit stay on the generic kernel.  fwgpu_debug_set_option(r, 11, 0) / FWGPU_NN_V2=0: off.
    {
        static const char *env_nn = getenv("FWGPU_NN_V2");
        const bool wish = r->launch.nn_v2 >= 0 ? r->launch.nn_v2 != 0 : !(env_nn && env_nn[0] == '0');
        const bool forced = r->launch.nn_v2 == 2;  // (tests: in-order launches too, whatever fits -- the head's wiring in that kernel against the oracle, per example)
        if (wish && p.nn.n_layers && update && (p.concurrent || forced) && p.window && p.k_log2 != 0xffu && p.R > 64 * 4 && (!r->launch.threads_set || r->launch.threads == nn_v2_threads()) && r->launch.kernel_version != 1 && (uint64_t)p.max_ffm <= 4ull * 512) {
            KernelParams q = p;
            q.nn_v2 = 1;
            q.lut_global = 1;
            q.prefetch = 0;
            const uint32_t th_q = nn_v2_threads();
            resolve_row_mode(q, th_q);
            const size_t lds_q = example_kernel_lds_bytes(q, r->cfg.optimizer);
            if (example_kernel_is_resident(q, th_q) && q.window && (forced ? lds_q <= r->lds_per_cu : 2 * lds_q <= r->lds_per_cu)) {
                p = q;
                threads = th_q;
            }
        }
    }
    resolve_row_mode(p, threads);
    size_t lds = example_kernel_lds_bytes(p, r->cfg.optimizer);
    // rows parked in LDS by the gather (KernelParams::lds_keep): as many per wave as still let a second workgroup live on the CU
    while (p.lds_keep > 0 && 2 * lds > r->lds_per_cu) {
        p.lds_keep--;
        resolve_row_mode(p, threads);
        lds = example_kernel_lds_bytes(p, r->cfg.optimizer);
    }
    // When the LDS footprint lets only ONE workgroup live on a CU (k = 16 rows: T alone is 57.6 KB), the CU's waves have
    // to come from that workgroup: 1024 threads (measured at k = 16: 1.53 -> 1.82 M examples/s; with the deep head 0.67 -> 0.79).
    if (!r->launch.threads_set && mode == FWGPU_MODE_HOGWILD && 2 * lds > r->lds_per_cu) {
        // ... unless giving up the LDS copy of the AdaGrad LUT (8 KB) is what lets a SECOND workgroup in: two examples in
        // flight with the LUT read through L1 beat one example with the LUT in LDS (k = 16: 1.82 -> 1.93 M examples/s)
        KernelParams q = p;
        q.lut_global = 1;
        const size_t lds_nolut = example_kernel_lds_bytes(q, r->cfg.optimizer);  // (== lds where the kernel keeps the LUT in LDS regardless: lut_lds_forced)
        if (!p.lut_global && 2 * lds_nolut <= r->lds_per_cu) {
            p.lut_global = 1;
            lds = lds_nolut;
        } else if (example_kernel_is_resident(p, threads)) {
            // ... and unless the launch runs on the v2 kernel (models without a deep head): ONE 512-thread workgroup of it per CU beats one 1024-thread workgroup
            // of the generic kernel on both axes -- headless k = 16, round 5: 2.62-2.63 M examples/s and hold-out 0.653-0.655 against 2.52-2.53 M and 0.673
            // (profiles/r05_k16_threads_ab.txt)
        } else {
            threads = 1024;
            p.lds_keep = 0;
            resolve_row_mode(p, threads);
            lds = example_kernel_lds_bytes(p, r->cfg.optimizer);
        }
    }
    if (lds > r->lds_per_cu)
        return fail(FWGPU_ERR_RANGE, "example does not fit the 160 KiB LDS (k*F^2 or features per example too large)");
    return FWGPU_OK;
}

static int run_batch_by_example(fwgpu_regressor *r, fwgpu_batch *b, int update, hipStream_t stream);
static bool head_predict_batched(const fwgpu_regressor *r, const fwgpu_batch *b, int mode, int update);
static int run_batch_head_predict(fwgpu_regressor *r, fwgpu_batch *b, hipStream_t stream);
static int run_batch(fwgpu_regressor *r, fwgpu_batch *b, int mode, int update, hipStream_t stream) {
    if (b->n == 0) return FWGPU_OK;
    if (b->host_copy) return run_batch_by_example(r, b, update, stream);  // (an example beyond what the fused kernel stages: see learn_one_chunked)
    if (head_predict_batched(r, b, mode, update)) {
        const int rcb = run_batch_head_predict(r, b, stream);
        if (rcb != FWGPU_ERR_RANGE) return rcb;  // (a shape the v2 kernel cannot stage: the per-example forward below)
    }
    KernelParams p;
    uint32_t threads = 0;
    int rc = prepare_launch(r, b, mode, update, p, threads);
    if (rc) return rc;
    const uint32_t grid = pick_grid(r, p, mode, threads);
    if (b->work_ring) {
        rc = mapped_batch_next_counter(b, stream);
        if (rc) return rc;
        p.work = b->work;
    } else {
        FWGPU_HIP(hipMemsetAsync(b->work, 0, sizeof(uint32_t), stream));
    }
    // An updating launch always uses device-scope (sc1) accesses; read-only launches use cached loads.
    FWGPU_HIP(launch_example_kernel(p, r->cfg.optimizer, update != 0, grid, threads, stream));
    return FWGPU_OK;
}

// Predict-only batches of a model with a deep head (hold-out passes, serving): the weights are frozen, so nothing orders the examples and the head need
// not run per example.  The FFM / LR part runs on the v2 kernel (two workgroups per CU, no dense traffic) and leaves every example's head input x
// (regressor.rs:185-189: LR combo sums | triangle) in a batch buffer; the layers then take the batch through v_mfma_f32_32x32x2_f32 GEMMs, slab by slab
// (head.hip head_step, update = false) -- the dense weights are read once per slab instead of once per example (block_neural.rs:196-222 per example:
// 0.77 MB of weights for 1.2 MFLOP).  Same numbers as the per-example forward up to the order of the f32 sums.
static constexpr uint32_t kHeadPredictSlab = 32768;
static bool head_predict_batched(const fwgpu_regressor *r, const fwgpu_batch *b, int mode, int update) {
    static const bool off = getenv("FWGPU_HEAD_PREDICT_PER_EXAMPLE") != nullptr;  // A/B runs and tests: the per-example forward for every launch
    if (off || update || !r->nn.n_layers || mode != FWGPU_MODE_HOGWILD || b->n < 256 || b->cache || b->emit_T) return false;
    const uint32_t k = r->cfg.ffm_k, R = k * r->cfg.ffm_num_fields;
    return k && k % 4 == 0 && b->aligned4 && (R <= 256 || (R <= 512 && 256 % k == 0)) && r->launch.kernel_version != 1;
}
static int run_batch_head_predict(fwgpu_regressor *r, fwgpu_batch *b, hipStream_t stream) {
    const uint32_t X = r->nn.X;
    if (r->pred_cap < b->n) {
        FWGPU_HIP(hipStreamSynchronize(stream));
        if (r->pred_x) (void)hipFree(r->pred_x);
        if (r->pred_yi) (void)hipFree(r->pred_yi);
        r->pred_x = r->pred_yi = nullptr;
        r->pred_cap = 0;
        if (hipMalloc((void **)&r->pred_x, (size_t)b->n * X * sizeof(float)) != hipSuccess || hipMalloc((void **)&r->pred_yi, (size_t)b->n * 2 * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FWGPU_ERR_OOM, "predict batch with a deep head: no memory for the head's inputs");
        }
        r->pred_cap = b->n;
    }
    KernelParams p;
    uint32_t threads = 0;
    {   // (prepare_launch with the head's per-example LDS taken out of the picture)
        p = make_params(r, b, 0);
        p.emit_x = 1;
        p.xbuf = r->pred_x;
        p.gbuf = r->pred_yi;
        threads = std::min<uint32_t>(r->launch.threads, 512);
        if ((uint64_t)p.max_ffm > 4ull * threads) return fail(FWGPU_ERR_RANGE, "an example has more than 2048 FFM features");  // (the caller falls back)
        resolve_row_mode(p, threads);
        if (example_kernel_lds_bytes(p, r->cfg.optimizer) > r->lds_per_cu) return fail(FWGPU_ERR_RANGE, "example does not fit the LDS");
    }
    if (b->work_ring) {
        int rc = mapped_batch_next_counter(b, stream);
        if (rc) return rc;
        p.work = b->work;
    } else {
        FWGPU_HIP(hipMemsetAsync(b->work, 0, sizeof(uint32_t), stream));
    }
    FWGPU_HIP(launch_example_kernel(p, r->cfg.optimizer, false, 0, threads, stream));
    fwgpu_split view;
    view.owner = r;
    view.d_x = r->pred_x;
    view.d_g = r->pred_yi;
    for (uint32_t first = 0; first < b->n; first += kHeadPredictSlab) {
        int rc = head_step(r, &view, first, std::min(kHeadPredictSlab, b->n - first), b->pred + first, /*update=*/false, stream);
        if (rc) return rc;
    }
    return FWGPU_OK;
}

int run_batch_peer(fwgpu_regressor *r, fwgpu_batch *b, int mode, int update, const PeerShards *d_shards, hipStream_t stream,
                   const PushRings *d_push, uint32_t stream_consumers, uint32_t device_share, uint32_t stream_max_consumer_waves, uint32_t n_ranks) {
    if (b->n == 0 && !stream_consumers) return FWGPU_OK;  // (a streaming step launches for an empty batch too: the rank's consumers serve the peers)
    if (r->nn.n_layers) return fail(FWGPU_ERR_INVALID, "peer-sharded tables: models with a deep head are not covered");
    KernelParams p;
    uint32_t threads = 0;
    const int32_t kv = r->launch.kernel_version;
    r->launch.kernel_version = 1;  // the generic kernel carries the owner lookup (kernels.hip ffm_w_base / lr_base)
    int rc = prepare_launch(r, b, mode, update, p, threads);
    r->launch.kernel_version = kv;
    if (rc) return rc;
    p.shards = d_shards;
    p.push = d_push;
    if (d_push) p.hot_lr_every = 0;  // (owner-side apply: every LR gradient travels to its owner, the constant feature's included)
    p.host_extra_wgs = stream_consumers;
    p.host_share = device_share;
    p.host_stream_max_consumer_waves = stream_max_consumer_waves;
    p.host_stream_min_consumer_waves = (d_push && stream_consumers) ? 2u * (n_ranks ? n_ranks : 1u) : 0u;
    const uint32_t grid = pick_grid(r, p, mode, threads);
    FWGPU_HIP(hipMemsetAsync(b->work, 0, sizeof(uint32_t), stream));
    FWGPU_HIP(launch_example_kernel(p, r->cfg.optimizer, update != 0, grid, threads, stream));
    return FWGPU_OK;
}

// a batch with an example beyond what the fused kernel stages keeps its entries on the host: it is walked example by example (run_batch)
// does the batch's largest example fit what a workgroup of the fused kernel stages?  (fewer than 4096 entries of each kind can still be more than the
// LDS holds: at k = 8 x 30 fields about 2.5 k features, at k = 16 far fewer -- ADVICE r4)
static bool batch_fits_fused_kernel(fwgpu_batch *b) {
    if (b->max_ffm > 4096 || b->max_lr > 4096) return false;
    if (!b->owner) return true;
    KernelParams p;
    uint32_t threads = 0;
    const std::string keep = fwgpu_last_error();
    const int rc = prepare_launch(b->owner, b, FWGPU_MODE_HOGWILD, 1, p, threads);
    if (rc == FWGPU_ERR_RANGE) set_error(keep);  // (not an error of the caller's: the batch takes the example-by-example path)
    return rc != FWGPU_ERR_RANGE;
}
void keep_host_copy_if_oversize(fwgpu_batch *b, HostBatch &&hb) {
    if (hb.max_ffm > 4096 || hb.max_lr > 4096 || !batch_fits_fused_kernel(b)) b->host_copy.reset(new HostBatch(std::move(hb)));
}
// ... a RECORD batch (just uploaded: max_ffm / max_lr are known) with a record that translates to more entries than that: translated on the host
int record_batch_host_copy_if_oversize(fwgpu_regressor *r, const fwgpu_translator_config *t, fwgpu_batch *b, const uint32_t *records,
                                       const uint64_t *rec_off, uint32_t n) {
    b->host_copy.reset();
    if (batch_fits_fused_kernel(b)) return FWGPU_OK;
    HostBatch hb;
    hb.clear();
    std::vector<fwgpu_lr_entry> lr;
    std::vector<fwgpu_ffm_entry> ffm;
    for (uint32_t i = 0; i < n; i++) {
        float label, imp;
        int rc = translate_record(t, records + rec_off[i], (uint32_t)(rec_off[i + 1] - rec_off[i]), lr, ffm, &label, &imp);
        if (rc == FWGPU_OK) rc = append_example(r, hb, lr.data(), (uint32_t)lr.size(), ffm.data(), (uint32_t)ffm.size(), label, imp);
        if (rc) return rc;
    }
    keep_host_copy_if_oversize(b, std::move(hb));
    return FWGPU_OK;
}

}  // namespace fwgpu

using namespace fwgpu;

extern "C" {

const char *fwgpu_last_error(void) { return g_last_error.c_str(); }
int fwgpu_abi_version(void) { return FWGPU_ABI_VERSION; }

// Where the accumulator table goes relative to the weight table.  Device memory on MI355X falls into groups (tools/placement.hip,
// profiles/r02_placement.txt): the FFM update's pattern -- whole-line read-modify-write of w[h..] and acc[h..] at once -- runs 17 %
// slower when both tables come from the same group (2.45 ms vs 2.04 ms per 3.3 M rows), whatever their addresses.  hipMalloc hands
// out consecutive allocations from one group for several GB, so for tables beyond the Infinity Cache a few candidate allocations
// are tried and timed against w with that very pattern (a fraction of a millisecond each); the fastest is kept, the rest freed.
// FWGPU_PLACEMENT=0 switches the search off (the first allocation is used, as for small tables).
// The search is BOUNDED by who else is on the device: alone (more than 85 % of the device memory free -- a training box) it may hold
// up to 112 candidates / half of the free memory for the few milliseconds it takes (a contending stretch of the allocator is up to
// ~72 table-GB long, profiles/r02_placement.txt; 24 and 64 candidates each left whole boxes on a contending pair this round:
// -10 % examples/s); with other tenants present at most 16 candidates and a tenth of the free memory, and only two candidates when
// more than half of the device is in use (other regressors, ranks, serving replicas must not run out of memory because this
// one is probing).  FWGPU_PLACEMENT=wide forces the wide scan, =0 switches the search off.
static int place_ffm_acc(fwgpu_regressor *r, size_t fbytes) {
    const char *env = std::getenv("FWGPU_PLACEMENT");
    const bool search = fbytes > (256u << 20) && !(env && env[0] == '0');
    const bool wide = env && env[0] == 'w';
    std::vector<float *> cand;
    float single = 0.0f, lo = 1e30f, hi = 0.0f;
    int best = 0;
    if (search) {
        // the weight table alone, until two consecutive readings agree (the first launches of a process run on ramping clocks)
        float prev = 0.0f;
        for (int i = 0; i < 24; i++) {
            if (pair_probe_ms(r->d_ffm_w, nullptr, fbytes, 400000, 2, &single) != hipSuccess) {
                (void)hipGetLastError();
                single = 0.0f;
                break;
            }
            if (prev > 0.0f && std::fabs(single - prev) < 0.02f * prev) break;
            prev = single;
        }
    }
    // the pair costs 1.8x the single table when the two do not contend, 2.2x when they do (profiles/r02_placement.txt)
    const float fast_below = 1.96f * single;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        free_b = 0;
    }
    // Candidates are allocations of the table's own size, held until the search ends so that hipMalloc keeps walking forward.
    // (Large "spacer" allocations to skip ahead do not work: they are assembled from elsewhere and the next table-sized block
    // still comes from the same stretch.)  A stretch of contending memory is up to ~72 table-GB long (tools/placement scan of
    // 200 x 1 GiB, profiles/r02_placement.txt), a probe costs about a millisecond.
    const bool crowded = total_b && free_b < total_b / 2;  // someone else (another process, other regressors) holds half the device
    const bool alone = total_b && free_b > total_b / 100 * 85;
    const size_t budget = (wide || alone) ? std::min<size_t>(free_b / 2, 128ull << 30) : free_b / 10;
    const size_t cap = (wide || alone) ? 112 : (crowded ? 2 : 16);
    const int max_tries = search && single > 0.0f ? (int)std::max<size_t>(2, std::min<size_t>(cap, budget / fbytes)) : 1;
    for (int t = 0; t < max_tries; t++) {
        float *c = nullptr;
        if (hipMalloc((void **)&c, fbytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        cand.push_back(c);
        if (max_tries == 1) break;
        float t_ms = 0.0f;
        if (pair_probe_ms(r->d_ffm_w, c, fbytes, 400000, 2, &t_ms) != hipSuccess) {
            (void)hipGetLastError();
            break;  // (keep what there is; the probe is an optimisation)
        }
        hi = std::max(hi, t_ms);
        if (t_ms < lo) {
            lo = t_ms;
            best = t;
        }
        if (t_ms < fast_below) break;  // does not contend with w: done
    }
    if (cand.empty()) return fail(FWGPU_ERR_DEVICE, "out of device memory (FFM accumulators)");
    r->d_ffm_acc = cand[best];
    for (size_t i = 0; i < cand.size(); i++)
        if ((int)i != best) (void)hipFree(cand[i]);
    // (no search, or no candidate that does not contend: the launches assume contention and use whole-line accesses)
    r->placement_contended = fbytes > (256u << 20) && !(search && single > 0.0f && lo < fast_below);
    r->placement_tries = (int)cand.size();
    r->placement_ms_lo = hi > 0.0f ? lo : 0.0f;
    r->placement_ms_hi = hi;
    r->placement_ms_single = single;
    return FWGPU_OK;
}

int fwgpu_create(const fwgpu_config *cfg, fwgpu_regressor **out) {
    if (!cfg || !out) return fail(FWGPU_ERR_INVALID, "fwgpu_create: NULL argument");
    *out = nullptr;
    if (cfg->optimizer != FWGPU_OPT_SGD && cfg->optimizer != FWGPU_OPT_ADAGRAD_FLEX &&
        cfg->optimizer != FWGPU_OPT_ADAGRAD_LUT)
        return fail(FWGPU_ERR_INVALID, "fwgpu_create: unknown optimizer");
    if (cfg->bit_precision < 1 || cfg->bit_precision > 31) return fail(FWGPU_ERR_INVALID, "fwgpu_create: bit_precision out of range");
    if (cfg->ffm_k > 0) {
        if (cfg->ffm_bit_precision < 1 || cfg->ffm_bit_precision > 31)
            return fail(FWGPU_ERR_INVALID, "fwgpu_create: ffm_bit_precision out of range");
        if (cfg->ffm_num_fields == 0 || cfg->ffm_num_fields > 255)
            return fail(FWGPU_ERR_INVALID, "fwgpu_create: ffm_num_fields must be in 1..255");
        // block_ffm.rs:96-101
        if ((uint64_t)cfg->ffm_k * cfg->ffm_num_fields * cfg->ffm_num_fields > kFfmContraBufLen)
            return fail(FWGPU_ERR_INVALID, "FFM_CONTRA_BUF_LEN is 41472. It needs to be at least ffm_k * number_of_fields^2");
    }
    if (cfg->wiring != FWGPU_WIRING_REGRESSOR && cfg->wiring != FWGPU_WIRING_FFM_ONLY)
        return fail(FWGPU_ERR_INVALID, "fwgpu_create: unknown wiring");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FWGPU_ERR_DEVICE, "fwgpu_create: no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(FWGPU_ERR_DEVICE, "fwgpu_create: device ordinal out of range");
    std::unique_ptr<fwgpu_regressor> r(new fwgpu_regressor());
    r->cfg = *cfg;
    r->device = cfg->device;
    FWGPU_HIP(hipSetDevice(r->device));
    hipDeviceProp_t prop;
    FWGPU_HIP(hipGetDeviceProperties(&prop, r->device));
    r->num_cus = prop.multiProcessorCount;
    r->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 64 * 1024;
    r->lr_len = 1ull << cfg->bit_precision;
    r->ffm_len = cfg->ffm_k ? (1ull << cfg->ffm_bit_precision) + (uint64_t)cfg->ffm_num_fields * cfg->ffm_k : 0;
    r->lr_hash_mask = lr_hash_mask(cfg->bit_precision);
    r->ffm_hash_mask = ffm_hash_mask(cfg->ffm_bit_precision, cfg->ffm_k);
    // (where the LR table lands relative to the FFM tables was measured not to matter: 3.55 ms per launch either way)
    FWGPU_HIP(hipMalloc((void **)&r->d_lr, r->lr_len * 2 * sizeof(float)));
    FWGPU_HIP(hipMemset(r->d_lr, 0, r->lr_len * 2 * sizeof(float)));
    if (r->ffm_len) {
        // +64 floats of slack so that a 16 B vector at the very end of the spill-over tail stays in the allocation
        const size_t fbytes = (r->ffm_len + 64) * sizeof(float);
        FWGPU_HIP(hipMalloc((void **)&r->d_ffm_w, fbytes));
        int rc = place_ffm_acc(r.get(), fbytes);
        if (rc) return rc;
        FWGPU_HIP(hipMemset(r->d_ffm_w, 0, fbytes));
        FWGPU_HIP(hipMemset(r->d_ffm_acc, 0, fbytes));
    }
    // each block owns its optimizer instance: block_lr.rs:63-65, block_ffm.rs:86-91
    std::vector<float> lut(kLutSize);
    FWGPU_HIP(hipMalloc((void **)&r->d_lut_lr, kLutSize * sizeof(float)));
    FWGPU_HIP(hipMalloc((void **)&r->d_lut_ffm, kLutSize * sizeof(float)));
    lut_init(lut.data(), cfg->learning_rate, cfg->power_t, cfg->init_acc_gradient);
    FWGPU_HIP(hipMemcpy(r->d_lut_lr, lut.data(), kLutSize * sizeof(float), hipMemcpyHostToDevice));
    lut_init(lut.data(), cfg->ffm_learning_rate, cfg->ffm_power_t, cfg->ffm_init_acc_gradient);
    FWGPU_HIP(hipMemcpy(r->d_lut_ffm, lut.data(), kLutSize * sizeof(float), hipMemcpyHostToDevice));
    // accumulators start at initial_data() even before init_weights (Vec allocation happens there in the
    // reference; a zero table with initial accumulators is the natural "new_without_weights" state here)
    const float a0 = initial_acc(cfg->optimizer, cfg->init_acc_gradient);
    if (a0 != 0.0f) FWGPU_HIP(launch_fill_lr(r->d_lr, r->lr_len, 0.0f, a0, 0));
    const float fa0 = initial_acc(cfg->optimizer, cfg->ffm_init_acc_gradient);
    if (r->ffm_len && fa0 != 0.0f) FWGPU_HIP(launch_fill(r->d_ffm_acc, r->ffm_len, fa0, 0));
    FWGPU_HIP(hipDeviceSynchronize());
    *out = r.release();
    return FWGPU_OK;
}

int fwgpu_free(fwgpu_regressor *r) {
    if (!r) return FWGPU_OK;
    (void)hipSetDevice(r->device);
    if (r->one) fwgpu_batch_free(r->one);
    head_scratch_free(r);
    if (r->pred_x) (void)hipFree(r->pred_x);
    if (r->pred_yi) (void)hipFree(r->pred_yi);
    (void)hipFree(r->d_lr);
    (void)hipFree(r->d_ffm_w);
    (void)hipFree(r->d_ffm_acc);
    (void)hipFree(r->d_lut_lr);
    (void)hipFree(r->d_lut_ffm);
    (void)hipFree(r->d_nn_w);
    (void)hipFree(r->d_nn_acc);
    (void)hipFree(r->d_lut_nn);
    if (r->pinned) (void)hipHostFree(r->pinned);
    delete r;
    return FWGPU_OK;
}

static int nn_init_weights(fwgpu_regressor *r);

int fwgpu_init_weights(fwgpu_regressor *r) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    FWGPU_HIP(hipSetDevice(r->device));
    const fwgpu_config &c = r->cfg;
    // block_lr.rs:97-105
    FWGPU_HIP(launch_fill_lr(r->d_lr, r->lr_len, 0.0f, initial_acc(c.optimizer, c.init_acc_gradient), 0));
    // block_ffm.rs:784-829
    if (r->ffm_len)
        FWGPU_HIP(launch_ffm_init(r->d_ffm_w, r->d_ffm_acc, r->ffm_len, c.ffm_k, c.ffm_init_width, c.ffm_init_zero_band,
                                  c.ffm_init_center, initial_acc(c.optimizer, c.ffm_init_acc_gradient), 0));
    if (r->nn.n_layers) {
        int rc = nn_init_weights(r);
        if (rc) return rc;
    }
    FWGPU_HIP(hipDeviceSynchronize());
    return FWGPU_OK;
}

int fwgpu_set_nn(fwgpu_regressor *r, const fwgpu_nn_config *nn) {
    if (!r || !nn) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (r->cfg.wiring != FWGPU_WIRING_REGRESSOR) return fail(FWGPU_ERR_INVALID, "the deep head needs the REGRESSOR wiring");
    if (nn->n_layers == 0 || nn->n_layers > FWGPU_NN_MAX_LAYERS) return fail(FWGPU_ERR_INVALID, "nn: n_layers must be in 1..8");
    if (nn->topology != 1 && nn->topology != 2)
        return fail(FWGPU_ERR_INVALID, "unknown nn topology (\"one\" and \"two\" are supported; \"four\"/\"five\" use block_normalize, out of scope)");
    FWGPU_HIP(hipSetDevice(r->device));
    DevNN d{};
    const uint32_t F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0;
    d.n_layers = nn->n_layers;
    d.topology = nn->topology;
    d.X = r->cfg.num_combos + F * (F + 1) / 2;  // the Join span (regressor.rs:185-189)
    uint32_t in = d.X, off = 0;
    d.max_in = d.X;
    for (uint32_t l = 0; l < nn->n_layers; l++) {
        if (nn->width[l] == 0 || nn->width[l] > 4096) return fail(FWGPU_ERR_INVALID, "nn: layer width must be in 1..4096");
        if (nn->init[l] > FWGPU_NN_INIT_ZERO) return fail(FWGPU_ERR_INVALID, "nn: unknown init type");
        if (in >= 16000) return fail(FWGPU_ERR_INVALID, "nn: too many inputs (MAX_NUM_INPUTS, block_neural.rs:27)");
        d.in[l] = in;
        d.out[l] = nn->width[l];
        d.off[l] = off;
        d.relu[l] = nn->relu[l] ? 1 : 0;
        off += (in + 1) * nn->width[l];  // +1: bias term (block_neural.rs:86)
        d.sum_width += nn->width[l];
        d.max_out = std::max(d.max_out, nn->width[l]);
        in = nn->width[l];
        d.max_in = std::max(d.max_in, in);
    }
    const uint32_t L = nn->n_layers;
    d.in[L] = in + (nn->topology == 1 ? d.X : 0);  // Join(h_last, copy of x), regressor.rs:307-311
    d.out[L] = 1;
    d.off[L] = off;
    off += d.in[L] + 1;
    d.max_in = std::max(d.max_in, d.in[L]);
    d.rate = nn->nn_learning_rate;
    d.minus_power_t = -nn->nn_power_t;
    if (r->d_nn_w) (void)hipFree(r->d_nn_w);
    if (r->d_nn_acc) (void)hipFree(r->d_nn_acc);
    if (r->d_lut_nn) (void)hipFree(r->d_lut_nn);
    r->nn_len = off;
    FWGPU_HIP(hipMalloc((void **)&r->d_nn_w, (size_t)off * 4));
    FWGPU_HIP(hipMalloc((void **)&r->d_nn_acc, (size_t)off * 4));
    FWGPU_HIP(hipMalloc((void **)&r->d_lut_nn, kLutSize * sizeof(float)));
    FWGPU_HIP(hipMemset(r->d_nn_w, 0, (size_t)off * 4));
    std::vector<float> lut(kLutSize);
    lut_init(lut.data(), nn->nn_learning_rate, nn->nn_power_t, nn->nn_init_acc_gradient);  // block_neural.rs:111-112
    FWGPU_HIP(hipMemcpy(r->d_lut_nn, lut.data(), kLutSize * sizeof(float), hipMemcpyHostToDevice));
    FWGPU_HIP(launch_fill(r->d_nn_acc, off, initial_acc(r->cfg.optimizer, nn->nn_init_acc_gradient), 0));
    FWGPU_HIP(hipDeviceSynchronize());
    d.w = r->d_nn_w;
    d.acc = r->d_nn_acc;
    d.lut = r->d_lut_nn;
    r->nn = d;
    r->nn_cfg = *nn;
    return FWGPU_OK;
}

// block_neural.rs:367-424.  The reference seeds Xoshiro256++ per layer and samples rand_distr Normal / Uniform;
// that stream is third-party and unpinned, so the library uses its own deterministic generator with the same
// distributions: Hu = N(0, sqrt(2/in)), Xavier = U(+-sqrt(6)/sqrt(in*out)), One, Zero; biases are always 0.
static int nn_init_weights(fwgpu_regressor *r) {
    const DevNN &d = r->nn;
    std::vector<float> w(r->nn_len, 0.0f);
    auto rng = [](uint64_t &s) {
        uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    };
    auto u01 = [&](uint64_t &s) { return ((double)(rng(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
    for (uint32_t l = 0; l <= d.n_layers; l++) {
        const uint32_t in = d.in[l], out = d.out[l];
        const size_t bias = (size_t)in * out;
        const uint32_t init = l == d.n_layers ? (uint32_t)FWGPU_NN_INIT_ONE : r->nn_cfg.init[l];  // regressor.rs:312-319
        uint64_t st = 0x5eed0000ULL + 7919ULL * l + in + (bias + out);
        float *wl = w.data() + d.off[l];
        for (size_t i = 0; i < bias; i++) {
            switch (init) {
            case FWGPU_NN_INIT_XAVIER: wl[i] = (float)((2.0 * u01(st) - 1.0) * (sqrt(6.0) / sqrt((double)bias))); break;
            case FWGPU_NN_INIT_HU: {
                const double u1 = u01(st), u2 = u01(st);
                wl[i] = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * sqrt(2.0 / (double)in));
                break;
            }
            case FWGPU_NN_INIT_ONE: wl[i] = 1.0f; break;
            default: wl[i] = 0.0f;
            }
        }
    }
    FWGPU_HIP(hipMemcpy(r->d_nn_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    FWGPU_HIP(launch_fill(r->d_nn_acc, r->nn_len, initial_acc(r->cfg.optimizer, r->nn_cfg.nn_init_acc_gradient), 0));
    return FWGPU_OK;
}

int fwgpu_set_launch(fwgpu_regressor *r, uint32_t threads, uint32_t workgroups_per_cu) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    if (threads) {
        if (threads % 64 || threads > 1024) return fail(FWGPU_ERR_INVALID, "threads must be a multiple of 64, <= 1024");
        r->launch.threads = threads;
        r->launch.threads_set = true;
    }
    r->launch.workgroups_per_cu = workgroups_per_cu;
    return FWGPU_OK;
}

int fwgpu_set_max_in_flight(fwgpu_regressor *r, uint32_t n_examples) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    r->launch.max_in_flight = n_examples;
    return FWGPU_OK;
}

int fwgpu_debug_set_option(fwgpu_regressor *r, int option, int value) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    switch (option) {
    case 1: r->launch.lut_global = value ? 1 : 0; return FWGPU_OK;
    case 3: r->launch.no_chain = value ? 1 : 0; return FWGPU_OK;  // duplicate-row chains off (A/B runs)
    case 4:  // hogwild launches: the constant feature's LR entry is stepped in LDS and reaches the table every `value` examples (0: off)
        if (value < 0 || value > 1024) return fail(FWGPU_ERR_INVALID, "hot LR entry option: 0 .. 1024 examples");
        r->launch.hot_lr_every = (uint32_t)value;
        return FWGPU_OK;
    case 5:  // FFM row store policy of hogwild launches: 0 write-through, 1 weights write-back, 2 both tables write-back, -1 the build's default
        if (value < -1 || value > 4) return fail(FWGPU_ERR_INVALID, "store policy option: -1, 0, 1, 2, 3 or 4");
        r->launch.store_policy = value;
        return FWGPU_OK;
    case 13:  // rows kept from the gather in the large-table kernel: 0 = none (every row re-read by the update: no last-writer-wins over an example's lifetime), 1 / -1 = kept (default)
        if (value < -1 || value > 1) return fail(FWGPU_ERR_INVALID, "kept-rows option: -1, 0 or 1");
        r->launch.kept_rows = value;
        return FWGPU_OK;
    case 12:  // store policy 4 also on hot LR entries (weight stored alone, accumulator by thinned atomic adds): 0 / 1, -1 = default
        if (value < -1 || value > 1) return fail(FWGPU_ERR_INVALID, "LR thinning option: -1, 0 or 1");
        r->launch.lr_thin = value;
        return FWGPU_OK;
    case 11:  // the deep head of concurrent two-chunk launches as a phase of the v2 kernel (1, default where two workgroups fit a CU) or always on the generic kernel (0); -1 = default
        if (value < -1 || value > 2) return fail(FWGPU_ERR_INVALID, "deep-head kernel option: -1, 0, 1 or 2");
        r->launch.nn_v2 = value;
        return FWGPU_OK;
    case 9:  // policies 3 / 4: a row counts as hot when its accumulators have grown by more than value / 1024 (-1: the default, 0.5)
        if (value < -1 || value > (1 << 24)) return fail(FWGPU_ERR_INVALID, "hot-row threshold option: -1, or 0 .. 2^24 (in 1/1024)");
        r->launch.acc_hot_theta = value < 0 ? -1.0f : (float)value / 1024.0f;
        return FWGPU_OK;
    case 10:  // policies 3 / 4: one example in 2^value touches a hot row's accumulators (-1: the default, 3)
        if (value < -1 || value > 6) return fail(FWGPU_ERR_INVALID, "accumulator sampling option: -1, or 0 .. 6");
        r->launch.acc_sample_log2 = value;
        return FWGPU_OK;
    case 6:  // write-back interval of policies 1 / 2: a workgroup issues buffer_wbl2 every `value` of its examples (0 never, -1 the build's default)
        if (value < -1 || value > 65536) return fail(FWGPU_ERR_INVALID, "write-back interval option: -1 .. 65536 examples");
        r->launch.wb_flush_every = value;
        return FWGPU_OK;
    case 8:  // rows per wave kept in LDS beyond the register-kept ones (-1: automatic)
        if (value < -1 || value > 3) return fail(FWGPU_ERR_INVALID, "LDS-kept rows option: 0 .. 3, or -1 (automatic)");
        r->launch.lds_keep = value;
        return FWGPU_OK;
    case 7: r->launch.prefetch = value ? 1 : 0; return FWGPU_OK;  // next-record prefetch of the v2 kernel's updating launches
    case 2:  // whole-line FFM row updates: 0 off, 1 auto (tables larger than the Infinity Cache; default), 2 always, 3 chained path always with float-granular accesses
        if (value < 0 || value > 3) return fail(FWGPU_ERR_INVALID, "window option: 0, 1, 2 or 3");
        r->launch.window = value;
        return FWGPU_OK;
    }
    return fail(FWGPU_ERR_INVALID, "unknown debug option");
}

int fwgpu_debug_set_kernel_version(fwgpu_regressor *r, int version) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    if (version < 0 || version > 2) return fail(FWGPU_ERR_INVALID, "kernel version must be 0 (auto), 1 or 2");
    r->launch.kernel_version = version;
    return FWGPU_OK;
}

// ------------------------------------------------------------------ single example

static int ensure_one(fwgpu_regressor *r, uint32_t n_lr, uint32_t n_ffm) {
    if (r->one && r->one->n_lr >= n_lr && r->one->n_ffm >= n_ffm) return FWGPU_OK;
    if (r->one) {
        fwgpu_batch_free(r->one);
        r->one = nullptr;
    }
    return batch_alloc(r, 1, std::max<uint32_t>(n_lr * 2, 256), std::max<uint32_t>(n_ffm * 2, 256), &r->one, /*host_mapped=*/true);
}

// the single example's prediction, once the launch has finished
static int one_prediction(fwgpu_regressor *r, float *prediction) {
    FWGPU_HIP(hipStreamSynchronize(0));
    *prediction = *reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(r->one->pred) + r->one->host_delta);
    return FWGPU_OK;
}

// ---- examples beyond the fused kernel's LDS budget.
// The reference takes an example of any size (block_ffm.rs:294-312: the gradient cache moves to the heap beyond 170 393 floats, the algorithm
// stays what it is).  The fused kernel stages an example's entries in LDS: 4096 FFM features / LR entries at most.  A larger example takes the
// synchronous pipeline instead, CHUNK by chunk -- field sums and LR sums are additive over features:
//   FWD   every chunk of <= kChunkEntries entries is a pseudo-example of a small batch: its partial field sums, self-pair corrections and LR sum
//         go to its split record, its features' own slots to the selfw buffer (pre-update weights, as block_ffm.rs:219-261 uses them);
//   sum   the chunks' records are added in chunk order and the total is copied back into every chunk's record;
//   MID   logit, prediction and general gradient once, from the total;
//   UPD   the chunks IN ORDER on one workgroup, each with the total record: AdaGrad per occurrence in buffer order, rows that repeat or
//         overlap across chunks included (a later chunk re-reads what an earlier one wrote), exactly the reference's update loop (block_ffm.rs:265-288).
// Only the order of the field sums' additions differs from the reference (chunk subtotals): ~1e-7 relative.  Models with a deep head are not covered.
constexpr uint32_t kMaxStagedEntries = 4096;  // what one workgroup of the fused kernel stages (prepare_launch)
constexpr uint32_t kChunkEntriesMax = 2048;
// entries per chunk: what the pipeline's workgroups stage beside the field sums (k floats of own-field weights and ~9 words per entry), a power of two
static uint32_t chunk_entries(const fwgpu_regressor *r) {
    const size_t k = r->cfg.ffm_k, F = k ? r->cfg.ffm_num_fields : 0, fixed = 4 * F * F * k + 12288;
    uint32_t n = kChunkEntriesMax;
    while (n > 64 && fixed + (size_t)n * (4 * k + 36) > r->lds_per_cu) n >>= 1;
    return n;
}

static int learn_one_chunked(fwgpu_regressor *r, const HostBatch &hb, uint32_t e, int update, float *prediction) {
    // With a deep head (regressor.rs:191-323) the chunks' partial records still add up -- per-combo LR sums, field sums, corrections, counts -- so MID forms
    // the head's input x from the TOTAL record once, the head runs once on it (a "mini-batch" of one example is the reference's per-example rule: one
    // AdaGrad step per dense weight with this example's gradient, input gradients from the pre-step weights, block_neural.rs:252-340), and every chunk's
    // update takes the example's per-slot gradients dx.
    const bool head = r->nn.n_layers != 0;
    const uint32_t f0 = hb.ffm_off[e], f1 = hb.ffm_off[e + 1], l0 = hb.lr_off[e], l1 = hb.lr_off[e + 1];
    const uint32_t nf = f1 - f0, nl = l1 - l0, kChunkEntries = chunk_entries(r);
    const uint32_t m = std::max<uint32_t>(1, std::max((nf + kChunkEntries - 1) / kChunkEntries, (nl + kChunkEntries - 1) / kChunkEntries));
    HostBatch hc;
    hc.clear();
    hc.aligned4 = hb.aligned4;
    for (uint32_t j = 0; j < m; j++) {
        const uint32_t fa = std::min(nf, j * kChunkEntries), fb = std::min(nf, (j + 1) * kChunkEntries);
        const uint32_t la = std::min(nl, j * kChunkEntries), lb = std::min(nl, (j + 1) * kChunkEntries);
        hc.ffm_hash.insert(hc.ffm_hash.end(), hb.ffm_hash.begin() + f0 + fa, hb.ffm_hash.begin() + f0 + fb);
        hc.ffm_val.insert(hc.ffm_val.end(), hb.ffm_val.begin() + f0 + fa, hb.ffm_val.begin() + f0 + fb);
        hc.ffm_fld.insert(hc.ffm_fld.end(), hb.ffm_fld.begin() + f0 + fa, hb.ffm_fld.begin() + f0 + fb);
        hc.lr_hash.insert(hc.lr_hash.end(), hb.lr_hash.begin() + l0 + la, hb.lr_hash.begin() + l0 + lb);
        hc.lr_val.insert(hc.lr_val.end(), hb.lr_val.begin() + l0 + la, hb.lr_val.begin() + l0 + lb);
        hc.lr_combo.insert(hc.lr_combo.end(), hb.lr_combo.begin() + l0 + la, hb.lr_combo.begin() + l0 + lb);
        hc.ffm_off.push_back((uint32_t)hc.ffm_hash.size());
        hc.lr_off.push_back((uint32_t)hc.lr_hash.size());
        hc.label.push_back(hb.label[e]);
        hc.importance.push_back(hb.importance[e]);
        hc.max_ffm = std::max(hc.max_ffm, fb - fa);
        hc.max_lr = std::max(hc.max_lr, lb - la);
    }
    fwgpu_batch *cb = nullptr;
    fwgpu_split *sp = nullptr;
    int rc = batch_alloc(r, m, hc.lr_hash.size(), hc.ffm_hash.size(), &cb);
    if (rc == FWGPU_OK) rc = batch_upload(cb, hc, 0);
    if (rc == FWGPU_OK) rc = fwgpu_split_create(r, m, kChunkEntries, &sp);
    auto done = [&](int code) {
        (void)hipStreamSynchronize(0);
        if (cb) fwgpu_batch_free(cb);
        if (sp) fwgpu_split_free(sp);
        return code;
    };
    if (rc) return done(rc);
    SplitRanges rg;
    rg.home_lo = 0;
    rg.home_hi = m;  // every chunk is "home": the features-per-field counts of the chunks add up like the sums (the head's diagonal needs the example's)
    if ((rc = split_forward(r, cb, sp, FWGPU_MODE_HOGWILD, rg, 0))) return done(rc);
    const size_t SL = sp->split_len;
    for (uint32_t j = 1; j < m; j++)
        if (launch_add(sp->d_split, sp->d_split + (size_t)j * SL, SL, 0) != hipSuccess) return done(fail(FWGPU_ERR_DEVICE, "chunked example: add failed"));
    {   // ... the label and the importance enter the total ONCE (every chunk carried them)
        const uint32_t F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0, R = F * r->cfg.ffm_k;
        const float li[2] = {hb.label[e], hb.importance[e]};
        if (hipMemcpyAsync(sp->d_split + (size_t)F * R + 2 * (size_t)F + sp->nlr, li, sizeof(li), hipMemcpyHostToDevice, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess)
            return done(fail(FWGPU_ERR_DEVICE, "chunked example: label write failed"));
    }
    if ((rc = split_mid(r, sp, 0, 1, cb->pred, head, 0))) return done(rc);
    const bool upd = update && hb.importance[e] != 0.0f;  // regressor.rs:366
    if (head && (rc = head_step(r, sp, 0, 1, cb->pred, upd, 0))) return done(rc);
    if (upd) {
        const size_t X = head ? r->nn.X : 0;
        for (uint32_t j = 1; j < m; j++) {
            if (hipMemcpyAsync(sp->d_split + (size_t)j * SL, sp->d_split, SL * 4, hipMemcpyDeviceToDevice, 0) != hipSuccess ||
                (!head && hipMemcpyAsync(sp->d_g + j, sp->d_g, 4, hipMemcpyDeviceToDevice, 0) != hipSuccess) ||
                (head && hipMemcpyAsync(sp->d_dx + (size_t)j * X, sp->d_dx, X * 4, hipMemcpyDeviceToDevice, 0) != hipSuccess))
                return done(fail(FWGPU_ERR_DEVICE, "chunked example: copy failed"));
        }
        if ((rc = split_update(r, cb, sp, FWGPU_MODE_SEQUENTIAL, rg, head, 0))) return done(rc);
    }
    if (hipMemcpyAsync(prediction, cb->pred, 4, hipMemcpyDeviceToHost, 0) != hipSuccess) return done(fail(FWGPU_ERR_DEVICE, "chunked example: read-back failed"));
    return done(FWGPU_OK);
}

// one example of a host batch through the single-example path (fused kernel in order, or chunk by chunk)
static int learn_host_example(fwgpu_regressor *r, const HostBatch &hb, uint32_t e, int update, float *prediction) {
    const uint32_t nf = hb.ffm_off[e + 1] - hb.ffm_off[e], nl = hb.lr_off[e + 1] - hb.lr_off[e];
    if (nf > kMaxStagedEntries || nl > kMaxStagedEntries) return learn_one_chunked(r, hb, e, update, prediction);
    HostBatch one;
    one.clear();
    one.aligned4 = hb.aligned4;
    one.ffm_hash.assign(hb.ffm_hash.begin() + hb.ffm_off[e], hb.ffm_hash.begin() + hb.ffm_off[e + 1]);
    one.ffm_val.assign(hb.ffm_val.begin() + hb.ffm_off[e], hb.ffm_val.begin() + hb.ffm_off[e + 1]);
    one.ffm_fld.assign(hb.ffm_fld.begin() + hb.ffm_off[e], hb.ffm_fld.begin() + hb.ffm_off[e + 1]);
    one.lr_hash.assign(hb.lr_hash.begin() + hb.lr_off[e], hb.lr_hash.begin() + hb.lr_off[e + 1]);
    one.lr_val.assign(hb.lr_val.begin() + hb.lr_off[e], hb.lr_val.begin() + hb.lr_off[e + 1]);
    one.lr_combo.assign(hb.lr_combo.begin() + hb.lr_off[e], hb.lr_combo.begin() + hb.lr_off[e + 1]);
    one.ffm_off.push_back(nf);
    one.lr_off.push_back(nl);
    one.label.push_back(hb.label[e]);
    one.importance.push_back(hb.importance[e]);
    one.max_ffm = nf;
    one.max_lr = nl;
    int rc = ensure_one(r, nl, nf);
    if (rc) return rc;
    if ((rc = batch_upload(r->one, one, 0))) return rc;
    rc = run_batch(r, r->one, FWGPU_MODE_SEQUENTIAL, update, 0);
    if (rc == FWGPU_ERR_RANGE) return learn_one_chunked(r, hb, e, update, prediction);
    if (rc) return rc;
    return one_prediction(r, prediction);
}

// A batch that holds an oversize example keeps its host copy (fwgpu_batch::host_copy) and is walked example by example, in order, whatever the
// mode: the in-order result is one of the outcomes hogwild allows.  The rare path: three launches per chunked example.
int fwgpu::run_batch_by_example(fwgpu_regressor *r, fwgpu_batch *b, int update, hipStream_t stream) {
    FWGPU_HIP(hipStreamSynchronize(stream));
    const HostBatch &hb = *b->host_copy;
    std::vector<float> preds(hb.size());
    for (uint32_t e = 0; e < hb.size(); e++) {
        int rc = learn_host_example(r, hb, e, update, &preds[e]);
        if (rc) return rc;
    }
    FWGPU_HIP(hipMemcpy(b->pred, preds.data(), preds.size() * 4, hipMemcpyHostToDevice));
    return FWGPU_OK;
}

static int learn_one(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm,
                     uint32_t n_ffm, float label, float importance, int update, float *prediction) {
    if (!r || !prediction) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if ((n_lr && !lr) || (n_ffm && !ffm)) return fail(FWGPU_ERR_INVALID, "NULL entry buffer");
    FWGPU_HIP(hipSetDevice(r->device));
    HostBatch hb;
    hb.clear();
    int rc = append_example(r, hb, lr, n_lr, ffm, n_ffm, label, importance);
    if (rc) return rc;
    if (hb.max_ffm > kMaxStagedEntries || hb.max_lr > kMaxStagedEntries) return learn_one_chunked(r, hb, 0, update, prediction);
    rc = ensure_one(r, (uint32_t)hb.lr_hash.size(), (uint32_t)hb.ffm_hash.size());
    if (rc) return rc;
    rc = batch_upload(r->one, hb, 0);
    if (rc) return rc;
    rc = run_batch(r, r->one, FWGPU_MODE_SEQUENTIAL, update, 0);
    // (fewer than 4096 entries of each kind, but together more than the LDS holds: prepare_launch refuses before anything is launched)
    if (rc == FWGPU_ERR_RANGE) return learn_one_chunked(r, hb, 0, update, prediction);
    if (rc) return rc;
    return one_prediction(r, prediction);
}

int fwgpu_learn(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm, uint32_t n_ffm,
                float label, float importance, int update, float *prediction) {
    return learn_one(r, lr, n_lr, ffm, n_ffm, label, importance, update, prediction);
}

int fwgpu_predict(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm,
                  uint32_t n_ffm, float *prediction) {
    return learn_one(r, lr, n_lr, ffm, n_ffm, 0.0f, 1.0f, 0, prediction);
}

// ------------------------------------------------------------------ batches

int fwgpu_batch_create(fwgpu_regressor *r, const fwgpu_lr_entry *lr, const uint32_t *lr_off, const fwgpu_ffm_entry *ffm,
                       const uint32_t *ffm_off, const float *label, const float *importance, uint32_t n,
                       fwgpu_batch **out) {
    if (!r || !out || !lr_off || !ffm_off || (n && (!label || !importance)))
        return fail(FWGPU_ERR_INVALID, "fwgpu_batch_create: NULL argument");
    *out = nullptr;
    HostBatch hb;
    hb.clear();
    for (uint32_t i = 0; i < n; i++) {
        if (lr_off[i + 1] < lr_off[i] || ffm_off[i + 1] < ffm_off[i]) return fail(FWGPU_ERR_INVALID, "offsets must be non-decreasing");
        int rc = append_example(r, hb, lr ? lr + lr_off[i] : nullptr, lr_off[i + 1] - lr_off[i],
                                ffm ? ffm + ffm_off[i] : nullptr, ffm_off[i + 1] - ffm_off[i], label[i], importance[i]);
        if (rc) return rc;
    }
    fwgpu_batch *b = nullptr;
    int rc = batch_alloc(r, n, hb.lr_hash.size(), hb.ffm_hash.size(), &b);
    if (rc) return rc;
    rc = batch_upload(b, hb, 0);
    if (rc == FWGPU_OK && hipStreamSynchronize(0) != hipSuccess) rc = fail(FWGPU_ERR_DEVICE, "upload failed");
    if (rc) {
        fwgpu_batch_free(b);
        return rc;
    }
    keep_host_copy_if_oversize(b, std::move(hb));
    *out = b;
    return FWGPU_OK;
}

int fwgpu_record_batch_create(fwgpu_regressor *r, const fwgpu_translator_config *t, const uint32_t *records,
                              const uint64_t *rec_off, uint32_t n, fwgpu_batch **out) {
    if (!r || !t || !out || (n && (!records || !rec_off))) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *out = nullptr;
    fwgpu_batch *b = nullptr;
    const uint64_t words = n ? rec_off[n] - rec_off[0] : 0;
    int rc = record_batch_alloc(r, t, std::max<uint32_t>(n, 1), std::max<uint64_t>(words, 1), &b);
    if (rc) return rc;
    rc = record_batch_upload(b, t, records, rec_off, n, 0);
    if (rc == FWGPU_OK && hipStreamSynchronize(0) != hipSuccess) rc = fail(FWGPU_ERR_DEVICE, "upload failed");
    if (rc == FWGPU_OK) rc = record_batch_host_copy_if_oversize(r, t, b, records, rec_off, n);
    if (rc) {
        fwgpu_batch_free(b);
        return rc;
    }
    *out = b;
    return FWGPU_OK;
}

// ------------------------------------------------------------------ synchronous micro-batch ("split") pipeline
// FWD (gather -> split records) -> MID (records -> prediction + general gradient) -> UPD (AdaGrad from the records): every
// example of the batch sees the weights of the batch start, like the examples in flight at once in hogwild.rs, but with a
// defined staleness (the batch).  It is what the owner-sharded multi-GPU mode (dist.cpp) and the mini-batched deep head
// (head.hip) are built from; on one GPU fwgpu_learn_batch_sync runs the three steps back to back.

int fwgpu_split_create(fwgpu_regressor *r, uint32_t n_examples, uint32_t max_ffm_per_example, fwgpu_split **out) {
    if (!r || !out || !n_examples) return fail(FWGPU_ERR_INVALID, "split_create: bad argument");
    FWGPU_HIP(hipSetDevice(r->device));
    std::unique_ptr<fwgpu_split> sp(new fwgpu_split());
    sp->owner = r;
    sp->n_cap = n_examples;
    const uint32_t F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0, R = F * r->cfg.ffm_k;
    sp->nlr = r->nn.n_layers ? r->cfg.num_combos : 1;
    sp->split_len = split_record_len(F, R, sp->nlr);
    sp->selfw_stride = (std::max<uint32_t>(4, (max_ffm_per_example + 3) & ~3u)) * std::max<uint32_t>(1, r->cfg.ffm_k);
    const uint32_t X = r->nn.n_layers ? r->nn.X : 0;
    const size_t bytes[5] = {(size_t)n_examples * sp->split_len * 4, (size_t)n_examples * sp->selfw_stride * 4,
                             (size_t)n_examples * 2 * 4 + 64, (size_t)n_examples * X * 4, (size_t)n_examples * X * 4};
    float **ptrs[5] = {&sp->d_split, &sp->d_selfw, &sp->d_g, &sp->d_x, &sp->d_dx};
    for (int i = 0; i < 5; i++) {
        if (!bytes[i]) continue;
        if (hipMalloc((void **)ptrs[i], bytes[i]) != hipSuccess) {
            fwgpu_split_free(sp.release());
            return fail(FWGPU_ERR_DEVICE, "split_create: allocation failed");
        }
    }
    *out = sp.release();
    return FWGPU_OK;
}

int fwgpu_split_free(fwgpu_split *sp) {
    if (!sp) return FWGPU_OK;
    for (float *q : {sp->d_split, sp->d_selfw, sp->d_g, sp->d_x, sp->d_dx})
        if (q) (void)hipFree(q);
    delete sp;
    return FWGPU_OK;
}

}  // extern "C"
namespace fwgpu {
static int split_params(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, int update, const SplitRanges &rg,
                        KernelParams &p, uint32_t &threads) {
    if (!r || !b || !sp) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (b->owner != r || sp->owner != r) return fail(FWGPU_ERR_INVALID, "batch / split buffers belong to another regressor");
    if (b->n > sp->n_cap) return fail(FWGPU_ERR_RANGE, "split buffers are smaller than the batch");
    int rc = prepare_launch(r, b, mode, update, p, threads);
    if (rc) return rc;
    if ((uint64_t)p.max_ffm * std::max<uint32_t>(1, p.k) > sp->selfw_stride)
        return fail(FWGPU_ERR_RANGE, "split buffers: an example has more FFM features than they were created for");
    p.split = sp->d_split;
    p.split_len = sp->split_len;
    p.split_nlr = sp->nlr;
    p.split_selfw = sp->d_selfw;
    p.selfw_stride = sp->selfw_stride;
    p.gbuf = sp->d_g;
    p.own_lo_ffm = rg.ffm_lo;
    p.own_hi_ffm = rg.ffm_hi;
    p.own_lo_lr = rg.lr_lo;
    p.own_hi_lr = rg.lr_hi;
    p.home_lo = rg.home_lo;
    p.home_hi = rg.home_hi;
    return FWGPU_OK;
}

int split_forward(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, const SplitRanges &rg, hipStream_t stream,
                  const OccBuffers *occ, uint32_t *max_ffm, uint32_t *max_lr) {
    if (b && b->n == 0) return FWGPU_OK;
    KernelParams p;
    uint32_t threads = 0;
    int rc = split_params(r, b, sp, mode, 0, rg, p, threads);
    if (rc) return rc;
    if (occ) {
        p.occ_ffm_key = occ->ffm_key;
        p.occ_ffm_desc = occ->ffm_desc;
        p.occ_lr_key = occ->lr_key;
        p.occ_lr_desc = occ->lr_desc;
    }
    if (max_ffm) *max_ffm = p.max_ffm;  // the slot strides of the occurrence lists
    if (max_lr) *max_lr = p.max_lr;
    FWGPU_HIP(hipMemsetAsync(b->work, 0, sizeof(uint32_t), stream));
    FWGPU_HIP(launch_example_phase(p, r->cfg.optimizer, 1, pick_grid(r, p, mode, threads), threads, stream));
    return FWGPU_OK;
}

// records [first, first + n) of `sp` -> predictions into pred (device, n floats) and general gradients / head inputs
int split_mid(fwgpu_regressor *r, fwgpu_split *sp, uint32_t first, uint32_t n, float *d_pred, bool head, hipStream_t stream) {
    if (!n) return FWGPU_OK;
    KernelParams p{};
    p.F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0;
    p.k = r->cfg.ffm_k;
    p.R = p.F * p.k;
    p.has_lr = r->cfg.wiring == FWGPU_WIRING_REGRESSOR;
    p.nn = r->nn;
    p.split = sp->d_split + (size_t)first * sp->split_len;
    p.split_len = sp->split_len;
    p.split_nlr = sp->nlr;
    p.pred = d_pred;
    p.gbuf = head ? sp->d_g + 2 * (size_t)first : sp->d_g + first;
    p.xbuf = head ? sp->d_x + (size_t)first * r->nn.X : nullptr;
    FWGPU_HIP(launch_split_mid(p, n, stream));
    return FWGPU_OK;
}

int split_update(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, const SplitRanges &rg, bool head, hipStream_t stream) {
    if (b && b->n == 0) return FWGPU_OK;
    KernelParams p;
    uint32_t threads = 0;
    int rc = split_params(r, b, sp, mode, 1, rg, p, threads);
    if (rc) return rc;
    p.dxbuf = head ? sp->d_dx : nullptr;
    FWGPU_HIP(hipMemsetAsync(b->work, 0, sizeof(uint32_t), stream));
    FWGPU_HIP(launch_example_phase(p, r->cfg.optimizer, 3, pick_grid(r, p, mode, threads), threads, stream));
    return FWGPU_OK;
}
}  // namespace fwgpu
extern "C" {

// One synchronous micro-batch on one GPU: all of b's examples are scored with the weights as they are, then all updates are
// applied (FWGPU_MODE_SEQUENTIAL: in example order, on one workgroup -- the deterministic mode the oracle's micro-batch
// mode is compared with; FWGPU_MODE_HOGWILD: concurrently).  Models with a deep head take the mini-batched head (head.hip).
int fwgpu_learn_batch_sync(fwgpu_regressor *r, fwgpu_batch *b, fwgpu_split *sp, int mode, void *stream_) {
    if (!r || !b || !sp) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (mode != FWGPU_MODE_SEQUENTIAL && mode != FWGPU_MODE_HOGWILD) return fail(FWGPU_ERR_INVALID, "unknown mode");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    FWGPU_HIP(hipSetDevice(r->device));
    SplitRanges rg;
    rg.home_hi = b->n;
    const bool head = r->nn.n_layers != 0;
    int rc = split_forward(r, b, sp, mode, rg, stream);
    if (rc == FWGPU_OK) rc = split_mid(r, sp, 0, b->n, b->pred, head, stream);
    if (rc == FWGPU_OK && head) rc = head_step(r, sp, 0, b->n, b->pred, /*update=*/true, stream);
    if (rc == FWGPU_OK) rc = split_update(r, b, sp, mode, rg, head, stream);
    return rc;
}

// ------------------------------------------------------------------ context cache (serving)
// Regressor::setup_cache / predict_with_cache (regressor.rs:397-423) with BlockFFM's cache (block_ffm.rs:442-782): the field
// sums ("contra fields") and self-pair corrections of the context's features are computed once, on the device, and every
// candidate then gathers only the rows of the features that are NOT in `features_present`.  BlockLR's cache is inert in the
// reference (prepare_forward_cache skips every feature: `hash & IS_NOT_SINGLE_MASK == 0` always holds for a masked hash,
// block_lr.rs:236-239), so LR entries are always all read, exactly like there.

static uint64_t cache_key(const fwgpu_ffm_entry &e) { return ((uint64_t)e.hash << 32) | e.contra_field_index; }  // regressor.rs:25-38

int fwgpu_setup_cache(fwgpu_regressor *r, const fwgpu_lr_entry *lr, uint32_t n_lr, const fwgpu_ffm_entry *ffm,
                      uint32_t n_ffm, fwgpu_block_cache **cache) {
    if (!r || !cache) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if ((n_lr && !lr) || (n_ffm && !ffm)) return fail(FWGPU_ERR_INVALID, "NULL entry buffer");
    if (r->nn.n_layers) return fail(FWGPU_ERR_INVALID, "the context cache does not cover models with a deep head");
    FWGPU_HIP(hipSetDevice(r->device));
    const uint32_t F = r->cfg.ffm_k ? r->cfg.ffm_num_fields : 0, R = F * r->cfg.ffm_k;
    fwgpu_block_cache *c = *cache;
    if (c && c->owner != r) return fail(FWGPU_ERR_INVALID, "cache belongs to another regressor");
    if (!c) {  // should_create (regressor.rs:415-417)
        c = new fwgpu_block_cache();
        c->owner = r;
        if (hipMalloc((void **)&c->d_T, ((size_t)F * R + F + 4) * sizeof(float)) != hipSuccess) {
            delete c;
            return fail(FWGPU_ERR_DEVICE, "setup_cache: allocation failed");
        }
        c->d_dcf = c->d_T + (size_t)F * R;
    }
    HostBatch hb;
    hb.clear();
    float pred = 0.0f;
    int rc = append_example(r, hb, lr, n_lr, ffm, n_ffm, 0.0f, 1.0f);
    if (rc == FWGPU_OK) rc = ensure_one(r, (uint32_t)hb.lr_hash.size(), (uint32_t)hb.ffm_hash.size());
    if (rc == FWGPU_OK) rc = batch_upload(r->one, hb, 0);
    if (rc == FWGPU_OK) {
        r->one->cache = nullptr;
        r->one->emit_T = c->d_T;
        r->one->emit_dcf = c->d_dcf;
        rc = run_batch(r, r->one, FWGPU_MODE_SEQUENTIAL, 0, 0);
        r->one->emit_T = r->one->emit_dcf = nullptr;
    }
    if (rc == FWGPU_OK && one_prediction(r, &pred) != FWGPU_OK) rc = fail(FWGPU_ERR_DEVICE, "setup_cache: launch failed");
    if (rc != FWGPU_OK) {
        if (!*cache) {
            (void)hipFree(c->d_T);
            delete c;
        }
        return rc;
    }
    c->present.clear();
    for (uint32_t i = 0; i < (r->cfg.ffm_k ? n_ffm : 0); i++) c->present.push_back(cache_key(ffm[i]));
    std::sort(c->present.begin(), c->present.end());
    c->present.erase(std::unique(c->present.begin(), c->present.end()), c->present.end());
    *cache = c;
    return FWGPU_OK;
}

int fwgpu_block_cache_free(fwgpu_block_cache *c) {
    if (!c) return FWGPU_OK;
    if (c->d_cover) (void)hipFree(c->d_cover);
    if (c->d_T) (void)hipFree(c->d_T);
    delete c;
    return FWGPU_OK;
}

// The FFM entries forward_with_cache still gathers: those whose (hash, contra_field_index) is not in features_present
// (block_ffm.rs:548, 600).  out may alias ffm.
int fwgpu_block_cache_filter(const fwgpu_block_cache *c, const fwgpu_ffm_entry *ffm, uint32_t n_ffm, fwgpu_ffm_entry *out, uint32_t *n_out) {
    if (!c || !n_out || (n_ffm && (!ffm || !out))) return fail(FWGPU_ERR_INVALID, "NULL argument");
    uint32_t m = 0;
    for (uint32_t i = 0; i < n_ffm; i++)
        if (!std::binary_search(c->present.begin(), c->present.end(), cache_key(ffm[i]))) out[m++] = ffm[i];
    *n_out = m;
    return FWGPU_OK;
}

int fwgpu_predict_with_cache(fwgpu_regressor *r, const fwgpu_block_cache *c, const fwgpu_lr_entry *lr, uint32_t n_lr,
                             const fwgpu_ffm_entry *ffm, uint32_t n_ffm, float *prediction) {
    if (!r || !c || !prediction) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (c->owner != r) return fail(FWGPU_ERR_INVALID, "cache belongs to another regressor");
    if ((n_lr && !lr) || (n_ffm && !ffm)) return fail(FWGPU_ERR_INVALID, "NULL entry buffer");
    FWGPU_HIP(hipSetDevice(r->device));
    std::vector<fwgpu_ffm_entry> rest(n_ffm);
    uint32_t m = 0;
    int rc = fwgpu_block_cache_filter(c, ffm, n_ffm, rest.data(), &m);
    if (rc) return rc;
    HostBatch hb;
    hb.clear();
    rc = append_example(r, hb, lr, n_lr, rest.data(), m, 0.0f, 1.0f);
    if (rc) return rc;
    rc = ensure_one(r, (uint32_t)hb.lr_hash.size(), (uint32_t)hb.ffm_hash.size());
    if (rc) return rc;
    rc = batch_upload(r->one, hb, 0);
    if (rc) return rc;
    r->one->cache = c;
    rc = run_batch(r, r->one, FWGPU_MODE_SEQUENTIAL, 0, 0);
    r->one->cache = nullptr;
    if (rc) return rc;
    return one_prediction(r, prediction);
}

static inline uint32_t present_bit(uint64_t key) { return (uint32_t)((key * 0x9e3779b97f4a7c15ULL) >> 52); }  // 12 bits

// Record batches with a context cache.  `record` is the context's own record (what fw_setup_cache parsed): the namespace
// slots it fills are the covered ones.  A request's record (context + candidate, parser.rs:195-211) then goes to the device
// whole, and the stage phase leaves out the FFM features of covered slots -- which are exactly the features
// fwgpu_block_cache_filter drops, PROVIDED the request passes fwgpu_block_cache_record_ok.
int fwgpu_block_cache_cover_record(fwgpu_block_cache *c, const fwgpu_translator_config *t, const uint32_t *record, uint32_t len) {
    if (!c || !t || !record) return fail(FWGPU_ERR_INVALID, "NULL argument");
    uint32_t nl = 0, nf = 0;
    int rc = count_record(t, record, len, &nl, &nf);
    if (rc) return rc;
    uint32_t max_ns = 0;
    for (uint32_t m = 0; m < t->field_off[t->n_fields]; m++) max_ns = std::max(max_ns, t->field_ns[m]);
    const uint32_t n_slots = std::min<uint32_t>(len > 3 ? len - 3 : 0, max_ns + 1);
    const size_t cover_words = (max_ns + 32) / 32;
    FWGPU_HIP(hipSetDevice(c->owner->device));
    if (c->d_cover && (c->cover.size() != cover_words || c->ctx_rec_cap < len)) {
        // (a refilled cache keeps its allocation when the new context fits: fw_setup_cache runs once per request)
        (void)hipFree(c->d_cover);
        c->d_cover = nullptr;
    }
    if (!c->d_cover) c->ctx_rec_cap = (size_t)len * 2;
    c->cover.assign(cover_words, 0);
    c->ctx_slots.assign(record + 3, record + 3 + n_slots);
    for (uint32_t ns = 0; ns < n_slots; ns++)
        if (record[3 + ns] != 0x80000000u) c->cover[ns >> 5] |= 1u << (ns & 31);  // NO_FEATURES (parser.rs:19)
    if (!c->d_cover) FWGPU_HIP(hipMalloc((void **)&c->d_cover, (cover_words + c->ctx_rec_cap) * 4));
    c->d_ctx_rec = c->d_cover + cover_words;
    c->ctx_rec.assign(record, record + len);
    std::vector<uint32_t> up(c->cover);
    up.insert(up.end(), record, record + len);
    FWGPU_HIP(hipMemcpy(c->d_cover, up.data(), up.size() * 4, hipMemcpyHostToDevice));
    c->present_bits.assign(64, 0);
    for (uint64_t key : c->present) {
        const uint32_t b = present_bit(key);
        c->present_bits[b >> 6] |= 1ull << (b & 63);
    }
    return FWGPU_OK;
}

// 1 when leaving out the covered slots of this request's record is what fwgpu_block_cache_filter would do with its
// translation: the request has not rewritten a covered slot (a namespace named again replaces the context's features in the
// record, parser.rs:318-326, while the cache still holds them), and none of its own FFM features equals a cached one
// (hash and field: the filter would drop it).  0: take the entry route (translate, filter) for this request.
int fwgpu_block_cache_record_ok(const fwgpu_block_cache *c, const fwgpu_translator_config *t, const uint32_t *record, uint32_t len) {
    return block_cache_record_ok(c, t, record, len, false);
}

}  // extern "C"
namespace fwgpu {
// delta: `record` holds the candidate's namespaces only (a covered slot it does not touch reads NO_FEATURES there)
int block_cache_record_ok(const fwgpu_block_cache *c, const fwgpu_translator_config *t, const uint32_t *record, uint32_t len, bool delta) {
    if (!c || !t || !record || !c->d_cover) return 0;
    const uint32_t n_slots = (uint32_t)c->ctx_slots.size();
    if (len < 3 + n_slots) return 0;
    for (uint32_t ns = 0; ns < n_slots; ns++)
        if (((c->cover[ns >> 5] >> (ns & 31)) & 1u) && record[3 + ns] != (delta ? 0x80000000u : c->ctx_slots[ns])) return 0;
    const uint32_t ffm_mask = ffm_hash_mask(t->ffm_bit_precision, t->ffm_k);
    for (uint32_t f = 0; f < t->n_fields; f++)
        for (uint32_t m = t->field_off[f]; m < t->field_off[f + 1]; m++) {
            const uint32_t ns = t->field_ns[m];
            if (3 + ns >= len) return 0;
            if (ns < n_slots && ((c->cover[ns >> 5] >> (ns & 31)) & 1u)) continue;
            const uint32_t w = record[3 + ns];
            auto hit = [&](uint32_t hash) {
                const uint64_t key = ((uint64_t)(hash & ffm_mask) << 32) | (f * t->ffm_k);
                const uint32_t b = present_bit(key);
                return ((c->present_bits[b >> 6] >> (b & 63)) & 1ull) && std::binary_search(c->present.begin(), c->present.end(), key);
            };
            if (!(w & 0x80000000u)) {
                if (hit(w)) return 0;
                continue;
            }
            const uint32_t start = (w >> 16) & 0x3fff, end = w & 0xffff;
            if (end > len || end < start) return 0;
            for (uint32_t o = start; o + 1 < end; o += 2)
                if (hit(record[o])) return 0;
        }
    return 1;
}
}  // namespace fwgpu
extern "C" {

// Predict-only launches of this batch start every example's field sums from the cache (NULL detaches it): an entry batch must
// then hold only the entries fwgpu_block_cache_filter leaves; a record batch holds whole requests that pass
// fwgpu_block_cache_record_ok, and the cache must know the context's record (fwgpu_block_cache_cover_record).
int fwgpu_batch_set_cache(fwgpu_batch *b, const fwgpu_block_cache *c) {
    if (!b) return fail(FWGPU_ERR_INVALID, "NULL batch");
    if (c && c->owner != b->owner) return fail(FWGPU_ERR_INVALID, "cache belongs to another regressor");
    if (c && b->records && !c->d_cover)
        return fail(FWGPU_ERR_INVALID, "a record batch needs a context cache that knows the context's record (fwgpu_block_cache_cover_record)");
    b->cache = c;
    return FWGPU_OK;
}

int fwgpu_batch_free(fwgpu_batch *b) {
    if (!b) return FWGPU_OK;
    if (b->dev) (void)hipFree(b->dev);
    if (b->tr_dev) (void)hipFree(b->tr_dev);
    if (b->host_block) (void)hipHostFree(b->host_block);
    if (b->work_ring)
        (void)hipFree(b->work_ring);
    else if (b->work)
        (void)hipFree(b->work);
    delete b;
    return FWGPU_OK;
}

int fwgpu_batch_size(const fwgpu_batch *b, uint32_t *n, uint64_t *n_lr, uint64_t *n_ffm) {
    if (!b) return fail(FWGPU_ERR_INVALID, "NULL batch");
    if (n) *n = b->n;
    if (n_lr) *n_lr = b->n_lr;
    if (n_ffm) *n_ffm = b->n_ffm;
    return FWGPU_OK;
}

int fwgpu_learn_batch(fwgpu_regressor *r, fwgpu_batch *b, int mode, int update, void *stream) {
    if (!r || !b) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (b->owner != r) return fail(FWGPU_ERR_INVALID, "batch belongs to another regressor");
    if (mode != FWGPU_MODE_SEQUENTIAL && mode != FWGPU_MODE_HOGWILD) return fail(FWGPU_ERR_INVALID, "unknown mode");
    return run_batch(r, b, mode, update, static_cast<hipStream_t>(stream));
}

int fwgpu_batch_predictions(fwgpu_batch *b, float *host_out, uint32_t n, void *stream) {
    if (!b) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (n > b->n) return fail(FWGPU_ERR_RANGE, "n > batch size");
    if (n == 0) return FWGPU_OK;
    if (!host_out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FWGPU_HIP(hipMemcpyAsync(host_out, b->pred, n * sizeof(float), hipMemcpyDeviceToHost, s));
    FWGPU_HIP(hipStreamSynchronize(s));
    return FWGPU_OK;
}

int fwgpu_batch_predictions_device(fwgpu_batch *b, void **dev_ptr) {
    if (!b || !dev_ptr) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *dev_ptr = b->pred;
    return FWGPU_OK;
}

// ------------------------------------------------------------------ tables

static int table_ptr(fwgpu_regressor *r, int which, float **p, uint64_t *n) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    switch (which) {
    case FWGPU_TABLE_LR: *p = r->d_lr; *n = r->lr_len * 2; return FWGPU_OK;
    case FWGPU_TABLE_FFM_W: *p = r->d_ffm_w; *n = r->ffm_len; return FWGPU_OK;
    case FWGPU_TABLE_FFM_ACC: *p = r->d_ffm_acc; *n = r->ffm_len; return FWGPU_OK;
    case FWGPU_TABLE_NN_W: *p = r->d_nn_w; *n = r->nn_len; return FWGPU_OK;
    case FWGPU_TABLE_NN_ACC: *p = r->d_nn_acc; *n = r->nn_len; return FWGPU_OK;
    }
    return fail(FWGPU_ERR_INVALID, "unknown table");
}

int fwgpu_table_len(fwgpu_regressor *r, int which, uint64_t *n_floats) {
    float *p;
    uint64_t n;
    int rc = table_ptr(r, which, &p, &n);
    if (rc) return rc;
    *n_floats = n;
    return FWGPU_OK;
}

int fwgpu_table_device_ptr(fwgpu_regressor *r, int which, void **dev_ptr) {
    float *p;
    uint64_t n;
    int rc = table_ptr(r, which, &p, &n);
    if (rc) return rc;
    if (!dev_ptr) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *dev_ptr = p;
    return FWGPU_OK;
}

int fwgpu_table_read(fwgpu_regressor *r, int which, uint64_t offset, uint64_t count, float *host_out) {
    float *p;
    uint64_t n;
    int rc = table_ptr(r, which, &p, &n);
    if (rc) return rc;
    if (offset > n || count > n - offset) return fail(FWGPU_ERR_RANGE, "table_read out of range");
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipDeviceSynchronize());
    if (count) FWGPU_HIP(hipMemcpy(host_out, p + offset, count * sizeof(float), hipMemcpyDeviceToHost));
    return FWGPU_OK;
}

int fwgpu_table_write(fwgpu_regressor *r, int which, uint64_t offset, uint64_t count, const float *host_in) {
    float *p;
    uint64_t n;
    int rc = table_ptr(r, which, &p, &n);
    if (rc) return rc;
    if (offset > n || count > n - offset) return fail(FWGPU_ERR_RANGE, "table_write out of range");
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipDeviceSynchronize());
    if (count) FWGPU_HIP(hipMemcpy(p + offset, host_in, count * sizeof(float), hipMemcpyHostToDevice));
    return FWGPU_OK;
}

int fwgpu_table_fill(fwgpu_regressor *r, int which, float value) {
    float *p;
    uint64_t n;
    int rc = table_ptr(r, which, &p, &n);
    if (rc) return rc;
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(launch_fill(p, n, value, 0));
    FWGPU_HIP(hipDeviceSynchronize());
    return FWGPU_OK;
}

int fwgpu_table_checksum(fwgpu_regressor *r, int which, uint64_t *checksum) {
    float *p;
    uint64_t n;
    int rc = table_ptr(r, which, &p, &n);
    if (rc) return rc;
    FWGPU_HIP(hipSetDevice(r->device));
    unsigned long long *d = nullptr;
    FWGPU_HIP(hipMalloc((void **)&d, sizeof(unsigned long long)));
    hipError_t e = launch_checksum(p, n, d, 0);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FWGPU_ERR_DEVICE, std::string("checksum: ") + hipGetErrorString(e));
    *checksum = h;
    return FWGPU_OK;
}

int fwgpu_delta_start(const void *table, const void *snapshot, void *local_delta, void *summed_delta, uint64_t n_floats,
                      float scale, void *stream) {
    if (!table || !snapshot || !local_delta || !summed_delta) return fail(FWGPU_ERR_INVALID, "NULL argument");
    FWGPU_HIP(launch_delta_start(static_cast<const float *>(table), static_cast<const float *>(snapshot),
                                 static_cast<float *>(local_delta), static_cast<float *>(summed_delta), n_floats, scale,
                                 static_cast<hipStream_t>(stream)));
    return FWGPU_OK;
}

int fwgpu_delta_finish(void *table, void *snapshot, const void *local_delta, const void *summed_delta, uint64_t n_floats,
                       void *stream) {
    if (!table || !snapshot || !local_delta || !summed_delta) return fail(FWGPU_ERR_INVALID, "NULL argument");
    FWGPU_HIP(launch_delta_finish(static_cast<float *>(table), static_cast<float *>(snapshot),
                                  static_cast<const float *>(local_delta), static_cast<const float *>(summed_delta),
                                  n_floats, static_cast<hipStream_t>(stream)));
    return FWGPU_OK;
}

int fwgpu_debug_placement(const fwgpu_regressor *r, int *tries, float *ms_fastest, float *ms_slowest) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    if (tries) *tries = r->placement_tries;
    if (ms_fastest) *ms_fastest = r->placement_ms_lo;
    if (ms_slowest) *ms_slowest = r->placement_ms_hi;
    return FWGPU_OK;
}

int fwgpu_debug_phase_ticks(fwgpu_regressor *r, int enable, uint64_t *out16) {
    if (!r) return fail(FWGPU_ERR_INVALID, "NULL regressor");
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipDeviceSynchronize());
    if (out16) {
        if (r->d_ticks)
            FWGPU_HIP(hipMemcpy(out16, r->d_ticks, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        else
            memset(out16, 0, 16 * sizeof(uint64_t));
    }
    if (enable) {
        if (!r->d_ticks) FWGPU_HIP(hipMalloc((void **)&r->d_ticks, 16 * sizeof(uint64_t)));
        FWGPU_HIP(hipMemset(r->d_ticks, 0, 16 * sizeof(uint64_t)));
    } else if (r->d_ticks) {
        (void)hipFree(r->d_ticks);
        r->d_ticks = nullptr;
    }
    return FWGPU_OK;
}

int fwgpu_debug_coherence_probe(int device, int use_sc1, uint32_t iters, uint32_t *stale_words, uint32_t *timeouts) {
    if (!stale_words || !timeouts) return fail(FWGPU_ERR_INVALID, "NULL argument");
    FWGPU_HIP(hipSetDevice(device));
    unsigned *d = nullptr;
    FWGPU_HIP(hipMalloc((void **)&d, 512 * sizeof(unsigned)));
    hipError_t e = hipMemset(d, 0, 512 * sizeof(unsigned));
    if (e == hipSuccess) e = launch_coherence_probe(d, use_sc1, iters, 16, 0);
    unsigned out[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(out, d + 400, sizeof(out), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FWGPU_ERR_DEVICE, std::string("coherence probe: ") + hipGetErrorString(e));
    *stale_words = out[0];
    *timeouts = out[1];
    return FWGPU_OK;
}

// ------------------------------------------------------------------ weight blob (regressor.rs:426-469)

static uint64_t serialized_elems(const fwgpu_regressor *r) {
    // sum of get_serialized_len(): block_lr.rs:253-255 (weights_len), block_ffm.rs:831-833 (ffm_weights_len)
    return r->lr_len + r->ffm_len + r->nn_len;  // + block_neural.rs:426-428 (weights_len of every dense layer)
}

int fwgpu_serialized_len(fwgpu_regressor *r, uint64_t *n_bytes) {
    if (!r || !n_bytes) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const bool sgd = r->cfg.optimizer == FWGPU_OPT_SGD;
    // SGD's PerWeightStore is PhantomData (0 bytes): optimizer.rs:20
    *n_bytes = 8 + r->lr_len * (sgd ? 4 : 8) + r->ffm_len * (sgd ? 4 : 8) + r->nn_len * (sgd ? 4 : 8);
    return FWGPU_OK;
}

int fwgpu_write_weights(fwgpu_regressor *r, uint8_t *buf, uint64_t cap, uint64_t *written) {
    uint64_t need = 0;
    int rc = fwgpu_serialized_len(r, &need);
    if (rc) return rc;
    if (!buf || cap < need) return fail(FWGPU_ERR_RANGE, "write_weights: buffer too small");
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipDeviceSynchronize());
    const bool sgd = r->cfg.optimizer == FWGPU_OPT_SGD;
    const uint64_t total = serialized_elems(r);
    memcpy(buf, &total, 8);  // little-endian host
    uint8_t *o = buf + 8;
    if (!sgd) {
        FWGPU_HIP(hipMemcpy(o, r->d_lr, r->lr_len * 8, hipMemcpyDeviceToHost));
        o += r->lr_len * 8;
    } else {
        std::vector<float> tmp(r->lr_len * 2);
        FWGPU_HIP(hipMemcpy(tmp.data(), r->d_lr, r->lr_len * 8, hipMemcpyDeviceToHost));
        float *dst = reinterpret_cast<float *>(o);
        for (uint64_t i = 0; i < r->lr_len; i++) dst[i] = tmp[2 * i];
        o += r->lr_len * 4;
    }
    if (r->ffm_len) {
        FWGPU_HIP(hipMemcpy(o, r->d_ffm_w, r->ffm_len * 4, hipMemcpyDeviceToHost));
        o += r->ffm_len * 4;
        if (!sgd) {
            FWGPU_HIP(hipMemcpy(o, r->d_ffm_acc, r->ffm_len * 4, hipMemcpyDeviceToHost));
            o += r->ffm_len * 4;
        }
    }
    // dense layers in block order, each: weights then optimizer state (block_neural.rs:430-438)
    for (uint32_t l = 0; r->nn.n_layers && l <= r->nn.n_layers; l++) {
        const size_t len = ((size_t)r->nn.in[l] + 1) * r->nn.out[l];
        FWGPU_HIP(hipMemcpy(o, r->d_nn_w + r->nn.off[l], len * 4, hipMemcpyDeviceToHost));
        o += len * 4;
        if (!sgd) {
            FWGPU_HIP(hipMemcpy(o, r->d_nn_acc + r->nn.off[l], len * 4, hipMemcpyDeviceToHost));
            o += len * 4;
        }
    }
    if (written) *written = (uint64_t)(o - buf);
    return FWGPU_OK;
}

int fwgpu_read_weights(fwgpu_regressor *r, const uint8_t *buf, uint64_t len) {
    uint64_t need = 0;
    int rc = fwgpu_serialized_len(r, &need);
    if (rc) return rc;
    if (!buf || len < need) return fail(FWGPU_ERR_FORMAT, "read_weights: blob shorter than the model");
    uint64_t total = 0;
    memcpy(&total, buf, 8);
    // regressor.rs:452-457: "Lenghts of weights array in regressor file differ"
    if (total != serialized_elems(r)) return fail(FWGPU_ERR_FORMAT, "read_weights: weight count differs from the model's");
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipDeviceSynchronize());
    const bool sgd = r->cfg.optimizer == FWGPU_OPT_SGD;
    const uint8_t *o = buf + 8;
    if (!sgd) {
        FWGPU_HIP(hipMemcpy(r->d_lr, o, r->lr_len * 8, hipMemcpyHostToDevice));
        o += r->lr_len * 8;
    } else {
        std::vector<float> tmp(r->lr_len * 2, 0.0f);
        const float *src = reinterpret_cast<const float *>(o);
        for (uint64_t i = 0; i < r->lr_len; i++) tmp[2 * i] = src[i];
        FWGPU_HIP(hipMemcpy(r->d_lr, tmp.data(), r->lr_len * 8, hipMemcpyHostToDevice));
        o += r->lr_len * 4;
    }
    if (r->ffm_len) {
        FWGPU_HIP(hipMemcpy(r->d_ffm_w, o, r->ffm_len * 4, hipMemcpyHostToDevice));
        o += r->ffm_len * 4;
        if (!sgd) {
            FWGPU_HIP(hipMemcpy(r->d_ffm_acc, o, r->ffm_len * 4, hipMemcpyHostToDevice));
            o += r->ffm_len * 4;
        }
    }
    for (uint32_t l = 0; r->nn.n_layers && l <= r->nn.n_layers; l++) {  // block_neural.rs:440-448
        const size_t len = ((size_t)r->nn.in[l] + 1) * r->nn.out[l];
        FWGPU_HIP(hipMemcpy(r->d_nn_w + r->nn.off[l], o, len * 4, hipMemcpyHostToDevice));
        o += len * 4;
        if (!sgd) {
            FWGPU_HIP(hipMemcpy(r->d_nn_acc + r->nn.off[l], o, len * 4, hipMemcpyHostToDevice));
            o += len * 4;
        }
    }
    return FWGPU_OK;
}

}  // extern "C"
