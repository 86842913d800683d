// Host side: record -> FeatureBuffer translation (feature_buffer.rs:138-338) for primitive namespaces,
// murmur3 as the parser uses it (parser.rs:82-87, 382-385) and the synthetic record generator for the
// BASELINE.json configurations.  Hash arithmetic here is integer-only and must be bit-exact.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <random>

#include "fwgpu_internal.h"

namespace fwgpu {

static constexpr uint32_t kHeaderLen = 3;                // parser.rs:13 HEADER_LEN
static constexpr uint32_t kIsNotSingleMask = 1u << 31;   // parser.rs:17
static constexpr uint32_t kMask31 = ~kIsNotSingleMask;   // parser.rs:18
static constexpr uint32_t kNoFeatures = kIsNotSingleMask;  // parser.rs:19
static constexpr uint32_t kVowpalFnvPrime = 16777619u;   // feature_buffer.rs:6
static constexpr uint32_t kConstantHash = 11650396u;     // feature_buffer.rs:8

struct HV {
    uint32_t hash;
    float value;
};

// feature_reader! (feature_buffer.rs:47-108): decode one namespace slot of a record.
static int read_namespace(const uint32_t *rec, uint32_t rec_len, uint32_t ns, bool is_f32, std::vector<HV> &out) {
    out.clear();
    if (ns + kHeaderLen >= rec_len) return fail(FWGPU_ERR_FORMAT, "record: namespace slot beyond the record");
    const uint32_t first_token = rec[ns + kHeaderLen];
    if ((first_token & kIsNotSingleMask) == 0) {
        out.push_back({first_token, 1.0f});
        return FWGPU_OK;
    }
    const uint32_t start = (first_token >> 16) & 0x3fff, end = first_token & 0xffff;
    if (end > rec_len) return fail(FWGPU_ERR_FORMAT, "record: feature range beyond the record");
    for (uint32_t o = start; o + 1 < end; o += 2) {
        float v = 1.0f;
        if (!is_f32) memcpy(&v, &rec[o + 1], 4);
        out.push_back({rec[o], v});
    }
    return FWGPU_OK;
}

uint32_t lr_hash_mask(uint32_t bit_precision) { return (uint32_t)((1ull << bit_precision) - 1); }  // feature_buffer.rs:140

uint32_t ffm_hash_mask(uint32_t ffm_bits, uint32_t ffm_k) {  // feature_buffer.rs:141-148
    uint32_t bits = 0;
    while (ffm_k > (1u << bits)) bits++;
    return ((uint32_t)((1ull << ffm_bits) - 1)) ^ ((1u << bits) - 1);
}

int translate_record(const fwgpu_translator_config *t, const uint32_t *rec, uint32_t rec_len,
                     std::vector<fwgpu_lr_entry> &lr, std::vector<fwgpu_ffm_entry> &ffm, float *label,
                     float *importance) {
    lr.clear();
    ffm.clear();
    if (rec_len < kHeaderLen) return fail(FWGPU_ERR_FORMAT, "record shorter than its header");
    *label = (float)rec[1];            // feature_buffer.rs:187
    memcpy(importance, &rec[2], 4);    // feature_buffer.rs:188-189
    const uint32_t lr_mask = lr_hash_mask(t->bit_precision), ffm_mask = ffm_hash_mask(t->ffm_bit_precision, t->ffm_k);
    thread_local std::vector<HV> a, b, cur;
    for (uint32_t ci = 0; ci < t->n_combos; ci++) {  // feature_buffer.rs:194-267
        const uint32_t s = t->combo_off[ci], e = t->combo_off[ci + 1];
        if (e <= s) return fail(FWGPU_ERR_INVALID, "translator: empty combo");
        const float cw = t->combo_weight[ci];
        int rc = read_namespace(rec, rec_len, t->combo_ns[s], t->combo_ns_f32[s] != 0, a);
        if (rc) return rc;
        if (e - s == 1) {
            for (const HV &x : a) lr.push_back({x.hash & lr_mask, x.value * cw, ci});
            continue;
        }
        std::vector<HV> *in = &a, *out = &b;
        for (uint32_t mi = s + 1; mi < e; mi++) {  // 235-258: h = h_prev * FNV (wrapping) ^ h_next
            rc = read_namespace(rec, rec_len, t->combo_ns[mi], t->combo_ns_f32[mi] != 0, cur);
            if (rc) return rc;
            out->clear();
            for (const HV &x : *in) {
                const uint32_t half_hash = x.hash * kVowpalFnvPrime;
                for (const HV &y : cur) out->push_back({y.hash ^ half_hash, x.value * y.value});
            }
            std::swap(in, out);
        }
        for (const HV &x : *in) lr.push_back({x.hash & lr_mask, x.value * cw, ci});
    }
    if (t->add_constant_feature) lr.push_back({kConstantHash & lr_mask, 1.0f, t->n_combos});  // 270-276
    if (t->ffm_k > 0) {  // 279-335
        for (uint32_t f = 0; f < t->n_fields; f++)
            for (uint32_t mi = t->field_off[f]; mi < t->field_off[f + 1]; mi++) {
                int rc = read_namespace(rec, rec_len, t->field_ns[mi], t->field_ns_f32[mi] != 0, cur);
                if (rc) return rc;
                for (const HV &y : cur) ffm.push_back({y.hash & ffm_mask, y.value, f * t->ffm_k});
            }
    }
    return FWGPU_OK;
}

int count_record(const fwgpu_translator_config *t, const uint32_t *rec, uint32_t rec_len, uint32_t *n_lr, uint32_t *n_ffm) {
    return count_record(t, rec, rec_len, nullptr, 0, n_lr, n_ffm);
}

int count_record(const fwgpu_translator_config *t, const uint32_t *rec, uint32_t rec_len, const uint32_t *ctx, uint32_t ctx_len,
                 uint32_t *n_lr, uint32_t *n_ffm) {
    if (rec_len < kHeaderLen || (ctx && ctx_len < kHeaderLen)) return fail(FWGPU_ERR_FORMAT, "record shorter than its header");
    auto cnt = [&](uint32_t ns, uint32_t *c) -> int {
        if (ns + kHeaderLen >= rec_len) return fail(FWGPU_ERR_FORMAT, "record: namespace slot beyond the record");
        uint32_t w = rec[ns + kHeaderLen], len = rec_len;
        if (ctx && w == kNoFeatures) {  // a candidate-only record inherits the namespace from its context's record
            if (ns + kHeaderLen >= ctx_len) return fail(FWGPU_ERR_FORMAT, "record: namespace slot beyond the context's record");
            w = ctx[ns + kHeaderLen];
            len = ctx_len;
        }
        if ((w & kIsNotSingleMask) == 0) {
            *c = 1;
            return FWGPU_OK;
        }
        const uint32_t start = (w >> 16) & 0x3fff, end = w & 0xffff;
        if (end < start || end > len || ((end - start) & 1) || (end > start && start < kHeaderLen))
            return fail(FWGPU_ERR_FORMAT, "record: malformed feature range");
        *c = (end - start) / 2;
        return FWGPU_OK;
    };
    uint64_t nl = 0, nf = 0;
    for (uint32_t ci = 0; ci < t->n_combos; ci++) {
        uint64_t prod = 1;
        for (uint32_t m = t->combo_off[ci]; m < t->combo_off[ci + 1]; m++) {
            uint32_t c;
            int rc = cnt(t->combo_ns[m], &c);
            if (rc) return rc;
            prod *= c;
        }
        nl += prod;
    }
    if (t->add_constant_feature) nl++;
    if (t->ffm_k > 0)
        for (uint32_t m = 0; m < t->field_off[t->n_fields]; m++) {
            uint32_t c;
            int rc = cnt(t->field_ns[m], &c);
            if (rc) return rc;
            nf += c;
        }
    if (nl > 1000000 || nf > 1000000) return fail(FWGPU_ERR_RANGE, "record translates to too many entries");
    *n_lr = (uint32_t)nl;
    *n_ffm = (uint32_t)nf;
    return FWGPU_OK;
}

int check_translator(const fwgpu_regressor *r, const fwgpu_translator_config *t) {
    if (!r || !t) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (t->bit_precision != r->cfg.bit_precision || t->ffm_k != r->cfg.ffm_k ||
        (t->ffm_k && t->ffm_bit_precision != r->cfg.ffm_bit_precision))
        return fail(FWGPU_ERR_INVALID, "translator: bit_precision / ffm_k / ffm_bit_precision differ from the regressor's");
    if (t->n_combos + (t->add_constant_feature ? 1 : 0) != r->cfg.num_combos && r->cfg.wiring == FWGPU_WIRING_REGRESSOR)
        return fail(FWGPU_ERR_INVALID, "translator: n_combos (+constant) differs from the regressor's num_combos");
    if (r->cfg.ffm_k > 0 && t->n_fields != r->cfg.ffm_num_fields)
        return fail(FWGPU_ERR_INVALID, "translator: n_fields differs from the regressor's ffm_num_fields");
    return FWGPU_OK;
}

// murmur3 x86_32 == fasthash::murmur3::hash32_with_seed (parser.rs:83, 382-385)
static uint32_t murmur3_32(const uint8_t *data, size_t len, uint32_t seed) {
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
    const size_t nblocks = len / 4;
    for (size_t i = 0; i < nblocks; i++) {
        uint32_t k1;
        memcpy(&k1, data + 4 * i, 4);
        k1 *= c1;
        k1 = (k1 << 15) | (k1 >> 17);
        k1 *= c2;
        h1 ^= k1;
        h1 = (h1 << 13) | (h1 >> 19);
        h1 = h1 * 5 + 0xe6546b64u;
    }
    const uint8_t *tail = data + nblocks * 4;
    uint32_t k1 = 0;
    switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; [[fallthrough]];
    case 2: k1 ^= (uint32_t)tail[1] << 8; [[fallthrough]];
    case 1:
        k1 ^= tail[0];
        k1 *= c1;
        k1 = (k1 << 15) | (k1 >> 17);
        k1 *= c2;
        h1 ^= k1;
    }
    h1 ^= (uint32_t)len;
    h1 ^= h1 >> 16;
    h1 *= 0x85ebca6bu;
    h1 ^= h1 >> 13;
    h1 *= 0xc2b2ae35u;
    h1 ^= h1 >> 16;
    return h1;
}

// ---- synthetic stream -------------------------------------------------------------------------

static inline uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t &s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }

// Bounded Zipf(s) over {0..n-1} by inverting the continuous approximation of the CDF
// (rank r has mass ~ r^-s): cheap, deterministic, no tables.
static inline uint32_t zipf_id(uint64_t &st, double s, uint32_t n) {
    const double u = u01(st);
    double x;
    if (fabs(s - 1.0) < 1e-9) {
        x = exp(u * log((double)n + 1.0));
    } else {
        const double a = 1.0 - s;
        const double hi = pow((double)n + 1.0, a);
        x = pow(1.0 + u * (hi - 1.0), 1.0 / a);
    }
    uint32_t id = (uint32_t)(x - 1.0);
    return id >= n ? n - 1 : id;
}

static inline uint32_t poisson(uint64_t &st, double mean) {
    if (mean <= 0.0) return 0;
    const double L = exp(-mean);
    double p = 1.0;
    uint32_t k = 0;
    do {
        k++;
        p *= u01(st);
    } while (p > L);
    return k - 1;
}

}  // namespace fwgpu

using namespace fwgpu;

extern "C" {

uint32_t fwgpu_murmur3_32(const uint8_t *data, size_t len, uint32_t seed) { return murmur3_32(data, len, seed); }

uint32_t fwgpu_lr_hash_mask(uint32_t bit_precision) { return lr_hash_mask(bit_precision); }
uint32_t fwgpu_ffm_hash_mask(uint32_t ffm_bit_precision, uint32_t ffm_k) { return ffm_hash_mask(ffm_bit_precision, ffm_k); }

int fwgpu_translate(const fwgpu_translator_config *t, const uint32_t *record, uint32_t record_len, fwgpu_lr_entry *lr_out, uint32_t lr_cap, uint32_t *n_lr, fwgpu_ffm_entry *ffm_out, uint32_t ffm_cap,
                    uint32_t *n_ffm, float *label, float *importance) {
    if (!record || !n_lr || !n_ffm || !label || !importance) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (!t) return fail(FWGPU_ERR_INVALID, "NULL argument");
    std::vector<fwgpu_lr_entry> lr;
    std::vector<fwgpu_ffm_entry> ffm;
    int rc = translate_record(t, record, record_len, lr, ffm, label, importance);
    if (rc) return rc;
    if (lr.size() > lr_cap || ffm.size() > ffm_cap) return fail(FWGPU_ERR_RANGE, "translate: output buffer too small");
    if (!lr.empty()) memcpy(lr_out, lr.data(), lr.size() * sizeof(fwgpu_lr_entry));
    if (!ffm.empty()) memcpy(ffm_out, ffm.data(), ffm.size() * sizeof(fwgpu_ffm_entry));
    *n_lr = (uint32_t)lr.size();
    *n_ffm = (uint32_t)ffm.size();
    return FWGPU_OK;
}

int fwgpu_batch_from_records(fwgpu_regressor *r, const fwgpu_translator_config *t, const uint32_t *records,
                             const uint64_t *rec_off, uint32_t n, fwgpu_batch **out) {
    if (!out || (n && (!records || !rec_off))) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = check_translator(r, t);
    if (rc) return rc;
    HostBatch hb;
    hb.clear();
    std::vector<fwgpu_lr_entry> lr;
    std::vector<fwgpu_ffm_entry> ffm;
    for (uint32_t i = 0; i < n; i++) {
        float label, imp;
        const uint32_t len = (uint32_t)(rec_off[i + 1] - rec_off[i]);
        rc = translate_record(t, records + rec_off[i], len, lr, ffm, &label, &imp);
        if (rc) return rc;
        rc = append_example(r, hb, lr.data(), (uint32_t)lr.size(), ffm.data(), (uint32_t)ffm.size(), label, imp);
        if (rc) return rc;
    }
    fwgpu_batch *b = nullptr;
    rc = batch_alloc(r, n, hb.lr_hash.size(), hb.ffm_hash.size(), &b);
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

int fwgpu_synth_records(const fwgpu_synth_config *c, uint64_t first_example, uint32_t n, uint32_t *records,
                        uint64_t records_cap, uint64_t *rec_off, uint64_t *n_words) {
    if (!c || !n_words) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (c->n_namespaces == 0 || c->n_namespaces > 255 || c->ids_per_ns == 0) return fail(FWGPU_ERR_INVALID, "bad synth config");
    const uint32_t N = c->n_namespaces;
    // namespace seeds as the parser computes them: murmur3(vwname, 0) (parser.rs:82-87); names "A0".."A254"
    std::vector<uint32_t> ns_seed(N);
    for (uint32_t i = 0; i < N; i++) {
        char name[16];
        int l = snprintf(name, sizeof(name), "N%u", i);
        ns_seed[i] = murmur3_32(reinterpret_cast<const uint8_t *>(name), (size_t)l, 0);
    }
    uint64_t w = 0;
    std::vector<uint32_t> tmp;
    for (uint32_t ex = 0; ex < n; ex++) {
        // every example has its own stream, so any sub-range of the stream can be generated independently
        uint64_t st = c->seed * 0x9e3779b97f4a7c15ULL + (first_example + ex) * 0xd1342543de82ef95ULL + 1;
        splitmix(st);
        tmp.assign(kHeaderLen + N, 0);
        double teacher = 0.0;
        for (uint32_t ns = 0; ns < N; ns++) {
            const uint32_t cnt = 1 + poisson(st, c->mean_extra);
            bool single = cnt == 1;
            uint32_t hashes[64];
            float vals[64];
            const uint32_t m = std::min<uint32_t>(cnt, 64);
            for (uint32_t j = 0; j < m; j++) {
                const uint32_t id = zipf_id(st, c->zipf_s, c->ids_per_ns);
                char name[16];
                int l = snprintf(name, sizeof(name), "%u", id);
                hashes[j] = murmur3_32(reinterpret_cast<const uint8_t *>(name), (size_t)l, ns_seed[ns]) & kMask31;
                vals[j] = 1.0f;
                if (u01(st) < c->p_weighted) {
                    vals[j] = (float)(0.5 + 1.5 * u01(st));
                    single = false;
                }
                // fixed random teacher: a per-(namespace,id) score from the hash bits
                uint64_t hs = ((uint64_t)ns << 32) ^ hashes[j] ^ (c->seed << 17);
                teacher += ((double)(splitmix(hs) >> 40) / 16777216.0 - 0.5) * vals[j];
            }
            if (single) {
                tmp[kHeaderLen + ns] = hashes[0];
            } else {
                const uint32_t start = (uint32_t)tmp.size();
                for (uint32_t j = 0; j < m; j++) {
                    tmp.push_back(hashes[j]);
                    uint32_t vb;
                    memcpy(&vb, &vals[j], 4);
                    tmp.push_back(vb);
                }
                const uint32_t end = (uint32_t)tmp.size();
                if (start > 0x3fff || end > 0xffff) return fail(FWGPU_ERR_RANGE, "synthetic record too long for the slot encoding");
                tmp[kHeaderLen + ns] = kIsNotSingleMask | (start << 16) | end;
            }
        }
        const double pr = 1.0 / (1.0 + exp(-1.5 * teacher / sqrt((double)N * (1.0 + c->mean_extra)) * 3.0));
        tmp[0] = (uint32_t)tmp.size();
        tmp[1] = u01(st) < pr ? 1u : 0u;  // parser: label 1 -> 1, -1 -> 0
        const float one = 1.0f;
        memcpy(&tmp[2], &one, 4);
        if (rec_off) rec_off[ex] = w;
        if (records) {
            if (w + tmp.size() > records_cap) return fail(FWGPU_ERR_RANGE, "synth: records buffer too small");
            memcpy(records + w, tmp.data(), tmp.size() * 4);
        }
        w += tmp.size();
    }
    if (rec_off) rec_off[n] = w;
    *n_words = w;
    (void)kNoFeatures;
    return FWGPU_OK;
}

}  // extern "C"
