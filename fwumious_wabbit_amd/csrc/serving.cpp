// The reference's serving FFI, same symbols and signatures (lib.rs:150-234): new_fw_predictor_prototype, clone_lite,
// fw_predict, fw_predict_with_cache, fw_setup_cache, free_predictor -- a binary that links libfw.so can link this library
// instead.  Each call parses the VW text, translates the record and runs the example kernel on the device.
//
// Context cache (lib.rs:88-148, block_ffm.rs:442-782, block_lr.rs:165-255): fw_setup_cache scans the context line once
// (fwgpu_parse_prefix: a request's scan resumes at the context's last token boundary instead of going over the context's text
// again, and gives the very record next_vowpal_with_cache builds from context + candidate, parser.rs:195-211), translates it
// and has the device compute its features' field sums once (fwgpu_setup_cache).  fw_predict_with_cache translates the request's
// record on the host and ships only the entries that are not in the cache.  fwgpu_predictor_predict_batch sends all candidates
// of a request in one launch, as records: the kernel's own translation leaves out the namespaces the cache covers
// (fwgpu_block_cache_cover_record); a request in which some candidate names a context namespace again, or carries a feature
// equal to a cached one, takes the entry route (host translation + fwgpu_block_cache_filter), which is the reference's rule
// for those.  Models with a deep head keep the uncached route (whole line scored), which gives the same result.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fw_ffi.h"
#include "fwgpu_internal.h"

using namespace fwgpu;

namespace {

struct SharedModel {  // what clone_lite shares: the immutable regressor and the descriptions it was built from
    fwgpu_regressor *re = nullptr;
    fwgpu_vwmap *vw = nullptr;
    fwgpu_model_instance *mi = nullptr;
    fwgpu_translator_config tr{};
    std::mutex mu;  // the regressor's single-example staging buffers are shared
    fwgpu_batch *batch = nullptr;  // device buffers of fwgpu_predictor_predict_batch, grown on demand and reused
    fwgpu_batch *one = nullptr;    // single requests: a host-mapped batch (no copy calls, no memset: launch + one synchronisation)
    ~SharedModel() {
        if (batch) fwgpu_batch_free(batch);
        if (one) fwgpu_batch_free(one);
        if (re) fwgpu_free(re);
        if (vw) fwgpu_vwmap_free(vw);
        if (mi) fwgpu_mi_free(mi);
    }
};

// shellwords::split (lib.rs:161): whitespace separated words, '...' and "..." quoting, backslash escapes
std::vector<std::string> shell_split(const char *s) {
    std::vector<std::string> out;
    std::string cur;
    bool in_word = false;
    char quote = 0;
    for (const char *p = s; *p; ++p) {
        const char c = *p;
        if (quote) {
            if (c == quote) quote = 0;
            else if (c == '\\' && quote == '"' && p[1]) cur += *++p;
            else cur += c;
        } else if (c == '\'' || c == '"') {
            quote = c;
            in_word = true;
        } else if (c == '\\' && p[1]) {
            cur += *++p;
            in_word = true;
        } else if (c == ' ' || c == '\t' || c == '\n') {
            if (in_word) out.push_back(cur);
            cur.clear();
            in_word = false;
        } else {
            cur += c;
            in_word = true;
        }
    }
    if (in_word) out.push_back(cur);
    return out;
}

}  // namespace

// one parser thread's share of a batched request (kept between requests: no fresh pages to fault in, no parser to build)
struct Part {
    std::vector<uint32_t> words;
    std::vector<uint64_t> len;
    std::vector<uint32_t> slot;
    // entry route: translated on the host, filtered (forward_with_cache gathers only what features_present lacks)
    HostBatch hb;
    bool hb_ok = true;
    RecordStats stats;  // record route: what record_batch_upload would otherwise count again, serially
    bool stats_ok = true;
    bool entry_route = false;  // some candidate of this slice is not covered record-wise (fwgpu_block_cache_record_ok)
    fwgpu_parser *parser = nullptr;  // VowpalParser is not thread safe (clone_lite's reason): one per thread
    std::vector<uint32_t> rec;
    void reset() {
        words.clear();
        len.clear();
        slot.clear();
        hb.clear();
        hb_ok = stats_ok = true;
        entry_route = false;
        stats = RecordStats();
    }
    ~Part() {
        if (parser) fwgpu_parser_free(parser);
    }
};

struct FfiPredictor {
    std::shared_ptr<SharedModel> model;
    std::vector<std::unique_ptr<Part>> parts;
    Workers workers;  // the parser threads of batched requests, kept between requests: starting 31 threads costs more than scanning a 20 000-candidate request on them takes
    fwgpu_parser *parser = nullptr;
    std::string cached_text;  // PredictorCache.input_buffer_size bytes of the context line (lib.rs:64-67)
    fwgpu_parse_prefix *prefix = nullptr;  // the same bytes, scanned
    bool has_cache = false;
    bool delta_ok = false;  // the scanned context is the context's own record: requests can travel as candidate-only records
    fwgpu_block_cache *cache = nullptr;  // PredictorCache.blocks: the context's field sums, on the device
    std::vector<uint32_t> record;
    std::vector<fwgpu_lr_entry> lr;
    std::vector<fwgpu_ffm_entry> ffm;
    ~FfiPredictor() {
        if (prefix) fwgpu_parse_prefix_free(prefix);
        if (parser) fwgpu_parser_free(parser);
        if (cache) fwgpu_block_cache_free(cache);
    }
};

namespace {

constexpr float kEofErrorCode = -1.0f, kExceptionErrorCode = -1.0f;  // lib.rs:47-48

// parse -> translate -> predict (lib.rs:70-86).  The record goes to the device as it is (one small copy) and is translated
// in the example kernel's stage phase, like every other record batch; fwgpu_predictor_predict_batch is the same path with n > 1.
float predict_line(FfiPredictor *p, const char *prefix, size_t prefix_len, const char *text, size_t len) {
    uint32_t n_words = 0;
    p->record.resize(std::max<size_t>(p->record.size(), 4096));
    int rc;
    for (;;) {
        rc = prefix && p->prefix
                 ? fwgpu_parser_parse_after_prefix(p->parser, p->prefix, text, len, p->record.data(), (uint32_t)p->record.size(), &n_words)
                 : fwgpu_parser_parse_with_prefix(p->parser, prefix, prefix_len, text, len, p->record.data(), (uint32_t)p->record.size(), &n_words);
        if (rc == FWGPU_ERR_RANGE && p->record.size() < (1u << 24)) {
            p->record.resize(p->record.size() * 4);
            continue;
        }
        break;
    }
    if (rc != FWGPU_OK) return kExceptionErrorCode;  // parse errors and commands alike ("Reading result ... returns error")
    if (n_words == 0) return kEofErrorCode;
    p->record[1] = 0;  // no label in a request (NO_LABEL = 0xff); the prediction does not depend on it
    SharedModel &m = *p->model;
    std::lock_guard<std::mutex> g(m.mu);
    if (prefix && p->cache) {  // predict_with_cache (lib.rs:88-108): translate, then only the uncached features are gathered
        float label, imp, out = 0.0f;
        if (translate_record(&m.tr, p->record.data(), n_words, p->lr, p->ffm, &label, &imp) != FWGPU_OK) return kExceptionErrorCode;
        if (fwgpu_predict_with_cache(m.re, p->cache, p->lr.data(), (uint32_t)p->lr.size(), p->ffm.data(), (uint32_t)p->ffm.size(), &out) != FWGPU_OK)
            return kExceptionErrorCode;
        return out;
    }
    // The request goes into host memory the device has mapped; the kernel reads it over PCIe in its stage phase and writes the
    // prediction back the same way.  Per call: one kernel launch and one synchronisation (before: two copies in, a memset, the
    // launch, a copy out and the synchronisation).
    if (!m.one || m.one->words_cap < n_words) {
        if (m.one) fwgpu_batch_free(m.one);
        m.one = nullptr;
        if (record_batch_alloc(m.re, &m.tr, 1, std::max<uint64_t>(2 * (uint64_t)n_words, 1 << 14), &m.one, /*host_mapped=*/true) != FWGPU_OK)
            return kExceptionErrorCode;
    }
    fwgpu_batch *b = m.one;
    uint32_t c_lr = 0, c_ffm = 0;
    if (count_record(&m.tr, p->record.data(), n_words, &c_lr, &c_ffm) != FWGPU_OK) return kExceptionErrorCode;  // validates the record
    std::memcpy(b->h_records, p->record.data(), (size_t)n_words * 4);
    b->h_rec_off[0] = 0;
    b->h_rec_off[1] = n_words;
    b->n = 1;
    b->n_lr = c_lr;
    b->n_ffm = c_ffm;
    b->n_words = n_words;
    b->max_lr = m.re->cfg.wiring == FWGPU_WIRING_FFM_ONLY ? 0 : c_lr;
    b->max_ffm = c_ffm;
    b->max_rec = n_words;
    b->aligned4 = true;
    b->h_pred[0] = kExceptionErrorCode;  // (a launch that processed nothing must not return the previous request's answer)
    if (fwgpu_learn_batch(m.re, b, FWGPU_MODE_SEQUENTIAL, /*update=*/0, nullptr) != FWGPU_OK) return kExceptionErrorCode;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return kExceptionErrorCode;
    return b->h_pred[0];
}

}  // namespace

extern "C" {

FfiPredictor *new_fw_predictor_prototype(const char *command) {  // lib.rs:150-185
    if (!command) {
        set_error("new_fw_predictor_prototype: NULL command");
        return nullptr;
    }
    const std::vector<std::string> words = shell_split(command);
    std::string weights;
    int device = 0;
    for (size_t i = 0; i + 1 < words.size(); i++) {
        if (words[i] == "-i" || words[i] == "--initial_regressor") weights = words[i + 1];
        if (words[i] == "--device") device = std::atoi(words[i + 1].c_str());  // ours: which GPU serves
    }
    for (const auto &w : words)
        if (w.rfind("--initial_regressor=", 0) == 0) weights = w.substr(std::strlen("--initial_regressor="));
    if (weights.empty()) {
        set_error("Cannot resolve input weights file name");  // the reference panics here (lib.rs:164-167)
        return nullptr;
    }
    auto model = std::make_shared<SharedModel>();
    if (fwgpu_model_load(weights.c_str(), device, /*immutable=*/1, &model->vw, &model->mi, &model->re) != FWGPU_OK) return nullptr;
    if (fwgpu_mi_configs(model->mi, device, nullptr, &model->tr, nullptr) != FWGPU_OK) return nullptr;
    auto p = std::make_unique<FfiPredictor>();
    p->model = model;
    if (fwgpu_parser_create(model->vw, &p->parser) != FWGPU_OK) return nullptr;
    return p.release();
}

FfiPredictor *clone_lite(FfiPredictor *prototype) {  // lib.rs:187-205: cheap copy, one per thread, shared weights
    if (!prototype) return nullptr;
    auto p = std::make_unique<FfiPredictor>();
    p->model = prototype->model;
    if (fwgpu_parser_create(p->model->vw, &p->parser) != FWGPU_OK) return nullptr;
    return p.release();
}

float fw_predict(FfiPredictor *ptr, const char *input_buffer) {  // lib.rs:207-212
    if (!ptr || !input_buffer) return kExceptionErrorCode;
    return predict_line(ptr, nullptr, 0, input_buffer, std::strlen(input_buffer));
}

float fw_setup_cache(FfiPredictor *ptr, const char *input_buffer) {  // lib.rs:224-232, 110-147
    if (!ptr || !input_buffer) return kExceptionErrorCode;
    size_t len = std::strlen(input_buffer);
    // the context line must itself parse (next_vowpal_with_size), and an empty one is EOF
    uint32_t n_words = 0;
    const int rc = fwgpu_parser_parse_line(ptr->parser, input_buffer, len, nullptr, 0, &n_words);
    if (rc != FWGPU_OK) return kExceptionErrorCode;
    if (n_words == 0) return kEofErrorCode;
    ptr->record.resize(std::max<size_t>(ptr->record.size(), (size_t)n_words));
    if (fwgpu_parser_parse_line(ptr->parser, input_buffer, len, ptr->record.data(), (uint32_t)ptr->record.size(), &n_words) != FWGPU_OK)
        return kExceptionErrorCode;
    if (len && input_buffer[len - 1] == '\n') len -= 1;  // "ignore last newline byte" (parser.rs:184-193)
    ptr->cached_text.assign(input_buffer, len);
    ptr->has_cache = true;
    ptr->delta_ok = false;
    if (ptr->prefix) fwgpu_parse_prefix_free(ptr->prefix);
    ptr->prefix = nullptr;
    if (fwgpu_parse_prefix_create(ptr->parser, ptr->cached_text.data(), ptr->cached_text.size(), &ptr->prefix) != FWGPU_OK) return kExceptionErrorCode;
    SharedModel &m = *ptr->model;
    if (m.re->nn.n_layers == 0 && m.re->cfg.ffm_k != 0) {
        // translate_and_filter(buffer, 0, Some(Primitive)) + Regressor::setup_cache (lib.rs:133-146); every namespace this
        // library accepts is primitive (transformed namespaces are refused when the model is loaded)
        float label, imp;
        std::lock_guard<std::mutex> g(m.mu);
        if (translate_record(&m.tr, ptr->record.data(), n_words, ptr->lr, ptr->ffm, &label, &imp) != FWGPU_OK) return kExceptionErrorCode;
        if (fwgpu_setup_cache(m.re, ptr->lr.data(), (uint32_t)ptr->lr.size(), ptr->ffm.data(), (uint32_t)ptr->ffm.size(), &ptr->cache) != FWGPU_OK)
            return kExceptionErrorCode;
        if (fwgpu_block_cache_cover_record(ptr->cache, &m.tr, ptr->record.data(), n_words) != FWGPU_OK) return kExceptionErrorCode;
        ptr->delta_ok = fwgpu_parse_prefix_is_record(ptr->prefix, ptr->record.data(), n_words) != 0;
    }
    return 0.0f;
}

float fw_predict_with_cache(FfiPredictor *ptr, const char *input_buffer) {  // lib.rs:214-222, 88-108
    if (!ptr || !input_buffer) return kExceptionErrorCode;
    return predict_line(ptr, ptr->cached_text.data(), ptr->cached_text.size(), input_buffer, std::strlen(input_buffer));
}

void free_predictor(FfiPredictor *ptr) { delete ptr; }  // lib.rs:234-236

// All candidates of a request in one launch.  inputs[i] is what fw_predict (with_cache == 0) or fw_predict_with_cache
// (with_cache != 0) would be given; out[i] what it would return (-1.0 for a line that does not parse).
int fwgpu_predictor_predict_batch(FfiPredictor *ptr, const char *const *inputs, uint32_t n, int with_cache, float *out) {
    if (!ptr || (!inputs && n) || (!out && n)) return fail(FWGPU_ERR_INVALID, "NULL argument");
    SharedModel &m = *ptr->model;
    // parse on a few host threads (each with its own parser: VowpalParser is not thread safe, clone_lite's reason)
    // (FWGPU_SERVING_THREADS overrides; by default one thread per 128 lines, up to the cores there are and at most 32)
    unsigned T = 1;
    if (n >= 256) {
        const char *env = std::getenv("FWGPU_SERVING_THREADS");
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        T = env && std::atoi(env) > 0 ? (unsigned)std::atoi(env) : std::min<unsigned>(std::min<unsigned>(32, hw), std::max<unsigned>(1, n / 128));
        T = std::min<unsigned>(T, 64);
    }
    const bool cached = with_cache && ptr->cache;  // candidates reduced to the entries the context cache does not cover
    // records + the kernel's own translation, unless a candidate turns out to need the entry route (FWGPU_SERVING_ENTRY_ROUTE=1 forces that one)
    bool by_record = cached && !std::getenv("FWGPU_SERVING_ENTRY_ROUTE");
    // ... and of those records only what the candidate adds to the context's record, which the device already holds
    // (FWGPU_SERVING_MERGED_RECORDS=1: whole context + candidate records instead)
    const bool delta_wanted = ptr->delta_ok && !std::getenv("FWGPU_SERVING_MERGED_RECORDS");
    bool delta = false;
    const bool timing = std::getenv("FWGPU_SERVING_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing)
            std::fprintf(stderr, "[predict_batch] %-10s %.3f ms since entry\n", what,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    while (ptr->parts.size() < T) ptr->parts.push_back(std::make_unique<Part>());
    std::vector<Part *> parts(T);
    for (unsigned k = 0; k < T; k++) parts[k] = ptr->parts[k].get();
    for (uint32_t i = 0; i < n; i++) out[i] = kExceptionErrorCode;  // a worker that cannot start leaves its slice marked
    auto work = [&](unsigned k) {
        Part &pt = *parts[k];
        pt.reset();
        if (!pt.parser && fwgpu_parser_create(m.vw, &pt.parser) != FWGPU_OK) return;
        fwgpu_parser *parser = pt.parser;
        std::vector<uint32_t> &rec = pt.rec;
        if (rec.size() < 4096) rec.resize(4096);
        std::vector<fwgpu_lr_entry> t_lr;
        std::vector<fwgpu_ffm_entry> t_ffm;
        const uint32_t a = (uint32_t)((uint64_t)n * k / T), e = (uint32_t)((uint64_t)n * (k + 1) / T);
        for (uint32_t i = a; i < e; i++) {
            out[i] = kExceptionErrorCode;
            if (!inputs[i]) continue;
            uint32_t nw = 0;
            int rc;
            int is_delta = 0;
            for (;;) {
                rc = delta ? fwgpu_parser_parse_candidate(parser, ptr->prefix, inputs[i], std::strlen(inputs[i]), rec.data(), (uint32_t)rec.size(), &nw, &is_delta)
                     : with_cache && ptr->prefix
                         ? fwgpu_parser_parse_after_prefix(parser, ptr->prefix, inputs[i], std::strlen(inputs[i]), rec.data(), (uint32_t)rec.size(), &nw)
                         : fwgpu_parser_parse_with_prefix(parser, with_cache ? ptr->cached_text.data() : nullptr,
                                                          with_cache ? ptr->cached_text.size() : 0, inputs[i], std::strlen(inputs[i]),
                                                          rec.data(), (uint32_t)rec.size(), &nw);
                if (rc == FWGPU_ERR_RANGE && rec.size() < (1u << 24)) {
                    rec.resize(rec.size() * 4);
                    continue;
                }
                break;
            }
            if (rc != FWGPU_OK || nw == 0) {
                if (rc == FWGPU_OK) out[i] = kEofErrorCode;
                continue;
            }
            rec[1] = 0;  // a request carries no label (NO_LABEL = 0xff); the prediction does not depend on it
            if (by_record && ((delta && !is_delta) || !block_cache_record_ok(ptr->cache, &m.tr, rec.data(), nw, delta))) {
                pt.entry_route = true;
                break;  // the whole request goes again, by entries
            }
            if (cached && !by_record) {
                float label, imp;
                if (translate_record(&m.tr, rec.data(), nw, t_lr, t_ffm, &label, &imp) != FWGPU_OK) continue;
                uint32_t kept = 0;
                if (fwgpu_block_cache_filter(ptr->cache, t_ffm.data(), (uint32_t)t_ffm.size(), t_ffm.data(), &kept) != FWGPU_OK) continue;
                if (append_example(m.re, pt.hb, t_lr.data(), (uint32_t)t_lr.size(), t_ffm.data(), kept, 0.0f, 1.0f) != FWGPU_OK) {
                    pt.hb_ok = false;
                    continue;
                }
                pt.slot.push_back(i);
                continue;
            }
            uint32_t c_lr = 0, c_ffm = 0;
            const uint32_t ctx_len = delta ? (uint32_t)ptr->cache->ctx_rec.size() : 0;
            if (count_record(&m.tr, rec.data(), nw, delta ? ptr->cache->ctx_rec.data() : nullptr, ctx_len, &c_lr, &c_ffm) != FWGPU_OK) {
                if (delta) {
                    pt.entry_route = true;
                    break;
                }
                pt.stats_ok = false;  // (the upload validates again and reports)
            } else {
                pt.stats.max_lr = std::max(pt.stats.max_lr, c_lr);
                pt.stats.max_ffm = std::max(pt.stats.max_ffm, c_ffm);
                pt.stats.max_rec = std::max(pt.stats.max_rec, nw + ctx_len);  // (LDS holds the context's record behind the candidate's)
                pt.stats.tot_lr += c_lr;
                pt.stats.tot_ffm += c_ffm;
            }
            pt.words.insert(pt.words.end(), rec.begin(), rec.begin() + nw);
            pt.len.push_back(nw);
            pt.slot.push_back(i);
        }
    };
    for (;;) {
        delta = by_record && delta_wanted;
        ptr->workers.run(T, work);
        bool again = false;
        for (const Part *pt : parts) again = again || pt->entry_route;
        if (!again) break;
        by_record = false;
    }
    lap("parsed");
    std::vector<uint64_t> off(1, 0);
    std::vector<uint32_t> slot;  // which input each record belongs to
    RecordStats stats;
    bool stats_ok = true;
    std::vector<size_t> wbase(parts.size() + 1, 0);
    for (size_t q = 0; q < parts.size(); q++) {
        const Part &pt = *parts[q];
        wbase[q + 1] = wbase[q] + pt.words.size();
        for (uint64_t l : pt.len) off.push_back(off.back() + l);
        slot.insert(slot.end(), pt.slot.begin(), pt.slot.end());
        stats.merge(pt.stats);
        stats_ok = stats_ok && pt.stats_ok;
    }
    const size_t total_words = wbase[parts.size()];
    lap("merged");
    if (slot.empty()) return FWGPU_OK;
    std::lock_guard<std::mutex> g(m.mu);
    const uint32_t nrec = (uint32_t)slot.size();
    if (cached && !by_record) {  // one entry batch, every example starting from the cached field sums
        HostBatch hb;
        hb.clear();
        {   // the parts' SoA arrays land in one host batch at their offsets (copied by one thread per part), example offsets rebased
            const size_t P = parts.size();
            std::vector<size_t> fb(P + 1, 0), lb(P + 1, 0), eb_(P + 1, 0);
            for (size_t q = 0; q < P; q++) {
                if (!parts[q]->hb_ok) return fail(FWGPU_ERR_RANGE, "predict_batch: a candidate's entries were refused (see append_example)");
                fb[q + 1] = fb[q] + parts[q]->hb.ffm_hash.size();
                lb[q + 1] = lb[q] + parts[q]->hb.lr_hash.size();
                eb_[q + 1] = eb_[q] + parts[q]->hb.label.size();
            }
            hb.ffm_hash.resize(fb[P]);
            hb.ffm_val.resize(fb[P]);
            hb.ffm_fld.resize(fb[P]);
            hb.lr_hash.resize(lb[P]);
            hb.lr_val.resize(lb[P]);
            hb.lr_combo.resize(lb[P]);
            hb.label.resize(eb_[P]);
            hb.importance.resize(eb_[P]);
            hb.ffm_off.resize(eb_[P] + 1);
            hb.lr_off.resize(eb_[P] + 1);
            hb.ffm_off[0] = hb.lr_off[0] = 0;
            auto put = [&](size_t q) {
                const HostBatch &src = parts[q]->hb;
                auto cp = [](auto &dst, size_t at, const auto &from) {
                    if (!from.empty()) std::memcpy(dst.data() + at, from.data(), from.size() * sizeof(from[0]));
                };
                cp(hb.ffm_hash, fb[q], src.ffm_hash);
                cp(hb.ffm_val, fb[q], src.ffm_val);
                cp(hb.ffm_fld, fb[q], src.ffm_fld);
                cp(hb.lr_hash, lb[q], src.lr_hash);
                cp(hb.lr_val, lb[q], src.lr_val);
                cp(hb.lr_combo, lb[q], src.lr_combo);
                cp(hb.label, eb_[q], src.label);
                cp(hb.importance, eb_[q], src.importance);
                for (size_t j = 1; j < src.ffm_off.size(); j++) hb.ffm_off[eb_[q] + j] = src.ffm_off[j] + (uint32_t)fb[q];
                for (size_t j = 1; j < src.lr_off.size(); j++) hb.lr_off[eb_[q] + j] = src.lr_off[j] + (uint32_t)lb[q];
            };
            ptr->workers.run((unsigned)P, [&](unsigned q) { put(q); });
            for (const Part *pt : parts) {
                hb.max_lr = std::max(hb.max_lr, pt->hb.max_lr);
                hb.max_ffm = std::max(hb.max_ffm, pt->hb.max_ffm);
                hb.aligned4 = hb.aligned4 && pt->hb.aligned4;
            }
        }
        lap("batched");
        fwgpu_batch *eb = nullptr;
        int rc0 = batch_alloc(m.re, nrec, hb.lr_hash.size(), hb.ffm_hash.size(), &eb);
        if (rc0 != FWGPU_OK) return rc0;
        rc0 = batch_upload(eb, hb, 0);
        if (rc0 == FWGPU_OK) rc0 = fwgpu_batch_set_cache(eb, ptr->cache);
        if (rc0 == FWGPU_OK) rc0 = fwgpu_learn_batch(m.re, eb, FWGPU_MODE_HOGWILD, /*update=*/0, nullptr);
        std::vector<float> preds(nrec);
        if (rc0 == FWGPU_OK) rc0 = fwgpu_batch_predictions(eb, preds.data(), nrec, nullptr);
        fwgpu_batch_free(eb);
        if (rc0 != FWGPU_OK) return rc0;
        for (size_t j = 0; j < slot.size(); j++) out[slot[j]] = preds[j];
        lap("predicted");
        return FWGPU_OK;
    }
    if (!m.batch || m.batch->n_cap < nrec || m.batch->words_cap < total_words) {
        if (m.batch) fwgpu_batch_free(m.batch);
        m.batch = nullptr;
        int rc0 = record_batch_alloc(m.re, &m.tr, std::max<uint32_t>(nrec * 2, 256), std::max<uint64_t>(total_words * 2, 1 << 16), &m.batch);
        if (rc0 != FWGPU_OK) return rc0;
    }
    fwgpu_batch *b = m.batch;
    int rc = FWGPU_OK;
    if (stats_ok) {
        // every thread's records go to the device from where the thread left them (no merged host copy); the threads
        // counted and validated their records as they parsed (what record_batch_upload would do again, serially)
        for (size_t q = 0; q < parts.size() && rc == FWGPU_OK; q++)
            if (!parts[q]->words.empty() &&
                hipMemcpyAsync(b->records + wbase[q], parts[q]->words.data(), parts[q]->words.size() * 4, hipMemcpyHostToDevice, nullptr) != hipSuccess)
                rc = fail(FWGPU_ERR_DEVICE, "predict_batch: copy to the device failed");
        if (rc == FWGPU_OK && hipMemcpyAsync(b->rec_off, off.data(), ((size_t)nrec + 1) * 8, hipMemcpyHostToDevice, nullptr) != hipSuccess)
            rc = fail(FWGPU_ERR_DEVICE, "predict_batch: copy to the device failed");
        b->n = nrec;
        b->n_lr = stats.tot_lr;
        b->n_ffm = stats.tot_ffm;
        b->n_words = total_words;
        b->max_lr = m.re->cfg.wiring == FWGPU_WIRING_FFM_ONLY ? 0 : stats.max_lr;
        b->max_ffm = stats.max_ffm;
        b->max_rec = stats.max_rec;
        b->aligned4 = true;
    } else {  // some record did not validate: one host copy, and record_batch_upload says which and why
        std::vector<uint32_t> words(total_words);
        for (size_t q = 0; q < parts.size(); q++)
            if (!parts[q]->words.empty()) std::memcpy(words.data() + wbase[q], parts[q]->words.data(), parts[q]->words.size() * 4);
        rc = record_batch_upload(b, &m.tr, words.data(), off.data(), nrec, 0, nullptr);
    }
    if (rc != FWGPU_OK) return rc;
    lap("uploaded");
    if (by_record) rc = fwgpu_batch_set_cache(b, ptr->cache);  // the kernel's translation leaves the covered namespaces out
    b->delta_records = delta;
    if (rc == FWGPU_OK) rc = fwgpu_learn_batch(m.re, b, FWGPU_MODE_HOGWILD, /*update=*/0, nullptr);
    std::vector<float> preds(slot.size());
    if (rc == FWGPU_OK) rc = fwgpu_batch_predictions(b, preds.data(), (uint32_t)preds.size(), nullptr);
    (void)fwgpu_batch_set_cache(b, nullptr);
    b->delta_records = false;
    if (rc != FWGPU_OK) return rc;
    for (size_t j = 0; j < slot.size(); j++) out[slot[j]] = preds[j];
    lap("predicted");
    return FWGPU_OK;
}

}  // extern "C"
