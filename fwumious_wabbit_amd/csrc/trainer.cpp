// HogwildTrainer replacement (hogwild.rs:13-103).  Records are copied as they are into a host staging buffer,
// shipped to HBM in micro-batches and translated + learned on the device (the stage phase of the example kernel is
// FeatureBufferTranslator::translate).  Two staging/device buffers alternate so that filling and uploading
// micro-batch i+1 overlaps the kernel of micro-batch i.  The host only validates slot words (count_record).
#include <string.h>

#include <string>

#include <algorithm>
#include <memory>
#include <chrono>
#include <thread>

#include "fwgpu_internal.h"

using namespace fwgpu;

struct fwgpu_trainer {
    fwgpu_regressor *r = nullptr;
    uint32_t micro_batch = 0;
    // deep copy of the translator description (the caller's arrays are only borrowed for the create call)
    std::vector<uint32_t> combo_off, combo_ns, field_off, field_ns;
    std::vector<uint8_t> combo_f32, field_f32;
    std::vector<float> combo_weight;
    fwgpu_translator_config t{};
    // raw record words of the micro-batch being filled, in PINNED host memory (H2D at PCIe rate, truly asynchronous)
    uint32_t *rec[2] = {nullptr, nullptr};
    uint64_t rec_used[2] = {0, 0}, rec_cap[2] = {0, 0};
    std::vector<uint64_t> off[2];   // [n+1] word offsets
    RecordStats stats[2];           // accumulated while the records are copied in
    unsigned host_threads = 1;
    fwgpu_batch *dev[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipEvent_t uploaded[2] = {nullptr, nullptr};  // micro-batch c's records are in device buffer c (copy stream -> compute stream)
    hipStream_t copy_stream = nullptr;            // uploads run here, next to the previous micro-batch's kernel on `stream`
    bool in_flight[2] = {false, false};
    int cur = 0;
    hipStream_t stream = nullptr;
    uint64_t seen = 0;
    // hold-out / test-only protocol (main.rs:184-185, 238-241): examples numbered >= holdout_after (from 1) are predicted,
    // never learned; their predictions are kept in stream order
    uint64_t holdout_after = 0;
    bool testonly = false;
    bool slot_predict[2] = {false, false};
    uint32_t slot_n[2] = {0, 0};
    float *pred_host[2] = {nullptr, nullptr};  // pinned
    uint32_t pred_cap[2] = {0, 0};
    std::vector<float> preds;
    // chunk buffers of fwgpu_trainer_digest_cache
    std::unique_ptr<uint32_t[]> cache_words[2];
    std::unique_ptr<uint64_t[]> cache_off[2];
    uint64_t cache_off_cap = 0;
    // host threads kept between calls: staging copies of fwgpu_digest_records, parser threads of
    // fwgpu_trainer_digest_text.  Two pools: a chunk of text is parsed while the previous chunk's records are digested.
    Workers copy_pool, parse_pool;
};

static bool predict_mode(const fwgpu_trainer *tr) {  // for the NEXT example (number seen + 1)
    return tr->testonly || (tr->holdout_after && tr->seen + 1 >= tr->holdout_after);
}

// waits for slot c's launch and, if it was a predict-only one, appends its predictions (slots retire in launch order)
static int retire(fwgpu_trainer *tr, int c) {
    if (!tr->in_flight[c]) return FWGPU_OK;
    FWGPU_HIP(hipEventSynchronize(tr->done[c]));
    tr->in_flight[c] = false;
    if (tr->slot_predict[c]) tr->preds.insert(tr->preds.end(), tr->pred_host[c], tr->pred_host[c] + tr->slot_n[c]);
    tr->slot_predict[c] = false;
    return FWGPU_OK;
}

static int flush(fwgpu_trainer *tr, bool predict = false) {
    const int c = tr->cur;
    const uint32_t n = (uint32_t)(tr->off[c].size() - 1);
    if (n == 0) return FWGPU_OK;
    fwgpu_regressor *r = tr->r;
    FWGPU_HIP(hipSetDevice(r->device));
    // device buffer c may still be read by its previous kernel
    int rrc = retire(tr, c);
    if (rrc) return rrc;
    const uint64_t words = tr->rec_used[c];
    fwgpu_batch *b = tr->dev[c];
    if (!b || b->n_cap < n || b->words_cap < words) {
        if (b) fwgpu_batch_free(b);
        tr->dev[c] = nullptr;
        int rc = record_batch_alloc(r, &tr->t, std::max(n, tr->micro_batch), words * 3 / 2 + 4096, &tr->dev[c]);
        if (rc) return rc;
        b = tr->dev[c];
    }
    // The upload goes down its own stream: it overlaps the kernel of the previous micro-batch (other device buffer), and the
    // compute stream only waits for THIS buffer's copy (one stream for both cost the copy's 0.6 ms per 16 384 examples: 4.0 -> 4.6 M ex/s).
    int rc = record_batch_upload(b, &tr->t, tr->rec[c], tr->off[c].data(), n, tr->copy_stream, &tr->stats[c]);
    if (rc) return rc;
    // (a record that translates to more entries than a workgroup stages: the micro-batch is walked example by example, regressor.cpp learn_one_chunked)
    if ((rc = record_batch_host_copy_if_oversize(r, &tr->t, b, tr->rec[c], tr->off[c].data(), n))) return rc;
    FWGPU_HIP(hipEventRecord(tr->uploaded[c], tr->copy_stream));
    FWGPU_HIP(hipStreamWaitEvent(tr->stream, tr->uploaded[c], 0));
    rc = fwgpu_learn_batch(r, b, FWGPU_MODE_HOGWILD, predict ? 0 : 1, tr->stream);
    if (rc) return rc;
    if (predict) {
        if (tr->pred_cap[c] < n) {
            if (tr->pred_host[c]) (void)hipHostFree(tr->pred_host[c]);
            tr->pred_host[c] = nullptr;
            FWGPU_HIP(hipHostMalloc((void **)&tr->pred_host[c], (size_t)std::max(n, tr->micro_batch) * 4, hipHostMallocDefault));
            tr->pred_cap[c] = std::max(n, tr->micro_batch);
        }
        FWGPU_HIP(hipMemcpyAsync(tr->pred_host[c], b->pred, (size_t)n * 4, hipMemcpyDeviceToHost, tr->stream));
    }
    tr->slot_predict[c] = predict;
    tr->slot_n[c] = n;
    FWGPU_HIP(hipEventRecord(tr->done[c], tr->stream));
    tr->in_flight[c] = true;
    // Staging buffer c stays untouched until its copies and kernel are done; switch to the other buffer, which may be
    // refilled once ITS previous micro-batch has completed.
    tr->cur ^= 1;
    const int c2 = tr->cur;
    rrc = retire(tr, c2);
    if (rrc) return rrc;
    tr->rec_used[c2] = 0;
    tr->off[c2].assign(1, 0);
    tr->stats[c2] = RecordStats();
    return FWGPU_OK;
}

extern "C" {

int fwgpu_trainer_create(fwgpu_regressor *r, const fwgpu_translator_config *t, uint32_t micro_batch, fwgpu_trainer **out) {
    if (!r || !t || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (micro_batch == 0) return fail(FWGPU_ERR_INVALID, "micro_batch must be > 0");
    int crc = check_translator(r, t);
    if (crc) return crc;
    std::unique_ptr<fwgpu_trainer> tr(new fwgpu_trainer());
    tr->r = r;
    tr->micro_batch = micro_batch;
    tr->combo_off.assign(t->combo_off, t->combo_off + t->n_combos + 1);
    const uint32_t ncm = t->n_combos ? t->combo_off[t->n_combos] : 0;
    tr->combo_ns.assign(t->combo_ns, t->combo_ns + ncm);
    tr->combo_f32.assign(t->combo_ns_f32, t->combo_ns_f32 + ncm);
    tr->combo_weight.assign(t->combo_weight, t->combo_weight + t->n_combos);
    tr->field_off.assign(t->field_off, t->field_off + t->n_fields + 1);
    const uint32_t nfm = t->n_fields ? t->field_off[t->n_fields] : 0;
    tr->field_ns.assign(t->field_ns, t->field_ns + nfm);
    tr->field_f32.assign(t->field_ns_f32, t->field_ns_f32 + nfm);
    tr->t = *t;
    tr->t.combo_off = tr->combo_off.data();
    tr->t.combo_ns = tr->combo_ns.data();
    tr->t.combo_ns_f32 = tr->combo_f32.data();
    tr->t.combo_weight = tr->combo_weight.data();
    tr->t.field_off = tr->field_off.data();
    tr->t.field_ns = tr->field_ns.data();
    tr->t.field_ns_f32 = tr->field_f32.data();
    tr->off[0].assign(1, 0);
    tr->off[1].assign(1, 0);
    tr->host_threads = std::max(1u, std::thread::hardware_concurrency());
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipStreamCreateWithFlags(&tr->stream, hipStreamNonBlocking));
    FWGPU_HIP(hipStreamCreateWithFlags(&tr->copy_stream, hipStreamNonBlocking));
    FWGPU_HIP(hipEventCreateWithFlags(&tr->uploaded[0], hipEventDisableTiming));
    FWGPU_HIP(hipEventCreateWithFlags(&tr->uploaded[1], hipEventDisableTiming));
    FWGPU_HIP(hipEventCreateWithFlags(&tr->done[0], hipEventDisableTiming));
    FWGPU_HIP(hipEventCreateWithFlags(&tr->done[1], hipEventDisableTiming));
    *out = tr.release();
    return FWGPU_OK;
}

int fwgpu_digest_records(fwgpu_trainer *tr, const uint32_t *records, const uint64_t *rec_off, uint32_t n) {
    if (!tr || (n && (!records || !rec_off))) return fail(FWGPU_ERR_INVALID, "NULL argument");
    uint32_t i = 0;
    while (i < n) {
        const int c = tr->cur;
        const uint32_t have = (uint32_t)(tr->off[c].size() - 1);
        uint32_t take = std::min<uint32_t>(n - i, tr->micro_batch - have);
        const bool predicting = predict_mode(tr);
        if (!predicting && tr->holdout_after) {  // a micro-batch never straddles the hold-out boundary
            const uint64_t until = tr->holdout_after - 1 - tr->seen;  // examples that may still be learned (>= 1 here)
            if (take > until) take = (uint32_t)until;
        }
        // the records are copied (main.rs:243 `Vec::from(buffer)`)
        const uint64_t w0 = rec_off[i], w1 = rec_off[i + take];
        const uint64_t base = tr->rec_used[c];
        if (base + (w1 - w0) > tr->rec_cap[c]) {
            // (pinning memory is slow, 0.3-0.5 ms per MB: grow generously, a micro-batch a few percent larger than the last must not pin again)
            const uint64_t ncap = std::max<uint64_t>((base + (w1 - w0)) * 2, 1u << 20);
            uint32_t *nbuf = nullptr;
            FWGPU_HIP(hipSetDevice(tr->r->device));
            FWGPU_HIP(hipHostMalloc((void **)&nbuf, ncap * 4, hipHostMallocDefault));
            if (base) memcpy(nbuf, tr->rec[c], base * 4);
            if (tr->rec[c]) (void)hipHostFree(tr->rec[c]);
            tr->rec[c] = nbuf;
            tr->rec_cap[c] = ncap;
        }
        // copy into the pinned staging buffer and validate/count, on a few host threads for large slices
        {
            const unsigned T = take >= 4096 ? std::min<unsigned>(tr->host_threads, 8) : 1;
            std::vector<RecordStats> st(T);
            std::vector<int> rcs(T, FWGPU_OK);
            std::vector<std::string> msgs(T);
            auto work = [&](unsigned k) {
                const uint32_t a = i + (uint32_t)((uint64_t)take * k / T), e = i + (uint32_t)((uint64_t)take * (k + 1) / T);
                memcpy(tr->rec[c] + base + (rec_off[a] - w0), records + rec_off[a], (rec_off[e] - rec_off[a]) * 4);
                rcs[k] = count_records(&tr->t, records, rec_off + a, e - a, &st[k]);
                if (rcs[k]) msgs[k] = fwgpu_last_error();
            };
            tr->copy_pool.run(T, work);
            for (unsigned k = 0; k < T; k++) {
                if (rcs[k]) return fail(rcs[k], msgs[k]);
                tr->stats[c].merge(st[k]);
            }
        }
        tr->rec_used[c] = base + (w1 - w0);
        for (uint32_t j = 1; j <= take; j++) tr->off[c].push_back(base + (rec_off[i + j] - w0));
        tr->seen += take;
        i += take;
        if (tr->off[c].size() - 1 >= tr->micro_batch || (!predicting && predict_mode(tr))) {
            int rc = flush(tr, predicting);  // the last learned example closes its micro-batch
            if (rc) return rc;
        }
    }
    return FWGPU_OK;
}

// Cache file -> trainer without leaving native code (main.rs:213-270 with a cache: get_next_record -> digest_example).
// A reader thread fills one chunk from the file while the previous chunk is copied to pinned memory and shipped.
int fwgpu_trainer_digest_cache(fwgpu_trainer *tr, fwgpu_cache *cache, uint64_t max_records, uint64_t *n_digested) {
    if (!tr || !cache) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const uint64_t chunk_records = std::max<uint64_t>(4 * (uint64_t)tr->micro_batch, 65536);
    const uint64_t chunk_words = 8ull << 20;  // 32 MiB of records per chunk
    struct Chunk {
        uint32_t *words = nullptr;
        uint64_t *off = nullptr;
        uint64_t n = 0, nw = 0;
        int rc = FWGPU_OK;
        std::string msg;
    } chunks[2];
    for (int k = 0; k < 2; k++) {  // buffers live in the trainer: no per-call allocation or zero fill
        if (!tr->cache_words[k]) tr->cache_words[k].reset(new uint32_t[chunk_words]);
        if (!tr->cache_off[k] || tr->cache_off_cap < chunk_records + 1) tr->cache_off[k].reset(new uint64_t[chunk_records + 1]);
        chunks[k].words = tr->cache_words[k].get();
        chunks[k].off = tr->cache_off[k].get();
    }
    tr->cache_off_cap = chunk_records + 1;
    uint64_t left = max_records ? max_records : ~0ull, done = 0;
    auto read_chunk = [&](Chunk &c, uint64_t want) {
        c.rc = fwgpu_cache_next_records(cache, c.words, chunk_words, c.off, std::min(want, chunk_records), &c.n, &c.nw);
        if (c.rc) c.msg = fwgpu_last_error();
    };
    int cur = 0;
    read_chunk(chunks[cur], left);
    int rc = FWGPU_OK;
    while (true) {
        Chunk &c = chunks[cur];
        if (c.rc) {
            rc = fail(c.rc, c.msg);
            break;
        }
        if (c.n == 0) break;  // end of file
        left -= c.n;
        std::thread reader;
        Chunk &next = chunks[cur ^ 1];
        const bool more = left > 0;
        if (more) reader = std::thread(read_chunk, std::ref(next), left);
        uint64_t i = 0;
        while (i < c.n && rc == FWGPU_OK) {  // fwgpu_digest_records takes a 32-bit count
            const uint32_t take = (uint32_t)std::min<uint64_t>(c.n - i, 1u << 30);
            rc = fwgpu_digest_records(tr, c.words, c.off + i, take);
            i += take;
        }
        if (reader.joinable()) reader.join();
        done += c.n;
        if (rc || !more) break;
        cur ^= 1;
    }
    if (n_digested) *n_digested = done;
    return rc;
}

// Text -> trainer in native code (main.rs:213-270 without a cache, or its first pass with `-c`): the buffer is cut at line
// boundaries into one slice per host thread, every slice is parsed by its own parser clone (VowpalParser is not thread
// safe), the records are digested in the original order and, when `cache` is open for writing, appended to it.
int fwgpu_trainer_digest_text(fwgpu_trainer *tr, fwgpu_parser *parser, fwgpu_cache *cache, const char *text, uint64_t len,
                              uint32_t threads, uint64_t *n_examples, uint64_t *consumed) {
    if (!tr || !parser || (!text && len)) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const unsigned T = std::max(1u, std::min<unsigned>(threads ? threads : 8, 64));
    struct Slice {
        uint64_t begin = 0, end = 0, used = 0, nr = 0, nw = 0;
        std::unique_ptr<uint32_t[]> words;  // uninitialised on purpose: zero-filling hundreds of MB costs more than parsing them
        uint64_t words_cap = 0;
        std::vector<uint64_t> off;
        int rc = FWGPU_OK;
        std::string msg;
        void grow(uint64_t cap) {
            std::unique_ptr<uint32_t[]> nb(new uint32_t[cap]);
            if (nw) memcpy(nb.get(), words.get(), nw * 4);
            words = std::move(nb);
            words_cap = cap;
        }
    };
    struct Clones {  // one parser clone per extra thread (VowpalParser is not thread safe), for the length of this call
        std::vector<fwgpu_parser *> p;
        ~Clones() {
            for (fwgpu_parser *q : p) fwgpu_parser_free(q);
        }
    } clones;
    // Two stages, pipelined over chunks of the text: chunk i+1 is parsed (T threads) while chunk i's records are digested (staging
    // copy, upload, learn launches) on a helper thread, in order.  One pass over the whole buffer (parse everything, then digest
    // everything) left the GPU idle while parsing and the parsers idle while learning: 1.3 M lines/s on 16 threads.
    auto parse_range = [&](uint64_t r_begin, uint64_t r_end, std::vector<Slice> &sl) {
    if (sl.size() != T) sl.resize(T);
    for (Slice &q : sl) {  // (the record buffers of the previous chunk are kept: first-touch page faults of 16 threads at once do not scale)
        q.begin = q.end = q.used = q.nr = q.nw = 0;
        q.rc = FWGPU_OK;
        q.msg.clear();
    }
    uint64_t pos = r_begin;
    const uint64_t rlen = r_end - r_begin;
    for (unsigned k = 0; k < T; k++) {  // slice k ends at the first line break at or after its proportional share
        sl[k].begin = pos;
        uint64_t e = k + 1 == T ? r_end : std::max<uint64_t>(pos, r_begin + rlen * (k + 1) / T);
        if (e < r_end) {
            const void *nl = memchr(text + e, '\n', r_end - e);
            e = nl ? (uint64_t)(static_cast<const char *>(nl) - text) + 1 : r_end;
        }
        sl[k].end = pos = e;
    }
    auto work = [&](unsigned k) {
        Slice &s = sl[k];
        const uint64_t n = s.end - s.begin;
        if (!n) return;
        fwgpu_parser *p = k > 0 ? clones.p[k - 1] : parser;
        uint64_t lines = 0;
        for (const char *q = text + s.begin, *e = text + s.end; q < e; lines++) {
            const void *nl = memchr(q, '\n', (size_t)(e - q));
            q = nl ? static_cast<const char *>(nl) + 1 : e;
        }
        // a feature token takes >= 2 bytes of text and <= 2 words of record; the per-line header depends on the namespace
        // map, so the buffer grows whenever a pass stops short without an error
        if (s.words_cap < n + lines * 64 + 4096) s.grow(n + lines * 64 + 4096);
        s.off.assign(lines + 1, 0);
        std::vector<uint64_t> tmp(lines + 1);
        while (s.used < n) {
            uint64_t nr = 0, nw = 0, used = 0;
            s.rc = fwgpu_parser_parse_buffer(p, text + s.begin + s.used, n - s.used, s.words.get() + s.nw, s.words_cap - s.nw,
                                             tmp.data(), lines - s.nr, &nr, &nw, &used);
            for (uint64_t j = 1; j <= nr; j++) s.off[s.nr + j] = s.nw + tmp[j];
            s.nr += nr;
            s.nw += nw;
            s.used += used;
            if (s.rc != FWGPU_OK) {
                s.msg = fwgpu_last_error();
                break;
            }
            if (s.used < n) {  // out of room (not a command, not an error): grow and go on
                if (s.words_cap > (1ull << 33)) {
                    s.rc = FWGPU_ERR_RANGE;
                    s.msg = "digest_text: a slice needs more than 32 GiB of records";
                    break;
                }
                s.grow(s.words_cap * 2);
            }
        }
    };
    tr->parse_pool.run(T, work);
    };
    for (unsigned k = 1; k < T; k++) {
        fwgpu_parser *c = nullptr;
        const int crc = fwgpu_parser_clone(parser, &c);
        if (crc) return crc;
        clones.p.push_back(c);
    }
    uint64_t done = 0, used = 0;
    // digests the slices of one chunk in order; false: stop (rc says why: an error, or a command line reached)
    auto digest_slices = [&](std::vector<Slice> &sl, int &rc, std::string &msg) -> bool {
        for (unsigned k = 0; k < T && rc == FWGPU_OK; k++) {
            Slice &s = sl[k];
            uint64_t i = 0;
            while (i < s.nr && rc == FWGPU_OK) {
                const uint32_t take = (uint32_t)std::min<uint64_t>(s.nr - i, 1u << 30);
                // offsets are relative to the slice's first word
                rc = fwgpu_digest_records(tr, s.words.get(), s.off.data() + i, take);
                i += take;
            }
            if (rc == FWGPU_OK && cache && s.nw) rc = fwgpu_cache_push_records(cache, s.words.get(), s.nw);
            if (rc != FWGPU_OK) {
                msg = fwgpu_last_error();  // (thread-local: carried back to the caller's thread)
                return false;
            }
            done += s.nr;
            used = s.begin + s.used;
            if (s.rc != FWGPU_OK) {  // a command or a bad line stopped this slice: stop here, in order
                rc = s.rc;
                msg = s.msg;
                return false;
            }
        }
        return rc == FWGPU_OK;
    };
    // chunks of about 1 MB of text per parser thread (the threads are a pool: a chunk costs no thread start), 8 to 64 MB in all
    // (FWGPU_TEXT_CHUNK_MB overrides), cut at line breaks
    uint64_t chunk_target = std::min<uint64_t>(std::max<uint64_t>((uint64_t)T * (1u << 20), 8u << 20), 64u << 20);
    if (const char *env = getenv("FWGPU_TEXT_CHUNK_MB"))
        if (atoi(env) > 0) chunk_target = (uint64_t)atoi(env) << 20;
    std::vector<Slice> cur, next;
    int rc = FWGPU_OK;
    std::string msg;
    uint64_t pos = 0;
    auto chunk_end = [&](uint64_t from) {
        uint64_t e = std::min<uint64_t>(len, from + chunk_target);
        if (e < len) {
            const void *nl = memchr(text + e, '\n', len - e);
            e = nl ? (uint64_t)(static_cast<const char *>(nl) - text) + 1 : len;
        }
        return e;
    };
    bool have = false;
    if (len) {
        const uint64_t e = chunk_end(0);
        parse_range(0, e, cur);
        pos = e;
        have = true;
    }
    const bool timing = getenv("FWGPU_TRAINER_TIMING") != nullptr;  // per chunk: parse of the next chunk / digest of this one, ms
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    while (have) {
        bool go_on = true;
        double t_digest = 0.0;
        const auto t0 = now();
        std::thread dg([&] {
            go_on = digest_slices(cur, rc, msg);
            t_digest = ms(t0, now());
        });
        bool parsed_next = false;
        if (pos < len) {
            const uint64_t e = chunk_end(pos);
            parse_range(pos, e, next);
            pos = e;
            parsed_next = true;
        }
        const double t_parse = ms(t0, now());
        dg.join();
        if (timing) fprintf(stderr, "[digest_text] chunk: parse of the next %.2f ms, digest of this one %.2f ms, both %.2f ms\n", t_parse, t_digest, ms(t0, now()));
        if (!go_on || !parsed_next) break;  // (a chunk parsed past a command line or an error is dropped: `consumed` says where to resume)
        cur.swap(next);
    }
    if (rc != FWGPU_OK && rc != FWGPU_PARSE_FLUSH && rc != FWGPU_PARSE_HOGWILD_LOAD) rc = fail(rc, msg);
    if (n_examples) *n_examples = done;
    if (consumed) *consumed = used;
    return rc;
}

int fwgpu_trainer_set_holdout(fwgpu_trainer *tr, uint64_t holdout_after, int testonly) {
    if (!tr) return fail(FWGPU_ERR_INVALID, "NULL trainer");
    if (tr->off[tr->cur].size() > 1) return fail(FWGPU_ERR_INVALID, "set the hold-out before digesting (or right after fwgpu_finish)");
    tr->holdout_after = holdout_after;
    tr->testonly = testonly != 0;
    return FWGPU_OK;
}

int fwgpu_trainer_predictions(fwgpu_trainer *tr, float *out, uint64_t cap, uint64_t *n) {
    if (!tr || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *n = tr->preds.size();
    if (!out) return FWGPU_OK;
    if (cap < tr->preds.size()) return fail(FWGPU_ERR_RANGE, "buffer too small for the predictions");
    if (!tr->preds.empty()) memcpy(out, tr->preds.data(), tr->preds.size() * 4);
    return FWGPU_OK;
}

int fwgpu_finish(fwgpu_trainer *tr) {
    if (!tr) return fail(FWGPU_ERR_INVALID, "NULL trainer");
    // the open micro-batch holds examples of ONE kind; they were predicted iff the example before `seen + 1` was
    const bool last_predict = tr->testonly || (tr->holdout_after && tr->seen >= tr->holdout_after);
    int rc = flush(tr, last_predict);
    if (rc) return rc;
    FWGPU_HIP(hipSetDevice(tr->r->device));
    FWGPU_HIP(hipStreamSynchronize(tr->stream));
    // slots retire in launch order: the one launched before the current filling slot's predecessor first
    rc = retire(tr, tr->cur);
    if (rc) return rc;
    rc = retire(tr, tr->cur ^ 1);
    if (rc) return rc;
    return FWGPU_OK;
}

int fwgpu_trainer_free(fwgpu_trainer *tr) {
    if (!tr) return FWGPU_OK;
    (void)hipSetDevice(tr->r->device);
    if (tr->stream) (void)hipStreamSynchronize(tr->stream);
    for (int i = 0; i < 2; i++) {
        if (tr->dev[i]) fwgpu_batch_free(tr->dev[i]);
        if (tr->rec[i]) (void)hipHostFree(tr->rec[i]);
        if (tr->pred_host[i]) (void)hipHostFree(tr->pred_host[i]);
        if (tr->done[i]) (void)hipEventDestroy(tr->done[i]);
    }
    if (tr->stream) (void)hipStreamDestroy(tr->stream);
    if (tr->copy_stream) (void)hipStreamDestroy(tr->copy_stream);
    for (int i = 0; i < 2; i++)
        if (tr->uploaded[i]) (void)hipEventDestroy(tr->uploaded[i]);
    delete tr;
    return FWGPU_OK;
}

int fwgpu_trainer_examples_seen(const fwgpu_trainer *tr, uint64_t *n) {
    if (!tr || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *n = tr->seen;
    return FWGPU_OK;
}

}  // extern "C"
