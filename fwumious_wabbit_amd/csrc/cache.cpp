// `.fwcache` input cache: cache.rs:12-26 layout ("FWCA", u32 version 11, u64-length-prefixed JSON of
// vw_source, then the parser's u32 records back to back), cache.rs:70-131 open/verify/create rules, cache.rs:133-232
// record I/O; LZ4 frame variant when the input file name ends in "gz" (cache.rs:73, 88-92, 112-121).
// The reader hands out records in bulk (words + offsets), the shape fwgpu_trainer_digest_records and
// fwgpu_record_batch_create take, so a cache file streams straight into HBM.
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "fwgpu_internal.h"
#include "lz4frame.hpp"

namespace fwgpu {
std::string vwmap_json(const fwgpu_vwmap *vw);                         // parser.cpp
bool vwmap_source_equal(const fwgpu_vwmap *a, const fwgpu_vwmap *b);   // parser.cpp
constexpr uint32_t kCacheVersion = 11;                                 // cache.rs:13
}  // namespace fwgpu

struct fwgpu_cache {
    std::string final_name, tmp_name;
    bool reading = false, writing = false, lz4 = false;
    FILE *f = nullptr;
    std::unique_ptr<fwlz4::FrameReader> zr;
    std::unique_ptr<fwlz4::FrameWriter> zw;
    std::vector<uint8_t> carry;  // bytes of a record split across reads
    uint64_t total_read = 0, records_read = 0, records_written = 0;

    size_t raw_read(uint8_t *dst, size_t n) {
        if (lz4) return zr->read(dst, n);
        return std::fread(dst, 1, n, f);
    }
    void raw_write(const uint8_t *p, size_t n) {
        if (lz4) {
            zw->write(p, n);
        } else if (std::fwrite(p, 1, n, f) != n) {
            throw std::runtime_error("cache: write failed");
        }
    }
    ~fwgpu_cache() {
        if (f) std::fclose(f);
    }
};

using namespace fwgpu;

namespace {

// cache.rs:163-185 verify_header
void verify_header(fwgpu_cache *c, const fwgpu_vwmap *vw) {
    uint8_t h[16];
    if (c->raw_read(h, 4) != 4 || std::memcmp(h, "FWCA", 4) != 0)
        throw std::runtime_error("Cache header does not begin with magic bytes FWFW");  // sic (cache.rs:167)
    if (c->raw_read(h, 4) != 4) throw std::runtime_error("cache: truncated header");
    const uint32_t version = fwlz4::rd32(h);
    if (version != kCacheVersion)
        throw std::runtime_error("Cache file version of this binary: " + std::to_string(kCacheVersion) +
                                 ", version of the cache file: " + std::to_string(version));
    if (c->raw_read(h, 8) != 8) throw std::runtime_error("cache: truncated header");
    uint64_t len = 0;
    for (int i = 7; i >= 0; i--) len = (len << 8) | h[i];
    if (len > (64u << 20)) throw std::runtime_error("cache: implausible vw_source length");
    std::string json(len, '\0');
    if (c->raw_read(reinterpret_cast<uint8_t *>(&json[0]), len) != len) throw std::runtime_error("cache: truncated vw_source");
    fwgpu_vwmap *from_cache = nullptr;
    if (fwgpu_vwmap_from_json(json.data(), json.size(), &from_cache) != FWGPU_OK)
        throw std::runtime_error(std::string("cache: ") + fwgpu_last_error());
    const bool same = vwmap_source_equal(from_cache, vw);
    fwgpu_vwmap_free(from_cache);
    if (!same) throw std::runtime_error("vw_namespace_map.csv and the one from cache file differ");
}

void write_header(fwgpu_cache *c, const fwgpu_vwmap *vw) {  // cache.rs:154-161
    uint8_t h[16];
    std::memcpy(h, "FWCA", 4);
    fwlz4::wr32(h + 4, kCacheVersion);
    const std::string json = vwmap_json(vw);
    uint64_t len = json.size();
    for (int i = 0; i < 8; i++) h[8 + i] = (uint8_t)(len >> (8 * i));
    c->raw_write(h, 16);
    c->raw_write(reinterpret_cast<const uint8_t *>(json.data()), json.size());
}

bool ends_with(const std::string &s, const char *suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

extern "C" {

int fwgpu_cache_open(const char *input_filename, const fwgpu_vwmap *vw, fwgpu_cache **out) {
    if (!input_filename || !vw || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    auto c = std::make_unique<fwgpu_cache>();
    c->final_name = std::string(input_filename) + ".fwcache";
    c->tmp_name = std::string(input_filename) + ".fwcache.writing";
    c->lz4 = ends_with(input_filename, "gz");
    struct stat st;
    if (::stat(c->final_name.c_str(), &st) == 0) {
        c->f = std::fopen(c->final_name.c_str(), "rb");
        if (!c->f) return fail(FWGPU_ERR_IO, "cannot open " + c->final_name);
        c->reading = true;
        try {
            if (c->lz4) c->zr = std::make_unique<fwlz4::FrameReader>(c->f);
            verify_header(c.get(), vw);
        } catch (const std::exception &e) {
            // "Couldn't use the existing cache file": fall back to (re)writing it (cache.rs:96-102)
            set_error(std::string("Couldn't use the existing cache file: ") + e.what());
            c->reading = false;
            c->zr.reset();
            std::fclose(c->f);
            c->f = nullptr;
        }
    }
    if (!c->reading) {
        c->f = std::fopen(c->tmp_name.c_str(), "wb");
        if (!c->f) return fail(FWGPU_ERR_IO, "cannot create " + c->tmp_name);
        c->writing = true;
        try {
            if (c->lz4) c->zw = std::make_unique<fwlz4::FrameWriter>(c->f);
            write_header(c.get(), vw);
        } catch (const std::exception &e) {
            return fail(FWGPU_ERR_IO, e.what());
        }
    }
    *out = c.release();
    return FWGPU_OK;
}

int fwgpu_cache_is_reading(const fwgpu_cache *c) { return c && c->reading; }
int fwgpu_cache_is_writing(const fwgpu_cache *c) { return c && c->writing; }

int fwgpu_cache_push_records(fwgpu_cache *c, const uint32_t *words, uint64_t n_words) {  // cache.rs:133-144
    if (!c || (!words && n_words)) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (!c->writing) return FWGPU_OK;  // like the reference: a no-op unless the cache is being written
    try {
        c->raw_write(reinterpret_cast<const uint8_t *>(words), n_words * 4);
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_IO, e.what());
    }
    return FWGPU_OK;
}

int fwgpu_cache_write_finish(fwgpu_cache *c) {  // cache.rs:146-152
    if (!c) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (!c->writing) return FWGPU_OK;
    try {
        if (c->zw) c->zw->finish();
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_IO, e.what());
    }
    if (std::fclose(c->f) != 0) {
        c->f = nullptr;
        return fail(FWGPU_ERR_IO, "cache: close failed");
    }
    c->f = nullptr;
    c->writing = false;
    if (std::rename(c->tmp_name.c_str(), c->final_name.c_str()) != 0)
        return fail(FWGPU_ERR_IO, "cannot rename " + c->tmp_name + " to " + c->final_name);
    return FWGPU_OK;
}

// Bulk get_next_record (cache.rs:187-232): fills `words` with whole records, rec_off[i] = word offset of record i
// (n_records + 1 entries).  *n_records == 0 means end of file.  A record longer than words_cap is an error.
int fwgpu_cache_next_records(fwgpu_cache *c, uint32_t *words, uint64_t words_cap, uint64_t *rec_off, uint64_t max_records,
                             uint64_t *n_records, uint64_t *n_words) {
    if (!c || !words || !rec_off || !n_records || !n_words) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (!c->reading)
        return fail(FWGPU_ERR_INVALID, "next_recrod() called on reading cache, when not opened in reading mode");  // sic
    uint8_t *dst = reinterpret_cast<uint8_t *>(words);
    const uint64_t cap = words_cap * 4;
    uint64_t have = c->carry.size();
    if (have > cap) return fail(FWGPU_ERR_RANGE, "record buffer too small");
    std::memcpy(dst, c->carry.data(), have);
    c->carry.clear();
    try {
        while (have < cap) {
            const size_t got = c->raw_read(dst + have, cap - have);
            if (got == 0) break;
            have += got;
            c->total_read += got;
        }
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_FORMAT, e.what());
    }
    uint64_t pos = 0, nr = 0;
    rec_off[0] = 0;
    while (nr < max_records && have - pos >= 4) {
        const uint64_t len = words[pos / 4];
        if (len < 3) return fail(FWGPU_ERR_FORMAT, "cache: record shorter than its header");
        if (len > words_cap) return fail(FWGPU_ERR_RANGE, "cache: a record is longer than the buffer");
        if (pos + len * 4 > have) break;
        pos += len * 4;
        rec_off[++nr] = pos / 4;
    }
    c->carry.assign(dst + pos, dst + have);
    if (nr == 0 && have - pos > 0 && have < cap)
        return fail(FWGPU_ERR_FORMAT, "cache: file ends inside a record");
    c->records_read += nr;
    *n_records = nr;
    *n_words = pos / 4;
    return FWGPU_OK;
}

void fwgpu_cache_free(fwgpu_cache *c) { delete c; }

}  // extern "C"
