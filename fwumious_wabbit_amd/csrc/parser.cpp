// Input side of the feed path: vw_namespace_map.csv (vwmap.rs:106-151), its JSON form
// (persistence.rs:36-53) and the VW text parser (parser.rs:214-461) producing the u32 records (parser.rs:57-74)
// that fwgpu_record_batch_create / fwgpu_trainer_digest_records consume.  Host code, no device work.
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "fwgpu_internal.h"
#include "json.hpp"

namespace fwgpu {

struct VwEntry {
    std::string vwname, verbose;
    uint32_t index;
    bool f32;
};

}  // namespace fwgpu

struct fwgpu_vwmap {
    std::vector<fwgpu::VwEntry> entries;  // vw_source.entries
    uint32_t skip_prefix = 0;             // vw_source.namespace_skip_prefix
    uint32_t num_namespaces = 0;          // max index + 1 (vwmap.rs:83-88)
    std::unordered_map<std::string, uint32_t> by_vwname, by_verbose;  // -> entries index

    void finish() {  // vwmap.rs:54-89 new_from_source
        num_namespaces = 0;
        by_vwname.clear();
        by_verbose.clear();
        for (uint32_t i = 0; i < entries.size(); i++) {
            by_vwname[entries[i].vwname] = i;
            by_verbose[entries[i].verbose] = i;
            num_namespaces = std::max(num_namespaces, entries[i].index);
        }
        num_namespaces += 1;
    }
};

struct fwgpu_parser {
    const fwgpu_vwmap *vw;  // copied
    fwgpu_vwmap vw_copy;
    std::vector<uint32_t> seed;  // murmur3::hash32(vwname) per entry (parser.rs:83)
    std::vector<uint32_t> out;   // output_buffer
    std::string scratch;         // padded copy of a line that has no readable byte after it
    std::vector<uint32_t> delta; // candidate-only form of the last record (fwgpu_parser_parse_candidate)
    std::string cmd_arg;         // filename of the last hogwild_load command
    // vwname -> entry: the reference walks a 256-ary radix tree (radix_tree.rs); a small open-addressing table on the
    // name's bytes does the same exact-match lookup without allocating
    std::vector<int32_t> ns_table;
    uint32_t ns_mask = 0;
    static uint32_t name_hash(const unsigned char *s, size_t n) {
        uint32_t h = 2166136261u;
        for (size_t i = 0; i < n; i++) h = (h ^ s[i]) * 16777619u;
        return h;
    }
    void build_ns_table() {
        uint32_t cap = 16;
        while (cap < 4 * vw_copy.entries.size()) cap <<= 1;
        ns_table.assign(cap, -1);
        ns_mask = cap - 1;
        for (size_t i = 0; i < vw_copy.entries.size(); i++) {
            const std::string &nm = vw_copy.entries[i].vwname;
            // later entries with the same vwname replace earlier ones, like HashMap::insert (vwmap.rs:75-78)
            uint32_t slot = name_hash(reinterpret_cast<const unsigned char *>(nm.data()), nm.size()) & ns_mask;
            while (ns_table[slot] >= 0 && vw_copy.entries[ns_table[slot]].vwname != nm) slot = (slot + 1) & ns_mask;
            ns_table[slot] = (int32_t)i;
        }
    }
    int find_ns(const unsigned char *s, size_t n) const {
        uint32_t slot = name_hash(s, n) & ns_mask;
        while (ns_table[slot] >= 0) {
            const std::string &nm = vw_copy.entries[ns_table[slot]].vwname;
            if (nm.size() == n && std::memcmp(nm.data(), s, n) == 0) return ns_table[slot];
            slot = (slot + 1) & ns_mask;
        }
        return -1;
    }
};

namespace fwgpu {

static inline uint32_t murmur3_32(const uint8_t *data, size_t len, uint32_t seed) { return fwgpu_murmur3_32(data, len, seed); }

// ---- the csv crate's reader as the reference configures it: no headers, flexible, "..." quoting, blank lines skipped
static std::vector<std::vector<std::string>> csv_records(const char *p, size_t n) {
    std::vector<std::vector<std::string>> recs;
    std::vector<std::string> cur;
    std::string field;
    bool in_quotes = false, any = false, field_started = false;
    auto end_field = [&]() {
        cur.push_back(field);
        field.clear();
        field_started = false;
    };
    auto end_record = [&]() {
        if (any) {
            end_field();
            recs.push_back(cur);
        }
        cur.clear();
        field.clear();
        any = false;
        field_started = false;
    };
    for (size_t i = 0; i < n; i++) {
        const char c = p[i];
        if (in_quotes) {
            if (c == '"') {
                if (i + 1 < n && p[i + 1] == '"') {
                    field += '"';
                    i++;
                } else {
                    in_quotes = false;
                }
            } else {
                field += c;
            }
            continue;
        }
        if (c == '"' && !field_started) {
            in_quotes = true;
            any = true;
            field_started = true;
        } else if (c == ',') {
            any = true;
            end_field();
        } else if (c == '\n' || c == '\r') {
            end_record();
        } else {
            field += c;
            any = true;
            field_started = true;
        }
    }
    end_record();
    return recs;
}

// Rust's f32::from_str grammar: [+-]? ( "inf" | "infinity" | "nan" (any case) | digits [ "." digits? ] | "." digits )
// [ (e|E) [+-]? digits ]; nothing else (no whitespace, no hex, no "nan(...)").  Correctly rounded like strtof.
static bool parse_f32_rust(const char *s, size_t n, float *out) {
    if (n == 0 || n > 4096) return false;
    size_t i = 0;
    if (s[i] == '+' || s[i] == '-') i++;
    if (i == n) return false;
    auto ieq = [&](const char *w) {
        const size_t l = std::strlen(w);
        if (n - i != l) return false;
        for (size_t j = 0; j < l; j++) {
            char c = s[i + j];
            if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
            if (c != w[j]) return false;
        }
        return true;
    };
    bool special = ieq("inf") || ieq("infinity") || ieq("nan");
    if (!special) {
        size_t j = i, nd = 0;
        while (j < n && s[j] >= '0' && s[j] <= '9') j++, nd++;
        if (j < n && s[j] == '.') {
            j++;
            while (j < n && s[j] >= '0' && s[j] <= '9') j++, nd++;
        }
        if (nd == 0) return false;
        if (j < n && (s[j] == 'e' || s[j] == 'E')) {
            j++;
            if (j < n && (s[j] == '+' || s[j] == '-')) j++;
            size_t ne = 0;
            while (j < n && s[j] >= '0' && s[j] <= '9') j++, ne++;
            if (ne == 0) return false;
        }
        if (j != n) return false;
    }
    std::string tmp(s, n);
    *out = std::strtof(tmp.c_str(), nullptr);
    return true;
}

enum { kHeaderLen = 3, kLabelOffset = 1, kImportanceOffset = 2 };
constexpr uint32_t kNotSingle = 1u << 31, kMask31 = ~kNotSingle, kNoFeatures = kNotSingle, kNoLabel = 0xff,
                   kFloatOne = 1065353216u;

static inline uint32_t f32_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// The scanner's namespace state between two tokens of the feature loop (parser.rs:318-326)
struct NsState {
    uint32_t ns_seed = 0;
    size_t ns_slot = kHeaderLen;
    bool ns_f32 = false;
    size_t ns_start = 0;
    float ns_weight = 1.0f;
    uint32_t ns_count = 0;
};

static int parse_float_or_error(const unsigned char *p, size_t a, size_t b, const char *what, float *out) {
    if (b - a == 4 && std::memcmp(p + a, "NONE", 4) == 0) {  // parser.rs:123-130
        *out = std::nanf("");
        return FWGPU_OK;
    }
    if (!parse_f32_rust(reinterpret_cast<const char *>(p + a), b - a, out))
        return fail(FWGPU_ERR_PARSE, std::string(what) + ": " + std::string(reinterpret_cast<const char *>(p + a), b - a));
    return FWGPU_OK;
}

// Label, importance and the scan to the first '|' (parser.rs:226-316).  `rowlen` bounds every scan; on FWGPU_OK *pos is
// where the feature loop starts.
static int parse_head(fwgpu_parser *ps, const unsigned char *p, size_t size, size_t rowlen, size_t *pos) {
    const fwgpu_vwmap &vw = ps->vw_copy;
    std::vector<uint32_t> &ob = ps->out;
    const size_t bufpos = vw.num_namespaces + kHeaderLen;
    ob.assign(bufpos, kNoFeatures);
    size_t i_start, i_end = 0;

    switch (p[0]) {
    case 0x31: ob[kLabelOffset] = 1; break;
    case 0x2d: ob[kLabelOffset] = 0; break;
    case 0x7c: ob[kLabelOffset] = kNoLabel; break;
    default:
        if (size >= 5 && std::memcmp(p, "flush", 5) == 0) return FWGPU_PARSE_FLUSH;
        if (size >= std::strlen("hogwild_load ")) {
            // parse_cmd (parser.rs:149-164): space separated tokens
            std::vector<std::string> toks;
            size_t e = 0;
            while (e < size) {
                std::string t;
                while (e < size && p[e] != 0x20) t += (char)p[e++];
                toks.push_back(t);
                while (e < size && p[e] == 0x20) e++;
            }
            if (toks.size() == 2) {
                if (toks[0] == "hogwild_load") {
                    ps->cmd_arg = toks[1];
                    return FWGPU_PARSE_HOGWILD_LOAD;
                }
                // any other two-token line falls through to the scanning below with the label word left at
                // NO_FEATURES, exactly as the reference does (parser.rs:243-252)
                break;
            }
            return fail(FWGPU_ERR_PARSE, "Cannot parse an example");
        }
        return fail(FWGPU_ERR_PARSE, "Cannot parse an example");
    }

    if (ob[kLabelOffset] == kNoLabel) {
        ob[kImportanceOffset] = kFloatOne;
    } else {
        while (p[i_end] != 0x20 && i_end < rowlen) i_end++;
        while (p[i_end] == 0x20 && i_end < rowlen) i_end++;
        if (p[i_end] == 0x7c) {
            ob[kImportanceOffset] = kFloatOne;
        } else {
            i_start = i_end;
            while (p[i_end] != 0x20 && i_end < rowlen) i_end++;
            float imp;
            int rc = parse_float_or_error(p, i_start, i_end, "Failed parsing example importance", &imp);
            if (rc) return rc;
            if (imp < 0.0f) {
                // Rust prints the f32 with {:?}: shortest round-trip digits
                return fail(FWGPU_ERR_PARSE, "Example importance cannot be negative: " + fwjson::format_f32(imp) + "! ");
            }
            ob[kImportanceOffset] = f32_bits(imp);
        }
    }
    while (p[i_end] != 0x7c && i_end < rowlen) i_end++;
    *pos = i_end;
    return FWGPU_OK;
}

// The feature loop (parser.rs:318-457) from token boundary `i_end` while i_end < stop; scans are bounded by `rowlen`.
// Positions are only ever used relative to `p`, so the loop can go on in another buffer that holds the rest of the line.
static int parse_body(fwgpu_parser *ps, const unsigned char *p, size_t rowlen, size_t i_end, size_t stop, NsState &st) {
    const fwgpu_vwmap &vw = ps->vw_copy;
    std::vector<uint32_t> &ob = ps->out;
    uint32_t &ns_seed = st.ns_seed;
    size_t &ns_slot = st.ns_slot;
    bool &ns_f32 = st.ns_f32;
    size_t &ns_start = st.ns_start;
    float &ns_weight = st.ns_weight;
    uint32_t &ns_count = st.ns_count;
    size_t i_start;
    while (i_end < stop) {
        while (p[i_end] == 0x20 && i_end < rowlen) i_end++;
        i_start = i_end;
        while (p[i_end] != 0x20 && p[i_end] != 0x3a && i_end < rowlen) i_end++;
        const size_t i_end_first = i_end;
        while (p[i_end] != 0x20 && i_end < rowlen) i_end++;

        if (p[i_start] == 0x7c) {
            i_start += 1;
            if (i_end_first != i_end) {
                int rc = parse_float_or_error(p, i_end_first + 1, i_end, "Failed parsing namespace weight", &ns_weight);
                if (rc) return rc;
            } else {
                ns_weight = 1.0f;
            }
            const int ei = ps->find_ns(p + i_start, i_end_first - i_start);
            if (ei < 0)
                return fail(FWGPU_ERR_PARSE, "Feature name was not predeclared in vw_namespace_map.csv: " +
                                                 std::string(reinterpret_cast<const char *>(p + i_start), i_end_first - i_start));
            const VwEntry &e = vw.entries[ei];
            ns_seed = ps->seed[ei];
            ns_slot = e.index + kHeaderLen;
            ns_f32 = e.f32;
            ns_count = 0;
            ns_start = ob.size();
        } else {
            const uint32_t h = murmur3_32(p + i_start, i_end_first - i_start, ns_seed) & kMask31;
            float fw = 1.0f;
            if (i_end_first != i_end) {
                int rc = parse_float_or_error(p, i_end_first + 1, i_end, "Failed parsing feature weight", &fw);
                if (rc) return rc;
            }
            if (ns_count == 0 && !ns_f32 && ns_weight == 1.0f && fw == 1.0f) {
                ob[ns_slot] = h;
            } else {
                const uint32_t cur = ob[ns_slot];
                if (ns_count == 1 && (cur & kNotSingle) == 0) {  // promote the in-place feature
                    ob.push_back(cur);
                    ob.push_back(kFloatOne);
                }
                ob.push_back(h);
                if (ns_f32) {
                    const size_t fs = i_start + vw.skip_prefix;
                    float v;
                    if (fs > i_end_first)  // name shorter than _namespace_skip_prefix: the reference slices start > end here (UB)
                        return fail(FWGPU_ERR_PARSE, "Failed parsing feature value to float (for float namespace): feature name "
                                                     "shorter than _namespace_skip_prefix");
                    if (i_end_first != fs) {
                        int rc = parse_float_or_error(p, fs, i_end_first, "Failed parsing feature value to float (for float namespace)", &v);
                        if (rc) return rc;
                    } else {
                        v = std::nanf("");
                    }
                    ob.push_back(f32_bits(v));
                    if (ns_weight * fw != 1.0f)
                        return fail(FWGPU_ERR_PARSE,
                                    "Namespaces that are f32 can not have weight attached neither to namespace nor to a single "
                                    "feature (basically they can' use :weight syntax");
                } else {
                    ob.push_back(f32_bits(ns_weight * fw));
                }
                ob[ns_slot] = kNotSingle | (uint32_t)((ns_start << 16) + ob.size());
            }
            ns_count += 1;
        }
        i_end += 1;
    }
    return FWGPU_OK;
}

// parser.rs:214-461 next_vowpal_to_size.  `p[0..size)` is the line as read by read_until('\n') (newline included when
// present).  Mirrors the reference's scanning literally, including `rowlen = size - 1` whether or not the last byte is
// a newline.  The buffer is padded so the reference's one-past reads (`*p.add(i_end)` with i_end == rowlen) are defined.
static int parse_line(fwgpu_parser *ps, const char *line, size_t size) {
    // `line` has at least one readable byte after `size` (callers copy the rare line that does not)
    const unsigned char *p = reinterpret_cast<const unsigned char *>(line);
    const size_t rowlen = size - 1;  // "ignore last newline byte"
    size_t pos = 0;
    int rc = parse_head(ps, p, size, rowlen, &pos);
    if (rc != FWGPU_OK) return rc;
    NsState st;
    rc = parse_body(ps, p, rowlen, pos, rowlen, st);
    if (rc != FWGPU_OK) return rc;
    ps->out[0] = (uint32_t)ps->out.size();
    return FWGPU_OK;
}

}  // namespace fwgpu

// A context line scanned once (fw_setup_cache), up to the last token boundary before its end: the record built so far and the
// scanner's state there.  Requests then scan only `tail` + their own bytes.
struct fwgpu_parse_prefix {
    std::string text;           // the whole prefix (the fallback concatenates, like next_vowpal_with_cache)
    bool resumable = false;
    std::string tail;           // prefix bytes from the resume point on
    std::vector<uint32_t> ob;   // output buffer at the resume point
    fwgpu::NsState st;
};

namespace fwgpu {

bool vwmap_source_equal(const fwgpu_vwmap *a, const fwgpu_vwmap *b) {  // VwNamespaceMapSource: derive(PartialEq)
    if (a->skip_prefix != b->skip_prefix || a->entries.size() != b->entries.size()) return false;
    for (size_t i = 0; i < a->entries.size(); i++) {
        const VwEntry &x = a->entries[i], &y = b->entries[i];
        if (x.vwname != y.vwname || x.verbose != y.verbose || x.index != y.index || x.f32 != y.f32) return false;
    }
    return true;
}

std::string vwmap_json(const fwgpu_vwmap *vw) {  // serde_json::to_vec_pretty(&vw_source), persistence.rs:37-42
    fwjson::Writer w;
    w.begin_obj();
    w.key("namespace_skip_prefix");
    w.u64(vw->skip_prefix);
    w.key("entries");
    w.begin_arr();
    for (const auto &e : vw->entries) {
        w.begin_obj();
        w.key("namespace_vwname");
        w.str(e.vwname);
        w.key("namespace_verbose");
        w.str(e.verbose);
        w.key("namespace_index");
        w.u64(e.index);
        w.key("namespace_format");
        w.str(e.f32 ? "F32" : "Categorical");
        w.end_obj();
    }
    w.end_arr();
    w.end_obj();
    return w.out;
}

}  // namespace fwgpu

using namespace fwgpu;

extern "C" {

int fwgpu_vwmap_from_csv(const char *csv, uint64_t len, fwgpu_vwmap **out) {
    if (!csv || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    auto vw = std::make_unique<fwgpu_vwmap>();
    const auto recs = csv_records(csv, len);
    for (size_t i = 0; i < recs.size(); i++) {
        const auto &r = recs[i];
        if (r[0] == "_namespace_skip_prefix") {  // vwmap.rs:123-133
            if (r.size() < 2) return fail(FWGPU_ERR_PARSE, "Couldn't parse _namespace_skip_prefix in vw_namespaces_map.csv");
            char *end = nullptr;
            const unsigned long v = std::strtoul(r[1].c_str(), &end, 10);
            if (r[1].empty() || *end) return fail(FWGPU_ERR_PARSE, "Couldn't parse _namespace_skip_prefix in vw_namespaces_map.csv");
            vw->skip_prefix = (uint32_t)v;
            continue;
        }
        if (r.size() < 2) return fail(FWGPU_ERR_PARSE, "vw_namespace_map.csv: a record needs a vw name and a verbose name");
        bool f32 = false;
        if (r.size() > 2) {
            if (r[2] == "f32") f32 = true;
            else if (!r[2].empty())
                return fail(FWGPU_ERR_PARSE, "Unknown type used for the feature in vw_namespace_map.csv: \"" + r[2] +
                                                 "\". Only \"f32\" is possible.");
        }
        vw->entries.push_back({r[0], r[1], (uint32_t)(i & 0xffff), f32});  // namespace_index: i as u16
    }
    vw->finish();
    *out = vw.release();
    return FWGPU_OK;
}

int fwgpu_vwmap_from_json(const char *json, uint64_t len, fwgpu_vwmap **out) {
    if (!json || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    try {
        const fwjson::Value v = fwjson::Parser(json, len).parse();
        auto vw = std::make_unique<fwgpu_vwmap>();
        vw->skip_prefix = (uint32_t)v.at("namespace_skip_prefix").as_u64();
        for (const auto &e : v.at("entries").arr) {
            const std::string &fmt = e.at("namespace_format").as_str();
            if (fmt != "Categorical" && fmt != "F32") throw std::runtime_error("unknown variant `" + fmt + "`");
            vw->entries.push_back({e.at("namespace_vwname").as_str(), e.at("namespace_verbose").as_str(),
                                   (uint32_t)e.at("namespace_index").as_u64(), fmt == "F32"});
        }
        vw->finish();
        *out = vw.release();
        return FWGPU_OK;
    } catch (const std::exception &ex) {
        return fail(FWGPU_ERR_FORMAT, std::string("vw_source JSON: ") + ex.what());
    }
}

void fwgpu_vwmap_free(fwgpu_vwmap *vw) { delete vw; }

uint32_t fwgpu_vwmap_num_namespaces(const fwgpu_vwmap *vw) { return vw ? vw->num_namespaces : 0; }
uint32_t fwgpu_vwmap_num_entries(const fwgpu_vwmap *vw) { return vw ? (uint32_t)vw->entries.size() : 0; }

int fwgpu_vwmap_to_json(const fwgpu_vwmap *vw, char *buf, uint64_t cap, uint64_t *len) {
    if (!vw || !len) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const std::string s = vwmap_json(vw);
    *len = s.size();
    if (!buf) return FWGPU_OK;  // size query
    if (cap < s.size()) return fail(FWGPU_ERR_RANGE, "buffer too small for the vw_source JSON");
    std::memcpy(buf, s.data(), s.size());
    return FWGPU_OK;
}

int fwgpu_vwmap_lookup(const fwgpu_vwmap *vw, const char *name, uint64_t len, int verbose, uint32_t *index,
                       uint32_t *is_f32) {
    if (!vw || !name) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const auto &m = verbose ? vw->by_verbose : vw->by_vwname;
    auto it = m.find(std::string(name, len));
    if (it == m.end()) return fail(FWGPU_ERR_INVALID, "Unknown namespace: " + std::string(name, len));
    if (index) *index = vw->entries[it->second].index;
    if (is_f32) *is_f32 = vw->entries[it->second].f32;
    return FWGPU_OK;
}

int fwgpu_parser_create(const fwgpu_vwmap *vw, fwgpu_parser **out) {
    if (!vw || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    auto p = std::make_unique<fwgpu_parser>();
    p->vw_copy = *vw;
    p->vw = &p->vw_copy;
    for (const auto &e : vw->entries)  // murmur3::hash32(vwname): seed 0 (parser.rs:82-84)
        p->seed.push_back(murmur3_32(reinterpret_cast<const uint8_t *>(e.vwname.data()), e.vwname.size(), 0));
    p->build_ns_table();
    *out = p.release();
    return FWGPU_OK;
}

void fwgpu_parser_free(fwgpu_parser *p) { delete p; }

int fwgpu_parser_clone(const fwgpu_parser *src, fwgpu_parser **out) {  // VowpalParser: derive(Clone) (parser.rs:23)
    if (!src || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    return fwgpu_parser_create(&src->vw_copy, out);
}

int fwgpu_parser_parse_line(fwgpu_parser *p, const char *line, uint64_t len, uint32_t *out, uint32_t cap, uint32_t *n_words) {
    if (!p || !n_words || (!line && len)) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *n_words = 0;
    if (len == 0) return FWGPU_OK;  // end of stream: empty record (parser.rs:172)
    p->scratch.assign(line, len);
    p->scratch.push_back('\0');
    int rc;
    try {
        rc = parse_line(p, p->scratch.data(), len);
    } catch (const std::exception &e) {  // nothing may unwind across the C ABI
        return fail(FWGPU_ERR_PARSE, std::string("Cannot parse an example: ") + e.what());
    }
    if (rc != FWGPU_OK) return rc;
    *n_words = (uint32_t)p->out.size();
    if (!out) return FWGPU_OK;
    if (cap < p->out.size()) return fail(FWGPU_ERR_RANGE, "record buffer too small");
    std::memcpy(out, p->out.data(), p->out.size() * 4);
    return FWGPU_OK;
}

int fwgpu_parser_parse_with_prefix(fwgpu_parser *p, const char *prefix, uint64_t prefix_len, const char *line, uint64_t len,
                                   uint32_t *out, uint32_t cap, uint32_t *n_words) {
    // next_vowpal_with_cache (parser.rs:195-211): the cached context text, truncated to its size, then the new bytes
    if (!p || !n_words) return fail(FWGPU_ERR_INVALID, "NULL argument");
    std::string s(prefix ? prefix : "", prefix ? prefix_len : 0);
    s.append(line ? line : "", line ? len : 0);
    return fwgpu_parser_parse_line(p, s.data(), s.size(), out, cap, n_words);
}

// next_vowpal_with_cache (parser.rs:195-211) without scanning the context again for every request.  The scanner is a
// left-to-right state machine over tokens that end at a space, and everything it knows between two tokens is `NsState` + the
// output buffer; so the context is scanned once up to its last token boundary whose token ended at a space INSIDE the context
// (a last token that runs to the context's end may continue in the request), and a request resumes from there over the
// context's remaining bytes + its own.  Contexts the head of the scanner does not leave inside the context (no '|', a
// command, an error) are not resumable and take the concatenating route; so does an empty request.
int fwgpu_parse_prefix_create(fwgpu_parser *p, const char *prefix, uint64_t len, fwgpu_parse_prefix **out) {
    if (!p || !out || (!prefix && len)) return fail(FWGPU_ERR_INVALID, "NULL argument");
    auto px = std::make_unique<fwgpu_parse_prefix>();
    px->text.assign(prefix ? prefix : "", len);
    *out = nullptr;
    if (len && (prefix[0] == 0x31 || prefix[0] == 0x2d || prefix[0] == 0x7c)) {
        p->scratch.assign(prefix, len);
        p->scratch.push_back('\0');
        const unsigned char *q = reinterpret_cast<const unsigned char *>(p->scratch.data());
        const size_t rowlen = len;  // as if more bytes followed: the request's
        size_t pos = 0;
        int rc = FWGPU_ERR_PARSE;
        try {
            rc = parse_head(p, q, len, rowlen, &pos);
        } catch (const std::exception &) {
        }
        if (rc == FWGPU_OK && pos < len && q[pos] == 0x7c) {
            // token boundaries of the feature loop: the last one whose token (and every one before it) ended at a space
            size_t top = pos, i = pos;
            for (;;) {
                while (q[i] == 0x20 && i < rowlen) i++;
                while (q[i] != 0x20 && i < rowlen) i++;
                if (i >= rowlen) break;
                i += 1;
                top = i;
            }
            fwgpu::NsState st;
            try {
                rc = parse_body(p, q, rowlen, pos, top, st);
            } catch (const std::exception &) {
                rc = FWGPU_ERR_PARSE;
            }
            if (rc == FWGPU_OK) {
                px->resumable = true;
                px->tail.assign(prefix + top, len - top);
                px->ob = p->out;
                px->st = st;
            }
        }
    }
    *out = px.release();
    return FWGPU_OK;
}

void fwgpu_parse_prefix_free(fwgpu_parse_prefix *px) { delete px; }

int fwgpu_parse_prefix_resumable(const fwgpu_parse_prefix *px) { return px && px->resumable; }

int fwgpu_parser_parse_after_prefix(fwgpu_parser *p, const fwgpu_parse_prefix *px, const char *line, uint64_t len, uint32_t *out,
                                    uint32_t cap, uint32_t *n_words) {
    if (!p || !px || !n_words || (!line && len)) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (!px->resumable || len == 0) return fwgpu_parser_parse_with_prefix(p, px->text.data(), px->text.size(), line, len, out, cap, n_words);
    *n_words = 0;
    p->scratch.assign(px->tail);
    p->scratch.append(line, len);
    p->scratch.push_back('\0');
    const size_t size = px->tail.size() + len;
    p->out = px->ob;
    fwgpu::NsState st = px->st;
    int rc;
    try {
        rc = parse_body(p, reinterpret_cast<const unsigned char *>(p->scratch.data()), size - 1, 0, size - 1, st);
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_PARSE, std::string("Cannot parse an example: ") + e.what());
    }
    if (rc != FWGPU_OK) return rc;
    p->out[0] = (uint32_t)p->out.size();
    *n_words = (uint32_t)p->out.size();
    if (!out) return FWGPU_OK;
    if (cap < p->out.size()) return fail(FWGPU_ERR_RANGE, "record buffer too small");
    std::memcpy(out, p->out.data(), p->out.size() * 4);
    return FWGPU_OK;
}

// 1 when the scanned part of the context is the whole of it: the context's own record (what fw_setup_cache parsed, `record`)
// is word for word the output buffer at the resume point.  Requests can then be returned as candidate-only records.
int fwgpu_parse_prefix_is_record(const fwgpu_parse_prefix *px, const uint32_t *record, uint32_t len) {
    if (!px || !px->resumable || !record || len != px->ob.size() || len < kHeaderLen) return 0;
    return std::memcmp(record + 1, px->ob.data() + 1, ((size_t)len - 1) * 4) == 0;  // (word 0 is the length, written last)
}

// fwgpu_parser_parse_after_prefix, but what comes back (*is_delta = 1) holds only what the request ADDED to the context's
// record: the header, a slot word for every namespace the request filled (feature ranges relocated to this record; every other
// slot NO_FEATURES = "as in the context") and the request's feature words.  The context's record + this one carry the same
// information as the merged record; a request that goes on inside a namespace the context began cannot be split that way and
// comes back merged (*is_delta = 0), as does everything when the context is not resumable.
int fwgpu_parser_parse_candidate(fwgpu_parser *p, const fwgpu_parse_prefix *px, const char *line, uint64_t len, uint32_t *out,
                                 uint32_t cap, uint32_t *n_words, int *is_delta) {
    if (!is_delta) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *is_delta = 0;
    int rc = fwgpu_parser_parse_after_prefix(p, px, line, len, nullptr, 0, n_words);
    if (rc != FWGPU_OK || *n_words == 0) return rc;
    const std::vector<uint32_t> &m = p->out;
    const size_t L0 = px->ob.size(), H = kHeaderLen + p->vw_copy.num_namespaces;
    bool delta = px->resumable && len != 0 && m.size() >= L0 && L0 >= H;
    std::vector<uint32_t> &d = p->delta;
    if (delta) {
        d.resize(H + (m.size() - L0));
        d[kLabelOffset] = m[kLabelOffset];
        d[kImportanceOffset] = m[kImportanceOffset];
        for (size_t sl = kHeaderLen; sl < H && delta; sl++) {
            const uint32_t w = m[sl];
            if (w == px->ob[sl]) {
                d[sl] = kNoFeatures;
            } else if (!(w & kNotSingle)) {
                d[sl] = w;
            } else {
                const uint32_t start = (w >> 16) & 0x3fff, end = w & 0xffff;
                if (start < L0 || end < start) delta = false;  // the namespace began in the context
                else d[sl] = kNotSingle | (uint32_t)(((start - L0 + H) << 16) + (end - L0 + H));
            }
        }
        if (delta) {
            std::memcpy(d.data() + H, m.data() + L0, (m.size() - L0) * 4);
            d[0] = (uint32_t)d.size();
        }
    }
    const std::vector<uint32_t> &res = delta ? d : m;
    *is_delta = delta ? 1 : 0;
    *n_words = (uint32_t)res.size();
    if (!out) return FWGPU_OK;
    if (cap < res.size()) return fail(FWGPU_ERR_RANGE, "record buffer too small");
    std::memcpy(out, res.data(), res.size() * 4);
    return FWGPU_OK;
}

int fwgpu_debug_format_f32(float v, char *buf, uint32_t cap) {
    const std::string t = fwjson::format_f32(v);
    if (!buf || cap < t.size() + 1) return fail(FWGPU_ERR_RANGE, "buffer too small");
    std::memcpy(buf, t.c_str(), t.size() + 1);
    return FWGPU_OK;
}

const char *fwgpu_parser_command_argument(const fwgpu_parser *p) { return p ? p->cmd_arg.c_str() : ""; }

int fwgpu_parser_parse_buffer(fwgpu_parser *p, const char *text, uint64_t len, uint32_t *words, uint64_t words_cap,
                              uint64_t *rec_off, uint64_t max_records, uint64_t *n_records, uint64_t *n_words,
                              uint64_t *consumed) {
    if (!p || !text || !words || !rec_off || !n_records || !n_words || !consumed) return fail(FWGPU_ERR_INVALID, "NULL argument");
    uint64_t pos = 0, nr = 0, nw = 0;
    rec_off[0] = 0;
    while (pos < len && nr < max_records) {
        const char *nl = static_cast<const char *>(std::memchr(text + pos, '\n', len - pos));
        const uint64_t line_len = nl ? (uint64_t)(nl - (text + pos)) + 1 : len - pos;
        int rc;
        try {
            if (pos + line_len < len) {
                rc = parse_line(p, text + pos, line_len);  // in place: the next line's first byte is readable
            } else {
                p->scratch.assign(text + pos, line_len);
                p->scratch.push_back('\0');
                rc = parse_line(p, p->scratch.data(), line_len);
            }
        } catch (const std::exception &e) {
            rc = fail(FWGPU_ERR_PARSE, std::string("Cannot parse an example: ") + e.what());
        }
        if (rc != FWGPU_OK) {
            *n_records = nr;
            *n_words = nw;
            *consumed = pos;  // the offending / command line starts here
            return rc;
        }
        if (nw + p->out.size() > words_cap) break;  // caller drains and calls again from `consumed`
        std::memcpy(words + nw, p->out.data(), p->out.size() * 4);
        nw += p->out.size();
        rec_off[++nr] = nw;
        pos += line_len;
    }
    *n_records = nr;
    *n_words = nw;
    *consumed = pos;
    return FWGPU_OK;
}

}  // extern "C"
