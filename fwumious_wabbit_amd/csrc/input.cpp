// Input files (buffer_handler.rs:8-37 create_buffered_input): ".vw" as it is, ".gz" through zlib (MultiGzDecoder: several
// gzip members back to back), ".zst" through libzstd's streaming API (resolved at run time: the image ships the library
// but not its header).  Plus the example loop over such a file (main.rs:213-270) in native code.
#include <dlfcn.h>
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fwgpu_internal.h"

namespace {

// the four libzstd entry points the streaming decoder needs (zstd.h, stable ABI since 1.0)
struct ZstdApi {
    struct InBuf {
        const void *src;
        size_t size, pos;
    };
    struct OutBuf {
        void *dst;
        size_t size, pos;
    };
    void *lib = nullptr;
    void *(*createDStream)() = nullptr;
    size_t (*freeDStream)(void *) = nullptr;
    size_t (*decompressStream)(void *, OutBuf *, InBuf *) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char *(*getErrorName)(size_t) = nullptr;
    std::once_flag once;
    bool ok = false;
    bool load() {  // several threads may open .zst inputs at the same time
        std::call_once(once, [this] { ok = load_once(); });
        return ok;
    }
    bool load_once() {
        for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        createDStream = reinterpret_cast<void *(*)()>(dlsym(lib, "ZSTD_createDStream"));
        freeDStream = reinterpret_cast<size_t (*)(void *)>(dlsym(lib, "ZSTD_freeDStream"));
        decompressStream = reinterpret_cast<size_t (*)(void *, OutBuf *, InBuf *)>(dlsym(lib, "ZSTD_decompressStream"));
        isError = reinterpret_cast<unsigned (*)(size_t)>(dlsym(lib, "ZSTD_isError"));
        getErrorName = reinterpret_cast<const char *(*)(size_t)>(dlsym(lib, "ZSTD_getErrorName"));
        return createDStream && freeDStream && decompressStream && isError;
    }
};
ZstdApi g_zstd;

}  // namespace

struct fwgpu_input {
    enum Kind { Plain, Gz, Zst } kind = Plain;
    FILE *f = nullptr;
    std::vector<unsigned char> in;  // compressed bytes read from the file
    size_t in_pos = 0, in_len = 0;
    bool eof = false;
    z_stream z{};
    bool z_open = false, z_mid_member = false;  // inside a gzip member (its trailer not reached yet)
    void *zd = nullptr;
    ~fwgpu_input() {
        if (z_open) inflateEnd(&z);
        if (zd) g_zstd.freeDStream(zd);
        if (f) std::fclose(f);
    }
    bool refill() {
        if (in_pos < in_len) return true;
        if (eof) return false;
        in_len = std::fread(in.data(), 1, in.size(), f);
        in_pos = 0;
        if (in_len == 0) eof = true;
        return in_len > 0;
    }
};

using namespace fwgpu;

extern "C" {

int fwgpu_input_open(const char *filename, fwgpu_input **out) {
    if (!filename || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const std::string name(filename);
    const size_t dot = name.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : name.substr(dot + 1);
    auto h = std::make_unique<fwgpu_input>();
    if (ext == "vw") h->kind = fwgpu_input::Plain;
    else if (ext == "gz") h->kind = fwgpu_input::Gz;
    else if (ext == "zst") h->kind = fwgpu_input::Zst;
    else return fail(FWGPU_ERR_INVALID, "Please specify a valid input format (.vw, .zst, .gz)");  // buffer_handler.rs:33-35
    h->f = std::fopen(filename, "rb");
    if (!h->f) return fail(FWGPU_ERR_IO, "Could not open the input file.");
    h->in.resize(1 << 20);
    if (h->kind == fwgpu_input::Gz) {
        if (inflateInit2(&h->z, 15 + 16) != Z_OK) return fail(FWGPU_ERR_IO, "zlib: inflateInit2 failed");
        h->z_open = true;
    } else if (h->kind == fwgpu_input::Zst) {
        if (!g_zstd.load()) return fail(FWGPU_ERR_IO, "libzstd.so.1 is not available on this machine: cannot read .zst input");
        h->zd = g_zstd.createDStream();
        if (!h->zd) return fail(FWGPU_ERR_OOM, "zstd: cannot create a decompression stream");
    }
    *out = h.release();
    return FWGPU_OK;
}

// reads up to cap decompressed bytes; *n == 0 at end of input
int fwgpu_input_read(fwgpu_input *h, char *buf, uint64_t cap, uint64_t *n) {
    if (!h || !buf || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *n = 0;
    if (h->kind == fwgpu_input::Plain) {
        *n = std::fread(buf, 1, cap, h->f);
        return FWGPU_OK;
    }
    uint64_t got = 0;
    while (got < cap) {
        if (!h->refill()) {
            if (h->kind == fwgpu_input::Gz && h->z_mid_member && got == 0)
                return fail(FWGPU_ERR_FORMAT, "gzip input: unexpected end of file");  // flate2: UnexpectedEof
            break;
        }
        if (h->kind == fwgpu_input::Gz) {
            h->z.next_in = h->in.data() + h->in_pos;
            h->z.avail_in = (uInt)(h->in_len - h->in_pos);
            h->z.next_out = reinterpret_cast<Bytef *>(buf + got);
            h->z.avail_out = (uInt)std::min<uint64_t>(cap - got, 1u << 30);
            const uInt out0 = h->z.avail_out;
            const int rc = inflate(&h->z, Z_NO_FLUSH);
            h->in_pos = h->in_len - h->z.avail_in;
            got += out0 - h->z.avail_out;
            h->z_mid_member = rc != Z_STREAM_END;
            if (rc == Z_STREAM_END) {
                // MultiGzDecoder: another member may follow
                if (h->refill() || h->in_pos < h->in_len) inflateReset(&h->z);
                else break;
            } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                return fail(FWGPU_ERR_FORMAT, std::string("gzip input: ") + (h->z.msg ? h->z.msg : "corrupt stream"));
            }
        } else {
            ZstdApi::InBuf ib{h->in.data(), h->in_len, h->in_pos};
            ZstdApi::OutBuf ob{buf, (size_t)cap, (size_t)got};
            const size_t rc = g_zstd.decompressStream(h->zd, &ob, &ib);
            h->in_pos = ib.pos;
            got = ob.pos;
            if (g_zstd.isError(rc))
                return fail(FWGPU_ERR_FORMAT, std::string("zstd input: ") + (g_zstd.getErrorName ? g_zstd.getErrorName(rc) : "corrupt stream"));
        }
    }
    *n = got;
    return FWGPU_OK;
}

void fwgpu_input_close(fwgpu_input *h) { delete h; }

// main.rs:213-270 over an input FILE: windows of decompressed text are cut at their last line break and handed to
// fwgpu_trainer_digest_text.  Stops at the first line that is not an example, like digest_text.
int fwgpu_trainer_digest_file(fwgpu_trainer *tr, fwgpu_parser *parser, fwgpu_cache *cache, const char *filename,
                              uint32_t threads, uint64_t *n_examples) {
    if (!tr || !parser || !filename) return fail(FWGPU_ERR_INVALID, "NULL argument");
    fwgpu_input *in = nullptr;
    int rc = fwgpu_input_open(filename, &in);
    if (rc) return rc;
    std::unique_ptr<fwgpu_input> guard(in);
    const uint64_t window = 64ull << 20;
    std::unique_ptr<char[]> buf(new char[window + 1]);
    uint64_t have = 0, total = 0;
    bool end = false;
    while (!end || have) {
        while (!end && have < window) {
            uint64_t n = 0;
            rc = fwgpu_input_read(in, buf.get() + have, window - have, &n);
            if (rc) return rc;
            if (n == 0) end = true;
            have += n;
        }
        if (!have) break;
        uint64_t cut = have;
        if (!end) {  // keep the incomplete last line for the next window
            while (cut > 0 && buf[cut - 1] != '\n') cut--;
            if (cut == 0) {
                if (have == window) return fail(FWGPU_ERR_RANGE, "a single line is longer than 64 MiB");
                continue;
            }
        }
        uint64_t n = 0, used = 0;
        rc = fwgpu_trainer_digest_text(tr, parser, cache, buf.get(), cut, threads, &n, &used);
        total += n;
        if (n_examples) *n_examples = total;
        if (rc != FWGPU_OK || used < cut) return rc;
        std::memmove(buf.get(), buf.get() + cut, have - cut);
        have -= cut;
    }
    if (n_examples) *n_examples = total;
    return FWGPU_OK;
}

}  // extern "C"
