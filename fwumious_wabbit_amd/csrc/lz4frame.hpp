// LZ4 frame format (magic 0x184D2204) reader and writer for the `.fwcache` files of gz inputs (cache.rs:88-92, 112-121:
// the reference wraps the cache stream in lz4::Decoder / lz4::Encoder, i.e. liblz4's frame API).  The `lz4` crate is a
// third-party dependency that is not vendored in the reference tree; this restates the published frame + block formats:
// linked or independent blocks, optional block / content checksums (xxHash32), optional content size and dictionary id.
// The writer emits independent 4 MiB blocks with a content checksum, which every conforming reader accepts.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace fwlz4 {

inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void wr32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

// xxHash32 (streaming), used for the frame's header, block and content checksums
class XXH32 {
  public:
    explicit XXH32(uint32_t seed = 0) { reset(seed); }
    void reset(uint32_t seed) {
        v_[0] = seed + P1 + P2;
        v_[1] = seed + P2;
        v_[2] = seed;
        v_[3] = seed - P1;
        total_ = 0;
        fill_ = 0;
        seed_ = seed;
    }
    void update(const uint8_t *p, size_t n) {
        total_ += n;
        if (fill_ + n < 16) {
            std::memcpy(buf_ + fill_, p, n);
            fill_ += n;
            return;
        }
        if (fill_) {
            const size_t take = 16 - fill_;
            std::memcpy(buf_ + fill_, p, take);
            stripe(buf_);
            p += take;
            n -= take;
            fill_ = 0;
        }
        while (n >= 16) {
            stripe(p);
            p += 16;
            n -= 16;
        }
        std::memcpy(buf_, p, n);
        fill_ = n;
    }
    uint32_t digest() const {
        uint32_t h = total_ >= 16 ? rotl(v_[0], 1) + rotl(v_[1], 7) + rotl(v_[2], 12) + rotl(v_[3], 18) : seed_ + P5;
        h += (uint32_t)total_;
        size_t i = 0;
        for (; i + 4 <= fill_; i += 4) h = rotl(h + rd32(buf_ + i) * P3, 17) * P4;
        for (; i < fill_; i++) h = rotl(h + buf_[i] * P5, 11) * P1;
        h ^= h >> 15;
        h *= P2;
        h ^= h >> 13;
        h *= P3;
        h ^= h >> 16;
        return h;
    }
    static uint32_t hash(const uint8_t *p, size_t n, uint32_t seed = 0) {
        XXH32 x(seed);
        x.update(p, n);
        return x.digest();
    }

  private:
    static constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
    void stripe(const uint8_t *p) {
        for (int i = 0; i < 4; i++) v_[i] = rotl(v_[i] + rd32(p + 4 * i) * P2, 13) * P1;
    }
    uint32_t v_[4], seed_;
    uint64_t total_;
    uint8_t buf_[16];
    size_t fill_;
};

// Decodes one LZ4 block into buf[pos..cap); buf[0..pos) is history the block may refer back to.  Returns the new end.
inline size_t decode_block(const uint8_t *src, size_t n, uint8_t *buf, size_t pos, size_t cap) {
    const uint8_t *p = src, *e = src + n;
    while (p < e) {
        const unsigned tok = *p++;
        size_t lit = tok >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (p >= e) throw std::runtime_error("LZ4: truncated literal length");
                b = *p++;
                lit += b;
            } while (b == 255);
        }
        if ((size_t)(e - p) < lit) throw std::runtime_error("LZ4: literals run past the block");
        if (cap - pos < lit) throw std::runtime_error("LZ4: block decodes past the declared maximum");
        std::memcpy(buf + pos, p, lit);
        pos += lit;
        p += lit;
        if (p >= e) break;  // the last sequence has no match part
        if (e - p < 2) throw std::runtime_error("LZ4: truncated offset");
        const size_t offset = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
        if (offset == 0 || offset > pos) throw std::runtime_error("LZ4: offset outside the window");
        size_t ml = tok & 15;
        if (ml == 15) {
            unsigned b;
            do {
                if (p >= e) throw std::runtime_error("LZ4: truncated match length");
                b = *p++;
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (cap - pos < ml) throw std::runtime_error("LZ4: block decodes past the declared maximum");
        // overlapping copies replicate the last `offset` bytes: copy in non-overlapping pieces that double each time
        uint8_t *d = buf + pos;
        const uint8_t *sfrom = d - offset;
        size_t done = 0;
        while (done < ml) {
            const size_t piece = std::min(ml - done, offset + done);
            std::memcpy(d + done, sfrom, piece);
            done += piece;
        }
        pos += ml;
    }
    return pos;
}

// Greedy single-pass LZ4 block compressor (hash of 4 bytes -> last position).  Returns the compressed size.
inline size_t encode_block(const uint8_t *src, size_t n, std::vector<uint8_t> &dst) {
    dst.clear();
    const size_t kMinMatch = 4, kLastLiterals = 5, kMfLimit = 12;
    std::vector<int32_t> table(1 << 16, -1);
    auto hash4 = [](const uint8_t *p) { return (rd32(p) * 2654435761u) >> 16; };
    size_t anchor = 0, i = 0;
    auto emit = [&](size_t lit_start, size_t lit_len, size_t offset, size_t match_len) {
        const size_t ml = match_len ? match_len - kMinMatch : 0;
        const uint8_t tok = (uint8_t)(((lit_len < 15 ? lit_len : 15) << 4) | (match_len ? (ml < 15 ? ml : 15) : 0));
        dst.push_back(tok);
        if (lit_len >= 15) {
            size_t r = lit_len - 15;
            while (r >= 255) dst.push_back(255), r -= 255;
            dst.push_back((uint8_t)r);
        }
        dst.insert(dst.end(), src + lit_start, src + lit_start + lit_len);
        if (match_len) {
            dst.push_back((uint8_t)offset);
            dst.push_back((uint8_t)(offset >> 8));
            if (ml >= 15) {
                size_t r = ml - 15;
                while (r >= 255) dst.push_back(255), r -= 255;
                dst.push_back((uint8_t)r);
            }
        }
    };
    if (n >= kMfLimit + 1) {
        const size_t limit = n - kMfLimit;  // last match must start before this
        while (i < limit) {
            const uint32_t h = hash4(src + i);
            const int32_t cand = table[h];
            table[h] = (int32_t)i;
            if (cand >= 0 && i - (size_t)cand <= 65535 && rd32(src + cand) == rd32(src + i)) {
                size_t ml = 4;
                const size_t max_ml = n - kLastLiterals - i;
                while (ml < max_ml && src[cand + ml] == src[i + ml]) ml++;
                emit(anchor, i - anchor, i - (size_t)cand, ml);
                i += ml;
                anchor = i;
            } else {
                i++;
            }
        }
    }
    emit(anchor, n - anchor, 0, 0);
    return dst.size();
}

class FrameReader {
  public:
    explicit FrameReader(FILE *f) : f_(f) {}
    // reads up to n decompressed bytes; 0 at end of stream
    size_t read(uint8_t *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (pos_ == end_) {
                if (done_ || !next_block()) break;
            }
            const size_t take = std::min(n - got, end_ - pos_);
            std::memcpy(dst + got, window_.data() + pos_, take);
            pos_ += take;
            got += take;
        }
        return got;
    }

  private:
    FILE *f_;
    bool header_read_ = false, done_ = false;
    bool block_indep_ = true, block_checksum_ = false, content_checksum_ = false;
    size_t max_block_ = 0;
    std::vector<uint8_t> window_, block_;
    size_t pos_ = 0, end_ = 0;  // window_[pos_..end_) = decoded bytes not handed out yet; [0..pos_) = history
    XXH32 content_hash_;

    void need(uint8_t *b, size_t n, const char *what) {
        if (std::fread(b, 1, n, f_) != n) throw std::runtime_error(std::string("LZ4 frame: truncated ") + what);
    }
    void read_header() {
        uint8_t m[4];
        if (std::fread(m, 1, 4, f_) != 4) {  // empty stream
            done_ = true;
            return;
        }
        uint32_t magic = rd32(m);
        while ((magic & 0xfffffff0u) == 0x184D2A50u) {  // skippable frame
            uint8_t l[4];
            need(l, 4, "skippable frame");
            if (std::fseek(f_, (long)rd32(l), SEEK_CUR) != 0) throw std::runtime_error("LZ4 frame: bad skippable frame");
            need(m, 4, "magic");
            magic = rd32(m);
        }
        if (magic != 0x184D2204u) throw std::runtime_error("LZ4 frame: bad magic");
        uint8_t d[16];
        need(d, 2, "descriptor");
        const uint8_t flg = d[0], bd = d[1];
        if ((flg >> 6) != 1) throw std::runtime_error("LZ4 frame: unsupported version");
        block_indep_ = flg & 0x20;
        block_checksum_ = flg & 0x10;
        const bool has_size = flg & 0x08;
        content_checksum_ = flg & 0x04;
        const bool has_dict = flg & 0x01;
        const unsigned bs = (bd >> 4) & 7;
        if (bs < 4) throw std::runtime_error("LZ4 frame: bad block size id");
        max_block_ = (size_t)1 << (8 + 2 * bs);
        size_t n = 2;
        if (has_size) {
            need(d + n, 8, "content size");
            n += 8;
        }
        if (has_dict) {
            need(d + n, 4, "dictionary id");
            n += 4;
        }
        uint8_t hc;
        need(&hc, 1, "header checksum");
        if (((XXH32::hash(d, n) >> 8) & 0xff) != hc) throw std::runtime_error("LZ4 frame: header checksum mismatch");
        header_read_ = true;
        content_hash_.reset(0);
    }
    bool next_block() {
        if (!header_read_) {
            read_header();
            if (done_) return false;
        }
        uint8_t b[4];
        need(b, 4, "block size");
        const uint32_t bsz = rd32(b);
        if (bsz == 0) {  // EndMark
            if (content_checksum_) {
                need(b, 4, "content checksum");
                if (rd32(b) != content_hash_.digest()) throw std::runtime_error("LZ4 frame: content checksum mismatch");
            }
            header_read_ = false;  // a concatenated frame may follow
            pos_ = end_ = 0;
            read_header();
            if (done_) return false;
            return next_block();
        }
        const bool raw = bsz & 0x80000000u;
        const size_t len = bsz & 0x7fffffffu;
        if (len > max_block_) throw std::runtime_error("LZ4 frame: block larger than the declared maximum");
        block_.resize(len);
        need(block_.data(), len, "block");
        if (block_checksum_) {
            need(b, 4, "block checksum");
            if (rd32(b) != XXH32::hash(block_.data(), len)) throw std::runtime_error("LZ4 frame: block checksum mismatch");
        }
        // keep the last 64 KiB as history for linked blocks, drop everything already handed out
        const size_t keep = block_indep_ ? 0 : std::min<size_t>(end_, 65536);
        if (window_.size() < 65536 + max_block_) window_.resize(65536 + max_block_);
        if (keep && end_ > keep) std::memmove(window_.data(), window_.data() + end_ - keep, keep);
        pos_ = keep;
        if (raw) {
            std::memcpy(window_.data() + pos_, block_.data(), len);
            end_ = pos_ + len;
        } else {
            end_ = decode_block(block_.data(), len, window_.data(), pos_, pos_ + max_block_);
        }
        if (content_checksum_) content_hash_.update(window_.data() + pos_, end_ - pos_);
        return true;
    }
};

class FrameWriter {
  public:
    explicit FrameWriter(FILE *f) : f_(f) {
        uint8_t h[7];
        wr32(h, 0x184D2204u);
        h[4] = 0x40 | 0x20 | 0x04;  // version 01, independent blocks, content checksum
        h[5] = 7 << 4;              // 4 MiB blocks
        h[6] = (uint8_t)((XXH32::hash(h + 4, 2) >> 8) & 0xff);
        put(h, 7);
        buf_.reserve(kBlock);
    }
    void write(const uint8_t *p, size_t n) {
        hash_.update(p, n);
        while (n) {
            const size_t take = std::min(n, kBlock - buf_.size());
            buf_.insert(buf_.end(), p, p + take);
            p += take;
            n -= take;
            if (buf_.size() == kBlock) flush_block();
        }
    }
    void finish() {
        if (finished_) return;
        flush_block();
        uint8_t t[8];
        wr32(t, 0);
        wr32(t + 4, hash_.digest());
        put(t, 8);
        finished_ = true;
    }

  private:
    static constexpr size_t kBlock = 4u << 20;
    FILE *f_;
    std::vector<uint8_t> buf_, comp_;
    XXH32 hash_;
    bool finished_ = false;
    void put(const uint8_t *p, size_t n) {
        if (std::fwrite(p, 1, n, f_) != n) throw std::runtime_error("LZ4 frame: write failed");
    }
    void flush_block() {
        if (buf_.empty()) return;
        encode_block(buf_.data(), buf_.size(), comp_);
        uint8_t h[4];
        if (comp_.size() < buf_.size()) {
            wr32(h, (uint32_t)comp_.size());
            put(h, 4);
            put(comp_.data(), comp_.size());
        } else {
            wr32(h, (uint32_t)buf_.size() | 0x80000000u);
            put(h, 4);
            put(buf_.data(), buf_.size());
        }
        buf_.clear();
    }
};

}  // namespace fwlz4
