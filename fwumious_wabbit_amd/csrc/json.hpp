// Minimal JSON for the two serde_json documents the reference embeds in its cache and model files
// (persistence.rs:20-53): a DOM parser, and a writer that reproduces serde_json::to_vec_pretty byte for byte
// (two-space indent, "key": value, "[]" / "{}" for empty containers, ryu-style shortest floats with ".0").
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace fwjson {

struct Value {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string num_text;  // the literal as written (so u64 / f32 values survive untouched)
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;  // insertion order, like serde's struct order

    const Value *get(const std::string &key) const {
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Value &at(const std::string &key) const {
        const Value *v = get(key);
        if (!v) throw std::runtime_error("missing field `" + key + "`");
        return *v;
    }
    float as_f32() const {
        if (kind != Num) throw std::runtime_error("expected a number");
        return std::strtof(num_text.c_str(), nullptr);
    }
    uint64_t as_u64() const {
        if (kind != Num) throw std::runtime_error("expected a number");
        return std::strtoull(num_text.c_str(), nullptr, 10);
    }
    bool as_bool() const {
        if (kind != Bool) throw std::runtime_error("expected a boolean");
        return b;
    }
    const std::string &as_str() const {
        if (kind != Str) throw std::runtime_error("expected a string");
        return str;
    }
};

class Parser {
  public:
    Parser(const char *p, size_t n) : p_(p), e_(p + n) {}
    Value parse() {
        Value v = value();
        ws();
        if (p_ != e_) fail("trailing characters");
        return v;
    }

  private:
    const char *p_, *e_;
    int depth_ = 0;
    [[noreturn]] void fail(const char *what) { throw std::runtime_error(std::string("JSON: ") + what); }
    void ws() {
        while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    bool lit(const char *s) {
        const size_t n = std::strlen(s);
        if ((size_t)(e_ - p_) >= n && std::memcmp(p_, s, n) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }
    static void utf8(std::string &o, uint32_t c) {
        if (c < 0x80) {
            o += (char)c;
        } else if (c < 0x800) {
            o += (char)(0xc0 | (c >> 6));
            o += (char)(0x80 | (c & 0x3f));
        } else if (c < 0x10000) {
            o += (char)(0xe0 | (c >> 12));
            o += (char)(0x80 | ((c >> 6) & 0x3f));
            o += (char)(0x80 | (c & 0x3f));
        } else {
            o += (char)(0xf0 | (c >> 18));
            o += (char)(0x80 | ((c >> 12) & 0x3f));
            o += (char)(0x80 | ((c >> 6) & 0x3f));
            o += (char)(0x80 | (c & 0x3f));
        }
    }
    uint32_t hex4() {
        if (e_ - p_ < 4) fail("bad \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        std::string o;
        ++p_;  // opening quote
        while (true) {
            if (p_ >= e_) fail("unterminated string");
            const char c = *p_++;
            if (c == '"') break;
            if (c != '\\') {
                o += c;
                continue;
            }
            if (p_ >= e_) fail("unterminated escape");
            const char x = *p_++;
            switch (x) {
            case '"': o += '"'; break;
            case '\\': o += '\\'; break;
            case '/': o += '/'; break;
            case 'b': o += '\b'; break;
            case 'f': o += '\f'; break;
            case 'n': o += '\n'; break;
            case 'r': o += '\r'; break;
            case 't': o += '\t'; break;
            case 'u': {
                uint32_t c1 = hex4();
                if (c1 >= 0xd800 && c1 < 0xdc00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                    p_ += 2;
                    const uint32_t c2 = hex4();
                    c1 = 0x10000 + ((c1 - 0xd800) << 10) + (c2 - 0xdc00);
                }
                utf8(o, c1);
                break;
            }
            default: fail("bad escape");
            }
        }
        return o;
    }
    Value value() {
        struct Depth {  // the documents read here nest 5 deep; a hostile file must not overflow the stack
            int &d;
            explicit Depth(int &x) : d(x) { ++d; }
            ~Depth() { --d; }
        } guard(depth_);
        if (depth_ > 64) fail("nested too deeply");
        ws();
        if (p_ >= e_) fail("unexpected end");
        Value v;
        const char c = *p_;
        if (c == '{') {
            v.kind = Value::Obj;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == '}') {
                ++p_;
                return v;
            }
            while (true) {
                ws();
                if (p_ >= e_ || *p_ != '"') fail("expected a key");
                std::string k = string();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (p_ < e_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < e_ && *p_ == '}') {
                    ++p_;
                    break;
                }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = Value::Arr;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == ']') {
                ++p_;
                return v;
            }
            while (true) {
                v.arr.push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < e_ && *p_ == ']') {
                    ++p_;
                    break;
                }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Value::Str;
            v.str = string();
        } else if (lit("true")) {
            v.kind = Value::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = Value::Bool;
        } else if (lit("null")) {
            v.kind = Value::Null;
        } else {
            const char *s = p_;
            if (p_ < e_ && *p_ == '-') ++p_;
            while (p_ < e_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) ++p_;
            if (p_ == s) fail("unexpected character");
            v.kind = Value::Num;
            v.num_text.assign(s, p_);
            v.num = std::strtod(v.num_text.c_str(), nullptr);
        }
        return v;
    }
};

// f32 as serde_json prints it (ryu::Buffer::format_finite for f32): shortest digits that round-trip, laid out as
// ddd.0 / d.ddd / 0.00ddd for decimal exponents in (-5, 16], exponent form d.ddde-7 otherwise.  Non-finite -> null.
inline std::string format_f32(float f) {
    if (!std::isfinite(f)) return "null";
    if (f == 0.0f) return std::signbit(f) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, f, std::chars_format::scientific);  // shortest round-trip digits
    std::string sci(buf, r.ptr);
    std::string out;
    size_t i = 0;
    if (sci[0] == '-') {
        out += '-';
        i = 1;
    }
    const size_t epos = sci.find('e');
    std::string digits;
    for (size_t j = i; j < epos; j++)
        if (sci[j] != '.') digits += sci[j];
    const int exp10 = std::atoi(sci.c_str() + epos + 1);
    const int k = (int)digits.size();
    const int kk = exp10 + 1;  // position of the decimal point relative to the first digit
    if (k <= kk && kk <= 13) {
        // 1234e7 -> 12340000000.0   (ryu f32 pretty printer: integer layouts up to 13 digits)
        out += digits;
        out.append((size_t)(kk - k), '0');
        out += ".0";
    } else if (0 < kk && kk <= 13) {
        out += digits.substr(0, (size_t)kk);
        out += '.';
        out += digits.substr((size_t)kk);
    } else if (-6 < kk && kk <= 0) {
        out += "0.";
        out.append((size_t)(-kk), '0');
        out += digits;
    } else if (k == 1) {
        out += digits;
        out += 'e';
        out += std::to_string(kk - 1);
    } else {
        out += digits[0];
        out += '.';
        out += digits.substr(1);
        out += 'e';
        out += std::to_string(kk - 1);
    }
    return out;
}

// serde_json pretty writer
class Writer {
  public:
    std::string out;
    void begin_obj() { open('{'); }
    void end_obj() { close('}'); }
    void begin_arr() { open('['); }
    void end_arr() { close(']'); }
    void key(const char *k) {
        sep();
        str_raw(k);
        out += ": ";
        after_key_ = true;
    }
    void str(const std::string &s) {
        sep();
        str_raw(s);
    }
    void u64(uint64_t v) {
        sep();
        out += std::to_string(v);
    }
    void f32(float v) {
        sep();
        out += format_f32(v);
    }
    void boolean(bool v) {
        sep();
        out += v ? "true" : "false";
    }
    void null() {
        sep();
        out += "null";
    }
    void raw(const std::string &text) {
        sep();
        out += text;
    }

  private:
    struct Level {
        bool has_items;
    };
    std::vector<Level> stack_;
    bool after_key_ = false;
    void indent() {
        out += '\n';
        out.append(2 * stack_.size(), ' ');
    }
    void sep() {
        if (after_key_) {
            after_key_ = false;
            return;
        }
        if (!stack_.empty()) {
            if (stack_.back().has_items) out += ',';
            stack_.back().has_items = true;
            indent();
        }
    }
    void open(char c) {
        sep();
        out += c;
        stack_.push_back({false});
    }
    void close(char c) {
        const bool had = stack_.back().has_items;
        stack_.pop_back();
        if (had) indent();
        out += c;
    }
    void str_raw(const std::string &s) {
        out += '"';
        for (unsigned char c : s) {
            switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) {
                    char b[8];
                    std::snprintf(b, sizeof b, "\\u%04x", c);
                    out += b;
                } else {
                    out += (char)c;
                }
            }
        }
        out += '"';
    }
};

}  // namespace fwjson
