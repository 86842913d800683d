// Model files: "FWRE", u32 version 6, u64 + JSON of vw_source, u64 + JSON of ModelInstance, weights blob
// (persistence.rs:17-97, 127-187; regressor.rs:426-469; model_instance.rs:47-97), read AND written, including the
// inference conversion (main.rs:136-148: optimizer -> SGD, weights only) and the f16 bucket quantisation of the FFM
// weights (quantization.rs:42-98).  ModelInstance crosses the C ABI as an opaque handle built from / rendered to the
// same JSON the reference stores.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "fwgpu_internal.h"
#include "json.hpp"

namespace fwgpu {
std::string vwmap_json(const fwgpu_vwmap *vw);  // parser.cpp

struct ND {  // vwmap.rs:22-27 NamespaceDescriptor
    uint32_t index;
    bool transformed, f32;
};
}  // namespace fwgpu

struct fwgpu_model_instance {
    float learning_rate = 0.5f, minimum_learning_rate = 0.0f, power_t = 0.5f;
    uint32_t bit_precision = 18;
    bool add_constant_feature = true;
    std::vector<std::pair<std::vector<fwgpu::ND>, float>> combos;
    std::vector<std::vector<fwgpu::ND>> fields;
    uint32_t ffm_k = 0, ffm_bit_precision = 18;
    bool fastmath = true;
    std::string ffm_initialization_type = "default";
    float ffm_k_threshold = 0, ffm_init_center = 0, ffm_init_width = 0, ffm_init_zero_band = 0, ffm_init_acc_gradient = 0,
          init_acc_gradient = 1.0f, ffm_learning_rate = 0.5f, ffm_power_t = 0.5f, nn_init_acc_gradient = 0,
          nn_learning_rate = 0.02f, nn_power_t = 0.45f;
    std::vector<std::map<std::string, std::string>> nn_layers;
    std::string nn_topology = "one";
    int optimizer = FWGPU_OPT_SGD;
    fwjson::Value transform_namespaces;  // kept verbatim; must be empty to build a regressor here
    int dequantize_weights = 0;          // Option<bool>: -1 = null
    // flattened views handed out by fwgpu_mi_configs (valid while the handle lives)
    std::vector<uint32_t> combo_off, combo_ns, field_off, field_ns;
    std::vector<uint8_t> combo_f32, field_f32;
    std::vector<float> combo_w;
};

using namespace fwgpu;

namespace {

ND nd_from(const fwjson::Value &v) {
    ND d;
    d.index = (uint32_t)v.at("namespace_index").as_u64();
    const std::string &t = v.at("namespace_type").as_str(), &f = v.at("namespace_format").as_str();
    if (t != "Primitive" && t != "Transformed") throw std::runtime_error("unknown variant `" + t + "`");
    if (f != "Categorical" && f != "F32") throw std::runtime_error("unknown variant `" + f + "`");
    d.transformed = t == "Transformed";
    d.f32 = f == "F32";
    return d;
}
void nd_write(fwjson::Writer &w, const ND &d) {
    w.begin_obj();
    w.key("namespace_index");
    w.u64(d.index);
    w.key("namespace_type");
    w.str(d.transformed ? "Transformed" : "Primitive");
    w.key("namespace_format");
    w.str(d.f32 ? "F32" : "Categorical");
    w.end_obj();
}
void value_write(fwjson::Writer &w, const fwjson::Value &v) {  // verbatim re-emission of a parsed subtree
    switch (v.kind) {
    case fwjson::Value::Null: w.null(); break;
    case fwjson::Value::Bool: w.boolean(v.b); break;
    case fwjson::Value::Num:
        if (v.num_text.find_first_of(".eE") == std::string::npos) w.raw(v.num_text);  // integers stay integers
        else w.f32(std::strtof(v.num_text.c_str(), nullptr));
        break;
    case fwjson::Value::Str: w.str(v.str); break;
    case fwjson::Value::Arr:
        w.begin_arr();
        for (const auto &e : v.arr) value_write(w, e);
        w.end_arr();
        break;
    case fwjson::Value::Obj:
        w.begin_obj();
        for (const auto &kv : v.obj) {
            w.key(kv.first.c_str());
            value_write(w, kv.second);
        }
        w.end_obj();
        break;
    }
}
float opt_f32(const fwjson::Value &v, const char *k, float dflt) {  // #[serde(default = ...)]
    const fwjson::Value *x = v.get(k);
    return x ? x->as_f32() : dflt;
}

void mi_parse(const fwjson::Value &v, fwgpu_model_instance *m) {  // model_instance.rs:47-97 (field order = struct order)
    m->learning_rate = v.at("learning_rate").as_f32();
    m->minimum_learning_rate = opt_f32(v, "minimum_learning_rate", 0.0f);
    m->power_t = v.at("power_t").as_f32();
    m->bit_precision = (uint32_t)v.at("bit_precision").as_u64();
    m->add_constant_feature = v.at("add_constant_feature").as_bool();
    for (const auto &c : v.at("feature_combo_descs").arr) {
        std::vector<ND> nds;
        for (const auto &d : c.at("namespace_descriptors").arr) nds.push_back(nd_from(d));
        m->combos.emplace_back(nds, c.at("weight").as_f32());
    }
    for (const auto &f : v.at("ffm_fields").arr) {
        std::vector<ND> nds;
        for (const auto &d : f.arr) nds.push_back(nd_from(d));
        m->fields.push_back(nds);
    }
    m->ffm_k = v.get("ffm_k") ? (uint32_t)v.at("ffm_k").as_u64() : 0;
    m->ffm_bit_precision = v.get("ffm_bit_precision") ? (uint32_t)v.at("ffm_bit_precision").as_u64() : 0;
    m->fastmath = v.get("fastmath") ? v.at("fastmath").as_bool() : false;
    m->ffm_initialization_type = v.at("ffm_initialization_type").as_str();
    m->ffm_k_threshold = opt_f32(v, "ffm_k_threshold", 0);
    m->ffm_init_center = opt_f32(v, "ffm_init_center", 0);
    m->ffm_init_width = opt_f32(v, "ffm_init_width", 0);
    m->ffm_init_zero_band = opt_f32(v, "ffm_init_zero_band", 0);
    m->ffm_init_acc_gradient = opt_f32(v, "ffm_init_acc_gradient", 0);
    m->init_acc_gradient = opt_f32(v, "init_acc_gradient", 0);
    m->ffm_learning_rate = opt_f32(v, "ffm_learning_rate", 0);
    m->ffm_power_t = opt_f32(v, "ffm_power_t", 0);
    m->nn_init_acc_gradient = opt_f32(v, "nn_init_acc_gradient", 0);
    m->nn_learning_rate = opt_f32(v, "nn_learning_rate", 0);
    m->nn_power_t = opt_f32(v, "nn_power_t", 0);
    const fwjson::Value &nn = v.at("nn_config");
    for (const auto &l : nn.at("layers").arr) {
        std::map<std::string, std::string> layer;
        for (const auto &kv : l.obj) layer[kv.first] = kv.second.as_str();
        m->nn_layers.push_back(layer);
    }
    m->nn_topology = nn.at("topology").as_str();
    if (const fwjson::Value *o = v.get("optimizer")) {
        const std::string &s = o->as_str();
        if (s == "SGD") m->optimizer = FWGPU_OPT_SGD;
        else if (s == "AdagradFlex") m->optimizer = FWGPU_OPT_ADAGRAD_FLEX;
        else if (s == "AdagradLUT") m->optimizer = FWGPU_OPT_ADAGRAD_LUT;
        else throw std::runtime_error("unknown variant `" + s + "`, expected one of `SGD`, `AdagradFlex`, `AdagradLUT`");
    } else {
        m->optimizer = FWGPU_OPT_ADAGRAD_FLEX;  // default_optimizer_adagrad (model_instance.rs:108-110)
    }
    m->transform_namespaces = v.at("transform_namespaces");
    const fwjson::Value *dq = v.get("dequantize_weights");
    m->dequantize_weights = (!dq || dq->kind == fwjson::Value::Null) ? -1 : (dq->as_bool() ? 1 : 0);
}

std::string mi_json(const fwgpu_model_instance *m) {  // serde_json::to_vec_pretty(&ModelInstance), persistence.rs:21-26
    fwjson::Writer w;
    w.begin_obj();
    w.key("learning_rate"); w.f32(m->learning_rate);
    w.key("minimum_learning_rate"); w.f32(m->minimum_learning_rate);
    w.key("power_t"); w.f32(m->power_t);
    w.key("bit_precision"); w.u64(m->bit_precision);
    w.key("add_constant_feature"); w.boolean(m->add_constant_feature);
    w.key("feature_combo_descs");
    w.begin_arr();
    for (const auto &c : m->combos) {
        w.begin_obj();
        w.key("namespace_descriptors");
        w.begin_arr();
        for (const auto &d : c.first) nd_write(w, d);
        w.end_arr();
        w.key("weight"); w.f32(c.second);
        w.end_obj();
    }
    w.end_arr();
    w.key("ffm_fields");
    w.begin_arr();
    for (const auto &f : m->fields) {
        w.begin_arr();
        for (const auto &d : f) nd_write(w, d);
        w.end_arr();
    }
    w.end_arr();
    w.key("ffm_k"); w.u64(m->ffm_k);
    w.key("ffm_bit_precision"); w.u64(m->ffm_bit_precision);
    w.key("fastmath"); w.boolean(m->fastmath);
    w.key("ffm_initialization_type"); w.str(m->ffm_initialization_type);
    w.key("ffm_k_threshold"); w.f32(m->ffm_k_threshold);
    w.key("ffm_init_center"); w.f32(m->ffm_init_center);
    w.key("ffm_init_width"); w.f32(m->ffm_init_width);
    w.key("ffm_init_zero_band"); w.f32(m->ffm_init_zero_band);
    w.key("ffm_init_acc_gradient"); w.f32(m->ffm_init_acc_gradient);
    w.key("init_acc_gradient"); w.f32(m->init_acc_gradient);
    w.key("ffm_learning_rate"); w.f32(m->ffm_learning_rate);
    w.key("ffm_power_t"); w.f32(m->ffm_power_t);
    w.key("nn_init_acc_gradient"); w.f32(m->nn_init_acc_gradient);
    w.key("nn_learning_rate"); w.f32(m->nn_learning_rate);
    w.key("nn_power_t"); w.f32(m->nn_power_t);
    w.key("nn_config");
    w.begin_obj();
    w.key("layers");
    w.begin_arr();
    for (const auto &l : m->nn_layers) {  // HashMap<String, String>: the reference's key order is arbitrary; ours is sorted
        w.begin_obj();
        for (const auto &kv : l) {
            w.key(kv.first.c_str());
            w.str(kv.second);
        }
        w.end_obj();
    }
    w.end_arr();
    w.key("topology"); w.str(m->nn_topology);
    w.end_obj();
    w.key("optimizer");
    w.str(m->optimizer == FWGPU_OPT_SGD ? "SGD" : m->optimizer == FWGPU_OPT_ADAGRAD_FLEX ? "AdagradFlex" : "AdagradLUT");
    w.key("transform_namespaces");
    if (m->transform_namespaces.kind == fwjson::Value::Obj) {
        value_write(w, m->transform_namespaces);
    } else {
        w.begin_obj();
        w.key("v");
        w.begin_arr();
        w.end_arr();
        w.end_obj();
    }
    w.key("dequantize_weights");
    if (m->dequantize_weights < 0) w.null();
    else w.boolean(m->dequantize_weights != 0);
    w.end_obj();
    return w.out;
}

// table sizes of the blocks a ModelInstance describes (block_lr.rs:67, block_ffm.rs:86-94, block_neural.rs:86 + regressor.rs:185-320)
struct BlobShape {
    uint64_t lr_len = 0, ffm_len = 0;
    std::vector<uint64_t> nn_layer_len;
    uint64_t elems() const {
        uint64_t n = lr_len + ffm_len;
        for (auto x : nn_layer_len) n += x;
        return n;
    }
};
int nn_from_mi(const fwgpu_model_instance *m, fwgpu_nn_config *nn) {
    std::memset(nn, 0, sizeof *nn);
    if (m->nn_layers.empty()) return FWGPU_OK;
    if (m->nn_layers.size() > FWGPU_NN_MAX_LAYERS) return fail(FWGPU_ERR_INVALID, "nn: more than 8 hidden layers");
    if (m->nn_topology == "one") nn->topology = 1;
    else if (m->nn_topology == "two") nn->topology = 2;
    else if (m->nn_topology == "four" || m->nn_topology == "five")
        return fail(FWGPU_ERR_INVALID, "nn topology \"" + m->nn_topology + "\" needs block_normalize, which is out of scope here");
    else return fail(FWGPU_ERR_INVALID, "unknown nn topology: \"" + m->nn_topology + "\"");
    nn->n_layers = (uint32_t)m->nn_layers.size();
    for (size_t i = 0; i < m->nn_layers.size(); i++) {  // regressor.rs:217-275
        auto layer = m->nn_layers[i];
        auto take = [&](const char *k, const char *dflt) {
            auto it = layer.find(k);
            std::string v = it == layer.end() ? dflt : it->second;
            if (it != layer.end()) layer.erase(it);
            return v;
        };
        const std::string act = take("activation", "none"), ln = take("layernorm", "none"), width = take("width", "20"),
                          maxnorm = take("maxnorm", "0.0"), dropout = take("dropout", "0.0"), init = take("init", "hu");
        if (!layer.empty())
            return fail(FWGPU_ERR_INVALID, "Unknown --nn parameter for layer number " + std::to_string(i) + " : " + layer.begin()->first);
        if (act == "relu") nn->relu[i] = 1;
        else if (act != "none") return fail(FWGPU_ERR_INVALID, "unknown nn activation type: \"" + act + "\"");
        if (ln != "none") return fail(FWGPU_ERR_INVALID, "nn layernorm needs block_normalize, which is out of scope here");
        if (std::strtof(maxnorm.c_str(), nullptr) != 0.0f || std::strtof(dropout.c_str(), nullptr) != 0.0f)
            return fail(FWGPU_ERR_INVALID, "nn maxnorm / dropout are not supported on the device path");
        nn->width[i] = (uint32_t)std::strtoul(width.c_str(), nullptr, 10);
        if (!nn->width[i]) return fail(FWGPU_ERR_INVALID, "nn: bad layer width \"" + width + "\"");
        if (init == "xavier") nn->init[i] = FWGPU_NN_INIT_XAVIER;
        else if (init == "hu") nn->init[i] = FWGPU_NN_INIT_HU;
        else if (init == "one") nn->init[i] = FWGPU_NN_INIT_ONE;
        else if (init == "zero") nn->init[i] = FWGPU_NN_INIT_ZERO;
        else return fail(FWGPU_ERR_INVALID, "unknown nn initialization type: \"" + init + "\"");
    }
    nn->nn_learning_rate = m->nn_learning_rate;
    nn->nn_power_t = m->nn_power_t;
    nn->nn_init_acc_gradient = m->nn_init_acc_gradient;
    return FWGPU_OK;
}
int shape_from_mi(const fwgpu_model_instance *m, BlobShape *s) {
    if (m->bit_precision > 31 || m->ffm_bit_precision > 31) return fail(FWGPU_ERR_INVALID, "bit_precision out of range");
    s->lr_len = 1ull << m->bit_precision;
    const uint64_t F = m->ffm_k ? m->fields.size() : 0;
    s->ffm_len = m->ffm_k ? (1ull << m->ffm_bit_precision) + F * m->ffm_k : 0;
    fwgpu_nn_config nn;
    int rc = nn_from_mi(m, &nn);
    if (rc) return rc;
    if (nn.n_layers) {
        const uint64_t C = m->combos.size() + (m->add_constant_feature ? 1 : 0);
        const uint64_t X = C + F * (F + 1) / 2;
        uint64_t in = X;
        for (uint32_t l = 0; l < nn.n_layers; l++) {
            s->nn_layer_len.push_back((in + 1) * nn.width[l]);
            in = nn.width[l];
        }
        s->nn_layer_len.push_back(in + (nn.topology == 1 ? X : 0) + 1);
    }
    return FWGPU_OK;
}

// ---- f16 (IEEE binary16) conversions, round to nearest even like half::f16::from_f32
uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t exp = (x >> 23) & 0xff, man = x & 0x7fffffu;
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0));
    const int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        const uint32_t m = man | 0x800000u;
        const int shift = 14 - e;
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;  // may carry into the exponent, which is the right result
    return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, x;
    if (exp == 0) {
        if (man == 0) {
            x = sign;
        } else {
            int e = -1;
            do {
                man <<= 1;
                e++;
            } while (!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7f800000u | (man << 13);
    } else {
        x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

// quantization.rs:18-41 emit_weight_statistics + 42-80 quantize_ffm_weights: 8-byte header {increment, min} + f16 buckets
void quantize_ffm(const float *w, uint64_t n, std::vector<uint8_t> &out) {
    float mn = w[0], mx = w[0];
    for (uint64_t i = 0; i < n; i++) {
        mx = std::fmax(mx, w[i]);
        mn = std::fmin(mn, w[i]);
    }
    mn = std::round(mn * 10000.0f) / 10000.0f;  // MIN_PREC / MAX_PREC
    mx = std::round(mx * 10000.0f) / 10000.0f;
    const float inc = (mx - mn) / 65025.0f;     // NUM_BUCKETS
    out.resize(8 + 2 * n);
    std::memcpy(out.data(), &inc, 4);
    std::memcpy(out.data() + 4, &mn, 4);
    for (uint64_t i = 0; i < n; i++) {
        const uint16_t h = f32_to_f16(std::round((w[i] - mn) / inc));
        out[8 + 2 * i] = (uint8_t)h;
        out[9 + 2 * i] = (uint8_t)(h >> 8);
    }
}
void dequantize_ffm(const uint8_t *src, uint64_t n, float *w) {  // quantization.rs:82-98
    float inc, mn;
    std::memcpy(&inc, src, 4);
    std::memcpy(&mn, src + 4, 4);
    for (uint64_t i = 0; i < n; i++) {
        const uint16_t h = (uint16_t)(src[8 + 2 * i] | (src[9 + 2 * i] << 8));
        w[i] = mn + f16_to_f32(h) * inc;
    }
}

struct File {
    FILE *f = nullptr;
    ~File() {
        if (f) std::fclose(f);
    }
    void need(void *p, size_t n, const char *what) {
        if (n && std::fread(p, 1, n, f) != n) throw std::runtime_error(std::string("model file: truncated ") + what);
    }
    void put(const void *p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error("model file: write failed");
    }
    uint64_t u64() {
        uint8_t b[8];
        need(b, 8, "length");
        uint64_t v = 0;
        for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
        return v;
    }
    uint64_t remaining() {  // bytes between the read position and the end of the file
        const long here = std::ftell(f);
        std::fseek(f, 0, SEEK_END);
        const long end = std::ftell(f);
        std::fseek(f, here, SEEK_SET);
        return end > here ? (uint64_t)(end - here) : 0;
    }
    void put_u64(uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
        put(b, 8);
    }
};

// persistence.rs:189-204 verify_header + 91-125 load_regressor_without_weights
void read_header(File &in, std::unique_ptr<fwgpu_vwmap, void (*)(fwgpu_vwmap *)> &vw, std::unique_ptr<fwgpu_model_instance> &mi) {
    uint8_t h[8];
    in.need(h, 8, "header");
    if (std::memcmp(h, "FWRE", 4) != 0) throw std::runtime_error("Cache header does not begin with magic bytes FWFW");  // sic
    const uint32_t ver = (uint32_t)h[4] | (h[5] << 8) | (h[6] << 16) | ((uint32_t)h[7] << 24);
    if (ver != 6)
        throw std::runtime_error("Cache file version of this binary: 6, version of the cache file: " + std::to_string(ver));
    uint64_t len = in.u64();
    if (len > (64u << 20)) throw std::runtime_error("model file: implausible vw_source length");
    std::string js(len, '\0');
    in.need(&js[0], len, "vw_source");
    fwgpu_vwmap *v = nullptr;
    if (fwgpu_vwmap_from_json(js.data(), js.size(), &v) != FWGPU_OK) throw std::runtime_error(fwgpu_last_error());
    vw.reset(v);
    len = in.u64();
    if (len > (64u << 20)) throw std::runtime_error("model file: implausible ModelInstance length");
    js.assign(len, '\0');
    in.need(&js[0], len, "ModelInstance");
    mi = std::make_unique<fwgpu_model_instance>();
    mi_parse(fwjson::Parser(js.data(), js.size()).parse(), mi.get());
}
void write_header(File &out, const fwgpu_vwmap *vw, const fwgpu_model_instance *mi) {  // persistence.rs:73-97
    const uint8_t h[8] = {'F', 'W', 'R', 'E', 6, 0, 0, 0};
    out.put(h, 8);
    std::string js = vwmap_json(vw);
    out.put_u64(js.size());
    out.put(js.data(), js.size());
    js = mi_json(mi);
    out.put_u64(js.size());
    out.put(js.data(), js.size());
}

void fill_views(fwgpu_model_instance *m) {
    m->combo_off.assign(1, 0);
    m->combo_ns.clear();
    m->combo_f32.clear();
    m->combo_w.clear();
    for (const auto &c : m->combos) {
        for (const auto &d : c.first) {
            m->combo_ns.push_back(d.index);
            m->combo_f32.push_back(d.f32);
        }
        m->combo_off.push_back((uint32_t)m->combo_ns.size());
        m->combo_w.push_back(c.second);
    }
    m->field_off.assign(1, 0);
    m->field_ns.clear();
    m->field_f32.clear();
    for (const auto &f : m->fields) {
        for (const auto &d : f) {
            m->field_ns.push_back(d.index);
            m->field_f32.push_back(d.f32);
        }
        m->field_off.push_back((uint32_t)m->field_ns.size());
    }
}

// Per-block byte layout of a weights blob for optimizer `opt` (4 B per element for SGD, else weights + state)
// and the conversions between layouts.  `src` is the blob body after the u64 element count.
// training/any layout -> weights only (read_weights_from_buf_into_forward_only: block_lr.rs, block_ffm.rs:879-900, block_neural.rs)
void to_weights_only(const uint8_t *src, int src_opt, bool src_quantized, const BlobShape &s, std::vector<uint8_t> &dst) {
    const bool sgd = src_opt == FWGPU_OPT_SGD;
    dst.clear();
    const uint8_t *p = src;
    if (sgd) {
        dst.insert(dst.end(), p, p + s.lr_len * 4);
        p += s.lr_len * 4;
    } else {
        dst.resize(s.lr_len * 4);
        for (uint64_t i = 0; i < s.lr_len; i++) std::memcpy(&dst[4 * i], p + 8 * i, 4);  // {w, acc} pairs -> w
        p += s.lr_len * 8;
    }
    if (s.ffm_len) {
        const size_t at = dst.size();
        dst.resize(at + s.ffm_len * 4);
        if (src_quantized) {
            dequantize_ffm(p, s.ffm_len, reinterpret_cast<float *>(&dst[at]));
            p += 8 + 2 * s.ffm_len;
        } else {
            std::memcpy(&dst[at], p, s.ffm_len * 4);
            p += s.ffm_len * 4;
        }
        if (!sgd) p += s.ffm_len * 4;  // skip_weights_from_buf::<OptimizerData<L>>
    }
    for (uint64_t len : s.nn_layer_len) {
        dst.insert(dst.end(), p, p + len * 4);
        p += len * 4;
        if (!sgd) p += len * 4;
    }
}
uint64_t blob_body_bytes(int opt, bool quantized, const BlobShape &s) {
    const bool sgd = opt == FWGPU_OPT_SGD;
    uint64_t n = s.lr_len * (sgd ? 4 : 8);
    if (s.ffm_len) n += (quantized ? 8 + 2 * s.ffm_len : 4 * s.ffm_len) + (sgd ? 0 : 4 * s.ffm_len);
    for (uint64_t len : s.nn_layer_len) n += len * (sgd ? 4 : 8);
    return n;
}

}  // namespace

extern "C" {

int fwgpu_mi_from_json(const char *json, uint64_t len, fwgpu_model_instance **out) {
    if (!json || !out) return fail(FWGPU_ERR_INVALID, "NULL argument");
    try {
        auto m = std::make_unique<fwgpu_model_instance>();
        mi_parse(fwjson::Parser(json, len).parse(), m.get());
        fill_views(m.get());
        *out = m.release();
        return FWGPU_OK;
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_FORMAT, std::string("ModelInstance JSON: ") + e.what());
    }
}

void fwgpu_mi_free(fwgpu_model_instance *m) { delete m; }

int fwgpu_mi_to_json(const fwgpu_model_instance *m, char *buf, uint64_t cap, uint64_t *len) {
    if (!m || !len) return fail(FWGPU_ERR_INVALID, "NULL argument");
    const std::string s = mi_json(m);
    *len = s.size();
    if (!buf) return FWGPU_OK;
    if (cap < s.size()) return fail(FWGPU_ERR_RANGE, "buffer too small for the ModelInstance JSON");
    std::memcpy(buf, s.data(), s.size());
    return FWGPU_OK;
}

int fwgpu_mi_configs(fwgpu_model_instance *m, int device, fwgpu_config *cfg, fwgpu_translator_config *tr,
                     fwgpu_nn_config *nn) {
    if (!m) return fail(FWGPU_ERR_INVALID, "NULL argument");
    if (m->transform_namespaces.kind == fwjson::Value::Obj) {
        const fwjson::Value *v = m->transform_namespaces.get("v");
        if (v && !v->arr.empty())
            return fail(FWGPU_ERR_INVALID, "transformed namespaces (feature_transform_*.rs) are out of scope on the device path");
    }
    for (const auto &c : m->combos)
        for (const auto &d : c.first)
            if (d.transformed) return fail(FWGPU_ERR_INVALID, "transformed namespaces are out of scope on the device path");
    fill_views(m);
    if (cfg) {
        std::memset(cfg, 0, sizeof *cfg);
        cfg->optimizer = m->optimizer;
        cfg->learning_rate = m->learning_rate;
        cfg->power_t = m->power_t;
        cfg->init_acc_gradient = m->init_acc_gradient;
        cfg->bit_precision = m->bit_precision;
        cfg->num_combos = (uint32_t)m->combos.size() + (m->add_constant_feature ? 1 : 0);
        cfg->ffm_k = m->ffm_k;
        cfg->ffm_bit_precision = m->ffm_bit_precision;
        cfg->ffm_num_fields = (uint32_t)m->fields.size();
        cfg->ffm_learning_rate = m->ffm_learning_rate;
        cfg->ffm_power_t = m->ffm_power_t;
        cfg->ffm_init_acc_gradient = m->ffm_init_acc_gradient;
        cfg->ffm_init_center = m->ffm_init_center;
        cfg->ffm_init_width = m->ffm_init_width;
        cfg->ffm_init_zero_band = m->ffm_init_zero_band;
        cfg->wiring = FWGPU_WIRING_REGRESSOR;
        cfg->device = device;
    }
    if (tr) {
        std::memset(tr, 0, sizeof *tr);
        tr->n_combos = (uint32_t)m->combos.size();
        tr->combo_off = m->combo_off.data();
        tr->combo_ns = m->combo_ns.data();
        tr->combo_ns_f32 = m->combo_f32.data();
        tr->combo_weight = m->combo_w.data();
        tr->add_constant_feature = m->add_constant_feature;
        tr->n_fields = (uint32_t)m->fields.size();
        tr->field_off = m->field_off.data();
        tr->field_ns = m->field_ns.data();
        tr->field_ns_f32 = m->field_f32.data();
        tr->bit_precision = m->bit_precision;
        tr->ffm_k = m->ffm_k;
        tr->ffm_bit_precision = m->ffm_bit_precision;
    }
    if (nn) {
        int rc = nn_from_mi(m, nn);
        if (rc) return rc;
    }
    return FWGPU_OK;
}

int fwgpu_mi_set_inference(fwgpu_model_instance *m, int dequantize_weights) {  // main.rs:141-145
    if (!m) return fail(FWGPU_ERR_INVALID, "NULL argument");
    m->optimizer = FWGPU_OPT_SGD;
    if (dequantize_weights) m->dequantize_weights = 1;
    return FWGPU_OK;
}

// persistence.rs:73-89 save_regressor_to_filename
int fwgpu_model_save(const char *path, const fwgpu_vwmap *vw, const fwgpu_model_instance *mi, fwgpu_regressor *r,
                     int quantize_weights) {
    if (!path || !vw || !mi || !r) return fail(FWGPU_ERR_INVALID, "NULL argument");
    BlobShape s;
    int rc = shape_from_mi(mi, &s);
    if (rc) return rc;
    uint64_t need = 0;
    rc = fwgpu_serialized_len(r, &need);
    if (rc) return rc;
    if (mi->optimizer != r->cfg.optimizer || need != 8 + blob_body_bytes(r->cfg.optimizer, false, s))
        return fail(FWGPU_ERR_INVALID, "model_save: the regressor was not built from this ModelInstance");
    // The header must announce a quantised FFM blob (the reference only quantises on the convert path, after setting
    // dequantize_weights = Some(true): main.rs:141-145); otherwise the file could not be read back.
    if (quantize_weights && s.ffm_len && mi->dequantize_weights != 1)
        return fail(FWGPU_ERR_INVALID, "model_save: quantize_weights needs a ModelInstance with dequantize_weights = true");
    try {
        std::vector<uint8_t> blob(need);
        uint64_t written = 0;
        rc = fwgpu_write_weights(r, blob.data(), blob.size(), &written);
        if (rc) return rc;
        File out;
        out.f = std::fopen(path, "wb");
        if (!out.f) return fail(FWGPU_ERR_IO, std::string("Cannot open ") + path + " to save regressor to");
        write_header(out, vw, mi);
        if (!quantize_weights || !s.ffm_len) {
            out.put(blob.data(), written);
        } else {  // block_ffm.rs:835-848: the FFM weights become f16 buckets, everything else is unchanged
            const bool sgd = mi->optimizer == FWGPU_OPT_SGD;
            const uint64_t lr_bytes = s.lr_len * (sgd ? 4 : 8);
            out.put(blob.data(), 8 + lr_bytes);
            std::vector<uint8_t> q;
            quantize_ffm(reinterpret_cast<const float *>(blob.data() + 8 + lr_bytes), s.ffm_len, q);
            out.put(q.data(), q.size());
            const uint64_t rest = 8 + lr_bytes + s.ffm_len * 4;
            out.put(blob.data() + rest, written - rest);
        }
        if (std::fclose(out.f) != 0) {
            out.f = nullptr;
            return fail(FWGPU_ERR_IO, "model_save: close failed");
        }
        out.f = nullptr;
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_IO, e.what());
    }
    return FWGPU_OK;
}

// persistence.rs:91-125: header + vw_source + ModelInstance, no weights, no device
int fwgpu_model_read_header(const char *path, fwgpu_vwmap **vw_out, fwgpu_model_instance **mi_out) {
    if (!path) return fail(FWGPU_ERR_INVALID, "NULL argument");
    try {
        File in;
        in.f = std::fopen(path, "rb");
        if (!in.f) return fail(FWGPU_ERR_IO, std::string("cannot open ") + path);
        std::unique_ptr<fwgpu_vwmap, void (*)(fwgpu_vwmap *)> vw(nullptr, fwgpu_vwmap_free);
        std::unique_ptr<fwgpu_model_instance> mi;
        read_header(in, vw, mi);
        fill_views(mi.get());
        if (vw_out) *vw_out = vw.release();
        if (mi_out) *mi_out = mi.release();
        return FWGPU_OK;
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_FORMAT, e.what());
    }
}

// persistence.rs:127-174 new_regressor_from_filename.  immutable != 0: the optimizer becomes SGD and only the weights are
// kept ("forward only"); the file may be a training file (weights + optimizer state) or an already converted one.
// With *r_inout != NULL the weights are loaded into that regressor instead (hogwild_load, persistence.rs:176-187).
int fwgpu_model_load(const char *path, int device, int immutable, fwgpu_vwmap **vw_out, fwgpu_model_instance **mi_out,
                     fwgpu_regressor **r_inout) {
    if (!path || !r_inout) return fail(FWGPU_ERR_INVALID, "NULL argument");
    try {
        File in;
        in.f = std::fopen(path, "rb");
        if (!in.f) return fail(FWGPU_ERR_IO, std::string("cannot open ") + path);
        std::unique_ptr<fwgpu_vwmap, void (*)(fwgpu_vwmap *)> vw(nullptr, fwgpu_vwmap_free);
        std::unique_ptr<fwgpu_model_instance> mi;
        read_header(in, vw, mi);
        BlobShape s;
        int rc = shape_from_mi(mi.get(), &s);
        if (rc) return rc;
        const uint64_t count = in.u64();
        if (count != s.elems())
            return fail(FWGPU_ERR_FORMAT, "Lenghts of weights array in regressor file differ: got " + std::to_string(count) +
                                              ", expected " + std::to_string(s.elems()));  // sic, regressor.rs:458-462
        const int file_opt = mi->optimizer;
        const bool quantized = mi->dequantize_weights == 1;  // persistence.rs:147-153
        const uint64_t body = blob_body_bytes(file_opt, quantized, s);
        if (in.remaining() < body) return fail(FWGPU_ERR_FORMAT, "model file: truncated weights");  // before allocating `body` bytes
        std::vector<uint8_t> src(body);
        in.need(src.data(), body, "weights");

        fwgpu_regressor *r = *r_inout;
        const bool own = r == nullptr;
        struct Guard {  // frees a regressor created here unless the load completes
            fwgpu_regressor **r;
            bool armed;
            ~Guard() {
                if (armed && *r) fwgpu_free(*r);
            }
        } guard{&r, own};
        if (own) {
            if (immutable) mi->optimizer = FWGPU_OPT_SGD;  // persistence.rs:164
            fwgpu_config cfg;
            fwgpu_nn_config nn;
            rc = fwgpu_mi_configs(mi.get(), device, &cfg, nullptr, &nn);
            if (rc) return rc;
            rc = fwgpu_create(&cfg, &r);
            if (rc) return rc;
            if (nn.n_layers) rc = fwgpu_set_nn(r, &nn);
            if (!rc) rc = fwgpu_init_weights(r);  // allocate_and_init_weights, then overwritten (persistence.rs:160-161)
            if (rc) return rc;
        }
        // bring the file's layout to the regressor's
        std::vector<uint8_t> blob;
        const int dst_opt = r->cfg.optimizer;
        if (dst_opt == FWGPU_OPT_SGD) {
            std::vector<uint8_t> w;
            to_weights_only(src.data(), file_opt, quantized, s, w);
            blob.resize(8 + w.size());
            std::memcpy(blob.data() + 8, w.data(), w.size());
        } else {
            if (file_opt == FWGPU_OPT_SGD) {
                return fail(FWGPU_ERR_INVALID, "an inference (SGD) model file carries no optimizer state to resume training from");
            }
            blob.resize(8 + blob_body_bytes(file_opt, false, s));
            if (quantized) {  // dequantize in place of the FFM weights
                const uint64_t lr_bytes = s.lr_len * 8;
                std::memcpy(blob.data() + 8, src.data(), lr_bytes);
                dequantize_ffm(src.data() + lr_bytes, s.ffm_len, reinterpret_cast<float *>(blob.data() + 8 + lr_bytes));
                const uint64_t src_rest = lr_bytes + 8 + 2 * s.ffm_len, dst_rest = 8 + lr_bytes + 4 * s.ffm_len;
                std::memcpy(blob.data() + dst_rest, src.data() + src_rest, body - src_rest);
            } else {
                std::memcpy(blob.data() + 8, src.data(), body);
            }
        }
        std::memcpy(blob.data(), &count, 8);
        rc = fwgpu_read_weights(r, blob.data(), blob.size());
        if (rc) return rc;
        fill_views(mi.get());
        guard.armed = false;
        *r_inout = r;
        if (vw_out) *vw_out = vw.release();
        if (mi_out) *mi_out = mi.release();
        return FWGPU_OK;
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_FORMAT, e.what());
    }
}

// main.rs:136-148 --convert_inference_regressor: training file -> inference file (optimizer SGD, weights only, optional
// f16 buckets).  Pure host code: no device is touched.
int fwgpu_model_convert_inference(const char *in_path, const char *out_path, int quantize_weights) {
    if (!in_path || !out_path) return fail(FWGPU_ERR_INVALID, "NULL argument");
    try {
        File in;
        in.f = std::fopen(in_path, "rb");
        if (!in.f) return fail(FWGPU_ERR_IO, std::string("cannot open ") + in_path);
        std::unique_ptr<fwgpu_vwmap, void (*)(fwgpu_vwmap *)> vw(nullptr, fwgpu_vwmap_free);
        std::unique_ptr<fwgpu_model_instance> mi;
        read_header(in, vw, mi);
        BlobShape s;
        int rc = shape_from_mi(mi.get(), &s);
        if (rc) return rc;
        const uint64_t count = in.u64();
        if (count != s.elems())
            return fail(FWGPU_ERR_FORMAT, "Lenghts of weights array in regressor file differ: got " + std::to_string(count) +
                                              ", expected " + std::to_string(s.elems()));
        // the reference reads with weight_quantization = dequantize_weights && !conversion_flag == false (persistence.rs:147-153)
        const uint64_t body = blob_body_bytes(mi->optimizer, false, s);
        if (in.remaining() < body) return fail(FWGPU_ERR_FORMAT, "model file: truncated weights");
        std::vector<uint8_t> src(body), w;
        in.need(src.data(), body, "weights");
        to_weights_only(src.data(), mi->optimizer, false, s, w);
        fwgpu_mi_set_inference(mi.get(), quantize_weights);
        File out;
        out.f = std::fopen(out_path, "wb");
        if (!out.f) return fail(FWGPU_ERR_IO, std::string("Cannot open ") + out_path + " to save regressor to");
        write_header(out, vw.get(), mi.get());
        out.put_u64(count);
        if (quantize_weights && s.ffm_len) {
            out.put(w.data(), s.lr_len * 4);
            std::vector<uint8_t> q;
            quantize_ffm(reinterpret_cast<const float *>(w.data() + s.lr_len * 4), s.ffm_len, q);
            out.put(q.data(), q.size());
            const uint64_t rest = (s.lr_len + s.ffm_len) * 4;
            out.put(w.data() + rest, w.size() - rest);
        } else {
            out.put(w.data(), w.size());
        }
        if (std::fclose(out.f) != 0) {
            out.f = nullptr;
            return fail(FWGPU_ERR_IO, "convert: close failed");
        }
        out.f = nullptr;
        return FWGPU_OK;
    } catch (const std::exception &e) {
        return fail(FWGPU_ERR_FORMAT, e.what());
    }
}

// quantization.rs:42-98 on caller buffers (the reference's own tests exercise exactly these two functions)
int fwgpu_quantize_ffm_weights(const float *weights, uint64_t n, uint8_t *out, uint64_t cap) {
    if (!weights || !out || n == 0) return fail(FWGPU_ERR_INVALID, "NULL / empty argument");
    if (cap < 8 + 2 * n) return fail(FWGPU_ERR_RANGE, "buffer too small");
    std::vector<uint8_t> q;
    quantize_ffm(weights, n, q);
    std::memcpy(out, q.data(), q.size());
    return FWGPU_OK;
}
int fwgpu_dequantize_ffm_weights(const uint8_t *in, uint64_t n, float *weights) {
    if (!in || !weights) return fail(FWGPU_ERR_INVALID, "NULL argument");
    dequantize_ffm(in, n, weights);
    return FWGPU_OK;
}

}  // extern "C"
