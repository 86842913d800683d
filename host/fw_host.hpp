// fw_host.hpp -- C++ host-side mirror of the reference interface for the LR+FFM path, over the C ABI (include/fwgpu.h).
//
// The reference is Rust; Rust is not available in this image, so the layer a maintainer would write in Rust
//  is provided in C++ with the reference's names, argument meaning and error behaviour:
//
//   fw::ModelInstance            model_instance.rs:47-97, defaults = ModelInstance::new_empty() (120-150)
//   fw::HashAndValue[AndSeq]     feature_buffer.rs:10-22
//   fw::FeatureBuffer            feature_buffer.rs:24-31
//   fw::FeatureBufferTranslator  feature_buffer.rs:33-44, 138-338
//   fw::Regressor                regressor.rs:142-147; learn/predict 356-395; weights 426-469
//   fw::HogwildTrainer           hogwild.rs:13-61
//   fw::VwNamespaceMap           vwmap.rs:30-151, persistence.rs:36-53
//   fw::VowpalParser             parser.rs:24-461 (FlushCommand / HogwildLoadCommand as exceptions)
//   fw::RecordCache              cache.rs:54-232
//   fw::persistence::*           persistence.rs:55-187, main.rs:136-148 (convert_inference_regressor)
//   fw::Predictor                lib.rs:55-148 over the reference's own FFI symbols (include/fw_ffi.h)
//
// Errors: configuration errors throw std::runtime_error (the reference returns Err -> exit 1, main.rs:44-47);
// violated internal invariants in learn/predict also throw (the reference panics).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "fw_ffi.h"
#include "fwgpu.h"

namespace fw {

enum class Optimizer : int32_t { SGD = FWGPU_OPT_SGD, AdagradFlex = FWGPU_OPT_ADAGRAD_FLEX, AdagradLUT = FWGPU_OPT_ADAGRAD_LUT };

using HashAndValue = fwgpu_lr_entry;         // {hash, value, combo_index}
using HashAndValueAndSeq = fwgpu_ffm_entry;  // {hash, value, contra_field_index}

struct NamespaceDescriptor {  // vwmap.rs:23-27 (primitive namespaces)
    uint16_t namespace_index = 0;
    bool format_f32 = false;
};

struct FeatureComboDesc {
    std::vector<NamespaceDescriptor> namespace_descriptors;
    float weight = 1.0f;
};

struct ModelInstance {  // ModelInstance::new_empty(), model_instance.rs:120-150
    float learning_rate = 0.5f;
    float ffm_learning_rate = 0.5f;
    uint8_t bit_precision = 18;
    float power_t = 0.5f;
    float ffm_power_t = 0.5f;
    bool add_constant_feature = true;
    std::vector<FeatureComboDesc> feature_combo_descs;
    std::vector<std::vector<NamespaceDescriptor>> ffm_fields;
    uint32_t ffm_k = 0;
    uint32_t ffm_bit_precision = 18;
    float ffm_init_center = 0.0f, ffm_init_width = 0.0f, ffm_init_zero_band = 0.0f;
    float ffm_init_acc_gradient = 0.0f;
    float init_acc_gradient = 1.0f;
    Optimizer optimizer = Optimizer::SGD;
    // not in the reference: the wiring of the FFM block tests (block_ffm.rs:1253-1254) and the HIP device
    int32_t wiring = FWGPU_WIRING_REGRESSOR;
    int32_t device = 0;

    static ModelInstance new_empty() { return ModelInstance(); }

    fwgpu_config to_config() const {
        fwgpu_config c{};
        c.optimizer = static_cast<int32_t>(optimizer);
        c.learning_rate = learning_rate;
        c.power_t = power_t;
        c.init_acc_gradient = init_acc_gradient;
        c.bit_precision = bit_precision;
        c.num_combos = static_cast<uint32_t>(feature_combo_descs.size()) + (add_constant_feature ? 1u : 0u);  // block_lr.rs:53-56
        c.ffm_k = ffm_k;
        c.ffm_bit_precision = ffm_bit_precision;
        c.ffm_num_fields = static_cast<uint32_t>(ffm_fields.size());
        c.ffm_learning_rate = ffm_learning_rate;
        c.ffm_power_t = ffm_power_t;
        c.ffm_init_acc_gradient = ffm_init_acc_gradient;
        c.ffm_init_center = ffm_init_center;
        c.ffm_init_width = ffm_init_width;
        c.ffm_init_zero_band = ffm_init_zero_band;
        c.wiring = wiring;
        c.device = device;
        return c;
    }
};

struct FeatureBuffer {  // feature_buffer.rs:24-31
    float label = 0.0f;
    float example_importance = 1.0f;
    uint64_t example_number = 0;
    std::vector<HashAndValue> lr_buffer;
    std::vector<HashAndValueAndSeq> ffm_buffer;
};

inline void check(int rc) {
    if (rc != FWGPU_OK) throw std::runtime_error(std::string("fwgpu error ") + std::to_string(rc) + ": " + fwgpu_last_error());
}

// Owns the flattened arrays a fwgpu_translator_config points into.
class FeatureBufferTranslator {
  public:
    explicit FeatureBufferTranslator(const ModelInstance &mi) {
        combo_off_.push_back(0);
        for (const auto &cd : mi.feature_combo_descs) {
            for (const auto &nd : cd.namespace_descriptors) {
                combo_ns_.push_back(nd.namespace_index);
                combo_f32_.push_back(nd.format_f32 ? 1 : 0);
            }
            combo_off_.push_back(static_cast<uint32_t>(combo_ns_.size()));
            combo_w_.push_back(cd.weight);
        }
        field_off_.push_back(0);
        for (const auto &fld : mi.ffm_fields) {
            for (const auto &nd : fld) {
                field_ns_.push_back(nd.namespace_index);
                field_f32_.push_back(nd.format_f32 ? 1 : 0);
            }
            field_off_.push_back(static_cast<uint32_t>(field_ns_.size()));
        }
        // keep data() valid for empty vectors
        if (combo_ns_.empty()) { combo_ns_.reserve(1); combo_f32_.reserve(1); }
        if (combo_w_.empty()) combo_w_.reserve(1);
        if (field_ns_.empty()) { field_ns_.reserve(1); field_f32_.reserve(1); }
        c_.n_combos = static_cast<uint32_t>(mi.feature_combo_descs.size());
        c_.combo_off = combo_off_.data();
        c_.combo_ns = combo_ns_.data();
        c_.combo_ns_f32 = combo_f32_.data();
        c_.combo_weight = combo_w_.data();
        c_.add_constant_feature = mi.add_constant_feature ? 1 : 0;
        c_.n_fields = static_cast<uint32_t>(mi.ffm_fields.size());
        c_.field_off = field_off_.data();
        c_.field_ns = field_ns_.data();
        c_.field_ns_f32 = field_f32_.data();
        c_.bit_precision = mi.bit_precision;
        c_.ffm_k = mi.ffm_k;
        c_.ffm_bit_precision = mi.ffm_bit_precision;
        lr_hash_mask = fwgpu_lr_hash_mask(mi.bit_precision);
        ffm_hash_mask = fwgpu_ffm_hash_mask(mi.ffm_bit_precision, mi.ffm_k);
    }
    FeatureBufferTranslator(const FeatureBufferTranslator &) = delete;
    FeatureBufferTranslator &operator=(const FeatureBufferTranslator &) = delete;

    // feature_buffer.rs:174-176
    void translate(const std::vector<uint32_t> &record_buffer, uint64_t example_number) {
        feature_buffer.lr_buffer.resize(8192);
        feature_buffer.ffm_buffer.resize(8192);
        uint32_t n_lr = 0, n_ffm = 0;
        check(fwgpu_translate(&c_, record_buffer.data(), static_cast<uint32_t>(record_buffer.size()),
                              feature_buffer.lr_buffer.data(), 8192, &n_lr, feature_buffer.ffm_buffer.data(), 8192, &n_ffm,
                              &feature_buffer.label, &feature_buffer.example_importance));
        feature_buffer.lr_buffer.resize(n_lr);
        feature_buffer.ffm_buffer.resize(n_ffm);
        feature_buffer.example_number = example_number;
    }
    const fwgpu_translator_config *config() const { return &c_; }

    FeatureBuffer feature_buffer;
    uint32_t lr_hash_mask = 0, ffm_hash_mask = 0;

  private:
    std::vector<uint32_t> combo_off_, combo_ns_, field_off_, field_ns_;
    std::vector<uint8_t> combo_f32_, field_f32_;
    std::vector<float> combo_w_;
    fwgpu_translator_config c_{};
};

struct PortBuffer {};  // port_buffer.rs: the tape lives in LDS on the device; kept for call-site compatibility

class Regressor {
  public:
    // Regressor::new(&mi) = new_without_weights + allocate_and_init_weights (regressor.rs:338-342)
    explicit Regressor(const ModelInstance &mi) : mi_(mi) {
        fwgpu_config c = mi.to_config();
        check(fwgpu_create(&c, &h_));
        allocate_and_init_weights(mi);
    }
    ~Regressor() { fwgpu_free(h_); }
    Regressor(const Regressor &) = delete;
    Regressor &operator=(const Regressor &) = delete;

    void allocate_and_init_weights(const ModelInstance &) { check(fwgpu_init_weights(h_)); }  // regressor.rs:352-354
    PortBuffer new_portbuffer() const { return PortBuffer(); }                               // regressor.rs:348-350
    std::string get_name() const {                                                          // regressor.rs:177
        const char *n = mi_.optimizer == Optimizer::SGD ? "SGD" : mi_.optimizer == Optimizer::AdagradFlex ? "AdagradFlex" : "AdagradLUT";
        return std::string("Regressor with optimizer \"") + n + "\"";
    }

    // regressor.rs:356-379
    float learn(const FeatureBuffer &fb, PortBuffer &, bool update) {
        float p = 0.0f;
        check(fwgpu_learn(h_, fb.lr_buffer.data(), static_cast<uint32_t>(fb.lr_buffer.size()), fb.ffm_buffer.data(),
                          static_cast<uint32_t>(fb.ffm_buffer.size()), fb.label, fb.example_importance, update ? 1 : 0, &p));
        return p;
    }
    // regressor.rs:381-395
    float predict(const FeatureBuffer &fb, PortBuffer &) {
        float p = 0.0f;
        check(fwgpu_predict(h_, fb.lr_buffer.data(), static_cast<uint32_t>(fb.lr_buffer.size()), fb.ffm_buffer.data(),
                            static_cast<uint32_t>(fb.ffm_buffer.size()), &p));
        return p;
    }
    // regressor.rs:426-442 / 444-469
    std::vector<uint8_t> write_weights_to_buf() {
        uint64_t n = 0, w = 0;
        check(fwgpu_serialized_len(h_, &n));
        std::vector<uint8_t> buf(n);
        check(fwgpu_write_weights(h_, buf.data(), n, &w));
        buf.resize(w);
        return buf;
    }
    void overwrite_weights_from_buf(const std::vector<uint8_t> &buf) { check(fwgpu_read_weights(h_, buf.data(), buf.size())); }

    // the tests' ffm_init / ffm_fixed_init (block_ffm.rs:1228-1235, persistence.rs:315-330)
    void ffm_fill(float w) {
        check(fwgpu_table_fill(h_, FWGPU_TABLE_FFM_W, w));
        check(fwgpu_table_fill(h_, FWGPU_TABLE_FFM_ACC, mi_.optimizer == Optimizer::AdagradFlex ? mi_.ffm_init_acc_gradient : 0.0f));
    }
    fwgpu_regressor *handle() { return h_; }

  private:
    ModelInstance mi_;
    fwgpu_regressor *h_ = nullptr;
};

class HogwildTrainer {  // hogwild.rs:13-61; num_workers has no meaning on the device, micro_batch replaces it
  public:
    HogwildTrainer(Regressor &re, const ModelInstance &mi, uint32_t /*num_workers*/, uint32_t micro_batch = 4096) : fbt_(mi) {
        check(fwgpu_trainer_create(re.handle(), fbt_.config(), micro_batch, &h_));
    }
    ~HogwildTrainer() { fwgpu_trainer_free(h_); }
    HogwildTrainer(const HogwildTrainer &) = delete;
    HogwildTrainer &operator=(const HogwildTrainer &) = delete;

    void digest_example(const std::vector<uint32_t> &feature_buffer) {  // hogwild.rs:51-53
        const uint64_t off[2] = {0, feature_buffer.size()};
        check(fwgpu_digest_records(h_, feature_buffer.data(), off, 1));
    }
    void block_until_workers_finished() { check(fwgpu_finish(h_)); }  // hogwild.rs:55-60

  private:
    FeatureBufferTranslator fbt_;
    fwgpu_trainer *h_ = nullptr;
};

// ---------------------------------------------------------------- feed path (vwmap.rs, parser.rs, cache.rs)
class VwNamespaceMap {
  public:
    explicit VwNamespaceMap(const std::string &csv) { check(fwgpu_vwmap_from_csv(csv.data(), csv.size(), &h_)); }  // vwmap.rs:106
    explicit VwNamespaceMap(fwgpu_vwmap *adopt) : h_(adopt) {}
    ~VwNamespaceMap() { fwgpu_vwmap_free(h_); }
    VwNamespaceMap(const VwNamespaceMap &) = delete;
    VwNamespaceMap &operator=(const VwNamespaceMap &) = delete;
    size_t num_namespaces() const { return fwgpu_vwmap_num_namespaces(h_); }
    NamespaceDescriptor descriptor(const std::string &vwname) const {  // map_vwname_to_namespace_descriptor
        uint32_t idx = 0, f32 = 0;
        check(fwgpu_vwmap_lookup(h_, vwname.data(), vwname.size(), 0, &idx, &f32));
        NamespaceDescriptor nd;
        nd.namespace_index = static_cast<uint16_t>(idx);
        nd.format_f32 = f32 != 0;
        return nd;
    }
    std::string save_to_buf() const {  // the JSON of persistence.rs:37-42, without the u64 length prefix
        uint64_t n = 0;
        check(fwgpu_vwmap_to_json(h_, nullptr, 0, &n));
        std::string s(n, '\0');
        check(fwgpu_vwmap_to_json(h_, &s[0], n, &n));
        return s;
    }
    fwgpu_vwmap *handle() const { return h_; }

  private:
    fwgpu_vwmap *h_ = nullptr;
};

struct FlushCommand : std::runtime_error {  // parser.rs:31
    FlushCommand() : std::runtime_error("Not really an error: a \"flush\" command from client") {}
};
struct HogwildLoadCommand : std::runtime_error {  // parser.rs:33-37
    std::string filename;
    explicit HogwildLoadCommand(const std::string &f)
        : std::runtime_error("Not really an error: a \"hogwild_load\" command from client to load: " + f), filename(f) {}
};

class VowpalParser {
  public:
    explicit VowpalParser(const VwNamespaceMap &vw) { check(fwgpu_parser_create(vw.handle(), &h_)); }  // parser.rs:78-105
    ~VowpalParser() { fwgpu_parser_free(h_); }
    VowpalParser(const VowpalParser &) = delete;
    VowpalParser &operator=(const VowpalParser &) = delete;
    // parser.rs:166-176 on one line (newline included when present); "" = end of stream -> empty record.
    // Errors throw std::runtime_error carrying the reference's message.
    const std::vector<uint32_t> &next_vowpal(const std::string &line) { return run(nullptr, 0, line); }
    // parser.rs:195-211
    const std::vector<uint32_t> &next_vowpal_with_cache(const std::string &cached, const std::string &line) {
        return run(cached.data(), cached.size(), line);
    }
    std::vector<uint32_t> output_buffer;

  private:
    const std::vector<uint32_t> &run(const char *prefix, size_t plen, const std::string &line) {
        output_buffer.resize(1 << 16);
        uint32_t n = 0;
        const int rc = fwgpu_parser_parse_with_prefix(h_, prefix, plen, line.data(), line.size(), output_buffer.data(),
                                                      static_cast<uint32_t>(output_buffer.size()), &n);
        if (rc == FWGPU_PARSE_FLUSH) throw FlushCommand();
        if (rc == FWGPU_PARSE_HOGWILD_LOAD) throw HogwildLoadCommand(fwgpu_parser_command_argument(h_));
        if (rc != FWGPU_OK) throw std::runtime_error(fwgpu_last_error());
        output_buffer.resize(n);
        return output_buffer;
    }
    fwgpu_parser *h_ = nullptr;
};

class RecordCache {
  public:
    RecordCache(const std::string &input_filename, bool enabled, const VwNamespaceMap &vw) {  // cache.rs:70-131
        if (!enabled) return;
        check(fwgpu_cache_open(input_filename.c_str(), vw.handle(), &h_));
        reading = fwgpu_cache_is_reading(h_) != 0;
        writing = fwgpu_cache_is_writing(h_) != 0;
    }
    ~RecordCache() { fwgpu_cache_free(h_); }
    RecordCache(const RecordCache &) = delete;
    RecordCache &operator=(const RecordCache &) = delete;
    void push_record(const std::vector<uint32_t> &record_buf) {  // cache.rs:133-144
        if (h_) check(fwgpu_cache_push_records(h_, record_buf.data(), record_buf.size()));
    }
    void write_finish() {  // cache.rs:146-152
        if (h_) check(fwgpu_cache_write_finish(h_));
        writing = false;
    }
    std::vector<uint32_t> get_next_record() {  // cache.rs:187-232: empty at end of file
        if (!reading) throw std::runtime_error("next_recrod() called on reading cache, when not opened in reading mode");
        std::vector<uint32_t> rec(1 << 16);
        uint64_t off[2] = {0, 0}, nr = 0, nw = 0;
        check(fwgpu_cache_next_records(h_, rec.data(), rec.size(), off, 1, &nr, &nw));
        rec.resize(nr ? nw : 0);
        return rec;
    }
    fwgpu_cache *handle() const { return h_; }
    bool reading = false, writing = false;

  private:
    fwgpu_cache *h_ = nullptr;
};

// ---------------------------------------------------------------- persistence.rs
namespace persistence {

inline std::string model_instance_json(const ModelInstance &mi) {  // what serde would emit for the fields this mirror carries
    auto f = [](float v) {
        char b[64];
        check(fwgpu_debug_format_f32(v, b, sizeof b));
        return std::string(b);
    };
    auto nd = [](const NamespaceDescriptor &d) {
        return std::string("{\"namespace_index\":") + std::to_string(d.namespace_index) +
               ",\"namespace_type\":\"Primitive\",\"namespace_format\":\"" + (d.format_f32 ? "F32" : "Categorical") + "\"}";
    };
    std::string s = "{\"learning_rate\":" + f(mi.learning_rate) + ",\"minimum_learning_rate\":0.0,\"power_t\":" + f(mi.power_t) +
                    ",\"bit_precision\":" + std::to_string(mi.bit_precision) +
                    ",\"add_constant_feature\":" + (mi.add_constant_feature ? "true" : "false") + ",\"feature_combo_descs\":[";
    for (size_t i = 0; i < mi.feature_combo_descs.size(); i++) {
        s += std::string(i ? "," : "") + "{\"namespace_descriptors\":[";
        for (size_t j = 0; j < mi.feature_combo_descs[i].namespace_descriptors.size(); j++)
            s += std::string(j ? "," : "") + nd(mi.feature_combo_descs[i].namespace_descriptors[j]);
        s += "],\"weight\":" + f(mi.feature_combo_descs[i].weight) + "}";
    }
    s += "],\"ffm_fields\":[";
    for (size_t i = 0; i < mi.ffm_fields.size(); i++) {
        s += std::string(i ? "," : "") + "[";
        for (size_t j = 0; j < mi.ffm_fields[i].size(); j++) s += std::string(j ? "," : "") + nd(mi.ffm_fields[i][j]);
        s += "]";
    }
    const char *opt = mi.optimizer == Optimizer::SGD ? "SGD" : mi.optimizer == Optimizer::AdagradFlex ? "AdagradFlex" : "AdagradLUT";
    s += "],\"ffm_k\":" + std::to_string(mi.ffm_k) + ",\"ffm_bit_precision\":" + std::to_string(mi.ffm_bit_precision) +
         ",\"fastmath\":true,\"ffm_initialization_type\":\"default\",\"ffm_k_threshold\":0.0,\"ffm_init_center\":" +
         f(mi.ffm_init_center) + ",\"ffm_init_width\":" + f(mi.ffm_init_width) + ",\"ffm_init_zero_band\":" + f(mi.ffm_init_zero_band) +
         ",\"ffm_init_acc_gradient\":" + f(mi.ffm_init_acc_gradient) + ",\"init_acc_gradient\":" + f(mi.init_acc_gradient) +
         ",\"ffm_learning_rate\":" + f(mi.ffm_learning_rate) + ",\"ffm_power_t\":" + f(mi.ffm_power_t) +
         ",\"nn_init_acc_gradient\":0.0,\"nn_learning_rate\":0.02,\"nn_power_t\":0.45,\"nn_config\":{\"layers\":[],\"topology\":\"one\"}," +
         "\"optimizer\":\"" + opt + "\",\"transform_namespaces\":{\"v\":[]},\"dequantize_weights\":false}";
    return s;
}

// persistence.rs:73-89
inline void save_regressor_to_filename(const std::string &filename, const ModelInstance &mi, const VwNamespaceMap &vw, Regressor &re,
                                       bool quantize_weights = false) {
    const std::string js = model_instance_json(mi);
    fwgpu_model_instance *h = nullptr;
    check(fwgpu_mi_from_json(js.data(), js.size(), &h));
    const int rc = fwgpu_model_save(filename.c_str(), vw.handle(), h, re.handle(), quantize_weights ? 1 : 0);
    fwgpu_mi_free(h);
    check(rc);
}
// persistence.rs:176-187
inline void hogwild_load(Regressor &re, const std::string &filename, int device = 0) {
    fwgpu_regressor *r = re.handle();
    check(fwgpu_model_load(filename.c_str(), device, 0, nullptr, nullptr, &r));
}
// main.rs:136-148
inline void convert_inference_regressor(const std::string &in, const std::string &out, bool quantize_weights = false) {
    check(fwgpu_model_convert_inference(in.c_str(), out.c_str(), quantize_weights ? 1 : 0));
}

}  // namespace persistence

// lib.rs:55-148: the serving predictor, through the reference's own exported symbols
class Predictor {
  public:
    explicit Predictor(const std::string &command) : p_(new_fw_predictor_prototype(command.c_str())) {
        if (!p_) throw std::runtime_error(fwgpu_last_error());
    }
    ~Predictor() { free_predictor(p_); }
    Predictor(const Predictor &) = delete;
    Predictor &operator=(const Predictor &) = delete;
    float predict(const std::string &input) { return fw_predict(p_, input.c_str()); }
    float setup_cache(const std::string &input) { return fw_setup_cache(p_, input.c_str()); }
    float predict_with_cache(const std::string &input) { return fw_predict_with_cache(p_, input.c_str()); }

  private:
    FfiPredictor *p_;
};

}  // namespace fw
