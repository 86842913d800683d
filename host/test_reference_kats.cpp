// The reference's own unit tests for the LR+FFM path, re-expressed against the C++ host mirror (host/fw_host.hpp) so
// that they read like the originals.  Every test cites the reference #[test] it follows.  Run on the GPU box
// (pytest -m gpu runs this binary); exits non-zero on the first failed assertion.
//
// The reference asserts exact f32 equality; the device sums the logit in a different order, so values agree to ~1e-7
// and are compared with EPS = 2e-6 (north star: per-example log-loss within 1e-4).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "fw_host.hpp"

using namespace fw;

static int g_checks = 0;
#define ASSERT_NEAR(x, y)                                                                                  \
    do {                                                                                                   \
        const double x_ = (x), y_ = (y);                                                                   \
        g_checks++;                                                                                        \
        if (!(std::fabs(x_ - y_) < 2e-6)) {                                                                \
            std::printf("FAILED %s:%d: %s = %.9g, expected %.9g\n", __FILE__, __LINE__, #x, x_, y_);         \
            std::exit(1);                                                                                  \
        }                                                                                                  \
    } while (0)
#define ASSERT_TRUE(c)                                                       \
    do {                                                                     \
        g_checks++;                                                          \
        if (!(c)) {                                                          \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);        \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

// regressor.rs:544-552
static FeatureBuffer lr_vec(std::vector<HashAndValue> v) {
    FeatureBuffer fb;
    fb.label = 0.0f;
    fb.example_importance = 1.0f;
    fb.lr_buffer = std::move(v);
    return fb;
}
// block_ffm.rs:1219-1227
static FeatureBuffer ffm_vec(std::vector<HashAndValueAndSeq> v) {
    FeatureBuffer fb;
    fb.ffm_buffer = std::move(v);
    return fb;
}
// persistence.rs:420-433
static FeatureBuffer lr_and_ffm_vec(std::vector<HashAndValue> v1, std::vector<HashAndValueAndSeq> v2) {
    FeatureBuffer fb;
    fb.lr_buffer = std::move(v1);
    fb.ffm_buffer = std::move(v2);
    return fb;
}

// regressor.rs:556-595
static void test_learning_turned_off() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.optimizer = Optimizer::AdagradLUT;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    ASSERT_NEAR(re.learn(lr_vec({}), pb, false), 0.5);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, false), 0.5);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}, {2, 1.0f, 0}}), pb, false), 0.5);
}

// regressor.rs:597-628
static void test_power_t_zero() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.optimizer = Optimizer::AdagradFlex;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    const FeatureBuffer vec_in = lr_vec({{1, 1.0f, 0}});
    ASSERT_NEAR(re.learn(vec_in, pb, true), 0.5);
    ASSERT_NEAR(re.learn(vec_in, pb, true), 0.48750263);
    ASSERT_NEAR(re.learn(vec_in, pb, true), 0.47533244);
}

// regressor.rs:630-658: what happens on collision depends on the order of the math
static void test_double_same_feature() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.optimizer = Optimizer::AdagradLUT;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    const FeatureBuffer vec_in = lr_vec({{1, 1.0f, 0}, {1, 2.0f, 0}});
    ASSERT_NEAR(re.learn(vec_in, pb, true), 0.5);
    ASSERT_NEAR(re.learn(vec_in, pb, true), 0.38936076);
    ASSERT_NEAR(re.learn(vec_in, pb, true), 0.30993468);
}

// regressor.rs:660-708
static void test_power_t_half() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.5f;
    mi.init_acc_gradient = 0.0f;
    mi.optimizer = Optimizer::AdagradFlex;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, true), 0.5);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, true), 0.4750208);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, true), 0.45788094);
}

// regressor.rs:710-752 (FASTMATH_LR_LUT_BITS == 11)
static void test_power_t_half_fastmath() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.5f;
    mi.optimizer = Optimizer::AdagradLUT;
    mi.init_acc_gradient = 0.0f;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, true), 0.5);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, true), 0.475734);
}

// regressor.rs:754-816
static void test_power_t_half_two_features() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.5f;
    mi.bit_precision = 18;
    mi.init_acc_gradient = 0.0f;
    mi.optimizer = Optimizer::AdagradFlex;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}, {2, 1.0f, 0}}), pb, true), 0.5);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}, {2, 1.0f, 0}}), pb, true), 0.45016602);
    ASSERT_NEAR(re.learn(lr_vec({{1, 1.0f, 0}}), pb, true), 0.45836908);
}

// regressor.rs:818-866
static void test_non_one_weight() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.optimizer = Optimizer::AdagradLUT;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    ASSERT_NEAR(re.learn(lr_vec({{1, 2.0f, 0}}), pb, true), 0.5);
    ASSERT_NEAR(re.learn(lr_vec({{1, 2.0f, 0}}), pb, true), 0.45016602);
    ASSERT_NEAR(re.learn(lr_vec({{1, 2.0f, 0}}), pb, true), 0.40611085);
}

// regressor.rs:868-884
static void test_example_importance() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.optimizer = Optimizer::AdagradLUT;
    Regressor re(mi);
    PortBuffer pb = re.new_portbuffer();
    FeatureBuffer fb_instance = lr_vec({{1, 1.0f, 0}});
    fb_instance.example_importance = 0.5f;
    ASSERT_NEAR(re.learn(fb_instance, pb, true), 0.5);
    ASSERT_NEAR(re.learn(fb_instance, pb, true), 0.49375027);
    ASSERT_NEAR(re.learn(fb_instance, pb, true), 0.4875807);
}

// block_ffm.rs:1238-1327 (FFM block wired straight into the sigmoid)
static void test_ffm_k1() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.ffm_learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.ffm_power_t = 0.0f;
    mi.ffm_k = 1;
    mi.ffm_fields = {{}, {}};
    mi.wiring = FWGPU_WIRING_FFM_ONLY;
    mi.optimizer = Optimizer::AdagradFlex;
    {
        Regressor re(mi);
        PortBuffer pb;
        re.ffm_fill(1.0f);
        const FeatureBuffer fb = ffm_vec({{1, 1.0f, 0}, {100, 1.0f, mi.ffm_k}});
        ASSERT_NEAR(re.predict(fb, pb), 0.7310586);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.7310586);
        ASSERT_NEAR(re.predict(fb, pb), 0.7024794);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.7024794);
    }
    mi.optimizer = Optimizer::AdagradLUT;
    {
        Regressor re(mi);
        PortBuffer pb;
        re.ffm_fill(1.0f);
        const FeatureBuffer fb = ffm_vec({{1, 2.0f, 0}, {100, 2.0f, mi.ffm_k}});
        ASSERT_NEAR(re.predict(fb, pb), 0.98201376);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.98201376);
        ASSERT_NEAR(re.predict(fb, pb), 0.81377685);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.81377685);
    }
}

// block_ffm.rs:1449-1531
static void test_ffm_k4() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.ffm_learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.ffm_power_t = 0.0f;
    mi.ffm_k = 4;
    mi.ffm_fields = {{}, {}};
    mi.wiring = FWGPU_WIRING_FFM_ONLY;
    mi.optimizer = Optimizer::AdagradFlex;
    {
        Regressor re(mi);
        PortBuffer pb;
        re.ffm_fill(1.0f);
        const FeatureBuffer fb = ffm_vec({{1, 1.0f, 0}, {100, 1.0f, mi.ffm_k}});
        ASSERT_NEAR(re.predict(fb, pb), 0.98201376);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.98201376);
        ASSERT_NEAR(re.predict(fb, pb), 0.96277946);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.96277946);
    }
    mi.optimizer = Optimizer::AdagradLUT;
    {
        Regressor re(mi);
        PortBuffer pb;
        re.ffm_fill(1.0f);
        const FeatureBuffer fb = ffm_vec({{1, 2.0f, 0}, {100, 2.0f, mi.ffm_k}});
        ASSERT_NEAR(re.predict(fb, pb), 0.9999999);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.9999999);
        ASSERT_NEAR(re.predict(fb, pb), 0.99685884);
        ASSERT_NEAR(re.learn(fb, pb, true), 0.99685884);
    }
}

// block_ffm.rs:1657-1700
static void test_ffm_multivalue() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.ffm_k = 1;
    mi.ffm_power_t = 0.0f;
    mi.ffm_learning_rate = 0.1f;
    mi.ffm_fields = {{}, {}};
    mi.wiring = FWGPU_WIRING_FFM_ONLY;
    mi.optimizer = Optimizer::AdagradLUT;
    Regressor re(mi);
    PortBuffer pb;
    re.ffm_fill(1.0f);
    const FeatureBuffer fbuf = ffm_vec({{1, 1.0f, 0}, {3 * 1000, 1.0f, 0}, {100, 2.0f, mi.ffm_k}});
    ASSERT_NEAR(re.predict(fbuf, pb), 0.9933072);
    ASSERT_NEAR(re.learn(fbuf, pb, true), 0.9933072);
    ASSERT_NEAR(re.predict(fbuf, pb), 0.9395168);
    ASSERT_NEAR(re.learn(fbuf, pb, false), 0.9395168);
}

// persistence.rs:436-560 (full regressor: LR + FFM + Triangle), incl. the weight-blob round trip of 562-643
static void test_hogwild_load() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.power_t = 0.0f;
    mi.ffm_k = 1;
    mi.ffm_power_t = 0.0f;
    mi.ffm_learning_rate = 0.1f;
    mi.ffm_fields = {{}, {}};
    mi.optimizer = Optimizer::AdagradFlex;
    Regressor re_1(mi), re_2(mi);
    PortBuffer pb_1, pb_2;
    re_1.ffm_fill(1.0f);
    re_2.ffm_fill(1.0f);
    const FeatureBuffer fbuf_1 = lr_and_ffm_vec({{52, 0.5f, 0}, {2, 1.0f, 0}}, {{1, 0.5f, 0}, {3 * 1000, 1.0f, 0}, {101, 2.0f, 1}});
    const FeatureBuffer fbuf_2 = lr_and_ffm_vec({{1, 1.0f, 0}, {2, 1.0f, 0}}, {{1, 1.0f, 0}, {3 * 1000, 1.0f, 0}, {100, 2.0f, 1}});
    ASSERT_NEAR(re_1.learn(fbuf_1, pb_1, true), 0.97068775);
    const double expected_result_1_on_1 = 0.8922257;
    ASSERT_NEAR(re_1.learn(fbuf_1, pb_1, false), expected_result_1_on_1);
    ASSERT_NEAR(re_1.predict(fbuf_1, pb_1), expected_result_1_on_1);
    ASSERT_NEAR(re_2.learn(fbuf_2, pb_2, true), 0.9933072);
    const double expected_result_2_on_2 = 0.92719215;
    ASSERT_NEAR(re_2.learn(fbuf_2, pb_2, false), expected_result_2_on_2);
    ASSERT_NEAR(re_2.predict(fbuf_2, pb_2), expected_result_2_on_2);
    ASSERT_NEAR(re_2.learn(fbuf_1, pb_2, false), 0.93763095);
    ASSERT_NEAR(re_1.learn(fbuf_2, pb_1, false), 0.98559695);
    // save both, load 1 into a fresh regressor, hot-swap 2 over it and back (persistence.rs:562-600, hogwild_load 176-186)
    const std::vector<uint8_t> w1 = re_1.write_weights_to_buf(), w2 = re_2.write_weights_to_buf();
    Regressor new_re_1(mi);
    new_re_1.overwrite_weights_from_buf(w1);
    ASSERT_NEAR(new_re_1.learn(fbuf_1, pb_1, false), expected_result_1_on_1);
    ASSERT_NEAR(new_re_1.predict(fbuf_2, pb_2), 0.98559695);
    new_re_1.overwrite_weights_from_buf(w2);
    ASSERT_NEAR(new_re_1.learn(fbuf_2, pb_1, false), expected_result_2_on_2);
    new_re_1.overwrite_weights_from_buf(w1);
    ASSERT_NEAR(new_re_1.predict(fbuf_1, pb_1), expected_result_1_on_1);
    ASSERT_TRUE(new_re_1.get_name() == "Regressor with optimizer \"AdagradFlex\"");
}

// feature_buffer.rs:381-403, 507-543, 642-742
static void test_translation() {
    {
        ModelInstance mi = ModelInstance::new_empty();
        mi.add_constant_feature = true;
        mi.feature_combo_descs.push_back({{{0, false}}, 1.0f});
        FeatureBufferTranslator fbt(mi);
        fbt.translate({100, 1, 0x3f800000u, 0x80000000u}, 0);  // no feature
        ASSERT_TRUE(fbt.feature_buffer.lr_buffer.size() == 1 && fbt.feature_buffer.lr_buffer[0].hash == 116060 &&
                    fbt.feature_buffer.lr_buffer[0].value == 1.0f && fbt.feature_buffer.lr_buffer[0].combo_index == 1);
    }
    {
        ModelInstance mi = ModelInstance::new_empty();
        mi.add_constant_feature = false;
        mi.feature_combo_descs.push_back({{{0, false}, {1, false}}, 1.0f});
        FeatureBufferTranslator fbt(mi);
        fbt.translate({100, 1, 0x3f800000u, 2988156968u & 0x7fffffffu, 2422381320u & 0x7fffffffu, 0x80000000u}, 0);
        ASSERT_TRUE(fbt.feature_buffer.lr_buffer.size() == 1 && fbt.feature_buffer.lr_buffer[0].hash == 208368);
    }
    {
        ModelInstance mi = ModelInstance::new_empty();
        mi.add_constant_feature = false;
        mi.ffm_fields = {{{0, false}}, {{0, false}, {1, false}}, {{1, false}}};
        mi.ffm_k = 3;
        FeatureBufferTranslator fbt(mi);
        fbt.translate({100, 1, 0x3f800000u, 0x80000000u | (5u << 16) | 9u, 0x1, 0xfff, 0x40000000u, 0xfeb, 0x40400000u}, 0);
        const auto &f = fbt.feature_buffer.ffm_buffer;
        ASSERT_TRUE(f.size() == 6 && f[0].hash == 0xffc && f[0].value == 2.0f && f[0].contra_field_index == 0);
        ASSERT_TRUE(f[1].hash == 0xfe8 && f[1].value == 3.0f && f[3].contra_field_index == 3);
        ASSERT_TRUE(f[4].hash == 0x0 && f[4].contra_field_index == 3 && f[5].contra_field_index == 6);
    }
}

// hogwild.rs: records digested in micro-batches train the shared regressor; errors surface as exceptions
static void test_hogwild_trainer() {
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = 0.1f;
    mi.optimizer = Optimizer::AdagradLUT;
    mi.bit_precision = 14;
    mi.feature_combo_descs.push_back({{{0, false}}, 1.0f});
    Regressor re(mi);
    PortBuffer pb;
    FeatureBufferTranslator fbt(mi);
    const std::vector<uint32_t> rec = {4, 1, 0x3f800000u, 77};  // label 1, one feature
    fbt.translate(rec, 0);
    const float before = re.predict(fbt.feature_buffer, pb);
    {
        HogwildTrainer tr(re, mi, 16, 8);
        for (int i = 0; i < 50; i++) tr.digest_example(rec);
        tr.block_until_workers_finished();
    }
    const float after = re.predict(fbt.feature_buffer, pb);
    ASSERT_NEAR(before, 0.5);
    ASSERT_TRUE(after > 0.6f);  // it learned that this feature means label 1
    bool threw = false;
    try {
        ModelInstance bad = ModelInstance::new_empty();
        bad.ffm_k = 8;
        bad.ffm_fields.resize(80);  // 8 * 80 * 80 > FFM_CONTRA_BUF_LEN (block_ffm.rs:96-101)
        Regressor r2(bad);
    } catch (const std::runtime_error &) {
        threw = true;
    }
    ASSERT_TRUE(threw);
}

// parser.rs:474-859 test_vowpal (a representative subset; the whole table runs in tests/test_feed_cpu.py)
static void test_vowpal() {
    const uint32_t ONE = 1065353216u, NOF = 1u << 31, M31 = NOF - 1, NOT_SINGLE = NOF;
    auto nd = [&](uint32_t a, uint32_t b) { return ((a << 16) + b) | NOT_SINGLE; };
    VwNamespaceMap vw("\nA,featureA\nB,featureB\nC,featureC\n");
    VowpalParser rr(vw);
    auto eq = [](const std::vector<uint32_t> &a, std::vector<uint32_t> b) { return a == b; };
    ASSERT_TRUE(eq(rr.next_vowpal("1 |A a\n"), {6, 1, ONE, 2988156968u & M31, NOF, NOF}));
    ASSERT_TRUE(eq(rr.next_vowpal("-1 |B b\n"), {6, 0, ONE, NOF, 2422381320u & M31, NOF}));
    ASSERT_TRUE(eq(rr.next_vowpal("1 |A a b\n"), {10, 1, ONE, nd(6, 10), NOF, NOF, 2988156968u & M31, ONE, 3529656005u & M31, ONE}));
    ASSERT_TRUE(eq(rr.next_vowpal("1 |A:3 a:2.0\n"), {8, 1, ONE, nd(6, 8), NOF, NOF, 2988156968u & M31, 0x40c00000u}));  // 6.0f
    ASSERT_TRUE(eq(rr.next_vowpal("|A a\n"), {6, 0xff, ONE, 2988156968u & M31, NOF, NOF}));
    ASSERT_TRUE(rr.next_vowpal("").empty());
    bool flush = false, load = false;
    std::string msg;
    try { rr.next_vowpal("flush"); } catch (const FlushCommand &) { flush = true; }
    try { rr.next_vowpal("hogwild_load   /path/to/filename  "); } catch (const HogwildLoadCommand &e) { load = e.filename == "/path/to/filename"; }
    try { rr.next_vowpal("1 |UNDECLARED_NAMESPACE a\n"); } catch (const std::runtime_error &e) { msg = e.what(); }
    ASSERT_TRUE(flush && load);
    ASSERT_TRUE(msg == "Feature name was not predeclared in vw_namespace_map.csv: UNDECLARED_NAMESPACE");
    // parser.rs:1096-1123 test_cache: the cached context text followed by the request's text
    VwNamespaceMap vw2("\nAA,featureA\nBB,featureB\nCC,featureC\n");
    VowpalParser r2(vw2);
    const std::vector<uint32_t> full = {8, 255, 1065353216, 2147876872u, 1123906636, 2147483648u, 292540976, 1086324736};
    ASSERT_TRUE(r2.next_vowpal("|BB b |AA:3 a:2.0 \n") == full);
    ASSERT_TRUE(r2.next_vowpal_with_cache("|BB b ", "|AA:3 a:2.0 \n") == full);
}

// cache.rs + persistence.rs + lib.rs in one chain (persistence.rs:206-643 style: what is saved predicts the same when loaded)
static void test_cache_model_file_and_predictor() {
    VwNamespaceMap vw("A,animal\nB,food\n");
    VowpalParser pa(vw);
    const std::string dir = "/tmp/fw_host_test";
    std::system(("rm -rf " + dir + " && mkdir -p " + dir).c_str());
    const std::string input = dir + "/train.vw";
    std::vector<std::vector<uint32_t>> recs;
    {
        RecordCache c(input, true, vw);
        ASSERT_TRUE(c.writing && !c.reading);
        for (int i = 0; i < 200; i++) {
            const std::string line = std::string(i % 3 ? "1" : "-1") + " |A a" + std::to_string(i % 7) + " |B f" + std::to_string(i % 5) + (i % 4 ? "" : ":1.5") + "\n";
            recs.push_back(pa.next_vowpal(line));
            c.push_record(recs.back());
        }
        c.write_finish();
    }
    ModelInstance mi = ModelInstance::new_empty();
    mi.learning_rate = mi.ffm_learning_rate = 0.1f;
    mi.power_t = mi.ffm_power_t = 0.0f;
    mi.bit_precision = 14;
    mi.ffm_bit_precision = 14;
    mi.ffm_k = 4;
    mi.ffm_init_acc_gradient = 1.0f;
    mi.optimizer = Optimizer::AdagradLUT;
    const NamespaceDescriptor a = vw.descriptor("A"), b = vw.descriptor("B");
    mi.feature_combo_descs = {{{a}, 1.0f}, {{b}, 1.0f}, {{a, b}, 1.0f}};
    mi.ffm_fields = {{a}, {b}};
    Regressor re(mi);
    FeatureBufferTranslator fbt(mi);
    PortBuffer pb;
    {
        RecordCache c(input, true, vw);  // second open: the cache is read, record by record, identical to what was pushed
        ASSERT_TRUE(c.reading && !c.writing);
        size_t n = 0;
        for (;;) {
            const std::vector<uint32_t> r = c.get_next_record();
            if (r.empty()) break;
            ASSERT_TRUE(n < recs.size() && r == recs[n]);
            fbt.translate(r, n);
            re.learn(fbt.feature_buffer, pb, true);
            n++;
        }
        ASSERT_TRUE(n == 200);
    }
    const std::string model = dir + "/model.fw", inference = dir + "/model.fw.inference";
    persistence::save_regressor_to_filename(model, mi, vw, re);
    persistence::convert_inference_regressor(model, inference);
    Predictor pr("fw -i " + inference + " -t");
    for (int i = 0; i < 10; i++) {
        const std::string req = "|A a" + std::to_string(i % 7) + " |B f" + std::to_string(i % 5) + "\n";
        fbt.translate(pa.next_vowpal(req), 0);
        const float want = re.predict(fbt.feature_buffer, pb);
        ASSERT_NEAR(pr.predict(req), want);
        ASSERT_TRUE(pr.setup_cache("|A a" + std::to_string(i % 7) + " \n") == 0.0f);
        ASSERT_NEAR(pr.predict_with_cache("|B f" + std::to_string(i % 5) + "\n"), want);
    }
    ASSERT_TRUE(pr.predict("") == -1.0f && pr.predict("|Z z\n") == -1.0f);  // EOF / parse error codes (lib.rs:47-48)
    Regressor fresh(mi);
    persistence::hogwild_load(fresh, model);  // persistence.rs:176-187
    ASSERT_TRUE(fresh.write_weights_to_buf() == re.write_weights_to_buf());
}

int main() {
    struct T {
        const char *name;
        void (*fn)();
    } tests[] = {{"test_learning_turned_off", test_learning_turned_off},
                 {"test_power_t_zero", test_power_t_zero},
                 {"test_double_same_feature", test_double_same_feature},
                 {"test_power_t_half__", test_power_t_half},
                 {"test_power_t_half_fastmath", test_power_t_half_fastmath},
                 {"test_power_t_half_two_features", test_power_t_half_two_features},
                 {"test_non_one_weight", test_non_one_weight},
                 {"test_example_importance", test_example_importance},
                 {"test_ffm_k1", test_ffm_k1},
                 {"test_ffm_k4", test_ffm_k4},
                 {"test_ffm_multivalue", test_ffm_multivalue},
                 {"test_hogwild_load", test_hogwild_load},
                 {"test_translation", test_translation},
                 {"test_hogwild_trainer", test_hogwild_trainer},
                 {"test_vowpal", test_vowpal},
                 {"test_cache_model_file_and_predictor", test_cache_model_file_and_predictor}};
    for (const auto &t : tests) {
        t.fn();
        std::printf("ok %s\n", t.name);
    }
    std::printf("all %zu tests passed (%d assertions)\n", sizeof(tests) / sizeof(tests[0]), g_checks);
    return 0;
}
