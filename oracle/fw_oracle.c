/*
 * fw_oracle.c -- CPU restatement of the fwumious_wabbit LR+FFM hot path (see fw_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: parity oracle + CPU baseline.  Never linked into the product.
 *
 * Compile with -ffp-contract=off (no FMA contraction): the reference's default build
 * (`cargo build --release`, x86_64 baseline = SSE2) has no `target_feature="fma"`, so
 * block_ffm.rs:933-944 (mul then add) is the variant restated here.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference/src/).
 */
#include "fw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ bits */
static inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* ------------------------------------------------------------------ optimizers */

/* optimizer.rs:121-144 OptimizerAdagradLUT::init */
void fwo_lut_init(float *lut, float learning_rate, float power_t, float init_acc) {
    float minus_power_t = -power_t;
    for (uint32_t x = 0; x < FWO_LUT_SIZE; x++) {
        float float_x = u2f(x << (31 - FWO_LUT_BITS)) + init_acc;
        float float_x_plus_one = u2f((x + 1) << (31 - FWO_LUT_BITS)) + init_acc;
        float val = learning_rate * (powf(float_x, minus_power_t) + powf(float_x_plus_one, minus_power_t)) * 0.5f;
        if (isnan(val) || isinf(val)) val = learning_rate;
        lut[x] = val;
    }
}

/* optimizer.rs:36-38 */
float fwo_step_sgd(float lr, float g) { return g * lr; }

/* optimizer.rs:76-88 */
float fwo_step_flex(float lr, float minus_power_t, float g, float *acc) {
    float gradient_squared = g * g;
    float new_acc = *acc + gradient_squared;
    *acc = new_acc;
    float update = g * lr * powf(new_acc, minus_power_t);
    if (isnan(update) || isinf(update)) return 0.0f;
    return update;
}

/* optimizer.rs:147-156 */
float fwo_step_lut(const float *lut, float g, float *acc) {
    float gradient_squared = g * g;
    float new_acc = *acc + gradient_squared;
    *acc = new_acc;
    /* (a NaN with the sign bit set -- a diverged model -- would index past the table: the reference's bounds check panics there,
     * the restatement stays inside the table like the library does) */
    uint32_t key = (f2u(new_acc) >> (31 - FWO_LUT_BITS)) & (uint32_t)(FWO_LUT_SIZE - 1);
    return g * lut[key];
}

/* ------------------------------------------------------------------ hashing */

/* murmur3 x86_32 (Austin Appleby, public domain algorithm) == fasthash::murmur3::hash32_with_seed,
 * call sites parser.rs:83, 382-385; pinned by the parser.rs KATs in tests/golden/hash_kat.json. */
uint32_t fwo_murmur3_32(const uint8_t *data, size_t len, uint32_t seed) {
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
    size_t nblocks = len / 4;
    for (size_t i = 0; i < nblocks; i++) {
        uint32_t k1;
        memcpy(&k1, data + 4 * i, 4);
        k1 *= c1;
        k1 = (k1 << 15) | (k1 >> 17);
        k1 *= c2;
        h1 ^= k1;
        h1 = (h1 << 13) | (h1 >> 19);
        h1 = h1 * 5 + 0xe6546b64u;
    }
    const uint8_t *tail = data + nblocks * 4;
    uint32_t k1 = 0;
    switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1:
        k1 ^= tail[0];
        k1 *= c1;
        k1 = (k1 << 15) | (k1 >> 17);
        k1 *= c2;
        h1 ^= k1;
    }
    h1 ^= (uint32_t)len;
    h1 ^= h1 >> 16;
    h1 *= 0x85ebca6bu;
    h1 ^= h1 >> 13;
    h1 *= 0xc2b2ae35u;
    h1 ^= h1 >> 16;
    return h1;
}

/* merand48 0.1.0 (= VowpalWabbit rand48.cc merand48): one LCG step from `seed`, 23 mantissa bits
 * -> [0,1).  Call sites block_ffm.rs:801, 811.  PARITY UNPINNED (no reference test observes it). */
float fwo_merand48(uint64_t seed) {
    const uint64_t a = 0xeece66d5deece66dULL, c = 2147483647ULL;
    seed = a * seed + c;
    uint32_t temp = (uint32_t)((seed >> 25) & 0x7FFFFF) | (127u << 23);
    return u2f(temp) - 1.0f;
}

/* ------------------------------------------------------------------ model */

struct fwo_model {
    fwo_config cfg;
    uint64_t lr_len;  /* 1 << bit_precision, block_lr.rs:67 */
    float *lr;        /* interleaved {weight, acc}, block_helpers.rs:23-28 */
    uint64_t ffm_len; /* (1<<ffm_bits) + F*k, block_ffm.rs:92-94 */
    float *ffm_w, *ffm_acc; /* separate arrays, block_ffm.rs:40-41 */
    float lut_lr[FWO_LUT_SIZE], lut_ffm[FWO_LUT_SIZE];
    /* deep head */
    fwo_nn_config nn;
    uint32_t nn_in[FWO_NN_MAX_LAYERS + 1], nn_out[FWO_NN_MAX_LAYERS + 1]; /* per layer incl. the final neuron */
    float *nn_w[FWO_NN_MAX_LAYERS + 1], *nn_acc[FWO_NN_MAX_LAYERS + 1];
    float lut_nn[FWO_LUT_SIZE];
    /* per-thread scratch lives on the caller's stack/heap: see fwo_scratch */
};

typedef struct {
    float *contra;  /* F*F*k  (FFM_CONTRA_BUF_LEN role, regressor.rs:23) */
    float *grads;   /* n_ffm*F*k (local_data_ffm_values, block_ffm.rs:294-312) */
    uint32_t grads_cap;
    float *ffm_out; /* F*F */
    float *lr_out;  /* C */
    float *tri;     /* F(F+1)/2 */
} fwo_scratch;

static void scratch_init(fwo_scratch *s, const fwo_config *c) {
    uint32_t F = c->ffm_num_fields, k = c->ffm_k;
    s->contra = (float *)malloc(sizeof(float) * (size_t)(F * F * k + 16));
    s->grads = NULL;
    s->grads_cap = 0;
    s->ffm_out = (float *)malloc(sizeof(float) * (size_t)(F * F + 1));
    s->lr_out = (float *)malloc(sizeof(float) * (size_t)(c->num_combos + 1));
    s->tri = (float *)malloc(sizeof(float) * (size_t)(F * (F + 1) / 2 + 1));
}
static void scratch_free(fwo_scratch *s) {
    free(s->contra);
    free(s->grads);
    free(s->ffm_out);
    free(s->lr_out);
    free(s->tri);
}

fwo_model *fwo_create(const fwo_config *cfg) {
    /* block_ffm.rs:96-101 guard */
    if ((uint64_t)cfg->ffm_k * cfg->ffm_num_fields * cfg->ffm_num_fields > 41472) return NULL;
    fwo_model *m = (fwo_model *)calloc(1, sizeof(fwo_model));
    m->cfg = *cfg;
    m->lr_len = 1ULL << cfg->bit_precision;
    m->ffm_len = cfg->ffm_k > 0 ? (1ULL << cfg->ffm_bit_precision) + (uint64_t)cfg->ffm_num_fields * cfg->ffm_k : 0;
    m->lr = (float *)calloc(m->lr_len * 2, sizeof(float));
    if (m->ffm_len) {
        m->ffm_w = (float *)calloc(m->ffm_len, sizeof(float));
        m->ffm_acc = (float *)calloc(m->ffm_len, sizeof(float));
    }
    /* block_lr.rs:63-65, block_ffm.rs:86-91: each block owns its optimizer instance */
    fwo_lut_init(m->lut_lr, cfg->learning_rate, cfg->power_t, cfg->init_acc_gradient);
    fwo_lut_init(m->lut_ffm, cfg->ffm_learning_rate, cfg->ffm_power_t, cfg->ffm_init_acc_gradient);
    return m;
}

void fwo_free(fwo_model *m) {
    if (!m) return;
    free(m->lr);
    free(m->ffm_w);
    free(m->ffm_acc);
    for (int l = 0; l <= FWO_NN_MAX_LAYERS; l++) {
        free(m->nn_w[l]);
        free(m->nn_acc[l]);
    }
    free(m);
}

/* initial_data(): optimizer.rs:40-42 (SGD: none), 90-92 (Flex: init_acc), 158-161 (LUT: 0.0) */
static float initial_acc(int optimizer, float init_acc) {
    return optimizer == FWO_OPT_ADAGRAD_FLEX ? init_acc : 0.0f;
}

static void nn_init_weights(fwo_model *m);
static float chain_nn(fwo_model *m, fwo_scratch *s, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm,
                      uint32_t n_ffm, float label, float importance, int update, int training_numerics);

void fwo_init_weights(fwo_model *m) {
    const fwo_config *c = &m->cfg;
    /* block_lr.rs:97-105: weight 0.0, acc initial_data() */
    float a0 = initial_acc(c->optimizer, c->init_acc_gradient);
    for (uint64_t i = 0; i < m->lr_len; i++) {
        m->lr[2 * i] = 0.0f;
        m->lr[2 * i + 1] = a0;
    }
    if (m->nn.n_layers) nn_init_weights(m);
    if (!m->ffm_len) return;
    /* block_ffm.rs:784-829 */
    float fa0 = initial_acc(c->optimizer, c->ffm_init_acc_gradient);
    if (c->ffm_init_width == 0.0f) {
        float ffm_one_over_k_root = 1.0f / sqrtf((float)c->ffm_k) / 50.0f;
        for (uint64_t i = 0; i < m->ffm_len; i++) {
            m->ffm_w[i] = (1.0f * fwo_merand48(m->ffm_len + i) - 0.5f) * ffm_one_over_k_root;
            m->ffm_acc[i] = fa0;
        }
    } else {
        float zero_half_band_width = c->ffm_init_width * c->ffm_init_zero_band * 0.5f;
        float band_width = c->ffm_init_width * (1.0f - c->ffm_init_zero_band);
        for (uint64_t i = 0; i < m->ffm_len; i++) {
            float w = fwo_merand48(i) * band_width - band_width * 0.5f;
            if (w > 0.0f) w += zero_half_band_width;
            else w -= zero_half_band_width;
            w += c->ffm_init_center;
            m->ffm_w[i] = w;
            m->ffm_acc[i] = fa0;
        }
    }
}

/* block_ffm.rs:1228-1235 / persistence.rs:315-330 */
void fwo_ffm_fill(fwo_model *m, float w) {
    float fa0 = initial_acc(m->cfg.optimizer, m->cfg.ffm_init_acc_gradient);
    for (uint64_t i = 0; i < m->ffm_len; i++) {
        m->ffm_w[i] = w;
        m->ffm_acc[i] = fa0;
    }
}

float *fwo_lr_table(fwo_model *m, uint64_t *n) {
    if (n) *n = m->lr_len;
    return m->lr;
}
float *fwo_ffm_weights(fwo_model *m, uint64_t *n) {
    if (n) *n = m->ffm_len;
    return m->ffm_w;
}
float *fwo_ffm_acc(fwo_model *m, uint64_t *n) {
    if (n) *n = m->ffm_len;
    return m->ffm_acc;
}
const float *fwo_lut_lr(fwo_model *m) { return m->lut_lr; }
const float *fwo_lut_ffm(fwo_model *m) { return m->lut_ffm; }

static inline float opt_step(int optimizer, float lr, float power_t, const float *lut, float g, float *acc) {
    switch (optimizer) {
    case FWO_OPT_SGD: return fwo_step_sgd(lr, g);
    case FWO_OPT_ADAGRAD_FLEX: return fwo_step_flex(lr, -power_t, g, acc);
    default: return fwo_step_lut(lut, g, acc);
    }
}

/* ------------------------------------------------------------------ blocks */

/* block_loss_functions.rs:15-17 */
static inline float logistic(float t) { return 1.0f / (1.0f + expf(-t)); }

/* block_lr.rs:28-47 */
static void lr_forward(const fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, float *lr_out) {
    for (uint32_t c = 0; c < m->cfg.num_combos; c++) lr_out[c] = 0.0f;
    for (uint32_t i = 0; i < n_lr; i++) lr_out[lr[i].combo_index] += m->lr[2 * (uint64_t)lr[i].hash] * lr[i].value;
}

/* block_lr.rs:135-150 */
static void lr_update(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const float *lr_out) {
    const fwo_config *c = &m->cfg;
    for (uint32_t i = 0; i < n_lr; i++) {
        uint64_t h = lr[i].hash;
        float gradient = lr_out[lr[i].combo_index] * lr[i].value;
        float update = opt_step(c->optimizer, c->learning_rate, c->power_t, m->lut_lr, gradient, &m->lr[2 * h + 1]);
        m->lr[2 * h] -= update;
    }
}

/* block_misc.rs:864-883 */
void fwo_triangle_forward(const float *in, uint32_t width, float *out) {
    uint32_t o = 0;
    for (uint32_t i = 0; i < width; i++) {
        for (uint32_t j = 0; j < i; j++) out[o++] = in[i * width + j] * 2.0f;
        out[o++] = in[i * width + i];
    }
}

/* block_misc.rs:822-832 */
void fwo_triangle_backward(const float *gout, uint32_t width, float *gin) {
    uint32_t o = 0;
    for (uint32_t i = 0; i < width; i++)
        for (uint32_t j = 0; j <= i; j++) {
            gin[i * width + j] = gout[o];
            gin[j * width + i] = gout[o];
            o++;
        }
}

/* block_ffm.rs:163-261: pass 1 (contra fields) + pass 2 (gradient cache + outputs) */
static void ffm_fb_forward(const fwo_model *m, const fwo_ffm_entry *fb, uint32_t n, fwo_scratch *s) {
    const uint32_t F = m->cfg.ffm_num_fields, k = m->cfg.ffm_k, fc = F * k;
    const float *W = m->ffm_w;
    float *contra = s->contra, *out = s->ffm_out, *G = s->grads;
    for (uint32_t i = 0; i < F * F; i++) out[i] = 0.0f; /* 137-139 */

    uint32_t idx = 0;
    for (uint32_t field = 0; field < F; field++) { /* 165-217 */
        uint32_t field_k = field * k;
        if (idx >= n || fb[idx].contra_field_index > field_k) {
            uint32_t off = field_k;
            for (uint32_t z = 0; z < F; z++) {
                for (uint32_t kk = off; kk < off + k; kk++) contra[kk] = 0.0f;
                off += fc;
            }
            continue;
        }
        int first = 1;
        while (idx < n && fb[idx].contra_field_index == field_k) {
            float v = fb[idx].value;
            uint64_t fi = fb[idx].hash;
            uint32_t off = field_k;
            if (first) {
                for (uint32_t z = 0; z < F; z++) {
                    for (uint32_t kk = 0; kk < k; kk++) contra[off + kk] = W[fi + kk] * v;
                    off += fc;
                    fi += k;
                }
                first = 0;
            } else {
                for (uint32_t z = 0; z < F; z++) {
                    for (uint32_t kk = 0; kk < k; kk++) contra[off + kk] += W[fi + kk] * v;
                    off += fc;
                    fi += k;
                }
            }
            idx++;
        }
    }

    uint32_t goff = 0;
    for (uint32_t i = 0; i < n; i++) { /* 220-261 */
        float v = fb[i].value;
        uint64_t fi = fb[i].hash;
        uint32_t cfi = fb[i].contra_field_index;
        uint32_t contra_offset = cfi * F;
        uint32_t contra_offset2 = contra_offset / k;
        uint32_t vv = 0;
        for (uint32_t z = 0; z < F; z++) {
            float correction = 0.0f;
            uint64_t vfi = fi + vv;
            uint32_t vco = contra_offset + vv;
            if (vv == cfi) {
                for (uint32_t kk = 0; kk < k; kk++) {
                    float w = W[vfi + kk];
                    float contra_weight = contra[vco + kk] - w * v;
                    float gradient = v * contra_weight;
                    G[goff + kk] = gradient;
                    correction += w * gradient;
                }
            } else {
                for (uint32_t kk = 0; kk < k; kk++) {
                    float contra_weight = contra[vco + kk];
                    float gradient = v * contra_weight;
                    G[goff + kk] = gradient;
                    float w = W[vfi + kk];
                    correction += w * gradient;
                }
            }
            out[contra_offset2 + z] += correction * 0.5f;
            vv += k;
            goff += k;
        }
    }
}

/* block_ffm.rs:265-288 */
static void ffm_fb_update(fwo_model *m, const fwo_ffm_entry *fb, uint32_t n, fwo_scratch *s) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_num_fields, k = c->ffm_k;
    const float *out = s->ffm_out, *G = s->grads;
    uint32_t local_index = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t fi = fb[i].hash;
        uint32_t contra_offset = (fb[i].contra_field_index * F) / k;
        for (uint32_t z = 0; z < F; z++) {
            float general_gradient = out[contra_offset + z];
            for (uint32_t kk = 0; kk < k; kk++) {
                float feature_value = G[local_index];
                float gradient = general_gradient * feature_value;
                float update = opt_step(c->optimizer, c->ffm_learning_rate, c->ffm_power_t, m->lut_ffm, gradient,
                                        &m->ffm_acc[fi]);
                m->ffm_w[fi] -= update;
                local_index++;
                fi++;
            }
        }
    }
}

/* hadd_ps, block_ffm.rs:106-114: (r0+r2)+(r1+r3) */
static inline float hadd4(const float r[4]) { return (r[0] + r[2]) + (r[1] + r[3]); }

/* 8-wide dot as in calculate_interactions (block_ffm.rs:1132-1139, 1168-1176):
 * acc_0 = a[0..4]*b[0..4]; acc_1 = a[4..8]*b[4..8]; hadd(acc_0+acc_1) */
static inline float dot8_sse(const float *a, const float *b) {
    float s[4];
    for (int l = 0; l < 4; l++) s[l] = a[l] * b[l] + a[4 + l] * b[4 + l];
    return hadd4(s);
}

/* block_ffm.rs:316-440 (forward) with prepare_contra_fields 964-1104 and
 * calculate_interactions 1107-1201.  Non-FMA variant (933-944). */
static void ffm_forward(const fwo_model *m, const fwo_ffm_entry *fb, uint32_t n, fwo_scratch *s) {
    const uint32_t F = m->cfg.ffm_num_fields, k = m->cfg.ffm_k, R = F * k;
    const float *W = m->ffm_w;
    float *contra = s->contra, *out = s->ffm_out;
    for (uint32_t i = 0; i < F * F; i++) out[i] = 0.0f;

    uint32_t idx = 0;
    for (uint32_t field = 0; field < F; field++) {
        uint32_t field_k = field * k;
        uint32_t offset = field_k * F;
        if (idx >= n || fb[idx].contra_field_index > field_k) { /* 362-383 */
            for (uint32_t z = 0; z < R; z++) contra[offset + z] = 0.0f;
            continue;
        }
        uint32_t ffm_index = field * (F + 1);
        int first = 1;
        while (idx < n && fb[idx].contra_field_index == field_k) {
            uint64_t fi = fb[idx].hash;
            float v = fb[idx].value;
            /* prepare_contra_fields, 964-1104: element-wise, the 16-float unrolling does not
             * change per-element arithmetic */
            if (first) {
                first = 0;
                if (v == 1.0f) {
                    for (uint32_t z = 0; z < R; z++) contra[offset + z] = W[fi + z];
                } else {
                    for (uint32_t z = 0; z < R; z++) contra[offset + z] = W[fi + z] * v;
                }
            } else if (v == 1.0f) {
                for (uint32_t z = 0; z < R; z++) contra[offset + z] = W[fi + z] + contra[offset + z];
            } else {
                for (uint32_t z = 0; z < R; z++) {
                    float t = W[fi + z] * v; /* _mm_mul_ps then _mm_add_ps (933-944) */
                    contra[offset + z] = t + contra[offset + z];
                }
            }
            /* 418-426: diagonal pre-correction */
            float correction = 0.0f;
            for (uint64_t kk = fi + field_k; kk < fi + field_k + k; kk++) correction += W[kk] * W[kk];
            out[ffm_index] -= correction * 0.5f * v * v;
            idx++;
        }
    }

    /* calculate_interactions, 1107-1201 */
    const uint32_t LANES = 8;
    const uint32_t k_end = k - k % LANES;
    for (uint32_t f1 = 0; f1 < F; f1++) {
        uint32_t f1_offset = f1 * R;
        uint32_t f1_ffmk = f1 * k;
        uint32_t f1_offset_ffmk = f1_offset + f1_ffmk;
        float cf = 0.0f;
        if (k == LANES) {
            cf = dot8_sse(contra + f1_offset_ffmk, contra + f1_offset_ffmk);
        } else {
            for (uint32_t b = 0; b < k_end; b += LANES)
                cf += dot8_sse(contra + f1_offset_ffmk + b, contra + f1_offset_ffmk + b);
            for (uint32_t kk = k_end; kk < k; kk++)
                cf += contra[f1_offset_ffmk + kk] * contra[f1_offset_ffmk + kk];
        }
        out[f1 * F + f1] += cf * 0.5f;

        uint32_t f2_offset_ffmk = f1_offset + f1_ffmk;
        for (uint32_t f2 = f1 + 1; f2 < F; f2++) {
            f2_offset_ffmk += R;
            f1_offset_ffmk += k;
            float c = 0.0f;
            if (k == LANES) {
                c = dot8_sse(contra + f1_offset_ffmk, contra + f2_offset_ffmk);
            } else {
                for (uint32_t b = 0; b < k_end; b += LANES)
                    c += dot8_sse(contra + f1_offset_ffmk + b, contra + f2_offset_ffmk + b);
                for (uint32_t kk = k_end; kk < k; kk++) c += contra[f1_offset_ffmk + kk] * contra[f2_offset_ffmk + kk];
            }
            c *= 0.5f;
            out[f1 * F + f2] += c;
            out[f2 * F + f1] += c;
        }
    }
}

/* block_loss_functions.rs:105-153; returns p, writes general gradient */
static float sigmoid_block(const float *a, uint32_t na, const float *b, uint32_t nb, float label, float importance,
                           float *general_gradient) {
    float wsum = 0.0f; /* iter().sum(): left fold over the joined span (LR slots, then triangle) */
    for (uint32_t i = 0; i < na; i++) wsum += a[i];
    for (uint32_t i = 0; i < nb; i++) wsum += b[i];
    float p, g;
    if (isnan(wsum)) {
        p = logistic(0.0f);
        g = 0.0f;
    } else if (wsum < -50.0f) {
        p = logistic(-50.0f);
        g = 0.0f;
    } else if (wsum > 50.0f) {
        p = logistic(50.0f);
        g = 0.0f;
    } else {
        p = logistic(wsum);
        g = -(label - p) * importance;
    }
    if (general_gradient) *general_gradient = g;
    return p;
}

static void ensure_grads(fwo_scratch *s, uint32_t need) {
    if (need > s->grads_cap) {
        free(s->grads);
        s->grads_cap = need + 1024;
        s->grads = (float *)malloc(sizeof(float) * (size_t)s->grads_cap);
    }
}

/* The forward_backward chain (block_helpers.rs:219-228) for [LR, FFM, Triangle, Sigmoid]
 * (REGRESSOR) or [FFM, Sigmoid] (FFM_ONLY). */
static float chain_forward_backward(fwo_model *m, fwo_scratch *s, const fwo_lr_entry *lr, uint32_t n_lr,
                                    const fwo_ffm_entry *ffm, uint32_t n_ffm, float label, float importance,
                                    int update) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_num_fields;
    const int has_ffm = c->ffm_k > 0;
    float g, p;
    if (m->nn.n_layers && c->wiring == FWO_WIRING_REGRESSOR)
        return chain_nn(m, s, lr, n_lr, ffm, n_ffm, label, importance, update, 1);
    if (c->wiring == FWO_WIRING_FFM_ONLY) {
        ensure_grads(s, n_ffm * F * c->ffm_k);
        ffm_fb_forward(m, ffm, n_ffm, s);
        p = sigmoid_block(s->ffm_out, F * F, NULL, 0, label, importance, &g);
        for (uint32_t i = 0; i < F * F; i++) s->ffm_out[i] = g; /* block_loss_functions.rs:148-151 */
        if (update) ffm_fb_update(m, ffm, n_ffm, s);
        return p;
    }
    lr_forward(m, lr, n_lr, s->lr_out);
    uint32_t T = 0;
    if (has_ffm) {
        ensure_grads(s, n_ffm * F * c->ffm_k);
        ffm_fb_forward(m, ffm, n_ffm, s);
        fwo_triangle_forward(s->ffm_out, F, s->tri);
        T = F * (F + 1) / 2;
    }
    p = sigmoid_block(s->lr_out, c->num_combos, s->tri, T, label, importance, &g);
    for (uint32_t i = 0; i < c->num_combos; i++) s->lr_out[i] = g;
    if (has_ffm) {
        for (uint32_t i = 0; i < T; i++) s->tri[i] = g;
        /* Triangle mirrors only when update (block_misc.rs:812-833); FFM update reads ffm_out only if update */
        if (update) {
            fwo_triangle_backward(s->tri, F, s->ffm_out);
            ffm_fb_update(m, ffm, n_ffm, s); /* FFM update runs before LR's (block_lr.rs:133-150) */
        }
    }
    if (update) lr_update(m, lr, n_lr, s->lr_out);
    return p;
}

static float chain_forward(const fwo_model *m, fwo_scratch *s, const fwo_lr_entry *lr, uint32_t n_lr,
                           const fwo_ffm_entry *ffm, uint32_t n_ffm) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_num_fields;
    if (m->nn.n_layers && c->wiring == FWO_WIRING_REGRESSOR)
        return chain_nn((fwo_model *)m, s, lr, n_lr, ffm, n_ffm, 0.0f, 0.0f, 0, 0);
    if (c->wiring == FWO_WIRING_FFM_ONLY) {
        ffm_forward(m, ffm, n_ffm, s);
        return sigmoid_block(s->ffm_out, F * F, NULL, 0, 0.0f, 0.0f, NULL);
    }
    lr_forward(m, lr, n_lr, s->lr_out);
    uint32_t T = 0;
    if (c->ffm_k > 0) {
        ffm_forward(m, ffm, n_ffm, s);
        fwo_triangle_forward(s->ffm_out, F, s->tri);
        T = F * (F + 1) / 2;
    }
    return sigmoid_block(s->lr_out, c->num_combos, s->tri, T, 0.0f, 0.0f, NULL);
}

float fwo_forward_backward(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm,
                           uint32_t n_ffm, float label, float importance, int update) {
    fwo_scratch s;
    scratch_init(&s, &m->cfg);
    float p = chain_forward_backward(m, &s, lr, n_lr, ffm, n_ffm, label, importance, update);
    scratch_free(&s);
    return p;
}

float fwo_predict(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm, uint32_t n_ffm) {
    fwo_scratch s;
    scratch_init(&s, &m->cfg);
    float p = chain_forward(m, &s, lr, n_lr, ffm, n_ffm);
    scratch_free(&s);
    return p;
}

/* regressor.rs:356-379 */
static float learn_s(fwo_model *m, fwo_scratch *s, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm,
                     uint32_t n_ffm, float label, float importance, int update) {
    update = update && (importance != 0.0f);
    if (!update) return chain_forward(m, s, lr, n_lr, ffm, n_ffm);
    return chain_forward_backward(m, s, lr, n_lr, ffm, n_ffm, label, importance, 1);
}

float fwo_learn(fwo_model *m, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm, uint32_t n_ffm,
                float label, float importance, int update) {
    fwo_scratch s;
    scratch_init(&s, &m->cfg);
    float p = learn_s(m, &s, lr, n_lr, ffm, n_ffm, label, importance, update);
    scratch_free(&s);
    return p;
}

/* ------------------------------------------------------------------ translation */

#define FWO_HEADER_LEN 3u               /* parser.rs:13 */
#define FWO_IS_NOT_SINGLE_MASK (1u << 31) /* parser.rs:17 */
#define FWO_VOWPAL_FNV_PRIME 16777619u  /* feature_buffer.rs:6 */
#define FWO_CONSTANT_HASH 11650396u     /* feature_buffer.rs:8 */

uint32_t fwo_lr_hash_mask(uint32_t bit_precision) { return (uint32_t)((1ULL << bit_precision) - 1); }

/* feature_buffer.rs:141-148 */
uint32_t fwo_ffm_hash_mask(uint32_t ffm_bits, uint32_t ffm_k) {
    uint32_t bits = 0;
    while (ffm_k > (1u << bits)) bits++;
    uint32_t dimensions_mask = (1u << bits) - 1;
    return ((uint32_t)((1ULL << ffm_bits) - 1)) ^ dimensions_mask;
}

typedef struct {
    uint32_t hash;
    float value;
} hv;

/* feature_reader! (feature_buffer.rs:47-108), primitive namespaces only.
 * Appends (hash, value) pairs of namespace `ns` to out; returns count or -1 on overflow. */
static int read_ns(const uint32_t *rec, uint32_t ns, int is_f32, hv *out, uint32_t cap) {
    uint32_t first_token = rec[ns + FWO_HEADER_LEN];
    if ((first_token & FWO_IS_NOT_SINGLE_MASK) == 0) {
        if (cap < 1) return -1;
        out[0].hash = first_token;
        out[0].value = 1.0f;
        return 1;
    }
    uint32_t start = (first_token >> 16) & 0x3fff, end = first_token & 0xffff;
    int n = 0;
    for (uint32_t o = start; o < end; o += 2) {
        if ((uint32_t)n >= cap) return -1;
        out[n].hash = rec[o];
        out[n].value = is_f32 ? 1.0f : u2f(rec[o + 1]);
        n++;
    }
    return n;
}

#define FWO_TMP_CAP 4096

int fwo_translate(const fwo_translator *t, const uint32_t *rec, fwo_lr_entry *lr_out, uint32_t lr_cap, uint32_t *n_lr_out,
                  fwo_ffm_entry *ffm_out, uint32_t ffm_cap, uint32_t *n_ffm_out, float *label, float *importance) {
    const uint32_t lr_mask = fwo_lr_hash_mask(t->bit_precision);
    uint32_t n_lr = 0, n_ffm = 0;
    *label = (float)rec[1];   /* feature_buffer.rs:187 */
    *importance = u2f(rec[2]); /* 188-189 */
    hv a[FWO_TMP_CAP], b[FWO_TMP_CAP], cur[FWO_TMP_CAP];
    for (uint32_t ci = 0; ci < t->n_combos; ci++) { /* 194-267 */
        uint32_t s = t->combo_off[ci], e = t->combo_off[ci + 1];
        float cw = t->combo_weight[ci];
        int na = read_ns(rec, t->combo_ns[s], t->combo_ns_f32[s], a, FWO_TMP_CAP);
        if (na < 0) return -1;
        if (e - s == 1) {
            for (int i = 0; i < na; i++) {
                if (n_lr >= lr_cap) return -1;
                lr_out[n_lr].hash = a[i].hash & lr_mask;
                lr_out[n_lr].value = a[i].value * cw;
                lr_out[n_lr].combo_index = ci;
                n_lr++;
            }
            continue;
        }
        hv *in = a, *out = b;
        for (uint32_t mi = s + 1; mi < e; mi++) { /* 235-258 */
            int nc = read_ns(rec, t->combo_ns[mi], t->combo_ns_f32[mi], cur, FWO_TMP_CAP);
            if (nc < 0) return -1;
            int no = 0;
            for (int i = 0; i < na; i++) {
                uint32_t half_hash = in[i].hash * FWO_VOWPAL_FNV_PRIME; /* wrapping */
                for (int j = 0; j < nc; j++) {
                    if (no >= FWO_TMP_CAP) return -1;
                    out[no].hash = cur[j].hash ^ half_hash;
                    out[no].value = in[i].value * cur[j].value;
                    no++;
                }
            }
            hv *tmp = in;
            in = out;
            out = tmp;
            na = no;
        }
        for (int i = 0; i < na; i++) {
            if (n_lr >= lr_cap) return -1;
            lr_out[n_lr].hash = in[i].hash & lr_mask;
            lr_out[n_lr].value = in[i].value * cw;
            lr_out[n_lr].combo_index = ci;
            n_lr++;
        }
    }
    if (t->add_constant_feature) { /* 270-276 */
        if (n_lr >= lr_cap) return -1;
        lr_out[n_lr].hash = FWO_CONSTANT_HASH & lr_mask;
        lr_out[n_lr].value = 1.0f;
        lr_out[n_lr].combo_index = t->n_combos;
        n_lr++;
    }
    if (t->ffm_k > 0) { /* 279-335 */
        const uint32_t ffm_mask = fwo_ffm_hash_mask(t->ffm_bit_precision, t->ffm_k);
        for (uint32_t f = 0; f < t->n_fields; f++) {
            for (uint32_t mi = t->field_off[f]; mi < t->field_off[f + 1]; mi++) {
                int nc = read_ns(rec, t->field_ns[mi], t->field_ns_f32[mi], cur, FWO_TMP_CAP);
                if (nc < 0) return -1;
                for (int j = 0; j < nc; j++) {
                    if (n_ffm >= ffm_cap) return -1;
                    ffm_out[n_ffm].hash = cur[j].hash & ffm_mask;
                    ffm_out[n_ffm].value = cur[j].value;
                    ffm_out[n_ffm].contra_field_index = f * t->ffm_k;
                    n_ffm++;
                }
            }
        }
    }
    *n_lr_out = n_lr;
    *n_ffm_out = n_ffm;
    return 0;
}


/* ------------------------------------------------------------------ deep head (a18) */

static uint32_t nn_x_len(const fwo_model *m) { /* the Join span: LR slots + triangle */
    const uint32_t F = m->cfg.ffm_k ? m->cfg.ffm_num_fields : 0;
    return m->cfg.num_combos + F * (F + 1) / 2;
}

int fwo_set_nn(fwo_model *m, const fwo_nn_config *nn) {
    if (nn->n_layers == 0 || nn->n_layers > FWO_NN_MAX_LAYERS || (nn->topology != 1 && nn->topology != 2)) return -1;
    m->nn = *nn;
    uint32_t in = nn_x_len(m);
    for (uint32_t l = 0; l < nn->n_layers; l++) {
        m->nn_in[l] = in;
        m->nn_out[l] = nn->width[l];
        in = nn->width[l];
    }
    m->nn_in[nn->n_layers] = in + (nn->topology == 1 ? nn_x_len(m) : 0); /* Join(h_last, x copy), regressor.rs:307-311 */
    m->nn_out[nn->n_layers] = 1;
    for (uint32_t l = 0; l <= nn->n_layers; l++) {
        const size_t len = ((size_t)m->nn_in[l] + 1) * m->nn_out[l];
        free(m->nn_w[l]);
        free(m->nn_acc[l]);
        m->nn_w[l] = (float *)calloc(len, sizeof(float));
        m->nn_acc[l] = (float *)calloc(len, sizeof(float));
    }
    fwo_lut_init(m->lut_nn, nn->nn_learning_rate, nn->nn_power_t, nn->nn_init_acc_gradient); /* block_neural.rs:111-112 */
    return 0;
}

float *fwo_nn_weights(fwo_model *m, uint32_t layer, uint64_t *len) {
    if (len) *len = ((uint64_t)m->nn_in[layer] + 1) * m->nn_out[layer];
    return m->nn_w[layer];
}
float *fwo_nn_acc(fwo_model *m, uint32_t layer, uint64_t *len) {
    if (len) *len = ((uint64_t)m->nn_in[layer] + 1) * m->nn_out[layer];
    return m->nn_acc[layer];
}

static uint64_t nn_rng(uint64_t *s) { /* splitmix64 */
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static double nn_u01(uint64_t *s) { return ((double)(nn_rng(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

static void nn_init_weights(fwo_model *m) { /* block_neural.rs:367-424 (RNG stream: stand-in, see header) */
    const float a0 = initial_acc(m->cfg.optimizer, m->nn.nn_init_acc_gradient);
    for (uint32_t l = 0; l <= m->nn.n_layers; l++) {
        const uint32_t in = m->nn_in[l], out = m->nn_out[l];
        const size_t bias = (size_t)in * out, len = bias + out;
        const uint32_t init = l == m->nn.n_layers ? FWO_NN_INIT_ONE : m->nn.init[l];
        uint64_t st = 0x5eed0000ULL + 7919ULL * l + in + len;
        for (size_t i = 0; i < bias; i++) {
            float w;
            switch (init) {
            case FWO_NN_INIT_XAVIER: {
                const double bound = sqrt(6.0) / sqrt((double)bias);
                w = (float)((2.0 * nn_u01(&st) - 1.0) * bound);
                break;
            }
            case FWO_NN_INIT_HU: {
                const double u1 = nn_u01(&st), u2 = nn_u01(&st);
                w = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * sqrt(2.0 / (double)in));
                break;
            }
            case FWO_NN_INIT_ONE: w = 1.0f; break;
            default: w = 0.0f;
            }
            m->nn_w[l][i] = w;
        }
        for (size_t i = bias; i < len; i++) m->nn_w[l][i] = 0.0f; /* biases always zero, 419-422 */
        for (size_t i = 0; i < len; i++) m->nn_acc[l][i] = a0;
    }
}

/* block_neural.rs:196-222: y = bias; y += W x (sgemv 'T'); MKL's summation order is not observable, plain order here */
static void nn_layer_forward(const float *w, uint32_t in, uint32_t out, const float *x, float *y) {
    const float *bias = w + (size_t)in * out;
    for (uint32_t j = 0; j < out; j++) {
        float dot = 0.0f;
        const float *wj = w + (size_t)j * in;
        for (uint32_t i = 0; i < in; i++) dot += wj[i] * x[i];
        y[j] = bias[j] + dot;
    }
}

/* block_neural.rs:252-340 without dropout / max-norm / layer-norm (all off by default): per neuron j in order, per
 * input i in order: AdaGrad step on W[j][i]; output_errors[i] += W_old[j][i] * gg; then the bias; finally the input
 * tape is replaced by output_errors. */
static void nn_layer_backward(int optimizer, float lr, float power_t, const float *lut, float *w, float *acc, uint32_t in,
                              uint32_t out, float *x, const float *out_grad) {
    float *oe = (float *)calloc(in ? in : 1, sizeof(float));
    const size_t bias = (size_t)in * out;
    for (uint32_t j = 0; j < out; j++) {
        const float gg = out_grad[j] * 1.0f; /* dropout_inv == 1 */
        if (gg == 0.0f) continue;
        const size_t jo = (size_t)j * in;
        for (uint32_t i = 0; i < in; i++) {
            const float gradient = gg * x[i];
            const float update = opt_step(optimizer, lr, power_t, lut, gradient, &acc[jo + i]);
            oe[i] += w[jo + i] * gg;
            w[jo + i] -= update;
        }
        const float update = opt_step(optimizer, lr, power_t, lut, gg * 1.0f, &acc[bias + j]);
        w[bias + j] -= update;
    }
    memcpy(x, oe, sizeof(float) * in);
    free(oe);
}

void fwo_neuron_layer_fb(int optimizer, float lr, float power_t, float init_acc, float *w, float *acc, uint32_t n_in,
                         uint32_t n_out, float *x, float *y, const float *out_grad, int update) {
    float lut[FWO_LUT_SIZE];
    fwo_lut_init(lut, lr, power_t, init_acc);
    nn_layer_forward(w, n_in, n_out, x, y);
    if (update) nn_layer_backward(optimizer, lr, power_t, lut, w, acc, n_in, n_out, x, out_grad);
}

/* Whole chain with the deep head.  Tape order follows graph.rs:251-285; gradients flow as the recursion unwinds:
 * sigmoid -> final neuron -> [ReLU -> layer]* -> Copy (sums both branches) -> Triangle mirror -> FFM update, LR update. */
static float chain_nn(fwo_model *m, fwo_scratch *s, const fwo_lr_entry *lr, uint32_t n_lr, const fwo_ffm_entry *ffm,
                      uint32_t n_ffm, float label, float importance, int update, int training_numerics) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_k ? c->ffm_num_fields : 0, C = c->num_combos, T = F * (F + 1) / 2, X = C + T;
    const uint32_t L = m->nn.n_layers;
    float *x = (float *)malloc(sizeof(float) * (X + 1));
    float *xc = (float *)malloc(sizeof(float) * (X + 1)); /* BlockCopy output 1 */
    float *pre[FWO_NN_MAX_LAYERS], *post[FWO_NN_MAX_LAYERS];
    lr_forward(m, lr, n_lr, s->lr_out);
    if (F) {
        if (training_numerics) {
            ensure_grads(s, n_ffm * F * c->ffm_k);
            ffm_fb_forward(m, ffm, n_ffm, s);
        } else {
            ffm_forward(m, ffm, n_ffm, s);
        }
        fwo_triangle_forward(s->ffm_out, F, s->tri);
    }
    memcpy(x, s->lr_out, sizeof(float) * C);
    if (T) memcpy(x + C, s->tri, sizeof(float) * T);
    memcpy(xc, x, sizeof(float) * X);
    const float *h = x;
    for (uint32_t l = 0; l < L; l++) {
        pre[l] = (float *)malloc(sizeof(float) * m->nn_out[l]);
        post[l] = (float *)malloc(sizeof(float) * m->nn_out[l]);
        nn_layer_forward(m->nn_w[l], m->nn_in[l], m->nn_out[l], h, pre[l]);
        for (uint32_t j = 0; j < m->nn_out[l]; j++) { /* block_relu.rs:38-54, 93-101 */
            const float wv = pre[l][j];
            if (m->nn.relu[l]) {
                post[l][j] = wv < 0.0f ? 0.0f : wv;
                pre[l][j] = wv < 0.0f ? 0.0f : 1.0f; /* the input slot now holds the 0/1 mask */
            } else {
                post[l][j] = wv;
            }
        }
        h = post[l];
    }
    const uint32_t fin = m->nn_in[L];
    float *fx = (float *)malloc(sizeof(float) * (fin + 1));
    memcpy(fx, h, sizeof(float) * m->nn_out[L - 1]);
    if (m->nn.topology == 1) memcpy(fx + m->nn_out[L - 1], xc, sizeof(float) * X);
    float z;
    nn_layer_forward(m->nn_w[L], fin, 1, fx, &z);
    float g;
    const float p = sigmoid_block(&z, 1, NULL, 0, label, importance, &g);
    if (update) {
        const int o = c->optimizer;
        const float nlr = m->nn.nn_learning_rate, npt = m->nn.nn_power_t;
        nn_layer_backward(o, nlr, npt, m->lut_nn, m->nn_w[L], m->nn_acc[L], fin, 1, fx, &g);
        float *grad_h = fx; /* first width_last entries: gradient w.r.t. h_last */
        for (int l = (int)L - 1; l >= 0; l--) {
            float *og = (float *)malloc(sizeof(float) * m->nn_out[l]);
            for (uint32_t j = 0; j < m->nn_out[l]; j++)
                og[j] = m->nn.relu[l] ? pre[l][j] * grad_h[j] : grad_h[j]; /* block_relu.rs:105-110 */
            float *in_vec = l == 0 ? x : post[l - 1];
            nn_layer_backward(o, nlr, npt, m->lut_nn, m->nn_w[l], m->nn_acc[l], m->nn_in[l], m->nn_out[l], in_vec, og);
            free(og);
            grad_h = in_vec;
        }
        /* x now holds d/dx through the layers; BlockCopy sums the second branch (block_misc.rs:456-475) */
        if (m->nn.topology == 1)
            for (uint32_t i = 0; i < X; i++) x[i] += fx[m->nn_out[L - 1] + i];
        for (uint32_t i = 0; i < C; i++) s->lr_out[i] = x[i];
        if (F) {
            for (uint32_t i = 0; i < T; i++) s->tri[i] = x[C + i];
            fwo_triangle_backward(s->tri, F, s->ffm_out);
            ffm_fb_update(m, ffm, n_ffm, s);
        }
        lr_update(m, lr, n_lr, s->lr_out);
    }
    for (uint32_t l = 0; l < L; l++) {
        free(pre[l]);
        free(post[l]);
    }
    free(fx);
    free(x);
    free(xc);
    return p;
}

/* ------------------------------------------------------------------ stream runner */

#define FWO_EX_CAP 8192

typedef struct {
    fwo_model *m;
    const fwo_translator *t;
    const uint32_t *records;
    const uint64_t *rec_off;
    uint64_t n_train;
    volatile uint64_t *next; /* shared cursor: stands in for the mpsc queue (hogwild.rs:30-36) */
} hog_args;

static void *hog_worker(void *p) { /* hogwild.rs:89-103 */
    hog_args *a = (hog_args *)p;
    fwo_scratch s;
    scratch_init(&s, &a->m->cfg);
    fwo_lr_entry *lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * FWO_EX_CAP);
    fwo_ffm_entry *ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * FWO_EX_CAP);
    for (;;) {
        uint64_t i = __atomic_fetch_add(a->next, 1, __ATOMIC_RELAXED);
        if (i >= a->n_train) break;
        uint32_t n_lr, n_ffm;
        float label, imp;
        if (fwo_translate(a->t, a->records + a->rec_off[i], lr, FWO_EX_CAP, &n_lr, ffm, FWO_EX_CAP, &n_ffm, &label, &imp))
            continue;
        learn_s(a->m, &s, lr, n_lr, ffm, n_ffm, label, imp, 1);
    }
    free(lr);
    free(ffm);
    scratch_free(&s);
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double fwo_run_stream(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off,
                      uint64_t n, uint64_t holdout_after, int nthreads, float *preds) {
    /* main.rs:236-241: example_num starts at 1; update = example_num < holdout_after */
    uint64_t n_train = n;
    if (holdout_after > 0) n_train = holdout_after - 1 < n ? holdout_after - 1 : n;
    fwo_scratch s;
    scratch_init(&s, &m->cfg);
    fwo_lr_entry *lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * FWO_EX_CAP);
    fwo_ffm_entry *ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * FWO_EX_CAP);
    double t0 = now_s(), dt;
    uint64_t start_seq = 0;
    if (nthreads > 1) {
        volatile uint64_t next = 0;
        hog_args a = {m, t, records, rec_off, n_train, &next};
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, hog_worker, &a);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
        free(th);
        if (preds)
            for (uint64_t i = 0; i < n_train; i++) preds[i] = 0.0f; /* main.rs:242-243 */
        start_seq = n_train;
        dt = now_s() - t0;
    } else {
        for (uint64_t i = 0; i < n_train; i++) {
            uint32_t n_lr, n_ffm;
            float label, imp;
            fwo_translate(t, records + rec_off[i], lr, FWO_EX_CAP, &n_lr, ffm, FWO_EX_CAP, &n_ffm, &label, &imp);
            float p = learn_s(m, &s, lr, n_lr, ffm, n_ffm, label, imp, 1);
            if (preds) preds[i] = p;
        }
        start_seq = n_train;
        dt = now_s() - t0;
    }
    for (uint64_t i = start_seq; i < n; i++) {
        uint32_t n_lr, n_ffm;
        float label, imp;
        fwo_translate(t, records + rec_off[i], lr, FWO_EX_CAP, &n_lr, ffm, FWO_EX_CAP, &n_ffm, &label, &imp);
        float p = learn_s(m, &s, lr, n_lr, ffm, n_ffm, label, imp, 0);
        if (preds) preds[i] = p;
    }
    free(lr);
    free(ffm);
    scratch_free(&s);
    return dt;
}

/* read-only pass (update = false, main.rs:238-241) over n examples on nthreads threads: predictions do not depend on the order,
 * so the result equals the single-threaded tail of fwo_run_stream; the threads only shorten the hold-out passes of the long curves */
typedef struct {
    fwo_model *m;
    const fwo_translator *t;
    const uint32_t *records;
    const uint64_t *rec_off;
    uint64_t lo, hi;
    float *preds;
} pred_args;

static void *pred_worker(void *p) {
    pred_args *a = (pred_args *)p;
    fwo_scratch s;
    scratch_init(&s, &a->m->cfg);
    fwo_lr_entry *lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * FWO_EX_CAP);
    fwo_ffm_entry *ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * FWO_EX_CAP);
    for (uint64_t i = a->lo; i < a->hi; i++) {
        uint32_t n_lr, n_ffm;
        float label, imp;
        fwo_translate(a->t, a->records + a->rec_off[i], lr, FWO_EX_CAP, &n_lr, ffm, FWO_EX_CAP, &n_ffm, &label, &imp);
        a->preds[i] = learn_s(a->m, &s, lr, n_lr, ffm, n_ffm, label, imp, 0);
    }
    free(lr);
    free(ffm);
    scratch_free(&s);
    return NULL;
}

void fwo_predict_stream(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n,
                        int nthreads, float *preds) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    pred_args *a = (pred_args *)malloc(sizeof(pred_args) * (size_t)nthreads);
    for (int i = 0; i < nthreads; i++) {
        a[i] = (pred_args){m, t, records, rec_off, n * (uint64_t)i / (uint64_t)nthreads, n * (uint64_t)(i + 1) / (uint64_t)nthreads, preds};
        pthread_create(&th[i], NULL, pred_worker, &a[i]);
    }
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    free(a);
    free(th);
}

/* ------------------------------------------------------------------ synchronous micro-batch (see fw_oracle.h) */
typedef struct {
    fwo_lr_entry *lr;
    fwo_ffm_entry *ffm;
    uint32_t n_lr, n_ffm;
    float *G;   /* gradient cache of the frozen forward pass (block_ffm.rs:220-261) */
    float g;    /* general gradient */
    float *dx;  /* deep head: d logit / d x from the frozen dense weights, [X] */
    int update;
} mb_example;

void fwo_learn_minibatch(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n,
                         float *preds) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_k ? c->ffm_num_fields : 0, k = c->ffm_k, C = c->num_combos, T = F * (F + 1) / 2, X = C + T;
    const uint32_t L = m->nn.n_layers;
    const int head = L && c->wiring == FWO_WIRING_REGRESSOR;
    fwo_scratch s;
    scratch_init(&s, c);
    mb_example *ex = (mb_example *)calloc(n ? n : 1, sizeof(mb_example));
    fwo_lr_entry *lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * FWO_EX_CAP);
    fwo_ffm_entry *ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * FWO_EX_CAP);
    /* dense gradient sums, same layout as the weights (per layer W[j*in+i], then the biases) */
    double unused = 0.0;
    (void)unused;
    float *dW[FWO_NN_MAX_LAYERS + 1] = {0};
    if (head)
        for (uint32_t l = 0; l <= L; l++) dW[l] = (float *)calloc((size_t)(m->nn_in[l] + 1) * m->nn_out[l], sizeof(float));
    /* ---- pass 1: every example against the weights of the batch start */
    for (uint64_t e = 0; e < n; e++) {
        mb_example *x = &ex[e];
        float label, imp;
        fwo_translate(t, records + rec_off[e], lr, FWO_EX_CAP, &x->n_lr, ffm, FWO_EX_CAP, &x->n_ffm, &label, &imp);
        x->lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * (x->n_lr + 1));
        x->ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * (x->n_ffm + 1));
        memcpy(x->lr, lr, sizeof(fwo_lr_entry) * x->n_lr);
        memcpy(x->ffm, ffm, sizeof(fwo_ffm_entry) * x->n_ffm);
        x->update = imp != 0.0f; /* regressor.rs:366 */
        lr_forward(m, lr, x->n_lr, s.lr_out);
        if (F) {
            ensure_grads(&s, x->n_ffm * F * k);
            ffm_fb_forward(m, ffm, x->n_ffm, &s);
            x->G = (float *)malloc(sizeof(float) * ((size_t)x->n_ffm * F * k + 1));
            memcpy(x->G, s.grads, sizeof(float) * (size_t)x->n_ffm * F * k);
            fwo_triangle_forward(s.ffm_out, F, s.tri);
        }
        float p;
        if (!head) {
            if (c->wiring == FWO_WIRING_FFM_ONLY)
                p = sigmoid_block(s.ffm_out, F * F, NULL, 0, label, imp, &x->g);
            else
                p = sigmoid_block(s.lr_out, C, s.tri, T, label, imp, &x->g);
        } else {
            /* frozen head: forward, then backward for the input gradients and the weight-gradient sums */
            float *xin = (float *)malloc(sizeof(float) * (X + 1));
            memcpy(xin, s.lr_out, sizeof(float) * C);
            if (T) memcpy(xin + C, s.tri, sizeof(float) * T);
            float *pre[FWO_NN_MAX_LAYERS], *post[FWO_NN_MAX_LAYERS];
            const float *h = xin;
            for (uint32_t l = 0; l < L; l++) {
                pre[l] = (float *)malloc(sizeof(float) * m->nn_out[l]);
                post[l] = (float *)malloc(sizeof(float) * m->nn_out[l]);
                nn_layer_forward(m->nn_w[l], m->nn_in[l], m->nn_out[l], h, pre[l]);
                for (uint32_t j = 0; j < m->nn_out[l]; j++) {
                    const float wv = pre[l][j];
                    post[l][j] = (m->nn.relu[l] && wv < 0.0f) ? 0.0f : wv;
                    pre[l][j] = (m->nn.relu[l] && wv < 0.0f) ? 0.0f : 1.0f;
                }
                h = post[l];
            }
            const uint32_t fin = m->nn_in[L], wl = m->nn_out[L - 1];
            float *fx = (float *)malloc(sizeof(float) * (fin + 1));
            memcpy(fx, h, sizeof(float) * wl);
            if (m->nn.topology == 1) memcpy(fx + wl, xin, sizeof(float) * X);
            float z;
            nn_layer_forward(m->nn_w[L], fin, 1, fx, &z);
            p = sigmoid_block(&z, 1, NULL, 0, label, imp, &x->g);
            x->dx = (float *)calloc(X + 1, sizeof(float));
            if (x->update && x->g != 0.0f) {
                const float g = x->g;
                float *gin = (float *)calloc(fin + 1, sizeof(float)); /* final neuron */
                for (uint32_t i = 0; i < fin; i++) {
                    dW[L][i] += g * fx[i];
                    gin[i] = m->nn_w[L][i] * g;
                }
                dW[L][fin] += g;
                if (m->nn.topology == 1)
                    for (uint32_t i = 0; i < X; i++) x->dx[i] = gin[wl + i];
                float *grad_h = gin;
                float *tmp = NULL;
                for (int l = (int)L - 1; l >= 0; l--) {
                    const uint32_t in = m->nn_in[l], out = m->nn_out[l];
                    const float *in_vec = l == 0 ? xin : post[l - 1];
                    float *oe = (float *)calloc(in + 1, sizeof(float));
                    for (uint32_t j = 0; j < out; j++) {
                        const float gg = pre[l][j] * grad_h[j]; /* block_relu.rs:105-110 */
                        if (gg == 0.0f) continue;
                        const size_t jo = (size_t)j * in;
                        for (uint32_t i = 0; i < in; i++) {
                            dW[l][jo + i] += gg * in_vec[i];
                            oe[i] += m->nn_w[l][jo + i] * gg;
                        }
                        dW[l][(size_t)in * out + j] += gg;
                    }
                    free(tmp);
                    tmp = oe;
                    grad_h = oe;
                }
                for (uint32_t i = 0; i < X; i++) x->dx[i] += grad_h[i]; /* BlockCopy sums both branches */
                free(tmp);
                free(gin);
            }
            for (uint32_t l = 0; l < L; l++) {
                free(pre[l]);
                free(post[l]);
            }
            free(fx);
            free(xin);
        }
        if (preds) preds[e] = p;
    }
    /* ---- pass 2: the sparse updates, example by example */
    for (uint64_t e = 0; e < n; e++) {
        mb_example *x = &ex[e];
        if (x->update && (head || x->g != 0.0f)) {
            if (head) {
                for (uint32_t i = 0; i < C; i++) s.lr_out[i] = x->dx[i];
                for (uint32_t i = 0; i < T; i++) s.tri[i] = x->dx[C + i];
            } else {
                for (uint32_t i = 0; i < C; i++) s.lr_out[i] = x->g;
                for (uint32_t i = 0; i < T; i++) s.tri[i] = x->g;
            }
            if (F) {
                if (c->wiring == FWO_WIRING_FFM_ONLY)
                    for (uint32_t i = 0; i < F * F; i++) s.ffm_out[i] = x->g;
                else
                    fwo_triangle_backward(s.tri, F, s.ffm_out);
                ensure_grads(&s, x->n_ffm * F * k);
                memcpy(s.grads, x->G, sizeof(float) * (size_t)x->n_ffm * F * k);
                ffm_fb_update(m, x->ffm, x->n_ffm, &s);
            }
            if (c->wiring != FWO_WIRING_FFM_ONLY) lr_update(m, x->lr, x->n_lr, s.lr_out);
        }
        free(x->lr);
        free(x->ffm);
        free(x->G);
        free(x->dx);
    }
    /* ---- the dense weights: one optimizer step each with the summed gradient */
    if (head)
        for (uint32_t l = 0; l <= L; l++) {
            const size_t nw = (size_t)(m->nn_in[l] + 1) * m->nn_out[l];
            for (size_t i = 0; i < nw; i++) {
                if (dW[l][i] == 0.0f) continue;
                const float upd = opt_step(c->optimizer, m->nn.nn_learning_rate, m->nn.nn_power_t, m->lut_nn, dW[l][i], &m->nn_acc[l][i]);
                m->nn_w[l][i] -= upd;
            }
            free(dW[l]);
        }
    free(ex);
    free(lr);
    free(ffm);
    scratch_free(&s);
}

/* ---- An EMULATION of the device's concurrent mode, for analysis only (round 6).  Not a reference code path and not a parity yardstick: the
 * reference steps one example after the other (regressor.rs:356-379) or with 16 racing threads (hogwild.rs:89-103).  Here the stream is cut into windows of
 * `window` examples -- the examples the device has in flight -- and inside a window
 *   pass 1: every example is scored against the tables as they stand at the window's start (the device's gather reads a row ~one example lifetime before it
 *           writes it back), its gradient cache and general gradient kept (block_ffm.rs:219-261, block_loss_functions.rs:141);
 *   pass 2: the examples' steps are applied in order.  flags bit 0: an FFM weight is written back as (value at the window's start) - step -- the device's rows
 *           kept from the gather: the last writer of a row wins, the other holders' steps are lost; bit 1: the same for the accumulators (what write-through
 *           stores that lose their race do; without it every g^2 is counted: store policy 4's atomic adds).  The LR block steps per occurrence on the current
 *           table (its read-modify-write window on the device is a memory round trip, not an example's lifetime).
 * With window = 1 this is the sequential reference, bit for bit (rows of different hashes that overlap inside one example aside, with flags set). */
void fwo_learn_window_emulation(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n, uint32_t window,
                                uint32_t flags, float *preds) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_k ? c->ffm_num_fields : 0, k = c->ffm_k, C = c->num_combos, T = F * (F + 1) / 2, R = F * k;
    if (window < 1) window = 1;
    fwo_scratch s;
    scratch_init(&s, c);
    mb_example *ex = (mb_example *)calloc(window, sizeof(mb_example));
    float **W0 = (float **)calloc(window, sizeof(float *)), **A0 = (float **)calloc(window, sizeof(float *));
    fwo_lr_entry *lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * FWO_EX_CAP);
    fwo_ffm_entry *ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * FWO_EX_CAP);
    for (uint64_t w0 = 0; w0 < n; w0 += window) {
        const uint64_t wn = n - w0 < window ? n - w0 : window;
        for (uint64_t q = 0; q < wn; q++) { /* pass 1 */
            mb_example *x = &ex[q];
            float label, imp;
            fwo_translate(t, records + rec_off[w0 + q], lr, FWO_EX_CAP, &x->n_lr, ffm, FWO_EX_CAP, &x->n_ffm, &label, &imp);
            x->lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * (x->n_lr + 1));
            x->ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * (x->n_ffm + 1));
            memcpy(x->lr, lr, sizeof(fwo_lr_entry) * x->n_lr);
            memcpy(x->ffm, ffm, sizeof(fwo_ffm_entry) * x->n_ffm);
            x->update = imp != 0.0f;
            lr_forward(m, lr, x->n_lr, s.lr_out);
            const size_t ng = (size_t)x->n_ffm * R;
            x->G = NULL;
            W0[q] = A0[q] = NULL;
            if (F) {
                ensure_grads(&s, x->n_ffm * F * k);
                ffm_fb_forward(m, ffm, x->n_ffm, &s);
                x->G = (float *)malloc(sizeof(float) * (ng + 1));
                memcpy(x->G, s.grads, sizeof(float) * ng);
                fwo_triangle_forward(s.ffm_out, F, s.tri);
                if (flags & 1u) W0[q] = (float *)malloc(sizeof(float) * (ng + 1));
                if (flags & 2u) A0[q] = (float *)malloc(sizeof(float) * (ng + 1));
                for (uint32_t i = 0; i < x->n_ffm; i++) {
                    if (W0[q]) memcpy(W0[q] + (size_t)i * R, m->ffm_w + ffm[i].hash, sizeof(float) * R);
                    if (A0[q]) memcpy(A0[q] + (size_t)i * R, m->ffm_acc + ffm[i].hash, sizeof(float) * R);
                }
            }
            const float p = c->wiring == FWO_WIRING_FFM_ONLY ? sigmoid_block(s.ffm_out, F * F, NULL, 0, label, imp, &x->g)
                                                             : sigmoid_block(s.lr_out, C, s.tri, T, label, imp, &x->g);
            if (preds) preds[w0 + q] = p;
        }
        for (uint64_t q = 0; q < wn; q++) { /* pass 2 */
            mb_example *x = &ex[q];
            if (x->update && x->g != 0.0f) {
                if (F) {
                    size_t li = 0;
                    for (uint32_t i = 0; i < x->n_ffm; i++) {
                        uint64_t fi = x->ffm[i].hash;
                        /* a row the example holds twice is stepped twice on ONE copy, on the device as in the reference (duplicate-row chains): later occurrences run on */
                        int again = 0;
                        for (uint32_t j = 0; j < i && !again; j++) again = x->ffm[j].hash == x->ffm[i].hash;
                        const float *w0 = (W0[q] && !again) ? W0[q] + li : NULL, *a0 = (A0[q] && !again) ? A0[q] + li : NULL;
                        for (uint32_t e = 0; e < R; e++, li++, fi++) {
                            const float gradient = x->g * x->G[li]; /* block_ffm.rs:278: every pair's general gradient is g without a head */
                            float a = a0 ? a0[e] : m->ffm_acc[fi];
                            const float update = opt_step(c->optimizer, c->ffm_learning_rate, c->ffm_power_t, m->lut_ffm, gradient, &a);
                            m->ffm_acc[fi] = a;
                            m->ffm_w[fi] = (w0 ? w0[e] : m->ffm_w[fi]) - update;
                        }
                    }
                }
                if (c->wiring != FWO_WIRING_FFM_ONLY) {
                    for (uint32_t i = 0; i < C; i++) s.lr_out[i] = x->g;
                    lr_update(m, x->lr, x->n_lr, s.lr_out);
                }
            }
            free(x->lr);
            free(x->ffm);
            free(x->G);
            free(W0[q]);
            free(A0[q]);
        }
    }
    free(ex);
    free(W0);
    free(A0);
    free(lr);
    free(ffm);
    scratch_free(&s);
}

/* ---- row-sparse gradient buckets: the update rule of the library's multi-GPU "sparse" mode (fwumious_wabbit_amd/csrc/sparse.hip).
 * Not a reference code path: the reference has one shared table and steps per occurrence (hogwild.rs:24-103, block_ffm.rs:265-288).
 * Here all examples of the global batch are scored against the weights as they are; every table row (FFM row of R floats, LR
 * entry) then takes ONE optimizer step (optimizer.rs) with the gradient summed over all its occurrences.  The summation order is
 * the library's, so that the results can be compared bit for bit: per part (= rank) the occurrences are sorted by (row, example,
 * entry) and summed inside 64-element blocks of that sorted list (one bucket row per block and row); the bucket rows of a row are
 * then added in (part, bucket) order.  FFM rows are applied in ascending row order within hash / R blocks, even blocks first, then
 * odd ones (rows are R long from any start, so neighbours overlap: block_ffm.rs:92-94). */
typedef struct {
    uint32_t hash, ex, ent;
} sp_occ;
static int sp_occ_cmp(const void *a, const void *b) {
    const sp_occ *x = (const sp_occ *)a, *y = (const sp_occ *)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    if (x->ex != y->ex) return x->ex < y->ex ? -1 : 1;
    return x->ent < y->ent ? -1 : (x->ent > y->ent);
}
typedef struct {
    uint32_t hash;
    uint64_t seq; /* arrival order: part, then position */
    float *row;
} sp_bucket;
static int sp_bucket_cmp(const void *a, const void *b) {
    const sp_bucket *x = (const sp_bucket *)a, *y = (const sp_bucket *)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

void fwo_learn_sparse(fwo_model *m, const fwo_translator *t, const uint32_t *records, const uint64_t *rec_off, uint64_t n,
                      const uint64_t *part_end, uint32_t n_parts, float *preds) {
    const fwo_config *c = &m->cfg;
    const uint32_t F = c->ffm_k ? c->ffm_num_fields : 0, k = c->ffm_k, C = c->num_combos, T = F * (F + 1) / 2, R = F * k;
    const int has_lr = c->wiring != FWO_WIRING_FFM_ONLY;
    fwo_scratch s;
    scratch_init(&s, c);
    mb_example *ex = (mb_example *)calloc(n ? n : 1, sizeof(mb_example));
    fwo_lr_entry *lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * FWO_EX_CAP);
    fwo_ffm_entry *ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * FWO_EX_CAP);
    uint64_t tot_ffm = 0, tot_lr = 0;
    /* ---- pass 1: every example against the weights of the batch start; its per-entry gradient rows are kept */
    for (uint64_t e = 0; e < n; e++) {
        mb_example *x = &ex[e];
        float label, imp;
        fwo_translate(t, records + rec_off[e], lr, FWO_EX_CAP, &x->n_lr, ffm, FWO_EX_CAP, &x->n_ffm, &label, &imp);
        x->lr = (fwo_lr_entry *)malloc(sizeof(fwo_lr_entry) * (x->n_lr + 1));
        x->ffm = (fwo_ffm_entry *)malloc(sizeof(fwo_ffm_entry) * (x->n_ffm + 1));
        memcpy(x->lr, lr, sizeof(fwo_lr_entry) * x->n_lr);
        memcpy(x->ffm, ffm, sizeof(fwo_ffm_entry) * x->n_ffm);
        x->update = imp != 0.0f; /* regressor.rs:366 */
        lr_forward(m, lr, x->n_lr, s.lr_out);
        x->G = NULL;
        if (F) {
            ensure_grads(&s, x->n_ffm * F * k);
            ffm_fb_forward(m, ffm, x->n_ffm, &s);
            x->G = (float *)malloc(sizeof(float) * ((size_t)x->n_ffm * R + 1));
            memcpy(x->G, s.grads, sizeof(float) * (size_t)x->n_ffm * R);
            fwo_triangle_forward(s.ffm_out, F, s.tri);
        }
        float p;
        if (c->wiring == FWO_WIRING_FFM_ONLY)
            p = sigmoid_block(s.ffm_out, F * F, NULL, 0, label, imp, &x->g);
        else
            p = sigmoid_block(s.lr_out, C, s.tri, T, label, imp, &x->g);
        if (preds) preds[e] = p;
        if (x->update) {
            /* occurrence gradients: general gradient (uniform g without a head) x the cached feature gradient, block_ffm.rs:278 */
            for (size_t i = 0; i < (size_t)x->n_ffm * R; i++) x->G[i] = x->g * x->G[i];
            tot_ffm += x->n_ffm;
            tot_lr += x->n_lr;
        }
    }
    /* ---- pass 2: bucket rows per part */
    for (int side = 0; side < 2; side++) {
        const int is_ffm = side == 0;
        if (is_ffm ? !F : !has_lr) continue;
        const uint32_t W = is_ffm ? R : 1;
        const uint64_t tot = is_ffm ? tot_ffm : tot_lr;
        sp_occ *occ = (sp_occ *)malloc(sizeof(sp_occ) * (tot + 1));
        sp_bucket *bk = (sp_bucket *)malloc(sizeof(sp_bucket) * (tot + 1));
        uint64_t nb = 0, lo = 0;
        for (uint32_t part = 0; part < n_parts; part++) {
            const uint64_t hi = part_end[part];
            uint64_t no = 0;
            for (uint64_t e = lo; e < hi; e++) {
                mb_example *x = &ex[e];
                if (!x->update) continue;
                const uint32_t cnt = is_ffm ? x->n_ffm : x->n_lr;
                for (uint32_t i = 0; i < cnt; i++) {
                    occ[no].hash = is_ffm ? x->ffm[i].hash : x->lr[i].hash;
                    occ[no].ex = (uint32_t)e;
                    occ[no].ent = i;
                    no++;
                }
            }
            qsort(occ, no, sizeof(sp_occ), sp_occ_cmp);
            for (uint64_t j = 0; j < no;) {
                const uint64_t blk_end = (j / 64 + 1) * 64 < no ? (j / 64 + 1) * 64 : no;
                uint64_t j2 = j;
                float *row = (float *)calloc(W, sizeof(float));
                while (j2 < blk_end && occ[j2].hash == occ[j].hash) {
                    const mb_example *x = &ex[occ[j2].ex];
                    if (is_ffm) {
                        const float *g = x->G + (size_t)occ[j2].ent * R;
                        for (uint32_t q = 0; q < R; q++) row[q] = row[q] + g[q];
                    } else {
                        row[0] = row[0] + x->g * x->lr[occ[j2].ent].value; /* block_lr.rs:141 */
                    }
                    j2++;
                }
                bk[nb].hash = occ[j].hash;
                bk[nb].seq = nb;
                bk[nb].row = row;
                nb++;
                j = j2;
            }
            lo = hi;
        }
        /* ---- pass 3: one step per row */
        qsort(bk, nb, sizeof(sp_bucket), sp_bucket_cmp);
        float *Gs = (float *)malloc(sizeof(float) * W);
        for (int parity = 0; parity < (is_ffm ? 2 : 1); parity++) {
            for (uint64_t j = 0; j < nb;) {
                uint64_t j2 = j;
                while (j2 < nb && bk[j2].hash == bk[j].hash) j2++;
                const uint32_t h = bk[j].hash;
                if (!is_ffm || (int)((h / R) & 1u) == parity) {
                    for (uint32_t q = 0; q < W; q++) Gs[q] = 0.0f;
                    for (uint64_t b = j; b < j2; b++)
                        for (uint32_t q = 0; q < W; q++) Gs[q] = Gs[q] + bk[b].row[q];
                    if (is_ffm) {
                        for (uint32_t q = 0; q < R; q++) {
                            if (Gs[q] == 0.0f) continue;
                            const float upd = opt_step(c->optimizer, c->ffm_learning_rate, c->ffm_power_t, m->lut_ffm, Gs[q], &m->ffm_acc[h + q]);
                            m->ffm_w[h + q] -= upd;
                        }
                    } else if (Gs[0] != 0.0f) {
                        const float upd = opt_step(c->optimizer, c->learning_rate, c->power_t, m->lut_lr, Gs[0], &m->lr[2 * (size_t)h + 1]);
                        m->lr[2 * (size_t)h] -= upd;
                    }
                }
                j = j2;
            }
        }
        for (uint64_t b = 0; b < nb; b++) free(bk[b].row);
        free(Gs);
        free(bk);
        free(occ);
    }
    for (uint64_t e = 0; e < n; e++) {
        free(ex[e].lr);
        free(ex[e].ffm);
        free(ex[e].G);
    }
    free(ex);
    free(lr);
    free(ffm);
    scratch_free(&s);
}
