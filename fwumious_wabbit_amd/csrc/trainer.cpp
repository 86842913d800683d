// HogwildTrainer replacement (hogwild.rs:13-103): records are translated on the host, packed into
// micro-batches and run on the device in HOGWILD mode on an internal stream.  Two device batches are
// used alternately so that translating/uploading micro-batch i+1 overlaps the kernel of micro-batch i.
#include <string.h>

#include <algorithm>
#include <memory>

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
    HostBatch hb[2];
    fwgpu_batch *dev[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool in_flight[2] = {false, false};
    int cur = 0;
    hipStream_t stream = nullptr;
    uint64_t seen = 0;
    std::vector<fwgpu_lr_entry> lr;
    std::vector<fwgpu_ffm_entry> ffm;
};

static int flush(fwgpu_trainer *tr) {
    const int c = tr->cur;
    HostBatch &hb = tr->hb[c];
    const uint32_t n = hb.size();
    if (n == 0) return FWGPU_OK;
    fwgpu_regressor *r = tr->r;
    FWGPU_HIP(hipSetDevice(r->device));
    // device batch c may still be read by its previous kernel
    if (tr->in_flight[c]) {
        FWGPU_HIP(hipEventSynchronize(tr->done[c]));
        tr->in_flight[c] = false;
    }
    fwgpu_batch *b = tr->dev[c];
    if (!b || b->n < n || b->n_lr < hb.lr_hash.size() || b->n_ffm < hb.ffm_hash.size()) {
        if (b) fwgpu_batch_free(b);
        tr->dev[c] = nullptr;
        int rc = batch_alloc(r, std::max(n, tr->micro_batch), hb.lr_hash.size() * 5 / 4 + 1024,
                             hb.ffm_hash.size() * 5 / 4 + 1024, &tr->dev[c]);
        if (rc) return rc;
        b = tr->dev[c];
    }
    int rc = batch_upload(b, hb, tr->stream);
    if (rc) return rc;
    const uint32_t cap = b->n;
    b->n = n;  // run only the filled part
    rc = fwgpu_learn_batch(r, b, FWGPU_MODE_HOGWILD, 1, tr->stream);
    b->n = cap;
    if (rc) return rc;
    FWGPU_HIP(hipEventRecord(tr->done[c], tr->stream));
    tr->in_flight[c] = true;
    // hipMemcpyAsync from pageable vectors has returned => the host staging can be reused
    hb.clear();
    tr->cur ^= 1;
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
    tr->hb[0].clear();
    tr->hb[1].clear();
    FWGPU_HIP(hipSetDevice(r->device));
    FWGPU_HIP(hipStreamCreateWithFlags(&tr->stream, hipStreamNonBlocking));
    FWGPU_HIP(hipEventCreateWithFlags(&tr->done[0], hipEventDisableTiming));
    FWGPU_HIP(hipEventCreateWithFlags(&tr->done[1], hipEventDisableTiming));
    *out = tr.release();
    return FWGPU_OK;
}

int fwgpu_digest_records(fwgpu_trainer *tr, const uint32_t *records, const uint64_t *rec_off, uint32_t n) {
    if (!tr || (n && (!records || !rec_off))) return fail(FWGPU_ERR_INVALID, "NULL argument");
    for (uint32_t i = 0; i < n; i++) {
        float label, imp;
        const uint32_t len = (uint32_t)(rec_off[i + 1] - rec_off[i]);
        int rc = translate_record(&tr->t, records + rec_off[i], len, tr->lr, tr->ffm, &label, &imp);
        if (rc) return rc;
        HostBatch &hb = tr->hb[tr->cur];
        rc = append_example(tr->r, hb, tr->lr.data(), (uint32_t)tr->lr.size(), tr->ffm.data(), (uint32_t)tr->ffm.size(),
                            label, imp);
        if (rc) return rc;
        tr->seen++;
        if (hb.size() >= tr->micro_batch) {
            rc = flush(tr);
            if (rc) return rc;
        }
    }
    return FWGPU_OK;
}

int fwgpu_finish(fwgpu_trainer *tr) {
    if (!tr) return fail(FWGPU_ERR_INVALID, "NULL trainer");
    int rc = flush(tr);
    if (rc) return rc;
    FWGPU_HIP(hipSetDevice(tr->r->device));
    FWGPU_HIP(hipStreamSynchronize(tr->stream));
    tr->in_flight[0] = tr->in_flight[1] = false;
    return FWGPU_OK;
}

int fwgpu_trainer_free(fwgpu_trainer *tr) {
    if (!tr) return FWGPU_OK;
    (void)hipSetDevice(tr->r->device);
    if (tr->stream) (void)hipStreamSynchronize(tr->stream);
    for (int i = 0; i < 2; i++) {
        if (tr->dev[i]) fwgpu_batch_free(tr->dev[i]);
        if (tr->done[i]) (void)hipEventDestroy(tr->done[i]);
    }
    if (tr->stream) (void)hipStreamDestroy(tr->stream);
    delete tr;
    return FWGPU_OK;
}

int fwgpu_trainer_examples_seen(const fwgpu_trainer *tr, uint64_t *n) {
    if (!tr || !n) return fail(FWGPU_ERR_INVALID, "NULL argument");
    *n = tr->seen;
    return FWGPU_OK;
}

}  // extern "C"
