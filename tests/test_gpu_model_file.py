"""Model files on the device: save / load / hogwild_load / immutable load / quantised files, and the
end-to-end text -> cache -> train -> save -> serve chain.  Mirrors persistence.rs:206-643's tests in spirit: a regressor
saved and loaded back predicts the same."""
import os
import struct

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import capi
from fwumious_wabbit_amd import persistence as P
from fwumious_wabbit_amd.feed import RecordCache, VowpalParser, VwNamespaceMap
from helpers import make_pair, logloss, record_labels

pytestmark = pytest.mark.gpu

VW6 = "".join(f"A{i},ns{i}\n" for i in range(6))


def _trained(opt, nn=False, n=600, seed=31):
    mi, _, _ = make_pair(6, 4, 12, 12, opt, lr=0.05, ffm_lr=0.05)
    if nn:
        mi.nn_layers = [dict(width="9", activation="relu"), dict(width="5", activation="relu", init="xavier")]
    recs, off = fw.synth_records(6, 1.0, 1.1, 3000, 0.2, seed, 0, n)
    re = fw.Regressor(mi)
    b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    b.close()
    return mi, re, recs, off


def _predict_all(re, mi, recs, off):
    b = re.record_batch(fw.FeatureBufferTranslator(mi), recs, off)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, False)
    p = b.predictions().copy()
    b.close()
    return p


@pytest.mark.parametrize("opt,nn", [(fw.Optimizer.AdagradLUT, False), (fw.Optimizer.AdagradFlex, True), (fw.Optimizer.SGD, False)])
def test_save_then_load_predicts_the_same(tmp_path, opt, nn):
    vw = VwNamespaceMap(VW6)
    mi, re, recs, off = _trained(opt, nn)
    p0 = _predict_all(re, mi, recs, off)
    path = str(tmp_path / "model.fw")
    P.save_regressor_to_filename(path, mi, vw, re)
    # file = header + JSONs + exactly write_weights_to_buf's blob (regressor.rs:426-442)
    raw = open(path, "rb").read()
    blob = re.write_weights_to_buf()
    assert raw.endswith(blob) and raw[:4] == b"FWRE"
    # mutable load: resume training exactly where the first regressor is
    mi2, vw2, re2 = P.new_regressor_from_filename(path, immutable=False)
    assert mi2 == mi and vw2.to_json() == vw.to_json()
    assert re2.write_weights_to_buf() == blob
    assert np.array_equal(_predict_all(re2, mi2, recs, off), p0)
    more, moff = fw.synth_records(6, 1.0, 1.1, 3000, 0.2, 99, 0, 100)
    for r_, m_ in ((re, mi), (re2, mi2)):
        b = r_.record_batch(fw.FeatureBufferTranslator(m_), more, moff)
        r_.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        b.close()
    assert re2.write_weights_to_buf() == re.write_weights_to_buf()
    # immutable load: SGD regressor holding only the weights, same predictions (persistence.rs:159-172)
    P.save_regressor_to_filename(path, mi, vw, re)
    p1 = _predict_all(re, mi, recs, off)
    mi3, _, re3 = P.new_regressor_from_filename(path, immutable=True)
    assert mi3.optimizer == fw.Optimizer.SGD
    assert np.array_equal(_predict_all(re3, mi3, recs, off), p1)
    # --convert_inference_regressor output loads to the same predictor and is what the immutable regressor would write
    inf = str(tmp_path / "inference.fw")
    P.convert_inference_regressor(path, inf)
    mi4, _, re4 = P.new_regressor_from_filename(inf, immutable=True)
    assert np.array_equal(_predict_all(re4, mi4, recs, off), p1)
    assert open(inf, "rb").read().endswith(re3.write_weights_to_buf())
    if opt != fw.Optimizer.SGD:
        # an inference file says optimizer SGD: loaded mutable it is an SGD regressor; it carries no optimizer state, so
        # it cannot be poured into a live AdaGrad regressor
        mi5, _, re5 = P.new_regressor_from_filename(inf, immutable=False)
        assert mi5.optimizer == fw.Optimizer.SGD
        re5.close()
        with pytest.raises(capi.FwgpuError):
            P.hogwild_load(re, inf)
    for x in (re, re2, re3, re4):
        x.close()


def test_hogwild_load_overwrites_a_live_regressor(tmp_path):
    vw = VwNamespaceMap(VW6)
    mi, re, recs, off = _trained(fw.Optimizer.AdagradLUT, seed=41)
    path = str(tmp_path / "model.fw")
    P.save_regressor_to_filename(path, mi, vw, re)
    fresh = fw.Regressor(mi)
    assert not np.array_equal(_predict_all(fresh, mi, recs, off), _predict_all(re, mi, recs, off))
    P.hogwild_load(fresh, path)  # persistence.rs:176-187
    assert fresh.write_weights_to_buf() == re.write_weights_to_buf()
    # an immutable (SGD) server regressor takes the weights of a training file
    mi_s = fw.ModelInstance(**{**mi.__dict__, "optimizer": fw.Optimizer.SGD})
    server = fw.Regressor(mi_s)
    P.hogwild_load(server, path)
    assert np.array_equal(_predict_all(server, mi_s, recs, off), _predict_all(re, mi, recs, off))
    # a file of another shape is refused
    other, _, _ = make_pair(6, 4, 13, 12, fw.Optimizer.AdagradLUT)
    ro = fw.Regressor(other)
    P.save_regressor_to_filename(path, other, vw, ro)
    with pytest.raises(capi.FwgpuError):
        P.hogwild_load(fresh, path)
    for x in (re, fresh, server, ro):
        x.close()


def test_quantised_model_file_round_trip(tmp_path):
    vw = VwNamespaceMap(VW6)
    mi, re, recs, off = _trained(fw.Optimizer.AdagradLUT, seed=51)
    p0 = _predict_all(re, mi, recs, off)
    path = str(tmp_path / "q.fw")
    # main.rs:141-147: --convert_inference_regressor --weight_quantization
    P.save_regressor_to_filename(str(tmp_path / "t.fw"), mi, vw, re)
    P.convert_inference_regressor(str(tmp_path / "t.fw"), path, quantize_weights=True)
    assert os.path.getsize(path) < os.path.getsize(str(tmp_path / "t.fw")) / 2
    mi2, _, re2 = P.new_regressor_from_filename(path, immutable=True)
    w0 = re.table_read(capi.TABLE_FFM_W)
    w1 = re2.table_read(capi.TABLE_FFM_W)
    inc = (np.round(w0.max() * 1e4) / 1e4 - np.round(w0.min() * 1e4) / 1e4) / 65025.0
    assert np.abs(w1 - w0).max() <= 20 * inc and not np.array_equal(w1, w0)
    assert np.array_equal(w1, P.dequantize_ffm_weights(P.quantize_ffm_weights(w0), w0.size))
    assert np.abs(_predict_all(re2, mi2, recs, off) - p0).max() < 2e-3
    # a training file saved with quantised FFM weights (main.rs:283-286 --weight_quantization) keeps its optimizer state
    mi_q = fw.ModelInstance(**mi.__dict__)
    P.save_regressor_to_filename(path, mi_q, vw, re, quantize_weights=True, extra={"dequantize_weights": True})
    mi3, _, re3 = P.new_regressor_from_filename(path, immutable=False)
    assert np.array_equal(re3.table_read(capi.TABLE_FFM_ACC), re.table_read(capi.TABLE_FFM_ACC))
    assert np.array_equal(re3.table_read(capi.TABLE_LR), re.table_read(capi.TABLE_LR))
    assert np.array_equal(re3.table_read(capi.TABLE_FFM_W), w1)
    for x in (re, re2, re3):
        x.close()


def test_text_to_cache_to_training_to_serving_chain(tmp_path):
    """config A's shape end to end (examples/ffm/run.sh:16-18): .vw text -> parser -> .fwcache -> trainer -> model file ->
    immutable regressor -> predictions, against the oracle fed with the same records"""
    from oracle import fwo
    rng = np.random.default_rng(12)
    vw = VwNamespaceMap("A,animal\nB,food\n")
    animals, foods = [f"a{i}" for i in range(40)], [f"f{i}" for i in range(40)]
    likes = rng.random((40, 40)) < 0.5
    lines = []
    for _ in range(3000):
        a, f = rng.integers(0, 40), rng.integers(0, 40)
        lines.append(f"{1 if likes[a, f] else -1} |A {animals[a]} |B {foods[f]}\n")
    inp = str(tmp_path / "train.vw")
    open(inp, "w").write("".join(lines))
    # pass 1: parse the text, write the cache (main.rs:213-270 with -c)
    parser = VowpalParser(vw)
    cache = RecordCache(inp, True, vw)
    assert cache.writing
    words, off, used, rc = parser.parse_buffer(open(inp, "rb").read())
    assert rc == capi.OK and len(off) == 3001
    cache.push_records(words)
    cache.write_finish()
    cache.close()
    # pass 2: the cache alone feeds the trainer
    mi = fw.ModelInstance(learning_rate=0.1, ffm_learning_rate=0.1, power_t=0.0, ffm_power_t=0.0, bit_precision=16,
                          ffm_k=10, ffm_bit_precision=16, add_constant_feature=False, init_acc_gradient=1.0,
                          ffm_init_acc_gradient=1.0, optimizer=fw.Optimizer.AdagradLUT,
                          feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(0)]), fw.FeatureComboDesc([fw.NamespaceDescriptor(1)]),
                                               fw.FeatureComboDesc([fw.NamespaceDescriptor(0), fw.NamespaceDescriptor(1)])],
                          ffm_fields=[[fw.NamespaceDescriptor(0)], [fw.NamespaceDescriptor(1)]])
    cache = RecordCache(inp, True, vw)
    assert cache.reading
    re = fw.Regressor(mi)
    fbt = fw.FeatureBufferTranslator(mi)
    preds, all_words = [], []
    while True:
        w, o = cache.next_records(words_cap=8192, max_records=700)
        if len(o) <= 1:
            break
        b = re.record_batch(fbt, w, o)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)  # single-thread reference semantics
        preds.append(b.predictions().copy())
        all_words.append(w)
        b.close()
    preds = np.concatenate(preds)
    assert np.array_equal(np.concatenate(all_words), words)
    # oracle on the same records
    ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=0.1, ffm_learning_rate=0.1, power_t=0.0, ffm_power_t=0.0,
                           init_acc_gradient=1.0, ffm_init_acc_gradient=1.0, bit_precision=16, num_combos=3, ffm_k=10,
                           ffm_bit_precision=16, ffm_num_fields=2)
    ots = fwo.TranslatorSpec([([(0, False)], 1.0), ([(1, False)], 1.0), ([(0, False), (1, False)], 1.0)],
                             [[(0, False)], [(1, False)]], False, 16, 10, 16)
    om = fwo.Model(ocfg)
    _, p_ref = om.run_stream(ots, words, off, holdout_after=0, nthreads=1)
    y = record_labels(words, off)
    assert np.abs(logloss(preds, y) - logloss(p_ref, y)).max() < 1e-4
    assert logloss(preds[-500:], y[-500:]).mean() < 0.67 < logloss(preds[:300], y[:300]).mean()  # it is learning the pairing
    # save, convert, serve: the immutable regressor answers like the trained one
    model = str(tmp_path / "model.fw")
    P.save_regressor_to_filename(model, mi, vw, re)
    P.convert_inference_regressor(model, model + ".inference")
    mi_s, vw_s, server = P.new_regressor_from_filename(model + ".inference", immutable=True)
    sp = VowpalParser(vw_s)
    fbt_s = fw.FeatureBufferTranslator(mi_s)
    for line in lines[:20]:
        rec = sp.next_vowpal(("|" + line.split("|", 1)[1]).encode())  # serving requests carry no label
        assert abs(server.predict(fbt_s.translate(rec)) - re.predict(fbt.translate(rec))) < 1e-7
    re.close()
    server.close()


def test_serving_ffi_matches_the_regressor(tmp_path):
    """lib.rs:150-236 through the exported reference symbols: fw_predict == Regressor::predict on the parsed line;
    setup_cache + predict_with_cache == predicting the whole line (block_ffm.rs:1325-1447 test_ffm_k1_with_cache and
    friends assert exactly this equality); the batched call == the single calls"""
    from fwumious_wabbit_amd.serving import Predictor
    vw = VwNamespaceMap(VW6)
    mi, re, recs, off = _trained(fw.Optimizer.AdagradLUT, seed=61)
    path = str(tmp_path / "model.fw")
    P.save_regressor_to_filename(path, mi, vw, re)
    pr = Predictor(f"fw -i {path} -t --foreground")
    parser = VowpalParser(vw)
    fbt = fw.FeatureBufferTranslator(mi)
    rng = np.random.default_rng(5)

    def feats(ns):
        return f"|A{ns} " + " ".join(f"{rng.integers(0, 3000)}" + (f":{rng.random() * 2:.3f}" if rng.random() < 0.3 else "")
                                      for _ in range(rng.integers(1, 4)))

    lines = []
    for _ in range(64):
        lines.append(" ".join(feats(ns) for ns in rng.permutation(6)[: rng.integers(2, 7)]) + "\n")
    want = np.array([re.predict(fbt.translate(parser.next_vowpal(l.encode()))) for l in lines], dtype=np.float32)
    got = np.array([pr.predict(l) for l in lines], dtype=np.float32)
    assert np.array_equal(got, want)
    assert np.array_equal(pr.predict_batch(lines), want)
    # context + candidates: the context namespaces first, each candidate adds the rest
    ctx = "|A0 17 23:0.5 |A1 99 "
    cands = [f"|A2 {i} |A3 {i * 7}:1.5 |A5 {i % 3}\n" for i in range(40)]
    assert pr.setup_cache(ctx + "\n") == 0.0
    with_cache = np.array([pr.predict_with_cache(c) for c in cands], dtype=np.float32)
    whole = np.array([pr.predict(ctx + c) for c in cands], dtype=np.float32)
    # the context's field sums now come from the device-side cache: same numbers up to the order of the f32 sums
    # (the reference's own *_with_cache tests use assert_epsilon!, 5e-6: block_helpers.rs:30-40)
    assert np.abs(with_cache - whole).max() < 2e-6
    assert np.abs(pr.predict_batch(cands, with_cache=True) - whole).max() < 2e-6
    # ... and a THIRD party: the CPU oracle's Regressor::predict (regressor.rs:381-395, block_ffm.rs:316-440 numerics) of the whole
    # line on the same weights -- the cached route, the whole-line route and the batched route against the reference algorithm
    # itself (assert_epsilon! 5e-6, block_helpers.rs:30-40), not only against each other
    from oracle import fwo
    _, ocfg, _ = make_pair(6, 4, 12, 12, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    om = fwo.Model(ocfg)
    om.lr_table[:] = np.asarray(re.table_read(capi.TABLE_LR))
    om.ffm_weights[:] = np.asarray(re.table_read(capi.TABLE_FFM_W))[: len(om.ffm_weights)]
    om.ffm_acc[:] = np.asarray(re.table_read(capi.TABLE_FFM_ACC))[: len(om.ffm_acc)]

    def oracle_predict(line):
        fb = fbt.translate(parser.next_vowpal(line.encode()))
        return om.predict(np.asarray(fb.lr_buffer), np.asarray(fb.ffm_buffer))

    oracle = np.array([oracle_predict(ctx + c) for c in cands], dtype=np.float32)
    assert np.abs(with_cache - oracle).max() < 5e-6 and np.abs(whole - oracle).max() < 5e-6
    assert np.abs(pr.predict_batch(cands, with_cache=True) - oracle).max() < 5e-6
    assert np.abs(got - np.array([oracle_predict(l) for l in lines], dtype=np.float32)).max() < 5e-6
    assert np.array_equal(pr.predict_batch(cands, with_cache=True), with_cache)  # batched == single, same route
    # the batched call sends records and lets the kernel's translation skip the cached namespaces; the entry route
    # (host translation + features_present filter, what the single call does) gives the same numbers
    os.environ["FWGPU_SERVING_ENTRY_ROUTE"] = "1"
    try:
        assert np.array_equal(pr.predict_batch(cands, with_cache=True), with_cache)
    finally:
        del os.environ["FWGPU_SERVING_ENTRY_ROUTE"]
    # by default the records hold only what a candidate adds to the context's record (the device keeps that one); whole
    # context + candidate records are the same numbers
    os.environ["FWGPU_SERVING_MERGED_RECORDS"] = "1"
    try:
        assert np.array_equal(pr.predict_batch(cands, with_cache=True), with_cache)
    finally:
        del os.environ["FWGPU_SERVING_MERGED_RECORDS"]
    big = [f"|A2 {i} {i + 1}:0.5 |A3 {i * 7}:1.5 |A4 {i % 11} |A5 {i % 3}\n" for i in range(3000)]  # several parser threads
    assert np.array_equal(pr.predict_batch(big, with_cache=True), np.array([pr.predict_with_cache(c) for c in big], dtype=np.float32))
    assert np.abs(pr.predict_batch(big, with_cache=True) - pr.predict_batch([ctx + c for c in big])).max() < 2e-6
    # requests the record route must hand to the entry route: a candidate that names a context namespace again (its
    # features replace the context's in the record while the cache still holds the context's, parser.rs:318-326) ...
    again = cands[:5] + ["|A1 5 |A2 3\n"] + cands[5:10]
    assert np.array_equal(pr.predict_batch(again, with_cache=True), np.array([pr.predict_with_cache(c) for c in again], dtype=np.float32))
    # ... and a context whose last token the candidates continue ("|A1 9" + "9 |A2 ..." is the feature 99)
    assert pr.setup_cache("|A0 17 23:0.5 |A1 9\n") == 0.0  # (the cached text ends before the newline, lib.rs:64-67)
    glued = [f"9 |A2 {i} |A3 {i * 7}:1.5\n" for i in range(12)] + [" |A2 1\n", "|A2 1\n"]
    single = np.array([pr.predict_with_cache(c) for c in glued], dtype=np.float32)
    assert np.array_equal(pr.predict_batch(glued, with_cache=True), single)
    assert abs(pr.predict_with_cache(glued[-2]) - pr.predict("|A0 17 23:0.5 |A1 9 |A2 1\n")) < 2e-6
    assert pr.setup_cache(ctx + "\n") == 0.0
    assert np.array_equal(pr.predict_batch(cands, with_cache=True), with_cache)
    # a clone shares the weights, has its own (empty) cache
    cl = pr.clone_lite()
    assert cl.predict(lines[0]) == want[0] and cl.predict_with_cache(cands[0]) == pr.predict(cands[0])
    # lib.rs:187-190: "it is safe to use multiple threads, each accessing only one predictor"
    import threading
    clones = [pr.clone_lite() for _ in range(4)]
    got_mt = [None] * 4

    def serve(k):
        got_mt[k] = np.array([clones[k].predict(l) for l in lines], dtype=np.float32)

    threads = [threading.Thread(target=serve, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(np.array_equal(g, want) for g in got_mt)
    for c in clones:
        c.close()
    # error codes (lib.rs:47-48): EOF and unparsable lines give -1.0; the batch marks only the bad entries
    assert pr.predict("") == -1.0 and pr.predict("|UNKNOWN x\n") == -1.0 and pr.setup_cache("") == -1.0
    mixed = pr.predict_batch([lines[0], "|UNKNOWN x\n", lines[1]])
    assert mixed[0] == want[0] and mixed[1] == -1.0 and mixed[2] == want[1]
    with pytest.raises(capi.FwgpuError):
        Predictor("fw -t")  # "Cannot resolve input weights file name"
    with pytest.raises(capi.FwgpuError):
        Predictor(f"fw -i {path}.missing")
    for x in (cl, pr, re):
        x.close()


def test_trainer_digests_a_cache_file_natively(tmp_path):
    """fwgpu_trainer_digest_cache: the example loop over a .fwcache (main.rs:213-270) without leaving native code"""
    vw = VwNamespaceMap(VW6)
    mi, ocfg, ots = make_pair(6, 4, 16, 16, fw.Optimizer.AdagradLUT, lr=0.05, ffm_lr=0.05)
    n_train, n_hold = 30000, 3000
    recs, off = fw.synth_records(6, 1.0, 1.1, 20000, 0.2, 71, 0, n_train + n_hold)
    y = record_labels(recs, off)
    for gz in (False, True):
        inp = str(tmp_path / ("s.vw.gz" if gz else "s.vw"))
        c = RecordCache(inp, True, vw)
        c.push_records(recs[: int(off[n_train])])
        c.write_finish()
        c.close()
        re = fw.Regressor(mi)
        tr = fw.HogwildTrainer(re, mi, micro_batch=2048)
        c = RecordCache(inp, True, vw)
        assert tr.digest_cache(c, max_records=10000) == 10000  # a bounded slice first ...
        assert tr.digest_cache(c) == n_train - 10000           # ... then the rest of the file
        assert tr.digest_cache(c) == 0                         # end of file
        tr.block_until_workers_finished()
        assert tr.examples_seen() == n_train
        hb = re.record_batch(fw.FeatureBufferTranslator(mi), recs[int(off[n_train]):], off[n_train:] - off[n_train])
        re.learn_batch(hb, capi.MODE_HOGWILD, False)
        gpu_hold = float(logloss(hb.predictions(), y[n_train:]).mean())
        from oracle import fwo
        om = fwo.Model(ocfg)
        om.run_stream(ots, recs[: int(off[n_train])], off[: n_train + 1], nthreads=1, want_preds=False)
        _, p = om.run_stream(ots, recs[int(off[n_train]):], off[n_train:] - off[n_train], holdout_after=1, nthreads=1)
        ref_hold = float(logloss(p, y[n_train:]).mean())
        assert abs(gpu_hold - ref_hold) < 0.03, (gpu_hold, ref_hold)
        for x in (c, tr, re, hb):
            x.close()


def test_fw_predict_many_calls_reuse_the_mapped_request_buffer(tmp_path):
    """fw_predict keeps the request in device-mapped host memory and takes a fresh work counter from a ring of 1024 per launch:
    2500 calls (the ring wraps twice) alternating between lines of different lengths all return their own prediction"""
    from fwumious_wabbit_amd.serving import Predictor
    vw = VwNamespaceMap(VW6)
    mi, re, recs, off = _trained(fw.Optimizer.AdagradLUT, seed=63)
    path = str(tmp_path / "model.fw")
    P.save_regressor_to_filename(path, mi, vw, re)
    pr = Predictor(f"fw -i {path} -t --foreground")
    lines = ["|A0 1 2 |A1 5\n", "|A2 7 |A3 9:0.5 |A0 3 |A5 1 2 3 4 5 6 7\n", "|A4 11\n"]
    want = pr.predict_batch(lines)
    assert len(set(want.tolist())) == 3
    for i in range(2500):
        assert pr.predict(lines[i % 3]) == want[i % 3], i
    pr.close()
