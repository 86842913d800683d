"""BASELINE config A end to end on the device, on the reference's own example data (examples/ffm: generate.py output committed
under tests/golden/ffm_example, see make_ffm_example_data.py) with the reference's own command line (examples/ffm/run.sh:16-18)
and the acceptance checks of its CI-style script (examples/ffm/run_fw_with_prediction_tests.sh):
  * predictions of the converted inference weights == predictions of the full weights, line by line (l.121-129),
  * predictions are not all the same (l.131-151),
  * balanced accuracy on the "hard" test set (unseen animal/food combinations, needs the factorisation) > 0.80 (l.26, 244-249),
plus what the reference cannot check about itself: every training prediction against the sequential CPU oracle."""
import gzip
import os

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import capi
from fwumious_wabbit_amd import persistence as P
from fwumious_wabbit_amd.feed import RecordCache, VowpalParser, VwNamespaceMap
from helpers import logloss, record_labels

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ffm_example")


def _unpack(name, tmp_path):
    dst = str(tmp_path / name)
    with gzip.open(os.path.join(DATA, name + ".gz"), "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


def _model_instance(vw):
    """--keep A --keep B --interactions AB --ffm_k 10 --ffm_field A --ffm_field B -l 0.1 -b 25 -c --adaptive --sgd
    --power_t 0.0 --noconstant  ->  model_instance.rs:330-492 (AdagradLUT; FFM inherits lr / power_t / init_acc_gradient)"""
    a, b = vw.lookup("A")[0], vw.lookup("B")[0]
    nd = fw.NamespaceDescriptor
    return fw.ModelInstance(learning_rate=0.1, ffm_learning_rate=0.1, power_t=0.0, ffm_power_t=0.0, bit_precision=25,
                            ffm_k=10, ffm_bit_precision=18, add_constant_feature=False, init_acc_gradient=1.0,
                            ffm_init_acc_gradient=1.0, optimizer=fw.Optimizer.AdagradLUT,
                            feature_combo_descs=[fw.FeatureComboDesc([nd(a)]), fw.FeatureComboDesc([nd(b)]),
                                                 fw.FeatureComboDesc([nd(a), nd(b)])],
                            ffm_fields=[[nd(a)], [nd(b)]])


def _records_of(path, vw, use_cache):
    """main.rs:213-270 input side: the text parser, or the cache when `-c` finds one"""
    cache = RecordCache(path, use_cache, vw)
    if cache.reading:
        out_w, out_o = [], [np.zeros(1, dtype=np.uint64)]
        base = 0
        while True:
            w, o = cache.next_records()
            if len(o) <= 1:
                break
            out_w.append(w)
            out_o.append(o[1:] + np.uint64(base))
            base += len(w)
        cache.close()
        return np.concatenate(out_w), np.concatenate(out_o), True
    words, off, used, rc = VowpalParser(vw).parse_buffer(open(path, "rb").read())
    assert rc == capi.OK and used == os.path.getsize(path)
    cache.push_records(words)
    cache.write_finish()
    cache.close()
    return words, off, False


def _predict(re, mi, words, off):
    b = re.record_batch(fw.FeatureBufferTranslator(mi), words, off)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, False)
    p = b.predictions().copy()
    b.close()
    return p


def _balanced_accuracy(p, y, threshold=0.5):
    pos, neg = y == 1, y == 0
    return 0.5 * ((p[pos] >= threshold).mean() + (p[neg] < threshold).mean())


def test_examples_ffm_end_to_end(tmp_path):
    from oracle import fwo
    vw = VwNamespaceMap(gzip.open(os.path.join(DATA, "vw_namespace_map.csv.gz"), "rt").read())
    train = _unpack("train.vw", tmp_path)
    mi = _model_instance(vw)
    # ---- training pass: text -> records (-c writes the cache), single-thread semantics, predictions like `-p`
    words, off, from_cache = _records_of(train, vw, True)
    assert not from_cache and len(off) == 30001
    re = fw.Regressor(mi)
    b = re.record_batch(fw.FeatureBufferTranslator(mi), words, off)
    re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
    p_train = b.predictions().copy()
    b.close()
    y = record_labels(words, off)
    # the sequential CPU oracle on the same records
    ocfg = fwo.make_config(optimizer=fwo.OPT_ADAGRAD_LUT, learning_rate=0.1, ffm_learning_rate=0.1, power_t=0.0, ffm_power_t=0.0,
                           init_acc_gradient=1.0, ffm_init_acc_gradient=1.0, bit_precision=25, num_combos=3, ffm_k=10,
                           ffm_bit_precision=18, ffm_num_fields=2)
    ots = fwo.TranslatorSpec([([(0, False)], 1.0), ([(1, False)], 1.0), ([(0, False), (1, False)], 1.0)],
                             [[(0, False)], [(1, False)]], False, 25, 10, 18)
    _, p_ref = fwo.Model(ocfg).run_stream(ots, words, off, holdout_after=0, nthreads=1)
    assert np.abs(logloss(p_train, y) - logloss(p_ref, y)).max() < 1e-4
    # ---- save (--save_resume -f), convert (--convert_inference_regressor)
    full = str(tmp_path / "full_weights.fw.model")
    inf = str(tmp_path / "inference_weights.fw.model")
    P.save_regressor_to_filename(full, mi, vw, re)
    P.convert_inference_regressor(full, inf)
    # ---- `-t` passes over the training data: second open finds the cache (ignoring the text, cache.rs:93-95)
    words2, off2, from_cache = _records_of(train, vw, True)
    assert from_cache and np.array_equal(words2, words) and np.array_equal(off2, off)
    mi_full, vw_full, re_full = P.new_regressor_from_filename(full, immutable=True)  # testonly -> immutable (main.rs:159)
    mi_inf, _, re_inf = P.new_regressor_from_filename(inf, immutable=True)
    assert vw_full.to_json() == vw.to_json()
    p_full = _predict(re_full, mi_full, words, off)
    p_inf = _predict(re_inf, mi_inf, words, off)
    # 1. "inference weights produce different predictions to full weights!" must not happen
    assert np.array_equal(p_full, p_inf)
    # 2. "all predictions are the same" must not happen, neither during training nor in the -t passes
    assert len(np.unique(p_full)) > 100 and len(np.unique(p_train)) > 100
    # 3. the learner is far better than a coin on its own training data
    assert _balanced_accuracy(p_full, y) > 0.95
    # ---- the hard test set: combinations never seen in training, only the latent factors can get them right
    hard = _unpack("test-hard.vw", tmp_path)
    hw, ho, _ = _records_of(hard, vw, False)
    hy = record_labels(hw, ho)
    p_hard = _predict(re_inf, mi_inf, hw, ho)
    ba = _balanced_accuracy(p_hard, hy)
    assert ba > 0.80, ba  # MARGIN_OF_PERFORMANCE_HARD_TEST_BA
    easy = _unpack("test-easy.vw", tmp_path)
    ew, eo, _ = _records_of(easy, vw, False)
    assert _balanced_accuracy(_predict(re_inf, mi_inf, ew, eo), record_labels(ew, eo)) > 0.95
    for x in (re, re_full, re_inf):
        x.close()


BASIC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_basic")
KEEPS = "B C D F G H L O S U W e f g h i o p q r v x".split()
INTERACTIONS = "4G 4GHX 4GUW 4K 4c 4go 4v BC BD BGO BX CO DG DW GU Gx KR MN UW Ug eg".split()


@pytest.mark.parametrize("name,interactions,power_t,optimizer", [
    ("basic", INTERACTIONS, 0.39, fw.Optimizer.AdagradLUT),           # examples/basic/run.sh:12-13
    ("vw-compatibility", [], 0.35, fw.Optimizer.AdagradFlex)])        # examples/vw-compatibility/run.sh:15-16, --vwcompat
def test_reference_example_datasets_lr(tmp_path, name, interactions, power_t, optimizer):
    """The reference's production-like example lines (58 namespaces, 2/3/4-way interactions, namespace weights, several
    features per namespace) through parser -> device translation -> LR learner, against the oracle and the host translator"""
    from oracle import fwo
    vw = VwNamespaceMap(gzip.open(os.path.join(BASIC, "vw_namespace_map.csv.gz"), "rt").read())
    with gzip.open(os.path.join(BASIC, "train.vw.gz"), "rb") as f:
        text = f.read()
    words, off, used, rc = VowpalParser(vw).parse_buffer(text)
    assert rc == capi.OK and used == len(text) and len(off) == 101
    nd = fw.NamespaceDescriptor
    combos = [[vw.lookup(c)[0] for c in k] for k in KEEPS + interactions]  # single-letter vw names
    mi = fw.ModelInstance(learning_rate=0.025, power_t=power_t, bit_precision=25, add_constant_feature=True, init_acc_gradient=1.0,
                          optimizer=optimizer, feature_combo_descs=[fw.FeatureComboDesc([nd(i) for i in c]) for c in combos])
    ocfg = fwo.make_config(optimizer=optimizer, learning_rate=0.025, power_t=power_t, init_acc_gradient=1.0, bit_precision=25,
                           num_combos=len(combos) + 1, ffm_k=0, ffm_bit_precision=18, ffm_num_fields=0, ffm_learning_rate=0.025,
                           ffm_power_t=power_t, ffm_init_acc_gradient=1.0)
    ots = fwo.TranslatorSpec([([(i, False) for i in c], 1.0) for c in combos], [], True, 25, 0, 18)
    y = record_labels(words, off)
    # three passes over the 100 lines so that weights matter
    w3 = np.concatenate([words] * 3)
    o3 = np.concatenate([off[:-1], off[:-1] + off[-1], off + 2 * off[-1]])
    _, p_ref = fwo.Model(ocfg).run_stream(ots, w3, o3, holdout_after=0, nthreads=1)
    out = []
    for kind in ("entries", "records"):
        re = fw.Regressor(mi)
        fbt = fw.FeatureBufferTranslator(mi)
        b = re.batch_from_records(fbt, w3, o3) if kind == "entries" else re.record_batch(fbt, w3, o3)
        re.learn_batch(b, capi.MODE_SEQUENTIAL, True)
        p = b.predictions().copy()
        assert np.abs(logloss(p, np.tile(y, 3)) - logloss(p_ref, np.tile(y, 3))).max() < 1e-4
        assert np.abs(p - p_ref).max() < 5e-6
        out.append((p, re.table_checksum(capi.TABLE_LR)))
        b.close()
        re.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]  # device translation == host translation
    assert len(np.unique(out[0][0][200:])) > 50  # and it is not predicting a constant


@pytest.mark.statistical
def test_text_to_trainer_natively_with_cache_writing(tmp_path):
    """fwgpu_trainer_digest_text: the reference's training loop over VW text with `-c` (main.rs:213-270), all in native
    code: several parser threads, records learned in file order (hogwild on the device), cache written on the way"""
    vw = VwNamespaceMap(gzip.open(os.path.join(DATA, "vw_namespace_map.csv.gz"), "rt").read())
    train = _unpack("train.vw", tmp_path)
    text = open(train, "rb").read()
    mi = _model_instance(vw)
    re = fw.Regressor(mi)
    # 30 000 two-feature examples: with 768 of them in flight every example reads weights that are thousands of updates stale
    # and the latent factors do not form (balanced accuracy 0.49 on the hard set); 16 in flight = hogwild.rs's default
    re.set_max_in_flight(16)
    tr = fw.HogwildTrainer(re, mi, micro_batch=2048)
    parser = VowpalParser(vw)
    cache = RecordCache(train, True, vw)
    assert cache.writing
    n, used, rc = tr.digest_text(parser, text, cache=cache, threads=5)
    tr.block_until_workers_finished()
    cache.write_finish()
    cache.close()
    assert rc == capi.OK and n == 30000 and used == len(text) and tr.examples_seen() == 30000
    # the cache holds exactly what a single parser produces for the whole file
    words, off, _, _ = VowpalParser(vw).parse_buffer(text)
    raw = open(train + ".fwcache", "rb").read()
    assert raw.endswith(words.tobytes()) and raw[:4] == b"FWCA"
    # the hogwild-trained model solves the reference's hard test set too
    hw, ho, _ = _records_of(_unpack("test-hard.vw", tmp_path), vw, False)
    assert _balanced_accuracy(_predict(re, mi, hw, ho), record_labels(hw, ho)) > 0.80
    # a command in the middle stops the digestion exactly there, in order
    lines = text.split(b"\n")
    mixed = b"\n".join(lines[:1000]) + b"\nflush\n" + b"\n".join(lines[1000:2000]) + b"\n"
    re2 = fw.Regressor(mi)
    tr2 = fw.HogwildTrainer(re2, mi, micro_batch=512)
    n2, used2, rc2 = tr2.digest_text(parser, mixed, threads=4)
    tr2.block_until_workers_finished()
    assert rc2 == capi.PARSE_FLUSH and n2 == 1000 and used2 == len(b"\n".join(lines[:1000]) + b"\n") and tr2.examples_seen() == 1000
    with pytest.raises(capi.FwgpuError):
        tr2.digest_text(parser, b"1 |A x\n1 |NOPE y\n", threads=1)
    for x in (tr, tr2, re, re2):
        x.close()


@pytest.mark.statistical
def test_gz_input_file_to_trainer_then_cache_pass(tmp_path):
    """`fw --data train.vw.gz -c`: gzip text in (buffer_handler.rs:19-23), LZ4 cache out (cache.rs:73), everything native;
    the second pass trains from the cache alone"""
    import shutil
    vw = VwNamespaceMap(gzip.open(os.path.join(DATA, "vw_namespace_map.csv.gz"), "rt").read())
    src = str(tmp_path / "train.vw.gz")
    shutil.copy(os.path.join(DATA, "train.vw.gz"), src)
    mi = _model_instance(vw)
    re = fw.Regressor(mi)
    re.set_max_in_flight(16)
    tr = fw.HogwildTrainer(re, mi, micro_batch=1024)
    parser = VowpalParser(vw)
    cache = RecordCache(src, True, vw)
    assert cache.writing
    n, rc = tr.digest_file(parser, src, cache=cache, threads=4)
    tr.block_until_workers_finished()
    cache.write_finish()
    cache.close()
    assert rc == capi.OK and n == 30000
    raw = open(src + ".fwcache", "rb").read()
    assert raw[:4] == (0x184D2204).to_bytes(4, "little")  # an LZ4 frame, because the input name ends in "gz"
    cache = RecordCache(src, True, vw)
    assert cache.reading
    assert tr.digest_cache(cache) == 30000  # second pass over the same data, from the cache
    tr.block_until_workers_finished()
    hw, ho, _ = _records_of(_unpack("test-hard.vw", tmp_path), vw, False)
    assert _balanced_accuracy(_predict(re, mi, hw, ho), record_labels(hw, ho)) > 0.80
    for x in (cache, tr, re):
        x.close()
