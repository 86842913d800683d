"""Feed path on the CPU: namespace map, VW text parser against the reference's own known-answer tests
(tests/golden/parser_kats.json), serde-compatible JSON, LZ4 frames against an independent implementation (pyarrow's
liblz4), and the .fwcache reader / writer."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import fwumious_wabbit_amd as fw
from fwumious_wabbit_amd import capi
from fwumious_wabbit_amd.feed import FlushCommand, HogwildLoadCommand, RecordCache, VowpalParser, VwNamespaceMap

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "parser_kats.json")))


@pytest.mark.parametrize("group", KATS["groups"], ids=[g["name"] for g in KATS["groups"]])
def test_parser_reference_kats(group):
    vw = VwNamespaceMap(group["vwmap"])
    rr = VowpalParser(vw)  # ONE parser for the whole group, like the reference tests: no state may leak between lines
    for case in group["cases"]:
        line = case["line"].encode()
        if "error" in case:
            with pytest.raises(capi.FwgpuError) as e:
                rr.next_vowpal(line)
            assert e.value.code == capi.ERR_PARSE and e.value.message == case["error"], case
        elif case.get("command") == "flush":
            with pytest.raises(FlushCommand):
                rr.next_vowpal(line)
        elif case.get("command") == "hogwild_load":
            with pytest.raises(HogwildLoadCommand) as e:
                rr.next_vowpal(line)
            assert e.value.filename == case["filename"]
        elif "cached" in case:
            got = rr.next_vowpal_with_cache(case["cached"].encode(), line)
            assert got.tolist() == case["record"], case
        else:
            got = rr.next_vowpal(line)
            assert got.tolist() == case["record"], case
            if "size" in case:  # next_vowpal_with_size: the line's length without its newline
                assert len(line.rstrip(b"\n")) == case["size"]


def _parse_outcome(fn):
    try:
        return ("ok", fn().tolist())
    except capi.FwgpuError as e:
        return ("error", e.code, e.message)
    except FlushCommand:
        return ("flush",)
    except HogwildLoadCommand as e:
        return ("hogwild_load", e.filename)


def test_scanned_context_equals_concatenation_reference_kats():
    """next_vowpal_with_cache's cases (parser.rs:1142-1183) through the scan-once route."""
    for group in KATS["groups"]:
        rr = VowpalParser(VwNamespaceMap(group["vwmap"]))
        for case in group["cases"]:
            if "cached" in case and "record" in case:
                px = rr.scan_context(case["cached"].encode())
                assert rr.next_vowpal_after(px, case["line"].encode()).tolist() == case["record"], case


def test_scanned_context_equals_concatenation_at_every_split():
    """Every way of cutting a line into context + request gives what the concatenating parser gives: records, error messages
    and commands alike (tokens cut in the middle, weights cut at the colon, cuts inside runs of spaces, before the first bar,
    repeated namespaces on either side, float namespaces, a context that is itself an error)."""
    vw = VwNamespaceMap("A,featureA\nB,featureB\nC,featureC,f32\nDD,featureD\n_namespace_skip_prefix,2\n")
    rr = VowpalParser(vw)
    lines = [
        b"1 |A a b:2.5 c |B x  y |C  yy3.5 |DD:0.5 q r:3 \n",
        b"-1 0.25 |B only |A  a:0.1  |A again |DD z\n",
        b"|A a |B b |C zz1 zz2\n",
        b"1 |A a:1.0e-3 |B:2 b c:4|x |DD  \n",
        b"-1 3 |A |B |DD d\n",
        b"1 |A a |Q nope |B b\n",          # unknown namespace: error wherever the cut is
        b"1 |A a:notanumber |B b\n",       # bad weight
        b"1 |C zzbad |A a\n",              # bad float in a float namespace
        b"1 |C:2 zz1 |A a\n",              # weighted float namespace
        b"1 abc |A a\n",                   # bad importance
        b"1 -2 |A a\n",                    # negative importance
        b"flush\n",
        b"hogwild_load some/file.fw\n",
        b"two tokens\n",
        b"1 no bar at all\n",
        b"1 |A a",                          # no newline: the last byte is dropped (parser.rs:264)
        b"1 |A  a   ",
    ]
    resumable = 0
    for line in lines:
        for cut in range(len(line) + 1):
            ctx, rest = line[:cut], line[cut:]
            want = _parse_outcome(lambda: rr.next_vowpal_with_cache(ctx, rest))
            px = rr.scan_context(ctx)
            resumable += px.resumable
            got = _parse_outcome(lambda: rr.next_vowpal_after(px, rest))
            assert got == want, (line, cut, px.resumable)
            # one scanned context serves many requests, and another parser of the same map
            other = VowpalParser(vw)
            assert _parse_outcome(lambda: other.next_vowpal_after(px, rest)) == want
            assert _parse_outcome(lambda: rr.next_vowpal_after(px, rest)) == want
    assert resumable > 100  # (the scan-once route is what most cuts take)


def test_scanned_context_equals_concatenation_random_lines():
    rng = np.random.default_rng(11)
    names = ["A", "B", "C", "DD"]
    vw = VwNamespaceMap("A,featureA\nB,featureB\nC,featureC,f32\nDD,featureD\n_namespace_skip_prefix,2\n")
    rr = VowpalParser(vw)

    def feature(ns):
        if ns == "C":
            return "zz%g" % rng.normal()
        f = "f%d" % rng.integers(0, 50)
        return f + (":%g" % rng.uniform(0.1, 3)) * bool(rng.random() < 0.3)

    def part(nss):
        out = ""
        for ns in nss:
            w = ":%g" % rng.uniform(0.5, 2) if (ns != "C" and rng.random() < 0.2) else ""
            out += "|" + ns + w + " " * int(rng.integers(1, 3))
            out += "".join(feature(ns) + " " * int(rng.integers(1, 3)) for _ in range(rng.integers(0, 5)))
        return out

    for _ in range(300):
        order = list(rng.permutation(names))
        k = int(rng.integers(0, 5))
        ctx = (rng.choice(["1 ", "-1 ", "1 0.5 ", ""]) + part(order[:k])).encode()
        if rng.random() < 0.3:
            ctx = ctx.rstrip()  # the request then continues the context's last token
        px = rr.scan_context(ctx)
        for _ in range(4):
            extra = list(rng.permutation(names))[: int(rng.integers(0, 4))]  # may repeat a namespace of the context
            cand = (rng.choice(["", " ", "x "]) + part(order[k:] + extra) + rng.choice(["\n", ""])).encode()
            want = _parse_outcome(lambda: rr.next_vowpal_with_cache(ctx, cand))
            assert _parse_outcome(lambda: rr.next_vowpal_after(px, cand)) == want, (ctx, cand)


def _namespaces(rec, n_ns, base=None):
    """record -> per namespace slot the list of (hash, value bits); slots reading NO_FEATURES come from `base` (a context's record)"""
    out = []
    for ns in range(n_ns):
        r, w = rec, int(rec[3 + ns])
        if w == 0x80000000 and base is not None:
            r, w = base, int(base[3 + ns])
        if not (w & 0x80000000):
            out.append([(w, 0x3f800000)])
        else:
            a, b = (w >> 16) & 0x3fff, w & 0xffff
            out.append([(int(r[o]), int(r[o + 1])) for o in range(a, b - 1, 2)])
    return out


def test_candidate_only_records_carry_what_the_merged_record_carries():
    """fwgpu_parser_parse_candidate: context record + candidate-only record == the record of context + candidate, namespace by
    namespace (hashes and value bits), incl. candidates that name a context namespace again; a candidate that continues the
    context's last namespace comes back merged"""
    rng = np.random.default_rng(5)
    names = ["A", "B", "C", "DD", "E"]
    vw = VwNamespaceMap("A,fa\nB,fb\nC,fc,f32\nDD,fd\nE,fe\n_namespace_skip_prefix,2\n")
    rr = VowpalParser(vw)
    n_ns = vw.num_namespaces

    def part(nss):
        out = ""
        for ns in nss:
            feats = ["zz%g" % rng.normal() if ns == "C" else "f%d" % rng.integers(0, 50) + (":%g" % rng.uniform(0.1, 3)) * bool(rng.random() < 0.3)
                     for _ in range(rng.integers(0, 4))]
            out += "|" + ns + (":2" if ns != "C" and rng.random() < 0.15 else "") + " " + "".join(f + " " for f in feats)
        return out

    seen = {True: 0, False: 0}
    for _ in range(200):
        order = list(rng.permutation(names))
        k = int(rng.integers(1, 4))
        ctx = (rng.choice(["1 ", "-1 0.5 ", ""]) + part(order[:k])).encode()
        px = rr.scan_context(ctx)
        ctx_rec = rr.next_vowpal(ctx)
        assert px.is_record(ctx_rec)  # (every context here ends with a space: the scan reaches its end)
        for _ in range(4):
            extra = list(rng.permutation(order[:k]))[: int(rng.integers(0, 2))]  # sometimes a context namespace again
            lead = rng.choice(["", "", "f7 "])                                     # sometimes a feature that continues the context's last namespace
            cand = (lead + part(order[k:] + extra) + "\n").encode()
            merged = rr.next_vowpal_with_cache(ctx, cand)
            got, is_delta = rr.next_vowpal_candidate(px, cand)
            seen[is_delta] += 1
            if not is_delta:
                assert got.tolist() == merged.tolist() and lead
                continue
            # (a lead feature can still split off when the context's last namespace left no feature words behind)
            assert got[0] == len(got) and got[1] == merged[1] and got[2] == merged[2]
            assert len(got) <= len(merged) - len(ctx_rec) + 3 + n_ns
            assert _namespaces(got, n_ns, base=ctx_rec) == _namespaces(merged, n_ns)
    assert seen[True] > 300 and seen[False] > 30
    # a context whose last token the request continues is not the context's own record: no candidate-only form
    px = rr.scan_context(b"1 |A f1 |B f2")
    assert not px.is_record(rr.next_vowpal(b"1 |A f1 |B f2"))


def test_vwmap_reference_kats():
    for k in KATS["vwmap"]:
        if "error" in k:
            with pytest.raises(capi.FwgpuError) as e:
                VwNamespaceMap(k["csv"])
            assert e.value.message == k["error"]
            continue
        vw = VwNamespaceMap(k["csv"])
        src = json.loads(vw.to_json())
        assert src["namespace_skip_prefix"] == k["skip_prefix"]
        got = [[e["namespace_vwname"], e["namespace_verbose"], e["namespace_index"], e["namespace_format"]] for e in src["entries"]]
        assert got == k["entries"]
        assert vw.num_namespaces == max(e[2] for e in k["entries"]) + 1


def test_vw_source_json_is_serde_pretty_and_round_trips():
    vw = VwNamespaceMap("A,featureA\nB,featureB,f32\n_namespace_skip_prefix,1\n")
    expect = (b'{\n  "namespace_skip_prefix": 1,\n  "entries": [\n    {\n      "namespace_vwname": "A",\n'
              b'      "namespace_verbose": "featureA",\n      "namespace_index": 0,\n      "namespace_format": "Categorical"\n'
              b'    },\n    {\n      "namespace_vwname": "B",\n      "namespace_verbose": "featureB",\n'
              b'      "namespace_index": 1,\n      "namespace_format": "F32"\n    }\n  ]\n}')
    assert vw.to_json() == expect
    vw2 = VwNamespaceMap.new_from_buf(expect)
    assert vw2.to_json() == expect and vw2.lookup("B") == (1, True) and vw2.lookup("featureA", verbose=True) == (0, False)
    assert VwNamespaceMap("").to_json() == b'{\n  "namespace_skip_prefix": 0,\n  "entries": []\n}'
    with pytest.raises(capi.FwgpuError):
        VwNamespaceMap.new_from_buf(b'{"namespace_skip_prefix": 0}')  # missing field `entries`


def test_f32_text_matches_ryu_layout():
    L = capi.lib()
    buf = C.create_string_buffer(64)

    def fmt(x):
        capi.check(L.fwgpu_debug_format_f32(C.c_float(x), buf, 64))
        return buf.value.decode()

    known = {0.1: "0.1", 1.0: "1.0", 0.025: "0.025", 0.38: "0.38", 0.0: "0.0", -2.5: "-2.5", 1e-7: "1e-7", 1e-6: "0.000001",
             1e-5: "0.00001", 1e12: "1000000000000.0", 1e13: "1e13", 1.5e10: "15000000000.0", 3.4028235e38: "3.4028235e38",
             1.17549435e-38: "1.1754944e-38", 16777216.0: "16777216.0", 0.3: "0.3", 123456.79: "123456.79"}
    for x, t in known.items():
        assert fmt(x) == t, (x, fmt(x), t)
    # every output parses back to the same f32 and is the shortest digit string numpy finds
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.standard_normal(300).astype(np.float32) * np.float32(10.0) ** rng.integers(-12, 12, 300).astype(np.float32),
                         rng.integers(0, 2 ** 32, 300, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    for x in xs[np.isfinite(xs)]:
        t = fmt(float(x))
        assert np.float32(t) == x
        digits = np.format_float_scientific(x, unique=True, trim="-").split("e")[0].replace(".", "").replace("-", "")
        assert t.replace(".", "").replace("-", "").split("e")[0].strip("0") == digits.strip("0"), (x, t, digits)


def _frame_via_pyarrow(data: bytes) -> bytes:
    pa = pytest.importorskip("pyarrow")
    return pa.compress(data, codec="lz4", asbytes=True)  # arrow's "lz4" codec is the LZ4 FRAME format (liblz4)


def _records(n, seed=3):
    recs, off = fw.synth_records(6, 2.0, 1.1, 5000, 0.3, seed, 0, n)
    return recs, off


VW6 = "".join(f"{chr(65 + i)},ns{i}\n" for i in range(6))


@pytest.mark.parametrize("gz", [False, True])
def test_cache_write_then_read(tmp_path, gz):
    vw = VwNamespaceMap(VW6)
    inp = str(tmp_path / ("train.vw.gz" if gz else "train.vw"))
    recs, off = _records(3000)
    rc = RecordCache(inp, True, vw)
    assert rc.writing and not rc.reading and os.path.exists(inp + ".fwcache.writing")
    for i in range(0, 3000, 500):  # pushed in pieces, like push_record per example
        rc.push_records(recs[int(off[i]):int(off[i + 500])])
    rc.write_finish()
    rc.close()
    assert os.path.exists(inp + ".fwcache") and not os.path.exists(inp + ".fwcache.writing")
    raw = open(inp + ".fwcache", "rb").read()
    if gz:
        pa = pytest.importorskip("pyarrow")
        assert raw[:4] == struct.pack("<I", 0x184D2204)
        # an independent LZ4 implementation reads the frame we wrote
        raw = pa.Codec("lz4").decompress(raw, decompressed_size=16 + len(vw.to_json()) + recs.nbytes, asbytes=True)
        assert len(open(inp + ".fwcache", "rb").read()) < len(raw)  # and it did compress
    # cache.rs:12-26 layout
    assert raw[:4] == b"FWCA" and struct.unpack("<I", raw[4:8])[0] == 11
    jl = struct.unpack("<Q", raw[8:16])[0]
    assert raw[16:16 + jl] == vw.to_json()
    assert raw[16 + jl:] == recs.tobytes()
    # second open: reading mode; records come back identical, in bulk, across small buffers
    rc = RecordCache(inp, True, vw)
    assert rc.reading and not rc.writing
    got, n = [], 0
    while True:
        w, o = rc.next_records(words_cap=4096, max_records=100)
        if len(o) <= 1:
            break
        assert o[0] == 0 and o[-1] == len(w) and all(w[int(a)] == int(b) - int(a) for a, b in zip(o[:-1], o[1:]))
        got.append(w)
        n += len(o) - 1
    assert n == 3000 and np.array_equal(np.concatenate(got), recs)
    rc.close()
    # a different namespace map invalidates the cache: it is rewritten (cache.rs:96-102, 177-182)
    rc = RecordCache(inp, True, VwNamespaceMap(VW6 + "G,ns6\n"))
    assert rc.writing and not rc.reading
    rc.close()


def test_cache_reads_frames_written_by_liblz4(tmp_path):
    """the reference writes gz caches through liblz4 (linked blocks, content checksum): decode such a stream"""
    vw = VwNamespaceMap(VW6)
    recs, _ = _records(20000, seed=9)
    js = vw.to_json()
    payload = b"FWCA" + struct.pack("<I", 11) + struct.pack("<Q", len(js)) + js + recs.tobytes()
    inp = str(tmp_path / "big.vw.gz")
    with open(inp + ".fwcache", "wb") as f:
        f.write(_frame_via_pyarrow(payload))
    rc = RecordCache(inp, True, vw)
    assert rc.reading
    got = []
    while True:
        w, o = rc.next_records()
        if len(o) <= 1:
            break
        got.append(w)
    assert np.array_equal(np.concatenate(got), recs)
    rc.close()
    # a frame WITH a content checksum (ours has one, like the lz4 crate's default): a flipped payload byte is caught,
    # by the block decoder or at the latest by the checksum at the end of the frame
    inp2 = str(tmp_path / "own.vw.gz")
    rc = RecordCache(inp2, True, vw)
    rc.push_records(recs)
    rc.write_finish()
    rc.close()
    raw = bytearray(open(inp2 + ".fwcache", "rb").read())
    assert raw[4] & 0x04  # FLG.C_Checksum
    raw[len(raw) // 2] ^= 0x55
    open(inp2 + ".fwcache", "wb").write(bytes(raw))
    rc = RecordCache(inp2, True, vw)
    assert rc.reading
    with pytest.raises(capi.FwgpuError):
        while len(rc.next_records()[1]) > 1:
            pass
    rc.close()


def test_cache_rejects_bad_headers(tmp_path):
    vw = VwNamespaceMap(VW6)
    inp = str(tmp_path / "x.vw")
    js = vw.to_json()
    for blob in (b"FWFW" + struct.pack("<I", 11) + struct.pack("<Q", len(js)) + js,   # wrong magic
                 b"FWCA" + struct.pack("<I", 10) + struct.pack("<Q", len(js)) + js,   # older version
                 b"FWCA" + struct.pack("<I", 11) + struct.pack("<Q", 5) + b"{bad}"):  # broken JSON
        open(inp + ".fwcache", "wb").write(blob)
        rc = RecordCache(inp, True, vw)
        assert rc.writing and not rc.reading  # falls back to rebuilding the cache
        rc.close()
    # a file that ends inside a record
    recs, off = _records(10)
    open(inp + ".fwcache", "wb").write(b"FWCA" + struct.pack("<I", 11) + struct.pack("<Q", len(js)) + js + recs.tobytes()[:-8])
    rc = RecordCache(inp, True, vw)
    assert rc.reading
    w, o = rc.next_records()
    assert len(o) - 1 == 9
    with pytest.raises(capi.FwgpuError):
        rc.next_records()
    rc.close()
    assert not RecordCache(inp, False, vw).reading  # enabled = false: neither reads nor writes


def test_text_to_records_in_bulk_equals_line_by_line():
    vw = VwNamespaceMap(VW6)
    rng = np.random.default_rng(11)
    lines = []
    for i in range(2000):
        parts = [str(1 if rng.random() < 0.4 else -1)]
        if rng.random() < 0.2:
            parts.append(f"{rng.random() * 3:.3f}")
        for ns in rng.permutation(6)[: rng.integers(1, 7)]:
            feats = " ".join(f"f{rng.integers(0, 1000)}" + (f":{rng.random() * 2:.2f}" if rng.random() < 0.2 else "")
                             for _ in range(rng.integers(1, 5)))
            parts.append(f"|{chr(65 + ns)}" + (":0.5" if rng.random() < 0.1 else "") + " " + feats)
        lines.append(" ".join(parts) + "\n")
    text = "".join(lines).encode()
    p = VowpalParser(vw)
    one_by_one = [p.next_vowpal(l.encode()) for l in lines]
    words, off, used, rc = p.parse_buffer(text)
    assert rc == capi.OK and used == len(text) and len(off) == 2001
    assert np.array_equal(words, np.concatenate(one_by_one))
    # a command line stops the bulk parse exactly there
    words2, off2, used2, rc2 = p.parse_buffer("".join(lines[:5]).encode() + b"flush\n" + lines[5].encode())
    assert rc2 == capi.PARSE_FLUSH and len(off2) == 6 and used2 == len("".join(lines[:5]))
    # records from text feed the translator like synthetic ones: hashes masked, values kept
    mi = fw.ModelInstance(bit_precision=18, ffm_k=4, ffm_bit_precision=18, add_constant_feature=True,
                          feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(i)]) for i in range(6)],
                          ffm_fields=[[fw.NamespaceDescriptor(i)] for i in range(6)])
    fb = fw.FeatureBufferTranslator(mi).translate(one_by_one[0])
    assert len(fb.lr_buffer) >= 2 and fb.label in (0.0, 1.0)


def test_reference_example_lines_parse_and_translate_like_the_oracle():
    """examples/basic/datasets/train.vw (the reference's own production-like lines: 58 namespaces, several features per
    namespace, `|w:2` namespace weights, 2/3/4-way --interactions): parser -> records -> host translator vs the oracle's
    translator, bit for bit; record structure checked against the format description (parser.rs:57-74)"""
    import gzip
    from oracle import fwo
    base = os.path.join(HERE, "golden", "example_basic")
    vw = VwNamespaceMap(gzip.open(os.path.join(base, "vw_namespace_map.csv.gz"), "rt").read())
    NS = vw.num_namespaces
    assert NS == 58
    with gzip.open(os.path.join(base, "train.vw.gz"), "rb") as f:
        text = f.read()
    p = VowpalParser(vw)
    words, off, used, rc = p.parse_buffer(text)
    assert rc == capi.OK and used == len(text) and len(off) == 101
    keeps = "B C D F G H L O S U W e f g h i o p q r v x".split()
    inter = "4G 4GHX 4GUW 4K 4c 4go 4v BC BD BGO BX CO DG DW GU Gx KR MN UW Ug eg".split()
    combos = [[vw.lookup(c)[0] for c in k] for k in keeps + inter]
    nd = fw.NamespaceDescriptor
    mi = fw.ModelInstance(bit_precision=25, add_constant_feature=True,
                          feature_combo_descs=[fw.FeatureComboDesc([nd(i) for i in c]) for c in combos],
                          ffm_k=4, ffm_bit_precision=20, ffm_fields=[[nd(vw.lookup("B")[0]), nd(vw.lookup("C")[0])], [nd(vw.lookup("O")[0])],
                                                                     [nd(vw.lookup("w")[0])]])
    ots = fwo.TranslatorSpec([([(i, False) for i in c], 1.0) for c in combos],
                             [[(vw.lookup("B")[0], False), (vw.lookup("C")[0], False)], [(vw.lookup("O")[0], False)], [(vw.lookup("w")[0], False)]],
                             True, 25, 4, 20)
    fbt = fw.FeatureBufferTranslator(mi)
    lines = text.split(b"\n")
    n_multi = 0
    for i in range(100):
        r = words[int(off[i]):int(off[i + 1])]
        assert r[0] == len(r) and r[1] in (0, 1) and r[2] == 0x3F800000
        assert np.array_equal(r, p.next_vowpal(lines[i] + b"\n"))
        # `|w:2 w...` : one feature with namespace weight 2 -> out-of-place pair (hash, 2.0)
        slot = int(r[3 + vw.lookup("w")[0]])
        assert slot & 0x80000000 and struct.unpack("<f", struct.pack("<I", int(r[(slot >> 16) & 0x3fff] if False else r[((slot >> 16) & 0x3fff) + 1])))[0] == 2.0
        n_multi += sum(1 for ns in range(NS) if (int(r[3 + ns]) & 0x80000000) and int(r[3 + ns]) != 0x80000000
                       and ((int(r[3 + ns]) & 0xffff) - ((int(r[3 + ns]) >> 16) & 0x3fff)) > 2)
        fb = fbt.translate(r)
        lr, ffm, label, imp = ots.translate(r)
        assert fb.lr_buffer.tobytes() == lr.tobytes() and fb.ffm_buffer.tobytes() == ffm.tobytes()
        assert fb.label == label and fb.example_importance == imp
    assert n_multi > 20  # the data does exercise several-features-per-namespace slots


def test_lz4_frame_flag_variants_and_xxh32_against_the_xxhash_library(tmp_path):
    """Frames carrying content size, block checksums and a content checksum (every optional part of the frame descriptor):
    built here from liblz4's blocks with checksums from the independent `xxhash` package, read by the library's decoder
    (which verifies all of them with its own xxHash32)."""
    xxhash = pytest.importorskip("xxhash")
    vw = VwNamespaceMap(VW6)
    recs, _ = _records(30000, seed=13)
    js = vw.to_json()
    payload = b"FWCA" + struct.pack("<I", 11) + struct.pack("<Q", len(js)) + js + recs.tobytes()
    src = _frame_via_pyarrow(payload)
    assert src[:4] == struct.pack("<I", 0x184D2204)
    flg, bd = src[4], src[5]
    pos = 6 + (8 if flg & 0x08 else 0) + (4 if flg & 0x01 else 0) + 1
    blocks = []
    while True:
        (sz,) = struct.unpack("<I", src[pos:pos + 4])
        pos += 4
        if sz == 0:
            break
        n = sz & 0x7fffffff
        blocks.append((sz, src[pos:pos + n]))
        pos += n + (4 if flg & 0x10 else 0)
    assert len(blocks) >= 2  # several blocks, so linked-block history is exercised if liblz4 linked them

    def build(with_size, with_block_sums, with_content_sum, corrupt=None):
        f = 0x40 | (flg & 0x20) | (0x08 if with_size else 0) | (0x10 if with_block_sums else 0) | (0x04 if with_content_sum else 0)
        desc = bytes([f, bd]) + (struct.pack("<Q", len(payload)) if with_size else b"")
        out = struct.pack("<I", 0x184D2204) + desc + bytes([(xxhash.xxh32(desc, seed=0).intdigest() >> 8) & 0xff])
        for i, (sz, data) in enumerate(blocks):
            out += struct.pack("<I", sz) + data
            if with_block_sums:
                d = xxhash.xxh32(data, seed=0).intdigest()
                out += struct.pack("<I", d ^ (1 if corrupt == ("block", i) else 0))
        out += struct.pack("<I", 0)
        if with_content_sum:
            out += struct.pack("<I", xxhash.xxh32(payload, seed=0).intdigest() ^ (1 if corrupt == ("content",) else 0))
        return out

    def read_all(frame):
        inp = str(tmp_path / "v.vw.gz")
        open(inp + ".fwcache", "wb").write(frame)
        rc = RecordCache(inp, True, vw)
        if not rc.reading:
            rc.close()
            os.remove(inp + ".fwcache.writing")
            raise capi.FwgpuError(5, "header rejected")
        got = []
        try:
            while True:
                w, o = rc.next_records()
                if len(o) <= 1:
                    break
                got.append(w)
        finally:
            rc.close()
        return np.concatenate(got)

    for flags in [(False, False, False), (True, False, False), (False, True, False), (False, False, True), (True, True, True)]:
        assert np.array_equal(read_all(build(*flags)), recs), flags
    for corrupt in (("block", 0), ("block", len(blocks) - 1), ("content",)):
        with pytest.raises(capi.FwgpuError):
            read_all(build(True, True, True, corrupt=corrupt))
    # a wrong header checksum is rejected before any block is read (the cache is then rebuilt)
    bad = bytearray(build(True, False, False))
    bad[4 + 2 + 8] ^= 0xff
    with pytest.raises(capi.FwgpuError):
        read_all(bytes(bad))


def test_parser_survives_garbage():
    """No input may crash the parser or make it read outside the line: random bytes, truncated and spliced valid lines,
    absurdly long tokens.  Every call must end in a record, a command or a parse error."""
    vw = VwNamespaceMap(VW6 + "_namespace_skip_prefix,1\nG,fl,f32\n")
    p = VowpalParser(vw)
    rng = np.random.default_rng(99)
    good = [b"1 |A a b:2 |B:0.5 c |G G1.5 Gx\n", b"-1 0.25 |C x |D y:3e2 |E z:NONE\n", b"|F only |A q\n", b"flush\n",
            b"hogwild_load /x/y\n"]
    outcomes = {"ok": 0, "err": 0, "cmd": 0}
    for i in range(4000):
        kind = i % 4
        if kind == 0:
            line = bytes(rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8))
        elif kind == 1:
            g = good[int(rng.integers(0, len(good)))]
            line = g[: int(rng.integers(1, len(g) + 1))]
        elif kind == 2:
            a, b_ = good[int(rng.integers(0, 3))], good[int(rng.integers(0, 3))]
            cut = int(rng.integers(0, len(a)))
            line = a[:cut] + bytes(rng.integers(32, 127, size=int(rng.integers(0, 8)), dtype=np.uint8)) + b_[int(rng.integers(0, len(b_))):]
        else:
            line = b"1 |A " + b"x" * int(rng.integers(1, 5000)) + b":" + b"9" * int(rng.integers(0, 400)) + b"\n"
        try:
            r = p.next_vowpal(line)
            assert len(r) == 0 or r[0] == len(r)
            outcomes["ok"] += 1
        except capi.FwgpuError as e:
            assert e.code == capi.ERR_PARSE
            outcomes["err"] += 1
        except (FlushCommand, HogwildLoadCommand):
            outcomes["cmd"] += 1
    assert outcomes["ok"] > 500 and outcomes["err"] > 500 and outcomes["cmd"] > 10, outcomes
    # the bulk entry point on a buffer that ends without a newline and contains NUL bytes
    words, off, used, rc = p.parse_buffer(b"1 |A a\n-1 |B b\x00c\n1 |C tail")
    assert rc == capi.OK and len(off) == 4 and used == len(b"1 |A a\n-1 |B b\x00c\n1 |C tail")


def test_cache_and_json_readers_survive_garbage(tmp_path):
    """Corrupted cache files (raw and LZ4) and corrupted JSON must end in an error or in clean records, never in a crash."""
    from fwumious_wabbit_amd import persistence as P
    vw = VwNamespaceMap(VW6)
    recs, _ = _records(300, seed=17)
    rng = np.random.default_rng(123)
    for gz in (False, True):
        inp = str(tmp_path / ("g.vw.gz" if gz else "g.vw"))
        c = RecordCache(inp, True, vw)
        c.push_records(recs)
        c.write_finish()
        c.close()
        good = open(inp + ".fwcache", "rb").read()
        for trial in range(150):
            b = bytearray(good)
            for _ in range(int(rng.integers(1, 6))):
                kind = int(rng.integers(0, 3))
                pos = int(rng.integers(0, len(b)))
                if kind == 0:
                    b[pos] = int(rng.integers(0, 256))
                elif kind == 1:
                    del b[pos:pos + int(rng.integers(1, 64))]
                else:
                    b[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8))
            open(inp + ".fwcache", "wb").write(bytes(b))
            rc = RecordCache(inp, True, vw)
            try:
                if rc.reading:
                    for _ in range(1000):
                        w, o = rc.next_records(words_cap=1 << 16, max_records=64)
                        if len(o) <= 1:
                            break
                        assert all(w[int(a)] == int(b_) - int(a) for a, b_ in zip(o[:-1], o[1:]))
            except capi.FwgpuError:
                pass
            finally:
                rc.close()
            for leftover in (inp + ".fwcache.writing",):
                if os.path.exists(leftover):
                    os.remove(leftover)
    # JSON documents: mutated vw_source / ModelInstance text
    mi = fw.ModelInstance(ffm_k=2, ffm_fields=[[fw.NamespaceDescriptor(0)]], feature_combo_descs=[fw.FeatureComboDesc([fw.NamespaceDescriptor(0)])])
    docs = [vw.to_json(), P.ModelInstanceHandle.from_model_instance(mi).to_json()]
    for doc in docs:
        for trial in range(300):
            b = bytearray(doc)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, len(b)))
                b[pos:pos + int(rng.integers(0, 3))] = bytes(rng.integers(32, 127, size=int(rng.integers(0, 4)), dtype=np.uint8))
            try:
                if doc is docs[0]:
                    VwNamespaceMap.new_from_buf(bytes(b)).to_json()
                else:
                    P.ModelInstanceHandle.from_json(bytes(b)).to_json()
            except capi.FwgpuError:
                pass


def test_create_buffered_input_vw_gz_zst(tmp_path):
    """buffer_handler.rs:77-137: the same bytes come back from .vw, .gz (also multi-member) and .zst; other extensions are refused"""
    import gzip
    from fwumious_wabbit_amd.feed import create_buffered_input
    contents = b"".join(b"%d |A a%d |B b%d\n" % (i % 2 * 2 - 1, i, i * 7) for i in range(50000))
    (tmp_path / "x.vw").write_bytes(contents)
    assert b"".join(create_buffered_input(str(tmp_path / "x.vw"))) == contents
    with gzip.open(tmp_path / "x.gz", "wb") as f:
        f.write(contents)
    assert b"".join(create_buffered_input(str(tmp_path / "x.gz"))) == contents
    with open(tmp_path / "multi.gz", "wb") as f:  # MultiGzDecoder: members back to back
        f.write(gzip.compress(contents[:100000]) + gzip.compress(contents[100000:]))
    assert b"".join(create_buffered_input(str(tmp_path / "multi.gz"))) == contents
    pa = pytest.importorskip("pyarrow")
    (tmp_path / "x.zst").write_bytes(pa.compress(contents, codec="zstd", asbytes=True))
    assert b"".join(create_buffered_input(str(tmp_path / "x.zst"))) == contents
    (tmp_path / "x.txt").write_bytes(contents)
    with pytest.raises(capi.FwgpuError) as e:
        create_buffered_input(str(tmp_path / "x.txt"))
    assert e.value.message == "Please specify a valid input format (.vw, .zst, .gz)"
    (tmp_path / "bad.gz").write_bytes(gzip.compress(contents)[:-200] + b"garbage" * 40)
    with pytest.raises(capi.FwgpuError):
        b"".join(create_buffered_input(str(tmp_path / "bad.gz")))
