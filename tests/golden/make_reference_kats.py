#!/usr/bin/env python3
"""Transcribes the reference's own known-answer tests for the LR+FFM hot path into
tests/golden/reference_kats.json (data only: inputs and the expected outputs asserted by the
reference's #[test] functions; every scenario cites the reference file:line it was read from).

Run:  python tests/golden/make_reference_kats.py

Scenario schema:
  name, source, wiring ("regressor" | "ffm_only"), config {...ModelInstance fields the blocks read...},
  ffm_fill (optional: tests' ffm_init forcing every FFM weight to this value),
  steps: [{op: learn|predict|forward_backward, lr: [[hash,value,combo]], ffm: [[hash,value,field*k]],
           label, importance, update, expect, cmp: "eq" | "eps"}]
"eq"  = the reference asserts exact f32 equality (assert_eq!);
"eps" = the reference asserts |x-y| < 5e-6 (assert_epsilon!, block_helpers.rs:30-40).
"stale" (optional) = the assertion no longer matches the reference's CURRENT code (the FFM block
tests are #[ignore]d in this snapshot); see the note in the scenario.
"""
import json
import os

SGD, FLEX, LUT = 100, 200, 300

# ModelInstance::new_empty(), model_instance.rs:120-150.  num_combos=1: add_constant_feature=true
# with no feature_combo_descs gives the LR block one output slot (block_lr.rs:53-56).
DEFAULT = dict(optimizer=SGD, learning_rate=0.5, power_t=0.5, init_acc_gradient=1.0, bit_precision=18,
               num_combos=1, ffm_k=0, ffm_bit_precision=18, ffm_num_fields=0, ffm_learning_rate=0.5,
               ffm_power_t=0.5, ffm_init_acc_gradient=0.0)


def cfg(**kw):
    c = dict(DEFAULT)
    c.update(kw)
    return c


def step(op, expect, lr=(), ffm=(), label=0.0, importance=1.0, update=True, cmp="eq", **extra):
    d = dict(op=op, lr=[list(x) for x in lr], ffm=[list(x) for x in ffm], label=label, importance=importance,
             update=update, expect=expect, cmp=cmp)
    d.update(extra)
    return d


S = []

# ------------------------------------------------------------------ regressor.rs (LR only)
one = [(1, 1.0, 0)]
S.append(dict(name="test_learning_turned_off", source="regressor.rs:556-595", wiring="regressor",
              config=cfg(optimizer=LUT),
              steps=[step("learn", 0.5, lr=[], update=False), step("learn", 0.5, lr=one, update=False),
                     step("learn", 0.5, lr=[(1, 1.0, 0), (2, 1.0, 0)], update=False)]))
S.append(dict(name="test_power_t_zero", source="regressor.rs:597-628", wiring="regressor",
              config=cfg(optimizer=FLEX, learning_rate=0.1, power_t=0.0),
              steps=[step("learn", 0.5, lr=one), step("learn", 0.48750263, lr=one),
                     step("learn", 0.47533244, lr=one)]))
dup = [(1, 1.0, 0), (1, 2.0, 0)]
S.append(dict(name="test_double_same_feature", source="regressor.rs:630-658", wiring="regressor",
              config=cfg(optimizer=LUT, learning_rate=0.1, power_t=0.0),
              steps=[step("learn", 0.5, lr=dup), step("learn", 0.38936076, lr=dup),
                     step("learn", 0.30993468, lr=dup)]))
S.append(dict(name="test_power_t_half__", source="regressor.rs:660-708", wiring="regressor",
              config=cfg(optimizer=FLEX, learning_rate=0.1, power_t=0.5, init_acc_gradient=0.0),
              steps=[step("learn", 0.5, lr=one), step("learn", 0.4750208, lr=one),
                     step("learn", 0.45788094, lr=one)]))
S.append(dict(name="test_power_t_half_fastmath", source="regressor.rs:710-752", wiring="regressor",
              config=cfg(optimizer=LUT, learning_rate=0.1, power_t=0.5, init_acc_gradient=0.0),
              steps=[step("learn", 0.5, lr=one), step("learn", 0.475734, lr=one)]))
two = [(1, 1.0, 0), (2, 1.0, 0)]
S.append(dict(name="test_power_t_half_two_features", source="regressor.rs:754-816", wiring="regressor",
              config=cfg(optimizer=FLEX, learning_rate=0.1, power_t=0.5, init_acc_gradient=0.0),
              steps=[step("learn", 0.5, lr=two), step("learn", 0.45016602, lr=two),
                     step("learn", 0.45836908, lr=one)]))
v2 = [(1, 2.0, 0)]
S.append(dict(name="test_non_one_weight", source="regressor.rs:818-866", wiring="regressor",
              config=cfg(optimizer=LUT, learning_rate=0.1, power_t=0.0),
              steps=[step("learn", 0.5, lr=v2), step("learn", 0.45016602, lr=v2),
                     step("learn", 0.40611085, lr=v2)]))
S.append(dict(name="test_example_importance", source="regressor.rs:868-884", wiring="regressor",
              config=cfg(optimizer=LUT, learning_rate=0.1, power_t=0.0),
              steps=[step("learn", 0.5, lr=one, importance=0.5), step("learn", 0.49375027, lr=one, importance=0.5),
                     step("learn", 0.4875807, lr=one, importance=0.5)]))

# ------------------------------------------------------------------ persistence.rs (LR)
S.append(dict(name="save_load_and_test_mode_lr", source="persistence.rs:250-313", wiring="regressor",
              config=cfg(optimizer=FLEX, learning_rate=0.1, power_t=0.5, init_acc_gradient=0.0),
              steps=[step("learn", 0.5, lr=two), step("learn", 0.45016602, lr=two),
                     step("learn", 0.41731137, lr=two, update=False),
                     step("learn", 0.41731137, lr=two, update=False), step("predict", 0.41731137, lr=two)]))

# ------------------------------------------------------------------ block_ffm.rs (FFM block -> sigmoid)
# spredict2 = BlockFFM::forward (inference numerics), slearn2 = forward_backward (block_helpers.rs:162-216).


def ffm_cfg(k, F, opt, lr=0.1, pt=0.0):
    return cfg(optimizer=opt, learning_rate=lr, power_t=pt, ffm_learning_rate=lr, ffm_power_t=pt, ffm_k=k,
               ffm_num_fields=F)


def pl(fb, a, b=None, cmp_p="eps", upd=True):
    """spredict2 asserting a, then slearn2 asserting b (default a)."""
    b = a if b is None else b
    return [step("predict", a, ffm=fb, cmp=cmp_p), step("forward_backward", b, ffm=fb, update=upd)]


k = 1
f1 = [(1, 1.0, 0)]
f2 = [(1, 1.0, 0), (100, 1.0, k)]
f2v = [(1, 2.0, 0), (100, 2.0, k)]
S.append(dict(name="test_ffm_k1/single_field", source="block_ffm.rs:1238-1269", wiring="ffm_only",
              config=ffm_cfg(1, 2, LUT), steps=[step("predict", 0.5, ffm=f1, cmp="eps"),
                                                step("forward_backward", 0.5, ffm=f1, cmp="eps")]))
S.append(dict(name="test_ffm_k1/two_fields_flex", source="block_ffm.rs:1271-1299", wiring="ffm_only",
              config=ffm_cfg(1, 2, FLEX), ffm_fill=1.0, steps=pl(f2, 0.7310586) + pl(f2, 0.7024794)))
S.append(dict(name="test_ffm_k1/two_fields_values_lut", source="block_ffm.rs:1301-1327", wiring="ffm_only",
              config=ffm_cfg(1, 2, LUT), ffm_fill=1.0,
              steps=pl(f2v, 0.98201376, cmp_p="eq") + pl(f2v, 0.81377685, cmp_p="eq")))
k = 4
g1 = [(1, 1.0, 0)]
g2 = [(1, 1.0, 0), (100, 1.0, k)]
g2v = [(1, 2.0, 0), (100, 2.0, k)]
S.append(dict(name="test_ffm_k4/single_field", source="block_ffm.rs:1449-1478", wiring="ffm_only",
              config=ffm_cfg(4, 2, LUT), steps=pl(g1, 0.5, cmp_p="eq") + pl(g1, 0.5, cmp_p="eq")))
S.append(dict(name="test_ffm_k4/two_fields_flex", source="block_ffm.rs:1480-1505", wiring="ffm_only",
              config=ffm_cfg(4, 2, FLEX), ffm_fill=1.0,
              steps=pl(g2, 0.98201376, cmp_p="eq") + pl(g2, 0.96277946, cmp_p="eq")))
S.append(dict(name="test_ffm_k4/two_fields_values_lut", source="block_ffm.rs:1507-1531", wiring="ffm_only",
              config=ffm_cfg(4, 2, LUT), ffm_fill=1.0,
              steps=pl(g2v, 0.9999999, cmp_p="eq") + pl(g2v, 0.99685884, cmp_p="eq")))
k = 1
mv = [(1, 1.0, 0), (3000, 1.0, 0), (100, 2.0, k)]
S.append(dict(name="test_ffm_multivalue", source="block_ffm.rs:1657-1700", wiring="ffm_only",
              config=ffm_cfg(1, 2, LUT), ffm_fill=1.0,
              steps=pl(mv, 0.9933072) + pl(mv, 0.9395168, upd=False) + pl(mv, 0.9395168, upd=False)))
k = 4
mv4 = [(1, 1.0, 0), (3000, 1.0, 0), (100, 2.0, k)]
S.append(dict(name="test_ffm_multivalue_k4_nonzero_powert", source="block_ffm.rs:1777-1817", wiring="ffm_only",
              config=cfg(optimizer=LUT, ffm_k=4, ffm_num_fields=2), ffm_fill=1.0,
              steps=pl(mv4, 1.0, cmp_p="eq") + [step("predict", 0.9949837, ffm=mv4),
                                                step("forward_backward", 0.9949837, ffm=mv4, update=False),
                                                step("forward_backward", 0.9949837, ffm=mv4, update=False)]))
k = 1
m3 = [(1, 1.0, 0), (5, 1.0, k), (100, 1.0, 2 * k)]
mid = [(5, 1.0, k)]
S.append(dict(name="test_ffm_missing_field", source="block_ffm.rs:1881-1943", wiring="ffm_only",
              config=ffm_cfg(1, 3, FLEX), ffm_fill=1.0,
              steps=[step("predict", 0.95257413, ffm=m3, cmp="eps"),
                     step("forward_backward", 0.95257413, ffm=m3, update=False),
                     step("predict", 0.5, ffm=mid),
                     step("forward_backward", 0.62245935, ffm=mid, stale=True, current_code=0.5,
                          note="This #[ignore]d assertion predates the current code: with every weight 1.0 and a "
                               "single feature in the middle field, block_ffm.rs:236-245 subtracts the feature's "
                               "own contribution (contra - w*v = 0) so forward_backward yields logit 0 -> 0.5, "
                               "the same value spredict2 is asserted to give one line earlier (block_ffm.rs:1941). "
                               "0.62245935 = sigmoid(0.5) is the self-interaction an older revision kept.")]))
# second half of test_ffm_missing_field_with_cache: fields 0 and 2 only (block_ffm.rs:2016-2035)
m02 = [(1, 1.0, 0), (100, 1.0, 2 * k)]
S.append(dict(name="test_ffm_missing_field_with_cache/tail", source="block_ffm.rs:1946-2036", wiring="ffm_only",
              config=ffm_cfg(1, 3, FLEX), ffm_fill=1.0,
              steps=[step("predict", 0.95257413, ffm=m3, cmp="eps"),
                     step("forward_backward", 0.95257413, ffm=m3, update=False),
                     step("predict", 0.7310586, ffm=m02), step("forward_backward", 0.7310586, ffm=m02)]))

# ------------------------------------------------------------------ persistence.rs (full regressor: LR + FFM + Triangle)
pcfg = cfg(optimizer=FLEX, learning_rate=0.1, power_t=0.0, ffm_k=1, ffm_num_fields=2, ffm_power_t=0.0,
           ffm_learning_rate=0.1)
pmv = [(1, 1.0, 0), (3000, 1.0, 0), (100, 2.0, 1)]
S.append(dict(name="save_load_and_test_mode_ffm", source="persistence.rs:341-418", wiring="regressor", config=pcfg,
              ffm_fill=1.0,
              steps=[step("learn", 0.9933072, ffm=pmv), step("learn", 0.9395168, ffm=pmv, update=False, cmp="eps"),
                     step("predict", 0.9395168, ffm=pmv, cmp="eps")]))
lr1 = [(52, 0.5, 0), (2, 1.0, 0)]
ff1 = [(1, 0.5, 0), (3000, 1.0, 0), (101, 2.0, 1)]
lr2 = [(1, 1.0, 0), (2, 1.0, 0)]
ff2 = [(1, 1.0, 0), (3000, 1.0, 0), (100, 2.0, 1)]
S.append(dict(name="test_hogwild_load/re_1", source="persistence.rs:436-560", wiring="regressor", config=pcfg,
              ffm_fill=1.0,
              steps=[step("learn", 0.97068775, lr=lr1, ffm=ff1), step("learn", 0.8922257, lr=lr1, ffm=ff1, update=False),
                     step("predict", 0.8922257, lr=lr1, ffm=ff1),
                     step("learn", 0.98559695, lr=lr2, ffm=ff2, update=False),
                     step("predict", 0.98559695, lr=lr2, ffm=ff2)]))
S.append(dict(name="test_hogwild_load/re_2", source="persistence.rs:436-560", wiring="regressor", config=pcfg,
              ffm_fill=1.0,
              steps=[step("learn", 0.9933072, lr=lr2, ffm=ff2), step("learn", 0.92719215, lr=lr2, ffm=ff2, update=False),
                     step("predict", 0.92719215, lr=lr2, ffm=ff2),
                     step("learn", 0.93763095, lr=lr1, ffm=ff1, update=False),
                     step("predict", 0.93763095, lr=lr1, ffm=ff1)]))

# ------------------------------------------------------------------ optimizer.rs:170-226
OPT = dict(
    source="optimizer.rs:170-226",
    sgd=[dict(lr=0.15, g=0.1, expect_expr="0.1f32*0.15f32")],
    flex=[dict(lr=0.15, power_t=0.4, acc=0.9, g=0.1, expect=0.015576674, acc_expr="0.9+0.1*0.1"),
          dict(lr=0.15, power_t=0.4, acc=0.0, g=0.1, expect=0.09464361, acc_expr="0.1*0.1"),
          dict(lr=0.15, power_t=0.4, acc=0.0, g=0.0, expect=None, acc_expr="0.0")],
    lut=[dict(lr=0.15, power_t=0.4, init_acc=0.0, acc=0.9, g=0.1, expect=0.015607622, acc_expr="0.9+0.1*0.1"),
         dict(lr=0.15, power_t=0.4, init_acc=0.0, acc=0.0, g=0.1, expect=0.09375872, acc_expr="0.1*0.1"),
         dict(lr=0.15, power_t=0.4, init_acc=0.0, acc=0.0, g=0.0, expect=0.0, acc_expr="0.0")],
    comparison=dict(source="optimizer.rs:229-268", lr=0.15, power_t=0.4, init_acc=0.0,
                    gradients=[-1.0, -0.9, -0.1, -0.00001, 0.0, 0.00001, 0.1, 0.5, 0.9, 1.0],
                    accumulations=[0.0000000001, 0.00001, 0.1, 0.5, 1.1, 2.0, 20.0, 200.0, 2000.0, 200000.0, 2000000.0],
                    max_rel_err=0.05),
)

# ------------------------------------------------------------------ block_misc.rs:942-969 (triangle)
TRI = dict(source="block_misc.rs:942-969", width=2, input=[2.0, 4.0, 4.0, 5.0], forward=[2.0, 8.0, 5.0],
           backward=[2.0, 8.0, 8.0, 5.0])

# ------------------------------------------------------------------ feature_buffer.rs:374-797 (translation)
NOF = 0x80000000  # parser.rs:19 NO_FEATURES
NS = 0x80000000   # parser.rs:17 IS_NOT_SINGLE_MASK
M31 = 0x7FFFFFFF


def nd(s, e):
    return (s << 16) + e


def f32bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


def rec(*words):
    return [100, 1, f32bits(1.0)] + list(words)  # add_header, feature_buffer.rs:349-353


def tr(name, source, combos, fields, const, ffm_k, cases):
    return dict(name=name, source=source, combos=combos, fields=fields, add_constant_feature=const, bit_precision=18,
                ffm_k=ffm_k, ffm_bit_precision=18, cases=cases)


c0 = [[[[0, 0]], 1.0]]  # one combo: namespace 0 categorical, weight 1.0
TR = [
    tr("test_constant", "feature_buffer.rs:381-403", c0, [], True, 0,
       [dict(record=rec(NOF), lr=[[116060, 1.0, 1]], ffm=[])]),
    tr("test_single_once", "feature_buffer.rs:405-451", c0, [], False, 0,
       [dict(record=rec(NOF), lr=[], ffm=[]), dict(record=rec(0xfea), lr=[[0xfea, 1.0, 0]], ffm=[]),
        dict(record=rec(NS | nd(4, 8), 0xfea, f32bits(1.0), 0xfeb, f32bits(1.0)),
             lr=[[0xfea, 1.0, 0], [0xfeb, 1.0, 0]], ffm=[])]),
    tr("test_single_twice", "feature_buffer.rs:453-503", [[[[0, 0]], 1.0], [[[1, 0]], 1.0]], [], False, 0,
       [dict(record=rec(NOF, NOF), lr=[], ffm=[]), dict(record=rec(0xfea, NOF), lr=[[0xfea, 1.0, 0]], ffm=[]),
        dict(record=rec(0xfea, 0xfeb), lr=[[0xfea, 1.0, 0], [0xfeb, 1.0, 1]], ffm=[])]),
    tr("test_double_vowpal", "feature_buffer.rs:507-543", [[[[0, 0], [1, 0]], 1.0]], [], False, 0,
       [dict(record=rec(NOF, NOF), lr=[], ffm=[]), dict(record=rec(123456789, NOF), lr=[], ffm=[]),
        dict(record=rec(2988156968 & M31, 2422381320 & M31, NOF), lr=[[208368, 1.0, 0]], ffm=[])]),
    tr("test_single_with_weight_vowpal", "feature_buffer.rs:545-567", [[[[0, 0]], 2.0]], [], False, 0,
       [dict(record=rec(0xfea), lr=[[0xfea, 2.0, 0]], ffm=[])]),
    tr("test_ffm_empty", "feature_buffer.rs:560-570", [], [[]], False, 1, [dict(record=rec(0xfea), lr=[], ffm=[])]),
    tr("test_ffm_one", "feature_buffer.rs:572-590", [], [[[0, 0]]], False, 1,
       [dict(record=rec(0xfea), lr=[], ffm=[[0xfea, 1.0, 0]])]),
    tr("test_ffm_two_fields", "feature_buffer.rs:592-640", [], [[[0, 0]], [[0, 0], [1, 0]]], False, 1,
       [dict(record=rec(NS | nd(5, 9), 0xfec, 0xfea, f32bits(2.0), 0xfeb, f32bits(3.0)), lr=[],
             ffm=[[0xfea, 2.0, 0], [0xfeb, 3.0, 0], [0xfea, 2.0, 1], [0xfeb, 3.0, 1], [0xfec, 1.0, 1]])]),
    tr("test_ffm_three_fields/k1", "feature_buffer.rs:642-693", [], [[[0, 0]], [[0, 0], [1, 0]], [[1, 0]]], False, 1,
       [dict(record=rec(NS | nd(5, 9), 0x1, 0xfff, f32bits(2.0), 0xfeb, f32bits(3.0)), lr=[],
             ffm=[[0xfff, 2.0, 0], [0xfeb, 3.0, 0], [0xfff, 2.0, 1], [0xfeb, 3.0, 1], [0x1, 1.0, 1], [0x1, 1.0, 2]])]),
    tr("test_ffm_three_fields/k3", "feature_buffer.rs:694-742", [], [[[0, 0]], [[0, 0], [1, 0]], [[1, 0]]], False, 3,
       [dict(record=rec(NS | nd(5, 9), 0x1, 0xfff, f32bits(2.0), 0xfeb, f32bits(3.0)), lr=[],
             ffm=[[0xffc, 2.0, 0], [0xfe8, 3.0, 0], [0xffc, 2.0, 3], [0xfe8, 3.0, 3], [0x0, 1.0, 3], [0x0, 1.0, 6]])]),
    tr("test_single_namespace_float", "feature_buffer.rs:763-796", [[[[1, 1]], 1.0]], [], False, 0,
       [dict(record=rec(NOF, nd(6, 10) | NS, NOF, 0xffc & M31, f32bits(3.0), 0xffa & M31, f32bits(4.0)),
             lr=[[0xffc, 1.0, 0], [0xffa, 1.0, 0]], ffm=[])]),
]

# ------------------------------------------------------------------ parser.rs:474-1183 (murmur3 as used by the parser)
# feature hash = murmur3_32(feature_name, seed = murmur3_32(namespace_vwname, 0)) & MASK31 (parser.rs:82-87, 382-385).
HASH = dict(
    source="parser.rs:474-1183",
    cases=[
        dict(ns="A", feature="a", hash=2988156968 & M31, at="parser.rs:489-502 ('1 |A a')"),
        dict(ns="B", feature="b", hash=2422381320 & M31, at="parser.rs:545-556 ('-1 |B b')"),
        dict(ns="A", feature="b", hash=3529656005 & M31, at="parser.rs:558-573 ('1 |A a b')"),
        dict(ns="A", feature="c", hash=906509 & M31, at="parser.rs:709-727 ('1 |A a b:2.0 c:3.0')"),
        dict(ns="B", feature="3", hash=1775699190 & M31, at="parser.rs:872-884 ('-1 |B 3')"),
        dict(ns="B", feature="4", hash=382082293 & M31, at="parser.rs:909-925 ('-1 |B 3 4')"),
        dict(ns="AA", feature="a", hash=292540976 & M31, at="parser.rs:1033-1045 ('1 |AA a')"),
    ],
)

out = dict(_doc=__doc__, scenarios=S, optimizer=OPT, triangle=TRI, translation=TR, hash=HASH)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(S), "scenarios,", len(TR), "translation cases")
