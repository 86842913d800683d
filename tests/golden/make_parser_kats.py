"""Builds tests/golden/parser_kats.json: the known-answer vectors of the reference's text parser and namespace map,
transcribed as DATA (input line -> expected u32 record / error message) from the reference's own #[test]s:
  parser.rs:474-859  test_vowpal            parser.rs:861-1047 test_float_namespaces
  parser.rs:1049-1094 test_multibyte_namespaces   parser.rs:1096-1183 test_cache* (next_vowpal_with_cache)
  vwmap.rs:159-236   test_simple, test_f32
Run from the repo root:  python tests/golden/make_parser_kats.py"""
import json
import os
import struct

ONE = 1065353216          # FLOAT32_ONE  (parser.rs:21)
NOT_SINGLE = 1 << 31      # IS_NOT_SINGLE_MASK
MASK31 = NOT_SINGLE - 1
NOF = NOT_SINGLE          # NO_FEATURES
NO_LABEL = 0xff


def f(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def nd(a, b):
    return ((a << 16) + b) | NOT_SINGLE


NAN = 0x7fc00000  # f32::NAN.to_bits()
ha, hb_, hc, hB = 2988156968 & MASK31, 3529656005 & MASK31, 906509 & MASK31, 2422381320 & MASK31

ABC = "\nA,featureA\nB,featureB\nC,featureC\n"
ABC_F32 = "\nA,featureA\nB,featureB,f32\nC,featureC\n"
ABC_F32_SKIP = "\nA,featureA\nB,featureB,f32\nC,featureC\n_namespace_skip_prefix,1\n"
AABBCC = "\nAA,featureA\nBB,featureB\nCC,featureC\n"

single_a = [6, 1, ONE, ha, NOF, NOF]
groups = [
    {"name": "test_vowpal", "vwmap": ABC, "cases": [
        {"line": "1 |A a\n", "record": single_a},
        {"line": "1 |A a \n", "record": single_a},
        {"line": "1  |A a\n", "record": single_a},
        {"line": "1 |A  a\n", "record": single_a},
        {"line": "-1 |B b\n", "record": [6, 0, ONE, NOF, hB, NOF]},
        {"line": "1 |A a b\n", "record": [10, 1, ONE, nd(6, 10), NOF, NOF, ha, ONE, hb_, ONE]},
        {"line": "-1 |A a |B b\n", "record": [6, 0, ONE, ha, hB, NOF]},
        {"line": "-1 |A a  |B b\n", "record": [6, 0, ONE, ha, hB, NOF]},
        {"line": "1 |UNDECLARED_NAMESPACE a\n",
         "error": "Feature name was not predeclared in vw_namespace_map.csv: UNDECLARED_NAMESPACE"},
        {"line": "1 |A:1.0 a\n", "record": single_a},
        {"line": "1 |A:not_a_parsable_number a\n", "error": "Failed parsing namespace weight: not_a_parsable_number"},
        {"line": "1 |A:1:1 a\n", "error": "Failed parsing namespace weight: 1:1"},
        {"line": "1 |A:2.0 a\n", "record": [8, 1, ONE, nd(6, 8), NOF, NOF, ha, f(2.0)]},
        {"line": "1 |A a:2.0\n", "record": [8, 1, ONE, nd(6, 8), NOF, NOF, ha, f(2.0)]},
        {"line": "1 |A a:2.0 b:3.0\n", "record": [10, 1, ONE, nd(6, 10), NOF, NOF, ha, f(2.0), hb_, f(3.0)]},
        {"line": "1 |A:3 a:2.0\n", "record": [8, 1, ONE, nd(6, 8), NOF, NOF, ha, f(6.0)]},
        {"line": "1 |A a:2x0\n", "error": "Failed parsing feature weight: 2x0"},
        {"line": "1 |A a b:2.0 c:3.0\n",
         "record": [12, 1, ONE, nd(6, 12), NOF, NOF, ha, f(1.0), hb_, f(2.0), hc, f(3.0)]},
        {"line": "|A a\n", "record": [6, NO_LABEL, ONE, ha, NOF, NOF]},
        {"line": "", "record": []},
        {"line": "flush", "command": "flush"},
        {"line": "$1", "error": "Cannot parse an example"},
        {"line": "1 -0.1 |A a\n", "error": "Example importance cannot be negative: -0.1! "},
        {"line": "1 fdsa |A a\n", "error": "Failed parsing example importance: fdsa"},
        {"line": "1 0.1 |A a\n", "record": [6, 1, f(0.1), ha, NOF, NOF]},
        {"line": "1  0.1  |A  a \n", "record": [6, 1, f(0.1), ha, NOF, NOF]},
        {"line": "hogwild_load /path/to/filename", "command": "hogwild_load", "filename": "/path/to/filename"},
        {"line": "hogwild_load   /path/to/filename", "command": "hogwild_load", "filename": "/path/to/filename"},
        {"line": "hogwild_load   /path/to/filename  ", "command": "hogwild_load", "filename": "/path/to/filename"},
        {"line": "hogwild_load", "error": "Cannot parse an example"},
        {"line": "hogwild_load ", "error": "Cannot parse an example"},
    ]},
    {"name": "test_float_namespaces/categorical", "vwmap": ABC, "cases": [
        {"line": "-1 |B 3\n", "record": [6, 0, ONE, NOF, 1775699190 & MASK31, NOF]},
    ]},
    {"name": "test_float_namespaces/f32", "vwmap": ABC_F32, "cases": [
        {"line": "-1 |B 3\n", "record": [8, 0, ONE, NOF, nd(6, 8), NOF, 1775699190 & MASK31, f(3.0)]},
        {"line": "-1 |B 3 4\n",
         "record": [10, 0, ONE, NOF, nd(6, 10), NOF, 1775699190 & MASK31, f(3.0), 382082293 & MASK31, f(4.0)]},
        {"line": "-1 |B not_a_number\n",
         "error": "Failed parsing feature value to float (for float namespace): not_a_number"},
        {"line": "-1 |B 3 4\n",
         "record": [10, 0, ONE, NOF, nd(6, 10), NOF, 1775699190 & MASK31, f(3.0), 382082293 & MASK31, f(4.0)]},
        {"line": "-1 |B 3:3\n",
         "error": "Namespaces that are f32 can not have weight attached neither to namespace nor to a single feature "
                  "(basically they can' use :weight syntax"},
        {"line": "-1 |B:3 3\n",
         "error": "Namespaces that are f32 can not have weight attached neither to namespace nor to a single feature "
                  "(basically they can' use :weight syntax"},
    ]},
    {"name": "test_float_namespaces/skip_prefix", "vwmap": ABC_F32_SKIP, "cases": [
        {"line": "-1 |B B3\n", "record": [8, 0, ONE, NOF, nd(6, 8), NOF, 1416737454 & MASK31, f(3.0)]},
        {"line": "-1 |B B\n", "record": [8, 0, ONE, NOF, nd(6, 8), NOF, 25602353 & MASK31, NAN]},
        {"line": "-1 |B BNONE\n", "record": [8, 0, ONE, NOF, nd(6, 8), NOF, 1846432377 & MASK31, NAN]},
    ]},
    {"name": "test_multibyte_namespaces", "vwmap": AABBCC, "cases": [
        {"line": "1 |AA a\n", "record": [6, 1, ONE, 292540976 & MASK31, NOF, NOF]},
        {"line": "1 |AA:3 a:2.0\n", "record": [8, 1, ONE, nd(6, 8), NOF, NOF, 292540976 & MASK31, f(6.0)]},
    ]},
]

full = [8, 255, 1065353216, 2147876872, 1123906636, 2147483648, 292540976, 1086324736]
with_cache = {"name": "test_cache", "vwmap": AABBCC, "cases": [
    {"line": "|BB b |AA:3 a:2.0 \n", "record": full},
    # next_vowpal_with_size: (record, size without the newline); next_vowpal_with_cache(cached text, new text)
    {"line": "|BB b \n", "record": [6, 255, 1065353216, 2147483648, 1123906636, 2147483648], "size": 6},
    {"cached": "|BB b ", "line": "|AA:3 a:2.0 \n", "record": full},
    {"line": "|BB b |AA:3 a:2.0 \n", "record": full, "size": 18},
    {"cached": "|BB b |AA:3 a:2.0 ", "line": "", "record": full},
    {"cached": "", "line": "|BB b |AA:3 a:2.0 \n", "record": full},
]}
groups.append(with_cache)

vwmap_kats = [
    {"csv": ABC, "skip_prefix": 0, "entries": [["A", "featureA", 0, "Categorical"], ["B", "featureB", 1, "Categorical"],
                                                ["C", "featureC", 2, "Categorical"]]},
    {"csv": "A,featureA,f32\n_namespace_skip_prefix,2", "skip_prefix": 2, "entries": [["A", "featureA", 0, "F32"]]},
    {"csv": "A,featureA,blah\n",
     "error": 'Unknown type used for the feature in vw_namespace_map.csv: "blah". Only "f32" is possible.'},
]

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parser_kats.json")
with open(out, "w") as fh:
    json.dump({"source": "reference parser.rs / vwmap.rs #[test] assertions, transcribed as data",
               "groups": groups, "vwmap": vwmap_kats}, fh, indent=1)
print("wrote", out, sum(len(g["cases"]) for g in groups), "parser cases")
