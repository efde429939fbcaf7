#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

1. split_vectors.json -- piece boundaries of the hard-coded pattern (reference
   src/tekkenizer.rs:123) computed with the Python `regex` module: an engine INDEPENDENT of both
   the reference and this repo's code (the reference's own engine, fancy-regex inside
   tiktoken-rs, cannot be run here: no Rust toolchain).  Hand-picked cases (SURVEY App. A.4 T12
   and the reference's edge-case inputs, tests/test_integration.rs:204-216) plus seeded random
   strings over an alphabet whose L/N/White_Space status is stable across Unicode 13-17.
2. reference_vectors.json -- DATA copied from the reference's tests: the 20 (text, ids) pairs of
   tests/test_tokenizer_output.rs, the decode known-answer of
   tests/test_rust_tokenizer.rs:16-19 and asset facts.  Their ids need the missing tekken.json;
   what can be checked without it is derived in tests/test_reference_vectors.py.
"""
import json
import os
import random

import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERN = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
R = regex.compile(PATTERN)

HAND = [
    "", " ", "\n", "\t", "   \n\t   ", "\U0001f680", "a" * 100, "Hello\x00World", "Line1\nLine2\rLine3\r\nLine4",
    "Hello, world!", "The quick brown fox jumps over the lazy dog.", "   whitespace   handling   ",
    "Another test case with numbers: 123, 456, 789.", "Special characters: @#$%^&*()_+-={}[]|\\:;\"'<>,.?/",
    "Mixed CaSe WoRdS", "it's IT'S x'\u017f", "a\x0bb", "a  1", "x\t\ty", "1234", "\u00b2\u2167\u00bd\u0663",
    "hi !!\n\nyo", "  \n  \n  x", "'abc", "'sabc", "a b", "end   ", "<s>[INST] hi [/INST]</s>",
    "caf\u00e9 na\u00efve \u4f60\u597d\u4e16\u754c \u043f\u0440\u0438\u0432\u0435\u0442 \u0645\u0631\u062d\u0628\u0627 \U0001f600\U0001f389",
    "x = 3.14159; y += 2**10  # comment\n\tif x >= y:\r\n\t\treturn 'ok'",
    "tabs\tand\u00a0nbsp\u3000ideographic\u2028linesep", "we're they've I'll he'd she's don't I'm 'RE 'Ve 'LL",
    "12345678 1 22 333 4444", "\u0661\u0662\u0663\u0664 \uff11\uff12\uff13\uff14\uff15", "!!!\n\n\n???   ...\r\n",
    "a" * 63 + " " + "b" * 64 + "\n" * 70 + " " * 130 + "x", "\n" * 5 + " " * 5, "e\u0301 o\u0308",
]

ALPHA = ["a", "S", "s", "t", "r", "e", "E", "l", "L", "v", "m", "d", "x", "Z", "1", "2", "9", "'", "!", "-", ".", ",",
         "(", " ", " ", " ", "\n", "\r", "\t", "\u017f", "\u00e9", "\u4e2d", "\ud55c", "\u0e01", "\u0436", "\u0663",
         "\uff11", "\u00b2", "\u00a0", "\u2028", "\u3000", "\U0001f680", "\u2200", "\u0301", "\x00", "\x0b"]


def pieces(text):
    return [m.group() for m in R.finditer(text)]


def main():
    rng = random.Random(0x7E44E2)
    cases = list(HAND)
    for _ in range(700):
        n = rng.randint(1, 48)
        cases.append("".join(rng.choice(ALPHA) for _ in range(n)))
    for _ in range(40):
        n = rng.randint(100, 400)
        cases.append("".join(rng.choice(ALPHA) for _ in range(n)))
    out = []
    for t in cases:
        p = pieces(t)
        assert "".join(p) == t, repr(t)
        starts, pos = [], 0
        for x in p:
            starts.append(pos)
            pos += len(x.encode("utf-8"))
        out.append({"text": t, "starts": starts})
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "golden", "split_vectors.json"), "w") as f:
        json.dump({"engine": "python regex %s" % regex.__version__, "pattern": PATTERN, "cases": out}, f,
                  ensure_ascii=True, indent=0)

    ref = {
        "source": "reference tests/test_tokenizer_output.rs (expected_tokens literals), tests/test_rust_tokenizer.rs:16-19",
        "asset": {"vocab_size": 131072, "version": "v7", "num_special_tokens": 1000},
        "encode": [
            ["Hello, world!", [22177, 1044, 4304, 1033]],
            ["The quick brown fox jumps over the lazy dog.", [1784, 7586, 22980, 94137, 72993, 2136, 1278, 42757, 10575, 1046]],
            ["This is a test of the Mistral Tekken tokenizer.", [4380, 1395, 1261, 2688, 1307, 1278, 42301, 2784, 47213, 3569, 128405, 1046]],
            ["Emojis and unicode characters work too!", [5969, 3659, 1275, 1321, 79219, 11084, 2196, 4382, 1033]],
            ["Hello", [22177]], ["world", [34049]], ["test", [4417]], ["a", [1097]], ["the", [3265]], ["Python", [46728]],
            ["Rust", [1082, 1616]], ["tokenizer", [15017, 7463]], ["encoding", [47130]], ["decoding", [18888, 7967]],
            ["comparison", [69959, 3693]], ["Simple sentence.", [28683, 19286, 1046]],
            ["Another test case with numbers: 123, 456, 789.",
             [18661, 2688, 2937, 1454, 8091, 1058, 1032, 1049, 1050, 1051, 1044, 1032, 1052, 1053, 1054, 1044, 1032, 1055, 1056, 1057, 1046]],
            ["Special characters: @#$%^&*()_+-={}[]|\\:;\"'<>,.?/",
             [40124, 11084, 1058, 2126, 1035, 1036, 1037, 1094, 1038, 1042, 1690, 1095, 104799, 3181, 1125, 4344, 17743, 1058, 36211, 96726, 24482, 1046, 1063, 1047]],
            ["Mixed CaSe WoRdS", [1077, 5422, 10645, 3201, 18739, 1082, 1100, 1083]],
            ["   whitespace   handling   ", [1256, 81024, 1256, 21490, 1293]],
        ],
        "decode": {
            "ids": [4998, 1878, 1044, 2036, 20574, 20999, 1044, 4237, 1605, 2549, 2143, 6816, 1710, 1653, 1394, 1636, 1044, 4237, 2549, 1636, 1710, 1653, 1394, 2143, 6816, 1046, 2],
            "text": "And so, my fellow Americans, ask not what your country can do for you, ask what you can do for your country."},
        "small_vocab": {"source": "reference tests/test_small_vocab.rs:11-67 (construction); expected ids derived by hand, SURVEY App. B.3",
                        "extra_tokens": ["hello", "world"], "num_special": 10, "vocab_size": 268,
                        "cases": [["hello world", True, True, [1, 266, 42, 129, 121, 124, 118, 110, 2]],
                                  ["world", False, False, [267]],
                                  ["worldhello", False, False, [129, 121, 124, 118, 110, 114, 111, 118, 118, 121]],
                                  ["", True, True, [1, 2]], ["", False, False, []]]},
    }
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json"), "w") as f:
        json.dump(ref, f, ensure_ascii=True, indent=1)
    print("wrote %d split cases" % len(out))


if __name__ == "__main__":
    main()
