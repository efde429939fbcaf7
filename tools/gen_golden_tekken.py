#!/usr/bin/env python3
"""Golden split vectors for SURVEY section 8 row f-3 (groundwork): the `pattern` of Mistral's tekken.json (literal of
reference tests/test_small_vocab.rs:62, which the reference ignores, src/tekkenizer.rs:74) evaluated by the Python
`regex` module -- an engine independent of the reference and of this repository.
    python tools/gen_golden_tekken.py  ->  tests/golden/split_vectors_tekken.json"""
import json
import os
import random
import sys

import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg          # the same hand-written cases and alphabet as the hard-coded pattern's vectors
import synth_vocab as sv

R = regex.compile(sv.MISTRAL_PATTERN)
CASE = ["HelloWorld", "helloWORLD", "HELLOworld", "ABc", "ABcD", "AXB", "camelCaseString XMLHttpRequest iPhone",
        "ǅungla ǈubav", "aʰB 中A A中 中a", "éÉ ́x", "path/to/file.txt //\n/x", "1a2B33",
        "it's IT'S don't", "xªyº", "ÉCOLE école École éCOLE", "ÁB̈c", " ́abc", "́",
        "UPPER lower Title mIxEd", "a/b/c\n/d", "!!/\r\n/", "ΑβΓδ ЖжЖ"]
ALPHA = gg.ALPHA + ["A", "B", "c", "É", "ǅ", "ʰ", "ª", "/", "̈", "Α", "β"]


def main():
    rng = random.Random(0x7E44E3)
    cases = list(gg.HAND) + CASE
    for _ in range(900):
        cases.append("".join(rng.choice(ALPHA) for _ in range(rng.randint(1, 48))))
    for _ in range(40):
        cases.append("".join(rng.choice(ALPHA) for _ in range(rng.randint(100, 400))))
    out = []
    for t in cases:
        p = [m.group() for m in R.finditer(t)]
        assert "".join(p) == t, repr(t)
        starts, pos = [], 0
        for x in p:
            starts.append(pos)
            pos += len(x.encode("utf-8"))
        out.append({"text": t, "starts": starts})
    with open(os.path.join(ROOT, "tests", "golden", "split_vectors_tekken.json"), "w") as f:
        json.dump({"engine": "python regex %s" % regex.__version__, "pattern": sv.MISTRAL_PATTERN, "cases": out}, f,
                  ensure_ascii=True, indent=0)
    print(len(out), "cases")


if __name__ == "__main__":
    main()
