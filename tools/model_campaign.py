#!/usr/bin/env python3
"""CPU hunt for split-rule bugs (test infrastructure): the lane-layout rules as the flat kernel evaluates them, region by region
(tools/flat_split_model.py: flat_split_chunked for the hard-coded pattern, flat_split_chunked_tekken for the JSON pattern), against the
oracle's sequential matchers on random documents made of RUNS of chars of every class -- multi-byte ones included, so that region
starts fall inside chars and runs cover halos.  ~300 cases a second and core; round 4: 2.6 M cases found one difference (JSON pattern:
a chain of tail chars, punctuation and marks from below the region), after the GPU fuzz had found the first of the kind.

    python tools/model_campaign.py SEED SECONDS        (prints "campaign ok" or writes the failing case to /tmp and exits 1)"""
import os
import pickle
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")]
import flat_split_model as fm  # noqa: E402
import tk_oracle  # noqa: E402

ALPHABET = ["a", "A", "1", " ", "\n", "\r", "\t", "!", "'", "/", "s", "t", "re", "ll", "d", "é", "٣", "３", "　", "…", "中",
            "\U0001f680", " ", "́", "ſ", "-", "ǅ", "ʰ"]


def check(docs, region, tekken):
    data = b"".join(docs)
    offs = [0]
    for d in docs:
        offs.append(offs[-1] + len(d))
    if tekken:
        starts, deferred = fm.flat_split_chunked_tekken(data, offs, region=region)
        split = tk_oracle.split_tekken
    else:
        starts, deferred = fm.flat_split_chunked(data, offs, region=region)
        split = tk_oracle.split
    exp = []
    for i, d in enumerate(docs):
        if i not in deferred:
            exp += [offs[i] + s for s in split(d)]
    return starts == exp


def main():
    seed0, seconds = int(sys.argv[1]), float(sys.argv[2])
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        rng = random.Random(seed0 * 1000003 + n)
        docs = ["".join(rng.choice(ALPHABET) * rng.choice([1, 1, 1, 2, 3, 5, 11, 22, 31, 33, 40, 70]) for _ in range(rng.randint(0, 40))).encode()
                for _ in range(rng.randint(1, 6))]
        region, tekken = rng.choice([256, 256, 2048]), rng.random() < 0.4
        if not check(docs, region, tekken):
            path = "/tmp/model_campaign_fail_%d_%d.pkl" % (seed0, n)
            pickle.dump((docs, region, tekken), open(path, "wb"))
            print("FAIL seed", seed0, "case", n, "region", region, "JSON pattern" if tekken else "default pattern", "->", path)
            sys.exit(1)
        n += 1
    print("campaign ok: seed", seed0, n, "cases")


if __name__ == "__main__":
    main()
