"""Shared helpers of the test-suite."""
import functools
import os
import pickle
import random

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@functools.lru_cache(maxsize=None)
def small_trained_vocab(n_merges=3000, n_ranks=6000, seed=0x7E44E2):
    """256 bytes + BPE merges trained on a small synthetic sample, grown to n_ranks tokens."""
    import synth_vocab as sv
    cache = os.path.join(ROOT, "assets", "test_vocab_%d_%d.pkl" % (n_merges, n_ranks))
    if os.path.exists(cache):
        with open(cache, "rb") as f:
            return pickle.load(f)
    toks = sv.build_tokens(n_ranks, n_merges, 1500, 300, seed)
    v = {"tokens": toks, "num_special": 1000, "bos": 1, "eos": 2}
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    with open(cache, "wb") as f:
        pickle.dump(v, f)
    return v


EDGE_DOCS = [b"", b" ", b"\n", b"\t", b"   \n\t   ", "\U0001f680".encode(), b"a" * 1000, b"Hello\x00World",
             b"Line1\nLine2\rLine3\r\nLine4", b"a", b"ab", b"'s", b"x's", b" " * 70, b"\n" * 70 + b"x", b"z" * 64,
             b"z" * 65, b"q" * 63 + b" ", ("中" * 40).encode(), ("é" * 33).encode(), b"1" * 100,
             b"!" * 90 + b"\n\n", b"the " * 40, b"<s>[INST] hi [/INST]</s>"]


def mixed_docs(n_ascii=60, n_mixed=25, n_zipf=60, max_len=5000):
    import corpus
    docs = []
    d, o = corpus.generate("ascii", n_ascii, 512, seed=corpus.BASE_SEED + 1)
    docs += corpus.docs_of(d, o)
    d, o = corpus.generate("mixed", n_mixed, 2048, seed=corpus.BASE_SEED + 2)
    docs += corpus.docs_of(d, o)
    d, o = corpus.generate("zipf", n_zipf, seed=corpus.BASE_SEED + 4)
    docs += [x for x in corpus.docs_of(d, o) if len(x) <= max_len]
    return docs + list(EDGE_DOCS)


def random_unicode_docs(n, seed=7, max_len=120):
    alpha = ["a", "S", "s", "t", "r", "e", "l", "v", "m", "d", "x", "1", "2", "'", "!", " ", " ", "\n", "\r", "\t",
             "ſ", "é", "中", "٣", " ", " ", "　", "\U0001f680", "́", "\x00", "-"]
    rng = random.Random(seed)
    return ["".join(rng.choice(alpha) for _ in range(rng.randint(0, max_len))).encode("utf-8") for _ in range(n)]


def oracle_for(v):
    import tk_oracle
    return tk_oracle.Oracle(v["tokens"], v["num_special"], v["bos"], v["eos"])


def starts_to_pieces(doc, starts):
    s = list(starts) + [len(doc)]
    return [doc[s[i]:s[i + 1]] for i in range(len(s) - 1)]
