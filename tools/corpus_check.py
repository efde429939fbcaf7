#!/usr/bin/env python3
"""FNV-1a (64-bit) of a seeded bench corpus -- the value reference_cpu (Rust) prints as `corpus_fnv1a`, to prove that
its port of tools/corpus_gen.c generated the same bytes before any token id is compared.

  python tools/corpus_check.py --kind ascii --docs 1000000 --doc-len 512
"""
import argparse
import json

import numpy as np

import corpus


def fnv1a_bytes(data: np.ndarray) -> int:
    # exact, but vectorised per block: h = (h ^ b) * P mod 2^64 is sequential; Python ints over a 512 MB corpus would
    # take minutes, so run it in C through the corpus library's compiler (gcc is what built tools/libtk_corpus.so)
    import ctypes
    import os
    import subprocess
    import tempfile
    src = r"""
#include <stdint.h>
uint64_t fnv(const uint8_t* p, uint64_t n) { uint64_t h = 1469598103934665603ull; for (uint64_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; } return h; }
"""
    d = tempfile.mkdtemp()
    c, so = os.path.join(d, "f.c"), os.path.join(d, "f.so")
    with open(c, "w") as f:
        f.write(src)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, c])
    L = ctypes.CDLL(so)
    L.fnv.restype = ctypes.c_uint64
    L.fnv.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    data = np.ascontiguousarray(data)
    return int(L.fnv(data.ctypes.data, data.size))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="ascii", choices=list(corpus.KINDS))
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--doc-len", type=int, default=512)
    ap.add_argument("--first-doc", type=int, default=0)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=corpus.BASE_SEED + 1)
    a = ap.parse_args()
    data, offs = corpus.generate(a.kind, a.docs, a.doc_len, seed=a.seed, first_doc=a.first_doc)
    print(json.dumps({"kind": a.kind, "docs": a.docs, "input_bytes": int(offs[-1]), "corpus_fnv1a": "%016x" % fnv1a_bytes(data)}))
