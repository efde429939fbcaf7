"""Seeded synthetic corpora (ctypes front-end of tools/corpus_gen.c).  Bench/test infrastructure.

kinds: "ascii" (G-ascii), "mixed" (G-mixed), "zipf" (Zipf length mix) -- SURVEY.md section 8d.
The seed convention follows the survey: 0x7E44E2 + config index.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
KINDS = {"ascii": 0, "mixed": 1, "zipf": 2}
BASE_SEED = 0x7E44E2


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libtk_corpus.so")
        src = os.path.join(_HERE, "corpus_gen.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            # (several ranks may get here at once: build under a private name, publish with an atomic rename)
            tmp = "%s.tmp%d" % (so, os.getpid())
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", tmp, src, "-lm", "-lpthread"])
            os.replace(tmp, so)
        L = ctypes.CDLL(so)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.tkc_fill_offsets.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u64p]
        L.tkc_gen_docs.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u64p,
                                   u8p, ctypes.c_int]
        L.tkc_n_words.restype = ctypes.c_int
        L.tkc_word.restype = ctypes.c_char_p
        L.tkc_word.argtypes = [ctypes.c_int]
        _LIB = L
    return _LIB


def generate(kind, n_docs, doc_len=512, seed=BASE_SEED, first_doc=0, threads=None):
    """-> (uint8[n_bytes], uint64[n_docs+1]) packed documents."""
    k = KINDS[kind]
    threads = threads or min(16, os.cpu_count() or 1)
    offs = np.zeros(n_docs + 1, np.uint64)
    L = lib()
    L.tkc_fill_offsets(k, seed, first_doc, n_docs, doc_len, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    total = int(offs[-1])
    data = np.zeros(max(total, 1), np.uint8)
    L.tkc_gen_docs(k, seed, first_doc, n_docs, doc_len, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                   data.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), threads)
    return data[:total], offs


def offsets(kind, n_docs, doc_len=512, seed=BASE_SEED, first_doc=0):
    """uint64[n_docs+1] document offsets alone (lengths are a function of the seed and the document index)."""
    offs = np.zeros(n_docs + 1, np.uint64)
    lib().tkc_fill_offsets(KINDS[kind], seed, first_doc, n_docs, doc_len, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    return offs


def words():
    L = lib()
    return [L.tkc_word(i).decode() for i in range(L.tkc_n_words())]


def docs_of(data, offs):
    b = data.tobytes()
    return [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
