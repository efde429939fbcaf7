"""csrc/tk_utf8_swar.h -- what tk_decode_validate_kernel runs per lane: four bytes per 32-bit register, three 16-entry tables
(Keiser & Lemire) looked up with byte permutes, run starts as hard boundaries -- on the CPU against python's own decoder:
a document is valid iff every run (the stretch between two run starts) decodes on its own (reference src/tekkenizer.rs:552-555)."""
import ctypes
import os
import random
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    so = os.path.join(HERE, "libtk_utf8_swar_check.so")
    src = os.path.join(HERE, "utf8_swar_check.c")
    hdr = os.path.join(HERE, "..", "tekken-rs_amd", "csrc", "tk_utf8_swar.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-o", so, src])
    L = ctypes.CDLL(so)
    L.tku8_check_doc.restype = ctypes.c_int
    L.tku8_check_doc.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    return L


def _valid(doc, starts):
    cuts = [0] + sorted(s for s in starts if 0 < s < len(doc)) + [len(doc)]
    for a, b in zip(cuts, cuts[1:]):
        try:
            doc[a:b].decode("utf-8")
        except UnicodeDecodeError:
            return False
    return True


def _check(L, pre, doc, starts):
    """doc at offset len(pre) of a buffer (what is in front belongs to another document), run starts relative to the document."""
    s0, s1 = len(pre), len(pre) + len(doc)
    buf = np.frombuffer(pre + doc + b"\xff" * 12, dtype=np.uint8).copy()      # (bytes behind the document must not matter)
    bits = np.zeros((s1 >> 5) + 4, np.uint32)
    for s in starts:
        p = s0 + s
        bits[p >> 5] |= np.uint32(1 << (p & 31))
    got = L.tku8_check_doc(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), s0, s1, bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return got == 0


def test_swar_validation_matches_the_decoder():
    L = _lib()
    rng = random.Random(12)
    # the bytes where the rules change: ASCII, the continuation range and its sub-ranges, every kind of lead, the invalid bytes
    alpha = [0x00, 0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1, 0xEC, 0xED, 0xEE, 0xEF,
             0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xF8, 0xFF]
    good = ["a", "é", "中", "\U0001f680", "߿", "ࠀ", "퟿", "", "\U00010000", "\U0010ffff"]
    n = bad = 0
    for it in range(60000):
        k = rng.random()
        if k < 0.5:
            doc = bytes(rng.choice(alpha) for _ in range(rng.randint(1, 12)))
        elif k < 0.8:
            doc = "".join(rng.choice(good) for _ in range(rng.randint(1, 8))).encode()
            if rng.random() < 0.5 and doc:      # damage one byte
                i = rng.randrange(len(doc))
                doc = doc[:i] + bytes([rng.choice(alpha)]) + doc[i + 1:]
        else:
            doc = "".join(rng.choice(good) for _ in range(rng.randint(20, 90))).encode()   # several dwords, valid unless cut
        starts = [rng.randrange(len(doc) + 1) for _ in range(rng.choice([0, 0, 1, 2, 3]))]
        pre = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 7)))
        want = _valid(doc, starts)
        assert _check(L, pre, doc, starts) == want, (doc, starts, pre, want)
        n += 1
        bad += not want
    assert bad > n // 4 and n - bad > n // 6          # both outcomes well covered


def test_swar_validation_exhaustive_short():
    """every string of up to three bytes over the critical bytes, with and without a run start at every position"""
    L = _lib()
    alpha = [0x41, 0x80, 0xA0, 0xBF, 0xC1, 0xC2, 0xE0, 0xED, 0xEF, 0xF0, 0xF4, 0xF5]
    import itertools
    for ln in (1, 2, 3):
        for t in itertools.product(alpha, repeat=ln):
            doc = bytes(t)
            for starts in ([],) + tuple([s] for s in range(1, ln)):
                for pre in (b"", b"\xe4\xb8", b"abc\xf0"):
                    assert _check(L, pre, doc, list(starts)) == _valid(doc, list(starts)), (doc, starts, pre)
