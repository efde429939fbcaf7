"""ctypes front-end of the CPU wave emulator (tests/emu/libtk_emu.so).  Test infrastructure."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        name = "libtk_emu_asan.so" if os.environ.get("TK_TEST_SANITIZE") else "libtk_emu.so"   # tests/test_sanitizers.py
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
        L = ctypes.CDLL(os.path.join(_HERE, name))
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.emu_encode_batch.restype = ctypes.c_int
        L.emu_encode_batch.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                       u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, u32p,
                                       u64p, u8p, u64p, u64p, ctypes.c_int]
        L.emu_last_error.restype = ctypes.c_char_p
        L.emu_table_cache_roundtrip.restype = ctypes.c_int
        L.emu_table_cache_roundtrip.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        L.emu_table_info.restype = ctypes.c_int
        L.emu_table_info.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, u64p]
        L.emu_flat_encode_batch.restype = ctypes.c_int
        L.emu_flat_encode_batch.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                            u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, u32p, u64p, u8p, u8p,
                                            u64p, u64p, ctypes.c_int]
        L.emu_memo_set.restype = None
        L.emu_memo_set.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.emu_memo_info.restype = None
        L.emu_memo_info.argtypes = [u64p]
        L.emu_long_merge.restype = ctypes.c_int64
        L.emu_long_merge.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, u8p, ctypes.c_uint32, u32p, u64p]
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def pack_docs(docs):
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    data = np.frombuffer(b"".join(docs) or b"\0", dtype=np.uint8).copy()
    return data, offs


def encode_batch(token_bytes, num_special, bos, eos, docs, add_bos=True, add_eos=True, split_only=False, pattern=0):
    """Runs the device algorithm on the emulator.  Returns (list of id lists, starts flags, n_deferred)."""
    toffs = np.zeros(len(token_bytes) + 1, np.uint32)
    toffs[1:] = np.cumsum([len(t) for t in token_bytes], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(token_bytes), dtype=np.uint8).copy()
    data, offs = pack_docs(docs)
    n = int(offs[-1])
    D = len(docs)
    out = np.zeros(n + 2 * D + 1, np.uint32)
    oo = np.zeros(D + 1, np.uint64)
    dbg = np.zeros(max(n, 1), np.uint8)
    ndef = ctypes.c_uint64(0)
    nops = ctypes.c_uint64(0)
    rc = lib().emu_encode_batch(_p(blob, ctypes.c_uint8), _p(toffs, ctypes.c_uint32), len(token_bytes), num_special,
                                bos, eos, _p(data, ctypes.c_uint8), _p(offs, ctypes.c_uint64), D, int(add_bos),
                                int(add_eos), int(split_only), _p(out, ctypes.c_uint32), _p(oo, ctypes.c_uint64),
                                _p(dbg, ctypes.c_uint8), ctypes.byref(ndef), ctypes.byref(nops), int(pattern))
    if rc != 0:
        raise RuntimeError("emu_encode_batch rc=%d: %s" % (rc, lib().emu_last_error().decode()))
    ids = [out[int(oo[d]):int(oo[d + 1])].tolist() for d in range(D)]
    starts = []
    for d in range(D):
        a, b = int(offs[d]), int(offs[d + 1])
        starts.append([i for i in range(b - a) if dbg[a + i]])
    return ids, starts, int(ndef.value)


def flat_encode_batch(token_bytes, num_special, bos, eos, docs, add_bos=True, add_eos=True, pattern=0):
    """The flat path (tk_flat_impl.h) on the emulator.  Returns (id lists, starts per document, flagged list)."""
    toffs = np.zeros(len(token_bytes) + 1, np.uint32)
    toffs[1:] = np.cumsum([len(t) for t in token_bytes], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(token_bytes), dtype=np.uint8).copy()
    data, offs = pack_docs(docs)
    n = int(offs[-1])
    D = len(docs)
    out = np.zeros(n + 2 * D + 1, np.uint32)
    oo = np.zeros(D + 1, np.uint64)
    dbg = np.zeros(max(n, 1), np.uint8)
    fl = np.zeros(max(D, 1), np.uint8)
    nfl = ctypes.c_uint64(0)
    nops = ctypes.c_uint64(0)
    rc = lib().emu_flat_encode_batch(_p(blob, ctypes.c_uint8), _p(toffs, ctypes.c_uint32), len(token_bytes), num_special,
                                     bos, eos, _p(data, ctypes.c_uint8), _p(offs, ctypes.c_uint64), D, int(add_bos),
                                     int(add_eos), _p(out, ctypes.c_uint32), _p(oo, ctypes.c_uint64),
                                     _p(dbg, ctypes.c_uint8), _p(fl, ctypes.c_uint8), ctypes.byref(nfl), ctypes.byref(nops), int(pattern))
    if rc != 0:
        raise RuntimeError("emu_flat_encode_batch rc=%d: %s" % (rc, lib().emu_last_error().decode()))
    ids = [out[int(oo[d]):int(oo[d + 1])].tolist() for d in range(D)]
    starts = []
    for d in range(D):
        a, b = int(offs[d]), int(offs[d + 1])
        starts.append([i for i in range(b - a) if dbg[a + i]])
    return ids, starts, [d for d in range(D) if fl[d]]


_MEMO_BUF = None


def memo_set(log2):
    """A memo table (csrc/tk_hash.h MEMO) for the following flat_encode_batch calls, kept across them like a context keeps its
    own across tk_encode_batch calls; log2 = 0 switches it off.  The table is empty afterwards."""
    global _MEMO_BUF
    if not log2:
        lib().emu_memo_set(None, 0)
        _MEMO_BUF = None
        return
    raw = np.zeros((8 << log2) + 8, np.uint32)
    skip = (-raw.ctypes.data % 32) // 4
    _MEMO_BUF = raw                              # (kept alive: the emulator holds a pointer into it)
    lib().emu_memo_set(raw.ctypes.data + 4 * skip, log2)


def memo_info():
    out = np.zeros(5, np.uint64)
    lib().emu_memo_info(_p(out, ctypes.c_uint64))
    return dict(calls=int(out[0]), hits_last=int(out[1]), entries=int(out[2]), logged_last=int(out[3]), won_last=int(out[4]))


def table_info(token_bytes, num_special):
    """Facts about the device tables the host builder makes for this vocabulary."""
    toffs = np.zeros(len(token_bytes) + 1, np.uint32)
    toffs[1:] = np.cumsum([len(t) for t in token_bytes], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(token_bytes), dtype=np.uint8).copy()
    out = np.zeros(9, np.uint64)
    rc = lib().emu_table_info(_p(blob, ctypes.c_uint8), _p(toffs, ctypes.c_uint32), len(token_bytes), num_special, _p(out, ctypes.c_uint64))
    if rc != 0:
        raise RuntimeError("emu_table_info rc=%d: %s" % (rc, lib().emu_last_error().decode()))
    return dict(key_hash_mode=int(out[0]), key8_slots=int(out[1]), key16_slots=int(out[2]), keys_in_second_slot=int(out[3]),
                flagged_slots=int(out[4]), pair_buckets=int(out[5]), pairs=int(out[6]), pair_filter_set_bits=int(out[7]),
                pair_filter_bits=int(out[8]))


def table_cache_roundtrip(token_bytes, num_special, path):
    """build -> save -> load (a wrong key must be refused) -> compare every field.  Returns timings / file size."""
    toffs = np.zeros(len(token_bytes) + 1, np.uint32)
    toffs[1:] = np.cumsum([len(t) for t in token_bytes], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(token_bytes), dtype=np.uint8).copy()
    out = np.zeros(4, np.float64)
    rc = lib().emu_table_cache_roundtrip(_p(blob, ctypes.c_uint8), _p(toffs, ctypes.c_uint32), len(token_bytes), num_special,
                                         os.fsencode(path), _p(out, ctypes.c_double))
    if rc != 0:
        raise RuntimeError("emu_table_cache_roundtrip rc=%d: %s" % (rc, lib().emu_last_error().decode()))
    return dict(build_s=float(out[0]), save_s=float(out[1]), load_s=float(out[2]), file_bytes=int(out[3]))


def long_merge(token_bytes, num_special, piece, kind):
    """The round-based workgroup merge of ONE long piece (csrc/tk_long_impl.h) on 16 emulated waves: kind 0 = compacting
    rounds, 1 = lazy rounds.  Returns the ids (the pure merge: no whole-piece look-up)."""
    toffs = np.zeros(len(token_bytes) + 1, np.uint32)
    toffs[1:] = np.cumsum([len(t) for t in token_bytes], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(token_bytes), dtype=np.uint8).copy()
    data = np.frombuffer(piece, dtype=np.uint8).copy()
    out = np.zeros(len(piece) + 1, np.uint32)
    nb = ctypes.c_uint64(0)
    rc = lib().emu_long_merge(_p(blob, ctypes.c_uint8), _p(toffs, ctypes.c_uint32), len(token_bytes), num_special, int(kind),
                              _p(data, ctypes.c_uint8), len(piece), _p(out, ctypes.c_uint32), ctypes.byref(nb))
    if rc < 0:
        raise RuntimeError("emu_long_merge rc=%d: %s" % (rc, lib().emu_last_error().decode()))
    return out[:rc].tolist()
