"""ctypes binding of the CPU oracle (oracle/tk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  Parity status: see tk_oracle.h
("parity unpinned" at token-id level; split pinned against Python `regex`).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    name = "libtk_oracle_asan.so" if os.environ.get("TK_TEST_SANITIZE") else "libtk_oracle.so"   # tests/test_sanitizers.py
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "tk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.tk_oracle_new.restype = ctypes.c_void_p
        L.tk_oracle_new.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.tk_oracle_free.argtypes = [ctypes.c_void_p]
        L.tk_oracle_split.restype = ctypes.c_size_t
        L.tk_oracle_split.argtypes = [u8p, ctypes.c_size_t, u32p, ctypes.c_size_t]
        L.tk_oracle_encode.restype = ctypes.c_size_t
        L.tk_oracle_encode.argtypes = [ctypes.c_void_p, u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u32p,
                                       ctypes.c_size_t]
        L.tk_oracle_encode_batch.restype = ctypes.c_uint64
        L.tk_oracle_encode_batch.argtypes = [ctypes.c_void_p, u8p, u64p, ctypes.c_uint64, ctypes.c_int,
                                             ctypes.c_int, u32p, u64p, ctypes.c_int]
        L.tk_oracle_split_tekken.restype = ctypes.c_size_t
        L.tk_oracle_split_tekken.argtypes = [u8p, ctypes.c_size_t, u32p, ctypes.c_size_t]
        L.tk_oracle_set_pattern.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.tk_oracle_class2.restype = ctypes.c_int
        L.tk_oracle_class2.argtypes = [ctypes.c_uint32]
        L.tk_oracle_class.restype = ctypes.c_int
        L.tk_oracle_class.argtypes = [ctypes.c_uint32]
        L.tk_oracle_last_batch_seconds.restype = ctypes.c_double
        L.tk_oracle_miss_stats.argtypes = [ctypes.c_void_p, u8p, u64p, ctypes.c_uint64, u64p]
        L.tk_oracle_fnv1a.restype = ctypes.c_uint64
        L.tk_oracle_fnv1a.argtypes = [u32p, ctypes.c_uint64]
        _LIB = L
    return _LIB


def _p(arr, ct):
    return arr.ctypes.data_as(ctypes.POINTER(ct))


def split(text: bytes):
    """Piece start offsets of `text` under the hard-coded pattern (src/tekkenizer.rs:123)."""
    n = len(text)
    buf = np.frombuffer(text, dtype=np.uint8) if n else np.zeros(1, np.uint8)
    starts = np.zeros(max(n, 1), np.uint32)
    k = lib().tk_oracle_split(_p(buf, ctypes.c_uint8), n, _p(starts, ctypes.c_uint32), n)
    return starts[:k].tolist()


def split_tekken(text: bytes):
    """Piece start offsets under the JSON pattern of Mistral's tekken.json (row f-3 groundwork)."""
    n = len(text)
    buf = np.frombuffer(text, dtype=np.uint8) if n else np.zeros(1, np.uint8)
    starts = np.zeros(max(n, 1), np.uint32)
    k = lib().tk_oracle_split_tekken(_p(buf, ctypes.c_uint8), n, _p(starts, ctypes.c_uint32), n)
    return starts[:k].tolist()


def split_pieces(text: bytes):
    s = split(text) + [len(text)]
    return [text[s[i]:s[i + 1]] for i in range(len(s) - 1)]


class Oracle:
    """Rank table + encode, the CPU stand-in for CoreBPE + Tekkenizer::encode."""

    def __init__(self, token_bytes, num_special, bos_id, eos_id):
        self.tokens = list(token_bytes)
        offs = np.zeros(len(self.tokens) + 1, np.uint32)
        offs[1:] = np.cumsum([len(t) for t in self.tokens], dtype=np.uint64).astype(np.uint32)
        blob = np.frombuffer(b"".join(self.tokens) or b"\0", dtype=np.uint8)
        self.num_special, self.bos_id, self.eos_id = num_special, bos_id, eos_id
        self._h = lib().tk_oracle_new(_p(blob, ctypes.c_uint8), _p(offs, ctypes.c_uint32), len(self.tokens),
                                      num_special, bos_id, eos_id)

    def set_pattern(self, mode):
        """0 = the hard-coded pattern (reference behaviour), 1 = the JSON pattern of Mistral's tekken.json (row f-3)."""
        lib().tk_oracle_set_pattern(self._h, int(mode))

    def __del__(self):
        try:
            if self._h:
                lib().tk_oracle_free(self._h)
                self._h = None
        except Exception:
            pass

    def encode(self, text: bytes, add_bos=False, add_eos=False):
        n = len(text)
        buf = np.frombuffer(text, dtype=np.uint8) if n else np.zeros(1, np.uint8)
        out = np.zeros(n + 2, np.uint32)
        k = lib().tk_oracle_encode(self._h, _p(buf, ctypes.c_uint8), n, int(add_bos), int(add_eos),
                                   _p(out, ctypes.c_uint32), n + 2)
        return out[:k].tolist()

    def miss_stats(self, data: np.ndarray, offs: np.ndarray):
        """{pieces, missed (not a vocabulary key), missed_bytes, missed_ids} of a packed batch."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        out = np.zeros(4, np.uint64)
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        lib().tk_oracle_miss_stats(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), len(offs) - 1, _p(out, ctypes.c_uint64))
        return {"pieces": int(out[0]), "missed": int(out[1]), "missed_bytes": int(out[2]), "missed_ids": int(out[3])}

    def miss_records(self, data: np.ndarray, offs: np.ndarray, cap=1 << 24):
        """uint32[n, 3] = {hash of the bytes, bytes, ids produced} of every piece that is no vocabulary key (tools/miss_analysis.py)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        rec = np.zeros((cap, 3), np.uint32)
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        f = lib().tk_oracle_miss_records
        f.restype = ctypes.c_uint64
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        n = f(self._h, dbuf.ctypes.data, offs.ctypes.data, len(offs) - 1, rec.ctypes.data, cap)
        return rec[:min(int(n), cap)]

    def encode_batch(self, data: np.ndarray, offs: np.ndarray, add_bos=True, add_eos=True, threads=1):
        """data: uint8[n_bytes], offs: uint64[D+1] -> (ids uint32[T], out_offs uint64[D+1])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        D = len(offs) - 1
        n = int(offs[-1])
        dbuf = data if n else np.zeros(1, np.uint8)
        out = np.zeros(n + 2 * D + 1, np.uint32)
        oo = np.zeros(D + 1, np.uint64)
        t = lib().tk_oracle_encode_batch(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), D,
                                         int(add_bos), int(add_eos), _p(out, ctypes.c_uint32),
                                         _p(oo, ctypes.c_uint64), threads)
        return out[:t].copy(), oo


def last_batch_seconds() -> float:
    """Wall time spent inside the last Oracle.encode_batch call (the C loop alone, without this binding's buffers)."""
    return float(lib().tk_oracle_last_batch_seconds())


def fnv1a(ids: np.ndarray) -> int:
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    if len(ids) == 0:
        ids_p = np.zeros(1, np.uint32)
        return lib().tk_oracle_fnv1a(_p(ids_p, ctypes.c_uint32), 0)
    return lib().tk_oracle_fnv1a(_p(ids, ctypes.c_uint32), len(ids))


def decode_ref(token_bytes, special_strings, num_special, ids, policy):
    """Pure-Python restatement of Tekkenizer::decode (reference src/tekkenizer.rs:436-560) for tests:
    policy 0 Ignore / 1 Keep / 2 Raise.  Returns bytes, or raises ValueError("special"/"key"/"utf8")."""
    out = []
    g0 = 0
    ids = list(ids)
    while g0 < len(ids):
        sp = ids[g0] < num_special                      # :474
        g1 = g0 + 1
        while g1 < len(ids) and (ids[g1] < num_special) == sp:
            g1 += 1
        if sp:
            if policy == 2:
                raise ValueError("special")            # :531-535
            if policy == 1:
                out.extend(special_strings[i].encode("utf-8") for i in ids[g0:g1])   # :536-540 (by position)
        else:
            run = b""
            for i in ids[g0:g1]:
                r = i - num_special                    # :548-551
                if r >= len(token_bytes):
                    raise ValueError("key")
                run += token_bytes[r]
            try:
                run.decode("utf-8")                    # CoreBPE::decode -> String::from_utf8, :552-555
            except UnicodeDecodeError:
                raise ValueError("utf8")
            out.append(run)
        g0 = g1
    return b"".join(out)
