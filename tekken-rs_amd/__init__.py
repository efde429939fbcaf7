"""tekken-rs_amd -- MI355X-native batch tokenization behind tekken-rs's `Tekkenizer::encode`.

Python is plumbing here: this module only loads the in-tree C-ABI library
(`libtekken_hip.so`, built from csrc/ by `make -C tekken-rs_amd`) with ctypes and mirrors the
reference's public surface for this path --

    Tekkenizer.from_file / encode / decode / SpecialTokenPolicy     (reference src/tekkenizer.rs:222,378,436;
                                                                     src/special_tokens.rs:128-136)

plus the batch and device-resident entry points that the GPU path adds.  There is no CPU
fallback: if the library or a HIP device is missing, calls raise.

The directory name contains a hyphen, so import it with
`importlib.import_module("tekken-rs_amd")`.
"""
import ctypes
import enum
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TK_HIP_LIB: another build of the same library -- the development build with the timing ablations, `make ablate`, or an A / B
# variant --, so that no tool ever has to copy a variant over the shipped file)
LIB_PATH = os.environ.get("TK_HIP_LIB") or os.path.join(_HERE, "libtekken_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

TK_OK = 0
CHECK_OFFSETS, CHECK_UTF8 = 1, 2   # tk_encode_batch_device_ex
TK_ERR_INVALID_CONFIG = -1
TK_ERR_RUNTIME = -2
TK_ERR_INVALID_UTF8 = -3
TK_ERR_NO_DEVICE = -4
TK_ERR_INVALID_ARG = -5
TK_ERR_IO = -6
TK_ERR_JSON = -7
TK_ERR_BASE64 = -8
TK_ERR_TOKEN_NOT_FOUND = -9
TK_ERR_SPECIAL_POLICY = -10


class TokenizerError(Exception):
    """Mirror of tekken::errors::TokenizerError (reference src/errors.rs:23-59)."""
    KIND = {TK_ERR_INVALID_CONFIG: "InvalidConfig", TK_ERR_RUNTIME: "Tokenizers", TK_ERR_INVALID_UTF8: "Tokenizers",
            TK_ERR_NO_DEVICE: "Tokenizers", TK_ERR_INVALID_ARG: "InvalidConfig", TK_ERR_IO: "Io", TK_ERR_JSON: "Json",
            TK_ERR_BASE64: "Base64", TK_ERR_TOKEN_NOT_FOUND: "TokenNotFound",
            TK_ERR_SPECIAL_POLICY: "SpecialTokenPolicy"}

    def __init__(self, code, message):
        self.code = code
        self.kind = self.KIND.get(code, "Tokenizers")
        super().__init__("%s: %s" % (self.kind, message))


class SpecialTokenPolicy(enum.IntEnum):
    """reference src/special_tokens.rs:128-136"""
    Ignore = 0
    Keep = 1
    Raise = 2


class _TextResult(ctypes.Structure):
    _fields_ = [("bytes", ctypes.POINTER(ctypes.c_uint8)), ("offsets", ctypes.POINTER(ctypes.c_uint64)),
                ("n_bytes", ctypes.c_uint64), ("n_docs", ctypes.c_uint64)]


class _Result(ctypes.Structure):
    _fields_ = [("ids", ctypes.POINTER(ctypes.c_uint32)), ("offsets", ctypes.POINTER(ctypes.c_uint64)),
                ("n_ids", ctypes.c_uint64), ("n_docs", ctypes.c_uint64)]


_LIB = None


def build(force=False):
    """Compile libtekken_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", _HERE, "libtekken_hip.so"])
    return LIB_PATH


def lib():
    """The C-ABI library.  Raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libtekken_hip.so is missing: run `make -C tekken-rs_amd` (or __graft_entry__.build())")
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.
    # If torch is going to live in this process it must be loaded FIRST so that our DT_NEEDED
    # libamdhip64.so.7 resolves to the copy torch already mapped (two runtimes cannot both open the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)
    L.tk_ctx_create.restype = ctypes.c_int
    L.tk_ctx_create.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                ctypes.c_int, ctypes.POINTER(vp)]
    L.tk_ctx_destroy.argtypes = [vp]
    L.tk_last_error.restype = ctypes.c_char_p
    L.tk_last_error.argtypes = [vp]
    L.tk_encode_batch.restype = ctypes.c_int
    L.tk_encode_batch.argtypes = [vp, u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(_Result)]
    L.tk_free_result.argtypes = [ctypes.POINTER(_Result)]
    L.tk_encode_one.restype = ctypes.c_int
    L.tk_encode_one.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, u32p, ctypes.c_uint64, u64p]
    L.tk_small_path_calls.restype = ctypes.c_uint64
    L.tk_small_path_calls.argtypes = [vp]
    L.tk_round_path_docs.restype = ctypes.c_uint64
    L.tk_round_path_docs.argtypes = [vp]
    L.tk_long_piece_records.restype = ctypes.c_uint64
    L.tk_long_piece_records.argtypes = [vp]
    if hasattr(L, "tk_ctx_set_memo"):   # (memo of merged pieces: libraries built before it have neither)
        L.tk_ctx_set_memo.restype = ctypes.c_int
        L.tk_ctx_set_memo.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.tk_ctx_memo_clear.restype = ctypes.c_int
        L.tk_ctx_memo_clear.argtypes = [vp]
        L.tk_memo_stats.restype = ctypes.c_int
        L.tk_memo_stats.argtypes = [vp, u64p, u64p, u64p, u64p, ctypes.POINTER(ctypes.c_int)]
    if hasattr(L, "tk_last_host_syncs"):
        L.tk_last_host_syncs.restype = ctypes.c_uint64
        L.tk_last_host_syncs.argtypes = [vp]
    if hasattr(L, "tk_cut_chunks"):   # (diagnostics only; tools/ab_bench.sh also loads libraries built before it existed)
        L.tk_cut_chunks.restype = ctypes.c_uint64
        L.tk_cut_chunks.argtypes = [vp]
    L.tk_encode_batch_device.restype = ctypes.c_int
    L.tk_encode_batch_device.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, vp,
                                         ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    L.tk_encode_batch_device_ex.restype = ctypes.c_int
    L.tk_encode_batch_device_ex.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp,
                                            ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    L.tk_host_alloc.restype = ctypes.c_void_p
    L.tk_host_alloc.argtypes = [ctypes.c_size_t]
    L.tk_host_free.argtypes = [ctypes.c_void_p]
    L.tk_encode_batch_pipelined.restype = ctypes.c_int
    L.tk_encode_batch_pipelined.argtypes = [vp, u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u32p,
                                            ctypes.c_uint64, u64p, u64p]
    L.tk_ids18_bytes.restype = ctypes.c_uint64
    L.tk_ids18_bytes.argtypes = [ctypes.c_uint64]
    L.tk_pack_ids18_device.restype = ctypes.c_int
    L.tk_pack_ids18_device.argtypes = [vp, vp, ctypes.c_uint64, vp, vp]
    L.tk_unpack_ids18_device.restype = ctypes.c_int
    L.tk_unpack_ids18_device.argtypes = [vp, vp, ctypes.c_uint64, vp, vp]
    L.tk_ctx_set_pattern.restype = ctypes.c_int
    L.tk_ctx_set_pattern.argtypes = [vp, ctypes.c_int]
    L.tk_tokenizer_set_honour_pattern.restype = ctypes.c_int
    L.tk_tokenizer_set_honour_pattern.argtypes = [vp, ctypes.c_int]
    L.tk_ctx_set_special_tokens.restype = ctypes.c_int
    L.tk_ctx_set_special_tokens.argtypes = [vp, u8p, u32p, ctypes.c_uint32]
    L.tk_decode_batch.restype = ctypes.c_int
    L.tk_decode_batch.argtypes = [vp, u32p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_TextResult), u64p]
    L.tk_free_text_result.argtypes = [ctypes.POINTER(_TextResult)]
    L.tk_decode_batch_device.restype = ctypes.c_int
    L.tk_decode_batch_device.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, vp, ctypes.POINTER(vp),
                                         ctypes.POINTER(vp), u64p, u64p]
    L.tk_tokenizer_decode_batch.restype = ctypes.c_int
    L.tk_tokenizer_decode_batch.argtypes = [vp, u32p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_TextResult), u64p]
    L.tk_last_timing.restype = ctypes.c_int
    if hasattr(L, "tk_last_merge_ms"):   # (like tk_cut_chunks: a diagnostic that older builds of the library lack)
        L.tk_last_merge_ms.restype = ctypes.c_float
        L.tk_last_merge_ms.argtypes = [vp]
    L.tk_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.tk_last_stats.restype = ctypes.c_int
    L.tk_last_stats.argtypes = [vp, u64p, u64p]
    L.tk_split_batch.restype = ctypes.c_int
    L.tk_split_batch.argtypes = [vp, u8p, u64p, ctypes.c_uint64, u8p]
    L.tk_tokenizer_from_file.restype = ctypes.c_int
    L.tk_tokenizer_from_file.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp)]
    L.tk_tokenizer_from_json.restype = ctypes.c_int
    L.tk_tokenizer_from_json.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
    L.tk_tokenizer_destroy.argtypes = [vp]
    L.tk_tokenizer_last_error.restype = ctypes.c_char_p
    L.tk_tokenizer_last_error.argtypes = [vp]
    L.tk_tokenizer_encode.restype = ctypes.c_int
    L.tk_tokenizer_encode.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(u32p), ctypes.POINTER(ctypes.c_size_t)]
    L.tk_tokenizer_encode_batch.restype = ctypes.c_int
    L.tk_tokenizer_encode_batch.argtypes = [vp, u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(_Result)]
    L.tk_free_ids.argtypes = [u32p]
    L.tk_tokenizer_decode.restype = ctypes.c_int
    L.tk_tokenizer_decode.argtypes = [vp, u32p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_size_t)]
    L.tk_free_text.argtypes = [ctypes.c_void_p]
    u64pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_uint64))
    L.tk_free_offsets.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.tk_free_offsets.restype = None
    L.tk_tokenizer_decode_all.restype = ctypes.c_int
    L.tk_tokenizer_decode_all.argtypes = [vp, u32p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), u64pp,
                                          ctypes.POINTER(ctypes.c_size_t)]
    L.tk_tokenizer_vocab.restype = ctypes.c_int
    L.tk_tokenizer_vocab.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p), u64pp, ctypes.POINTER(ctypes.c_size_t)]
    L.tk_tokenizer_vocab_size.restype = ctypes.c_uint32
    L.tk_tokenizer_vocab_size.argtypes = [vp]
    L.tk_tokenizer_num_special_tokens.restype = ctypes.c_uint32
    L.tk_tokenizer_num_special_tokens.argtypes = [vp]
    L.tk_tokenizer_version.restype = ctypes.c_char_p
    L.tk_tokenizer_version.argtypes = [vp]
    L.tk_tokenizer_control_token.restype = ctypes.c_int
    L.tk_tokenizer_control_token.argtypes = [vp, ctypes.c_char_p, u32p]
    L.tk_tokenizer_is_special.restype = ctypes.c_int
    L.tk_tokenizer_is_special.argtypes = [vp, ctypes.c_uint32]
    L.tk_tokenizer_is_byte.restype = ctypes.c_int
    L.tk_tokenizer_is_byte.argtypes = [vp, ctypes.c_uint32]
    L.tk_tokenizer_id_to_piece.restype = ctypes.c_int
    L.tk_tokenizer_id_to_piece.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.POINTER(ctypes.c_size_t)]
    L.tk_tokenizer_id_to_byte_piece.restype = ctypes.c_int
    L.tk_tokenizer_id_to_byte_piece.argtypes = [vp, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                                ctypes.POINTER(ctypes.c_size_t)]
    L.tk_tokenizer_ctx.restype = vp
    L.tk_tokenizer_ctx.argtypes = [vp]
    L.tk_node_create.restype = ctypes.c_int
    L.tk_node_create.argtypes = [u8p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)]
    L.tk_node_destroy.argtypes = [vp]
    L.tk_node_last_error.restype = ctypes.c_char_p
    L.tk_node_last_error.argtypes = [vp]
    L.tk_node_encode_batch.restype = ctypes.c_int
    L.tk_node_encode_batch.argtypes = [vp, u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_Result)]
    L.tk_node_encode_batch_pinned.restype = ctypes.c_int
    L.tk_node_encode_batch_pinned.argtypes = [vp, u8p, u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, u32p, ctypes.c_uint64, u64p, u64p]
    L.tk_node_n_devices.restype = ctypes.c_int
    L.tk_node_n_devices.argtypes = [vp]
    L.tk_node_last_shards.restype = ctypes.c_int
    L.tk_node_last_shards.argtypes = [vp, u64p, u64p, ctypes.c_int]
    L.tk_node_last_timing.restype = ctypes.c_int
    L.tk_node_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.tk_tokenizer_json_pattern.restype = ctypes.c_char_p
    L.tk_tokenizer_json_pattern.argtypes = [vp]
    L.tk_tokenizer_from_cache.restype = ctypes.c_int
    L.tk_tokenizer_from_cache.argtypes = [vp]
    L.tk_tokenizer_rank_table.restype = ctypes.c_int
    L.tk_tokenizer_rank_table.argtypes = [vp, ctypes.POINTER(u8p), ctypes.POINTER(u32p), u32p]
    _LIB = L
    return L


def _p(arr, ct):
    return arr.ctypes.data_as(ctypes.POINTER(ct))


def pack_docs(docs):
    """list[bytes] -> (uint8[n_bytes], uint64[D+1]) packed byte buffer + doc-offset array."""
    offs = np.zeros(len(docs) + 1, np.uint64)
    if docs:
        offs[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    joined = b"".join(docs)
    data = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(0, np.uint8)
    return data, offs


def _take_result(res):
    n, D = int(res.n_ids), int(res.n_docs)
    ids = np.ctypeslib.as_array(res.ids, shape=(max(n, 1),))[:n].copy()
    offs = np.ctypeslib.as_array(res.offsets, shape=(D + 1,)).copy()
    lib().tk_free_result(ctypes.byref(res))
    return ids, offs


class DeviceView:
    """Zero-copy view of a context-owned device buffer for array libraries that understand
    `__cuda_array_interface__` (e.g. `torch.as_tensor(view, device="cuda")`).  Valid until the next
    call on the owning context."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


class Engine:
    """Engine-level context: the replacement for CoreBPE (reference src/tekkenizer.rs:125, 384-386)."""

    def __init__(self, token_bytes, num_special, bos_id, eos_id, device=0, _borrowed=None):
        self._own = _borrowed is None
        if _borrowed is not None:
            self._h = _borrowed
            return
        toks = list(token_bytes)
        offs = np.zeros(len(toks) + 1, np.uint32)
        offs[1:] = np.cumsum([len(t) for t in toks], dtype=np.uint64).astype(np.uint32)
        blob = np.frombuffer(b"".join(toks) or b"\0", dtype=np.uint8).copy()
        h = ctypes.c_void_p()
        rc = lib().tk_ctx_create(_p(blob, ctypes.c_uint8), _p(offs, ctypes.c_uint32), len(toks), num_special, bos_id,
                                 eos_id, device, ctypes.byref(h))
        if rc != TK_OK:
            raise TokenizerError(rc, lib().tk_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and self._own:
            lib().tk_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        return TokenizerError(rc, lib().tk_last_error(self._h).decode())

    def encode_batch(self, data, offs, add_bos=True, add_eos=True, validate_utf8=False):
        """Host buffers in, host buffers out: (ids uint32[T], out_offsets uint64[D+1])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        res = _Result()
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        rc = lib().tk_encode_batch(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), len(offs) - 1,
                                   int(add_bos), int(add_eos), int(validate_utf8), ctypes.byref(res))
        if rc != TK_OK:
            raise self._err(rc)
        return _take_result(res)

    def encode_batch_pipelined(self, data, offs, add_bos=True, add_eos=True, slice_bytes=0, ids_out=None, offsets_out=None):
        """tk_encode_batch_pipelined: the batch streams through the GPU in slices (copy up / kernels / copy down overlapped).
        data / offs / ids_out / offsets_out may be pinned arrays from `host_empty` (then every copy is an asynchronous
        DMA); ids_out (uint32) must hold offs[-1] + 2 * n_docs ids at most.  Returns (ids view, offsets)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n_docs = len(offs) - 1
        if ids_out is None:
            ids_out = np.empty(int(offs[-1]) + 2 * n_docs + 1, np.uint32)
        if offsets_out is None:
            offsets_out = np.empty(n_docs + 1, np.uint64)
        n = ctypes.c_uint64(0)
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        rc = lib().tk_encode_batch_pipelined(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), n_docs, int(add_bos),
                                             int(add_eos), int(slice_bytes), _p(ids_out, ctypes.c_uint32), len(ids_out),
                                             _p(offsets_out, ctypes.c_uint64), ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        return ids_out[:int(n.value)], offsets_out

    def encode_one(self, text, add_bos=False, add_eos=False, out=None):
        """tk_encode_one: one document, caller-owned output (numpy uint32 array of >= len + 2 entries; made if None)."""
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        if out is None:
            out = np.empty(len(raw) + 2, np.uint32)
        n = ctypes.c_uint64(0)
        rc = lib().tk_encode_one(self._h, raw, len(raw), int(add_bos), int(add_eos), _p(out, ctypes.c_uint32), len(out), ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        return out[:n.value]

    def round_path_docs(self):
        """Documents so far whose long piece was merged in rounds by a workgroup (csrc/tk_long.hip)."""
        return int(lib().tk_round_path_docs(self._h))

    def long_piece_records(self):
        """Pieces of 65..256 bytes of the last batch that stayed on the flat path as records."""
        return int(lib().tk_long_piece_records(self._h))

    def set_memo(self, log2_entries, policy=0):
        """Memo of merged pieces (tk_ctx_set_memo): 0 = off, 10..26 = 2^n entries; policy 0 adaptive, 1 always on."""
        rc = lib().tk_ctx_set_memo(self._h, int(log2_entries), int(policy))
        if rc != TK_OK:
            raise self._err(rc)

    def memo_clear(self):
        rc = lib().tk_ctx_memo_clear(self._h)
        if rc != TK_OK:
            raise self._err(rc)

    def memo_stats(self):
        """{lookups_last, hits_last, lookups_total, hits_total, active_last} (tk_memo_stats)."""
        if not hasattr(lib(), "tk_memo_stats"):
            return {"lookups_last": 0, "hits_last": 0, "lookups_total": 0, "hits_total": 0, "active_last": False}
        v = [ctypes.c_uint64(0) for _ in range(4)]
        act = ctypes.c_int(0)
        lib().tk_memo_stats(self._h, *[ctypes.byref(x) for x in v], ctypes.byref(act))
        return {"lookups_last": v[0].value, "hits_last": v[1].value, "lookups_total": v[2].value, "hits_total": v[3].value,
                "active_last": bool(act.value)}

    def cut_chunks(self):
        """Regions of the last batch whose long pieces were cut into independently merged fragments."""
        return int(lib().tk_cut_chunks(self._h)) if hasattr(lib(), "tk_cut_chunks") else 0

    def last_host_syncs(self):
        """Host waits of the last batch on the flat pipeline."""
        return int(lib().tk_last_host_syncs(self._h)) if hasattr(lib(), "tk_last_host_syncs") else 0

    def small_path_calls(self):
        """Calls served by the one-launch small-batch path so far."""
        return int(lib().tk_small_path_calls(self._h))

    def encode_docs(self, docs, add_bos=True, add_eos=True, validate_utf8=False):
        data, offs = pack_docs(docs)
        ids, oo = self.encode_batch(data, offs, add_bos, add_eos, validate_utf8)
        return [ids[int(oo[d]):int(oo[d + 1])].tolist() for d in range(len(docs))]

    def encode_batch_device(self, d_bytes_ptr, d_offs_ptr, n_docs, n_bytes, add_bos=True, add_eos=True, stream=0, checks=0):
        """Inputs resident in HBM (raw device pointers).  Returns (d_ids_ptr, d_out_offs_ptr, n_ids);
        the output buffers belong to the context and stay valid until the next call.  checks: CHECK_OFFSETS | CHECK_UTF8
        (tk_encode_batch_device_ex: the offsets / the documents are checked on the device first)."""
        d_ids, d_oo, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64(0)
        if checks:
            rc = lib().tk_encode_batch_device_ex(self._h, ctypes.c_void_p(d_bytes_ptr), ctypes.c_void_p(d_offs_ptr), n_docs,
                                                 n_bytes, int(add_bos), int(add_eos), int(checks), ctypes.c_void_p(stream),
                                                 ctypes.byref(d_ids), ctypes.byref(d_oo), ctypes.byref(n))
        else:
            rc = lib().tk_encode_batch_device(self._h, ctypes.c_void_p(d_bytes_ptr), ctypes.c_void_p(d_offs_ptr), n_docs,
                                              n_bytes, int(add_bos), int(add_eos), ctypes.c_void_p(stream),
                                              ctypes.byref(d_ids), ctypes.byref(d_oo), ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        return d_ids.value, d_oo.value, int(n.value)

    def encode_batch_device_views(self, d_bytes_ptr, d_offs_ptr, n_docs, n_bytes, add_bos=True, add_eos=True, stream=0):
        """Same, returning (ids view as int32[n_ids], offsets view as int64[n_docs+1])."""
        p_ids, p_oo, n = self.encode_batch_device(d_bytes_ptr, d_offs_ptr, n_docs, n_bytes, add_bos, add_eos, stream)
        return DeviceView(p_ids, n, "<i4"), DeviceView(p_oo, n_docs + 1, "<i8")

    def pack_ids18_device(self, d_ids_ptr, n_ids, d_packed_ptr, stream=0):
        """ids (uint32, device) -> 18-bit wire format (ids18_bytes(n_ids) bytes, device); raises if an id needs more bits."""
        rc = lib().tk_pack_ids18_device(self._h, ctypes.c_void_p(d_ids_ptr), n_ids, ctypes.c_void_p(d_packed_ptr), ctypes.c_void_p(stream))
        if rc != TK_OK:
            raise self._err(rc)

    def unpack_ids18_device(self, d_packed_ptr, n_ids, d_ids_ptr, stream=0):
        """18-bit wire format -> ids (uint32, device); enqueued on `stream`, not waited for."""
        rc = lib().tk_unpack_ids18_device(self._h, ctypes.c_void_p(d_packed_ptr), n_ids, ctypes.c_void_p(d_ids_ptr), ctypes.c_void_p(stream))
        if rc != TK_OK:
            raise self._err(rc)

    def set_pattern(self, mode):
        """0 = the reference's hard-coded pattern (default), 1 = the JSON pattern of Mistral's tekken.json (opt-in, row f-3)."""
        rc = lib().tk_ctx_set_pattern(self._h, int(mode))
        if rc != TK_OK:
            raise self._err(rc)

    def set_special_tokens(self, strings):
        """Special-token strings by position (needed by decode with SpecialTokenPolicy.Keep)."""
        raw = [x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in strings]
        offs = np.zeros(len(raw) + 1, np.uint32)
        offs[1:] = np.cumsum([len(x) for x in raw], dtype=np.uint64).astype(np.uint32)
        blob = np.frombuffer(b"".join(raw) or b"\0", dtype=np.uint8).copy()
        rc = lib().tk_ctx_set_special_tokens(self._h, _p(blob, ctypes.c_uint8), _p(offs, ctypes.c_uint32), len(raw))
        if rc != TK_OK:
            raise self._err(rc)

    def decode_batch(self, ids, offs, policy=SpecialTokenPolicy.Ignore):
        """Batch Tekkenizer::decode on the GPU: (uint32 ids, uint64 offsets[D+1]) -> (uint8 bytes, uint64 offsets[D+1])."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        res = _TextResult()
        bad = ctypes.c_uint64(0)
        ibuf = ids if len(ids) else np.zeros(1, np.uint32)
        rc = lib().tk_decode_batch(self._h, _p(ibuf, ctypes.c_uint32), _p(offs, ctypes.c_uint64), len(offs) - 1, int(policy),
                                   ctypes.byref(res), ctypes.byref(bad))
        if rc != TK_OK:
            e = self._err(rc)
            e.bad_doc = int(bad.value)
            raise e
        n, D = int(res.n_bytes), int(res.n_docs)
        data = np.ctypeslib.as_array(res.bytes, shape=(max(n, 1),))[:n].copy()
        oo = np.ctypeslib.as_array(res.offsets, shape=(D + 1,)).copy()
        lib().tk_free_text_result(ctypes.byref(res))
        return data, oo

    def decode_docs(self, id_lists, policy=SpecialTokenPolicy.Ignore):
        offs = np.zeros(len(id_lists) + 1, np.uint64)
        if id_lists:
            offs[1:] = np.cumsum([len(x) for x in id_lists], dtype=np.uint64)
        ids = np.array([i for x in id_lists for i in x], dtype=np.uint32)
        data, oo = self.decode_batch(ids, offs, policy)
        raw = data.tobytes()
        return [raw[int(oo[d]):int(oo[d + 1])] for d in range(len(id_lists))]

    def decode_batch_device(self, d_ids_ptr, d_offs_ptr, n_docs, n_ids, policy=SpecialTokenPolicy.Ignore, stream=0):
        """ids resident in HBM -> (bytes view uint8[n_bytes], offsets view int64[n_docs+1]) context-owned."""
        d_b, d_o, n, bad = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = lib().tk_decode_batch_device(self._h, ctypes.c_void_p(d_ids_ptr), ctypes.c_void_p(d_offs_ptr), n_docs, n_ids,
                                          int(policy), ctypes.c_void_p(stream), ctypes.byref(d_b), ctypes.byref(d_o),
                                          ctypes.byref(n), ctypes.byref(bad))
        if rc != TK_OK:
            e = self._err(rc)
            e.bad_doc = int(bad.value)
            raise e
        return DeviceView(d_b.value, int(n.value), "|u1"), DeviceView(d_o.value, n_docs + 1, "<i8")

    def last_timing(self):
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        lib().tk_last_timing(self._h, ctypes.byref(a), ctypes.byref(b))
        return {"pipeline_ms": a.value, "encode_kernel_ms": b.value, "merge_ms": float(lib().tk_last_merge_ms(self._h)) if hasattr(lib(), "tk_last_merge_ms") else 0.0}

    def last_stats(self):
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        lib().tk_last_stats(self._h, ctypes.byref(a), ctypes.byref(b))
        return {"long_docs": int(a.value), "handed_back": int(b.value)}

    def split_docs(self, docs):
        """Piece-start offsets per document (vocab-free split, debug / parity entry)."""
        data, offs = pack_docs(docs)
        n = int(offs[-1])
        out = np.zeros(max(n, 1), np.uint8)
        dbuf = data if n else np.zeros(1, np.uint8)
        rc = lib().tk_split_batch(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), len(docs),
                                  _p(out, ctypes.c_uint8))
        if rc != TK_OK:
            raise self._err(rc)
        res = []
        for d in range(len(docs)):
            a, b = int(offs[d]), int(offs[d + 1])
            res.append(np.nonzero(out[a:b])[0].tolist())
        return res


class Node:
    """tk_node_*: every listed GPU behind one call -- documents sharded whole by bytes, one RCCL gather of the id buffers to
    devices[0] (include/tekken_hip.h, csrc/tk_node.cpp).  One process; the path a Rust / C host calls."""

    def __init__(self, token_bytes, num_special, bos_id, eos_id, devices=(0,)):
        toks = list(token_bytes)
        offs = np.zeros(len(toks) + 1, np.uint32)
        offs[1:] = np.cumsum([len(t) for t in toks], dtype=np.uint64).astype(np.uint32)
        blob = np.frombuffer(b"".join(toks) or b"\0", dtype=np.uint8)
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = lib().tk_node_create(_p(blob, ctypes.c_uint8), _p(offs, ctypes.c_uint32), len(toks), num_special, bos_id, eos_id,
                                  devs, len(devices), ctypes.byref(h))
        if rc != TK_OK:
            raise TokenizerError(rc, lib().tk_node_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().tk_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_devices(self):
        return lib().tk_node_n_devices(self._h)

    def encode_batch(self, data, offs, add_bos=True, add_eos=True):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        res = _Result()
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        rc = lib().tk_node_encode_batch(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), len(offs) - 1, int(add_bos),
                                        int(add_eos), ctypes.byref(res))
        if rc != TK_OK:
            raise TokenizerError(rc, lib().tk_node_last_error(self._h).decode())
        return _take_result(res)

    def encode_batch_into(self, data, offs, ids_out, offs_out, add_bos=True, add_eos=True):
        """tk_node_encode_batch_pinned: caller-owned buffers (host_empty: pinned -- nothing allocated or pinned per call).  Returns the
        number of ids written into ids_out; offs_out[: n_docs + 1] holds the id offsets."""
        assert data.dtype == np.uint8 and offs.dtype == np.uint64 and ids_out.dtype == np.uint32 and offs_out.dtype == np.uint64
        # (the C side writes n_docs + 1 offsets and takes no capacity for them; every buffer is handed over as one flat block)
        assert len(offs_out) >= len(offs), "offs_out holds %d offsets, the call writes %d" % (len(offs_out), len(offs))
        for a in (data, offs, ids_out, offs_out):
            assert a.flags["C_CONTIGUOUS"], "tk_node_encode_batch_pinned wants contiguous buffers"
        n = ctypes.c_uint64(0)
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        rc = lib().tk_node_encode_batch_pinned(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), len(offs) - 1, int(add_bos),
                                               int(add_eos), _p(ids_out, ctypes.c_uint32), len(ids_out), _p(offs_out, ctypes.c_uint64), ctypes.byref(n))
        if rc != TK_OK:
            raise TokenizerError(rc, lib().tk_node_last_error(self._h).decode())
        return int(n.value)

    def last_timing(self):
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        lib().tk_node_last_timing(self._h, ctypes.byref(a), ctypes.byref(b))
        return {"kernels_ms_max": a.value, "gather_ms": b.value}

    def last_shards(self):
        """(text bytes, ids) of every device's run in the last batch (tk_node_last_shards)."""
        n = self.n_devices()
        b, i = (ctypes.c_uint64 * n)(), (ctypes.c_uint64 * n)()
        lib().tk_node_last_shards(self._h, b, i, n)
        return list(b), list(i)


class _Pinned:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            lib().tk_host_free(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


def ids18_bytes(n_ids):
    return int(lib().tk_ids18_bytes(int(n_ids)))


def host_empty(n, dtype):
    """numpy array of n elements in PINNED host memory (tk_host_alloc): the buffers tk_encode_batch_pipelined wants."""
    dt = np.dtype(dtype)
    nbytes = max(int(n) * dt.itemsize, 1)
    ptr = lib().tk_host_alloc(nbytes)
    if not ptr:
        raise MemoryError("tk_host_alloc(%d) failed" % nbytes)
    owner = _Pinned(ptr)
    raw = (ctypes.c_uint8 * nbytes).from_address(ptr)
    raw._tk_owner = owner                  # the array keeps `raw` alive (its base), `raw` keeps the allocation alive
    return np.frombuffer(raw, dtype=dt, count=int(n))


class Tekkenizer:
    """Mirror of tekken::tekkenizer::Tekkenizer for the text path (reference src/tekkenizer.rs)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_file(cls, path, device=0):
        """Tekkenizer::from_file (src/tekkenizer.rs:222-248).  device=-1: host-only (no encode)."""
        h = ctypes.c_void_p()
        rc = lib().tk_tokenizer_from_file(os.fsencode(path), device, ctypes.byref(h))
        if rc != TK_OK:
            raise TokenizerError(rc, lib().tk_tokenizer_last_error(None).decode())
        return cls(h)

    @classmethod
    def from_json(cls, text, device=0):
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        h = ctypes.c_void_p()
        rc = lib().tk_tokenizer_from_json(raw, len(raw), device, ctypes.byref(h))
        if rc != TK_OK:
            raise TokenizerError(rc, lib().tk_tokenizer_last_error(None).decode())
        return cls(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().tk_tokenizer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        return TokenizerError(rc, lib().tk_tokenizer_last_error(self._h).decode())

    def set_honour_pattern(self, honour=True):
        """Opt-in (row f-3): use the `pattern` of the loaded tekken.json instead of ignoring it like the reference does."""
        rc = lib().tk_tokenizer_set_honour_pattern(self._h, int(bool(honour)))
        if rc != TK_OK:
            raise self._err(rc)

    def encode(self, text, add_bos=False, add_eos=False):
        """Tekkenizer::encode (src/tekkenizer.rs:378-405)."""
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        ids = ctypes.POINTER(ctypes.c_uint32)()
        n = ctypes.c_size_t(0)
        rc = lib().tk_tokenizer_encode(self._h, raw, len(raw), int(add_bos), int(add_eos), ctypes.byref(ids),
                                       ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        out = [ids[i] for i in range(n.value)]
        lib().tk_free_ids(ids)
        return out

    def encode_batch(self, docs, add_bos=False, add_eos=False):
        data, offs = pack_docs([d.encode("utf-8") if isinstance(d, str) else bytes(d) for d in docs])
        res = _Result()
        dbuf = data if len(data) else np.zeros(1, np.uint8)
        rc = lib().tk_tokenizer_encode_batch(self._h, _p(dbuf, ctypes.c_uint8), _p(offs, ctypes.c_uint64), len(docs),
                                             int(add_bos), int(add_eos), ctypes.byref(res))
        if rc != TK_OK:
            raise self._err(rc)
        ids, oo = _take_result(res)
        return [ids[int(oo[d]):int(oo[d + 1])].tolist() for d in range(len(docs))]

    def decode(self, ids, policy=SpecialTokenPolicy.Ignore):
        """Tekkenizer::decode (src/tekkenizer.rs:436-443)."""
        arr = np.ascontiguousarray(ids, dtype=np.uint32)
        buf = arr if len(arr) else np.zeros(1, np.uint32)
        text = ctypes.c_void_p()
        n = ctypes.c_size_t(0)
        rc = lib().tk_tokenizer_decode(self._h, _p(buf, ctypes.c_uint32), len(arr), int(policy), ctypes.byref(text),
                                       ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        out = ctypes.string_at(text, n.value).decode("utf-8")
        lib().tk_free_text(text)
        return out

    def _strs(self, fn, *args):
        text = ctypes.c_void_p()
        ends = ctypes.POINTER(ctypes.c_uint64)()
        n = ctypes.c_size_t(0)
        rc = fn(self._h, *args, ctypes.byref(text), ctypes.byref(ends), ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        e = [int(ends[i]) for i in range(n.value)]
        raw = ctypes.string_at(text, e[-1] if e else 0)
        lib().tk_free_text(text)
        lib().tk_free_offsets(ends)
        return [raw[a:b].decode("utf-8") for a, b in zip([0] + e[:-1], e)]

    def decode_all(self, ids, policy=SpecialTokenPolicy.Ignore):
        """Tekkenizer::decode_all (src/tekkenizer.rs:463-560): one string per run of special / non-special ids."""
        arr = np.ascontiguousarray(ids, dtype=np.uint32)
        buf = arr if len(arr) else np.zeros(1, np.uint32)
        return self._strs(lib().tk_tokenizer_decode_all, _p(buf, ctypes.c_uint32), len(arr), int(policy))

    def vocab(self):
        """Tekkenizer::vocab (src/tekkenizer.rs:348-350): the piece string of every id."""
        return self._strs(lib().tk_tokenizer_vocab)

    def decode_batch(self, id_lists, policy=SpecialTokenPolicy.Ignore):
        """Batch decode on the GPU: list of id lists -> list of str (an addition; the reference decodes one at a time)."""
        eng = self.engine()
        if eng is None:
            raise TokenizerError(TK_ERR_NO_DEVICE, "tokenizer was created without a device (host-only object)")
        return [b.decode("utf-8") for b in eng.decode_docs(id_lists, policy)]

    def _piece(self, fn, *args):
        text = ctypes.c_void_p()
        n = ctypes.c_size_t(0)
        rc = fn(self._h, *args, ctypes.byref(text), ctypes.byref(n))
        if rc != TK_OK:
            raise self._err(rc)
        out = ctypes.string_at(text, n.value)
        lib().tk_free_text(text)
        return out

    def id_to_piece(self, token_id):
        return self._piece(lib().tk_tokenizer_id_to_piece, token_id).decode("utf-8")

    def id_to_byte_piece(self, token_id, policy=SpecialTokenPolicy.Raise):
        return self._piece(lib().tk_tokenizer_id_to_byte_piece, token_id, int(policy))

    def vocab_size(self):
        return lib().tk_tokenizer_vocab_size(self._h)

    def num_special_tokens(self):
        return lib().tk_tokenizer_num_special_tokens(self._h)

    def version(self):
        return lib().tk_tokenizer_version(self._h).decode()

    def get_control_token(self, name):
        v = ctypes.c_uint32(0)
        rc = lib().tk_tokenizer_control_token(self._h, name.encode("utf-8"), ctypes.byref(v))
        if rc != TK_OK:
            raise self._err(rc)
        return v.value

    def bos_id(self):
        return self.get_control_token("<s>")

    def eos_id(self):
        return self.get_control_token("</s>")

    def pad_id(self):
        return self.get_control_token("<pad>")

    def unk_id(self):
        return self.get_control_token("<unk>")

    def is_special_token(self, token_id):
        return bool(lib().tk_tokenizer_is_special(self._h, token_id))

    def is_byte(self, token_id):
        return bool(lib().tk_tokenizer_is_byte(self._h, token_id))

    def json_pattern(self):
        """config.pattern of the loaded tekken.json (parsed and ignored by the reference, src/tekkenizer.rs:74)."""
        return lib().tk_tokenizer_json_pattern(self._h).decode("utf-8")

    def from_cache(self):
        """True when the object was loaded from a TK_TABLE_CACHE_DIR side file instead of the JSON (row f-2)."""
        return bool(lib().tk_tokenizer_from_cache(self._h))

    def engine(self):
        """The engine context behind this tokenizer (None for host-only objects)."""
        h = lib().tk_tokenizer_ctx(self._h)
        return Engine(None, 0, 0, 0, _borrowed=ctypes.c_void_p(h)) if h else None

    def rank_table(self):
        """list[bytes]: token bytes by rank (what reload_mergeable_ranks produced)."""
        blob = ctypes.POINTER(ctypes.c_uint8)()
        offs = ctypes.POINTER(ctypes.c_uint32)()
        n = ctypes.c_uint32(0)
        lib().tk_tokenizer_rank_table(self._h, ctypes.byref(blob), ctypes.byref(offs), ctypes.byref(n))
        o = np.ctypeslib.as_array(offs, shape=(n.value + 1,))
        total = int(o[-1])
        b = ctypes.string_at(blob, total)
        return [b[int(o[i]):int(o[i + 1])] for i in range(n.value)]
