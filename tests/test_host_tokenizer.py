"""Host-side mirror of the reference's Tekkenizer (loader, construction checks, decode surface,
SpecialTokenPolicy): tekken-rs_amd/csrc/tekkenizer.cpp through the C ABI, host-only objects
(device = -1), so no GPU is needed.  Expected behaviour is read off the reference source
(file:line in each test); cases mirror reference tests/test_tokenizer_detailed.rs and test_tekken.rs."""
import base64
import json

import pytest


def model(tokens, num_special=10, specials=("<unk>", "<s>", "</s>"), version="v7", vocab_size=None, audio=None):
    m = {"config": {"pattern": "ignored", "num_vocab_tokens": len(tokens),
                    "default_vocab_size": vocab_size if vocab_size is not None else len(tokens) + num_special,
                    "default_num_special_tokens": num_special, "version": version},
         "vocab": [{"rank": r, "token_bytes": base64.b64encode(t).decode(), "token_str": None} for r, t in enumerate(tokens)]}
    if specials is not None:
        m["special_tokens"] = [{"rank": i, "token_str": s, "is_control": True} for i, s in enumerate(specials)]
    if audio is not None:
        m["audio"] = audio
    return m


@pytest.fixture()
def small(tk, small_vocab):
    return tk.Tekkenizer.from_json(json.dumps(model(small_vocab["tokens"])), device=-1)


def test_accessors(tk, small):
    # reference src/tekkenizer.rs:260-350, 574-600
    assert small.vocab_size() == 268 and small.num_special_tokens() == 10 and small.version() == "v7"
    assert small.bos_id() == 1 and small.eos_id() == 2 and small.unk_id() == 0
    assert small.get_control_token("<SPECIAL_7>") == 7          # fillers, src/tekkenizer.rs:108-116
    with pytest.raises(tk.TokenizerError) as e:
        small.get_control_token("[NOPE]")
    assert e.value.kind == "TokenNotFound"                       # :331-341
    with pytest.raises(tk.TokenizerError) as e:
        small.pad_id()                                           # "<pad>" not among these specials
    assert e.value.kind == "TokenNotFound"
    assert small.is_special_token(9) and not small.is_special_token(10)   # boundary at num_special (:574-577)
    assert small.is_byte(10) and small.is_byte(265) and not small.is_byte(266) and not small.is_byte(3)


def test_decode_policies(tk, small):
    # reference src/tekkenizer.rs:436-560 ; tests/test_tekken.rs:54-86
    ids = [1, 266, 42, 129, 121, 124, 118, 110, 2]
    P = tk.SpecialTokenPolicy
    assert small.decode(ids, P.Ignore) == "hello world"
    assert small.decode(ids, P.Keep) == "<s>hello world</s>"
    with pytest.raises(tk.TokenizerError) as e:
        small.decode(ids, P.Raise)
    assert e.value.kind == "SpecialTokenPolicy"
    assert small.decode([], P.Raise) == ""
    assert small.decode([266, 267], P.Raise) == "helloworld"
    # each non-special run is decoded on its own and must be valid UTF-8 (:552-555)
    e_acute = [0xC3 + 10, 0xA9 + 10]
    assert small.decode(e_acute, P.Ignore) == "é"
    with pytest.raises(tk.TokenizerError) as e:
        small.decode([0xC3 + 10, 1, 0xA9 + 10], P.Ignore)       # split by a special token -> two invalid runs
    assert e.value.kind == "Tokenizers"
    with pytest.raises(tk.TokenizerError):
        small.decode([5000], P.Ignore)                            # unknown rank


def test_decode_all_and_vocab(tk, small):
    # reference src/tekkenizer.rs:463-560: one element per run of non-special ids, one per special id under Keep,
    # nothing for a special run under Ignore, an error under Raise
    P = tk.SpecialTokenPolicy
    ids = [1, 1, 266, 42, 267, 2, 266, 3]
    assert small.decode_all(ids, P.Keep) == ["<s>", "<s>", "hello world", "</s>", "hello", small.id_to_piece(3)]
    assert small.decode_all(ids, P.Ignore) == ["hello world", "hello"]
    assert small.decode_all([], P.Raise) == [] and small.decode_all([266, 267], P.Raise) == ["helloworld"]
    with pytest.raises(tk.TokenizerError) as e:
        small.decode_all(ids, P.Raise)
    assert e.value.kind == "SpecialTokenPolicy"
    assert "".join(small.decode_all(ids, P.Keep)) == small.decode(ids, P.Keep)
    # src/tekkenizer.rs:348-350: index = id, specials first, then the (lossy) string of every rank
    v = small.vocab()
    assert len(v) == small.vocab_size() == 268
    assert v[1] == "<s>" and v[266] == "hello" and v[267] == "world" and v[10 + ord("a")] == "a"
    assert all(v[i] == small.id_to_piece(i) for i in list(range(10 + 0x80)) + [266, 267])
    assert v[10 + 0x80] == "\ufffd"            # a lone continuation byte: the lossy string (src/tekkenizer.rs:150-160); id_to_piece fails there


def test_id_to_piece(tk, small):
    # reference src/tekkenizer.rs:617-695 ; tests/test_tokenizer_detailed.rs:15-55
    P = tk.SpecialTokenPolicy
    assert small.id_to_piece(266) == "hello" and small.id_to_piece(1) == "<s>"
    with pytest.raises(tk.TokenizerError) as e:
        small.id_to_piece(268)
    assert e.value.kind == "InvalidConfig"
    assert small.id_to_byte_piece(267, P.Raise) == b"world"
    assert small.id_to_byte_piece(1, P.Keep) == b"<s>" and small.id_to_byte_piece(1, P.Ignore) == b""
    with pytest.raises(tk.TokenizerError) as e:
        small.id_to_byte_piece(1, P.Raise)
    assert e.value.kind == "SpecialTokenPolicy"
    # a lone continuation byte is not UTF-8: the reference falls back to the lossy vocab string (:683-686)
    assert small.id_to_byte_piece(0x80 + 10, P.Raise) == "�".encode()


def test_encode_without_device_fails_loudly(tk, small):
    with pytest.raises(tk.TokenizerError):
        small.encode("hello", False, False)


def test_loader_validation(tk, small_vocab):
    toks = small_vocab["tokens"]

    def load(m):
        return tk.Tekkenizer.from_json(json.dumps(m), device=-1)

    def kind(m):
        with pytest.raises(tk.TokenizerError) as e:
            load(m)
        return e.value.kind

    assert kind(model(toks, version="v9")) == "InvalidConfig"                      # src/tekkenizer.rs:226-232
    assert kind(model(toks, vocab_size=400)) == "InvalidConfig"                    # :80-87
    assert kind(model(toks, specials=("<s>", "<s>"))) == "InvalidConfig"           # :90-98 duplicates
    assert kind(model(toks, num_special=2)) == "InvalidConfig"                     # :100-106 too many specials
    bad = list(toks); bad[65] = b"B"                                               # rank 65 must be byte 65 (:793-798)
    assert kind(model(bad)) == "InvalidConfig"
    dup = list(toks); dup[257] = b"hello"                                          # duplicate bytes -> not contiguous (:801-813)
    assert kind(model(dup)) == "InvalidConfig"
    m = model(toks); m["vocab"][256]["token_bytes"] = "!!!not base64"
    assert kind(m) == "Base64"                                                     # :789
    m = model(toks); del m["config"]["pattern"]
    assert kind(m) == "Json"                                                       # serde: no defaults (src/config.rs:38-49)
    m = model(toks); m["vocab"][3]["rank"] = -1
    assert kind(m) == "Json"
    with pytest.raises(tk.TokenizerError) as e:
        tk.Tekkenizer.from_json("{not json", device=-1)
    assert e.value.kind == "Json"
    with pytest.raises(tk.TokenizerError) as e:
        tk.Tekkenizer.from_file("/nonexistent/tekken.json", device=-1)
    assert e.value.kind == "Io"
    # audio present but the audio special tokens are missing (:158-178)
    audio = {"sampling_rate": 16000, "frame_rate": 12.5, "audio_encoding_config": {"num_mel_bins": 128, "hop_length": 160, "window_size": 400}, "chunk_length_s": None}
    assert kind(model(toks, audio=audio)) == "TokenNotFound"


def test_loader_semantics(tk, small_vocab):
    toks = small_vocab["tokens"]
    # truncation to the first vocab_size - num_special entries (T5, src/tekkenizer.rs:118-119,780-784)
    t = tk.Tekkenizer.from_json(json.dumps(model(toks + [b"zzz", b"yyy"], vocab_size=268)), device=-1)
    assert len(t.rank_table()) == 258 and t.vocab_size() == 268
    # the JSON pattern is ignored, unknown fields are tolerated, legacy specials when the field is absent (:234-237)
    m = model(toks, num_special=30, specials=None); m["extra"] = {"x": 1}
    t = tk.Tekkenizer.from_json(json.dumps(m), device=-1)
    assert t.get_control_token("[TOOL_CONTENT]") == 19 and t.pad_id() == 11 and t.get_control_token("<SPECIAL_20>") == 20
    # ranks may appear in any order in the file as long as they are contiguous (:804-813)
    m = model(toks); m["vocab"][256], m["vocab"][257] = m["vocab"][257], m["vocab"][256]
    t = tk.Tekkenizer.from_json(json.dumps(m), device=-1)
    assert t.rank_table()[256] == b"hello"
    # \u escapes incl. surrogate pairs in special token strings
    m = model(toks, specials=("<unk>", "<s>", "</s>", "🚀"))
    t = tk.Tekkenizer.from_json(json.dumps(m), device=-1)
    assert t.get_control_token("\U0001f680") == 3


def test_c_abi_exports(tk):
    """The library loads and exports every symbol include/tekken_hip.h declares."""
    import os
    import re
    hdr = open(os.path.join(tk.INCLUDE_DIR, "tekken_hip.h")).read()
    names = set(re.findall(r"\b(tk_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    L = tk.lib()
    for n in sorted(names):
        assert hasattr(L, n), n


def test_table_cache_round_trip(tmp_path, test_vocab, bench_vocab):
    """the derived device tables written to / read from a side file keyed by the rank table are
    identical, field by field, to freshly built ones; a wrong key, a truncated or a corrupted file is refused."""
    import emu
    r = emu.table_cache_roundtrip(test_vocab["tokens"], test_vocab["num_special"], str(tmp_path / "small.bin"))
    assert r["file_bytes"] > 0
    p = tmp_path / "bench.bin"
    r = emu.table_cache_roundtrip(bench_vocab["tokens"], bench_vocab["num_special"], str(p))
    assert r["load_s"] < r["build_s"]
    raw = p.read_bytes()
    for bad in (raw[:len(raw) // 2], raw[:-1], b"\x00" * 64):
        q = tmp_path / "bad.bin"
        q.write_bytes(bad)
        with pytest.raises(RuntimeError):   # the round trip's own load of a damaged file must fail -> rc != 0
            _load_only(emu, bench_vocab, str(q))


def _load_only(emu, v, path):
    import ctypes
    import numpy as np
    L = emu.lib()
    if not hasattr(L, "_tk_load_only"):
        L.emu_table_cache_load.restype = ctypes.c_int
        L._tk_load_only = True
    toks = v["tokens"]
    toffs = np.zeros(len(toks) + 1, np.uint32)
    toffs[1:] = np.cumsum([len(t) for t in toks], dtype=np.uint64).astype(np.uint32)
    blob = np.frombuffer(b"".join(toks), dtype=np.uint8).copy()
    rc = L.emu_table_cache_load(blob.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), toffs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                ctypes.c_uint32(len(toks)), ctypes.c_uint32(v["num_special"]), ctypes.c_char_p(path.encode()))
    if rc != 0:
        raise RuntimeError("refused")


def _piece_or_err(tk, t, i):
    try:
        return t.id_to_piece(i)
    except tk.TokenizerError as e:           # a lone non-ASCII byte is not a string (src/tekkenizer.rs:617-640)
        return ("error", e.kind)


def test_model_cache_from_file(tk, small_vocab, bench_vocab, tmp_path, monkeypatch):
    """Row f-2, file level: with TK_TABLE_CACHE_DIR set, from_file of the same bytes comes from the side file and the
    object behaves identically; other bytes get their own key; failed loads are never cached; a damaged side file is
    ignored."""
    import glob
    import time
    cache = tmp_path / "cache"
    cache.mkdir()
    P = tk.SpecialTokenPolicy
    specials = ("<unk>", "<s>", "</s>", "[INST]", "é☃")
    f1 = tmp_path / "a.json"
    f1.write_text(json.dumps(model(small_vocab["tokens"], specials=specials)))
    plain = tk.Tekkenizer.from_file(str(f1), device=-1)
    monkeypatch.setenv("TK_TABLE_CACHE_DIR", str(cache))
    first = tk.Tekkenizer.from_file(str(f1), device=-1)       # parses, writes
    side = glob.glob(str(cache / "tk_model_*.bin"))
    assert len(side) == 1
    second = tk.Tekkenizer.from_file(str(f1), device=-1)      # from the side file
    ids = [1, 266, 42, 129, 121, 124, 118, 110, 4, 2]
    assert not plain.from_cache() and not first.from_cache() and second.from_cache()
    for t in (first, second):
        assert t.json_pattern() == plain.json_pattern() == "ignored"       # config.pattern survives the side file
        assert (t.vocab_size(), t.num_special_tokens(), t.version()) == (plain.vocab_size(), plain.num_special_tokens(), plain.version())
        assert t.get_control_token("é☃") == 4 and t.get_control_token("<SPECIAL_7>") == 7
        assert t.decode(ids, P.Keep) == plain.decode(ids, P.Keep) == "<s>hello worldé☃</s>"
        assert [_piece_or_err(tk, t, i) for i in range(t.vocab_size())] == [_piece_or_err(tk, plain, i) for i in range(plain.vocab_size())]
    # different bytes (even if the same model) -> different key
    f2 = tmp_path / "b.json"
    f2.write_text(json.dumps(model(small_vocab["tokens"], specials=specials)) + "\n")
    tk.Tekkenizer.from_file(str(f2), device=-1)
    assert len(glob.glob(str(cache / "tk_model_*.bin"))) == 2
    # a file that fails to load leaves nothing behind and fails the same way again
    f3 = tmp_path / "c.json"
    f3.write_text(json.dumps(model(small_vocab["tokens"], version="v99")))
    for _ in range(2):
        with pytest.raises(tk.TokenizerError) as e:
            tk.Tekkenizer.from_file(str(f3), device=-1)
        assert e.value.kind == "InvalidConfig"
    assert len(glob.glob(str(cache / "tk_model_*.bin"))) == 2
    # damaged side file: ignored, rewritten
    raw = open(side[0], "rb").read()
    for bad in (raw[:len(raw) // 2], raw[:-2], raw[:40] + bytes([raw[40] ^ 1]) + raw[41:60]):
        open(side[0], "wb").write(bad)
        t = tk.Tekkenizer.from_file(str(f1), device=-1)
        assert t.decode(ids, P.Keep) == "<s>hello worldé☃</s>"
        assert open(side[0], "rb").read() == raw
    # load time on the bench-size vocabulary (the reference's profiling tests look at exactly this)
    t0 = time.perf_counter()
    a = tk.Tekkenizer.from_file(bench_vocab["path"], device=-1)
    t1 = time.perf_counter()
    b = tk.Tekkenizer.from_file(bench_vocab["path"], device=-1)
    t2 = time.perf_counter()
    print("from_file host-only: parse %.3f s, from cache %.3f s" % (t1 - t0, t2 - t1))   # (printed, not asserted: shared CI box)
    assert a.vocab_size() == b.vocab_size() and a.id_to_piece(100000) == b.id_to_piece(100000)
    assert not a.from_cache() and b.from_cache() and a.json_pattern() == b.json_pattern() != ""


def test_honour_pattern_needs_device_and_known_pattern(tk, small):
    """Row f-3 at the tokenizer level: the opt-in exists only for a device-backed object (the pattern is applied by the
    kernels); a host-only object refuses it loudly."""
    with pytest.raises(tk.TokenizerError) as e:
        small.set_honour_pattern(True)
    assert e.value.kind in ("NoDevice", "Tokenizers", "InvalidConfig")


def test_fewer_than_256_ranks_host_only_yes_device_no(tk, small_vocab):
    """A documented divergence.  The reference constructs a Tekkenizer from a rank table with fewer than 256 entries
    (reload_mergeable_ranks only checks the byte tokens that ARE there, src/tekkenizer.rs:793-798) and then panics inside
    tiktoken-rs on the first text byte whose single-byte token is missing.  Here the host-side mirror loads such a file
    like the reference does (loader, accessors, decode work), but a DEVICE context is refused at construction with
    InvalidConfig: the kernels take the rank of a single byte from the byte itself and have no panic to mirror."""
    import ctypes
    import numpy as np
    toks = small_vocab["tokens"][:100]
    t = tk.Tekkenizer.from_json(json.dumps(model(toks)), device=-1)
    assert t.vocab_size() == 110 and t.decode([10 + 65], tk.SpecialTokenPolicy.Ignore) == "A"
    t.close()
    blob = np.frombuffer(b"".join(toks), dtype=np.uint8).copy()
    offs = np.arange(len(toks) + 1, dtype=np.uint32)
    h = ctypes.c_void_p()
    rc = tk.lib().tk_ctx_create(blob.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                len(toks), 10, 1, 2, 0, ctypes.byref(h))
    assert rc == tk.TK_ERR_INVALID_CONFIG and b"256 single-byte" in tk.lib().tk_last_error(None)


def test_node_create_argument_checks(tk, small_vocab):
    """tk_node_create (csrc/tk_node.cpp): the device list is checked before any device is touched -- the same GPU listed
    twice is refused cleanly, as is an empty / oversized list; without a GPU a valid list fails with NoDevice (no CPU fallback)."""
    v = small_vocab
    for devs in ((0, 0), (1, 0, 1), ()):
        with pytest.raises(tk.TokenizerError) as e:
            tk.Node(v["tokens"], v["num_special"], v["bos"], v["eos"], devices=devs)
        assert e.value.kind == "InvalidConfig"           # TK_ERR_INVALID_ARG
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(tk.TokenizerError) as e:
            tk.Node(v["tokens"], v["num_special"], v["bos"], v["eos"], devices=(0,))
        assert e.value.kind in ("NoDevice", "Tokenizers")
