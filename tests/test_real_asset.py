"""Asset-gated closure of id-level parity: the reference's OWN golden id vectors on a real tekken.json.

The 20 exact `encode(text, false, false)` vectors of reference tests/test_tokenizer_output.rs:22-373 and the decode
known answer of tests/test_rust_tokenizer.rs:16-19 (data in tests/golden/reference_vectors.json) need the asset
tests/assets/tekken.json, which is absent from the reference mount.  Point

    TEKKEN_JSON=/path/to/tekken.json            (the asset: vocab_size 131072, version v7 -- asserted)
    TEKKEN_EXPECTED=/path/to/vectors.json       (optional: another file of the same layout; then the asset facts asserted
                                                 are the ones in ITS "asset" object)

at this file and every vector runs through the product loader (`Tekkenizer.from_file`, host-only), the oracle, the
device source on the CPU wave emulator and -- with `-m gpu` -- the HIP path.  Without TEKKEN_JSON those tests skip.

`test_gate_runs_on_synthetic_asset` shows the gate working end to end without the real asset: it writes an expected
file for the SYNTHETIC tekken.json (ids from the independent list-of-parts restatement of tools/gen_golden_merge.py,
split by Python `regex` -- neither the oracle nor the product), sets the two variables and runs the same checks.
"""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_EXPECTED = os.path.join(HERE, "golden", "reference_vectors.json")


def _load(json_path, expected_path, tk):
    with open(expected_path) as f:
        exp = json.load(f)
    t = tk.Tekkenizer.from_file(json_path, device=-1)            # the product's loader (src/tekkenizer.rs:222-248)
    asset = exp["asset"]
    assert t.vocab_size() == asset["vocab_size"], (t.vocab_size(), asset)            # tests/test_tokenizer_output.rs:18
    assert t.version() == asset["version"]                                          # :19
    assert t.num_special_tokens() == asset["num_special_tokens"]
    return exp, t


def check_oracle(json_path, expected_path, tk):
    import tk_oracle
    exp, t = _load(json_path, expected_path, tk)
    o = tk_oracle.Oracle(t.rank_table(), t.num_special_tokens(), t.bos_id(), t.eos_id())
    for text, ids in exp["encode"]:
        assert o.encode(text.encode("utf-8"), False, False) == ids, text
    # decode known answer through the host-side decode mirror (Keep / Ignore must agree up to the trailing </s>)
    dec = exp.get("decode")
    if dec:
        assert t.decode(dec["ids"], tk.SpecialTokenPolicy.Ignore) == dec["text"]
        body = [i for i in dec["ids"] if i >= t.num_special_tokens()]
        assert o.encode(dec["text"].encode("utf-8"), False, False) == body
    t.close()
    return len(exp["encode"])


def check_emulator(json_path, expected_path, tk):
    import emu
    exp, t = _load(json_path, expected_path, tk)
    toks, ns = t.rank_table(), t.num_special_tokens()
    docs = [text.encode("utf-8") for text, _ in exp["encode"]]
    ids, _, _ = emu.flat_encode_batch(toks, ns, t.bos_id(), t.eos_id(), docs, False, False)
    for (text, e), g in zip(exp["encode"], ids):
        assert g == e, text
    t.close()
    return len(docs)


def check_gpu(json_path, expected_path, tk):
    with open(expected_path) as f:
        exp = json.load(f)
    t = tk.Tekkenizer.from_file(json_path, device=0)
    assert t.vocab_size() == exp["asset"]["vocab_size"] and t.version() == exp["asset"]["version"]
    for text, ids in exp["encode"]:                              # one call per text: the reference's own signature
        assert t.encode(text, False, False) == ids, text
    got = t.encode_batch([text for text, _ in exp["encode"]], False, False)
    assert got == [ids for _, ids in exp["encode"]]
    dec = exp.get("decode")
    if dec:
        assert t.decode(dec["ids"], tk.SpecialTokenPolicy.Ignore) == dec["text"]
        assert t.decode_batch([dec["ids"]], tk.SpecialTokenPolicy.Ignore) == [dec["text"]]
    t.close()
    return len(exp["encode"])


def _env():
    p = os.environ.get("TEKKEN_JSON", "")
    if not p or not os.path.exists(p):
        pytest.skip("TEKKEN_JSON is not set (the reference's tests/assets/tekken.json is absent from the mount)")
    return p, os.environ.get("TEKKEN_EXPECTED", "") or DEFAULT_EXPECTED


def test_real_asset_oracle(tk):
    p, e = _env()
    assert check_oracle(p, e, tk) >= 1


def test_real_asset_emulator(tk):
    p, e = _env()
    assert check_emulator(p, e, tk) >= 1


@pytest.mark.gpu
def test_real_asset_gpu(tk):
    p, e = _env()
    assert check_gpu(p, e, tk) >= 1


# ---------------------------------------------------------------------------------------------------------------
# the gate itself, demonstrated on the synthetic asset
# ---------------------------------------------------------------------------------------------------------------
SYN_TEXTS = ["Hello, world!", "The quick brown fox jumps over the lazy dog.", "tokenizer", "Rust", "decoding",
             "Another test case with numbers: 123, 456, 789.", "   whitespace   handling   ", "Mixed CaSe WoRdS",
             "Special characters: @#$%^&*()_+-={}[]|\\:;\"'<>,.?/", "it's we'Re x'ſ", "Zq Xj QQQQ unwordish kjhgf", "a" * 40, ""]


def synthetic_expected(bench_vocab, tmp_path):
    """Expected file for the synthetic tekken.json, ids from the independent restatement (tools/gen_golden_merge.py)."""
    import gen_golden_merge as ggm
    toks, ns = bench_vocab["tokens"], bench_vocab["num_special"]
    ranks = {t: i for i, t in enumerate(toks)}
    enc = [[t, ggm.encode_text(ranks, t, ns)] for t in SYN_TEXTS]
    ids = [bench_vocab["bos"]] + enc[1][1] + [bench_vocab["eos"]]
    with open(bench_vocab["path"]) as f:
        cfg = json.load(f)["config"]
    exp = {"source": "synthetic asset; ids from tools/gen_golden_merge.py (independent restatement, not the reference)",
           "asset": {"vocab_size": cfg["default_vocab_size"], "version": cfg["version"], "num_special_tokens": ns},
           "encode": enc, "decode": {"ids": ids, "text": SYN_TEXTS[1]}}
    path = os.path.join(str(tmp_path), "synthetic_expected.json")
    with open(path, "w") as f:
        json.dump(exp, f)
    return path


def test_gate_runs_on_synthetic_asset(tk, bench_vocab, tmp_path, monkeypatch):
    pytest.importorskip("regex")
    path = synthetic_expected(bench_vocab, tmp_path)
    monkeypatch.setenv("TEKKEN_JSON", bench_vocab["path"])
    monkeypatch.setenv("TEKKEN_EXPECTED", path)
    p, e = _env()                                               # the same gate the real-asset tests go through
    assert (p, e) == (bench_vocab["path"], path)
    assert check_oracle(p, e, tk) == len(SYN_TEXTS)
    assert check_emulator(p, e, tk) == len(SYN_TEXTS)
    # and the gate does fail when an id is wrong
    with open(path) as f:
        exp = json.load(f)
    exp["encode"][0][1][0] += 1
    with open(path, "w") as f:
        json.dump(exp, f)
    with pytest.raises(AssertionError):
        check_oracle(p, e, tk)


@pytest.mark.gpu
def test_gate_runs_on_synthetic_asset_gpu(tk, bench_vocab, tmp_path, monkeypatch):
    pytest.importorskip("regex")
    path = synthetic_expected(bench_vocab, tmp_path)
    monkeypatch.setenv("TEKKEN_JSON", bench_vocab["path"])
    monkeypatch.setenv("TEKKEN_EXPECTED", path)
    p, e = _env()
    assert check_gpu(p, e, tk) == len(SYN_TEXTS)
