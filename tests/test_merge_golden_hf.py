"""Oracle, emulator and GPU against tests/golden/merge_vectors_hf.json: ids from a THIRD-PARTY engine.

Label: independent engine, different authors, NOT the reference.  tools/gen_golden_hf.py builds the vocabulary (BPE training),
splits the text (Oniguruma on the literal pattern of reference src/tekkenizer.rs:123) and runs the merge loop (merge-list BPE
with the whole-pre-token look-up first) entirely inside HuggingFace `tokenizers`; it keeps only texts on which tiktoken's
bytes-keyed ranks and the merge list provably pick the same merges (checked per pre-token by the generator; the file says
how many it dropped).  What this pins that nothing else here does: split AND merge order against code this repository's
author did not write.  It does not turn id-level parity against the reference binary green -- only the asset and a Rust
toolchain can (tests/test_real_asset.py).
"""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hf():
    with open(os.path.join(HERE, "golden", "merge_vectors_hf.json")) as f:
        g = json.load(f)
    assert "NOT the reference" in g["label"] and "HuggingFace tokenizers" in g["label"]
    toks = [bytes.fromhex(t) for t in g["tokens_hex"]]
    docs = [bytes.fromhex(v["text_hex"]) for v in g["vectors"]]
    exp = [v["ids"] for v in g["vectors"]]
    assert len(docs) > 250 and g["n_excluded_outside_class"] * 10 < g["n_texts"]
    return {"tokens": toks, "num_special": g["num_special"], "docs": docs, "exp": exp}


def _diff(name, docs, exp, got):
    for d, e, g in zip(docs, exp, got):
        assert g == e, (name, d[:80], e[:12], g[:12])


def test_generator_uses_only_the_third_party_engine():
    """Neither the oracle nor the package nor this repository's BPE trainer is imported by the generator: the expected ids are
    HuggingFace's."""
    src = open(os.path.join(HERE, "..", "tools", "gen_golden_hf.py")).read()
    for banned in ("import tk_oracle", "tekken-rs_amd", "import synth_vocab", "import gen_golden_merge"):
        assert banned not in src, banned
    assert "from tokenizers import" in src


def test_oracle_matches_hf_vectors(hf):
    import tk_oracle
    o = tk_oracle.Oracle(hf["tokens"], hf["num_special"], 1, 2)
    _diff("oracle", hf["docs"], hf["exp"], [o.encode(d, False, False) for d in hf["docs"]])
    assert o.encode(hf["docs"][0], True, True) == [1] + hf["exp"][0] + [2]


def test_emulator_matches_hf_vectors(hf):
    """The device source on the CPU wave emulator, flat and per-document path (every third text: the emulator is slow)."""
    import emu
    docs, exp = hf["docs"][::3], hf["exp"][::3]
    ids, _, _ = emu.flat_encode_batch(hf["tokens"], hf["num_special"], 1, 2, docs, False, False)
    _diff("emulator (flat)", docs, exp, ids)
    ids2, _, _ = emu.encode_batch(hf["tokens"], hf["num_special"], 1, 2, docs[::2], False, False)
    _diff("emulator (per-document)", docs[::2], exp[::2], ids2)


@pytest.mark.gpu
def test_gpu_matches_hf_vectors(tk, hf, monkeypatch):
    for pipeline in ("flat", "doc"):
        monkeypatch.setenv("TK_PIPELINE", pipeline)
        e = tk.Engine(hf["tokens"], hf["num_special"], 1, 2, device=0)
        _diff("GPU (%s)" % pipeline, hf["docs"], hf["exp"], e.encode_docs(hf["docs"], False, False))
        rev = e.encode_docs(hf["docs"][::-1], True, True)
        _diff("GPU (%s, reversed)" % pipeline, hf["docs"][::-1], [[1] + x + [2] for x in hf["exp"][::-1]], rev)
        e.close()
