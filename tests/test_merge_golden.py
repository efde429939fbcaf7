"""The byte-pair MERGE loop against tests/golden/merge_vectors.json.

The vectors come from tools/gen_golden_merge.py: an INDEPENDENT RESTATEMENT (not the reference, whose merge loop
lives in the absent crate tiktoken-rs) of tiktoken's published algorithm in list-of-parts form, written separately
from oracle/tk_oracle.c's (start, rank)-array form, on adversarial vocabularies (tokens no merge sequence reaches,
runs of equal pairs, random rank orders, UTF-8 tokens) with pieces of every length class of the merge kernels
(2..8, 9..16, 17..32, 33..64, > 64 bytes).  Checked here: the oracle and the device source on the CPU wave emulator
(both pipelines); on the GPU (-m gpu) the HIP path through the C ABI.
"""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "merge_vectors.json")) as f:
        g = json.load(f)
    out = []
    for v in g["vocabs"]:
        toks = [bytes.fromhex(t) for t in v["tokens_hex"]]
        docs = [bytes.fromhex(p) for p, _ in v["pieces"]] + [t.encode("utf-8") for t, _ in v["texts"]]
        exp = [ids for _, ids in v["pieces"]] + [ids for _, ids in v["texts"]]
        out.append({"name": v["name"], "tokens": toks, "num_special": v["num_special"], "docs": docs, "exp": exp})
    assert len(out) >= 5 and sum(len(v["docs"]) for v in out) > 2000
    return out


def _diff(name, docs, exp, got):
    for d, e, g in zip(docs, exp, got):
        assert g == e, (name, d[:80], e[:12], g[:12])


def test_generator_is_deterministic_and_self_contained():
    """The committed file is what the committed script writes (the script imports nothing of oracle/ or the package)."""
    src = open(os.path.join(HERE, "..", "tools", "gen_golden_merge.py")).read()
    for banned in ("tk_oracle", "tekken-rs_amd", "synth_vocab", "import corpus"):
        assert banned not in src.replace("oracle/tk_oracle.c", "").replace("tools/synth_vocab.py", "").replace("tekken-rs_amd/", ""), banned


def test_oracle_matches_independent_merge_vectors(vectors):
    import tk_oracle
    for v in vectors:
        o = tk_oracle.Oracle(v["tokens"], v["num_special"], 1, 2)
        _diff(v["name"], v["docs"], v["exp"], [o.encode(d, False, False) for d in v["docs"]])
        # BOS / EOS only add ids around the same sequence (src/tekkenizer.rs:394-402)
        d0, e0 = v["docs"][0], v["exp"][0]
        assert o.encode(d0, True, True) == [1] + e0 + [2]


def test_emulator_matches_independent_merge_vectors(vectors):
    """The device source (tk_flat_impl.h / tk_encode_impl.h) on the CPU wave emulator, flat and per-document path."""
    import emu
    for v in vectors:
        ids, _, _ = emu.flat_encode_batch(v["tokens"], v["num_special"], 1, 2, v["docs"], False, False)
        _diff(v["name"] + " (flat)", v["docs"], v["exp"], ids)
        ids2, _, _ = emu.encode_batch(v["tokens"], v["num_special"], 1, 2, v["docs"], False, False)
        _diff(v["name"] + " (per-document)", v["docs"], v["exp"], ids2)


@pytest.mark.gpu
def test_gpu_matches_independent_merge_vectors(tk, vectors, monkeypatch):
    for pipeline in ("flat", "doc"):
        monkeypatch.setenv("TK_PIPELINE", pipeline)
        for v in vectors:
            e = tk.Engine(v["tokens"], v["num_special"], 1, 2, device=0)
            _diff("%s (%s)" % (v["name"], pipeline), v["docs"], v["exp"], e.encode_docs(v["docs"], False, False))
            # the same documents in reverse order and with BOS / EOS: every batch position, both flags
            rev = e.encode_docs(v["docs"][::-1], True, True)
            _diff("%s (%s, reversed)" % (v["name"], pipeline), v["docs"][::-1], [[1] + x + [2] for x in v["exp"][::-1]], rev)
            e.close()
