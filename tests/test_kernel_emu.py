"""The DEVICE SOURCE (tekken-rs_amd/csrc/tk_encode_impl.h) executed on the CPU wave emulator
(tests/emu) and compared with the oracle: the same code hipcc compiles for gfx950, minus the
hardware.  The emulator aborts if lanes reach a wave primitive in non-uniform control flow."""
import emu
import helpers


def test_emu_small_vocab_known_answer(golden, small_vocab):
    sv = golden["ref"]["small_vocab"]
    for text, bos, eos, ids in sv["cases"]:
        got, _, _ = emu.encode_batch(small_vocab["tokens"], sv["num_special"], 1, 2, [text.encode()], bos, eos)
        assert got[0] == ids, text


def test_emu_split_only(golden, small_vocab):
    cases = golden["split"]["cases"]
    docs = [c["text"].encode("utf-8") for c in cases]
    _, starts, _ = emu.encode_batch(small_vocab["tokens"], 10, 1, 2, docs, split_only=True)
    for c, s in zip(cases, starts):
        assert s == c["starts"], c["text"]


def test_emu_encode_matches_oracle(test_vocab):
    o = helpers.oracle_for(test_vocab)
    docs = helpers.mixed_docs(25, 10, 40, max_len=3000) + helpers.random_unicode_docs(150)
    for bos, eos in ((True, True), (False, False)):
        got, _, n_def = emu.encode_batch(test_vocab["tokens"], test_vocab["num_special"], 1, 2, docs, bos, eos)
        for doc, g in zip(docs, got):
            assert g == o.encode(doc, bos, eos), doc[:80]
    assert n_def > 0  # the long-piece (second pass) path was exercised


def test_emu_reference_vectors_on_consistent_vocab(golden):
    import ref_consistent_vocab as rcv
    toks = rcv.build(golden["ref"])
    texts = [t for t, _ in golden["ref"]["encode"]]
    got, _, _ = emu.encode_batch(toks, 1000, 1, 2, [t.encode("utf-8") for t in texts], False, False)
    for (t, ids), g in zip(golden["ref"]["encode"], got):
        assert g == ids, t


def test_emu_split_exhaustive_ascii(small_vocab):
    """Every string of length <= 5 over an 8-symbol ASCII alphabet through the device split (the
    scalar mask-algebra path of ASCII windows), against the oracle."""
    import itertools
    import tk_oracle
    alpha = ["a", "s", "1", "'", "!", " ", "\n", "\t"]
    docs = ["".join(t).encode() for n in range(0, 6) for t in itertools.product(alpha, repeat=n)]
    _, starts, _ = emu.encode_batch(small_vocab["tokens"], 10, 1, 2, docs, split_only=True)
    for d, s in zip(docs, starts):
        assert s == tk_oracle.split(d), d


def test_emu_split_long_ascii_windows(small_vocab):
    """Multi-window ASCII documents: digit runs (prefix doubling), long white-space runs with CR/LF,
    runs crossing window boundaries."""
    import random
    import tk_oracle
    rng = random.Random(21)
    alpha = ["a", "b", "1", "2", "'", "s", "t", "!", "-", " ", " ", " ", "\n", "\r", "\t"]
    docs = []
    for _ in range(400):
        n = rng.randint(60, 400)
        parts = []
        while sum(map(len, parts)) < n:
            c = rng.choice(alpha)
            parts.append(c * rng.choice([1, 1, 1, 2, 3, 5, 9, 17, 40]))
        docs.append("".join(parts).encode())
    _, starts, _ = emu.encode_batch(small_vocab["tokens"], 10, 1, 2, docs, split_only=True)
    for d, s in zip(docs, starts):
        assert s == tk_oracle.split(d), d


def test_emu_long_single_piece(test_vocab):
    """One piece of several thousand bytes (pass 2, the cooperative merge with its block minima in registers: more than
    64 blocks, so more than one register slot per lane), random and periodic content."""
    import random
    rng = random.Random(9)
    o = helpers.oracle_for(test_vocab)
    docs = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(4500)).encode(), b"ab" * 2300, b"x" * 4200 + b"yz"]
    got, _, n_def = emu.encode_batch(test_vocab["tokens"], test_vocab["num_special"], 1, 2, docs, True, True)
    assert n_def == len(docs)
    for doc, g in zip(docs, got):
        assert g == o.encode(doc, True, True)


def test_emu_json_pattern_opt_in(test_vocab):
    """Row f-3: the device's sequential matcher for the JSON pattern of Mistral's tekken.json (tk_match_end2, pass 2),
    on the emulator, id for id against the oracle in the same mode -- golden texts, case mixes, Unicode categories."""
    import json
    import os
    import tk_oracle
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "split_vectors_tekken.json")) as f:
        g = json.load(f)
    docs = [c["text"].encode("utf-8") for c in g["cases"][:400]] + [b"HelloWorld XMLHttpRequest 1234 a/b/c\n", b""]
    o = tk_oracle.Oracle(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"])
    o.set_pattern(1)
    got, _, _ = emu.encode_batch(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], docs, True, True,
                                 pattern=1)
    for doc, ids in zip(docs, got):
        assert ids == o.encode(doc, True, True), doc[:60]
    # and it differs from the reference's behaviour where it should
    plain = helpers.oracle_for(test_vocab)
    assert plain.encode(b"HelloWorld 1234", False, False) != o.encode(b"HelloWorld 1234", False, False)
