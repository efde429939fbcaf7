"""The oracle (oracle/tk_oracle.c) against the committed golden vectors.

* split_vectors.json: boundaries from Python `regex` on the literal pattern of reference
  src/tekkenizer.rs:123 (independent engine).
* reference_vectors.json: data of the reference's own tests; without the missing tekken.json only
  the vocab-free facts and the small-vocab known answer (App. B.3) are checkable.
"""
import tk_oracle


def test_split_matches_independent_engine(golden):
    bad = []
    for c in golden["split"]["cases"]:
        got = tk_oracle.split(c["text"].encode("utf-8"))
        if got != c["starts"]:
            bad.append((c["text"], c["starts"], got))
    assert not bad, bad[:3]


def test_split_live_against_python_regex():
    """Same comparison on fresh random strings when the `regex` module is importable."""
    regex = __import__("pytest").importorskip("regex")
    import helpers
    pat = regex.compile(r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")
    for doc in helpers.random_unicode_docs(3000, seed=11, max_len=60):
        exp = [m.group().encode("utf-8") for m in pat.finditer(doc.decode("utf-8"))]
        assert tk_oracle.split_pieces(doc) == exp, doc


def test_small_vocab_known_answer(golden, small_vocab):
    """reference tests/test_small_vocab.rs construction; ids hand-derived in SURVEY App. B.3."""
    sv = golden["ref"]["small_vocab"]
    assert [t.decode() for t in small_vocab["tokens"][256:]] == sv["extra_tokens"]
    o = tk_oracle.Oracle(small_vocab["tokens"], sv["num_special"], 1, 2)
    for text, bos, eos, ids in sv["cases"]:
        assert o.encode(text.encode(), bos, eos) == ids, text


def test_reference_vectors_vocab_free_facts(golden):
    """What the reference's golden id vectors pin WITHOUT the asset."""
    ref = golden["ref"]
    ns = ref["asset"]["num_special_tokens"]
    by_text = {t: ids for t, ids in ref["encode"]}

    def pieces(t):
        return tk_oracle.split_pieces(t.encode("utf-8"))

    # every piece is a whole-vocab hit in these rows => #pieces == #ids
    for t in ("Hello, world!", "The quick brown fox jumps over the lazy dog.", "Simple sentence.",
              "   whitespace   handling   ", "Hello", "world", "the"):
        assert len(pieces(t)) == len(by_text[t]), t
    assert pieces("   whitespace   handling   ") == [b"  ", b" whitespace", b"  ", b" handling", b"   "]
    ids = by_text["   whitespace   handling   "]
    assert ids[0] == ids[2]  # the same two-space piece twice

    # single-byte pieces must come out as byte tokens: id = byte + num_special (src/tekkenizer.rs:793-798, :390-392)
    def check_bytes(text):
        ps, ids = pieces(text), list(by_text[text])
        k = 0
        for p in ps:
            if len(p) == 1:
                assert ids[k] == p[0] + ns, (text, p)
                k += 1
            else:
                # advance over this piece's ids: unknown count, so re-sync on the next single-byte piece
                nxt = next((q for q in ps[ps.index(p) + 1:] if len(q) == 1), None)
                if nxt is None:
                    break
                while ids[k] != nxt[0] + ns:
                    k += 1

    check_bytes("Hello, world!")
    check_bytes("Simple sentence.")
    # numbers: \p{N}{1,3} pieces expand to single digits, each preceded by a stand-alone space (alt 7)
    t = "Another test case with numbers: 123, 456, 789."
    ps = pieces(t)
    assert ps == [b"Another", b" test", b" case", b" with", b" numbers", b":", b" ", b"123", b",", b" ", b"456", b",",
                  b" ", b"789", b"."]
    ids = by_text[t]
    assert ids[5:] == [ord(":") + ns, 32 + ns, 49 + ns, 50 + ns, 51 + ns, 44 + ns, 32 + ns, 52 + ns, 53 + ns, 54 + ns,
                       44 + ns, 32 + ns, 55 + ns, 56 + ns, 57 + ns, 46 + ns]
    # the cl100k pattern (not Mistral's case-aware one) is in effect: 3 pieces for 8 ids
    assert pieces("Mixed CaSe WoRdS") == [b"Mixed", b" CaSe", b" WoRdS"]
    # the best merge-order vector: 4 pieces -> 24 ids, the last piece has 30 bytes
    ps = pieces("Special characters: @#$%^&*()_+-={}[]|\\:;\"'<>,.?/")
    assert [len(p) for p in ps] == [7, 11, 1, 30]
    # special-token strings in the input are plain text (reference tests/test_integration.rs:259-291)
    assert b"".join(pieces("<s>[INST] hi [/INST]</s>")) == b"<s>[INST] hi [/INST]</s>"


def test_reference_vectors_on_consistent_vocab(golden):
    """The 20 reference vectors reproduced end-to-end on a vocabulary CONSTRUCTED to be consistent
    with them (tests/ref_consistent_vocab.py): pins split + whole-piece shortcut + merge order +
    id shift against ids taken from the reference's tests."""
    import ref_consistent_vocab as rcv
    toks = rcv.build(golden["ref"])
    o = tk_oracle.Oracle(toks, 1000, 1, 2)
    for text, ids in golden["ref"]["encode"]:
        assert o.encode(text.encode("utf-8"), False, False) == ids, text


def test_encode_properties(test_vocab):
    import helpers
    o = helpers.oracle_for(test_vocab)
    toks, ns = test_vocab["tokens"], test_vocab["num_special"]
    for doc in helpers.mixed_docs(20, 8, 20) + helpers.random_unicode_docs(300):
        ids = o.encode(doc, True, True)
        assert ids[0] == 1 and ids[-1] == 2
        body = ids[1:-1]
        assert all(i >= ns for i in body)
        assert b"".join(toks[i - ns] for i in body) == doc  # concatenated piece bytes == input
        assert o.encode(doc, False, False) == body          # BOS/EOS only add ids (src/tekkenizer.rs:394-402)
        assert o.encode(doc, True, False) == ids[:-1]


def test_batch_equals_single(test_vocab):
    import numpy as np
    import helpers
    o = helpers.oracle_for(test_vocab)
    docs = helpers.mixed_docs(10, 5, 10)
    data = np.frombuffer(b"".join(docs), dtype=np.uint8)
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    for threads in (1, 3):
        ids, oo = o.encode_batch(data, offs, True, True, threads=threads)
        for d, doc in enumerate(docs):
            assert ids[int(oo[d]):int(oo[d + 1])].tolist() == o.encode(doc, True, True)


def test_tekken_pattern_split_matches_independent_engine():
    """SURVEY 8 row f-3, groundwork: the oracle's matcher for the `pattern` string of Mistral's tekken.json (the one the
    reference ignores) against golden vectors from Python `regex` (tools/gen_golden_tekken.py) -- case-aware word
    splitting (Lu / Lt / Ll / Lm / Lo / M classes), single digits, '/' absorbed after punctuation."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "split_vectors_tekken.json")) as f:
        g = json.load(f)
    assert "\\p{Lu}" in g["pattern"] and len(g["cases"]) > 900
    for c in g["cases"]:
        assert tk_oracle.split_tekken(c["text"].encode("utf-8")) == c["starts"], c["text"]
    # the two patterns differ where they should
    assert tk_oracle.split(b"HelloWorld 1234") == [0, 10, 11, 14] and tk_oracle.split_tekken(b"HelloWorld 1234") == [0, 5, 10, 11, 12, 13, 14]


def test_tekken_pattern_live_against_python_regex():
    import random
    import pytest
    regex = pytest.importorskip("regex")
    import synth_vocab as sv
    R = regex.compile(sv.MISTRAL_PATTERN)
    rng = random.Random(8)
    alpha = list("aAbB zZ") + ["é", "É", "ǅ", "ʰ", "中", "ª", "́", "̈", "1", "٣", "²",
                               " ", "\n", "\r", "\t", "!", "/", "-", "'", "_", " ", "　", "\U0001f642"]
    for _ in range(20000):
        s = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 16)))
        exp = [len(s[:m.start()].encode()) for m in R.finditer(s)]
        assert tk_oracle.split_tekken(s.encode()) == exp, repr(s)


def test_corpora_and_golden_texts_avoid_unicode_drift():
    """SURVEY 7.3-3 / trap T11: the reference's regex-syntax (its Unicode tables) is unpinned, ours are `regex`'s (Unicode 17 in this image).
    The only code points on which the two splits could differ without either being wrong are those whose L / N class changed after the
    reference's version -- bounded here by everything that differs from Unicode 13.0 (python's unicodedata), listed in
    tests/golden/unicode_drift.json by tools/unicode_drift.py.  No bench corpus alphabet, no golden text and no test alphabet may use
    one of them: every parity claim in this repository is then independent of the Unicode version."""
    import json
    import os
    import numpy as np
    import corpus
    import helpers
    root = os.path.dirname(os.path.abspath(__file__))
    drift = json.load(open(os.path.join(root, "golden", "unicode_drift.json")))
    assert drift["unicodedata_version"].startswith("13.") and drift["code_points"] > 0
    bad = np.zeros(0x110000, bool)
    for lo, hi, _, _ in drift["ranges_lo_hi_class13_classNow"]:
        bad[lo:hi + 1] = True

    def check(text, what):
        cps = np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
        hit = np.unique(cps[bad[np.minimum(cps, 0x10FFFF)]])
        assert len(hit) == 0, "%s uses code points whose class depends on the Unicode version: %s" % (what, [hex(int(c)) for c in hit[:8]])

    for kind, n, dl in (("ascii", 2000, 512), ("mixed", 4000, 2048), ("zipf", 3000, 0)):
        data, offs = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + 2)
        check(data.tobytes().decode("utf-8"), "the %s bench corpus" % kind)
    check(" ".join(corpus.words()), "the corpus word list")

    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for k, v in x.items():
                yield from strings(k)
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)

    gdir = os.path.join(root, "golden")
    for name in sorted(os.listdir(gdir)):
        if name.endswith(".json") and name != "unicode_drift.json":
            for s in strings(json.load(open(os.path.join(gdir, name)))):
                check(s, "tests/golden/" + name)
    for d in list(helpers.EDGE_DOCS) + helpers.random_unicode_docs(200):
        check(d.decode("utf-8", "replace"), "the test alphabets of tests/helpers.py")
