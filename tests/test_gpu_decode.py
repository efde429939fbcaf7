"""GPU batch decode through the C ABI against a restatement of the
reference's Tekkenizer::decode (oracle/tk_oracle.py decode_ref, reference src/tekkenizer.rs:436-560):
SpecialTokenPolicy behaviour, per-run UTF-8 validity, error classes, round trips at full size."""
import json

import numpy as np
import pytest

import corpus
import helpers
import tk_oracle

pytestmark = pytest.mark.gpu

SPECIALS = ["<unk>", "<s>", "</s>", "[INST]", "[/INST]", "é🚀"] + ["<SPECIAL_%d>" % i for i in range(6, 1000)]


@pytest.fixture(scope="module")
def eng(tk, test_vocab):
    e = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    e.set_special_tokens(SPECIALS)
    yield e
    e.close()


def ref(test_vocab, ids, policy):
    return tk_oracle.decode_ref(test_vocab["tokens"], SPECIALS, test_vocab["num_special"], ids, policy)


def test_round_trip_and_policies(tk, eng, test_vocab):
    orc = helpers.oracle_for(test_vocab)
    docs = helpers.mixed_docs(80, 30, 60, max_len=20000) + helpers.random_unicode_docs(300)
    id_lists = [orc.encode(d, True, True) for d in docs]
    P = tk.SpecialTokenPolicy
    assert eng.decode_docs(id_lists, P.Ignore) == docs                      # decode(encode(x)) == x (tests/test_tekken.rs)
    kept = eng.decode_docs(id_lists, P.Keep)
    assert kept == [b"<s>" + d + b"</s>" for d in docs]
    with pytest.raises(tk.TokenizerError) as e:
        eng.decode_docs(id_lists, P.Raise)
    assert e.value.kind == "SpecialTokenPolicy" and e.value.bad_doc == 0
    body = [x[1:-1] for x in id_lists]
    assert eng.decode_docs(body, P.Raise) == docs
    assert eng.decode_docs([], P.Ignore) == [] and eng.decode_docs([[], [1, 2], []], P.Ignore) == [b"", b"", b""]
    assert eng.decode_docs([[1, 5, 2]], P.Keep) == ["<s>é🚀</s>".encode()]


def test_per_run_utf8_and_error_classes(tk, eng, test_vocab):
    ns = test_vocab["num_special"]
    P = tk.SpecialTokenPolicy
    c3, a9 = ns + 0xC3, ns + 0xA9
    assert eng.decode_docs([[c3, a9]], P.Ignore) == ["é".encode()]
    # a special token splits the code point into two runs: each run alone is invalid (src/tekkenizer.rs:552-555)
    for pol in (P.Ignore, P.Keep):
        with pytest.raises(tk.TokenizerError) as e:
            eng.decode_docs([[ns + 97], [c3, 1, a9], [ns + 98]], pol)
        assert e.value.kind == "Tokenizers" and e.value.bad_doc == 1
    cases = [[c3], [a9], [ns + 0xE2, ns + 0x82], [ns + 0xC0, ns + 0xAF], [ns + 0xED, ns + 0xA0, ns + 0x80],
             [ns + 0xF4, ns + 0x90, ns + 0x80, ns + 0x80], [ns + 97, ns + 0x80, ns + 98], [ns + 0xFF]]
    for ids in cases:
        with pytest.raises(ValueError):
            ref(test_vocab, ids, 0)
        with pytest.raises(tk.TokenizerError) as e:
            eng.decode_docs([[ns + 120], ids], P.Ignore)
        assert e.value.kind == "Tokenizers" and e.value.bad_doc == 1, ids
    with pytest.raises(tk.TokenizerError) as e:                              # id outside the vocabulary
        eng.decode_docs([[ns + len(test_vocab["tokens"]) + 5]], P.Ignore)
    assert e.value.kind == "Tokenizers"
    # the FIRST offending group decides the class: Raise-special before / after an invalid run
    with pytest.raises(tk.TokenizerError) as e:
        eng.decode_docs([[1, c3]], P.Raise)
    assert e.value.kind == "SpecialTokenPolicy"
    with pytest.raises(tk.TokenizerError) as e:
        eng.decode_docs([[c3, 1]], P.Raise)
    assert e.value.kind == "Tokenizers"


def test_random_id_sequences_match_reference_restatement(tk, eng, test_vocab):
    rng = np.random.default_rng(3)
    ns, nr = test_vocab["num_special"], len(test_vocab["tokens"])
    for policy in (0, 1, 2):
        for _ in range(150):
            n = int(rng.integers(0, 30))
            ids = [int(rng.integers(0, 6)) if rng.random() < 0.15 else ns + int(rng.integers(0, nr)) for _ in range(n)]
            try:
                exp = ref(test_vocab, ids, policy)
            except ValueError as ex:
                with pytest.raises(tk.TokenizerError) as e:
                    eng.decode_docs([ids], policy)
                assert (e.value.kind == "SpecialTokenPolicy") == (str(ex) == "special"), (ids, str(ex))
                continue
            assert eng.decode_docs([ids], policy) == [exp], ids


def test_tokenizer_level_batch_decode(tk, small_vocab):
    from test_host_tokenizer import model
    t = tk.Tekkenizer.from_json(json.dumps(model(small_vocab["tokens"])), device=0)
    ids = t.encode_batch(["hello world", "", "héllo 🚀"], True, True)
    assert t.decode_batch(ids, tk.SpecialTokenPolicy.Ignore) == ["hello world", "", "héllo 🚀"]
    assert t.decode_batch(ids, tk.SpecialTokenPolicy.Keep)[0] == "<s>hello world</s>"
    assert [t.decode(x, tk.SpecialTokenPolicy.Keep) for x in ids] == t.decode_batch(ids, tk.SpecialTokenPolicy.Keep)
    t.close()


def test_full_size_round_trip_device_resident(tk, bench_vocab):
    """C2 at full size: encode 1 M x 512 B on the GPU, decode the ids on the GPU, compare bytes and offsets."""
    import torch
    e = tk.Engine(bench_vocab["tokens"], bench_vocab["num_special"], bench_vocab["bos"], bench_vocab["eos"], device=0)
    n_docs = 1_000_000
    data, offs = corpus.generate("ascii", n_docs, 512, seed=corpus.BASE_SEED + 1)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    v_ids, v_oo = e.encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, len(data), True, True, st)
    ids = torch.as_tensor(v_ids, device="cuda").clone()
    oo = torch.as_tensor(v_oo, device="cuda").clone()
    v_b, v_bo = e.decode_batch_device(ids.data_ptr(), oo.data_ptr(), n_docs, ids.numel(), tk.SpecialTokenPolicy.Ignore, st)
    out = torch.as_tensor(v_b, device="cuda")
    out_offs = torch.as_tensor(v_bo, device="cuda")
    assert out.numel() == len(data) and torch.equal(out, d_bytes)
    assert torch.equal(out_offs, d_offs)
    assert e.last_timing()["pipeline_ms"] > 0
    e.close()


def test_utf8_validation_at_every_window_offset(tk, eng, test_vocab):
    """tk_decode_validate_kernel judges 256 bytes per step, one aligned dword per lane, every lane inside a twelve-byte window
    {left neighbour's dword, own, right neighbour's} (round 2: 64 bytes per step, windows overlapping by six): every kind of
    sequence -- valid 2 / 3 / 4-byte code points, a lone continuation byte, a truncated lead, overlong forms, a surrogate, a code
    point beyond U+10FFFF, a special token inside a code point (two runs, each invalid alone) -- at every offset 0..130, 236..279
    and 500..519 of a document, so that each straddles the dword, lane and step seams (at 4, 256 and 512 bytes from the
    document's aligned start, which the documents in front shift through all four residues) in every way; the failing document
    is the one python's own decoder rejects."""
    ns = test_vocab["num_special"]
    P = tk.SpecialTokenPolicy

    def byte_ids(bs):
        return [ns + x for x in bs]

    valid = ["é".encode(), "中".encode(), "\U0001f680".encode(), "é中\U0001f680".encode()]
    docs = []
    offsets = list(range(0, 131)) + list(range(236, 280)) + list(range(500, 520))
    for k in offsets:
        for v in valid:
            docs.append(byte_ids(b"a" * k + v + b"b" * 70 + v))
    texts = eng.decode_docs(docs, P.Ignore)
    assert all(t == bytes(x - ns for x in ids) for t, ids in zip(texts, docs))
    bad = [bytes([0x80]), bytes([0xC3]), bytes([0xE4, 0xB8]), bytes([0xF0, 0x9F, 0x9A]), bytes([0xC0, 0xAF]), bytes([0xE0, 0x80, 0xAF]),
           bytes([0xED, 0xA0, 0x80]), bytes([0xF4, 0x90, 0x80, 0x80]), bytes([0xF8, 0x88, 0x80, 0x80]), bytes([0xE4, 0xB8, 0x41]),
           bytes([0xC3, 0xC3, 0xA9])]
    filler = [byte_ids(("zé中" * 40).encode()) for _ in range(5)]
    for k in offsets:
        for bseq in bad[k % 3::3]:                       # (a third of the kinds per offset: every kind meets every residue of 58)
            raw = b"a" * k + bseq + b"b" * 9
            with pytest.raises(UnicodeDecodeError):
                raw.decode("utf-8")
            batch = filler[:3] + [byte_ids(raw)] + filler[3:]
            with pytest.raises(tk.TokenizerError) as e:
                eng.decode_docs(batch, P.Ignore)
            assert e.value.kind == "Tokenizers" and e.value.bad_doc == 3, (k, bseq)
        # a special id between the bytes of a code point: two runs, each invalid on its own (src/tekkenizer.rs:552-555)
        ids = byte_ids(b"a" * k) + [ns + 0xE4, ns + 0xB8, 1, ns + 0xAD] + byte_ids(b"b" * 9)
        with pytest.raises(tk.TokenizerError) as e:
            eng.decode_docs(filler[:2] + [ids], P.Ignore)
        assert e.value.kind == "Tokenizers" and e.value.bad_doc == 2, k
        # ... and the same bytes without the special id are fine
        ok = byte_ids(b"a" * k) + [ns + 0xE4, ns + 0xB8, ns + 0xAD] + byte_ids(b"b" * 9)
        assert eng.decode_docs([ok], P.Ignore) == [b"a" * k + "中".encode() + b"b" * 9]


def test_length_pass_by_groups_and_by_documents_agree(tk, test_vocab, monkeypatch):
    """The decode pipeline sizes its output from the text lengths of GROUPS of 16 documents (tk_decode_grouplen_kernel) and lets the
    emit kernel place the documents inside a group; TK_DECODE_GROUPS=0 (and a group whose text reaches 2 GiB) takes the
    per-document length pass instead.  Both forms give the reference's text and offsets: empty documents at the beginning, in the
    middle and at the end of a group, groups of nothing but empty documents, a last group that is not full, special tokens under
    every policy, documents much longer than a step."""
    import random
    rng = random.Random(5)
    orc = helpers.oracle_for(test_vocab)
    words = [b"hello", b" world", b"\n", b" caf\xc3\xa9", b" \xf0\x9f\x9a\x80", b"12", b" x"]
    def doc():
        k = rng.choice([0, 0, 1, 3, 40, 400, 3000])
        return b"".join(rng.choice(words) for _ in range(k))
    shapes = [[doc() for _ in range(n)] for n in (1, 15, 16, 17, 33, 100)]
    shapes.append([b""] * 40)                                            # nothing but empty documents
    shapes.append([b"a"] + [b""] * 31 + [b"b"] + [b""] * 15)             # whole groups of empty documents between two texts
    shapes.append([b""] * 16 + [doc() for _ in range(16)] + [b""] * 5)
    P = tk.SpecialTokenPolicy
    engines = []
    for groups, limit in (("1", ""), ("0", ""), ("1", "2000")):           # (the third: groups, but a group of 2000 bytes already falls back)
        monkeypatch.setenv("TK_DECODE_GROUPS", groups)
        if limit:
            monkeypatch.setenv("TK_DECODE_GROUP_LIMIT", limit)
        e = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
        e.set_special_tokens(SPECIALS)
        engines.append(e)
    try:
        for docs in shapes:
            for bos, eos in ((True, True), (False, False)):
                id_lists = [orc.encode(d, bos, eos) for d in docs]
                # a few special ids in the middle of some documents
                for ids in id_lists[::3]:
                    if len(ids) > 4:
                        at = len(ids) // 2
                        ids.insert(at, 3)
                        try:
                            ref(test_vocab, ids, P.Ignore)
                        except ValueError:                               # (it cut a UTF-8 sequence into two runs: not this test's subject)
                            del ids[at]
                for policy in (P.Ignore, P.Keep):
                    want = [ref(test_vocab, ids, policy) for ids in id_lists]
                    for e in engines:
                        assert e.decode_docs(id_lists, policy) == want
    finally:
        for e in engines:
            e.close()
