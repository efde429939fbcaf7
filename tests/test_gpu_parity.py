"""Parity tests proper: the HIP path, called through the C ABI (include/tekken_hip.h), against the
oracle on the same seeded inputs -- bit-exact (integer work, no tolerance).

Small sizes compare every id with the oracle; the full BASELINE size (1 M x 512 B) is checked
through size-independent properties (bytes of the emitted tokens concatenate back to the input,
determinism, checksum of a sample vs the oracle).  Nothing here reads /root/reference.
"""
import json
import threading

import numpy as np
import pytest

import corpus
import helpers
import tk_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_small(tk, test_vocab):
    e = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_bench(tk, bench_vocab):
    e = tk.Engine(bench_vocab["tokens"], bench_vocab["num_special"], bench_vocab["bos"], bench_vocab["eos"], device=0)
    yield e
    e.close()


def check_batch(eng, orc, data, offs, bos=True, eos=True):
    ids, oo = eng.encode_batch(data, offs, bos, eos)
    eids, eoo = orc.encode_batch(data, offs, bos, eos, threads=8)
    if not (np.array_equal(oo, eoo) and np.array_equal(ids, eids)):
        for d in range(len(offs) - 1):
            a = ids[int(oo[d]):int(oo[d + 1])]
            b = eids[int(eoo[d]):int(eoo[d + 1])]
            assert a.tolist() == b.tolist(), (d, bytes(data[int(offs[d]):int(offs[d + 1])])[:120])
        raise AssertionError("offset arrays differ")
    return ids, oo


def test_library_loaded_is_the_hip_one(tk, eng_small):
    import ctypes
    assert tk.lib()._name.endswith("libtekken_hip.so")
    assert isinstance(eng_small._h, ctypes.c_void_p) and eng_small._h.value


def test_small_vocab_known_answer_via_tokenizer(tk, golden, small_vocab):
    """Tekkenizer::from_file -> encode -> decode on the reference's small-vocab construction."""
    from test_host_tokenizer import model
    t = tk.Tekkenizer.from_json(json.dumps(model(small_vocab["tokens"])), device=0)
    for text, bos, eos, ids in golden["ref"]["small_vocab"]["cases"]:
        assert t.encode(text, bos, eos) == ids
    ids = t.encode("hello world", True, True)
    assert t.decode(ids, tk.SpecialTokenPolicy.Ignore) == "hello world"
    assert t.encode_batch(["hello world", "", "world"], False, False) == [[266, 42, 129, 121, 124, 118, 110], [], [267]]
    t.close()


def test_reference_vectors_on_consistent_vocab(tk, golden):
    import ref_consistent_vocab as rcv
    toks = rcv.build(golden["ref"])
    e = tk.Engine(toks, 1000, 1, 2, device=0)
    texts = [t for t, _ in golden["ref"]["encode"]]
    got = e.encode_docs([t.encode("utf-8") for t in texts], False, False)
    for (t, ids), g in zip(golden["ref"]["encode"], got):
        assert g == ids, t
    e.close()


def test_split_matches_golden(eng_small, golden):
    cases = golden["split"]["cases"]
    got = eng_small.split_docs([c["text"].encode("utf-8") for c in cases])
    for c, s in zip(cases, got):
        assert s == c["starts"], c["text"]


@pytest.mark.parametrize("bos,eos", [(True, True), (False, False), (True, False), (False, True)])
def test_mixed_docs_small_vocab(eng_small, test_vocab, bos, eos):
    orc = helpers.oracle_for(test_vocab)
    docs = helpers.mixed_docs(200, 60, 300, max_len=40000) + helpers.random_unicode_docs(500)
    data = np.frombuffer(b"".join(docs), dtype=np.uint8)
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    check_batch(eng_small, orc, data, offs, bos, eos)


@pytest.mark.parametrize("kind,n,dl,seed", [("ascii", 20000, 512, 1), ("mixed", 4000, 2048, 2), ("zipf", 6000, 0, 4),
                                              ("ascii", 1000, 64, 0)])
def test_baseline_shapes_bench_vocab(eng_bench, bench_vocab, kind, n, dl, seed):
    """Reduced-size versions of the BASELINE.json configs (C1..C5 shapes), every id compared."""
    orc = helpers.oracle_for(bench_vocab)
    data, offs = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + seed)
    check_batch(eng_bench, orc, data, offs)
    if kind == "zipf":
        assert eng_bench.last_stats()["long_docs"] > 0  # the long-piece second pass ran


def test_both_pipelines(tk, test_vocab, monkeypatch):
    """The flat chunk-per-wave pipeline (default) and the per-document pipeline (TK_PIPELINE=doc) give the same ids;
    the flat one hands documents with very long runs / pieces back and says how many."""
    orc = helpers.oracle_for(test_vocab)
    # (a white-space run across a region end and a long digit run are always handed back; long letter runs are cut: step 4b)
    docs = helpers.mixed_docs(120, 40, 200, max_len=40000) + helpers.random_unicode_docs(300) + [b"a b" + b" " * 3000 + b"x", b"1" * 2500]
    exp = [orc.encode(d, True, True) for d in docs]
    monkeypatch.setenv("TK_PIPELINE", "flat")
    flat = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    assert flat.encode_docs(docs, True, True) == exp
    st = flat.last_stats()
    assert 0 < st["handed_back"] < len(docs)
    flat.close()
    monkeypatch.setenv("TK_PIPELINE", "doc")
    per_doc = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    assert per_doc.encode_docs(docs, True, True) == exp
    assert per_doc.last_stats()["handed_back"] == 0
    per_doc.close()
    # default = flat: multi-byte text stays on the fast path, nothing in the mixed UTF-8 shape is handed back
    monkeypatch.delenv("TK_PIPELINE")
    auto = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    d, o = corpus.generate("mixed", 200, 2048, seed=corpus.BASE_SEED + 2)
    mixed = corpus.docs_of(d, o)
    assert auto.encode_docs(mixed, True, True) == [orc.encode(x, True, True) for x in mixed]
    assert auto.last_stats()["handed_back"] == 0
    auto.close()


def test_flat_path_stress(eng_small, test_vocab):
    """What the flat path is sensitive to: documents of every length packed back to back (boundaries at every lane /
    chunk offset), many empty documents, pieces that miss the vocabulary in bulk, long runs across chunk boundaries."""
    import random
    orc = helpers.oracle_for(test_vocab)
    rng = random.Random(11)
    docs = [b"x" * n for n in range(0, 130)] + [b"ab " * n for n in range(0, 400, 7)] + [b""] * 50
    words = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(1, 30))) for _ in range(5000)]
    docs += [" ".join(rng.choice(words) for _ in range(rng.randint(1, 400))).encode() for _ in range(300)]
    alpha = ["a", "b", "1", "2", "'", "s", "t", "!", "-", " ", " ", " ", "\n", "\r", "\t"]
    for _ in range(400):
        n = rng.randint(10, 3000)
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(rng.choice(alpha) * rng.choice([1, 1, 1, 2, 3, 5, 9, 17, 40, 70]))
        docs.append("".join(parts).encode())
    docs += ["".join(rng.choice("abc123 ,.\n") for _ in range(rng.randint(0, 50))).encode() for _ in range(3000)]
    # multi-byte code points: chars across lane / region boundaries, multi-byte digit and white-space runs, long s
    ualpha = ["a", "S", "1", "\u0663", "\uff13", "'", "\u017f", "s", "!", " ", " ", "\n", "\r", "\u4e2d", "\u00e9", "\U0001f680",
              "\u00a0", "\u3000", "-", "\t"]
    docs += ["".join(rng.choice(ualpha) * rng.choice([1, 1, 1, 2, 3, 7, 20]) for _ in range(rng.randint(0, 60))).encode()
             for _ in range(1500)]
    docs += helpers.random_unicode_docs(1500, seed=12, max_len=300)
    # runs of every class that start / end around the region geometry (halos, commit size, piece limit), every phase
    for pad in list(range(860, 1000, 3)) + list(range(1880, 2030, 5)):
        for ch in ("1", "\n", " ", "a", "!", "\u4e2d", "\uff11"):
            for rl in (31, 32, 33, 63, 64, 65):
                docs.append(("x y " * (pad // 4) + "q" * (pad % 4) + ch * rl + " z").encode())
    rng.shuffle(docs)
    data = np.frombuffer(b"".join(docs), dtype=np.uint8)
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    for bos, eos in ((True, True), (False, False)):
        check_batch(eng_small, orc, data, offs, bos, eos)


def test_edge_batches(eng_small, test_vocab):
    orc = helpers.oracle_for(test_vocab)
    # empty batch, only-empty docs, one empty doc between others
    ids, oo = eng_small.encode_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(ids) == 0 and oo.tolist() == [0]
    assert eng_small.encode_docs([b"", b"", b""], True, True) == [[1, 2]] * 3
    assert eng_small.encode_docs([b"", b""], False, False) == [[], []]
    assert eng_small.encode_docs([b"a", b"", b"b"], True, False) == [orc.encode(b"a", True, False), [1],
                                                                    orc.encode(b"b", True, False)]
    # maximum sizes of the configs: a 32 KiB single letter run, a 32 KiB white-space run, a 300 KB document
    rng = np.random.default_rng(5)
    big = [bytes(rng.integers(97, 123, 32768, dtype=np.uint8)), b" " * 20000 + b"\n" + b" " * 12767,
           b"the quick brown fox " * 15000, ("中文" * 8000).encode(), b"x" + b"\n" * 5000]
    for doc in big:
        assert eng_small.encode_docs([doc], True, True)[0] == orc.encode(doc, True, True), doc[:40]


def test_utf8_validation(tk, eng_small):
    good = ["ok", "é中🚀"]
    eng_small.encode_docs([g.encode() for g in good], validate_utf8=True)
    for bad in (b"\xff", b"a\xc3", b"\xe2\x82", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"ab\x80cd"):
        with pytest.raises(tk.TokenizerError) as e:
            eng_small.encode_docs([b"fine", bad], validate_utf8=True)
        assert e.value.code == tk.TK_ERR_INVALID_UTF8, bad


def test_invalid_table_rejected(tk, small_vocab):
    toks = list(small_vocab["tokens"])
    with pytest.raises(tk.TokenizerError) as e:
        tk.Engine(toks[:200], 10, 1, 2)
    assert e.value.kind == "InvalidConfig"
    toks[257] = b"hello"
    with pytest.raises(tk.TokenizerError) as e:
        tk.Engine(toks, 10, 1, 2)
    assert e.value.kind == "InvalidConfig"


def test_concurrent_calls_on_one_context(eng_small, test_vocab):
    """The reference's Tekkenizer is Sync (tests/test_tokenizer_output.rs:5-12): one context, many threads."""
    orc = helpers.oracle_for(test_vocab)
    docs = helpers.mixed_docs(50, 10, 30)
    exp = [orc.encode(d, True, True) for d in docs]
    errs = []

    def run():
        try:
            for _ in range(5):
                assert eng_small.encode_docs(docs, True, True) == exp
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=run) for _ in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:1]


def test_device_resident_entry_and_full_size_properties(tk, eng_bench, bench_vocab):
    """BASELINE config C2 at full size (1 M x 512 B) with inputs resident in HBM."""
    import torch
    n_docs = 1_000_000
    data, offs = corpus.generate("ascii", n_docs, 512, seed=corpus.BASE_SEED + 1)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    sums = []
    for _ in range(2):
        v_ids, v_oo = eng_bench.encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, len(data), True,
                                                          True, stream)
        ids = torch.as_tensor(v_ids, device="cuda")
        oo = torch.as_tensor(v_oo, device="cuda")
        ids_h = ids.cpu().numpy().view(np.uint32)
        oo_h = oo.cpu().numpy().astype(np.uint64)
        sums.append(tk_oracle.fnv1a(ids_h))
    assert sums[0] == sums[1]  # deterministic
    assert int(oo_h[-1]) == len(ids_h) and np.all(np.diff(oo_h.astype(np.int64)) >= 2)
    ns = bench_vocab["num_special"]
    # size-independent property: the bytes of the emitted tokens concatenate back to the input
    tok_len = np.array([len(t) for t in bench_vocab["tokens"]], dtype=np.int64)
    body = ids_h[ids_h >= ns]
    assert int(tok_len[body - ns].sum()) == len(data)
    assert len(ids_h) - len(body) == 2 * n_docs  # exactly one BOS and one EOS per document
    first = oo_h[:-1].astype(np.int64)
    last = oo_h[1:].astype(np.int64) - 1
    assert np.all(ids_h[first] == bench_vocab["bos"]) and np.all(ids_h[last] == bench_vocab["eos"])
    # a sample of documents against the oracle, id for id (checksum of checksums)
    orc = helpers.oracle_for(bench_vocab)
    for lo in (0, 500_000, 990_000):
        hi = lo + 5000
        sub_offs = offs[lo:hi + 1] - offs[lo]
        eids, _ = orc.encode_batch(data[int(offs[lo]):int(offs[hi])], sub_offs, True, True, threads=8)
        got = ids_h[int(oo_h[lo]):int(oo_h[hi])]
        assert tk_oracle.fnv1a(got) == tk_oracle.fnv1a(eids)
    tm = eng_bench.last_timing()
    assert tm["encode_kernel_ms"] > 0 and tm["pipeline_ms"] >= tm["encode_kernel_ms"]


def test_full_size_mixed_utf8_properties(tk, eng_bench, bench_vocab):
    """BASELINE config C3 at FULL size (1 M x 2 KiB mixed UTF-8, 2.05 GB) with inputs resident in HBM: determinism, the
    bytes of the emitted tokens concatenate back to the input, one BOS / EOS per document, nothing handed back, three
    samples of 3 000 documents id for id against the oracle, and the GPU batch decode gives the text back."""
    import torch
    n_docs = 1_000_000
    data, offs = corpus.generate("mixed", n_docs, 2048, seed=corpus.BASE_SEED + 1)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    sums = []
    for _ in range(2):
        v_ids, v_oo = eng_bench.encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, len(data), True, True, stream)
        ids = torch.as_tensor(v_ids, device="cuda")
        oo = torch.as_tensor(v_oo, device="cuda")
        ids_h = ids.cpu().numpy().view(np.uint32)
        oo_h = oo.cpu().numpy().astype(np.uint64)
        sums.append(tk_oracle.fnv1a(ids_h))
    assert sums[0] == sums[1]
    assert eng_bench.last_stats()["handed_back"] == 0
    # round trip on the device while the ids are still there
    ids_keep, oo_keep = ids.clone(), oo.clone()
    v_b, _ = eng_bench.decode_batch_device(ids_keep.data_ptr(), oo_keep.data_ptr(), n_docs, ids_keep.numel(), tk.SpecialTokenPolicy.Ignore, stream)
    back = torch.as_tensor(v_b, device="cuda")
    assert back.numel() == len(data) and bool(torch.equal(back, d_bytes))
    del back, ids_keep, oo_keep
    ns = bench_vocab["num_special"]
    assert int(oo_h[-1]) == len(ids_h) and np.all(np.diff(oo_h.astype(np.int64)) >= 2)
    tok_len = np.array([len(t) for t in bench_vocab["tokens"]], dtype=np.int64)
    body = ids_h[ids_h >= ns]
    assert int(tok_len[body - ns].sum()) == len(data)
    assert len(ids_h) - len(body) == 2 * n_docs
    first = oo_h[:-1].astype(np.int64)
    last = oo_h[1:].astype(np.int64) - 1
    assert np.all(ids_h[first] == bench_vocab["bos"]) and np.all(ids_h[last] == bench_vocab["eos"])
    orc = helpers.oracle_for(bench_vocab)
    for lo in (0, 480_000, 997_000):
        hi = lo + 3000
        sub_offs = offs[lo:hi + 1] - offs[lo]
        eids, _ = orc.encode_batch(data[int(offs[lo]):int(offs[hi])], sub_offs, True, True, threads=8)
        got = ids_h[int(oo_h[lo]):int(oo_h[hi])]
        assert tk_oracle.fnv1a(got) == tk_oracle.fnv1a(eids) and len(got) == len(eids)


def test_zipf_shape_properties(tk, eng_bench, bench_vocab):
    """The shape of BASELINE configs[4] at one GPU's share (500 k documents of 16 B .. 32 KiB, 1.14 GB, resident in HBM): every
    route at once -- flat path, long-piece records, handed-back documents, the round-based workgroup merges.  Determinism,
    the GPU batch decode gives the text back byte for byte, one BOS / EOS per document, and three samples of 20 000
    documents id for id against the oracle."""
    import torch
    n_docs = 500_000
    data, offs = corpus.generate("zipf", n_docs, 0, seed=corpus.BASE_SEED + 1)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    sums = []
    for _ in range(2):
        v_ids, v_oo = eng_bench.encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, len(data), True, True, stream)
        ids = torch.as_tensor(v_ids, device="cuda")
        oo = torch.as_tensor(v_oo, device="cuda")
        ids_h = ids.cpu().numpy().view(np.uint32)
        oo_h = oo.cpu().numpy().astype(np.uint64)
        sums.append(tk_oracle.fnv1a(ids_h))
    assert sums[0] == sums[1]
    st = eng_bench.last_stats()
    assert 0 < st["handed_back"] < 2000 and eng_bench.long_piece_records() > 0 and eng_bench.round_path_docs() > 0
    ids_keep, oo_keep = ids.clone(), oo.clone()
    v_b, _ = eng_bench.decode_batch_device(ids_keep.data_ptr(), oo_keep.data_ptr(), n_docs, ids_keep.numel(), tk.SpecialTokenPolicy.Ignore, stream)
    back = torch.as_tensor(v_b, device="cuda")
    assert back.numel() == len(data) and bool(torch.equal(back, d_bytes))
    del back, ids_keep, oo_keep
    assert int(oo_h[-1]) == len(ids_h) and np.all(np.diff(oo_h.astype(np.int64)) >= 2)
    first = oo_h[:-1].astype(np.int64)
    last = oo_h[1:].astype(np.int64) - 1
    assert np.all(ids_h[first] == bench_vocab["bos"]) and np.all(ids_h[last] == bench_vocab["eos"])
    orc = helpers.oracle_for(bench_vocab)
    for lo in (0, 240_000, 480_000):
        hi = lo + 20_000
        sub_offs = offs[lo:hi + 1] - offs[lo]
        eids, _ = orc.encode_batch(data[int(offs[lo]):int(offs[hi])], sub_offs, True, True, threads=16)
        got = ids_h[int(oo_h[lo]):int(oo_h[hi])]
        assert tk_oracle.fnv1a(got) == tk_oracle.fnv1a(eids) and len(got) == len(eids)


def test_invalid_utf8_without_validation_is_safe(eng_small):
    """Callers that skip validation and pass malformed bytes get unspecified ids but no crash, no hang,
    and the id count stays within the documented bound (bytes + 2 per document)."""
    rng = np.random.default_rng(9)
    docs = [bytes(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8)) for _ in range(200)]
    docs += [b"\x80" * 500, b"\xf0" * 200, b"\xe4\xb8" * 100, b"a\xc3", b"\xff\xfe\xfd"]
    out = eng_small.encode_docs(docs, True, True)
    for d, ids in zip(docs, out):
        assert 2 <= len(ids) <= len(d) + 2 and ids[0] == 1 and ids[-1] == 2


def test_bench_contract_small(tk):
    """bench.py prints exactly one JSON line with the contract's keys (small workload)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--docs", "20000", "--steps", "2", "--warmup", "1",
                        "--cpu-sample-docs", "5000", "--cpu-passes", "1", "--decode-steps", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"] == "u8" and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1 and "workload" in j["config"]
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1 and j["bit_exact_vs_cpu"] is True
    assert j["decode"]["round_trip_exact"] is True
    # two contexts fed by two host threads (the leg that shows what a second batch in flight buys): both give the same ids
    assert j["two_in_flight"]["both_contexts_same_ids"] is True and j["two_in_flight"]["calls_timed"] == 20


def test_table_cache_context(tk, bench_vocab, tmp_path, monkeypatch):
    """Row f-2: a context created from the table cache (TK_TABLE_CACHE_DIR) encodes exactly like one that built its
    tables; a damaged cache file is ignored and rewritten."""
    import glob
    import time
    orc = helpers.oracle_for(bench_vocab)
    data, offs = corpus.generate("mixed", 500, 1024, seed=corpus.BASE_SEED + 9)
    args = (bench_vocab["tokens"], bench_vocab["num_special"], bench_vocab["bos"], bench_vocab["eos"])
    monkeypatch.setenv("TK_TABLE_CACHE_DIR", str(tmp_path))
    t0 = time.perf_counter()
    e1 = tk.Engine(*args, device=0)            # builds, writes the file
    t1 = time.perf_counter()
    files = glob.glob(str(tmp_path / "tk_tables_*.bin"))
    assert len(files) == 1
    e2 = tk.Engine(*args, device=0)            # loads
    t2 = time.perf_counter()
    check_batch(e1, orc, data, offs)
    check_batch(e2, orc, data, offs)
    e1.close()
    e2.close()
    print("context create: build+save %.3f s, from cache %.3f s" % (t1 - t0, t2 - t1))
    raw = open(files[0], "rb").read()
    open(files[0], "wb").write(raw[:len(raw) // 3])
    e3 = tk.Engine(*args, device=0)            # refuses the truncated file, builds, rewrites
    check_batch(e3, orc, data, offs)
    e3.close()
    assert open(files[0], "rb").read() == raw


def test_pipelined_ingestion(tk, eng_small, test_vocab):
    """Row f-4: the streaming entry (slices of whole documents, copy up / kernels / copy down overlapped, caller-owned
    pinned buffers) returns exactly what tk_encode_batch returns -- for slices of one document, slices larger than the
    batch, documents longer than a slice, empty documents at slice edges, and every BOS / EOS combination."""
    import random
    rng = random.Random(11)
    docs = helpers.mixed_docs(300, 50, 400, max_len=30000) + [b""] * 5 + helpers.random_unicode_docs(200) + [b"", b"x", b""]
    rng.shuffle(docs)
    data, offs = tk.pack_docs(docs)
    pin_data = tk.host_empty(len(data), np.uint8)
    pin_data[:] = data
    pin_offs = tk.host_empty(len(offs), np.uint64)
    pin_offs[:] = offs
    pin_ids = tk.host_empty(len(data) + 2 * len(docs) + 1, np.uint32)
    pin_oo = tk.host_empty(len(offs), np.uint64)
    for bos, eos in ((True, True), (False, False), (True, False)):
        exp_ids, exp_oo = eng_small.encode_batch(data, offs, bos, eos)
        for sl in (1, 700, 4096, 100000, 0):
            ids, oo = eng_small.encode_batch_pipelined(pin_data, pin_offs, bos, eos, slice_bytes=sl, ids_out=pin_ids, offsets_out=pin_oo)
            assert np.array_equal(oo, exp_oo), (bos, eos, sl)
            assert np.array_equal(ids, exp_ids), (bos, eos, sl)
        ids, oo = eng_small.encode_batch_pipelined(data, offs, bos, eos, slice_bytes=5000)      # pageable buffers
        assert np.array_equal(oo, exp_oo) and np.array_equal(ids, exp_ids)
    # no documents / only empty documents
    ids, oo = eng_small.encode_batch_pipelined(np.zeros(0, np.uint8), np.zeros(1, np.uint64), True, True)
    assert len(ids) == 0 and oo.tolist() == [0]
    ids, oo = eng_small.encode_batch_pipelined(np.zeros(0, np.uint8), np.zeros(4, np.uint64), True, True)
    assert ids.tolist() == [test_vocab["bos"], test_vocab["eos"]] * 3 and oo.tolist() == [0, 2, 4, 6]
    # an output buffer that is too small is refused, not overrun
    small = np.empty(10, np.uint32)
    with pytest.raises(tk.TokenizerError):
        eng_small.encode_batch_pipelined(data, offs, True, True, slice_bytes=4096, ids_out=small)


def test_key_hash_fallback_kernel(tk, test_vocab):
    """The kernel instantiation for tables built with the strong key hash (mode 1; tests/test_flat_path.py builds the
    same three colliding 12-byte tokens on the emulator): lookups stay exact for those tokens and for everything else."""
    import struct

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF
    x, k1 = 0x6C6C6568, 0x6F77206F
    extra = [struct.pack("<III", x ^ rotl(k2, 13), k1, k2) for k2 in (0x61616161, 0x62626262, 0x63636363)]
    toks = list(test_vocab["tokens"]) + [t for t in extra if t not in test_vocab["tokens"]]
    v = dict(test_vocab, tokens=toks)
    orc = helpers.oracle_for(v)
    e = tk.Engine(toks, v["num_special"], v["bos"], v["eos"], device=0)
    docs = [b"hello world aaaa", extra[0] + b" " + extra[1], extra[2], b"x" + extra[0][1:]] + helpers.mixed_docs(150, 40, 300, max_len=20000) \
        + helpers.random_unicode_docs(100)
    assert e.encode_docs(docs, True, True) == [orc.encode(d, True, True) for d in docs]
    e.close()


def test_dense_piece_regions(tk, eng_small, test_vocab):
    """Regions with more pieces than the LDS piece list holds (under two bytes per piece over 2 KB) take two enumeration
    passes; nothing is handed back because of it."""
    import random
    rng = random.Random(77)
    docs = [(b"1,2,3,4,5,6,7,8,9,0;" * 300), b"a b c d e f g h i j " * 250, b"!a!b!c" * 700,
            b"".join(bytes([rng.choice(b"0123456789,.;:-+ ")]) for _ in range(9000)),
            b"x" * 40 + b"1,2," * 600 + b" zzzzzz " + b"3;4;" * 500]
    docs += [b"7," * rng.randint(1, 40) for _ in range(2000)]
    docs += [(b"q" * 70 + b",1" * 900)]
    # piece counts of a region sweeping across the capacity of the LDS list (1 054): a dense stretch of a bytes, then sparse text
    docs += [b"1," * a + b"abcdefgh " * 300 for a in range(480, 560)]
    orc = helpers.oracle_for(test_vocab)
    data, offs = tk.pack_docs(docs)
    check_batch(eng_small, orc, data, offs)
    assert eng_small.last_stats()["handed_back"] <= 2


def test_ids18_wire_format(tk, eng_small):
    """The 18-bit wire format of the multi-GPU gather (tk_pack_ids18_device / tk_unpack_ids18_device): byte for byte the
    layout the header describes (checked against the numpy restatement the gloo tests use), exact round trip for every
    tail length, an id of 2^18 refused."""
    import torch
    from test_parallel_gloo import NumpyIds18Codec
    ref = NumpyIds18Codec()
    rng = np.random.default_rng(3)
    for n in (0, 1, 15, 16, 17, 31, 33, 1000, 65537, 1 << 20):
        ids = rng.integers(0, 1 << 18, n, dtype=np.int64).astype(np.uint32)
        if n > 3:
            ids[:3] = [0, (1 << 18) - 1, 1 << 16]
        d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
        nb = tk.ids18_bytes(n)
        assert nb == ref.nbytes(n)
        d_packed = torch.zeros((nb + 3) // 4 + 1, dtype=torch.int32, device="cuda")
        eng_small.pack_ids18_device(d_ids.data_ptr(), n, d_packed.data_ptr(), torch.cuda.current_stream().cuda_stream)
        exp = ref.pack(torch.from_numpy(ids.view(np.int32)))
        assert np.array_equal(d_packed.cpu().numpy()[:exp.numel()], exp.numpy())
        d_out = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        eng_small.unpack_ids18_device(d_packed.data_ptr(), n, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = d_out.cpu().numpy()
        assert np.array_equal(out[:n].view(np.uint32), ids) and out[n] == -1
        # id buffers that are only word-aligned (a rank's ids land at an arbitrary id index of the gathered buffer; per-rank
        # id counts are not multiples of 4): pack from / unpack to every offset 1..3
        for sh in (1, 2, 3):
            if n == 0:
                continue
            src = torch.zeros(n + sh, dtype=torch.int32, device="cuda")
            src[sh:] = d_ids
            d_p2 = torch.zeros_like(d_packed)
            eng_small.pack_ids18_device(src.data_ptr() + 4 * sh, n, d_p2.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert torch.equal(d_p2, d_packed)
            dst = torch.full((n + sh + 1,), -1, dtype=torch.int32, device="cuda")
            eng_small.unpack_ids18_device(d_packed.data_ptr(), n, dst.data_ptr() + 4 * sh, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            o2 = dst.cpu().numpy()
            assert np.array_equal(o2[sh:sh + n].view(np.uint32), ids) and o2[sh + n] == -1 and (o2[:sh] == -1).all()
    bad = torch.tensor([5, 1 << 18, 7], dtype=torch.int32, device="cuda")
    buf = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(tk.TokenizerError):
        eng_small.pack_ids18_device(bad.data_ptr(), 3, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)


def test_bench_distributed_path_single_rank():
    """bench.py's N > 1 code path (RCCL process group, size exchange, side-stream gather, deferred results) with a
    world of one rank: everything but the peer-to-peer transfers themselves."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TK_BENCH_FORCE_DIST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29577")
    for gather in ("overlap", "sync"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--docs", "20000", "--steps", "3", "--warmup", "1",
                            "--gather", gather], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["ids_total"] > 0


def test_json_pattern_opt_in(tk, test_vocab, bench_vocab):
    """Row f-3 (opt-in): tk_ctx_set_pattern(ctx, 1) makes the split follow the `pattern` of Mistral's tekken.json
    (case-aware words, single digits, '/' after punctuation) -- id for id the oracle in the same mode; mode 0 afterwards
    is the reference's behaviour again; the tokenizer level honours only the known pattern string."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "split_vectors_tekken.json")) as f:
        g = json.load(f)
    docs = [c["text"].encode("utf-8") for c in g["cases"]] + helpers.mixed_docs(60, 20, 80, max_len=9000) + helpers.random_unicode_docs(300) \
        + [b"", b"HelloWorld XMLHttpRequest iPhone 1234 a/b/c\n"]
    orc = tk_oracle.Oracle(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"])
    orc.set_pattern(1)
    plain = helpers.oracle_for(test_vocab)
    e = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    e.set_pattern(1)
    assert e.encode_docs(docs, True, True) == [orc.encode(d, True, True) for d in docs]
    e.set_pattern(0)
    assert e.encode_docs(docs[:200], True, False) == [plain.encode(d, True, False) for d in docs[:200]]
    with pytest.raises(tk.TokenizerError):
        e.set_pattern(7)
    e.close()
    # tokenizer level: the model file's pattern string decides
    from test_host_tokenizer import model
    import synth_vocab as sv
    m = model(test_vocab["tokens"], num_special=test_vocab["num_special"], specials=("<unk>", "<s>", "</s>"))
    m["config"]["pattern"] = sv.MISTRAL_PATTERN
    t = tk.Tekkenizer.from_json(json.dumps(m), device=0)
    assert t.encode("HelloWorld 12", False, False) == plain.encode(b"HelloWorld 12", False, False)      # ignored by default
    t.set_honour_pattern(True)
    assert t.encode("HelloWorld 12", False, False) == orc.encode(b"HelloWorld 12", False, False)
    t.set_honour_pattern(False)
    assert t.encode("HelloWorld 12", False, False) == plain.encode(b"HelloWorld 12", False, False)
    t.close()
    # ... also when the object comes from a TK_TABLE_CACHE_DIR side file (the side file keeps config.pattern)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.json")
        with open(path, "w") as f:
            json.dump(m, f)
        old = os.environ.get("TK_TABLE_CACHE_DIR")
        os.environ["TK_TABLE_CACHE_DIR"] = td
        try:
            t1 = tk.Tekkenizer.from_file(path, device=0)
            t2 = tk.Tekkenizer.from_file(path, device=0)
        finally:
            if old is None:
                del os.environ["TK_TABLE_CACHE_DIR"]
            else:
                os.environ["TK_TABLE_CACHE_DIR"] = old
        assert not t1.from_cache() and t2.from_cache() and t2.json_pattern() == sv.MISTRAL_PATTERN
        t2.set_honour_pattern(True)
        assert t2.encode("HelloWorld 12", False, False) == orc.encode(b"HelloWorld 12", False, False)
        t1.close()
        t2.close()
    m["config"]["pattern"] = "something else"
    t = tk.Tekkenizer.from_json(json.dumps(m), device=0)
    with pytest.raises(tk.TokenizerError) as ei:
        t.set_honour_pattern(True)
    assert ei.value.kind == "InvalidConfig"
    t.close()


@pytest.mark.gpu
def test_merge_kernels_every_piece_length(tk, eng_small, eng_bench, test_vocab, bench_vocab):
    """The merge kernels (csrc/tk_flat_impl.h tk_merge_lds<8|16|32>, the one-lane-per-byte loop for 33..64 bytes, pass 2
    beyond): words of every length 2..80 that miss the vocabulary -- random letters, runs of one letter, UTF-8 words --
    many per document, so that a wave's 64 queued pieces span several documents (the hole counts of a run of lanes
    are combined into one atomic) and the 16-byte id stores meet every alignment and every tail length."""
    import random
    rng = random.Random(4242)
    letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    docs = []
    for rep in range(6):
        words = []
        for n in range(2, 81):
            words.append("".join(rng.choice(letters) for _ in range(n)))
            words.append(rng.choice(letters) * n)
            words.append("".join(rng.choice("éü中文Ж") for _ in range(max(1, n // 3))))
        rng.shuffle(words)
        # short documents (a few words each: document boundaries inside one wave's items) and one long one
        for k in range(0, len(words), 5):
            docs.append((" ".join(words[k:k + 5])).encode())
        docs.append((" ".join(words)).encode())
    docs += [b"", b"zq", b"Zq" * 33, ("中" * 11).encode()]
    data, offs = tk.pack_docs(docs)
    for eng, v in ((eng_small, test_vocab), (eng_bench, bench_vocab)):
        orc = helpers.oracle_for(v)
        for bos, eos in ((True, True), (False, False)):
            check_batch(eng, orc, data, offs, bos, eos)


def test_sparse_miss_queues(tk, eng_bench, bench_vocab):
    """Few queued pieces in many chunks: the 64 items of a merge wave then span thousands of per-chunk sub-queues (the wave
    finds them by a window over the prefix sums, then by bisection: csrc/tk_flat_impl.h tk_merge_wave); every length class,
    also with one single queued piece in the whole batch."""
    import random
    rng = random.Random(99)
    letters = "abcdefghijklmnopqrstuvwxyz"
    bdocs = corpus.docs_of(*corpus.generate("ascii", 20000, 512, seed=corpus.BASE_SEED + 31))
    orc = helpers.oracle_for(bench_vocab)
    for every, lens in ((331, (5, 12, 24, 48)), (2917, (24,)), (19999, (40,)), (700, (20, 28, 60))):
        docs = list(bdocs)
        for d in range(every - 1, len(docs), every):
            n = lens[(d // every) % len(lens)]
            docs[d] = docs[d][:200] + b" " + "".join(rng.choice(letters) for _ in range(n)).encode() + b" " + docs[d][200:]
        data, offs = tk.pack_docs(docs)
        check_batch(eng_bench, orc, data, offs)
    # one-byte pieces never miss: the narrow queues are as sparse as the wide ones
    filler = "".join(c + "\n" for c in letters).encode() * 40
    docs = [filler] * 3000
    for d in range(0, 3000, 409):
        n = (3, 7, 12, 16, 20, 31, 40, 64)[(d // 409) % 8]
        docs[d] = filler[:520] + "".join(rng.choice(letters) for _ in range(n)).encode() + b"\n" + filler[520:]
    data, offs = tk.pack_docs(docs)
    check_batch(eng_bench, orc, data, offs)


def test_long_pieces_stay_on_the_flat_path(tk, eng_small, eng_bench, test_vocab, bench_vocab, monkeypatch):
    """Pieces of 65..256 bytes (csrc/tk_flat_impl.h step 6, tk_flat_long_kernel): their documents are not handed back; beyond 256
    bytes they are unless the piece can be cut (step 4b; TK_FLAT_CUT=0 shows the hand-back).  Letter runs, CJK paragraphs, rulers, the first byte of the piece walking over a chunk boundary, many
    such pieces per document; the same batch with the path switched off (TK_FLAT_LONG=0) gives the same ids."""
    import random
    rng = random.Random(5)
    letters = "abcdefghijklmnopqrstuvwxyz"
    filler = b"ab cd ef gh ij kl mn op qr st uv wx yz " * 60
    kept, handed = [], []
    for n in (64, 65, 100, 128, 200, 254, 255):
        w = "".join(rng.choice(letters) for _ in range(n)).encode()
        kept.append(b"x " + w + b" y")
        kept.append(filler[:1899] + w + b" " + filler[:300])
    for k in range(1880, 2030, 3):
        w = "".join(rng.choice(letters) for _ in range(150)).encode()
        kept.append(filler[:k] + b" " + w + b" " + filler[:200])
    ideo = [chr(0x4E00 + rng.randrange(0x5000)) for _ in range(500)]
    for _ in range(300):
        parts = []
        while sum(len(x) for x in parts) < 700:
            parts.append("".join(rng.choice(ideo) for _ in range(rng.randint(5, 80))) + rng.choice("，。"))
        kept.append("".join(parts).encode())
    kept.append(b"head " + b"=" * 200 + b"\n\ntail")
    kept.append((" " + "é" * 100 + " x").encode())
    for n in (256, 300, 1000):
        w = "".join(rng.choice(letters) for _ in range(n)).encode()
        handed.append(b"x " + w + b" y")
        handed.append(filler[:1899] + w + b" z")
    docs = kept + handed
    data, offs = tk.pack_docs(docs)
    for eng, v in ((eng_small, test_vocab), (eng_bench, bench_vocab)):
        orc = helpers.oracle_for(v)
        ids, oo = check_batch(eng, orc, data, offs)
        # (the random-letter pieces beyond 256 bytes are cut into fragments -- step 4b -- and stay as well)
        # (so are the CJK runs on these synthetic vocabularies -- their ideographs are random, no token spans two of them --;
        # the records proper are exercised by the TK_FLAT_CUT=0 engine below: a few thousand pieces)
        assert eng.last_stats()["handed_back"] == 0 and eng.cut_chunks() > 0
        check_batch(eng, orc, data, offs, False, False)
    monkeypatch.setenv("TK_FLAT_CUT", "0")              # without the cuts: beyond 256 bytes the document is handed back
    e1 = tk.Engine(bench_vocab["tokens"], bench_vocab["num_special"], bench_vocab["bos"], bench_vocab["eos"], device=0)
    ids1, oo1 = e1.encode_batch(data, offs, True, True)
    assert e1.last_stats()["handed_back"] == len(handed) and e1.cut_chunks() == 0 and e1.long_piece_records() > 2000
    assert np.array_equal(ids1, ids) and np.array_equal(oo1, oo)
    e1.close()
    monkeypatch.setenv("TK_FLAT_LONG", "0")
    e0 = tk.Engine(bench_vocab["tokens"], bench_vocab["num_special"], bench_vocab["bos"], bench_vocab["eos"], device=0)
    ids0, oo0 = e0.encode_batch(data, offs, True, True)
    assert e0.last_stats()["handed_back"] > len(handed) and e0.long_piece_records() == 0
    assert np.array_equal(ids0, ids) and np.array_equal(oo0, oo)
    e0.close()


def test_random_adversarial_vocabularies(tk):
    """Fresh vocabularies per run of this test (seeded): random multi-byte tokens over tiny alphabets in random rank order --
    tokens no merge sequence reaches, runs of equal pairs, chains that undercut -- and random texts over the same alphabets
    with pieces of every length class (2..8, 9..16, 17..32, 33..64 in the merge kernels' LDS columns, longer ones in pass 2,
    the longest through the round-based workgroup merges).  The HIP path against the oracle, id for id."""
    import random
    import gen_golden_merge as gg
    rng = random.Random(20260917)
    # (n_extra stays below the number of strings the alphabet can make: the generator draws until it has that many)
    for alphabet, n_extra, max_len in (("ab", 60, 6), ("abc", 250, 5), ("abcd ", 1500, 7), ("aé中", 90, 4), ("xyz'\n", 900, 9)):
        toks = gg.vocab_adversarial(rng, alphabet, n_extra, max_len)
        v = {"tokens": toks, "num_special": 7, "bos": 1, "eos": 2}
        eng = tk.Engine(toks, 7, 1, 2, device=0)
        orc = helpers.oracle_for(v)
        letters = [c for c in alphabet if c not in " '\n"]
        docs = []
        for n in list(range(1, 70)) + [80, 100, 130, 200, 513, 1500, 5000]:
            for _ in range(3 if n < 70 else 1):
                docs.append("".join(rng.choice(letters) for _ in range(n)).encode())            # one piece
                docs.append("".join(rng.choice(alphabet) for _ in range(n)).encode())           # whatever the split makes of it
        docs.append(("".join(rng.choice(letters) for _ in range(40)) + " ").encode() * 300)      # many class-3 pieces in one document
        data, offs = tk.pack_docs(docs)
        check_batch(eng, orc, data, offs)
        check_batch(eng, orc, data, offs, False, False)
        eng.close()


def test_small_batches_one_launch(tk, test_vocab, bench_vocab):
    """tk_encode_one and small tk_encode_batch calls (<= 1024 documents, <= 64 KiB) run as ONE launch (tk_small_kernel);
    ids identical to the oracle, a document that needs pass 2 falls back,
    the capacity check, every BOS / EOS combination, a batch just over either limit takes the pipeline."""
    for v in (test_vocab, bench_vocab):
        orc = helpers.oracle_for(v)
        e = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
        docs = [d for d in helpers.mixed_docs(30, 6, 30, max_len=3000) + helpers.random_unicode_docs(120) if len(d) <= 3000]
        n0 = e.small_path_calls()
        for i, d in enumerate(docs):
            bos, eos = bool(i & 1), bool(i & 2)
            got = e.encode_one(d, bos, eos).tolist()
            assert got == orc.encode(d, bos, eos), d[:80]
        served = e.small_path_calls() - n0
        assert served >= len(docs) - 15         # all but the few with a piece that does not fit a window
        # C1 shape: 1 000 x 64-byte ASCII strings, one call each (BASELINE configs[0])
        data, offs = corpus.generate("ascii", 1000, 64, seed=corpus.BASE_SEED)
        for d in corpus.docs_of(data, offs)[:300]:
            assert e.encode_one(d, True, True).tolist() == orc.encode(d, True, True)
        # small batches through tk_encode_batch
        n1 = e.small_path_calls()
        for n_docs in (1, 2, 17, 64, 200, 1024):
            sub = corpus.docs_of(data, offs)[:n_docs]
            if n_docs == 200:
                sub = sub[:100] + [b"", "é中".encode() * 5, b" " * 40, b"\n"] + sub[100:196]
            assert e.encode_docs(sub, True, False, validate_utf8=True) == [orc.encode(d, True, False) for d in sub]
        assert e.small_path_calls() - n1 == 6
        # over the limits: the batch pipeline (same ids)
        n2 = e.small_path_calls()
        many = corpus.docs_of(data, offs)[:1000] + [b"x"] * 30
        assert e.encode_docs(many, False, False) == [orc.encode(d, False, False) for d in many]
        big = [b"word " * 14000]                                      # 70 000 bytes
        assert e.encode_docs(big, False, True) == [orc.encode(big[0], False, True)]
        assert e.small_path_calls() == n2
        # a piece that does not fit a window: falls back to the pipeline, same ids
        longp = b"q" * 300 + b" tail"
        assert e.encode_one(longp, True, True).tolist() == orc.encode(longp, True, True)
        # invalid UTF-8 is reported on this path too
        with pytest.raises(tk.TokenizerError):
            e.encode_docs([b"ok", b"\xff\xfe"], False, False, validate_utf8=True)
        # capacity
        out = np.empty(3, np.uint32)
        with pytest.raises(tk.TokenizerError):
            e.encode_one(b"one two three four five six", True, True, out=out)
        e.close()


def test_tokenizer_encode_uses_one_launch(tk, bench_vocab):
    t = tk.Tekkenizer.from_file(bench_vocab["path"], device=0)
    orc = helpers.oracle_for(bench_vocab)
    eng = t.engine()
    n0 = eng.small_path_calls()
    for text in ("Hello, world!", "", "it's 12345 ...", "日本語 のテキスト"):
        for bos, eos in ((False, False), (True, True)):
            assert t.encode(text, bos, eos) == orc.encode(text.encode("utf-8"), bos, eos)
    assert eng.small_path_calls() - n0 == 8
    t.close()


def test_long_pieces_merged_in_rounds(tk, test_vocab, bench_vocab, monkeypatch):
    """csrc/tk_long.hip: a long piece that is not a vocabulary key is merged in ROUNDS by a workgroup (all occurrences of the
    minimum rank at once; tools/batched_merge_model.py) -- id for id what the one-merge-per-step order gives.  With
    TK_LONG_MIN=65 every piece beyond a window takes that path, so the adversarial vocabularies of
    tests/golden/merge_vectors.json (created pairs that rank BELOW the one being merged: the round is cut) and runs of
    one letter (every pair a candidate: only even offsets merge, across step and wave boundaries) go through it."""
    import json
    import os
    import random
    rng = random.Random(77)
    letters = "abcdefghijklmnopqrstuvwxyz"
    docs = [b"a" * 1000, b"a" * 32768, b"ab" * 9000, b"abc" * 700, bytes(rng.choice(letters.encode()) for _ in range(32768)),
            bytes(rng.choice(letters.encode()) for _ in range(5000)), b"x" * 65, b"y" * 127 + b" tail", b"head " + b"q" * 4097 + b" tail end",
            b"z" * 1023, b"z" * 1024, b"z" * 1025, b"ba" * 2000 + b"c", ("é" * 2000).encode(), ("中a" * 1500).encode(),
            b" " * 3000 + b"x", b"!" * 2500, b"a" * 8191 + b"b" + b"a" * 8192, b"hello world " * 50 + b"k" * 2050 + b" and " + b"j" * 3000]
    for n in (66, 129, 257, 2047, 2049, 16383, 16385):
        docs.append(bytes(rng.choice(b"ab") for _ in range(n)))
    for v in (test_vocab, bench_vocab):
        orc = helpers.oracle_for(v)
        exp = [orc.encode(d, True, True) for d in docs]
        monkeypatch.setenv("TK_FLAT_CUT", "0")   # (with the cut decomposition the random-letter pieces never get here)
        # (lm, force): force = every long piece takes the rounds; without it only the repetitive ones do (the shipped policy:
        # on a piece with many distinct pairs a round costs more than the handful of merges it makes)
        # force 1 = every long piece through the compacting rounds, 2 = through the lazy rounds, 0 = routed by pair diversity
        for lm, force in (("65", "1"), ("65", "2"), ("1024", "1"), ("1024", "2"), ("1024", "0"), ("0", "0")):
            monkeypatch.setenv("TK_LONG_MIN", lm)
            monkeypatch.setenv("TK_LONG_FORCE", force)
            e = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
            got = e.encode_docs(docs, True, True)
            for d, g, x in zip(docs, got, exp):
                assert g == x, (lm, force, d[:40], len(d))
            if lm == "0":
                assert e.round_path_docs() == 0
            else:
                assert e.round_path_docs() >= 10
            e.close()
    # the adversarial merge vocabularies (pairs that undercut): pieces of 65 .. 200 bytes through the round-based kernel
    # (TK_FLAT_LONG=0: such pieces hand their documents back instead of staying on the flat path as records)
    monkeypatch.setenv("TK_LONG_MIN", "65")
    monkeypatch.setenv("TK_FLAT_LONG", "0")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "merge_vectors.json")) as f:
        g = json.load(f)
    took = 0
    for vv, force in [(vv, f) for vv in g["vocabs"] for f in ("1", "2")]:
        monkeypatch.setenv("TK_LONG_FORCE", force)
        toks = [bytes.fromhex(t) for t in vv["tokens_hex"]]
        pieces = [(bytes.fromhex(p), ids) for p, ids in vv["pieces"] if len(p) // 2 >= 40]
        # longer pieces of the same alphabet, expected ids from the oracle (itself pinned by the vectors)
        orc = tk_oracle.Oracle(toks, vv["num_special"], 1, 2)
        alpha = sorted(set(b"".join(p for p, _ in pieces))) or [97]
        extra = [bytes(rng.choice(alpha) for _ in range(n)) for n in (300, 700, 1500, 4000)]
        try:
            for x in extra:
                x.decode("utf-8")
        except UnicodeDecodeError:
            extra = []
        e = tk.Engine(toks, vv["num_special"], 1, 2, device=0)
        got = e.encode_docs([p for p, _ in pieces] + extra, False, False)
        for (p, ids), gg in zip(pieces, got):
            assert gg == ids, (vv["name"], p[:60])
        for x, gg in zip(extra, got[len(pieces):]):
            assert gg == orc.encode(x, False, False), (vv["name"], len(x))
        took += e.round_path_docs()
        e.close()
    assert took > 100


def test_device_entry_checks_offsets_and_utf8(tk, eng_small, test_vocab):
    """tk_encode_batch_device_ex ("C callers get a validate flag"): corrupted offsets are TK_ERR_INVALID_ARG, a document that
    is not UTF-8 on its own -- one that starts inside a code point included -- is TK_ERR_INVALID_UTF8, a good batch gives the ids
    of the unchecked entry."""
    import torch
    docs = [b"hello world", "café 中文".encode(), b"", b"tail"]
    data, offs = helpers_pack(docs)
    d_b = torch.from_numpy(data).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    orc = helpers.oracle_for(test_vocab)
    eids, eoo = orc.encode_batch(data, offs, True, True)

    def run(o, checks, n_bytes=None):
        d_o = torch.from_numpy(np.asarray(o, dtype=np.uint64).astype(np.int64)).cuda()
        v_ids, v_oo = None, None
        p_ids, p_oo, n = eng_small.encode_batch_device(d_b.data_ptr(), d_o.data_ptr(), len(o) - 1, len(data) if n_bytes is None else n_bytes,
                                                       True, True, stream, checks=checks)
        ids = torch.as_tensor(tk.DeviceView(p_ids, n, "<i4"), device="cuda").cpu().numpy().view(np.uint32)
        return ids

    for checks in (tk.CHECK_OFFSETS, tk.CHECK_OFFSETS | tk.CHECK_UTF8, tk.CHECK_UTF8):
        assert np.array_equal(run(offs, checks), eids)
    good = [int(x) for x in offs]
    bad_sets = [[1] + good[1:], good[:2] + [good[1] - 1] + good[3:], good[:-1] + [good[-1] + 5], good[:-1] + [good[-1] - 1],
                [good[0], good[3], good[3] - 1, good[3], good[4]]]
    for o in bad_sets:
        with pytest.raises(tk.TokenizerError) as e:
            run(o, tk.CHECK_OFFSETS)
        assert e.value.code == tk.TK_ERR_INVALID_ARG, o
    # a boundary inside the two-byte char of document 1: offsets are fine, the documents are not UTF-8 on their own
    cut = good[1] + 4
    assert data[cut] & 0xC0 == 0x80
    split = [good[0], good[1], cut, good[3], good[4]]
    assert len(run(split, tk.CHECK_OFFSETS)) > 0             # (only the offsets are checked: accepted)
    with pytest.raises(tk.TokenizerError) as e:
        run(split, tk.CHECK_UTF8)
    assert e.value.code == tk.TK_ERR_INVALID_UTF8
    with pytest.raises(tk.TokenizerError) as e:
        run(offs, 64)
    assert e.value.code == tk.TK_ERR_INVALID_ARG


def helpers_pack(docs):
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    return np.frombuffer(b"".join(docs) or b"\0", dtype=np.uint8).copy(), offs


def test_memo_never_changes_an_id(tk, test_vocab, bench_vocab):
    """The memo of merged pieces (tk_ctx_set_memo; csrc/tk_hash.h MEMO): call after call on one context -- table empty, filling, full
    of another text's words, tiny and thrashing, cleared, switched off and on again, adaptive policy and always-on -- every call
    gives the oracle's ids, and a repeated text actually hits the table."""
    rng = np.random.default_rng(5)
    for v in (test_vocab, bench_vocab):
        orc = helpers.oracle_for(v)
        eng = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
        try:
            batches = []
            for sd, kind, n, dl in ((1, "ascii", 3000, 512), (2, "mixed", 400, 2048), (4, "zipf", 1500, 0), (7, "ascii", 3000, 512)):
                data, offs = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + sd)
                batches.append((data, offs))
            # words no vocabulary has: every one of them goes through the merge, and comes back in the next batch
            words = ["".join(rng.choice(list("qxzjkvwQXZ"), size=int(rng.integers(2, 8)))) for _ in range(400)]
            junk = [(" ".join(rng.choice(words, size=int(rng.integers(1, 80))))).encode() for _ in range(600)]
            batches.append(helpers_pack(junk))
            batches.append(helpers_pack(list(helpers.EDGE_DOCS)))
            for log2, policy in ((18, 1), (10, 1), (16, 0)):
                eng.set_memo(log2, policy)
                hits = 0
                for rnd in range(3):
                    for data, offs in batches:
                        check_batch(eng, orc, data, offs)
                        st = eng.memo_stats()
                        hits += st["hits_last"]
                        assert st["hits_last"] <= st["lookups_last"]
                    if rnd == 1:
                        eng.memo_clear()
                if policy == 1:
                    assert hits > 0, "the memo was never hit (log2 %d)" % log2
            eng.set_memo(0)
            check_batch(eng, orc, *batches[0])
            assert not eng.memo_stats()["active_last"]
            eng.set_memo(20, 1)
            check_batch(eng, orc, *batches[4])
            check_batch(eng, orc, *batches[4])
            st = eng.memo_stats()
            assert st["active_last"] and st["hits_last"] > 0.3 * st["lookups_last"], st
            for bad in (5, 27, -1):
                with pytest.raises(tk.TokenizerError):
                    eng.set_memo(bad)
        finally:
            eng.close()


def test_crlf_run_behind_a_char_the_region_cuts(tk, test_vocab, bench_vocab):
    """The GPU fuzz's find of round 4 (tests/test_flat_path.py has the story): a region that begins inside a multi-byte char does not
    know whether the CR / LF run behind it is that char's absorbed tail or white space; the document is handed back.  The fuzz's
    own text at 30 alignments around the region start, each document alone in its batch and all of them in one, both vocabularies."""
    rle = [(0x4e2d, 26), (0x663, 1), (0xe9, 28), (0x21, 3), (0x20, 13), (0x9, 17), (0x3000, 11), (0xd, 40), (0x9, 11), (0xd, 22), (0x27, 5),
           (0x663, 13), (0xff13, 32), (0x21, 13), (0x2d, 22), (0xd, 14), (0x4e2d, 16)]
    frag = "".join(chr(c) * n for c, n in rle).encode()
    at = frag.index(b"\r" * 40)
    docs = [(b"ab cd\n" * 800)[:2 * 1952 - 32 - at + 1 + shift] + frag for shift in range(-15, 15)]
    for ch in ("\u2026", "\u3000", "\U0001f680"):                  # punctuation (the run IS its tail), white space, an emoji
        for k in range(1, len(ch.encode())):
            docs.append((b"xy z\n" * 800)[:1952 - 32 - k] + ch.encode() + b"\n" * 45 + b"\t\t next" + b" words" * 30)
    for v in (test_vocab, bench_vocab):
        orc = helpers.oracle_for(v)
        eng = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
        try:
            for d in docs:
                check_batch(eng, orc, *helpers_pack([d]), False, False)
            check_batch(eng, orc, *helpers_pack(docs))
        finally:
            eng.close()


def test_json_pattern_chain_from_below_the_region(tk, test_vocab):
    """JSON pattern: a chain of tail chars (CR / LF / '/'), punctuation and marks that comes from below the region and covers the
    left halo hands its document back (tests/test_flat_path.py has the story; found by the model campaign of round 4).  The
    campaign's text with the region start at every offset inside the CRs, each document alone in its batch."""
    text = ("x1-----" + "\r" * 40 + "/" * 11 + "\u0301" * 31 + "'''''\r\r\U0001f680 And more Text 12.\n").encode()
    cr0 = text.index(b"\r")
    orc = tk_oracle.Oracle(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"])
    orc.set_pattern(1)
    e = tk.Engine(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], device=0)
    e.set_pattern(1)
    try:
        docs = [(b"ab cd\n" * 400)[:1952 - 32 - cr0 - off] + text for off in range(0, 52)]
        for d in docs:
            assert e.encode_docs([d], False, False) == [orc.encode(d, False, False)]
        assert e.encode_docs(docs, True, True) == [orc.encode(d, True, True) for d in docs]
    finally:
        e.close()
