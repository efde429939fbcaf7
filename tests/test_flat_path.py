"""The FLAT path (tekken-rs_amd/csrc/tk_flat_impl.h: one wave per 2048-byte region of the packed stream, masks
in lane layout, document boundaries as a mask) on the CPU:
  * tools/flat_split_model.py -- the rules as mask algebra on Python ints -- against the oracle split;
  * the device source on the wave emulator (tests/emu) against the oracle, id for id, including the
    documents the fast path hands back to the per-document algorithm."""
import itertools
import os
import random

import corpus
import emu
import flat_split_model as fm
import helpers
import tk_oracle


def _model_check(docs, **kw):
    data = b"".join(docs)
    offs = [0]
    for d in docs:
        offs.append(offs[-1] + len(d))
    starts, deferred = fm.flat_split_chunked(data, offs, **kw)
    exp = []
    for i, d in enumerate(docs):
        if i not in deferred:
            exp += [offs[i] + s for s in tk_oracle.split(d)]
    assert starts == exp
    return deferred


def test_model_baseline_shapes():
    d, o = corpus.generate("ascii", 300, 512, seed=corpus.BASE_SEED + 1)
    assert not _model_check(corpus.docs_of(d, o))          # nothing handed back on the headline workload
    d, o = corpus.generate("zipf", 200, seed=corpus.BASE_SEED + 4)
    _model_check([x for x in corpus.docs_of(d, o) if len(x) < 9000])
    _model_check(helpers.EDGE_DOCS)


def test_model_exhaustive_small_alphabet_packed():
    """every string of length <= 5 over an 8-symbol alphabet, packed into ONE stream: document boundaries
    fall at every offset of the lanes and of the chunk."""
    alpha = ["a", "s", "1", "'", "!", " ", "\n", "\t"]
    docs = ["".join(t).encode() for n in range(0, 6) for t in itertools.product(alpha, repeat=n)]
    assert not _model_check(docs, region=256)


def test_model_random_and_runs():
    rng = random.Random(5)
    alpha = list("aSstrelvmdx12'! \n\r\t-") + [" "] * 3
    for _ in range(120):
        docs = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, rng.choice([3, 40, 300])))).encode()
                for _ in range(rng.randint(1, 40))]
        _model_check(docs, region=rng.choice([256, 1024, 2048]))
    alpha2 = list("a1 \n!'s")
    for _ in range(120):
        docs = ["".join(rng.choice(alpha2) * rng.choice([1, 1, 1, 2, 5, 40, 100]) for _ in range(rng.randint(0, 30))).encode()
                for _ in range(rng.randint(1, 10))]
        _model_check(docs, region=256)
    # runs of multi-byte chars of every class: region starts fall inside chars, runs cover halos (round 4: the fuzz's find was of this kind)
    alpha3 = ["a", "1", " ", "\n", "\r", "\t", "!", "'", "\u00e9", "\u0663", "\uff13", "\u3000", "\u2026", "\u4e2d", "\U0001f680", "\u00a0", "\u0301"]
    for _ in range(400):
        docs = ["".join(rng.choice(alpha3) * rng.choice([1, 1, 1, 2, 5, 11, 22, 40]) for _ in range(rng.randint(0, 40))).encode()
                for _ in range(rng.randint(1, 6))]
        _model_check(docs, region=rng.choice([256, 256, 2048]))


def _emu_check(v, docs, bos=True, eos=True, check_split=True):
    o = helpers.oracle_for(v)
    ids, starts, flagged = emu.flat_encode_batch(v["tokens"], v["num_special"], v["bos"], v["eos"], docs, bos, eos)
    for i, d in enumerate(docs):
        assert ids[i] == o.encode(d, bos, eos), (i, d[:80], i in flagged)
        if check_split and i not in flagged:
            assert starts[i] == tk_oracle.split(d), (i, d[:80])
    return flagged


def test_emu_flat_baseline_shapes(test_vocab):
    d, o = corpus.generate("ascii", 40, 512, seed=corpus.BASE_SEED + 1)
    assert _emu_check(test_vocab, corpus.docs_of(d, o)) == []
    d, o = corpus.generate("zipf", 60, seed=corpus.BASE_SEED + 4)
    _emu_check(test_vocab, [x for x in corpus.docs_of(d, o) if len(x) < 6000])
    for bos, eos in ((False, False), (True, False)):
        _emu_check(test_vocab, helpers.EDGE_DOCS, bos, eos)


def test_crlf_run_behind_a_char_the_region_cuts(test_vocab):
    """A region that begins INSIDE a multi-byte char cannot know that char's class: a CR / LF run right behind it is either the tail
    its piece absorbs (the char was punctuation: `[^\\s\\p{L}\\p{N}]+[\\r\\n]*`) or part of a white-space run (it was U+3000) -- and the
    pieces behind the run differ.  Where such a run covers the rest of the left halo the document is handed back (hand-back rule A;
    found by the GPU fuzz of round 4: U+3000, forty CRs, eleven TABs, twenty-two CRs -- the CRs looked absorbed, a piece began at
    the first TAB and the fragment behind it was emitted twice).  Every alignment of the char over the region start, both kinds
    of char, run lengths around the halo, on the model and on the emulated kernel."""
    tails = [b"\r" * 40 + b"\t" * 11 + b"\r" * 22 + b"'''''abc", b"\n" * 33 + b"  x", b"\r\n" * 20 + b"\t\ty", b"\r" * 30 + b" z", b"\n" * 64 + b"\t" * 3 + b"\n\nq"]
    chars = ["\u3000".encode(), "\u2026".encode(), "\u00a0".encode(), "\U0001f680".encode(), "\u4e2d".encode()]   # white space, punctuation, NBSP, emoji, letter
    docs = []
    for region, commit in ((256, 160), (2048, 1952)):
        for ch in chars:
            for tail in tails:
                for k in range(len(ch) + 1):
                    # the char ends k bytes behind the start of the second region (= commit - 32)
                    pad_len = commit - 32 - (len(ch) - k) + (commit if region == 256 else 0)
                    pad = (b"ab cd\n" * 400)[:pad_len]
                    doc = pad + ch + tail + b" tail of the document.\n" * 3
                    if region == 256:
                        deferred = _model_check([doc], region=256)
                    docs.append(doc)
    # the fuzz's own text: the white-space piece is 136 bytes long (cut into fragments), the region of the next chunk begins
    # inside the last U+3000 -- 30 alignments around it
    rle = [(0x4e2d, 26), (0x663, 1), (0xe9, 28), (0x21, 3), (0x20, 13), (0x9, 17), (0x3000, 11), (0xd, 40), (0x9, 11), (0xd, 22), (0x27, 5),
           (0x663, 13), (0xff13, 32), (0x21, 13), (0x2d, 22), (0xd, 14), (0x4e2d, 16)]
    frag = "".join(chr(c) * n for c, n in rle).encode()
    at = frag.index(b"\r" * 40)                       # the CRs begin here; the last U+3000 ends here
    for shift in range(-15, 15):
        pad_len = 2 * 1952 - 32 - at + 1 + shift      # shift 0: the region of chunk 2 begins at the U+3000's last byte
        _emu_check(test_vocab, [(b"ab cd\n" * 800)[:pad_len] + frag], False, False, check_split=False)   # (alone: the alignment is the point)
    for doc in docs[::7]:
        _emu_check(test_vocab, [doc], False, False, check_split=False)
    _emu_check(test_vocab, docs, check_split=False)
    _emu_check(test_vocab, [b"".join(docs[:40])], False, False, check_split=False)       # the same runs at other alignments


def test_every_kind_of_run_behind_a_char_the_region_cuts(test_vocab):
    """The same situation for every class of run and of cut char: a region begins inside a 2-, 3- or 4-byte char (every byte
    offset) and a run of digits / letters / punctuation / blanks / line ends / mixed white space of 1..80 bytes follows, then a
    char of another class.  Model (256-byte regions) and emulated kernel (each document alone: the alignment is the point)."""
    chars = ["\u00e9", "\u0663", "\u3000", "\u2026", "\uff13", "\u4e2d", "\U0001f680", "\u0301"]   # letter, digit, space, punctuation, digit, letter, emoji, mark
    runs = ["7", "a", "!", " ", "\n", "\r\n", "\t", " \n", "\n "]
    after = ["x", "9", "?", " y", "\n\nz", "'s", "\t\t\n\nx", " \r\n!", "\t\r\r7"]
    rng = random.Random(17)
    emu_docs, n_model = [], 0
    for ch in chars:
        cb = ch.encode()
        for run in runs:
            for n in (1, 2, 3, 29, 30, 31, 32, 33, 34, 40, 63, 64, 65, 80):
                body = (run * n)[:n].encode() + rng.choice(after).encode() + b" and the rest of the document\n"
                for k in range(1, len(cb)):
                    # model: the second region of 256 bytes (commit 160) begins k bytes into the char
                    doc = (b"ab cd\n" * 60)[:160 - 32 + 160 - k] + cb + body
                    _model_check([doc], region=256)
                    n_model += 1
                    if (n in (30, 31, 32, 33, 40, 64, 80) or rng.random() < 0.15):
                        emu_docs.append((b"ab cd\n" * 400)[:1952 - 32 - k] + cb + body)
    assert n_model > 1500 and len(emu_docs) > 700
    for doc in emu_docs:
        _emu_check(test_vocab, [doc], False, False, check_split=False)


def test_emu_flat_sparse_miss_queues(test_vocab):
    """A few queued pieces in ~300 chunks: the later items of a merge wave lie more than three 64-entry windows of
    sub-queues behind the first, so the wave finds them by bisection (csrc/tk_flat_impl.h tk_merge_wave), narrow and wide."""
    filler = b"a\nb\nc\nd\ne\nf\ng\nh\ni\nj\nk\nl\nm\nn\no\np\nq\nr\ns\nt\nu\nv\nw\nx\ny\nz\n" * 40      # one-byte pieces: never queued
    docs = [filler] * 290
    for d, w in ((0, b"qzxjv"), (140, b"kqjxzvwpyfbg"), (280, b"zqxjkvbwpfgmhdcylrtn"), (3, b"xzqjvkwbpfmgyhdclrtnxzqjvkwbpfmgyhdclrtn"),
                 (285, b"jqzxv"), (288, b"vkqjxzwpbfgyhmdclrtnsaeio")):
        docs[d] = filler[:520] + w + b"\n" + filler[520:]
    assert _emu_check(test_vocab, docs) == []


def test_emu_flat_long_pieces_stay_on_the_flat_path(test_vocab, monkeypatch):
    """Pieces of 65..256 bytes (tk_flat_impl.h step 6, tk_flat_long_wave): their documents are NOT handed back -- the piece
    becomes a record, its ids come from one wave's lookup / merge into the reserved slots; where the chunk does not see the
    end of its last piece (the piece crosses the region end) the sequential matcher finds it; beyond 256 bytes the document
    is handed back after all unless the piece can be cut (step 4b; TK_FLAT_CUT=0 shows the hand-back).  Letter runs, CJK runs, punctuation runs, at every offset around a chunk boundary."""
    import random
    rng = random.Random(5)
    filler = b"ab cd ef gh ij kl mn op qr st uv wx yz " * 60                 # 2340 bytes of short pieces
    kept, handed = [], []
    for n in (64, 65, 100, 128, 200, 254, 255):                               # (the blank in front belongs to the piece: n + 1 bytes)
        w = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(n)).encode()
        kept.append(b"x " + w + b" y")
        kept.append(filler[:1899] + w + b" " + filler[:300])                  # crosses the end of the first chunk's region or not
    for k in range(1880, 2030, 7):                                            # the piece's first byte walks over the chunk boundary
        w = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(150)).encode()
        kept.append(filler[:k] + b" " + w + b" " + filler[:200])
    kept.append(("中" * 30 + "，" + "文" * 60 + "。" + "字" * 80).encode())          # 93- / 183- / 243-byte pieces
    kept.append(b"head " + b"=" * 200 + b"\n\ntail")
    kept.append((" " + "é" * 100 + " x").encode())
    for n in (256, 300, 1000):
        w = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(n)).encode()
        handed.append(b"x " + w + b" y")
        handed.append(filler[:1899] + w + b" z")
    docs = kept + handed
    assert _emu_check(test_vocab, docs) == []                                 # (beyond 256 bytes: cut into fragments, step 4b)
    monkeypatch.setenv("TK_FLAT_CUT", "0")                                    # without the cuts those documents are handed back
    flagged = _emu_check(test_vocab, docs)
    assert set(flagged) == set(range(len(kept), len(docs))), flagged
    for bos, eos in ((False, False), (True, False)):
        _emu_check(test_vocab, docs, bos, eos)


def test_emu_cut_decomposition(test_vocab):
    """Step 4b of the flat kernel (CUT instantiation; model tools/cut_model.py): a piece of more than 64 bytes is cut wherever no
    vocabulary token can span the boundary, its fragments merge independently (no whole-piece look-up for a fragment) and its
    document stays on the flat path -- letter runs of 65 bytes .. 6 KB that begin, end and continue at every kind of offset
    around the chunk boundaries, several per stream, on the trained vocabulary (cuts everywhere) and on adversarial ones
    (few cuts: fragments of more than 64 bytes become records, open fragments hand their document back)."""
    import random
    import gen_golden_merge as gg
    rng = random.Random(5)
    letters = "abcdefghijklmnopqrstuvwxyz"
    filler = b"ab cd ef gh ij kl mn op qr st uv wx yz " * 120
    docs = []
    for n in (65, 96, 97, 200, 256, 257, 300, 1000, 1888, 1952, 2000, 2048, 4000, 6000):
        w = "".join(rng.choice(letters) for _ in range(n)).encode()
        docs += [w, b"x " + w + b" y", filler[:1899] + w + b" " + filler[:300]]
    for k in range(1850, 2080, 9):                                            # the run's first byte / last byte walks over a chunk boundary
        w = "".join(rng.choice(letters) for _ in range(400)).encode()
        docs.append(filler[:k] + b" " + w + b" " + filler[:100])
        docs.append(w[:k % 400 + 70] + b" " + filler[:50])
    san = bool(os.environ.get("TK_TEST_SANITIZE"))          # (the sanitizer leg of tests/test_sanitizers.py: a thinned-out set)
    if san:
        docs = docs[::9]
    assert _emu_check(test_vocab, docs) == []
    for bos, eos in ((False, False), (True, False)):
        _emu_check(test_vocab, docs[:12], bos, eos)
    # a piece without a cut in the stretch two chunks share (the first 64 bytes behind a commit boundary) belongs to ONE of them:
    # 131 x 't' + 'm' from commit offset 1914 on -- the only cut (before the 'm') lies 93 bytes into the next chunk, which must
    # leave the whole piece to the long-piece record of the chunk it starts in (found by tools/gpu_fuzz_long.py: the 'm' came twice)
    for k in range(1890, 1960, 40 if san else 3):
        for run in (131, 250) if san else (100, 131, 180, 250):
            _emu_check(test_vocab, [filler[:k] + b" " + b"t" * run + b"m" + "٣٣٣٣٣中中中".encode() + b" " + filler[:100]], check_split=False)
    n_docs = n_flagged = 0
    for alphabet, n_extra, max_len in (("ab", 60, 6), ("abc", 250, 5), ("abcdefgh", 300, 5), ("aé中", 90, 4), ("abcdefghijklmnop", 200, 4), ("ab \n", 40, 5))[:1 if san else 6]:
        toks = gg.vocab_adversarial(rng, alphabet, n_extra, max_len)
        v = {"tokens": toks, "num_special": 5, "bos": 1, "eos": 2}
        docs = []
        for n in (65, 97, 700, 2048) if san else (64, 65, 66, 95, 96, 97, 130, 257, 700, 1900, 1952, 2048, 4100):
            w = "".join(rng.choice(alphabet) for _ in range(n)).encode()
            docs.append(filler[:rng.randint(0, 2100)] + w + b" " + filler[:rng.randint(0, 300)])
            docs.append(w)
        n_flagged += len(_emu_check(v, docs, check_split=False))
        n_docs += len(docs)
    assert san or n_flagged < n_docs // 2


def test_emu_long_records_on_adversarial_vocabularies():
    """The two merges behind the long-piece records -- 128-entry LDS columns (65..128 bytes), parts in registers (129..256) --
    on vocabularies built to hurt (random multi-byte tokens in random rank order over tiny alphabets: chains that undercut,
    runs of equal pairs), pieces of every length 65..256, against the oracle."""
    import random
    import gen_golden_merge as gg
    rng = random.Random(77)
    for alphabet, n_extra, max_len in (("ab", 60, 6), ("abc", 250, 5), ("aé中", 90, 4)):
        toks = gg.vocab_adversarial(rng, alphabet, n_extra, max_len)
        v = {"tokens": toks, "num_special": 5, "bos": 1, "eos": 2}
        docs = []
        for n in range(65, 257, 3):
            w = "".join(rng.choice(alphabet) for _ in range(n))
            while len(w.encode()) > 256:
                w = w[:-1]
            docs.append(b"x " + w.encode() + b" y")
            docs.append((rng.choice(alphabet) * (n // 2) + w)[:120].encode() + b"\n" + w.encode()[:200].decode("utf-8", "ignore").encode())
        flagged = _emu_check(v, docs, check_split=False)
        assert len(flagged) < len(docs) // 2        # (most of them stay on the flat path: the records are what is tested)


def test_emu_flat_handback_and_mixed(test_vocab):
    # (a white-space run that crosses a region end is always handed back; long letter runs no longer are: step 4b)
    docs = helpers.mixed_docs(8, 8, 8, max_len=3000) + helpers.random_unicode_docs(120) + [b"a b" + b" " * 3000 + b"x", b"1" * 2500]
    flagged = _emu_check(test_vocab, docs)
    assert flagged and len(flagged) < len(docs)           # both routes were taken inside one stream
    # empty documents in every position, a single document, only empty documents
    _emu_check(test_vocab, [b"", b"ab cd", b"", b"", b"x" * 100, b""])
    _emu_check(test_vocab, [b"hello world"])
    _emu_check(test_vocab, [b"", b""])


def test_emu_flat_small_alphabet_packed(small_vocab):
    alpha = ["a", "s", "1", "'", "!", " ", "\n", "\t"]
    docs = ["".join(t).encode() for n in range(0, 5) for t in itertools.product(alpha, repeat=n)]
    _emu_check(small_vocab, docs, False, False)


def test_emu_flat_runs_and_misses(test_vocab):
    """digit / white-space / CR-LF runs across lanes and chunk boundaries; random letter strings (pieces that
    miss the vocabulary: the packed-window merge, several pieces per window, extra ids shifting later slots)."""
    rng = random.Random(21)
    alpha = ["a", "b", "1", "2", "'", "s", "t", "!", "-", " ", " ", " ", "\n", "\r", "\t"]
    docs = []
    for _ in range(120):
        n = rng.randint(60, 700)
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(rng.choice(alpha) * rng.choice([1, 1, 1, 2, 3, 5, 9, 17, 40]))
        docs.append("".join(parts).encode())
    _emu_check(test_vocab, docs)
    words = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(1, 30))) for _ in range(3000)]
    docs = [" ".join(rng.choice(words) for _ in range(rng.randint(1, 120))).encode() for _ in range(25)]
    assert _emu_check(test_vocab, docs) == []


def test_emu_flat_every_ascii_byte_pair(small_vocab):
    """classification by bit planes: every pair of ASCII byte values, at every lane offset (documents of 4 bytes
    packed back to back, a 5-byte document now and then to shift the phase)."""
    docs = []
    for b1 in range(128):
        for b2 in range(0, 128, 1 if b1 in (0x27, 0x20, 10, 13, 9, 0x30, 0x41, 0x61, 0x7F, 0) else 7):
            docs.append(bytes([0x61, b1, b2, 0x31]))
        docs.append(b"'re s")
    _emu_check(small_vocab, docs, False, False)
    # multi-byte code points stay on the fast path
    assert _emu_check(small_vocab, [b"ab", bytes([0x61, 0xC3, 0xA9]), b"cd"], False, False) == []


def test_model_utf8():
    """multi-byte code points in the mask model: classes on all bytes of a char, char-level digit rule, U+017F."""
    _model_check(helpers.random_unicode_docs(1500, seed=3, max_len=80), region=256)
    d, o = corpus.generate("mixed", 60, 2048, seed=corpus.BASE_SEED + 2)
    assert not _model_check(corpus.docs_of(d, o))
    rng = random.Random(9)
    alpha = ["a", "S", "1", "\u0663", "\uff13", "'", "\u017f", "s", "!", " ", " ", "\n", "\r", "\u4e2d", "\u00e9", "\U0001f680",
             "\u00a0", "\u3000", "-", "\t"]
    for _ in range(150):
        docs = ["".join(rng.choice(alpha) * rng.choice([1, 1, 1, 2, 3, 7, 20]) for _ in range(rng.randint(0, 25))).encode()
                for _ in range(rng.randint(1, 12))]
        _model_check(docs, region=rng.choice([256, 1024, 2048]))


def test_emu_flat_utf8(test_vocab):
    """the same on the device source: trie classes per lead byte, chars that straddle lanes and regions, multi-byte
    digit runs (three chars at a time), multi-byte white space at the end of a run, the long s contraction."""
    d, o = corpus.generate("mixed", 12, 2048, seed=corpus.BASE_SEED + 2)
    assert _emu_check(test_vocab, corpus.docs_of(d, o)) == []
    _emu_check(test_vocab, helpers.random_unicode_docs(250, seed=5, max_len=200))
    rng = random.Random(10)
    alpha = ["a", "S", "1", "\u0663", "\uff13", "'", "\u017f", "s", "!", " ", " ", "\n", "\r", "\u4e2d", "\u00e9", "\U0001f680",
             "\u00a0", "\u3000", "-", "\t"]
    docs = ["".join(rng.choice(alpha) * rng.choice([1, 1, 1, 2, 3, 7, 20]) for _ in range(rng.randint(0, 40))).encode()
            for _ in range(150)]
    _emu_check(test_vocab, docs)
    _emu_check(test_vocab, ["x'\u017f y'\u017fz '\u017f".encode(), ("\uff11" * 300).encode(), ("\u4e2d\u6587" * 10 + " ").encode() * 30])


def test_emu_flat_every_bmp_code_point(small_vocab):
    """The range rules of tkf_classify (CJK unified, Hangul, basic Cyrillic / Greek / Arabic, Latin-1 letters: whole blocks named
    by lead byte and a range of the second byte, taken as letters without decode or table look-up) and the per-char walk for
    everything else: EVERY code point of the BMP (and a sample beyond it) between a letter, a digit and a blank, split by the
    flat path, against the oracle's split, which reads the class trie -- a block that a rule claims and the trie does not hold
    as letters shows up here."""
    cps = [c for c in range(0x80, 0x10000) if not 0xD800 <= c <= 0xDFFF] + list(range(0x1F600, 0x1F650)) + list(range(0x20000, 0x20040)) + [0x10FFFF]
    docs, cur = [], []
    for c in cps:
        ch = chr(c)
        cur.append("a%sb %s1 %s%s " % (ch, ch, ch, ch))
        if len(cur) == 400:
            docs.append("".join(cur).encode("utf-8"))
            cur = []
    docs.append("".join(cur).encode("utf-8"))
    ids, starts, flagged = emu.flat_encode_batch(small_vocab["tokens"], small_vocab["num_special"], small_vocab["bos"], small_vocab["eos"], docs, False, False)
    o = helpers.oracle_for(small_vocab)
    for i, d in enumerate(docs):
        if i not in flagged:
            assert starts[i] == tk_oracle.split(d), (i, d[:60])
        assert ids[i] == o.encode(d, False, False), i
    assert len(flagged) < len(docs) // 10


def test_key_hash_fallback_mode(test_vocab):
    """Three 12-byte tokens built to have the SAME cheap key hash (mode 0 folds the upper 8 bytes into the lower ones):
    a cuckoo slot pair cannot hold three keys, so the table builder has to fall back to the strong hash (mode 1) --
    and the lookups must still be exact, for these tokens and for everything else."""
    import struct

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF
    x, k1 = 0x6C6C6568, 0x6F77206F                       # "hell", "o wo"
    extra = []
    for k2 in (0x61616161, 0x62626262, 0x63636363):      # "aaaa", "bbbb", "cccc"
        extra.append(struct.pack("<III", x ^ rotl(k2, 13), k1, k2))
    assert len(set(extra)) == 3
    toks = list(test_vocab["tokens"]) + [t for t in extra if t not in test_vocab["tokens"]]
    v = dict(test_vocab, tokens=toks)
    assert emu.table_info(test_vocab["tokens"], test_vocab["num_special"])["key_hash_mode"] == 0
    info = emu.table_info(toks, v["num_special"])
    assert info["key_hash_mode"] == 1 and info["flagged_slots"] <= info["keys_in_second_slot"]
    docs = [b"hello world aaaa", b"plain text stays exact 123", extra[0] + b" " + extra[1], extra[2]]
    _emu_check(v, docs, check_split=False)
    o = helpers.oracle_for(v)
    base = len(test_vocab["tokens"])
    # a piece that IS one of the colliding tokens comes back as that single token (whole-piece lookup, not a merge)
    got, _, _ = emu.flat_encode_batch(v["tokens"], v["num_special"], v["bos"], v["eos"], [b"x" + extra[0][1:]], False, False)
    assert got[0] == o.encode(b"x" + extra[0][1:], False, False)
    assert base + 0 < len(toks)


def _boundary_docs(pads, lens):
    """runs of every class placed so that they start / end around the region geometry (32-byte left halo, 928 or 1952
    bytes committed -- 16 or 32 bytes per lane --, 64-byte right halo, 64-byte piece limit)"""
    docs = []
    for pad in pads:
        for ch in ("1", "\n", " ", "a", "!", "中", "１"):
            for rl in lens:
                docs.append(("x y " * (pad // 4) + "q" * (pad % 4) + ch * rl + " z").encode())
    return docs


def test_emu_flat_region_geometry(small_vocab):
    for pads in ((864, 896, 927, 928, 960), (1888, 1920, 1951, 1952, 1984)):
        docs = _boundary_docs(pads, (31, 32, 33, 64, 65))
        flagged = _emu_check(small_vocab, docs, False, False, check_split=True)
        assert 0 < len(flagged) < len(docs)
    # total length a multiple of the commit size, documents ending exactly on chunk / region boundaries
    _emu_check(small_vocab, [b"ab " * 309 + b"c", b"d" * 32, b"e f" * 298 + b"gh", b"", b"i" * 928, b"j k " * 232], True, True)
    _emu_check(small_vocab, [b"ab " * 650 + b"cd", b"d" * 32, b"e f" * 639 + b"ghi", b"", b"i" * 1952, b"j k " * 488], True, True)


def test_emu_flat_dense_pieces(test_vocab):
    """Regions with more pieces than the LDS list of the 32-bytes-per-lane kernel holds (an average of under two bytes
    per piece over 2 KB): the pieces are enumerated in two passes; document starts, misses and long runs on both sides."""
    rng = random.Random(77)
    docs = [(b"1,2,3,4,5,6,7,8,9,0;" * 300), b"a b c d e f g h i j " * 250, b"!a!b!c" * 700,
            b"".join(bytes([rng.choice(b"0123456789,.;:-+ ")]) for _ in range(9000)),
            b"x" * 40 + b"1,2," * 600 + b" zzzzzz " + b"3;4;" * 500]
    docs += [b"7," * rng.randint(1, 40) for _ in range(300)]          # many short documents inside dense regions
    docs += [(b"q" * 70 + b",1" * 900)]                                 # a piece over 64 bytes inside a dense region
    flagged = _emu_check(test_vocab, docs)
    assert len(flagged) <= 2


def _json_oracle(v):
    o = tk_oracle.Oracle(v["tokens"], v["num_special"], v["bos"], v["eos"])
    o.set_pattern(1)
    return o


def _emu_check_json(v, docs, bos=True, eos=True):
    o = _json_oracle(v)
    ids, starts, flagged = emu.flat_encode_batch(v["tokens"], v["num_special"], v["bos"], v["eos"], docs, bos, eos, pattern=1)
    for i, d in enumerate(docs):
        assert ids[i] == o.encode(d, bos, eos), (i, d[:80], i in flagged)
        if i not in flagged:
            assert starts[i] == tk_oracle.split_tekken(d), (i, d[:80])
    return flagged


def test_model_json_pattern():
    """Row f-3: the JSON pattern's rules as mask algebra (tools/flat_split_model.py flat_rules_tekken) against the oracle's
    matcher: case changes, single digits, the CR / LF / '/' tail, accented / Cyrillic letters, neutral letters (Lm / Lo:
    CJK, modifier letters, ordinal indicators), marks inside words and inside punctuation runs, titlecase."""
    rng = random.Random(4)
    alpha = list("aAbBzZ") + [" "] * 4 + ["1", "2", "!", "/", "-", "'", "\n", "\r", "\t", "é", "É", "Ж", "ж", "٣", "　",
                                            "中", "文", "ʰ", "ª", "́", "̈", "ǅ", "!", ".", "\U0001f680", "¿"]
    for _ in range(600):
        docs = ["".join(rng.choice(alpha) * rng.choice([1, 1, 1, 2, 3]) for _ in range(rng.randint(0, rng.choice([4, 30, 200, 1500])))).encode()
                for _ in range(rng.randint(1, 12))]
        data = b"".join(docs)
        offs = [0]
        for d in docs:
            offs.append(offs[-1] + len(d))
        starts, deferred = fm.flat_split_chunked_tekken(data, offs, region=rng.choice([256, 2048]))
        exp = []
        for i, d in enumerate(docs):
            if i not in deferred:
                exp += [offs[i] + s for s in tk_oracle.split_tekken(d)]
        assert starts == exp
    _, deferred = fm.flat_split_chunked_tekken("abc 中文 def".encode() + b"Plain TextHere 12", [0, 14, 31], region=256)
    assert deferred == set()                       # neutral letters (Lo) stay on the fast path


def test_model_json_pattern_runs_behind_a_char_the_region_cuts():
    """The JSON pattern's rules in the situation the round-4 fuzz found for the default pattern: a region that begins inside a
    multi-byte char, every class of run of 1..80 bytes behind it (its absorbed tail runs through CR / LF / '/')."""
    chars = ["\u00e9", "\u00c9", "\u0663", "\u3000", "\u2026", "\uff13", "\u4e2d", "\U0001f680", "\u0301"]
    runs = ["7", "a", "A", "!", " ", "\n", "\r\n", "/", "\n/", "\t", " \n", "\n "]
    after = ["x", "X", "9", "?", " y", "\n\nz", "\t\t\n\nx", " \r\n!", "\t\r\r7", "/a"]
    rng = random.Random(23)
    n_checked = 0
    for ch in chars:
        cb = ch.encode()
        for run in runs:
            for n in (1, 2, 3, 29, 30, 31, 32, 33, 34, 40, 63, 64, 65, 80):
                body = (run * n)[:n].encode() + rng.choice(after).encode() + b" And The Rest of the document\n"
                for k in range(1, len(cb)):
                    doc = (b"ab cd\n" * 60)[:160 - 32 + 160 - k] + cb + body
                    starts, deferred = fm.flat_split_chunked_tekken(doc, [0, len(doc)], region=256)
                    if not deferred:
                        assert starts == tk_oracle.split_tekken(doc), (ch, run, n, k)
                        n_checked += 1
    assert n_checked > 300


def test_json_pattern_chain_of_tail_punctuation_and_marks_from_below(test_vocab):
    """JSON pattern: whether a mark is a word char or punctuation depends on whether the 4th alternative is running, which depends on
    whether the chars before it are the TAIL [\\r\\n/]* of a punctuation piece -- a chain of tail chars, punctuation and marks that
    comes from below the region and covers the left halo leaves the state at the commit start unknown: the document is handed back
    (hand-back rule A of the JSON instantiation counted runs of ONE kind; found by the model campaign of round 4).  The campaign's
    text with the region start at every offset inside the CRs, model and emulated kernel."""
    text = ("x1-----" + "\r" * 40 + "/" * 11 + "\u0301" * 31 + "'''''\r\r\U0001f680 And more Text 12.\n").encode()
    cr0 = text.index(b"\r")
    for off in range(0, 52):
        doc = (b"ab cd\n" * 60)[:160 - 32 + 160 - cr0 - off] + text          # model: the second 256-byte region begins `off` bytes into the CRs
        starts, deferred = fm.flat_split_chunked_tekken(doc, [0, len(doc)], region=256)
        if not deferred:
            assert starts == tk_oracle.split_tekken(doc), off
        doc = (b"ab cd\n" * 400)[:1952 - 32 - cr0 - off] + text              # kernel: the second 2048-byte region
        _emu_check_json(test_vocab, [doc], False, False)


def test_emu_flat_json_pattern(test_vocab):
    """The flat kernel's JSON-pattern instantiation on the emulator, id for id against the oracle in that mode: case
    mixes and digits on the fast path, CJK / marks handed back to the sequential matcher, long runs, region geometry."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "split_vectors_tekken.json")) as f:
        g = json.load(f)
    docs = [c["text"].encode("utf-8") for c in g["cases"][:500]]
    flagged = _emu_check_json(test_vocab, docs)
    assert len(flagged) < len(docs) // 10          # (only the very long runs / pieces)
    rng = random.Random(6)
    alpha = list("aAbBzZxyQ") + [" "] * 5 + ["1", "2", "!", "/", "-", "'", "\n", "\r", "\t", "é", "É", "Ж", "ж", "٣", "　", ".", ",",
                                               "中", "文", "ʰ", "ª", "́", "̈", "ǅ"]
    docs = ["".join(rng.choice(alpha) * rng.choice([1, 1, 1, 2, 3, 40]) for _ in range(rng.randint(0, rng.choice([4, 60, 900])))).encode()
            for _ in range(150)]
    docs += [("x y " * (pad // 4) + "q" * (pad % 4) + ch * rl + " Zz").encode() for pad in (1888, 1920, 1951, 1952, 1984)
             for ch in ("\n", "/", " ", "a", "A", "!") for rl in (31, 33, 65)]
    d, o = corpus.generate("ascii", 30, 512, seed=corpus.BASE_SEED + 1)
    docs += corpus.docs_of(d, o)
    flagged = _emu_check_json(test_vocab, docs)
    assert len(flagged) < len(docs) // 2


def test_pair_filter_has_no_false_negatives(test_vocab, small_vocab):
    """The merge kernels take a clear bit of the PAIR filter (csrc/tk_hash.h) as proof that a pair is in no bucket and
    skip the probe: every stored pair must have its bit set, for the table as built and as reloaded from the cache
    (emu_table_info / emu_table_cache_roundtrip check it pair by pair and compare the two filters)."""
    import tempfile
    for v in (test_vocab, small_vocab):
        info = emu.table_info(v["tokens"], v["num_special"])
        assert info["pair_filter_set_bits"] <= info["pairs"] and (info["pairs"] == 0) == (info["pair_filter_set_bits"] == 0)
        assert info["pair_filter_bits"] >= 1 << 15
    assert emu.table_info(test_vocab["tokens"], test_vocab["num_special"])["pairs"] > 0
    with tempfile.TemporaryDirectory() as d:
        emu.table_cache_roundtrip(test_vocab["tokens"], test_vocab["num_special"], d + "/t.bin")


def test_emu_memo_of_merged_pieces(test_vocab, small_vocab):
    """The memo of merged pieces (csrc/tk_hash.h MEMO): pieces of 2..16 bytes that are no vocabulary key are looked up in a table
    the narrow merge kernel fills, from the next call on.  Every call -- table empty, half filled, tiny and thrashing, hit by
    fragments of cut pieces, under the other vocabulary's leftovers being impossible by construction (one table per vocabulary)
    -- must give the oracle's ids; and the second pass over the same text must actually hit."""
    rng = random.Random(11)
    words = ["zyqx", "Qwrtzu", "blorft", "xx", "Zz", "quuxly", "vvvvvv", "aeiouaeiou", "snorkelwhack", "pneumonoultra", "ZYXWVUTSRQPONMLK"]
    def text(n):
        out = []
        for _ in range(n):
            w = rng.choice(words) if rng.random() < 0.5 else "".join(rng.choice("abcxyzQZ") for _ in range(rng.randint(2, 18)))
            out.append(w)
        return (" ".join(out)).encode()
    docs_a = [text(rng.randint(0, 60)) for _ in range(30)] + list(helpers.EDGE_DOCS) + [b"x" * 300 + b" zyqx blorft", ("qz" * 80).encode()]
    docs_b = [text(rng.randint(0, 60)) for _ in range(30)] + docs_a[:10]
    d, o = corpus.generate("ascii", 30, 512, seed=corpus.BASE_SEED + 1)
    docs_c = corpus.docs_of(d, o)
    try:
        for v in (test_vocab, small_vocab):
            for log2 in (12, 4):                 # roomy, and 16 entries (every insert evicts something)
                emu.memo_set(log2)
                seen_hits = 0
                for docs in (docs_a, docs_a, docs_b, docs_c, docs_a, docs_c):
                    _emu_check(v, docs, check_split=False)
                    seen_hits += emu.memo_info()["hits_last"]
                info = emu.memo_info()
                assert info["calls"] == 6 and info["entries"] > 0
                assert seen_hits > 0, "no call ever hit the memo"
                _emu_check(v, docs_a, False, False, check_split=False)
        # a hit is reported only for what the table holds: first call 0, an identical second call > 0
        emu.memo_set(14)
        _emu_check(test_vocab, docs_a, check_split=False)
        info = emu.memo_info()
        assert info["hits_last"] == 0
        # the call that fills an empty table does not log a word again whose slot already carries a claim of this call (the hot words
        # would fill the log with copies of themselves; only the lanes of ONE group still see each other's word too late): far fewer
        # records than memo-sized pieces that missed, and at least one record in two wins its slot
        import numpy as np
        packed = np.frombuffer(b"".join(docs_a), np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(d) for d in docs_a])]).astype(np.uint64)
        rec = helpers.oracle_for(test_vocab).miss_records(packed, offs)
        eligible = int(((rec[:, 1] >= 2) & (rec[:, 1] <= 16) & (rec[:, 2] <= 5)).sum())
        assert 20 < info["logged_last"] <= eligible * 3 // 4 and info["won_last"] * 2 >= info["logged_last"], (info, eligible)
        _emu_check(test_vocab, docs_a, check_split=False)
        assert emu.memo_info()["hits_last"] > 0
    finally:
        emu.memo_set(0)


def test_memo_entry_packing_round_trip():
    """tk_memo_pack / tk_memo_id / tk_memo_n / tk_memo_len (csrc/tk_hash.h): five 21-bit ranks, the length and the count survive the 32-byte
    entry's packing for every count and at the extremes of every field; the fifth rank shares its word with the tag bit that tells
    a committed entry from a claim (a log-record index)."""
    import ctypes
    import numpy as np
    L = emu.lib()
    L.emu_memo_pack_roundtrip.restype = ctypes.c_int
    L.emu_memo_pack_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    rng = random.Random(3)
    top = (1 << 21) - 1
    for _ in range(4000):
        n = rng.randint(1, 5)
        r = [rng.choice([0, 1, top, top - 1, rng.randint(0, top)]) for _ in range(n)] + [0] * (5 - n)
        a = np.array(r, np.uint32)
        assert L.emu_memo_pack_roundtrip(a.ctypes.data, n, rng.randint(2, 16)) == 0, (r, n)
