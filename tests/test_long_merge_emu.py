"""The round-based WORKGROUP merges of one long piece (csrc/tk_long_impl.h: compacting rounds tkl_block_merge, lazy rounds
tks_block_merge) on the CPU emulator: 16 emulated waves, the workgroup barrier as a scheduling point, scratch / output between
guard words -- and, through tests/test_sanitizers.py, the same under AddressSanitizer / UBSan.  Their index arithmetic (the
sel-word carry ripple across waves, the u16 occurrence lists, Blr / Brr scratch indexing, the alive-bit updates) can silently
corrupt LDS or scratch on the GPU, where no sanitizer runs; before round 3 only a Python model of the round STRUCTURE
(tools/batched_merge_model.py) and one GPU test stood behind them.

Expected ids: the oracle on a piece that is one `\\p{L}+` match and longer than every token (no whole-piece look-up can hit),
i.e. the pure sequential merge.  TK_EMU_LONG_FULL=1 adds the 32 767- / 32 768-byte pieces (minutes each)."""
import os
import random

import pytest

import emu
import gen_golden_merge as gg
import tk_oracle


def _check(toks, ns, piece, kinds=(0, 1)):
    assert len(piece) > max(len(t) for t in toks)
    want = tk_oracle.Oracle(toks, ns, 1, 2).encode(piece, False, False)
    for kind in kinds:
        got = emu.long_merge(toks, ns, piece, kind)
        assert got == want, (kind, len(piece), piece[:40], got[:12], want[:12])


def test_block_merges_on_adversarial_vocabularies():
    """Random multi-byte tokens in random rank order over tiny alphabets (created pairs that undercut: the round is cut; runs of
    equal pairs: even offsets across step and wave boundaries), lengths around every boundary of the layout: one step (64),
    one part per thread (1024), two steps per wave (2048)."""
    rng = random.Random(9)
    for alphabet, n_extra, max_len, sizes in (("ab", 60, 6, (65, 128, 1025)), ("abc", 250, 5, (66, 129, 1024)),
                                              ("abcdefgh", 300, 5, (65, 1023, 2049, 4100))):
        toks = gg.vocab_adversarial(rng, alphabet, n_extra, max_len)
        if os.environ.get("TK_TEST_SANITIZE"):      # (the sanitizer leg of tests/test_sanitizers.py: three times slower)
            if alphabet == "abcdefgh":
                continue
            sizes = sizes[1:2] + sizes[-1:]
        for n in sizes:
            _check(toks, 5, "".join(rng.choice(alphabet) for _ in range(n)).encode())


def test_block_merges_on_repetitive_pieces_and_the_full_list():
    """One letter, short periods (every pair a candidate: the parity of a run that crosses a wave boundary), and a piece whose
    minimum rank has more than TKS_CAP = 512 isolated occurrences inside one wave's 2 048 slots (the lazy form's list is
    full: the round is cut at the first head left out).  (The lazy form merges one head per run and round: on one-letter
    pieces it needs a round per merge, which the emulator pays with a thousand fiber switches each -- short pieces only.)"""
    rng = random.Random(10)
    toks = [bytes([i]) for i in range(256)] + [b"ab", b"aa", b"abc", b"aaa", b"ca", b"bca", b"aaaa", b"abab", b"cab"]
    for piece in (b"a" * 1000, b"a" * 4097, b"ab" * 1500, b"abc" * 2800, b"aab" * 700 + b"b" * 70, b"abc" * 683 + b"a" * 3000):
        _check(toks, 5, piece, kinds=(0,))
    for piece in (b"a" * 300, b"ab" * 400, b"aab" * 100 + b"b" * 70, b"abc" * 683 + b"a" * 100):
        _check(toks, 5, piece, kinds=(1,))
    toks2 = gg.vocab_adversarial(rng, "abc", 90, 4)
    _check(toks2, 5, b"abc" * 2731)            # 8 193 bytes: four steps per wave, 683 heads per 2 048 slots


def test_block_merges_trained_vocabulary(test_vocab):
    rng = random.Random(11)
    for n in (300, 3000):
        _check(test_vocab["tokens"], test_vocab["num_special"], bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(n)))


@pytest.mark.skipif(not os.environ.get("TK_EMU_LONG_FULL"), reason="minutes per piece: TK_EMU_LONG_FULL=1 (last run clean: both forms, both lengths)")
def test_block_merges_at_the_maximum_length():
    rng = random.Random(4)
    toks = gg.vocab_adversarial(rng, "abc", 250, 5)
    for n, kind in ((32768, 0), (32767, 1), (32767, 0), (32768, 1)):
        _check(toks, 5, "".join(rng.choice("abc") for _ in range(n)).encode(), kinds=(kind,))
