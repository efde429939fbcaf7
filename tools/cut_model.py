#!/usr/bin/env python3
"""Model of the EXACT CUT DECOMPOSITION of a missed piece (csrc/tk_flat_impl.h, CUT instantiation of the flat kernel).

tiktoken's merge loop (the engine behind reference src/tekkenizer.rs:384-386) joins two adjacent parts
only if their concatenated BYTES are a vocabulary key.  A part that spans the boundary i | i+1 of a piece therefore is a
key whose bytes contain b[i] b[i+1] at that place.  If the vocabulary rules that out, no merge can ever span the
boundary, the pair (part ending at i, part starting at i+1) has rank MAX for the whole run of the loop, and the loop's
leftmost-minimum order restricted to either side is that side's own order: the piece decomposes into sub-pieces that
merge independently.  Rules (all pure functions of the bytes around the boundary, so that two chunks of the flat path
decide alike):

  G2     cut iff the bigram b[i] b[i+1] occurs inside NO token (8 KB bitmap);
  K2G3   cut iff the bigram is not itself a KEY and neither trigram b[i-1..i+1], b[i..i+2] occurs inside any token
         (a spanning key of 2 bytes is the bigram; one of >= 3 bytes contains one of the two trigrams).  Windows that
         reach outside the piece are evaluated on the raw text: that can only suppress a cut, never add one.
  Kk     the same idea one level up: keys of 2..k-1 bytes that span the boundary, k-grams inside tokens.

Two facts the device code relies on, both checked here:
  * a piece that IS a key has no cut inside (every n-gram of it occurs in a token: itself), so "has a cut" implies
    "misses the vocabulary" and the whole-piece shortcut needs no look-up for a cut piece;
  * sub-pieces go through the PURE merge, never through the whole-piece shortcut (parity trap T3): a sub-piece that
    happens to be a key is not necessarily what merging its bytes gives.

Run: python tools/cut_model.py  (self-check on adversarial vocabularies + statistics on the bench vocabulary).
"""
import random
import sys

from batched_merge_model import sequential, _random_vocab


def ngrams_inside(ranks, k):
    s = set()
    for t in ranks:
        for i in range(len(t) - k + 1):
            s.add(t[i:i + k])
    return s


class CutRule:
    def __init__(self, ranks, k):
        """k = 2: G2.  k >= 3: keys of 2..k-1 bytes + k-grams inside tokens."""
        self.k = k
        self.keys = ranks
        self.grams = ngrams_inside(ranks, k)

    def cut(self, text, i):
        """True: no part can span the boundary between text[i-1] and text[i] (0 < i < len(text))."""
        k = self.k
        for m in range(2, k):          # a spanning key of m < k bytes: text[s:s+m] with s < i < s+m
            for s in range(i - m + 1, i):
                if s >= 0 and s + m <= len(text) and text[s:s + m] in self.keys:
                    return False
        for s in range(i - k + 1, i):  # a spanning key of >= k bytes contains one of these windows
            if s >= 0 and s + k <= len(text) and text[s:s + k] in self.grams:
                return False
        return True


def encode_cut(ranks, piece, rule, context=(b"", b"")):
    """encode_piece through the cut decomposition.  context = raw text left / right of the piece (windows may reach into it)."""
    left, right = context
    text = left + piece + right
    o = len(left)
    cuts = [i for i in range(1, len(piece)) if rule.cut(text, o + i)]
    if not cuts:
        if piece in ranks:
            return [piece]
        return sequential(ranks, piece)
    assert piece not in ranks, "a key has no cut inside"
    out = []
    for a, b in zip([0] + cuts, cuts + [len(piece)]):
        out += sequential(ranks, piece[a:b])     # pure merge: no shortcut for the fragment
    return out


def encode_ref(ranks, piece):
    if piece in ranks:
        return [piece]
    return sequential(ranks, piece)


def self_check(n_vocabs=60, n_pieces=80, seed=11):
    rng = random.Random(seed)
    checked = n_cut = 0
    for v in range(n_vocabs):
        alphabet = rng.choice(["ab", "abc", "abcd", "abcdef", "abcdefgh"])
        ranks = _random_vocab(rng, alphabet, rng.randint(3, 200), rng.randint(2, 7))
        rules = [CutRule(ranks, k) for k in (2, 3, 4)]
        for _ in range(n_pieces):
            n = rng.choice([1, 2, 3, 5, 8, 13, 30, 64, 65, 130, 300])
            piece = "".join(rng.choice(alphabet) for _ in range(n)).encode()
            ctx = ("".join(rng.choice(alphabet) for _ in range(3)).encode(), "".join(rng.choice(alphabet) for _ in range(3)).encode())
            want = encode_ref(ranks, piece)
            for r in rules:
                for c in ((b"", b""), ctx):
                    got = encode_cut(ranks, piece, r, c)
                    assert got == want, (alphabet, r.k, piece, want, got)
            n_cut += sum(1 for i in range(1, n) if rules[0].cut(piece, i))
            checked += 1
    return checked, n_cut


def bench_stats():
    import synth_vocab
    toks, ns, bos, eos = synth_vocab.load_tokens(synth_vocab.ensure_default())
    ranks = {t: i for i, t in enumerate(toks)}
    rng = random.Random(3)
    piece = bytes(rng.randrange(26) + 97 for _ in range(32768))
    for k in (2, 3, 4):
        rule = CutRule(ranks, k)
        cuts = [0] + [i for i in range(1, len(piece)) if rule.cut(piece, i)] + [len(piece)]
        lens = [b - a for a, b in zip(cuts, cuts[1:])]
        print("k = %d on a 32 KiB random-letter piece: %d sub-pieces, longest %d bytes, mean %.1f; > 64 bytes: %d, > 16: %d"
              % (k, len(lens), max(lens), sum(lens) / len(lens), sum(1 for x in lens if x > 64), sum(1 for x in lens if x > 16)))


if __name__ == "__main__":
    c, n = self_check()
    print("cut decomposition == sequential merge on %d pieces (rules k = 2, 3, 4, with and without context; %d cuts under G2)" % (c, n))
    if len(sys.argv) > 1 and sys.argv[1] == "stats":
        bench_stats()
