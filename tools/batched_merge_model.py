#!/usr/bin/env python3
"""Model of the ROUND-BASED byte-pair merge used for long single pieces (csrc/tk_long.hip), checked against the
one-merge-per-step algorithm (tiktoken's _byte_pair_merge: leftmost minimum rank).

One step of the sequential algorithm merges the leftmost pair of minimum rank r*.  A ROUND merges, in one parallel
sweep, every occurrence the sequential algorithm would merge before it touches any other rank:

  1. r* = the global minimum pair rank; candidates = all positions i whose pair (part i, part i+1) has rank r*;
  2. overlapping candidates: in a maximal run of CONSECUTIVE candidates only those at an even offset are merged (the
     sequential algorithm takes the leftmost, which destroys the pair to its right, then the next one two further on);
  3. exactness: a merge creates two new pairs -- (left neighbour AT THAT MOMENT, merged) and (merged, right neighbour at
     that moment).  Going left to right, the left neighbour is the merged token of the previous occurrence if that one
     ends right before this one, the right neighbour is still unmerged.  If one of those pairs has a rank BELOW r* the
     sequential algorithm would merge it next, before the remaining occurrences of r* -- the round is then cut after the
     first such occurrence (it and everything to its left is committed; the next round starts from the new minimum);
  4. commit: parts are compacted, the new pair ranks are the ones probed in 3 (for two occurrences back to back the pair
     between them is (merged, merged)).

Not a parallel algorithm by itself: the model processes plain Python lists; what it pins is that the ROUND STRUCTURE
gives the sequential result on any vocabulary, including ones where step 3 fires (tokens whose halves have higher ranks).
"""
import random

MAX = 1 << 62


def sequential(ranks, piece):
    """tiktoken's loop, list-of-parts form: returns the part byte strings."""
    parts = [piece[i:i + 1] for i in range(len(piece))]
    while len(parts) > 1:
        best, where = None, None
        for i in range(len(parts) - 1):
            r = ranks.get(parts[i] + parts[i + 1])
            if r is not None and (best is None or r < best):
                best, where = r, i
        if where is None:
            break
        parts[where:where + 2] = [parts[where] + parts[where + 1]]
    return parts


def rounds(ranks, piece, stats=None):
    """The round-based form.  parts are byte strings (on the device: token ids; the pair rank comes from PAIR)."""
    parts = [piece[i:i + 1] for i in range(len(piece))]

    def rk(a, b):
        return ranks.get(a + b, MAX)

    pair = [rk(parts[i], parts[i + 1]) for i in range(len(parts) - 1)] + [MAX]
    n_rounds = n_cut = 0
    while True:
        n = len(parts)
        r_star = min(pair) if pair else MAX
        if r_star >= MAX:
            break
        n_rounds += 1
        cand = [pair[i] == r_star for i in range(n)]
        # even offsets inside runs of consecutive candidates
        sel = [False] * n
        off = 0
        for i in range(n):
            if cand[i]:
                sel[i] = off % 2 == 0
                off += 1
            else:
                off = 0
        # probes + undercut test, every occurrence on its own (this is the parallel part)
        left_rank = [MAX] * n
        right_rank = [MAX] * n
        under = [False] * n
        for i in range(n):
            if not sel[i]:
                continue
            merged = parts[i] + parts[i + 1]
            if i > 0:
                left = (parts[i - 2] + parts[i - 1]) if (i >= 2 and sel[i - 2]) else parts[i - 1]
                left_rank[i] = rk(left, merged)
            if i + 2 < n:
                right_rank[i] = rk(merged, parts[i + 2])
            under[i] = left_rank[i] < r_star or right_rank[i] < r_star
        if any(under):
            u = under.index(True)
            n_cut += 1
            for i in range(u + 1, n):
                sel[i] = False
        # commit + compact
        new_parts, new_pair = [], []
        i = 0
        while i < n:
            if sel[i]:
                merged = parts[i] + parts[i + 1]
                if new_pair:
                    new_pair[-1] = left_rank[i]
                new_parts.append(merged)
                if i + 2 < n and sel[i + 2]:
                    new_pair.append(MAX)          # written by the next occurrence (its left_rank)
                else:
                    new_pair.append(right_rank[i])
                i += 2
            else:
                new_parts.append(parts[i])
                new_pair.append(pair[i])
                i += 1
        parts, pair = new_parts, new_pair
        pair[-1] = MAX
    if stats is not None:
        stats["rounds"] = stats.get("rounds", 0) + n_rounds
        stats["cut_rounds"] = stats.get("cut_rounds", 0) + n_cut
    return parts


def rounds_heads(ranks, piece, stats=None):
    """The LAZY form (csrc/tk_long.hip, tk_long_sparse_kernel): parts are never compacted -- slot i keeps its place, an
    alive flag says whether a part starts there -- and a round merges only the HEADS of the candidate runs (a candidate whose
    predecessor is not a candidate), one lane each, without any run-parity bookkeeping.  Exactness needs two cuts:
      * the undercut cut of `rounds` (a created pair below r*: nothing to the right of that occurrence merges this round);
      * the CHAIN cut: in a run of three or more consecutive candidates the sequential order also merges the member at
        offset 2 in this "round"; the lazy form leaves it for the next one, so nothing to the RIGHT of the leftmost such
        member may merge before it (its merge could create a pair that undercuts and reaches into what lies to its right).
    A run of k candidates therefore takes ~k / 2 rounds: the form for pieces with many distinct pairs, where runs are rare."""
    n = len(piece)
    tok = [piece[i:i + 1] for i in range(n)]
    alive = [True] * n

    def rk(a, b):
        return ranks.get(a + b, MAX)

    def nxt(i):
        i += 1
        while i < n and not alive[i]:
            i += 1
        return i

    def prv(i):
        i -= 1
        while i >= 0 and not alive[i]:
            i -= 1
        return i

    pair = [rk(tok[i], tok[i + 1]) if i + 1 < n else MAX for i in range(n)]
    n_rounds = n_chain = n_cut = 0
    while True:
        live = [i for i in range(n) if alive[i]]
        r_star = min((pair[i] for i in live), default=MAX)
        if r_star >= MAX:
            break
        n_rounds += 1
        cand = {i for i in live if pair[i] == r_star}
        heads = [i for i in sorted(cand) if prv(i) not in cand]
        # chain cut: leftmost candidate whose two predecessors are candidates too
        z = min((i for i in cand if prv(i) in cand and prv(prv(i)) in cand), default=n)
        if z < n:
            n_chain += 1
        heads = [i for i in heads if i < z]
        headset = set(heads)
        res = {}
        under = []
        for i in heads:                                   # the parallel part: one lane per head
            j = nxt(i)
            k = nxt(j)
            p = prv(i)
            merged = tok[i] + tok[j]
            lr = rr = MAX
            left_slot = -1
            if p >= 0:
                pp = prv(p)
                if pp >= 0 and pp in headset and nxt(pp) == p:      # p is consumed by the occurrence before it
                    left, left_slot = tok[pp] + tok[p], pp
                else:
                    left, left_slot = tok[p], p
                lr = rk(left, merged)
            if k < n:
                rr = rk(merged, tok[k])
            res[i] = (j, k, left_slot, lr, rr, merged)
            if lr < r_star or rr < r_star:
                under.append(i)
        if under:
            n_cut += 1
            u = min(under)
            heads = [i for i in heads if i <= u]
            headset = set(heads)
        for i in heads:
            j, k, left_slot, lr, rr, merged = res[i]
            tok[i] = merged
            alive[j] = False
            pair[j] = MAX
            if not (k < n and k in headset):              # the pair behind me: mine unless the next occurrence starts right there
                pair[i] = rr
            if left_slot >= 0:
                pair[left_slot] = lr
    if stats is not None:
        stats["rounds"] = stats.get("rounds", 0) + n_rounds
        stats["chain_rounds"] = stats.get("chain_rounds", 0) + n_chain
        stats["cut_rounds"] = stats.get("cut_rounds", 0) + n_cut
    return [tok[i] for i in range(n) if alive[i]]


def _random_vocab(rng, alphabet, n_extra, max_len):
    toks = {bytes([b]): b for b in range(256)}
    words = []
    possible = sum(len(alphabet) ** k for k in range(2, max_len + 1))
    n_extra = min(n_extra, possible)
    while len(words) < n_extra:
        t = "".join(rng.choice(alphabet) for _ in range(rng.randint(2, max_len))).encode()
        if t not in toks:
            toks[t] = 0
            words.append(t)
    rng.shuffle(words)
    for r, t in enumerate(words):
        toks[t] = 256 + r
    return toks


def self_check(n_vocabs=40, n_pieces=60, seed=5):
    rng = random.Random(seed)
    stats, stats2 = {}, {}
    checked = 0
    for v in range(n_vocabs):
        alphabet = rng.choice(["ab", "abc", "a", "abcd", "ab"])
        ranks = _random_vocab(rng, alphabet, rng.randint(3, 120), rng.randint(2, 6))
        for _ in range(n_pieces):
            n = rng.choice([1, 2, 3, 5, 8, 13, 30, 64, 65, 130])
            piece = "".join(rng.choice(alphabet) for _ in range(n)).encode()
            a, b = sequential(ranks, piece), rounds(ranks, piece, stats)
            assert a == b, (alphabet, piece, a, b)
            c = rounds_heads(ranks, piece, stats2)
            assert a == c, ("lazy form", alphabet, piece, a, c)
            checked += 1
    stats["lazy"] = stats2
    return checked, stats


if __name__ == "__main__":
    c, st = self_check()
    print("round-based merge == sequential merge on %d pieces; rounds %d, of which cut short by an undercutting pair %d"
          % (c, st["rounds"], st["cut_rounds"]))
    lz = st["lazy"]
    print("lazy (heads-only) form == sequential merge on the same pieces; rounds %d, chain cuts %d, undercut cuts %d"
          % (lz["rounds"], lz["chain_rounds"], lz["cut_rounds"]))
