#!/usr/bin/env python3
"""Seeded synthetic tekken.json (the real tests/assets/tekken.json is absent from the reference
mount).  Bench/test infrastructure.

A small byte-level BPE trainer is run over pieces of the synthetic corpora (pieces obtained with
the oracle's split, i.e. the hard-coded pattern of reference src/tekkenizer.rs:123), then the
vocabulary is grown to the requested size with tokens that are concatenations of two existing
tokens (so that, as in a trained vocabulary, every token has a valid split), and the result is
written in the on-disk schema of reference src/config.rs:16-82:
ranks 0..255 are the single bytes, `default_vocab_size` = n_ranks + `default_num_special_tokens`.
"""
import argparse
import base64
import collections
import heapq
import json
import os
import random
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

MISTRAL_PATTERN = (r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+|"
                   r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n/]*|"
                   r"\s*[\r\n]+|\s+(?!\S)|\s+")

SPECIAL_NAMES = ["<unk>", "<s>", "</s>", "[INST]", "[/INST]", "[AVAILABLE_TOOLS]", "[/AVAILABLE_TOOLS]",
                 "[TOOL_RESULTS]", "[/TOOL_RESULTS]", "[TOOL_CALLS]", "[IMG]", "<pad>", "[IMG_BREAK]", "[IMG_END]",
                 "[PREFIX]", "[MIDDLE]", "[SUFFIX]", "[SYSTEM_PROMPT]", "[/SYSTEM_PROMPT]", "[TOOL_CONTENT]"]


def heldout_words():
    """The words of the generator's 4 096-word list that the HELD-OUT vocabulary never sees: every fourth word from index 19
    on (about 15 % of the word occurrences of G-ascii by Zipf mass).  bench.py --vocab-fit heldout tokenizes the ordinary
    corpus with that vocabulary: its words routinely take several tokens, like `tokenizer`, `decoding`, `comparison` in the
    reference's own vectors (tests/test_tokenizer_output.rs:217, 251, 268), and the merge kernels see a realistic miss rate
    instead of the 2.4 % of a vocabulary trained on the very word list the corpus draws from."""
    import corpus
    return {w for i, w in enumerate(corpus.words()) if i >= 16 and i % 4 == 3}


def piece_counts(n_ascii_docs, n_mixed_docs, seed, heldout=False):
    import corpus
    import tk_oracle
    cnt = collections.Counter()
    held = heldout_words() if heldout else set()
    for kind, n, dl, sd in (("ascii", n_ascii_docs, 512, seed + 101), ("mixed", n_mixed_docs, 2048, seed + 102)):
        if n == 0:
            continue
        data, offs = corpus.generate(kind, n, dl, seed=sd)
        raw = data.tobytes()
        for d in range(n):
            doc = raw[int(offs[d]):int(offs[d + 1])]
            pieces = tk_oracle.split_pieces(doc)
            if held:
                pieces = [p for p in pieces if bytes(c for c in p.lower() if 97 <= c <= 122).decode() not in held]
            cnt.update(pieces)
    return cnt


def train_bpe(cnt, n_merges, min_count=2):
    """Classic BPE over a piece-frequency table.  Returns the merged tokens (bytes) in merge order."""
    words = []   # list of lists of token ids
    freqs = []
    for piece, f in cnt.items():
        if len(piece) < 2:
            continue
        words.append(list(piece))
        freqs.append(f)
    tokens = [bytes([i]) for i in range(256)]
    pair_cnt = collections.defaultdict(int)
    pair_words = collections.defaultdict(set)
    for wi, w in enumerate(words):
        f = freqs[wi]
        for a, b in zip(w, w[1:]):
            pair_cnt[(a, b)] += f
            pair_words[(a, b)].add(wi)
    heap = [(-c, p) for p, c in pair_cnt.items()]
    heapq.heapify(heap)
    merges = []
    have = set(tokens)
    while len(merges) < n_merges and heap:
        negc, p = heapq.heappop(heap)
        c = pair_cnt.get(p, 0)
        if c != -negc:
            if c > 0:
                heapq.heappush(heap, (-c, p))
            continue
        if c < min_count:
            break
        new_bytes = tokens[p[0]] + tokens[p[1]]
        if new_bytes in have:
            # the same byte string was already produced through another split; skip this pair
            pair_cnt[p] = 0
            continue
        new_id = len(tokens)
        tokens.append(new_bytes)
        have.add(new_bytes)
        merges.append(new_bytes)
        touched = set()
        for wi in list(pair_words[p]):
            w = words[wi]
            f = freqs[wi]
            i = 0
            changed = False
            while i < len(w) - 1:
                if w[i] == p[0] and w[i + 1] == p[1]:
                    if i > 0:
                        q = (w[i - 1], w[i])
                        pair_cnt[q] -= f
                        touched.add(q)
                    if i + 2 < len(w):
                        q = (w[i + 1], w[i + 2])
                        pair_cnt[q] -= f
                        touched.add(q)
                    w[i:i + 2] = [new_id]
                    if i > 0:
                        q = (w[i - 1], new_id)
                        pair_cnt[q] += f
                        pair_words[q].add(wi)
                        touched.add(q)
                    if i + 1 < len(w):
                        q = (new_id, w[i + 1])
                        pair_cnt[q] += f
                        pair_words[q].add(wi)
                        touched.add(q)
                    changed = True
                else:
                    i += 1
            if changed:
                pass
        pair_cnt[p] = 0
        pair_words.pop(p, None)
        for q in touched:
            cq = pair_cnt.get(q, 0)
            if cq > 0:
                heapq.heappush(heap, (-cq, q))
    return merges


def grow(tokens, n_ranks, rng):
    """Append concatenations of two existing tokens until n_ranks tokens exist."""
    have = set(tokens)
    pool = list(tokens[256:]) or list(tokens)
    while len(tokens) < n_ranks:
        a = pool[rng.randrange(len(pool))] if rng.random() < 0.7 else tokens[rng.randrange(256)]
        b = pool[rng.randrange(len(pool))] if rng.random() < 0.7 else tokens[rng.randrange(256)]
        t = a + b
        if len(t) > 24 or t in have:
            continue
        have.add(t)
        tokens.append(t)
        if len(pool) < 200000:
            pool.append(t)
    return tokens


def build_tokens(n_ranks, n_merges, n_ascii_docs, n_mixed_docs, seed, heldout=False):
    cnt = piece_counts(n_ascii_docs, n_mixed_docs, seed, heldout)
    merges = train_bpe(cnt, min(n_merges, max(0, n_ranks - 256)))
    tokens = [bytes([i]) for i in range(256)] + merges
    tokens = grow(tokens, n_ranks, random.Random(seed))
    return tokens[:n_ranks]


def model_json(tokens, num_special=1000, version="v7", extra_specials=("[AUDIO]", "[BEGIN_AUDIO]")):
    vocab = []
    for r, t in enumerate(tokens):
        try:
            s = t.decode("utf-8")
        except UnicodeDecodeError:
            s = None
        vocab.append({"rank": r, "token_bytes": base64.b64encode(t).decode("ascii"), "token_str": s})
    specials = [{"rank": i, "token_str": n, "is_control": True} for i, n in enumerate(SPECIAL_NAMES)]
    for k, n in enumerate(extra_specials):
        specials.append({"rank": 24 + k, "token_str": n, "is_control": True})
    return {"config": {"pattern": MISTRAL_PATTERN, "num_vocab_tokens": len(tokens),
                       "default_vocab_size": len(tokens) + num_special, "default_num_special_tokens": num_special,
                       "version": version},
            "vocab": vocab, "special_tokens": specials}


def write_json(path, tokens, num_special=1000):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        json.dump(model_json(tokens, num_special), f, ensure_ascii=True)
    os.replace(tmp, path)


def default_path():
    return os.path.join(ROOT, "assets", "synth_tekken.json")


def ensure_default(path=None, n_ranks=130072, n_merges=40000, n_ascii_docs=20000, n_mixed_docs=3000, seed=0x7E44E2):
    """The bench vocabulary: same size class as the reference's test asset (131072 ids, 1000 specials)."""
    path = path or default_path()
    if not os.path.exists(path):
        write_json(path, build_tokens(n_ranks, n_merges, n_ascii_docs, n_mixed_docs, seed))
    return path


def heldout_path():
    return os.path.join(ROOT, "assets", "synth_tekken_heldout.json")


def ensure_heldout(path=None, n_ranks=130072, n_merges=40000, n_ascii_docs=20000, n_mixed_docs=3000, seed=0x7E44E2):
    """The same construction with heldout_words() withheld from the training pieces."""
    path = path or heldout_path()
    if not os.path.exists(path):
        write_json(path, build_tokens(n_ranks, n_merges, n_ascii_docs, n_mixed_docs, seed, heldout=True))
    return path


def load_tokens(path):
    """(tokens by rank, num_special, bos_id, eos_id) of a tekken.json, applying the reference's truncation
    (src/tekkenizer.rs:118-119,780-784)."""
    with open(path) as f:
        m = json.load(f)
    ns = m["config"]["default_num_special_tokens"]
    inner = m["config"]["default_vocab_size"] - ns
    toks = [base64.b64decode(e["token_bytes"]) for e in m["vocab"][:inner]]
    sp = {e["token_str"]: e["rank"] for e in (m.get("special_tokens") or [])}
    return toks, ns, sp.get("<s>", 1), sp.get("</s>", 2)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=default_path())
    ap.add_argument("--ranks", type=int, default=130072)
    ap.add_argument("--merges", type=int, default=40000)
    ap.add_argument("--ascii-docs", type=int, default=20000)
    ap.add_argument("--mixed-docs", type=int, default=3000)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x7E44E2)
    a = ap.parse_args()
    toks = build_tokens(a.ranks, a.merges, a.ascii_docs, a.mixed_docs, a.seed)
    write_json(a.out, toks)
    print("wrote %s: %d ranks" % (a.out, len(toks)))
