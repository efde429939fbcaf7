#!/usr/bin/env python3
"""tools/gen_golden_merge.py -- golden vectors for the byte-pair MERGE loop, from an independent restatement.

INDEPENDENT RESTATEMENT, NOT THE REFERENCE: the reference (tekken-rs) delegates the merge loop to the crate
tiktoken-rs ("0.7.0", reference Cargo.toml:40), whose source is not under /root/reference and cannot be built here.
This script is a second, separately written transcription of tiktoken's PUBLISHED algorithm in its list-of-parts
form -- the piece is a Python list of byte strings, the lowest-ranked adjacent concatenation (leftmost on ties) is
joined, until no adjacent concatenation is a key -- deliberately NOT the (start, rank)-array form with sentinels that
oracle/tk_oracle.c restates.  Two independently written forms agreeing on adversarial vocabularies is what these
vectors add; they do not replace a run of the Rust reference (reference_cpu/ closes that loop on a box with cargo).

It shares no code with oracle/, tekken-rs_amd/ or tools/synth_vocab.py (own vocabulary builders, own PRNG use), and
splits whole texts with Python `regex` on the pattern literal of reference src/tekkenizer.rs:123 (independent engine).

Output: tests/golden/merge_vectors.json
  {"vocabs": [{"name", "num_special", "tokens_hex": [...by rank...],
               "pieces": [[piece_hex, [ids...]], ...],      one pre-token each (letters only), ids already shifted
               "texts":  [[text, [ids...]], ...]}]}         whole texts: split (regex) + shortcut + merge + shift
Run:  python tools/gen_golden_merge.py        (deterministic; rewrites the file)
"""
import json
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERN = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+"
           r"|\s+(?!\S)|\s+")


# ---------------------------------------------------------------------------------------------
# the algorithm, list-of-parts form
# ---------------------------------------------------------------------------------------------
def merge_parts(ranks, piece):
    """ranks: dict bytes -> rank.  Returns the ranks of the parts the piece ends up as."""
    parts = [piece[i:i + 1] for i in range(len(piece))]
    while len(parts) > 1:
        where, lowest = None, None
        for i in range(len(parts) - 1):
            r = ranks.get(parts[i] + parts[i + 1])
            if r is not None and (lowest is None or r < lowest):     # strict '<': the leftmost of equal ranks wins
                where, lowest = i, r
        if where is None:
            break
        parts[where:where + 2] = [parts[where] + parts[where + 1]]
    return [ranks[p] for p in parts]


def encode_piece(ranks, piece):
    whole = ranks.get(piece)            # the whole-piece shortcut comes BEFORE any merging
    if whole is not None:
        return [whole]
    return merge_parts(ranks, piece)


def encode_text(ranks, text, num_special):
    import regex
    out = []
    for m in regex.finditer(PATTERN, text):
        out.extend(r + num_special for r in encode_piece(ranks, m.group().encode("utf-8")))
    return out


# ---------------------------------------------------------------------------------------------
# vocabularies (rank i < 256 is the single byte i: reference src/tekkenizer.rs:793-798)
# ---------------------------------------------------------------------------------------------
def base():
    return [bytes([b]) for b in range(256)]


def vocab_trained(rng, sample, n_merges):
    """A tiny BPE trainer of its own: most frequent adjacent pair over the words of `sample`, ties by first seen."""
    import regex
    words = {}
    for m in regex.finditer(PATTERN, sample):
        w = m.group().encode("utf-8")
        words[w] = words.get(w, 0) + 1
    seqs = {w: [w[i:i + 1] for i in range(len(w))] for w in words}
    toks = base()
    have = set(toks)
    for _ in range(n_merges):
        cnt = {}
        for w, parts in seqs.items():
            for i in range(len(parts) - 1):
                k = (parts[i], parts[i + 1])
                cnt[k] = cnt.get(k, 0) + words[w]
        if not cnt:
            break
        best = max(cnt.items(), key=lambda kv: kv[1])[0]
        new = best[0] + best[1]
        if new in have:
            break
        toks.append(new)
        have.add(new)
        for w, parts in seqs.items():
            i, out = 0, []
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    out.append(new)
                    i += 2
                else:
                    out.append(parts[i])
                    i += 1
            seqs[w] = out
    return toks


def vocab_adversarial(rng, alphabet, n_extra, max_len):
    """Random multi-byte tokens in random rank order: tokens no merge sequence can reach (only the whole-piece shortcut
    finds them), runs of equal pairs (leftmost tie-break), long tokens whose halves are missing."""
    toks = base()
    have = set(toks)
    while len(toks) < 256 + n_extra:
        n = rng.randint(2, max_len)
        t = "".join(rng.choice(alphabet) for _ in range(n)).encode("utf-8")
        if t not in have:
            toks.append(t)
            have.add(t)
    return toks


def vocab_chains():
    """Hand-made: powers of one letter (ties + chains), nested prefixes, a high-rank pair that must lose."""
    toks = base()
    for t in (b"aa", b"aaaa", b"ab", b"ba", b"abab", b"bb", b"bbb", b"abb", b"aab", b"baab", b"aaaaaaaa", b"cab",
              b"ca", b"abc", b"bc", b"bcb", b"cbc", b"cc", b"ccc", b"cccc", b"ccccc", b"acc", b"cca"):
        toks.append(t)
    return toks


def rand_piece(rng, alphabet, n):
    return "".join(rng.choice(alphabet) for _ in range(n)).encode("utf-8")


def main():
    rng = random.Random(0x7E44E2 + 77)
    sample_words = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "token", "tokenizer", "izer",
                    "hello", "world", "Hello", "World", "merge", "merging", "rank", "ranks", "byte", "bytes", "pair",
                    "encoding", "decoding", "comparison", "compare", "special", "characters", "number", "numbers"]
    sample = " ".join(rng.choice(sample_words) for _ in range(4000)) + " 123, 456. (a+b) [x] {y} it's we're\n\nend"
    vocabs = []

    def add(name, toks, num_special, alphabets, lens, n_pieces, texts=()):
        assert len(set(toks)) == len(toks) and all(toks[b] == bytes([b]) for b in range(256))
        ranks = {t: i for i, t in enumerate(toks)}
        pieces = []
        seen = set()
        for _ in range(n_pieces):
            p = rand_piece(rng, rng.choice(alphabets), rng.choice(lens))
            if p in seen:
                continue
            seen.add(p)
            pieces.append([p.hex(), [r + num_special for r in encode_piece(ranks, p)]])
        # every token itself (the shortcut) and every token doubled / with one letter appended
        for t in toks[256:256 + 60]:
            for p in (t, t + t, t + b"a", b"b" + t):
                try:
                    s = p.decode("utf-8")
                except UnicodeDecodeError:
                    continue
                if p in seen or not s.isalpha():
                    continue
                seen.add(p)
                pieces.append([p.hex(), [r + num_special for r in encode_piece(ranks, p)]])
        tv = [[t, encode_text(ranks, t, num_special)] for t in texts]
        vocabs.append({"name": name, "num_special": num_special, "tokens_hex": [t.hex() for t in toks],
                       "pieces": pieces, "texts": tv})

    every_len = list(range(1, 41)) + [48, 63, 64, 65, 66, 80, 100, 129, 200]
    add("chains", vocab_chains(), 10, ["ab", "abc", "a", "c", "ac"], every_len, 420)
    add("adversarial_ab", vocab_adversarial(rng, "ab", 120, 7), 3, ["ab", "a", "b"], every_len, 420)
    add("adversarial_abcd", vocab_adversarial(rng, "abcd", 400, 6), 1000, ["abcd", "abc", "ad"], every_len, 420)
    add("adversarial_utf8", vocab_adversarial(rng, "aé中б", 300, 5), 100, ["aé中б", "é中", "aб"], list(range(1, 30)) + [40, 70], 300)
    texts = ["Hello, world!", "The quick brown fox jumps over the lazy dog.", "it's we're THEY'LL x'd",
             "tokenizer tokenizers untokenizable", "  leading and   trailing   ", "numbers: 123, 4567, 89.",
             "Special characters: @#$%^&*()_+-={}[]|\\:;\"'<>,.?/", "line1\nline2\r\n\r\nline3\n", "a" * 70,
             "MiXeD cAsE wOrDs", "helloworldhelloworldhelloworldhelloworldhelloworld", ""]
    for _ in range(40):
        texts.append(" ".join(rng.choice(sample_words + ["xq", "zzz", "Qk", "12", "!?", "\n"]) for _ in range(rng.randint(1, 30))))
    add("trained_300", vocab_trained(rng, sample, 300), 1000, ["helowrdtkniz", "abcdefghijklmnopqrstuvwxyz", "etaoin"],
        every_len, 300, texts)

    path = os.path.join(ROOT, "tests", "golden", "merge_vectors.json")
    with open(path, "w") as f:
        json.dump({"source": "tools/gen_golden_merge.py: independent list-of-parts restatement of tiktoken's published "
                             "byte-pair encoding (NOT the reference, NOT oracle/tk_oracle.c's form); split by Python regex",
                   "pattern": PATTERN, "vocabs": vocabs}, f, ensure_ascii=True, separators=(",", ":"))
    n = sum(len(v["pieces"]) + len(v["texts"]) for v in vocabs)
    print("wrote %s: %d vocabularies, %d vectors, %d bytes" % (path, len(vocabs), n, os.path.getsize(path)))


if __name__ == "__main__":
    main()
