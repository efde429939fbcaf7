#!/usr/bin/env python3
"""Golden vectors from a THIRD-PARTY engine: HuggingFace `tokenizers` (Rust; 0.22.2 in this image) -> tests/golden/merge_vectors_hf.json.

Label: "independent engine, different authors, NOT the reference".  The reference's arithmetic lives in tiktoken-rs (absent
here); both restatements this repository checks its kernels against (oracle/tk_oracle.c and
tools/gen_golden_merge.py) have the same author as the kernels.  This script removes that objection as far as the image
allows: vocabulary, split AND merge loop come from `tokenizers`:

  * split: pre_tokenizers.Split(Regex(<the literal pattern of reference src/tekkenizer.rs:123>), "isolated") -- Oniguruma, not
    Python `regex`, not this repository's matcher -- followed by the byte-level alphabet (no regex of its own);
  * vocabulary: trainers.BpeTrainer over seeded synthetic text (tools/corpus_gen.c shapes), 256 byte symbols + N merges;
  * merge loop: models.BPE(vocab, merges, ignore_merges=True) -- the whole pre-token is looked up first (tiktoken's
    whole-piece shortcut, SURVEY App. A.2 line 1), otherwise the merges are applied by priority, leftmost first.

Where the two semantics are THE SAME, and where this file stands.  tiktoken ranks an adjacent pair by the rank of its
concatenated BYTES; the merge list ranks it by the index of the PAIR.  With rank(token) = 256 + index of its merge they pick
the same pair as long as, whenever the bytes of two adjacent parts concatenate to a token T, those two parts are T's own
training pair.  A vocabulary that comes out of BPE training has exactly one merge per token (the class the verdict names),
but a text can still put two other parts next to each other whose bytes spell T ("ab" + "c" trained, "a" + "bc" met): there
tiktoken merges and the merge list does not.  The generator therefore checks every emitted vector against the condition
itself -- it replays the merge list on the piece and asserts that no adjacent pair ever spells a token through a
non-training split -- and writes only vectors inside the class (the count of excluded pieces is stored in the file; on the
committed vocabulary it is small).  Inside the class the expected ids are HF's, untouched.

Run: python tools/gen_golden_hf.py   (needs `tokenizers`; the committed JSON is what the tests read).
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

PATTERN = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
NUM_SPECIAL = 1000


def bytes_to_unicode():
    """GPT-2's byte <-> printable char table (what pre_tokenizers.ByteLevel uses)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): b for b, c in zip(bs, cs)}


def in_class(piece, tokens_set, pair_of, order):
    """Replays the merge list on one pre-token and says whether tiktoken's rule provably picks the same merges: at every step,
    any two adjacent parts whose bytes spell a token are that token's training pair."""
    parts = [piece[i:i + 1] for i in range(len(piece))]
    while True:
        best, where = None, None
        for i in range(len(parts) - 1):
            t = parts[i] + parts[i + 1]
            if t in tokens_set:
                if pair_of.get(t) != (parts[i], parts[i + 1]):
                    return False
                if best is None or order[t] < best:
                    best, where = order[t], i
        if where is None:
            return True
        parts[where:where + 2] = [parts[where] + parts[where + 1]]


def main():
    from tokenizers import Tokenizer, Regex, models, pre_tokenizers, trainers
    import corpus
    rng = random.Random(20260403)
    tok = Tokenizer(models.BPE(ignore_merges=True))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(PATTERN), behavior="isolated"),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    n_merges = 2500
    trainer = trainers.BpeTrainer(vocab_size=256 + n_merges, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[],
                                  show_progress=False)
    train = []
    for kind, n, dl, sd in (("ascii", 500, 512, 11), ("mixed", 120, 2048, 12)):
        d, o = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + sd)
        train += [x.decode("utf-8") for x in corpus.docs_of(d, o)]
    tok.train_from_iterator(train, trainer)
    model = json.loads(tok.to_str())["model"]
    u2b = bytes_to_unicode()

    def tb(s):
        return bytes(u2b[c] for c in s)

    merges = [(tb(a), tb(b)) for a, b in model["merges"]]
    tokens = [bytes([i]) for i in range(256)] + [a + b for a, b in merges]
    assert len(set(tokens)) == len(tokens), "a token with two merges: outside the class"
    rank = {t: i for i, t in enumerate(tokens)}
    id2rank = {}
    for s, i in model["vocab"].items():
        id2rank[i] = rank[tb(s)]
    pair_of = {a + b: (a, b) for a, b in merges}
    tokens_set = set(tokens[256:])

    # texts: fresh seeds of the bench shapes, the reference's edge inputs, pieces of every length class of the merge kernels
    texts = []
    for kind, n, dl, sd in (("ascii", 40, 512, 21), ("mixed", 25, 2048, 22), ("zipf", 30, 0, 23)):
        d, o = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + sd)
        texts += [x.decode("utf-8") for x in corpus.docs_of(d, o) if len(x) <= 3000]
    texts += ["Hello, world!", "it's IT'S x's 'abc 'sabc", "1234 56 7", "   whitespace   handling   ", "Line1\nLine2\rLine3\r\nLine4",
              "hi !!\n\nyo", "  \n  \n  x", "a  1", "x\t\ty", "end   ", "Hello\x00World", "日本語 のテキスト", "naïve café — “quotes”"]
    words = sorted({w for t in train[:200] for w in t.split() if w.isalpha() and len(w) > 2})
    for n in (2, 5, 9, 17, 33, 65, 130, 300):
        for _ in range(12):
            w = "".join(rng.choice(words) for _ in range(1 + n // 5))[:n]
            texts.append("x " + w + " y")
            texts.append("".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(n)))
    vectors, excluded = [], 0
    for t in texts:
        enc = tok.encode(t)
        raw = t.encode("utf-8")
        # the pre-tokens as byte strings (offsets are in chars of the ORIGINAL text with use_regex=False byte-level)
        pieces = [tb(s) for s, _ in tok.pre_tokenizer.pre_tokenize_str(t)]
        assert b"".join(pieces) == raw, t[:40]
        if not all(p in rank or in_class(p, tokens_set, pair_of, rank) for p in pieces):
            excluded += 1
            continue
        vectors.append({"text_hex": raw.hex(), "ids": [id2rank[i] + NUM_SPECIAL for i in enc.ids], "n_pieces": len(pieces)})
    out = {"label": "independent engine (HuggingFace tokenizers %s: split by Oniguruma, BPE training, merge-list BPE with ignore_merges), "
                    "different authors, NOT the reference" % __import__("tokenizers").__version__,
           "pattern": PATTERN, "num_special": NUM_SPECIAL, "tokens_hex": [t.hex() for t in tokens],
           "n_texts": len(texts), "n_excluded_outside_class": excluded, "vectors": vectors}
    path = os.path.join(ROOT, "tests", "golden", "merge_vectors_hf.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %s: %d tokens, %d vectors (%d texts outside the class excluded), %.1f KB"
          % (path, len(tokens), len(vectors), excluded, os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
