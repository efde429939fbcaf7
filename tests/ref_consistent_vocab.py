"""A vocabulary CONSTRUCTED to be consistent with the reference's 20 golden id vectors
(reference tests/test_tokenizer_output.rs).

The asset those vectors were produced with (tests/assets/tekken.json) is absent from the
reference mount, so the real byte strings behind most ids are unknown.  What the vectors DO fix:
ids of single bytes (id = byte + 1000), which pieces are single tokens, and how many tokens every
other piece produces.  This module invents the missing byte strings (one admissible segmentation
per multi-token piece, plus low-rank intermediate tokens that make each segment reachable by
leftmost-lowest-rank merging) and lays them out on the ranks the vectors dictate.  Encoding the 20
texts with this vocabulary must reproduce the 20 id vectors exactly: an end-to-end known answer
for split + whole-piece shortcut + merge order + id shift whose expected values come from the
reference's tests.  It is test data, not a claim about the real asset.
"""

NUM_SPECIAL = 1000

# segmentation chosen for every piece that yields more than one id (token strings, in order)
MULTI = {
    "Emojis": ["Em", "ojis"],
    " unicode": [" un", "icode"],
    "Rust": ["R", "ust"],
    "tokenizer": ["token", "izer"],
    " tokenizer": [" to", "kenizer"],
    " Tekken": [" Tek", "ken"],
    "decoding": ["dec", "oding"],
    "comparison": ["compar", "ison"],
    "Mixed": ["M", "ixed"],
    " CaSe": [" Ca", "Se"],
    " WoRdS": [" Wo", "R", "d", "S"],
    "123": ["1", "2", "3"], "456": ["4", "5", "6"], "789": ["7", "8", "9"],
    " @#$%^&*()_+-={}[]|\\:;\"'<>,.?/": [" @", "#", "$", "%", "^", "&", "*", "()", "_", "+-", "={", "}", "[]", "|\\", ":",
                                         ";\"", "'<", ">,", ".", "?", "/"],
}

# intermediate tokens (fresh ranks, allocated in this order from the lowest unused rank >= 256)
INTERMEDIATE = ["oj", "oji", " u", "ic", "ico", "icod", "us", "ke", "iz", "ize", "to", " t", "keni", "keniz", "kenize",
                " T", " Te", "de", "od", "odi", "odin", "co", "com", "comp", "compa", "is", "iso", "ix", "ixe", " C",
                " W"]


def build(ref):
    """ref = tests/golden/reference_vectors.json.  Returns the token list by rank."""
    import tk_oracle
    by_rank = {}

    def put(rank, tok):
        assert by_rank.get(rank, tok) == tok, (rank, tok, by_rank[rank])
        by_rank[rank] = tok

    for b in range(256):
        put(b, bytes([b]))
    for text, ids in ref["encode"]:
        pieces = tk_oracle.split_pieces(text.encode("utf-8"))
        k = 0
        for p in pieces:
            seg = MULTI.get(p.decode("utf-8"))
            if seg is None:
                put(ids[k] - NUM_SPECIAL, p)
                k += 1
            else:
                assert "".join(seg).encode("utf-8") == p, p
                for s in seg:
                    put(ids[k] - NUM_SPECIAL, s.encode("utf-8"))
                    k += 1
        assert k == len(ids), (text, k, len(ids))
    assert len(set(by_rank.values())) == len(by_rank)
    n = max(by_rank) + 1
    free = (r for r in range(256, n) if r not in by_rank)
    have = set(by_rank.values())
    for s in INTERMEDIATE:
        t = s.encode("utf-8")
        if t in have:
            continue
        r = next(free)
        by_rank[r] = t
        have.add(t)
    # every other rank: a filler that can never occur in the test texts (private-use code points)
    toks = []
    for r in range(n):
        if r in by_rank:
            toks.append(by_rank[r])
        else:
            toks.append(("" + chr(0xE100 + (r % 0x1000)) + chr(0xF000 + (r // 0x1000))).encode("utf-8"))
    assert len(set(toks)) == len(toks)
    return toks
