"""Executable model of the FLAT split: the piece-start rules of tools/split_rules_model.py restated as
pure mask algebra over the packed byte stream of ALL documents (bit i = byte i of the packed buffer).

This is what tk_flat_kernel (tekken-rs_amd/csrc/tk_flat_impl.h) evaluates, one chunk of the stream per
wave with the masks held "lane layout" (lane l owns bits [W*l, W*l+W) of the chunk).  Document
boundaries are just another mask: every look-behind shift is cut at a document start (DS), every
look-ahead shift at a document end (DE), runs are broken at DS.  ASCII only: a document with a byte
>= 0x80, a digit / CR-LF run that covers the whole left halo, a white-space run that reaches the end
of the loaded region, or a piece longer than the right halo is DEFERRED to the per-document kernel.

Python ints are the masks here; tests/test_flat_model.py checks the result against the oracle split.
The pattern is the literal of reference src/tekkenizer.rs:123.
"""


def class_masks(buf: bytes):
    """bit masks over `buf` (ASCII classes; bytes >= 0x80 only set HI)."""
    m = dict(L=0, N=0, S=0, NL=0, SP=0, AP=0, HI=0, STMD=0, RV=0, E=0, LL=0)
    for i, b in enumerate(buf):
        bit = 1 << i
        if b >= 0x80:
            m["HI"] |= bit
            continue
        f = b | 0x20
        if 0x61 <= f <= 0x7A:
            m["L"] |= bit
            if f in (0x73, 0x74, 0x6D, 0x64):
                m["STMD"] |= bit
            elif f in (0x72, 0x76):
                m["RV"] |= bit
            elif f == 0x65:
                m["E"] |= bit
            elif f == 0x6C:
                m["LL"] |= bit
        elif 0x30 <= b <= 0x39:
            m["N"] |= bit
        elif 9 <= b <= 13 or b == 0x20:
            m["S"] |= bit
            if b in (10, 13):
                m["NL"] |= bit
            if b == 0x20:
                m["SP"] |= bit
        elif b == 0x27:
            m["AP"] |= bit
    return m


def _is_start(doc_starts, p):
    import bisect
    i = bisect.bisect_left(doc_starts, p)
    return i < len(doc_starts) and doc_starts[i] == p


def flat_rules(m, DS, n):
    """Piece starts of an n-byte region.  DS = mask of document starts inside the region (bit 0 is treated
    as a start of the region's context: nothing is known below it).  Returns PS."""
    full = (1 << n) - 1
    DE = (DS >> 1) | (1 << (n - 1))          # last byte of a document (or of the region)
    nDS, nDE = full & ~DS, full & ~DE

    def p1(x):                               # "the previous byte, same document, has x"
        return (x << 1) & nDS & full

    def n1(x):                               # "the next byte, same document, has x"
        return (x >> 1) & nDE

    mL, mN, mS, NL, SP, AP = m["L"], m["N"], m["S"], m["NL"], m["SP"], m["AP"]
    mO = full & ~(mL | mN | mS)
    # alt 1: contractions fire only where a match starts at the apostrophe
    ok = AP & ~p1(mO | SP)
    c2 = ok & n1(m["STMD"])
    c3 = ok & ~c2 & n1((m["RV"] & n1(m["E"])) | (m["LL"] & n1(m["LL"])))
    CEND = ((c2 << 2) | (c3 << 3)) & full
    L1, O1 = p1(mL), p1(mO)
    Lst = mL & ~L1
    psL = (mL & L1 & CEND) | (Lst & p1(mN | NL)) | (Lst & O1 & p1(p1(mO | SP)))
    psO = mO & ~O1 & ~p1(SP)
    # numbers: every 3rd char of a run (\p{N}{1,3}), by prefix doubling
    psN = mN & ~p1(mN)
    M = mN & p1(mN) & p1(p1(mN)) & p1(p1(p1(mN)))
    k = 3
    while M:
        psN |= (psN << k) & M
        M &= (M << k)
        k *= 2
    psN &= full
    # white space
    seeds = NL & O1
    Rn = NL & nDS
    ABS = (((Rn + seeds) ^ Rn) & Rn) & full if seeds else 0
    SPR = mS & ~ABS
    NLp = NL & SPR
    cont = SPR & p1(SPR)                     # continues a run from the previous byte
    Z = NLp                                  # positions up to and including the last CR/LF of their run
    C = cont >> 1                            # C[i]: byte i+1 continues the run of byte i
    k = 1
    while C and Z:
        Z |= (Z >> k) & C
        C &= (C >> k)
        k *= 2
    psS = (SPR & ~cont) | ((Z << 1) & cont & ~Z) | (SPR & ~(cont >> 1) & ~Z & nDE)
    return (psL | psN | psO | psS | DS | 1) & full, dict(SPR=SPR, cont=cont, mN=mN, NL=NL)


def flat_split_chunked(data: bytes, offs, region=1024, hl=32, hr=64):
    """Chunked evaluation as the kernel does it.  Returns (starts, deferred): starts = sorted global piece
    start positions of the non-deferred documents, deferred = set of document indices."""
    n = len(data)
    D = len(offs) - 1
    commit = region - hl - hr
    starts = set()
    deferred = set()
    doc_starts = sorted(set(int(o) for o in offs[:-1]))
    import bisect

    def doc_of(p):                           # the non-empty document containing byte p
        return bisect.bisect_right(offs, p) - 1

    c0 = 0
    while c0 < n:
        r0, r1 = c0 - hl, c0 - hl + region
        c1 = min(c0 + commit, n)
        lo, hi = max(r0, 0), min(r1, n)
        buf = bytes(data[lo:hi])
        shift = lo - r0                      # bytes below 0 do not exist
        m = {k: v << shift for k, v in class_masks(buf).items()}
        DS = 0
        for s in doc_starts[bisect.bisect_left(doc_starts, lo):bisect.bisect_left(doc_starts, hi)]:
            DS |= 1 << (s - r0)
        if hi < r1:
            DS |= 1 << (hi - r0)             # end of the stream: nothing follows
        PS, aux = flat_rules(m, DS, region)
        bad = 0
        a, b = c0 - r0, c1 - r0              # commit range in region coordinates
        bad |= m["HI"] & (((1 << b) - 1) ^ ((1 << a) - 1))
        # (A) a digit / CR-LF run that starts below the region and covers the whole left halo
        if r0 > 0 and not (DS & 1):
            for run in (aux["mN"], aux["NL"]):
                if run & 1:
                    e = 0
                    while e < region and (run >> e) & 1 and not (e > 0 and (DS >> e) & 1):
                        e += 1
                    if e >= a:
                        bad |= 1 << a
        # (B) a white-space run that reaches the end of the loaded region (and goes on in the same document)
        # and started inside the commit range: its decisions need bytes that were not loaded
        last = region - 1
        if r1 < n and (aux["SPR"] >> last) & 1 and not _is_start(doc_starts, r1):
            f = last
            while f > 0 and (aux["cont"] >> f) & 1:
                f -= 1
            if f < b:
                bad |= 1 << max(f, a)
        # pieces owned by this chunk: starts in [a, b); the end is the next start (sentinel at `region`)
        pos = [i for i in range(a, region) if (PS >> i) & 1] + [region]
        for j, p in enumerate(pos[:-1]):
            if p >= b:
                break
            if pos[j + 1] - p > 64:
                bad |= 1 << p
            starts.add(r0 + p)
        q = 0
        while bad >> q:
            if (bad >> q) & 1:
                deferred.add(doc_of(r0 + q))
            q += 1
        c0 += commit
    keep = sorted(s for s in starts if doc_of(s) not in deferred)
    return keep, deferred
