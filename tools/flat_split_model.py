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


def class_masks(buf: bytes, tail: bytes = b""):
    """bit masks over `buf`.  ASCII classes by value; a multi-byte code point gets the class of its code point on ALL of
    its bytes (runs stay contiguous), U8C marks continuation bytes.  `tail` = the bytes that follow buf in the stream
    (a code point that starts inside buf may end there)."""
    import tk_oracle
    cls_fn = tk_oracle.lib().tk_oracle_class
    m = dict(L=0, N=0, S=0, NL=0, SP=0, AP=0, HI=0, STMD=0, RV=0, E=0, LL=0, U8C=0, LS=0, NMB=0)
    n = len(buf)
    ext = buf + tail[:4]
    i = 0
    while i < n:
        b = ext[i]
        bit = 1 << i
        if b >= 0x80:
            m["HI"] |= bit
            if (b & 0xC0) == 0x80:
                m["U8C"] |= bit           # a continuation byte whose lead was classified (or lies below buf)
                i += 1
                continue
            ln, cp = 1, None
            if (b & 0xE0) == 0xC0 and i + 1 < len(ext) and (ext[i + 1] & 0xC0) == 0x80:
                ln, cp = 2, ((b & 0x1F) << 6) | (ext[i + 1] & 0x3F)
            elif (b & 0xF0) == 0xE0 and i + 2 < len(ext) and all((ext[i + k] & 0xC0) == 0x80 for k in (1, 2)):
                ln, cp = 3, ((b & 0x0F) << 12) | ((ext[i + 1] & 0x3F) << 6) | (ext[i + 2] & 0x3F)
            elif (b & 0xF8) == 0xF0 and i + 3 < len(ext) and all((ext[i + k] & 0xC0) == 0x80 for k in (1, 2, 3)):
                ln, cp = 4, ((b & 0x07) << 18) | ((ext[i + 1] & 0x3F) << 12) | ((ext[i + 2] & 0x3F) << 6) | (ext[i + 3] & 0x3F)
            c = cls_fn(cp) if cp is not None else 0
            for k in range(ln):
                if i + k < n:
                    kb = 1 << (i + k)
                    if c == 1:
                        m["L"] |= kb
                    elif c == 2:
                        m["N"] |= kb
                        m["NMB"] |= kb
                    elif c == 3:
                        m["S"] |= kb
            if b == 0xC5 and i + 1 < len(ext) and ext[i + 1] == 0xBF:
                m["LS"] |= bit            # U+017F LATIN SMALL LETTER LONG S folds to 's' ((?i) of the pattern)
            i += 1                        # the continuation bytes are marked by the loop
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
        i += 1
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
    U8C = m.get("U8C", 0)
    CS = full & ~U8C                         # first byte of every code point
    mO = full & ~(mL | mN | mS)
    pOS = p1(mO | SP)
    # alt 1: contractions fire only where a match starts at the apostrophe
    ok = AP & ~pOS
    c2 = ok & n1(m["STMD"])
    c3 = ok & ~c2 & n1((m["RV"] & n1(m["E"])) | (m["LL"] & n1(m["LL"])) | m.get("LS", 0))
    CEND = ((c2 << 2) | (c3 << 3)) & full
    L1, O1 = p1(mL), p1(mO)
    Lst = mL & ~L1
    # "the O char before me is not available as alt 2's prefix": it is itself preceded by O or U+0020.  X = O chars
    # (all their bytes) whose first byte has such a predecessor
    X = CS & mO & pOS
    for _ in range(3):
        X |= p1(X) & U8C
    psL = (mL & L1 & CEND) | (Lst & p1(mN | NL)) | (Lst & p1(X))
    psO = mO & ~O1 & ~p1(SP)
    # numbers: every 3rd char of a run (\p{N}{1,3})
    psN = mN & ~p1(mN)
    if not m.get("NMB", 0):
        M = mN & p1(mN) & p1(p1(mN)) & p1(p1(p1(mN)))      # one byte per char: prefix doubling on byte positions
        k = 3
        while M:
            psN |= (psN << k) & M
            M &= (M << k)
            k *= 2
        psN &= full
    else:
        def adv1(x):                         # the next char start after each start in x, inside the same run
            t = (x << 1) & full
            return ((U8C + t) & ~U8C) & full & mN & nDS
        cur = psN
        while cur:
            cur = adv1(adv1(adv1(cur))) & ~psN
            psN |= cur
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
    last = SPR & ~(cont >> 1) & ~Z & nDE     # last byte of a run that is followed by a non-space char
    for _ in range(3):                       # -> the first byte of that char
        last = (last & CS) | (n1(last & U8C))
    psS = (SPR & ~cont) | ((Z << 1) & cont & ~Z) | (last & CS)
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
        m = {k: v << shift for k, v in class_masks(buf, bytes(data[hi:hi + 4])).items()}
        DS = 0
        for s in doc_starts[bisect.bisect_left(doc_starts, lo):bisect.bisect_left(doc_starts, hi)]:
            DS |= 1 << (s - r0)
        if hi < r1:
            DS |= 1 << (hi - r0)             # end of the stream: nothing follows
        PS, aux = flat_rules(m, DS, region)
        bad = 0
        a, b = c0 - r0, c1 - r0              # commit range in region coordinates
        # (A) a digit / CR-LF run that starts below the region and covers the whole left halo; the region may begin
        # inside a code point: its leading continuation bytes have no class and count as part of the run
        if r0 > 0 and not (DS & 1):
            u8c = m.get("U8C", 0)
            lead_cont = u8c & ~(u8c + 1)
            for run in (aux["mN"] | lead_cont, aux["NL"] | lead_cont):
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


# ------------------------------------------------------------------------------------------
# the JSON pattern of Mistral's tekken.json in lane layout (model of tkf_rules<PAT = 1>).
# Scope of the fast path: chars of the classes U (Lu|Lt), W (Ll), N, S and "other"; a region that holds a neutral
# letter (Lm / Lo: every CJK, Arabic, Hebrew ... char) or a mark (M) is handed back to the sequential path.
# Differences to the hard-coded pattern: no contraction alternative; inside a letter run a piece starts at an upper-case
# char that follows a lower-case one; every digit is a piece; the run absorbed after a punctuation run is made of
# CR / LF / '/' (and whatever follows it starts a piece).
# ------------------------------------------------------------------------------------------
def class_masks_tekken(buf: bytes, tail: bytes = b""):
    import tk_oracle
    cls_fn = tk_oracle.lib().tk_oracle_class2
    m = dict(U=0, W=0, N=0, S=0, NL=0, SP=0, SL=0, HI=0, U8C=0, X=0, M=0)
    n = len(buf)
    ext = buf + tail[:4]
    i = 0
    while i < n:
        b = ext[i]
        bit = 1 << i
        if b >= 0x80:
            m["HI"] |= bit
            if (b & 0xC0) == 0x80:
                m["U8C"] |= bit
                i += 1
                continue
            ln, cp = 1, None
            if (b & 0xE0) == 0xC0 and i + 1 < len(ext) and (ext[i + 1] & 0xC0) == 0x80:
                ln, cp = 2, ((b & 0x1F) << 6) | (ext[i + 1] & 0x3F)
            elif (b & 0xF0) == 0xE0 and i + 2 < len(ext) and all((ext[i + k] & 0xC0) == 0x80 for k in (1, 2)):
                ln, cp = 3, ((b & 0x0F) << 12) | ((ext[i + 1] & 0x3F) << 6) | (ext[i + 2] & 0x3F)
            elif (b & 0xF8) == 0xF0 and i + 3 < len(ext) and all((ext[i + k] & 0xC0) == 0x80 for k in (1, 2, 3)):
                ln, cp = 4, ((b & 0x07) << 18) | ((ext[i + 1] & 0x3F) << 12) | ((ext[i + 2] & 0x3F) << 6) | (ext[i + 3] & 0x3F)
            c = cls_fn(cp) if cp is not None else 0
            for k in range(ln):
                if i + k < n:
                    kb = 1 << (i + k)
                    if c == 1:
                        m["U"] |= kb
                    elif c == 2:
                        m["W"] |= kb
                    elif c == 3:
                        m["X"] |= kb
                    elif c == 4:
                        m["M"] |= kb
                    elif c == 5:
                        m["N"] |= kb
                    elif c == 6:
                        m["S"] |= kb
            i += 1
            continue
        if 0x41 <= b <= 0x5A:
            m["U"] |= bit
        elif 0x61 <= b <= 0x7A:
            m["W"] |= bit
        elif 0x30 <= b <= 0x39:
            m["N"] |= bit
        elif 9 <= b <= 13 or b == 0x20:
            m["S"] |= bit
            if b in (10, 13):
                m["NL"] |= bit
            if b == 0x20:
                m["SP"] |= bit
        elif b == 0x2F:
            m["SL"] |= bit
        i += 1
    return m


def flat_rules_tekken(m, DS, n):
    full = (1 << n) - 1
    DE = (DS >> 1) | (1 << (n - 1))
    nDS, nDE = full & ~DS, full & ~DE

    def p1(x):
        return (x << 1) & nDS & full

    def n1(x):
        return (x >> 1) & nDE

    def fix_right(seed, through):            # seed plus everything reachable to the right through `through`
        a = seed
        while True:
            b = a | (p1(a) & through)
            if b == a:
                return a
            a = b

    def fix_left(seed, through):
        a = seed
        while True:
            b = a | (n1(a) & through)
            if b == a:
                return a
            a = b

    mU, mW, mN, mS, NL, SP, SL = m["U"], m["W"], m["N"], m["S"], m["NL"], m["SP"], m["SL"]
    mX, mM = m.get("X", 0), m.get("M", 0)
    U8C = m.get("U8C", 0)
    CS = full & ~U8C
    mO0 = full & ~(mU | mW | mX | mM | mN | mS)          # punctuation / symbols / everything else
    # Two sets that feed each other, both decided by the char to the LEFT: T = the tail [\\r\\n/]* of a punctuation piece
    # (a CR / LF behind any punctuation char, behind a mark that the 4th alternative swallowed, or behind a tail char; a '/'
    # behind a tail char) and A = where the 4th alternative is running (a punctuation char -- not one of a tail -- behind
    # U+0020, behind a punctuation char that is not in a tail, or behind A; a mark behind A: the class of the 4th
    # alternative holds \\p{M}).  A mark in A counts as punctuation, every other mark is a word char.  Position i needs
    # position i - 1 only, so iterating the two definitions from nothing settles chains of length k after k rounds.
    T, A = 0, 0
    if (NL & p1(mO0)) or mM:
        while True:
            T2 = (NL & p1(mO0 | T | (mM & A))) | (SL & p1(T))
            A2 = CS & ((mO0 & ~T2 & (p1(SP) | p1(mO0 & ~T2) | p1(A))) | (mM & p1(A)))   # decided at the char's first byte ...
            for _ in range(3):
                A2 |= p1(A2) & U8C                                                       # ... and valid for all its bytes
            if T2 == T and A2 == A:
                break
            T, A = T2, A2
    ABS = T
    Mabs = mM & A
    mO = mO0 | Mabs
    neut = mX | (mM & ~Mabs)                               # upper-side AND lower-side chars
    mWd = mU | mW | neut                                   # word chars
    after_abs = p1(ABS) & ~ABS
    Oe = mO & ~ABS
    pOS = p1(Oe | SP)
    Wst = mWd & ~p1(mWd)                                   # first byte of a word run
    X = CS & Oe & pOS                                      # an O char that is not available as a word's one-char prefix
    for _ in range(3):
        X |= p1(X) & U8C
    # inside a word run: an upper-case char starts a piece when the lower side has begun (a lower-case char, then any
    # neutral chars), and the all-upper tail of a run starts one behind a neutral char (the first alternative gives the
    # upper-side run back down to its last neutral char when nothing lower-side follows)
    LW = fix_right(mW, neut) if neut else mW
    psC = CS & mU & p1(LW)
    if neut:
        UE = fix_left(mU & ~n1(mWd), mU)                   # upper-case chars with only upper-case chars up to the run end
        psC |= CS & UE & p1(neut)
    psL = (Wst & p1(mN | NL)) | (Wst & p1(X)) | psC
    psO = Oe & ~p1(Oe) & ~p1(SP)
    psN = mN & CS
    SPR = mS & ~ABS
    NLp = NL & SPR
    cont = SPR & p1(SPR)
    Z = NLp
    C = cont >> 1
    k = 1
    while C and Z:
        Z |= (Z >> k) & C
        C &= (C >> k)
        k *= 2
    last = SPR & ~(cont >> 1) & ~Z & nDE
    for _ in range(3):
        last = (last & CS) | (n1(last & U8C))
    psS = (SPR & ~cont) | ((Z << 1) & cont & ~Z) | (last & CS)
    PS = (psL | psN | psO | psS | (after_abs & CS) | DS | 1) & full
    return PS, dict(SPR=SPR, cont=cont, mN=0, NL=NL | SL, WD=mWd, OM=mO0 | mM)


def flat_split_chunked_tekken(data: bytes, offs, region=2048, hl=32, hr=64):
    """The chunked evaluation of flat_split_chunked with the JSON pattern's classes and rules; a document that touches a
    region with a neutral letter / mark is deferred as well."""
    import bisect
    n = len(data)
    commit = region - hl - hr
    starts, deferred = set(), set()
    doc_starts = sorted(set(int(o) for o in offs[:-1]))

    def doc_of(p):
        return bisect.bisect_right(offs, p) - 1

    c0 = 0
    while c0 < n:
        r0, r1 = c0 - hl, c0 - hl + region
        c1 = min(c0 + commit, n)
        lo, hi = max(r0, 0), min(r1, n)
        buf = bytes(data[lo:hi])
        shift = lo - r0
        m = {k: v << shift for k, v in class_masks_tekken(buf, bytes(data[hi:hi + 4])).items()}
        DS = 0
        for s in doc_starts[bisect.bisect_left(doc_starts, lo):bisect.bisect_left(doc_starts, hi)]:
            DS |= 1 << (s - r0)
        if hi < r1:
            DS |= 1 << (hi - r0)
        PS, aux = flat_rules_tekken(m, DS, region)
        bad = 0
        a, b = c0 - r0, c1 - r0
        if r0 > 0 and not (DS & 1):
            # runs whose state comes from below the region: the CR / LF / '/' tail, a word run (upper or lower side?)
            # and a punctuation / mark run (is the 4th alternative running?) must not cover the whole left halo
            lead_cont = m["U8C"] & ~(m["U8C"] + 1)
            # (tail chars, punctuation and marks feed each other's state: a chain of them is one run here)
            for run in (aux["WD"] | lead_cont, aux["OM"] | aux["NL"] | lead_cont):
                if run & 1:
                    e = 0
                    while e < region and (run >> e) & 1 and not (e > 0 and (DS >> e) & 1):
                        e += 1
                    if e >= a:
                        bad |= 1 << a
        last = region - 1
        if r1 < n and (aux["SPR"] >> last) & 1 and not _is_start(doc_starts, r1):
            f = last
            while f > 0 and (aux["cont"] >> f) & 1:
                f -= 1
            if f < b:
                bad |= 1 << max(f, a)
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
