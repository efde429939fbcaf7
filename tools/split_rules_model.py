"""Executable model of the LOCAL piece-start rules the HIP split uses.

The reference matches the 7-alternative pattern of src/tekkenizer.rs:123 sequentially
(leftmost-first, each match starts where the previous one ended).  The GPU path instead lets
every byte position decide "does a piece start here?" from class masks of its neighbourhood
(DESIGN.md "Split rules").  This file states those rules position-by-position in plain
Python so that the CPU test-suite can check them against the oracle / Python `regex`
without a GPU.  tests/test_split_rules.py drives it; the kernel in
tekken-rs_amd/csrc/tk_kernels.hip implements exactly these predicates with wave ballots.

`window_split` models the 64-byte sliding window including the certainty logic that
decides how far a window may commit.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "oracle"))

O, L, N, S = 0, 1, 2, 3


def _classes(text: bytes):
    """Per-byte class (propagated to continuation bytes), char-start flags, char lengths."""
    import tk_oracle
    cls_fn = tk_oracle.lib().tk_oracle_class
    n = len(text)
    cls = [O] * n
    cs = [False] * n
    clen = [1] * n
    i = 0
    while i < n:
        b0 = text[i]
        ln, cp = 1, b0
        if b0 >= 0x80:
            cp = 0xFFFFFFFF
            if (b0 & 0xE0) == 0xC0 and i + 1 < n and (text[i + 1] & 0xC0) == 0x80:
                ln, cp = 2, ((b0 & 0x1F) << 6) | (text[i + 1] & 0x3F)
            elif (b0 & 0xF0) == 0xE0 and i + 2 < n and all((text[i + k] & 0xC0) == 0x80 for k in (1, 2)):
                ln, cp = 3, ((b0 & 0x0F) << 12) | ((text[i + 1] & 0x3F) << 6) | (text[i + 2] & 0x3F)
            elif (b0 & 0xF8) == 0xF0 and i + 3 < n and all((text[i + k] & 0xC0) == 0x80 for k in (1, 2, 3)):
                ln, cp = 4, ((b0 & 0x07) << 18) | ((text[i + 1] & 0x3F) << 12) | ((text[i + 2] & 0x3F) << 6) | (
                    text[i + 3] & 0x3F)
        c = cls_fn(cp) if cp != 0xFFFFFFFF else O
        cs[i] = True
        clen[i] = ln
        for k in range(ln):
            cls[i + k] = c
        i += ln
    return cls, cs, clen


def _contraction_len(text, i):
    """Byte length of (?i:'s|'t|'re|'ve|'m|'ll|'d) at i, 0 if it does not match there.
    (b | 0x20) == letter is an exact ASCII case-insensitive compare; U+017F folds to 's'."""
    n = len(text)
    if text[i] != 0x27 or i + 1 >= n:
        return 0
    f1 = text[i + 1] | 0x20
    if f1 in (0x73, 0x74, 0x6D, 0x64):  # s t m d
        return 2
    if i + 2 < n:
        b2 = text[i + 2]
        if text[i + 1] == 0xC5 and b2 == 0xBF:  # U+017F long s
            return 3
        f2 = b2 | 0x20
        if (f1, f2) in ((0x72, 0x65), (0x76, 0x65), (0x6C, 0x6C)):  # re ve ll
            return 3
    return 0


def rule_split(text: bytes, eot=True):
    """Piece starts of `text` (position 0 is a known piece start) by local rules.

    eot=False models a window whose document continues after `text`; then the second
    return value is the number of leading positions whose decision is certain.
    Returns (starts list, certain_upto)."""
    n = len(text)
    if n == 0:
        return [], 0
    if not eot:
        # drop a trailing char whose nominal length runs past the window: its class is unknown
        j = n - 1
        while j > 0 and (text[j] & 0xC0) == 0x80:
            j -= 1
        b0 = text[j]
        nominal = 1 if b0 < 0xC0 else 2 if b0 < 0xE0 else 3 if b0 < 0xF0 else 4 if b0 < 0xF8 else 1
        if j + nominal > n:
            text = text[:j]
            n = j
            if n == 0:
                return [], 0
    cls, cs, clen = _classes(text)
    nl = [b in (0x0A, 0x0D) for b in text]

    # contraction fire: apostrophe where a match starts and the letters fit
    fire = [0] * n
    for i in range(n):
        if text[i] == 0x27:
            ce = _contraction_len(text, i)
            if ce and (i == 0 or (cls[i - 1] != O and text[i - 1] != 0x20)):
                fire[i] = ce
    cend = [False] * (n + 4)
    for i in range(n):
        if fire[i]:
            cend[i + fire[i]] = True

    # NLs absorbed by alt 4's trailing [\r\n]* : leading CR/LF of a white-space run that
    # directly follows a class-O byte (never at position 0: that is a piece start)
    absb = [False] * n
    for i in range(1, n):
        absb[i] = nl[i] and (cls[i - 1] == O or absb[i - 1])
    sp = [cls[i] == S and not absb[i] for i in range(n)]  # effective white space S'

    starts = []
    uncertain_from = n
    for i in range(n):
        if not cs[i]:
            continue
        if i == 0:
            starts.append(0)
            continue
        c, pc = cls[i], cls[i - 1]
        st = False
        if c == L:
            if pc == L:
                st = cend[i]
            elif pc == N:
                st = True
            elif pc == S:
                st = nl[i - 1]  # a non-CR/LF white-space char is absorbed as alt 2's prefix
            else:  # pc == O
                qs = i - 1
                while not cs[qs]:
                    qs -= 1
                runstart = qs == 0 or cls[qs - 1] != O
                if not runstart:
                    st = True  # O-run of >= 2 chars was consumed whole by alt 4
                elif qs > 0 and text[qs - 1] == 0x20:
                    st = True  # ' q' consumed by alt 4
                else:
                    st = False  # q is alt 2's prefix, or a contraction covers position i
        elif c == N:
            if pc != N:
                st = True
            else:
                r = i
                k = 0
                while r > 0 and cls[r - 1] == N:
                    r -= 1
                    if cs[r]:
                        k += 1
                st = (k % 3 == 0)
        elif c == O:
            st = (pc != O) and text[i - 1] != 0x20
        else:  # white space
            if absb[i]:
                st = False
            elif not sp[i - 1]:
                st = True  # start of the effective run (a')
            else:
                # extent of the effective run from i
                e = i
                while e < n and sp[e]:
                    e += 1
                if not eot and e == n:
                    uncertain_from = min(uncertain_from, i)
                    continue
                later_nl = any(nl[k] for k in range(i, e))
                if later_nl:
                    st = False
                elif nl[i - 1]:
                    st = True  # u = k+1
                else:
                    last_char = (i + clen[i] == e)
                    st = last_char and e < n  # followed by a non-space char (not end of text)
        if st:
            starts.append(i)
    return starts, uncertain_from


def window_split(text: bytes, W=64):
    """Sliding-window driver: returns (starts, n_fallback) where n_fallback counts windows that
    could not commit a single piece (the kernel hands those documents to the long-piece path)."""
    n = len(text)
    out = []
    w0 = 0
    fallback = 0
    while w0 < n:
        nv = min(W, n - w0)
        at_end = (w0 + nv == n)
        starts, unc = rule_split(text[w0:w0 + nv], eot=at_end)
        if at_end:
            out.extend(w0 + s for s in starts)
            break
        certain = [s for s in starts if s < unc]
        estar = certain[-1]
        if estar == 0:
            # no progress: model the long-piece path with the unbounded rules
            fallback += 1
            starts2, _ = rule_split(text[w0:], eot=True)
            nxt = starts2[1] if len(starts2) > 1 else n - w0
            out.append(w0)
            w0 += nxt
            continue
        out.extend(w0 + s for s in certain[:-1])
        w0 += estar
    return out, fallback
