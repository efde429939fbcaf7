// tk_encode_impl.h -- the wave-level tokenization algorithm (split + lookup + merge + emit).
//
// This is the device code of the hot path named by BASELINE.json: what the reference does in
//   CoreBPE::encode(text, {})              reference src/tekkenizer.rs:384-386
//   token += num_special_tokens            reference src/tekkenizer.rs:390-392
//   BOS / EOS insertion                    reference src/tekkenizer.rs:394-402
// re-designed for 64-wide CDNA4 wavefronts: one wave walks one document with a sliding
// 64-byte window, ONE LANE PER BYTE.
//
//   1. load 64 bytes, classify every code point (ASCII by ALU, others via the class trie)
//   2. wave ballots turn the classes into 64-bit masks; every lane decides locally whether a
//      piece starts at its byte (the 7 alternatives of src/tekkenizer.rs:123 restated as
//      local rules -- DESIGN.md "Split rules", modelled in tools/split_rules_model.py)
//   3. the window commits up to its last start that no unseen byte can change
//   4. each piece-start lane probes SHORT/LONG for the whole-piece shortcut
//   5. pieces that miss are merged in place: lanes are parts, an `alive` ballot mask is the
//      part list, the leftmost minimum-rank pair of every piece is found with a segmented
//      min-scan over lanes and merged, one merge per piece per round (exactly the order of
//      tiktoken's _byte_pair_merge); pair ranks come from PAIR2/PAIR
//   6. surviving part heads are compacted with popcounts and written as final ids
//
// A document with a piece that does not fit a window (>~60 bytes) is handed to a second pass
// (tk_encode_doc_seq): sequential end search per piece, wave-wide polynomial hash, LONG probe
// and, on a miss, a wave-cooperative merge over scratch memory with per-64-part block minima.
//
// The file is included with a set of wave primitives already defined (wv_lane, wv_ballot,
// wv_shfl, wv_first, wv_first64, wv_up1, wv_dn1, wv_sync, wv_atomic_add, wv_atomic_add_all, wv_brev64, wv_min_u32, wv_readlane, TK_DEV): tk_wave_hip.h for gfx950, and a
// fiber emulator in tests/emu/ that lets the CPU test-suite run this very source.
#ifndef TK_ENCODE_IMPL_H
#define TK_ENCODE_IMPL_H
#include <stdint.h>

#include "tk_encode_impl_args.h"

#define TK_CLS_O 0u
#define TK_CLS_L 1u
#define TK_CLS_N 2u
#define TK_CLS_S 3u
#define TK_DEAD 0xFFFFFFFFu
#define TK_NONE 0xFFFFFFFFu
#define TK_DOC_CHUNK 8u /* documents taken per work-queue fetch */
#define TK_LONG_MAX 32768u /* longest piece the round-based merge takes (16 waves x 32 steps x 64 parts) */


struct TkPolyPow {  // per-lane powers of the two polynomial bases
    uint32_t pw1, ipw1, pw2, ipw2;
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
TK_DEV uint64_t tk_lowmask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
TK_DEV int tk_ctz64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }
TK_DEV int tk_msb64(uint64_t x) { return x ? 63 - __builtin_clzll(x) : 0; }
TK_DEV bool tk_bit(uint64_t m, int i) { return (m >> i) & 1ull; }
TK_DEV int tk_popc64(uint64_t x) { return __builtin_popcountll(x); }

TK_DEV uint32_t tk_ascii_class(uint32_t b) {
    if (((b | 0x20u) - 0x61u) < 26u) return TK_CLS_L;
    if ((b - 0x30u) < 10u) return TK_CLS_N;
    if ((b - 9u) < 5u || b == 0x20u) return TK_CLS_S;
    return TK_CLS_O;
}

TK_DEV uint32_t tk_uc_class(const TkTablesView& t, uint32_t cp) {
    if (cp >= 0x110000u) return TK_CLS_O;
    uint32_t blk = t.uc_stage1[cp >> 7];
    uint32_t w = t.uc_stage2[blk * 8u + ((cp & 127u) >> 4)];
    return (w >> (2u * (cp & 15u))) & 3u;
}

struct alignas(16) tk_u32x4 { uint32_t x, y, z, w; };

// One probe = ONE round trip: both candidate entries (cuckoo, tk_hash.h) are fetched together with 16-byte loads
// and compared with bitwise ops; there is no probe loop.
struct alignas(8) tk_u32x2 { uint32_t x, y; };

// Whole-piece lookup.  Pieces of up to 8 bytes -- most of the text -- take ONE 16-byte load (key, rank and length in one
// entry of KEY8); pieces of 9..16 bytes a 16-byte + an 8-byte load from KEY16, whose first 16 bytes have the KEY8 layout:
// a short piece has k2 = k3 = 0 and loads nothing for them, so ONE compare sequence without selects serves both.  The
// second cuckoo location is fetched only by the lanes whose first one did not match AND is flagged (first choices are
// filled first by the builder; an unflagged slot that does not match is a definite miss).
// One location: rank or TK_RANK_MAX; *lw = the slot's length word (TK_KEY_SPILL = something spilled from here).
TK_DEV uint32_t tk_key_slot(const uint8_t* p, bool s8, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len, uint32_t* lw) {
    tk_u32x4 a = *reinterpret_cast<const tk_u32x4*>(p);           // {k0, k1, rank, len | flag}
    tk_u32x2 b;
    b.x = 0u; b.y = 0u;
    if (!s8) b = *reinterpret_cast<const tk_u32x2*>(p + 16);      // KEY16: {k2, k3}
    WV_PIN(a.x); WV_PIN(a.y); WV_PIN(a.z); WV_PIN(a.w); WV_PIN(b.x); WV_PIN(b.y);   // both loads before the first compare
    const uint32_t d = (a.x ^ k0) | (a.y ^ k1) | ((a.w & ~TK_KEY_SPILL) ^ len) | (b.x ^ k2) | (b.y ^ k3);
    *lw = a.w;
    return d == 0u ? a.z : TK_RANK_MAX;                           // an empty entry has len 0
}
// h = tk_key_hash of the key; KEY8 / KEY16 lanes issue their loads TOGETHER and wait once
TK_DEV uint32_t tk_probe_key_h(const uint8_t* key8_tab, uint32_t key8_mask, const uint8_t* key16_tab, uint32_t key16_mask, uint32_t h,
                               uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len) {
    const bool s8 = len <= 8u;
    const uint8_t* base = s8 ? key8_tab : key16_tab;
    const uint32_t o1 = s8 ? (h & key8_mask) << 4 : (h & key16_mask) << 5;
    uint32_t lw;
    uint32_t r = tk_key_slot(base + o1, s8, k0, k1, k2, k3, len, &lw);
    if (r == TK_RANK_MAX && (lw & TK_KEY_SPILL)) {
        const uint32_t h2 = tk_hash_alt(h);
        const uint32_t o2 = s8 ? (h2 & key8_mask) << 4 : (h2 & key16_mask) << 5;
        r = tk_key_slot(base + o2, s8, k0, k1, k2, k3, len, &lw);
    }
    return r;
}
TK_DEV uint32_t tk_probe_key(const TkTablesView& t, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len) {
    return tk_probe_key_h(reinterpret_cast<const uint8_t*>(t.key8_tab), t.key8_mask, reinterpret_cast<const uint8_t*>(t.key_tab), t.key_mask,
                          tk_key_hash(t.key_hash_mode, k0, k1, k2, k3, len), k0, k1, k2, k3, len);
}

// text points at the piece bytes in the packed buffer; a tag match is verified byte by byte so
// that the result is exact, not probabilistic.
TK_DEV uint32_t tk_probe_long(const TkTablesView& t, uint32_t h1, uint32_t h2, uint32_t len, const uint8_t* text) {
    uint32_t s = tk_long_hash(h1, len) & t.long_mask;
    for (uint32_t tries = 0; tries <= t.long_mask; ++tries) {
        const tk_u32x4 e = *reinterpret_cast<const tk_u32x4*>(t.long_tab + s);  // {tag, rank, len, blob_off}
        if (e.z == 0u) return TK_RANK_MAX;
        if (((e.z ^ len) | (e.x ^ h2)) == 0u) {
            const uint8_t* q = t.blob + e.w;
            uint32_t diff = 0;
            for (uint32_t k = 0; k < len; ++k) diff |= (uint32_t)(q[k] ^ text[k]);  // no early exit: loads pipeline
            if (diff == 0u) return e.y;
        }
        s = (s + 1u) & t.long_mask;
    }
    return TK_RANK_MAX;
}

// KEY64 (tk_hash.h): the whole-piece look-up of a piece of 17..64 bytes by ONE LANE, the piece's bytes as dwords -- tw = the
// aligned word of the text copy that holds its first byte, sh = that byte's offset in it (the words behind the piece are
// readable).  One multiply per dword, the verification against the blob by dwords too.
TK_DEV uint32_t tk_load_u32_unaligned(const uint8_t* p) {
    typedef uint32_t __attribute__((aligned(1))) u32_u;
    return *reinterpret_cast<const u32_u*>(p);
}
TK_DEV uint32_t tk_probe_key64(const TkTablesView& t, const uint32_t* tw, uint32_t sh, uint32_t len, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
    // k0..k3: the piece's first 16 bytes (the caller has them in registers: the exact-key probe of the shorter pieces takes them too)
    const uint32_t nd = (len + 3u) >> 2;                    // 5..16
    const uint32_t tail = (len & 3u) ? ((1u << (8u * (len & 3u))) - 1u) : 0xFFFFFFFFu;
    uint32_t ha = tk_k64_start(k0, k1, k2, k3);
    {
        uint32_t lo = tw[4];
        for (uint32_t j = 4; j < nd; ++j) {
            const uint32_t hi = tw[j + 1];
            uint32_t w = wv_alignbyte(hi, lo, sh);
            if (j + 1u == nd) w &= tail;
            tk_k64_step(ha, w);
            lo = hi;
        }
    }
    const uint32_t hb = tk_k64_tag(ha);
    uint32_t s = tk_k64_slot(ha, len) & t.key64_mask;
    // Two slots per round trip: a piece that is no token (most long pieces of running text) ends at the first EMPTY slot, and with
    // one slot per trip the wave waited for the longest probe chain among its lanes.
    for (uint32_t tries = 0; tries <= t.key64_mask; tries += 2) {
        const uint32_t s1 = (s + 1u) & t.key64_mask;
        tk_u32x4 e0 = *reinterpret_cast<const tk_u32x4*>(t.key64_tab + s);   // {tag, rank, len, blob_off}
        tk_u32x4 e1 = *reinterpret_cast<const tk_u32x4*>(t.key64_tab + s1);
        WV_PIN(e0.x); WV_PIN(e1.x);                         // both loads before the first compare
        for (int q = 0; q < 2; ++q) {
            const tk_u32x4 e = q ? e1 : e0;
            if (e.z == 0u) return TK_RANK_MAX;
            if (((e.z ^ len) | (e.x ^ hb)) == 0u) {
                const uint8_t* qb = t.blob + e.w;                                 // (the blob has 16 bytes of slack behind its last token)
                uint32_t diff = 0, lo = tw[0];
                for (uint32_t j = 0; j < nd; ++j) {
                    const uint32_t hi = tw[j + 1];
                    uint32_t w = wv_alignbyte(hi, lo, sh) ^ tk_load_u32_unaligned(qb + 4u * j);
                    if (j + 1u == nd) w &= tail;
                    diff |= w;
                    lo = hi;
                }
                if (diff == 0u) return e.y;
            }
        }
        s = (s + 2u) & t.key64_mask;
    }
    return TK_RANK_MAX;
}

struct alignas(16) tk_u64x2 { uint64_t x, y; };

// PAIR: buckets of two entries (one 16-byte load).  Most probes of the merge loop are for pairs that are NOT tokens;
// the first-choice bucket alone settles them unless it is flagged (something spilled from it): one scattered load per
// probe instead of two -- the merge kernel is bound by exactly those loads.  An empty entry is all ones and matches
// no key.
TK_DEV uint32_t tk_pair_in(const tk_u64x2& p, uint64_t key) {
    uint32_t r = TK_RANK_MAX;
    if (tk_pair_key(p.x) == key) r = tk_pair_rank(p.x);
    if (tk_pair_key(p.y) == key) r = tk_pair_rank(p.y);
    return r;
}
TK_DEV bool tk_pair_spilled(const tk_u64x2& p) { return p.x != TK_PAIR_EMPTY && (p.x & TK_PAIR_SPILL) != 0ull; }

TK_DEV uint32_t tk_probe_pair(const TkTablesView& t, uint32_t a, uint32_t b) {
    const uint64_t key = ((uint64_t)a << TK_ID_BITS) | (uint64_t)b;
    const uint32_t h = tk_pair_hash(a, b);
    tk_u64x2 p = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (h & t.pair_mask));
    WV_PIN(p.x); WV_PIN(p.y);
    uint32_t r = tk_pair_in(p, key);
    if (r == TK_RANK_MAX && tk_pair_spilled(p)) {
        tk_u64x2 q = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (tk_hash_alt(h) & t.pair_mask));
        WV_PIN(q.x); WV_PIN(q.y);
        r = tk_pair_in(q, key);
    }
    return r;
}

// two independent PAIR probes: both first-choice buckets are in flight together, and so are the second-choice ones of
// the (few) lanes that need them
TK_DEV void tk_probe_pair_x2(const TkTablesView& t, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t& r0,
                             uint32_t& r1) {
    const uint64_t key0 = ((uint64_t)a0 << TK_ID_BITS) | (uint64_t)b0, key1 = ((uint64_t)a1 << TK_ID_BITS) | (uint64_t)b1;
    const uint32_t h0 = tk_pair_hash(a0, b0), h1 = tk_pair_hash(a1, b1);
    tk_u64x2 p0 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (h0 & t.pair_mask));
    tk_u64x2 p1 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (h1 & t.pair_mask));
    WV_PIN(p0.x); WV_PIN(p0.y); WV_PIN(p1.x); WV_PIN(p1.y);
    r0 = tk_pair_in(p0, key0);
    r1 = tk_pair_in(p1, key1);
    const bool n0 = r0 == TK_RANK_MAX && tk_pair_spilled(p0), n1 = r1 == TK_RANK_MAX && tk_pair_spilled(p1);
    if (n0 || n1) {
        tk_u64x2 q0, q1;
        q0.x = q0.y = q1.x = q1.y = TK_PAIR_EMPTY;
        if (n0) q0 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (tk_hash_alt(h0) & t.pair_mask));
        if (n1) q1 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (tk_hash_alt(h1) & t.pair_mask));
        WV_PIN(q0.x); WV_PIN(q0.y); WV_PIN(q1.x); WV_PIN(q1.y);
        if (n0) r0 = tk_pair_in(q0, key0);
        if (n1) r1 = tk_pair_in(q1, key1);
    }
}

TK_DEV uint32_t tk_probe_pair_f(const TkTablesView& t, const uint32_t* filt, uint32_t a, uint32_t b) {
    if (!filt) return tk_probe_pair(t, a, b);
    const uint32_t f = tk_pair_fbit(tk_pair_hash(a, b));
    return ((filt[f >> 5] >> (f & 31u)) & 1u) ? tk_probe_pair(t, a, b) : TK_RANK_MAX;
}

// the merge kernels' form: either probe may be unwanted, and `filt` (the PAIR filter, tk_hash.h, in LDS there) lets a
// probe whose bit is clear skip its gather -- the pair is in no bucket
TK_DEV void tk_probe_pair_x2f(const TkTablesView& t, const uint32_t* filt, bool want0, uint32_t a0, uint32_t b0, bool want1,
                              uint32_t a1, uint32_t b1, uint32_t& r0, uint32_t& r1) {
    const uint64_t key0 = ((uint64_t)a0 << TK_ID_BITS) | (uint64_t)b0, key1 = ((uint64_t)a1 << TK_ID_BITS) | (uint64_t)b1;
    const uint32_t h0 = tk_pair_hash(a0, b0), h1 = tk_pair_hash(a1, b1);
    const uint32_t f0 = tk_pair_fbit(h0), f1 = tk_pair_fbit(h1);
    bool m0 = want0, m1 = want1;
    if (filt) {
        const uint32_t w0 = filt[f0 >> 5], w1 = filt[f1 >> 5];
        m0 = m0 && ((w0 >> (f0 & 31u)) & 1u);
        m1 = m1 && ((w1 >> (f1 & 31u)) & 1u);
    }
    tk_u64x2 p0, p1;
    p0.x = p0.y = p1.x = p1.y = TK_PAIR_EMPTY;
    if (m0) p0 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (h0 & t.pair_mask));
    if (m1) p1 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (h1 & t.pair_mask));
    WV_PIN(p0.x); WV_PIN(p0.y); WV_PIN(p1.x); WV_PIN(p1.y);
    r0 = tk_pair_in(p0, key0);
    r1 = tk_pair_in(p1, key1);
    const bool n0 = r0 == TK_RANK_MAX && tk_pair_spilled(p0), n1 = r1 == TK_RANK_MAX && tk_pair_spilled(p1);
    if (n0 || n1) {
        tk_u64x2 q0, q1;
        q0.x = q0.y = q1.x = q1.y = TK_PAIR_EMPTY;
        if (n0) q0 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (tk_hash_alt(h0) & t.pair_mask));
        if (n1) q1 = *reinterpret_cast<const tk_u64x2*>(t.pair_tab + 2u * (tk_hash_alt(h1) & t.pair_mask));
        WV_PIN(q0.x); WV_PIN(q0.y); WV_PIN(q1.x); WV_PIN(q1.y);
        if (n0) r0 = tk_pair_in(q0, key0);
        if (n1) r1 = tk_pair_in(q1, key1);
    }
}

TK_DEV uint32_t tk_wave_sum(uint32_t v, int lane) {
    for (int d = 1; d < 64; d <<= 1) v += wv_shfl(v, lane ^ d);
    return v;
}

TK_DEV uint64_t tk_wave_min64(uint64_t v, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = wv_shfl((uint32_t)v, lane ^ d);
        uint32_t hi = wv_shfl((uint32_t)(v >> 32), lane ^ d);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// minimum of (rank << 32 | pos) over the wave with the DPP reduction (two 32-bit minima: the rank, then the position
// among the lanes that hold it) -- about 150 clocks instead of the 1 500 of the bpermute butterfly on 64-bit values
TK_DEV uint64_t tk_wave_min_key(uint32_t rank, uint32_t pos) {
    const uint32_t r = wv_min_u32(rank);
    const uint32_t p = wv_min_u32(rank == r ? pos : 0xFFFFFFFFu);
    return r == TK_RANK_MAX ? ~0ull : (((uint64_t)r << 32) | (uint64_t)p);
}

TK_DEV TkPolyPow tk_poly_pow(const TkTablesView& t, int lane) {
    TkPolyPow p;
    p.pw1 = 1u; p.ipw1 = 1u; p.pw2 = 1u; p.ipw2 = 1u;
    for (int i = 0; i < lane; ++i) {
        p.pw1 *= TK_POLY_P1; p.ipw1 *= t.p1inv;
        p.pw2 *= TK_POLY_P2; p.ipw2 *= t.p2inv;
    }
    return p;
}

// ------------------------------------------------------------------------------------------
// sequential matcher used only by the long-piece path: end of the match that starts at `pos`
// (a piece start) in the document [.., s1).  Same alternatives, same order as
// src/tekkenizer.rs:123.  Every lane computes the same value.
// ------------------------------------------------------------------------------------------
TK_DEV uint32_t tk_decode_at(const uint8_t* b, uint64_t p, uint64_t n, uint32_t* len) {
    uint32_t b0 = b[p];
    if (b0 < 0x80u) { *len = 1; return b0; }
    if ((b0 & 0xE0u) == 0xC0u && p + 1 < n && (b[p + 1] & 0xC0u) == 0x80u) {
        *len = 2; return ((b0 & 0x1Fu) << 6) | (b[p + 1] & 0x3Fu);
    }
    if ((b0 & 0xF0u) == 0xE0u && p + 2 < n && (b[p + 1] & 0xC0u) == 0x80u && (b[p + 2] & 0xC0u) == 0x80u) {
        *len = 3; return ((b0 & 0x0Fu) << 12) | ((uint32_t)(b[p + 1] & 0x3Fu) << 6) | (b[p + 2] & 0x3Fu);
    }
    if ((b0 & 0xF8u) == 0xF0u && p + 3 < n && (b[p + 1] & 0xC0u) == 0x80u && (b[p + 2] & 0xC0u) == 0x80u &&
        (b[p + 3] & 0xC0u) == 0x80u) {
        *len = 4;
        return ((b0 & 0x07u) << 18) | ((uint32_t)(b[p + 1] & 0x3Fu) << 12) | ((uint32_t)(b[p + 2] & 0x3Fu) << 6) |
               (b[p + 3] & 0x3Fu);
    }
    *len = 1;
    return 0xFFFFFFFFu;
}

TK_DEV uint32_t tk_class_at(const TkTablesView& t, const uint8_t* b, uint64_t p, uint64_t n, uint32_t* len,
                            uint32_t* cp) {
    uint32_t c = tk_decode_at(b, p, n, len);
    *cp = c;
    if (c < 0x80u) return tk_ascii_class(c);
    return tk_uc_class(t, c);
}

TK_DEV uint32_t tk_fold(uint32_t cp) {
    if (cp >= 'A' && cp <= 'Z') return cp + 32u;
    if (cp == 0x17Fu) return 's';
    return cp;
}

// Wave-cooperative fast-forward over a run of ASCII bytes of class `cls`, 64 bytes per step (called by all 64 lanes in
// convergence; every lane gets the same answer): the first position at or after q whose byte is not ASCII of that class
// (or n).  The per-byte loops of the matcher below are sequential dependent loads -- 32 KiB of letters took a wave 6 ms.
TK_DEV uint64_t tk_skip_ascii_run(const uint8_t* b, uint64_t q, uint64_t n, uint32_t cls) {
    const int lane = wv_lane();
    for (;;) {
        const uint64_t p = q + (uint64_t)lane;
        const uint32_t c = p < n ? (uint32_t)b[p] : 0x80u;
        const uint64_t stop = wv_ballot(c >= 0x80u || tk_ascii_class(c) != cls);
        if (stop) return q + (uint64_t)tk_ctz64(stop);
        q += 64;
    }
}

TK_DEV uint64_t tk_match_end(const TkTablesView& t, const uint8_t* b, uint64_t pos, uint64_t n) {
    uint32_t l0, l1, l2, c0, c1, c2;
    uint32_t k0 = tk_class_at(t, b, pos, n, &l0, &c0);
    if (c0 == '\'' && pos + 1 < n) {  // alt 1
        (void)tk_class_at(t, b, pos + 1, n, &l1, &c1);
        uint32_t f1 = tk_fold(c1);
        if (f1 == 's' || f1 == 't' || f1 == 'm' || f1 == 'd') return pos + 1 + l1;
        if ((f1 == 'r' || f1 == 'v' || f1 == 'l') && pos + 1 + l1 < n) {
            (void)tk_class_at(t, b, pos + 1 + l1, n, &l2, &c2);
            uint32_t f2 = tk_fold(c2);
            if (((f1 == 'r' || f1 == 'v') && f2 == 'e') || (f1 == 'l' && f2 == 'l')) return pos + 1 + l1 + l2;
        }
    }
    {  // alt 2
        bool crlf = (c0 == '\r' || c0 == '\n');
        bool with_prefix = !(crlf || k0 == TK_CLS_L || k0 == TK_CLS_N);
        for (int attempt = with_prefix ? 0 : 1; attempt < 2; ++attempt) {
            uint64_t p = attempt == 0 ? pos + l0 : pos;
            uint64_t q = p;
            while (q < n) {
                q = tk_skip_ascii_run(b, q, n, TK_CLS_L);           // ASCII letters 64 at a time, then char by char
                if (q >= n) break;
                uint32_t l, c;
                if (tk_class_at(t, b, q, n, &l, &c) != TK_CLS_L) break;
                q += l;
            }
            if (q > p) return q;
        }
    }
    if (k0 == TK_CLS_N) {  // alt 3
        uint64_t q = pos + l0;
        for (int cnt = 1; cnt < 3 && q < n; ++cnt) {
            uint32_t l, c;
            if (tk_class_at(t, b, q, n, &l, &c) != TK_CLS_N) break;
            q += l;
        }
        return q;
    }
    for (int attempt = (c0 == ' ') ? 0 : 1; attempt < 2; ++attempt) {  // alt 4
        uint64_t p = attempt == 0 ? pos + 1 : pos;
        uint64_t q = p;
        while (q < n) {
            q = tk_skip_ascii_run(b, q, n, TK_CLS_O);
            if (q >= n) break;
            uint32_t l, c;
            if (tk_class_at(t, b, q, n, &l, &c) != TK_CLS_O) break;
            q += l;
        }
        if (q > p) {
            while (q < n && (b[q] == '\r' || b[q] == '\n')) ++q;
            return q;
        }
    }
    // alts 5-7 over the maximal white-space run
    uint64_t e = pos, after_last_nl = 0, last_char = pos;
    bool has_nl = false;
    {
        // whole blocks of 64 ASCII white-space bytes at a time (the last CR / LF of a block from a ballot)
        const int lane = wv_lane();
        while (e + 64 <= n) {
            const uint32_t c = b[e + (uint64_t)lane];
            if (wv_ballot(c >= 0x80u || tk_ascii_class(c) != TK_CLS_S)) break;
            const uint64_t NL = wv_ballot(c == '\r' || c == '\n');
            if (NL) { has_nl = true; after_last_nl = e + (uint64_t)tk_msb64(NL) + 1; }
            last_char = e + 63;
            e += 64;
        }
    }
    while (e < n) {
        uint32_t l, c;
        if (tk_class_at(t, b, e, n, &l, &c) != TK_CLS_S) break;
        last_char = e;
        e += l;
        if (c == '\r' || c == '\n') { has_nl = true; after_last_nl = e; }
    }
    if (has_nl) return after_last_nl;
    if (e == n) return e;
    if (last_char > pos) return last_char;
    return e > pos ? e : pos + l0;
}

// ------------------------------------------------------------------------------------------
// SURVEY section 8 row f-3, opt-in: the `pattern` string that Mistral's tekken.json carries and the reference ignores
// (src/tekkenizer.rs:74; literal in tests/test_small_vocab.rs:62):
//   [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+ | [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*
//   | \p{N} | ' '?[^\s\p{L}\p{N}]+[\r\n/]* | \s*[\r\n]+ | \s+(?!\S) | \s+
// as a sequential matcher (leftmost-first alternation, greedy with backtracking), used by the piece-by-piece path of
// pass 2 only: first version of the opt-in, every document takes that path.  Classes from unicode_tables2.h:
// 1 U = Lu|Lt, 2 W = Ll, 3 X = Lm|Lo, 4 M, 5 N, 6 S, 0 anything else.  "Upper side" = U X M, "lower side" = W X M.
// ------------------------------------------------------------------------------------------
TK_DEV uint32_t tk_uc_class2(const TkTablesView& t, uint32_t cp) {
    if (cp >= 0x110000u) return 0u;
    const uint32_t blk = t.uc2_stage1[cp >> 7];
    const uint32_t w = t.uc2_stage2[blk * 16u + ((cp & 127u) >> 3)];
    return (w >> (4u * (cp & 7u))) & 15u;
}
TK_DEV uint32_t tk_class2_at(const TkTablesView& t, const uint8_t* b, uint64_t p, uint64_t n, uint32_t* len, uint32_t* cp) {
    const uint32_t c = tk_decode_at(b, p, n, len);
    *cp = c;
    return tk_uc_class2(t, c);
}
TK_DEV bool tk_c2_upper_side(uint32_t k) { return k == 1u || k == 3u || k == 4u; }
TK_DEV bool tk_c2_lower_side(uint32_t k) { return k == 2u || k == 3u || k == 4u; }

// word of the first / second alternative that starts at p (behind the optional prefix); p itself = no match
TK_DEV uint64_t tk_word2_at(const TkTablesView& t, const uint8_t* b, uint64_t p, uint64_t n, bool second_alt) {
    uint64_t q = p, last_both = ~0ull;           // last char of the upper-side run that is also lower-side (X or M)
    while (q < n) {
        uint32_t l, c;
        const uint32_t k = tk_class2_at(t, b, q, n, &l, &c);
        if (!tk_c2_upper_side(k)) break;
        if (tk_c2_lower_side(k)) last_both = q;
        q += l;
    }
    uint64_t from;
    if (second_alt) {                            // [upper side]+ [lower side]*
        if (q == p) return p;
        from = q;
    } else {                                     // [upper side]* [lower side]+ : give chars back until the lower side can start
        from = ~0ull;
        if (q < n) {
            uint32_t l, c;
            if (tk_c2_lower_side(tk_class2_at(t, b, q, n, &l, &c))) from = q;
        }
        if (from == ~0ull) from = last_both;
        if (from == ~0ull) return p;
    }
    uint64_t e = from;
    while (e < n) {
        uint32_t l, c;
        if (!tk_c2_lower_side(tk_class2_at(t, b, e, n, &l, &c))) break;
        e += l;
    }
    return e;
}

TK_DEV uint64_t tk_match_end2(const TkTablesView& t, const uint8_t* b, uint64_t pos, uint64_t n) {
    uint32_t l0, c0;
    const uint32_t k0 = tk_class2_at(t, b, pos, n, &l0, &c0);
    const bool letter = k0 == 1u || k0 == 2u || k0 == 3u;
    const bool prefix_ok = !(c0 == '\r' || c0 == '\n' || letter || k0 == 5u) && pos + l0 < n;
    for (int alt = 0; alt < 2; ++alt) {          // alternatives 1 and 2: with the prefix char first, then without
        if (prefix_ok) {
            const uint64_t e = tk_word2_at(t, b, pos + l0, n, alt == 1);
            if (e > pos + l0) return e;
        }
        const uint64_t e = tk_word2_at(t, b, pos, n, alt == 1);
        if (e > pos) return e;
    }
    if (k0 == 5u) return pos + l0;               // \p{N}: one digit
    for (int attempt = (c0 == ' ') ? 0 : 1; attempt < 2; ++attempt) {   // ' '?[^\s\p{L}\p{N}]+[\r\n/]*  (classes O and M)
        const uint64_t p = attempt == 0 ? pos + 1 : pos;
        uint64_t q = p;
        while (q < n) {
            uint32_t l, c;
            const uint32_t k = tk_class2_at(t, b, q, n, &l, &c);
            if (!(k == 0u || k == 4u)) break;
            q += l;
        }
        if (q > p) {
            while (q < n && (b[q] == '\r' || b[q] == '\n' || b[q] == '/')) ++q;
            return q;
        }
    }
    // the three white-space alternatives, over the maximal white-space run
    uint64_t e = pos, after_last_nl = 0, last_char = pos;
    bool has_nl = false;
    while (e < n) {
        uint32_t l, c;
        if (tk_class2_at(t, b, e, n, &l, &c) != 6u) break;
        last_char = e;
        e += l;
        if (c == '\r' || c == '\n') { has_nl = true; after_last_nl = e; }
    }
    if (e == pos) return pos + l0;               // (not reached for valid UTF-8)
    if (has_nl) return after_last_nl;
    if (e == n) return e;
    if (last_char > pos) return last_char;
    return e;
}

// ------------------------------------------------------------------------------------------
// pass 2, one piece [w0, e) of any length: whole-piece lookup (wave-wide polynomial hash for
// pieces of >= 9 bytes), and on a miss the wave-cooperative merge over scratch memory.
// ------------------------------------------------------------------------------------------
// Which merge a piece [w0, e) that is not a vocabulary key takes (wave-uniform; a pure function of the bytes and the
// thresholds, so every kernel that asks gets the same answer): 0 = one merge per step by this wave; 1 = the compacting
// rounds of tk_long.hip (repetitive pieces of >= long_min bytes: thousands of occurrences per rank -- 15x); 2 = the lazy
// rounds (pieces with many distinct pairs of >= 4 long_min bytes: 2.3x at 32 KiB, break-even near 4 KiB; below that the
// rounds' fixed cost of ~5 us loses against a dozen steps of 1.1 us -- but a job of the rounds has a CU of its own while the
// single-wave merge holds up the kernel that walks ALL handed-back documents, so the threshold is 2 long_min: Zipf shape,
// 500 k documents 10.0 -> 9.15 ms, 4 M documents 32.2 -> 32.6 ms).  long_force (tests): 1 / 2 = every long piece there.
TK_DEV bool tk_piece_repetitive(const TkEncodeArgs& a, int lane, uint64_t w0, uint64_t e);
TK_DEV uint32_t tk_piece_is_long(const TkEncodeArgs& a, int lane, uint64_t w0, uint64_t e) {
    const uint64_t len = e - w0;
    if (len < (uint64_t)a.long_min || len > TK_LONG_MAX) return 0u;
    if (a.long_force) return a.long_force == 1u ? 1u : 2u;
    if (tk_piece_repetitive(a, lane, w0, e)) return 1u;
    return len >= (uint64_t)(a.long_lazy_mul ? a.long_lazy_mul : 2u) * a.long_min ? 2u : 0u;
}

// whole-piece lookup of the piece [w0, e): its rank or TK_RANK_MAX (wave-uniform)
TK_DEV uint32_t tk_piece_lookup(const TkEncodeArgs& a, const TkPolyPow& pw, int lane, uint64_t w0, uint64_t e) {
    const TkTablesView& t = a.t;
    const uint64_t n = e - w0;
    uint32_t r = TK_RANK_MAX;
    if (n == 1) {
        r = a.bytes[w0];
    } else if (n <= 16) {
        uint32_t kk[4] = {0u, 0u, 0u, 0u};
        for (uint64_t j = 0; j < n; ++j) kk[j >> 2] |= (uint32_t)a.bytes[w0 + j] << (8 * (j & 3));
        r = tk_probe_key(t, kk[0], kk[1], kk[2], kk[3], (uint32_t)n);
    } else {
        // H = sum b_j P^(n-1-j), folded 64 bytes at a time: H = H * P^m + (sum b_i P^-i) * P^(m-1)
        uint32_t h1 = 0, h2 = 0;
        for (uint64_t c = w0; c < e; c += 64) {
            int m = (e - c) < 64 ? (int)(e - c) : 64;
            uint32_t bb = lane < m ? (uint32_t)a.bytes[c + lane] : 0u;
            uint32_t s1 = tk_wave_sum(bb * pw.ipw1, lane);
            uint32_t s2 = tk_wave_sum(bb * pw.ipw2, lane);
            uint32_t pm1 = wv_shfl(pw.pw1, m - 1), pm2 = wv_shfl(pw.pw2, m - 1);
            h1 = h1 * (pm1 * TK_POLY_P1) + s1 * pm1;
            h2 = h2 * (pm2 * TK_POLY_P2) + s2 * pm2;
        }
        r = tk_probe_long(t, h1, h2, (uint32_t)n, a.bytes + w0);
    }
    return wv_first(r);
}

TK_DEV void tk_piece_merge_coop(const TkEncodeArgs& a, int lane, uint64_t w0, uint64_t e, uint32_t* out, uint32_t& cursor, uint32_t* scratch,
                                bool light = false);

// Is the long piece [w0, e) REPETITIVE -- few distinct adjacent byte pairs among its first 512 bytes (runs of white space,
// of one symbol, of a short period)?  Such a piece merges many pairs per rank: the round-based workgroup merge
// (tk_long.hip) is up to 15x faster on it than one merge per step; on a piece with many distinct pairs (random letters: a
// dozen merges per rank) a round costs more than the dozen steps, and the piece stays with the single-wave merge
// (measured).  A pure function of the bytes: every kernel that asks gets the same answer.  Wave-uniform.
TK_DEV bool tk_piece_repetitive(const TkEncodeArgs& a, int lane, uint64_t w0, uint64_t e) {
    uint64_t seen = 0ull;                                        // pairs hashed into 64 buckets
    for (uint32_t k0 = 0; k0 < 512u; k0 += 64u) {
        const uint64_t p = w0 + k0 + (uint64_t)lane;
        if (p + 1 < e) {
            const uint32_t h = ((uint32_t)a.bytes[p] * 31u + (uint32_t)a.bytes[p + 1] * 7u) & 63u;
            seen |= 1ull << h;
        }
    }
    // OR over the lanes: one ballot per bucket would be 64 ballots; fold with shuffles instead (6 steps, two words)
    uint32_t lo = (uint32_t)seen, hi = (uint32_t)(seen >> 32);
    for (int d = 1; d < 64; d <<= 1) {
        lo |= wv_shfl(lo, lane ^ d);
        hi |= wv_shfl(hi, lane ^ d);
    }
    return (uint32_t)__builtin_popcount(wv_first(lo)) + (uint32_t)__builtin_popcount(wv_first(hi)) <= 20u;
}

TK_DEV void tk_piece_coop(const TkEncodeArgs& a, const TkPolyPow& pw, int lane, uint64_t w0, uint64_t e,
                          uint32_t* out, uint32_t& cursor, uint32_t* scratch) {
    const uint32_t r = tk_piece_lookup(a, pw, lane, w0, e);
    if (r != TK_RANK_MAX) {
        if (lane == 0) out[cursor] = r + a.t.num_special;
        cursor += 1;
        return;
    }
    tk_piece_merge_coop(a, lane, w0, e, out, cursor, scratch);
}

// the wave-cooperative merge of a piece that is not a vocabulary key (one merge per step, tiktoken's order)
// light (a compile-time constant at every call site): only the plain form with the block minima in scratch -- fewer registers,
// for the workgroup-per-document kernel, whose 1024-thread blocks leave a wave 128 of them
TK_DEV void tk_piece_merge_small(const TkTablesView& t, const uint8_t* bytes, int lane, uint64_t w0, uint32_t n, uint32_t* out,
                                 uint32_t& cursor);
TK_DEV void tk_piece_merge_coop(const TkEncodeArgs& a, int lane, uint64_t w0, uint64_t e, uint32_t* out, uint32_t& cursor, uint32_t* scratch,
                                bool light) {
    const TkTablesView& t = a.t;
    const uint64_t n = e - w0;
    if (n <= 256u && !a.dbg_mark) {            // short enough for the parts to live in registers: no scratch traffic at all
        tk_piece_merge_small(t, a.bytes, lane, w0, (uint32_t)n, out, cursor);
        return;
    }

    // ---- wave-cooperative merge over scratch: node[nn] = {tok, prk, nxt, prv} (16 B) | bmin[nb] (u64) ----
    // prk = rank of the pair (this part, next part); bmin[b] = min over the 64 nodes of block b of
    // (prk << 32 | position), so the global leftmost minimum is the min over bmin.  Per merge: one read
    // of bmin, node i, nodes j and p together, node k, both re-probes together, one store batch, and a
    // refresh of at most 3 block minima that patches the just-written ranks in registers (no wait on
    // the stores): about 5 dependent round trips.
    const uint32_t nn = (uint32_t)n;
    const uint32_t nb = (nn + 63u) / 64u;
    tk_u32x4* node = reinterpret_cast<tk_u32x4*>(scratch);
    uint32_t* bmin = scratch + 4u * nn;  // 2 words per block: lo = position, hi = rank
    if (a.dbg_mark && lane == 0) { a.dbg_mark[0] = 1u; a.dbg_mark[1] = nn; }
    for (uint32_t i = (uint32_t)lane; i < nn; i += 64u) {
        const uint32_t b0 = a.bytes[w0 + i];
        const uint32_t b1 = (i + 1u < nn) ? (uint32_t)a.bytes[w0 + i + 1u] : 0u;
        tk_u32x4 v;
        v.x = b0;
        v.y = (i + 1u < nn) ? t.pair2[b0 | (b1 << 8)] : TK_RANK_MAX;
        v.z = i + 1u;
        v.w = i ? i - 1u : TK_NONE;
        node[i] = v;
    }
    wv_sync();
    if (!light && nb <= 512u) {
        // ---- up to 32 KiB: the block minima live in REGISTERS (lane l holds blocks l, l + 64, ...: 8 slots), every node
        // carries the token of its successor (tokn), and the rows of the (at most three) blocks whose minimum changes are
        // requested together with nodes j and p.  Dependent round trips per merge: node i | nodes j, p, tokn[j], rows |
        // the two re-probes | the stores' acknowledgement -- four instead of seven.
        uint32_t* tokn = bmin + 2u * nb;
        for (uint32_t i = (uint32_t)lane; i < nn; i += 64u) tokn[i] = (i + 1u < nn) ? (uint32_t)a.bytes[w0 + i + 1u] : 0u;
        uint64_t bm[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) bm[q] = ~0ull;
        for (uint32_t blk = 0; blk < nb; ++blk) {
            const uint32_t x = blk * 64u + (uint32_t)lane;
            const uint32_t rk = x < nn ? node[x].y : TK_RANK_MAX;
            const uint64_t m = tk_wave_min_key(rk, x);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((uint32_t)q == (blk >> 6) && (uint32_t)lane == (blk & 63u)) bm[q] = m;
        }
        wv_sync();
        for (uint32_t merges = 0; merges < nn; ++merges) {  // at most nn - 1 merges can happen
            uint64_t best = bm[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) best = bm[q] < best ? bm[q] : best;
            best = tk_wave_min_key((uint32_t)(best >> 32), (uint32_t)best);   // (~0: rank MAX)
            if (best == ~0ull) break;                        // wave-uniform
            const uint32_t i = (uint32_t)best, rr = (uint32_t)(best >> 32);
            const tk_u32x4 Ni = node[i];
            const uint32_t j = Ni.z, p = Ni.w;
            const uint32_t bi = i / 64u, bj = j / 64u, bp = (p != TK_NONE) ? p / 64u : bi;
            // one round trip: the two neighbours, the token behind j, the rank rows of the blocks to refresh
            const tk_u32x4 Nj = node[j];
            const tk_u32x4 Np = node[p != TK_NONE ? p : i];
            const uint32_t tok_k = tokn[j];
            uint32_t row[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const uint32_t blk = q == 0 ? bi : q == 1 ? bj : bp;
                const uint32_t x = blk * 64u + (uint32_t)lane;
                row[q] = x < nn ? node[x].y : TK_RANK_MAX;
            }
            const uint32_t k = Nj.z;
            uint32_t new_i, new_p;
            tk_probe_pair_x2(t, rr, tok_k, Np.x, rr, new_i, new_p);
            if (k >= nn) new_i = TK_RANK_MAX;
            if (p == TK_NONE) new_p = TK_RANK_MAX;
            wv_sync();  // every lane has read the nodes before lane 0 rewrites them
            if (lane == 0) {
                tk_u32x4 v;
                v.x = rr; v.y = new_i; v.z = k; v.w = p;
                node[i] = v;
                tokn[i] = tok_k;                             // i's successor is now k
                v.x = TK_DEAD; v.y = TK_RANK_MAX; v.z = Nj.z; v.w = Nj.w;
                node[j] = v;
                if (k < nn) node[k].w = i;
                if (p != TK_NONE) { node[p].y = new_p; tokn[p] = rr; }
            }
            // refresh the block minima of i, j and p from the rows, with the three just-written ranks patched in
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const uint32_t blk = q == 0 ? bi : q == 1 ? bj : bp;
                if ((q == 1 && bj == bi) || (q == 2 && (bp == bi || bp == bj))) continue;   // (wave-uniform)
                const uint32_t x = blk * 64u + (uint32_t)lane;
                uint32_t rk = row[q];
                if (x == i) rk = new_i;
                if (x == j) rk = TK_RANK_MAX;
                if (p != TK_NONE && x == p) rk = new_p;
                const uint64_t m = tk_wave_min_key(rk, x);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if ((uint32_t)u == (blk >> 6) && (uint32_t)lane == (blk & 63u)) bm[u] = m;
            }
            wv_sync();  // this merge's stores are visible before the next merge's loads
        }
    } else {
        for (uint32_t blk = 0; blk < nb; ++blk) {
            const uint32_t x = blk * 64u + (uint32_t)lane;
            const uint32_t rk = x < nn ? node[x].y : TK_RANK_MAX;
            const uint64_t key = rk == TK_RANK_MAX ? ~0ull : (((uint64_t)rk << 32) | x);
            const uint64_t m = tk_wave_min64(key, lane);
            if (lane == 0) { bmin[2 * blk] = (uint32_t)m; bmin[2 * blk + 1] = (uint32_t)(m >> 32); }
        }
        wv_sync();
        for (uint32_t merges = 0; merges < nn; ++merges) {  // at most nn - 1 merges can happen
            if (a.dbg_mark && lane == 0) a.dbg_mark[2] = merges;
            uint64_t best = ~0ull;
            for (uint32_t bq = (uint32_t)lane; bq < nb; bq += 64u) {
                const uint64_t v = ((uint64_t)bmin[2 * bq + 1] << 32) | bmin[2 * bq];
                best = v < best ? v : best;
            }
            best = tk_wave_min64(best, lane);
            if (wv_ballot(best != ~0ull) == 0) break;  // decided on a ballot => a scalar branch
            const uint32_t i = (uint32_t)best, rr = (uint32_t)(best >> 32);
            const tk_u32x4 Ni = node[i];
            const uint32_t j = Ni.z, p = Ni.w;
            const tk_u32x4 Nj = node[j];
            const tk_u32x4 Np = node[p != TK_NONE ? p : i];
            const uint32_t k = Nj.z;
            const uint32_t tok_k = node[k < nn ? k : i].x;
            uint32_t new_i, new_p;
            tk_probe_pair_x2(t, rr, tok_k, Np.x, rr, new_i, new_p);
            if (k >= nn) new_i = TK_RANK_MAX;
            if (p == TK_NONE) new_p = TK_RANK_MAX;
            wv_sync();  // every lane has read the nodes before lane 0 rewrites them
            if (lane == 0) {
                tk_u32x4 v;
                v.x = rr; v.y = new_i; v.z = k; v.w = p;
                node[i] = v;
                v.x = TK_DEAD; v.y = TK_RANK_MAX; v.z = Nj.z; v.w = Nj.w;
                node[j] = v;
                if (k < nn) node[k].w = i;
                if (p != TK_NONE) node[p].y = new_p;
            }
            // refresh the block minima of i, j and p; the three just-written ranks are patched in registers
            const uint32_t bi = i / 64u, bj = j / 64u, bp = (p != TK_NONE) ? p / 64u : bi;
            const bool skip1 = wv_ballot(bj != bi) == 0;
            const bool skip2 = wv_ballot(bp != bi && bp != bj) == 0;
            for (int q = 0; q < 3; ++q) {
                if ((q == 1 && skip1) || (q == 2 && skip2)) continue;
                const uint32_t blk = q == 0 ? bi : q == 1 ? bj : bp;
                const uint32_t x = blk * 64u + (uint32_t)lane;
                uint32_t rk = x < nn ? node[x].y : TK_RANK_MAX;
                if (x == i) rk = new_i;
                if (x == j) rk = TK_RANK_MAX;
                if (p != TK_NONE && x == p) rk = new_p;
                const uint64_t key = rk == TK_RANK_MAX ? ~0ull : (((uint64_t)rk << 32) | x);
                const uint64_t m = tk_wave_min64(key, lane);
                if (lane == 0) { bmin[2 * blk] = (uint32_t)m; bmin[2 * blk + 1] = (uint32_t)(m >> 32); }
            }
            wv_sync();  // this merge's stores are visible before the next merge's loads
        }
    }
    if (a.dbg_mark && lane == 0) a.dbg_mark[0] = 5u;
    for (uint32_t blk = 0; blk < nb; ++blk) {
        const uint32_t x = blk * 64u + (uint32_t)lane;
        const uint32_t tv = x < nn ? node[x].x : TK_DEAD;
        const uint64_t am = wv_ballot(tv != TK_DEAD);
        if (tv != TK_DEAD) out[cursor + (uint32_t)tk_popc64(am & tk_lowmask(lane))] = tv + t.num_special;
        cursor += (uint32_t)tk_popc64(am);
    }
    if (a.dbg_mark && lane == 0) a.dbg_mark[0] = 6u;
}

// ------------------------------------------------------------------------------------------
// The byte-pair merge of ONE piece of up to 256 bytes by one wave with the parts in REGISTERS: position p = 64 j + lane
// (j = 0..3) holds the id of the part that starts there and the rank of its pair with the next part; four wave-uniform
// 64-bit masks say which positions are live.  A merge (tiktoken's order: the leftmost smallest rank) is one wave
// minimum, a few bit scans over the masks for the neighbours, two readlanes for their ids, ONE round trip for the two
// pairs the merge creates, and a handful of predicated moves -- no memory but the probes.  (tk_piece_merge_coop keeps
// the parts of a piece of ANY length in scratch and refreshes block minima from 64-lane rows: three wave-wide gathers
// per merge, which is what bounded it on pieces this short -- ~3 merges per microsecond and CU whatever the occupancy.)
// ------------------------------------------------------------------------------------------
TK_DEV uint32_t tkp_sel4(const uint32_t* v, uint32_t j) { return j == 0u ? v[0] : j == 1u ? v[1] : j == 2u ? v[2] : v[3]; }
TK_DEV uint64_t tkp_sel4(const uint64_t* v, uint32_t j) { return j == 0u ? v[0] : j == 1u ? v[1] : j == 2u ? v[2] : v[3]; }
// first live position above pos (256: none) / last live position below pos (TK_NONE: none); A, pos wave-uniform
TK_DEV uint32_t tkp_next(const uint64_t* A, uint32_t pos) {
    const uint32_t j = pos >> 6, l = pos & 63u;
    const uint64_t m = l < 63u ? (tkp_sel4(A, j) >> (l + 1u)) << (l + 1u) : 0ull;
    if (m) return 64u * j + (uint32_t)__builtin_ctzll(m);
    for (uint32_t jj = j + 1u; jj < 4u; ++jj) {
        const uint64_t a = tkp_sel4(A, jj);
        if (a) return 64u * jj + (uint32_t)__builtin_ctzll(a);
    }
    return 256u;
}
TK_DEV uint32_t tkp_prev(const uint64_t* A, uint32_t pos) {
    const uint32_t j = pos >> 6, l = pos & 63u;
    const uint64_t m = tkp_sel4(A, j) & ((1ull << l) - 1ull);
    if (m) return 64u * j + 63u - (uint32_t)__builtin_clzll(m);
    for (uint32_t jj = j; jj-- > 0u;) {
        const uint64_t a = tkp_sel4(A, jj);
        if (a) return 64u * jj + 63u - (uint32_t)__builtin_clzll(a);
    }
    return TK_NONE;
}
TK_DEV void tk_piece_merge_small(const TkTablesView& t, const uint8_t* bytes, int lane, uint64_t w0, uint32_t n, uint32_t* out,
                                 uint32_t& cursor) {
    uint32_t tok[4], rk[4];
    uint64_t A[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t p = 64u * j + (uint32_t)lane;
        const uint32_t b0 = p < n ? (uint32_t)bytes[w0 + p] : 0u;
        const uint32_t b1 = p + 1u < n ? (uint32_t)bytes[w0 + p + 1u] : 0u;
        tok[j] = b0;
        rk[j] = p + 1u < n ? t.pair2[b0 | (b1 << 8)] : TK_RANK_MAX;
        A[j] = wv_first64(n >= 64u * (j + 1u) ? ~0ull : n <= 64u * j ? 0ull : ((1ull << (n - 64u * j)) - 1ull));
    }
    for (uint32_t merges = 0; merges < n; ++merges) {          // at most n - 1 merges can happen
        uint32_t key = 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t k = rk[j] == TK_RANK_MAX ? 0xFFFFFFFFu : ((rk[j] << 8) | (64u * j + (uint32_t)lane));
            key = k < key ? k : key;
        }
        key = wv_first(wv_min_u32(key));
        if (key == 0xFFFFFFFFu) break;                         // wave-uniform
        const uint32_t i = key & 255u, rr = key >> 8;
        // (readfirstlane on everything that is the same in all lanes: the mask arithmetic then runs on the scalar unit)
        const uint32_t jn = wv_first(tkp_next(A, i));           // exists: the pair at i was live
        const uint32_t nn = wv_first(tkp_next(A, jn)), p = wv_first(tkp_prev(A, i));
        const uint32_t tok_nn = nn < 256u ? wv_shfl(tkp_sel4(tok, nn >> 6), (int)(nn & 63u)) : 0u;
        const uint32_t tok_p = p != TK_NONE ? wv_shfl(tkp_sel4(tok, p >> 6), (int)(p & 63u)) : 0u;
        uint32_t new_i, new_p;
        tk_probe_pair_x2(t, rr, tok_nn, tok_p, rr, new_i, new_p);   // (every lane asks for the same two buckets: one request each)
        if (nn >= 256u) new_i = TK_RANK_MAX;
        if (p == TK_NONE) new_p = TK_RANK_MAX;
        // the part at i takes the merged id and its new pair rank, the part at jn dies, the part at p gets its new pair rank
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t pos = 64u * j + (uint32_t)lane;
            if (pos == i) { tok[j] = rr; rk[j] = new_i; }
            if (pos == jn) rk[j] = TK_RANK_MAX;
            if (p != TK_NONE && pos == p) rk[j] = new_p;
            if ((jn >> 6) == j) A[j] = wv_first64(A[j] & ~(1ull << (jn & 63u)));
        }
    }
    uint32_t base = cursor;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        if ((A[j] >> lane) & 1ull) out[base + (uint32_t)tk_popc64(A[j] & tk_lowmask(lane))] = tok[j] + t.num_special;
        base += (uint32_t)tk_popc64(A[j]);
    }
    cursor = base;
}

// ------------------------------------------------------------------------------------------
// one document
// ------------------------------------------------------------------------------------------
// MODE 0: pass 1 (a document whose first unfinished piece does not end inside a window is handed
//         to pass 2: return false);  MODE 2: split only (tk_split_batch), long pieces are skipped
//         over with the sequential matcher.
template <int MODE>
TK_DEV bool tk_encode_doc(const TkEncodeArgs& a, uint64_t d, int lane, const TkPolyPow& pw) {
    const TkTablesView& t = a.t;
    const uint64_t s0 = wv_first64(a.doc_offs[d]), s1 = wv_first64(a.doc_offs[d + 1]);
    uint32_t* out = a.staging + s0 + 2 * d;
    uint32_t cursor = 0;
    if (a.add_bos) {
        if (lane == 0) out[0] = t.bos_id;
        cursor = 1;
    }
    uint64_t w0 = s0;
    // software prefetch: `cur` = bytes [w0, w0+64), `nxt` = bytes [w0+64, w0+128) (0 past the document end).
    // The next window starts at w0 + region_end <= w0 + 64, so its bytes are a lane rotation of
    // (cur, nxt); the global load issued for the new `nxt` is only consumed one window later.
    uint32_t cur = (w0 + (uint64_t)lane < s1) ? (uint32_t)a.bytes[w0 + lane] : 0u;
    uint32_t nxt = (w0 + 64 + (uint64_t)lane < s1) ? (uint32_t)a.bytes[w0 + 64 + lane] : 0u;
    while (w0 < s1) {
        const uint64_t rem = s1 - w0;
        int nv = rem < 64 ? (int)rem : 64;
        const bool at_end = rem <= 64;

        // ---- 1. load + classify ---------------------------------------------------------
        uint32_t b0 = lane < nv ? cur : 0u;
        const bool ascii = wv_ballot(b0 >= 0x80u) == 0;  // wave-uniform: the common case takes the scalar rules
        if (!at_end && !ascii) {
            // a char cut by the window end has an unknown class: leave it to the next window
            uint32_t nominal = b0 < 0xC0u ? 1u : b0 < 0xE0u ? 2u : b0 < 0xF0u ? 3u : b0 < 0xF8u ? 4u : 1u;
            uint64_t csraw = wv_ballot(lane < nv && ((b0 & 0xC0u) != 0x80u || lane == 0));
            int last = tk_msb64(csraw);
            uint32_t nl = wv_first(wv_shfl(nominal, last));
            if (last + (int)nl > nv && last > 0) nv = last;
            if (lane >= nv) b0 = 0u;
        }
        const bool valid = lane < nv;
        const uint32_t b1 = wv_up1(b0), b2 = wv_up1(b1), b3 = wv_up1(b2);
        uint64_t PS;  // piece starts decided by this window
        int fu;       // first position whose decision unseen bytes could change
        if (ascii) {
            // ---- 2a. ASCII window: one byte = one char, the rules are pure 64-bit mask algebra on
            //          SGPR pairs (no per-lane work, no divergent branch).  Same rules as 2b below.
            const uint32_t cls = tk_ascii_class(b0);
            const uint64_t VAL = tk_lowmask(nv);
            // bytes past nv are 0 (class O), so single compares are already masked by validity
            const uint64_t mL = wv_ballot(cls == TK_CLS_L);
            const uint64_t mN = wv_ballot(cls == TK_CLS_N);
            const uint64_t mS = wv_ballot(cls == TK_CLS_S);
            const uint64_t mO = VAL & ~(mL | mN | mS);
            const uint64_t NLm = wv_ballot(b0 == 10u) | wv_ballot(b0 == 13u);
            const uint64_t SPm = wv_ballot(b0 == 0x20u);
            const uint64_t APm = wv_ballot(b0 == 0x27u);
            uint64_t CEND = 0;
            if (APm) {  // alt 1 fires only where a match starts at the apostrophe
                const uint32_t f1 = b1 | 0x20u, f2 = b2 | 0x20u;
                const bool c2 = f1 == 's' || f1 == 't' || f1 == 'm' || f1 == 'd';
                const bool c3 = ((f1 == 'r' || f1 == 'v') && f2 == 'e') || (f1 == 'l' && f2 == 'l');
                const uint64_t ok = APm & ~((mO | SPm) << 1);
                CEND = ((wv_ballot(c2) & ok) << 2) | ((wv_ballot(c3 && !c2) & ok) << 3);
            }
            const uint64_t L1 = mL << 1, O1 = mO << 1;
            const uint64_t Lst = mL & ~L1;
            const uint64_t psL = (mL & L1 & CEND) | (Lst & ((mN | NLm) << 1)) | (Lst & O1 & ((mO | SPm) << 2));
            const uint64_t psO = mO & ~O1 & ~(SPm << 1);
            uint64_t psN = mN & ~(mN << 1);
            {
                const uint64_t M3 = mN & (mN << 1) & (mN << 2) & (mN << 3);  // q-3..q all numbers
                if (M3) {  // a run of >= 4 numbers: starts every 3 chars (\p{N}{1,3}), by prefix doubling
                    const uint64_t M6 = M3 & (M3 << 3), M12 = M6 & (M6 << 6), M24 = M12 & (M12 << 12);
                    const uint64_t M48 = M24 & (M24 << 24);
                    psN |= (psN << 3) & M3;
                    psN |= (psN << 6) & M6;
                    psN |= (psN << 12) & M12;
                    psN |= (psN << 24) & M24;
                    psN |= (psN << 48) & M48;
                }
            }
            // white space: CR/LF absorbed by alt 4 directly after an O run (carry ripple)
            const uint64_t seeds = NLm & O1;
            const uint64_t ABS = seeds ? (((NLm + seeds) ^ NLm) & NLm) : 0ull;
            const uint64_t SPR = mS & ~ABS;
            const uint64_t NLp = NLm & SPR;
            uint64_t Z = 0;  // positions up to and including the last CR/LF of their run (\s*[\r\n]+)
            if (NLp) {
                // T = the part of each run above its last CR/LF, found in bit-reversed space by a
                // ripple from the run tops through the non-CR/LF bits
                const uint64_t r = wv_brev64(SPR), s = wv_brev64(NLp);
                const uint64_t u = r & ~s;
                const uint64_t tops = r & ~(r << 1) & u;
                const uint64_t T = wv_brev64(((u + tops) ^ u) & u);
                Z = SPR & ~T;
            }
            const uint64_t psS = (SPR & ~(SPR << 1)) | ((Z << 1) & SPR & ~Z) | (SPR & ~(SPR >> 1) & ~Z & (VAL >> 1));
            PS = (psL | psN | psO | psS | 1ull) & VAL;
            fu = nv;
            if (!at_end && tk_bit(SPR, nv - 1)) {  // the run that touches the window end is undecided past its first char
                const uint64_t below = ~SPR & VAL;
                const int rs = below ? tk_msb64(below) + 1 : 0;
                fu = rs + 1 < nv ? rs + 1 : nv;
            }
        } else {
            // ---- 2b. general window (multi-byte code points): per-lane rules
            const bool lead = (b0 & 0xC0u) != 0x80u || lane == 0;
            uint32_t cls = TK_CLS_O, clen = 1;
            if (b0 < 0x80u) {
                cls = tk_ascii_class(b0);
            } else if (lead && b0 >= 0xC0u) {
                uint32_t cp = 0xFFFFFFFFu;
                if (b0 < 0xE0u) {
                    if ((b1 & 0xC0u) == 0x80u) { cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu); clen = 2; }
                } else if (b0 < 0xF0u) {
                    if ((b1 & 0xC0u) == 0x80u && (b2 & 0xC0u) == 0x80u) {
                        cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu); clen = 3;
                    }
                } else if (b0 < 0xF8u) {
                    if ((b1 & 0xC0u) == 0x80u && (b2 & 0xC0u) == 0x80u && (b3 & 0xC0u) == 0x80u) {
                        cp = ((b0 & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu); clen = 4;
                    }
                }
                if (cp != 0xFFFFFFFFu) cls = tk_uc_class(t, cp);
            }
            const uint64_t CS = wv_ballot(valid && lead);
            {
                // continuation bytes take the class of their lead byte so that runs are contiguous
                const uint32_t c1 = wv_dn1(cls), c2 = wv_dn1(c1), c3 = wv_dn1(c2);
                if (!lead) {
                    int dist = lane - tk_msb64(CS & tk_lowmask(lane));
                    cls = dist == 1 ? c1 : dist == 2 ? c2 : dist == 3 ? c3 : TK_CLS_O;
                }
            }
            const uint64_t mL = wv_ballot(valid && cls == TK_CLS_L);
            const uint64_t mN = wv_ballot(valid && cls == TK_CLS_N);
            const uint64_t mS = wv_ballot(valid && cls == TK_CLS_S);
            const uint64_t mO = wv_ballot(valid && cls == TK_CLS_O);
            const uint64_t NLm = wv_ballot(valid && (b0 == 10u || b0 == 13u));
            const uint64_t SPm = wv_ballot(valid && b0 == 0x20u);

            // (local piece-start rules, per lane) ---------------------------------------------------
            // alt 1 (?i:'s|'t|'re|'ve|'m|'ll|'d): fires only where a match starts at the apostrophe
            uint32_t ce = 0;
            if (b0 == 0x27u) {
                const uint32_t f1 = b1 | 0x20u, f2 = b2 | 0x20u;
                if (f1 == 's' || f1 == 't' || f1 == 'm' || f1 == 'd') ce = 2;
                else if (b1 == 0xC5u && b2 == 0xBFu) ce = 3;  // U+017F folds to 's'
                else if (((f1 == 'r' || f1 == 'v') && f2 == 'e') || (f1 == 'l' && f2 == 'l')) ce = 3;
            }
            const bool fire = ce && (lane == 0 || (!tk_bit(mO, lane - 1) && !tk_bit(SPm, lane - 1)));
            const uint64_t F2 = wv_ballot(valid && fire && ce == 2);
            const uint64_t F3 = wv_ballot(valid && fire && ce == 3);
            const uint64_t CEND = (F2 << 2) | (F3 << 3);
            // CR/LF absorbed by alt 4's trailing [\r\n]*: leading CR/LF of a white-space run that
            // directly follows a class-O byte (carry ripple through the CR/LF run)
            const uint64_t seeds = NLm & (mO << 1);
            const uint64_t ABS = ((NLm + seeds) ^ NLm) & NLm;
            const uint64_t SPR = mS & ~ABS;  // effective white space
            const uint64_t Nst = mN & ~(mN << 1);

            bool st = false, unc = false;
            if (valid && lead) {
                if (lane == 0) {
                    st = true;
                } else {
                    const int pm = lane - 1;
                    const bool pL = tk_bit(mL, pm), pN = tk_bit(mN, pm), pS = tk_bit(mS, pm), pO = tk_bit(mO, pm);
                    if (cls == TK_CLS_L) {
                        if (pL) st = tk_bit(CEND, lane);             // only right after a contraction
                        else if (pN) st = true;
                        else if (pS) st = tk_bit(NLm, pm);           // other white space is alt 2's prefix
                        else {
                            const int qs = tk_msb64(CS & tk_lowmask(lane));
                            const bool runstart = qs == 0 || !tk_bit(mO, qs - 1);
                            st = !runstart || (qs > 0 && tk_bit(SPm, qs - 1));
                        }
                    } else if (cls == TK_CLS_N) {
                        if (!pN) st = true;
                        else {
                            const int rs = tk_msb64(Nst & tk_lowmask(lane + 1));
                            const int k = tk_popc64(CS & tk_lowmask(lane) & ~tk_lowmask(rs));
                            st = (k % 3) == 0;                       // \p{N}{1,3}
                        }
                    } else if (cls == TK_CLS_O) {
                        st = !pO && !tk_bit(SPm, pm);                // ' ?' of alt 4 takes one U+0020
                    } else {
                        (void)pS;
                        if (tk_bit(ABS, lane)) st = false;
                        else if (!tk_bit(SPR, pm)) st = true;        // start of the effective run
                        else {
                            const uint64_t x = SPR >> lane;
                            const int rl = tk_ctz64(~x);
                            const int e = lane + rl;
                            if (!at_end && e >= nv) unc = true;      // run reaches unseen bytes
                            else {
                                const bool later_nl = ((NLm >> lane) & tk_lowmask(rl)) != 0;
                                if (later_nl) st = false;            // inside \s*[\r\n]+
                                else if (tk_bit(NLm, pm)) st = true; // first char after the last CR/LF
                                else st = (lane + (int)clen == e) && (e < nv);  // \s+(?!\S) leaves the last char
                            }
                        }
                    }
                }
            }
            PS = wv_ballot(st);
            {
                const uint64_t UNC = wv_ballot(unc);
                fu = UNC ? tk_ctz64(UNC) : nv;
            }

        }

        // ---- 3. how far may this window commit? -------------------------------------------
        int region_end;
        uint64_t PSp;
        if (at_end) {
            region_end = nv;
            PSp = PS;
        } else {
            const uint64_t cert = PS & tk_lowmask(fu);
            const int estar = tk_msb64(cert);
            if (estar == 0) {
                // the first piece does not end inside the window
                if (MODE != 2) return false;
                const uint64_t e = wv_first64(tk_match_end(t, a.bytes, w0, s1));
                for (uint64_t q = w0 + (uint64_t)lane; q < e; q += 64) a.dbg_starts[q] = (q == w0);
                w0 = e;
                cur = (w0 + (uint64_t)lane < s1) ? (uint32_t)a.bytes[w0 + lane] : 0u;
                nxt = (w0 + 64 + (uint64_t)lane < s1) ? (uint32_t)a.bytes[w0 + 64 + lane] : 0u;
                continue;
            }
            region_end = estar;
            PSp = cert & ~(1ull << estar);
        }
        const bool inreg = lane < region_end;
        // slide: rotate (cur, nxt) by region_end lanes and start the load of the following 64 bytes
        {
            const int idx = region_end + lane;
            const uint32_t ra = wv_shfl(cur, idx & 63), rb = wv_shfl(nxt, idx & 63);
            const uint64_t nw = w0 + (uint64_t)region_end;
            cur = idx < 64 ? ra : rb;
            nxt = (nw + 64 + (uint64_t)lane < s1) ? (uint32_t)a.bytes[nw + 64 + lane] : 0u;
        }
        if (MODE == 2) {
            if (inreg) a.dbg_starts[w0 + lane] = tk_bit(PSp, lane);
            w0 += (uint64_t)region_end;
            continue;
        }

        // ---- 4. whole-piece lookup ----------------------------------------------------------
        const uint64_t E = PSp | (region_end < 64 ? (1ull << region_end) : 0ull);
        const int ps = tk_msb64(PSp & tk_lowmask(lane + 1));
        int pe;
        {
            const uint64_t y = lane < 63 ? (E >> (lane + 1)) : 0ull;
            pe = y ? lane + 1 + tk_ctz64(y) : 64;
        }
        const bool isstart = inreg && tk_bit(PSp, lane);
        const int len = pe - lane;
        // exact-key material: every lane masks its own 4 bytes to its piece (bytes of the NEXT piece are
        // zeroed at the source), the start lane then gathers the dwords at +4, +8, +12
        const uint32_t d4 = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        const uint32_t d4m = len >= 4 ? d4 : (d4 & ((1u << (8 * len)) - 1u));  // len = bytes left in my piece (>= 1)
        const uint32_t g1 = wv_shfl(d4m, (lane + 4) & 63), g2 = wv_shfl(d4m, (lane + 8) & 63);
        const uint32_t g3 = wv_shfl(d4m, (lane + 12) & 63);

        uint32_t h1 = 0, h2 = 0;
        const uint64_t LONGM = wv_ballot(isstart && len >= 17);
        if (LONGM) {
            const uint32_t t1 = b0 * pw.ipw1, t2 = b0 * pw.ipw2;
            uint32_t f1 = t1, f2 = t2;
            for (int dd = 1; dd < 64; dd <<= 1) {
                const uint32_t o1 = wv_shfl(f1, lane >= dd ? lane - dd : lane);
                const uint32_t o2 = wv_shfl(f2, lane >= dd ? lane - dd : lane);
                if (lane >= dd) { f1 += o1; f2 += o2; }
            }
            const int el = pe - 1;
            const uint32_t fe1 = wv_shfl(f1, el), fe2 = wv_shfl(f2, el);
            const uint32_t pe1 = wv_shfl(pw.pw1, el), pe2 = wv_shfl(pw.pw2, el);
            h1 = (fe1 - (f1 - t1)) * pe1;
            h2 = (fe2 - (f2 - t2)) * pe2;
        }

        uint32_t tokv = TK_RANK_MAX;
        if (a.dbg_ablate & 1) tokv = 7u;
        else if (isstart) {
            if (len == 1) {
                tokv = b0;  // rank of a single byte is the byte (src/tekkenizer.rs:793-798)
            } else if (len <= 16) {
                // exact key: the piece bytes themselves, zero padded to 16 (lane + 4j is in my piece iff len > 4j)
                const uint32_t k0 = d4m, k1 = len > 4 ? g1 : 0u, k2 = len > 8 ? g2 : 0u, k3 = len > 12 ? g3 : 0u;
                tokv = tk_probe_key(t, k0, k1, k2, k3, (uint32_t)len);
            } else {
                tokv = tk_probe_long(t, h1, h2, (uint32_t)len, a.bytes + w0 + lane);
            }
        }
        const uint64_t Hm = wv_ballot(isstart && tokv != TK_RANK_MAX);
        const uint64_t Mi = PSp & ~Hm;

        // ---- 5. in-window byte-pair merge for the pieces that missed ------------------------
        uint64_t A = 0;
        uint32_t tok = b0;
        const bool inmiss = inreg && tk_bit(Mi, ps);
        if (Mi && !(a.dbg_ablate & 2)) {
            A = wv_ballot(inmiss);
            uint32_t prank = TK_RANK_MAX;
            if (inmiss && lane + 1 < pe) prank = t.pair2[b0 | (b1 << 8)];
            const int sl = pe - 1 < 63 ? pe - 1 : 63;
            for (int round = 0; round < 64; ++round) {  // a window holds at most 63 merges per piece
                const uint32_t key = prank == TK_RANK_MAX ? 0xFFFFFFFFu : ((prank << 6) | (uint32_t)lane);
                uint32_t m = key;
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t o = wv_shfl(m, lane >= dd ? lane - dd : lane);
                    if (lane >= dd && lane - dd >= ps) m = o < m ? o : m;
                }
                const uint32_t segmin = wv_shfl(m, sl);
                const bool winner = inmiss && key != 0xFFFFFFFFu && key == segmin;
                const uint64_t W = wv_ballot(winner);
                if (!W) break;
                const bool alive = tk_bit(A, lane);
                const uint64_t below = A & tk_lowmask(lane);
                const bool dead = alive && below && tk_bit(W, tk_msb64(below));
                const uint64_t Dm = wv_ballot(dead);
                A &= ~Dm;
                if (winner) tok = prank;          // merged token id == rank of the pair
                if (dead) prank = TK_RANK_MAX;
                const uint64_t z = lane < 63 ? (A >> (lane + 1)) : 0ull;
                const int na = z ? lane + 1 + tk_ctz64(z) : 64;
                const bool has_next = na < pe;
                const uint32_t tn = wv_shfl(tok, na < 64 ? na : lane);
                const bool need = tk_bit(A, lane) && (winner || (has_next && tk_bit(W, na)));
                if (need) prank = has_next ? tk_probe_pair(t, tok, tn) : TK_RANK_MAX;
            }
        }

        // ---- 6. emit --------------------------------------------------------------------------
        const bool hitstart = isstart && tokv != TK_RANK_MAX;
        const bool istok = hitstart || (inmiss && tk_bit(A, lane));
        const uint64_t Tm = wv_ballot(istok);
        if (istok && !(a.dbg_ablate & 4)) {
            out[cursor + (uint32_t)tk_popc64(Tm & tk_lowmask(lane))] = (hitstart ? tokv : tok) + t.num_special;
        }
        cursor += (uint32_t)tk_popc64(Tm);
        w0 += (uint64_t)region_end;
    }
    if (MODE != 2) {
        if (a.add_eos) {
            if (lane == 0) out[cursor] = t.eos_id;
            cursor += 1;
        }
        if (lane == 0) a.counts[d] = cursor;
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// pass 2: one deferred document, piece by piece (sequential matcher + tk_piece_coop)
// ------------------------------------------------------------------------------------------
TK_DEV void tk_encode_doc_seq(const TkEncodeArgs& a, uint64_t d, int lane, const TkPolyPow& pw, uint32_t* scratch) {
    const TkTablesView& t = a.t;
    const uint64_t s0 = wv_first64(a.doc_offs[d]), s1 = wv_first64(a.doc_offs[d + 1]);
    uint32_t* out = a.staging + s0 + 2 * d;
    uint32_t cursor = 0;
    if (a.add_bos) {
        if (lane == 0) out[0] = t.bos_id;
        cursor = 1;
    }
    uint64_t w0 = s0;
    while (w0 < s1) {
        const uint64_t e = wv_first64(a.pattern ? tk_match_end2(t, a.bytes, w0, s1) : tk_match_end(t, a.bytes, w0, s1));
        const uint32_t r = tk_piece_lookup(a, pw, lane, w0, e);
        if (r != TK_RANK_MAX) {
            if (lane == 0) out[cursor] = r + t.num_special;
            cursor += 1;
        } else if (a.long_list && tk_piece_is_long(a, lane, w0, e) != 0u) {
            // a LONG piece that is not a vocabulary key: thousands of dependent merges for one wave.  The document goes to
            // tk_long.hip, which merges such a piece in rounds by a workgroup (all occurrences of a rank at once)
            if (lane == 0) {
                a.counts[d] = 0;
                a.long_list[wv_atomic_add(a.long_count, 1u)] = (uint32_t)d;
            }
            return;
        } else {
            tk_piece_merge_coop(a, lane, w0, e, out, cursor, scratch);
        }
        w0 = e;
    }
    if (a.add_eos) {
        if (lane == 0) out[cursor] = t.eos_id;
        cursor += 1;
    }
    if (lane == 0) a.counts[d] = cursor;
}

// ------------------------------------------------------------------------------------------
// one wave: pull documents from the work queue until it is empty
// ------------------------------------------------------------------------------------------
// MODE 0: pass 1 over all documents; MODE 1: pass 2 over todo_list; MODE 2: split only;
// MODE 3: pass 1 over todo_list (the documents the flat path, tk_flat_impl.h, handed back)
template <int MODE>
TK_DEV void tk_encode_wave(const TkEncodeArgs& a, int lane, uint64_t wave_id) {
    const TkPolyPow pw = tk_poly_pow(a.t, lane);
    uint32_t* scratch = MODE == 1 ? a.scratch + wave_id * a.scratch_words_per_wave : nullptr;
    const uint64_t total = (MODE == 1 || MODE == 3) ? (uint64_t)(a.n_todo_dev ? wv_first(*a.n_todo_dev) : a.n_todo) : a.n_docs;
    const uint32_t chunk = MODE == 1 ? 1u : TK_DOC_CHUNK;
    for (;;) {
        // Every lane takes part in the fetch (the compiler folds it into ONE atomic of 64*chunk and
        // hands lane i the value old + i*chunk), so no lane-dependent branch feeds the loop condition:
        // with `if (lane == 0) base = atomicAdd(..)` the optimizer threaded lanes 1..63 around the
        // atomic and they re-ran the loop without lane 0 (observed on gfx950, ROCm 7.2).
        const uint32_t ticket = wv_first(wv_atomic_add_all(a.work_counter, chunk));
        const uint32_t base = ticket / 64u;  // scalar: document index, offsets and the window loop are wave-uniform
        if ((uint64_t)base >= total) break;
        const uint64_t hi = (uint64_t)base + chunk < total ? (uint64_t)base + chunk : total;
        for (uint64_t q = base; q < hi; ++q) {
            if (MODE == 1) {
                tk_encode_doc_seq(a, (uint64_t)wv_first(a.todo_list[q]), lane, pw, scratch);
            } else {
                const uint64_t doc = MODE == 3 ? (uint64_t)wv_first(a.todo_list[q]) : q;
                const bool ok = tk_encode_doc<(MODE == 3 ? 0 : MODE)>(a, doc, lane, pw);
                if (!ok && lane == 0) {
                    // pass 1 only: the document has a piece that does not fit a window
                    a.counts[doc] = 0;
                    const uint32_t slot = wv_atomic_add(a.defer_count, 1u);
                    a.defer_list[slot] = (uint32_t)doc;
                }
            }
        }
    }
}

#endif
