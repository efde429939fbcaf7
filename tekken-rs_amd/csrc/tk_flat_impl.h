// tk_flat_impl.h -- the FLAT tokenization path: one wave per CHUNK of the packed byte stream.
//
// Same computation as tk_encode_impl.h (CoreBPE::encode behind reference src/tekkenizer.rs:384-386, pattern
// literal :123) but laid out for the machine instead of for the document:
//
//   * the packed text of ALL documents is cut into regions of 64 x TKF_W bytes (TKF_W = 32: 2048); a wave owns one
//     region at a time, lane l owns bytes [W l, W l + W) and holds every class / rule mask as W bits of a VGPR
//     ("lane layout": one VALU instruction = one mask operation over the whole region; shifts borrow from
//     the neighbour lane with DPP, ripple carries cross lanes through a 64-bit carry look-ahead);
//   * document boundaries are a mask (DS): look-behind shifts are cut at a document start, look-ahead
//     shifts at a document end, runs are broken there -- documents never cost a branch;
//   * piece starts are enumerated into LDS and probed ONE LANE PER PIECE (64 probes per instruction
//     instead of ~13 with one lane per byte); the piece bytes come from an LDS copy of the region, the
//     probe is one scattered load for most pieces (KEY8: 16-byte entries; spill flags, tk_hash.h);
//   * a piece that misses the vocabulary reserves `len` id slots and is queued; tk_merge_wave merges the
//     queue ONE LANE PER PIECE (tiktoken's order) -- 64 independent chains of dependent
//     PAIR probes per wave at high occupancy -- and marks the slots it does not need as holes;
//   * ids are written chunk-dense (chunk c at tmp[c*TKF_STRIDE ..]), K[c] slots per chunk; the slot of
//     every document start inside its chunk (lstart) and the holes per document let tk_flat_assemble
//     cut the stream into documents and squeeze the holes out.
//
// Regions overlap: 32 bytes of left halo (look-behind context), 64 of right halo (look-ahead, piece
// ends), the rest committed (1952 bytes of 2048).  ASCII is classified from the bit planes alone; a region with multi-byte code
// points additionally looks the class of every lead byte up in the trie (the char-level rules use the
// char-start mask).  A document with a digit / CR-LF run that covers the whole left halo, a white-space
// run that reaches the end of the region or a piece of more than TKF_LONGCAP (256) bytes is flagged and redone by the
// per-document kernel (tk_encode_impl.h), which handles everything.  A piece of 65..256 bytes becomes a RECORD
// (step 6: slots reserved, the merge left to tk_flat_long_wave / tk_merge_long_wave) and its document stays here.
//
// The rules are modelled in tools/flat_split_model.py (Python ints as masks, checked against the
// oracle); this file is that model in lane layout.  Runs on the CPU wave emulator (tests/emu).
#ifndef TK_FLAT_IMPL_H
#define TK_FLAT_IMPL_H
#include <stdint.h>

#include "tk_encode_impl.h"
#include "tk_flat_args.h"

#define TKF_WM 0xFFFFFFFFu
#define TKF_LOGW 5
#define TKF_TOPBIT 0x80000000u
#define TKF_NHL (TKF_HL / TKF_W)                     /* lanes of the left halo: 2 / 1 */
/* entries of the LDS piece list.  16 bytes per lane: every byte could start a piece.  32 bytes per lane: a chunk with
   more pieces than this (an average of under two bytes per piece over 2 KB) is handed back as a whole */
#ifndef TKF_LISTCAP
#define TKF_LISTCAP 1056
#endif
#define TKF_MAXPIECES (TKF_LISTCAP - 2)              /* + the sentinel */

// LDS words of one wave
#define TKF_L_LIST 0                                /* u16 [REGION + 4] piece positions, then the id slot of every piece */
#define TKF_L_DS (TKF_L_LIST + TKF_LISTCAP / 2)      /* [64] document-start mask words */
#define TKF_L_PS (TKF_L_DS + 64)                    /* [64] owned piece-start mask words */
#define TKF_L_PFX (TKF_L_PS + 64)                   /* [64] pieces before the lane */
#define TKF_L_BAD (TKF_L_PFX + 64)                  /* [64] positions that make their document fall back */
#define TKF_L_BPFX (TKF_L_BAD + 64)                 /* [64] bad positions before the lane */
/* the dealt walk's list of lead-byte positions (step 1, default pattern): u16 entries in the list words behind the class words */
#define TKF_WALKOFF 192                              /* words */
#define TKF_WALKCAP (2 * (TKF_LISTCAP / 2 - TKF_WALKOFF))   /* 672 positions; a region with more lead bytes is walked lane by lane */
#define TKF_L_CL TKF_L_LIST                          /* [3 * 64] classes L, N, S of the multi-byte code points (step 1 only: shares the
                                                        words of the piece list, which is built in step 5) */
#define TKF_L_KM (TKF_L_BPFX + 64)                  /* [17 * 4] byte masks of a zero-padded key of length 0..16 (filled once per wave) */
#define TKF_L_MEMO TKF_L_KM                          /* [4] row 0 of the key masks (length 0: never read): memo table base lo / hi, mask, the wave's hit count */
#define TKF_L_TXT (TKF_L_KM + 17 * 4)                /* [REGION / 4 + 4] the region's bytes: a piece's 16 bytes are read from here */
#define TKF_L_CONST (TKF_L_TXT + TKF_REGION / 4 + 4)              /* [8] wave-uniform constants of the probe: KEY8 mask, KEY16 mask, KEY8 base, KEY16 base.
                                                        The kernel is out of scalar registers (spilled SGPRs come back through
                                                        v_readlane, a VALU slot each, and VALU issue is what bounds it); an LDS
                                                        broadcast read costs an LDS slot, of which it has plenty */
#define TKF_LDS_WORDS (TKF_L_CONST + 8)
#define TKF_L_FS TKF_LDS_WORDS                       /* [64] CUT instantiation only: cut-only piece starts (fragments begin / end there) */
#define TKF_LDS_WORDS_CUT (TKF_LDS_WORDS + 64)


// ------------------------------------------------------------------------------------------
// lane-layout mask primitives (TKF_W bits per lane, bit i of lane l = region byte W l + i)
// ------------------------------------------------------------------------------------------
// (own word above / below the neighbour's in one register, then ONE bit-field extract: 3 VALU per shift, and the
// combined word is shared by shifts of the same mask)
// (one funnel shift over {own word : neighbour's word}: DPP move + v_alignbit)
TK_DEV uint32_t tkf_shl(uint32_t x, int k) { return (uint32_t)((((uint64_t)x << 32) | (uint64_t)wv_dn1(x)) >> (32 - k)); }  // 1 <= k <= 31
TK_DEV uint32_t tkf_shr(uint32_t x, int k) { return (uint32_t)((((uint64_t)wv_up1(x) << 32) | (uint64_t)x) >> k); }
TK_DEV uint32_t tkf_shl_any(uint32_t x, int k, int lane) {
    const int q = k >> TKF_LOGW, r = k & (TKF_W - 1);
    const int s1 = lane - q, s2 = lane - q - 1;
    uint32_t v1 = wv_shfl(x, s1 & 63), v2 = wv_shfl(x, s2 & 63);
    if (s1 < 0) v1 = 0;
    if (s2 < 0) v2 = 0;
    return (uint32_t)((((uint64_t)v1 << TKF_W) | (uint64_t)v2) >> (TKF_W - r)) & TKF_WM;
}
TK_DEV uint32_t tkf_shr_any(uint32_t x, int k, int lane) {
    const int q = k >> TKF_LOGW, r = k & (TKF_W - 1);
    const int s1 = lane + q, s2 = lane + q + 1;
    uint32_t v1 = wv_shfl(x, s1 & 63), v2 = wv_shfl(x, s2 & 63);
    if (s1 > 63) v1 = 0;
    if (s2 > 63) v2 = 0;
    return (uint32_t)((((uint64_t)v2 << TKF_W) | (uint64_t)v1) >> r) & TKF_WM;
}
TK_DEV bool tkf_any(uint32_t x) { return wv_ballot(x != 0u) != 0ull; }

// a + b over the whole region (64 x W bits): per-lane add, then a 64-bit carry look-ahead over the lane carries
// (G = lanes that generate a carry, P = lanes that would pass one on) hands every lane its carry-in
TK_DEV uint32_t tkf_add(uint32_t a, uint32_t b) {
    const uint32_t t = a + b;
    const uint64_t G = wv_ballot(t < a);                  // the word itself overflowed
    const uint64_t P = wv_ballot((t & TKF_WM) == TKF_WM);
    const uint64_t x = G | P;
    const uint64_t cin = (x + G) ^ x ^ G;  // carry into lane l = bit l
    return (t + (wv_inverse_ballot(cin) ? 1u : 0u)) & TKF_WM;
}
// bits of `run` covered by a carry that starts at `seeds` (subset of run) and ripples upwards through
// consecutive run bits, across lanes
TK_DEV uint32_t tkf_ripple(uint32_t run, uint32_t seeds) { return (tkf_add(run, seeds) ^ run) & run; }
// where a carry started at `seeds` comes to rest: the first position at or above each seed that is not in `run`
TK_DEV uint32_t tkf_land(uint32_t run, uint32_t seeds) { return tkf_add(run, seeds) & ~run & TKF_WM; }

TK_DEV uint32_t tkf_scan_excl(uint32_t v, int lane, uint32_t* total) {
    (void)lane;
    const uint32_t x = wv_scan_incl_u32(v);
    *total = wv_readlane(x, 63);
    return x - v;
}

// ------------------------------------------------------------------------------------------
// bit-plane classification of the lane's TKF_W bytes -> W-bit masks
// ------------------------------------------------------------------------------------------
struct TkfClass {
    uint32_t L, N, S, NL, SP, AP, HI, STMD, RV, E, LL;
    uint32_t U8C, LEAD, C5, BF;  // UTF-8: continuation bytes, lead bytes; C5 BF = U+017F (folds to 's')
    uint32_t UP, SL, X, M;    // upper-case letters, '/', neutral letters (Lm | Lo), marks (the JSON pattern of row f-3 only)
    uint32_t F2, F3;          // lead bytes of 2- / 3-byte chars that the RANGE RULES know to be letters (tkf_classify; 0 in an ASCII region)
    bool nmb;                 // wave-uniform: the region holds a multi-byte \p{N} char
};

// 8x8 bit-matrix transpose of 8 bytes (lo = bytes 0..3, hi = bytes 4..7): afterwards byte b holds bit b of every
// input byte (bit i of the result byte = bit b of input byte i).  Three delta swaps (1x1 blocks inside 2x2,
// 2x2 inside 4x4, 4x4 inside 8x8); the first two never cross the 32-bit halves.
TK_DEV void tkf_transpose8(uint32_t& lo, uint32_t& hi) {
    uint32_t t;
    t = ((lo >> 7) ^ lo) & 0x00AA00AAu; lo ^= t ^ (t << 7);
    t = ((hi >> 7) ^ hi) & 0x00AA00AAu; hi ^= t ^ (t << 7);
    t = ((lo >> 14) ^ lo) & 0x0000CCCCu; lo ^= t ^ (t << 14);
    t = ((hi >> 14) ^ hi) & 0x0000CCCCu; hi ^= t ^ (t << 14);
    t = (((lo >> 28) | (hi << 4)) ^ lo) & 0xF0F0F0F0u;   // delta 28 across the halves
    lo ^= t ^ (t << 28);
    hi ^= t >> 4;
}

// the lane's bytes -> the 8 bit planes (already in lane layout: bit i of plane b = bit b of byte i) -> the class
// masks as boolean functions of the planes.  ASCII classes of the pattern of src/tekkenizer.rs:123: L = [A-Za-z],
// N = [0-9], S = \s (9..13, 0x20); bytes >= 0x80 only set HI.
TK_DEV TkfClass tkf_classify(const uint32_t* x) {
    uint32_t a_lo = x[0], a_hi = x[1], b_lo = x[2], b_hi = x[3];
    tkf_transpose8(a_lo, a_hi);
    tkf_transpose8(b_lo, b_hi);
    // four groups of 8 bytes: plane p = byte p of the four transposed groups -- a 4 x 4 byte transpose (v_perm), twice
    uint32_t c_lo = x[4], c_hi = x[5], d_lo = x[6], d_hi = x[7];
    tkf_transpose8(c_lo, c_hi);
    tkf_transpose8(d_lo, d_hi);
    const uint32_t u0 = wv_perm(b_lo, a_lo, 0x05010400u), u1 = wv_perm(b_lo, a_lo, 0x07030602u);   // {a0 b0 a1 b1}, {a2 b2 a3 b3}
    const uint32_t v0 = wv_perm(d_lo, c_lo, 0x05010400u), v1 = wv_perm(d_lo, c_lo, 0x07030602u);
    const uint32_t p0 = wv_perm(v0, u0, 0x05040100u), p1 = wv_perm(v0, u0, 0x07060302u);
    const uint32_t p2 = wv_perm(v1, u1, 0x05040100u), p3 = wv_perm(v1, u1, 0x07060302u);
    const uint32_t u2 = wv_perm(b_hi, a_hi, 0x05010400u), u3 = wv_perm(b_hi, a_hi, 0x07030602u);
    const uint32_t v2 = wv_perm(d_hi, c_hi, 0x05010400u), v3 = wv_perm(d_hi, c_hi, 0x07030602u);
    const uint32_t p4 = wv_perm(v2, u2, 0x05040100u), p5 = wv_perm(v2, u2, 0x07060302u);
    const uint32_t p6 = wv_perm(v3, u3, 0x05040100u), p7 = wv_perm(v3, u3, 0x07060302u);
    TkfClass c;
    const uint32_t hz = TKF_WM & ~(p7 | p6 | p5 | p4);            // high nibble 0
    const uint32_t pre = p6 & ~p7;                                // 0x40..0x7F: letters differ in bit 5 only
    c.HI = p7;
    c.L = pre & (p4 | p3 | p2 | p1 | p0) & ~(p4 & p3 & (p2 | (p1 & p0)));       // low five bits in 1..26
    c.N = p5 & p4 & ~(p7 | p6) & ~(p3 & (p2 | p1));                             // 0x30..0x39
    c.SP = p5 & ~(p7 | p6 | p4) & ~(p3 | p2 | p1 | p0);                         // 0x20
    c.S = (hz & p3 & ((~p2 & (p1 | p0)) | (p2 & ~p1))) | c.SP;                  // 9..13, 0x20
    c.NL = hz & p3 & ((~p2 & p1 & ~p0) | (p2 & ~p1 & p0));                      // 10, 13
    c.AP = p5 & ~(p7 | p6 | p4) & ~p3 & p2 & p1 & p0;                           // 0x27
    const uint32_t q = pre & ~p3 & p4;                                          // low five bits 10xxx
    c.STMD = (q & ((~p2 & p1 & p0) | (p2 & ~p1 & ~p0))) |                       // s 10011, t 10100
             (pre & ~p4 & p2 & ~p1 & ((p3 & p0) | (~p3 & ~p0)));                // m 01101, d 00100
    c.RV = q & p1 & ~p0;                                                        // r 10010, v 10110
    c.E = pre & ~p4 & ~p3 & p2 & ~p1 & p0;                                      // e 00101
    c.LL = pre & ~p4 & p3 & p2 & ~p1 & ~p0;                                     // l 01100
    c.U8C = p7 & ~p6;
    c.LEAD = p7 & p6;
    c.C5 = p7 & p6 & ~(p5 | p4 | p3) & p2 & ~p1 & p0;                           // 1100 0101
    c.BF = p7 & ~p6 & p5 & p4 & p3 & p2 & p1 & p0;                              // 1011 1111
    c.UP = c.L & ~p5;                                                           // A..Z (letters differ in bit 5 only)
    c.SL = p5 & ~(p7 | p6 | p4) & p3 & p2 & p1 & p0;                             // 0x2F
    c.X = 0u;
    c.M = 0u;
    c.nmb = false;
    c.F2 = 0u;
    c.F3 = 0u;
    if (tkf_any(p7)) {
        // Range rules for the big letter blocks, as mask algebra on the planes (one instruction = 2048 bytes): a char they cover needs
        // no decode and no table look-up -- the per-char walk of tk_flat_chunk (two chars per round and lane: a third of this
        // kernel on mixed UTF-8 text) is left with what the rules do not cover.  Every rule names (lead byte, range of the
        // second byte) whose WHOLE block of code points is class L in the trie (tests/test_flat_path.py checks every code point
        // of the BMP through this path against the oracle, which reads the trie):
        //   3 bytes: E4 B8..BF, E5..E8, E9 80..BD (CJK unified U+4E00..9F7F); EA B0..BF, EB, EC, ED 80..9D (Hangul U+AC00..D77F)
        //   2 bytes: C3 except 97 / B7 (U+00C0..00FF without the two operators); D0, D1 80..8F (Cyrillic U+0400..044F);
        //            CE B1..BF, CF 80..8F (Greek U+03B1..03CF); D8 A0..BF (Arabic U+0620..063F)
        // The bytes behind the lead must be continuation bytes (anything else is for the walk to judge).
#define NX(m) tkf_shr((m), 1)
        const uint32_t cont = p7 & ~p6;
        const uint32_t nc1 = NX(cont), nc2 = tkf_shr(cont, 2);
        const uint32_t x54 = p5 & p4, x321 = p3 & p2 & p1;
        const uint32_t n_geB0 = NX(x54), n_geB8 = NX(x54 & p3), n_BEBF = NX(x54 & x321), n_ge90 = NX(p5 | p4), n_geA0 = NX(p5);
        const uint32_t n_ge9E = NX(p5 | (p4 & x321)), n_B1BF = NX(x54 & (p3 | p2 | p1 | p0)), n_x7 = NX(p4 & ~p3 & p2 & p1 & p0);
        const uint32_t hE = p7 & p6 & p5 & ~p4, hC = p7 & p6 & ~p5 & ~p4, hD = p7 & p6 & ~p5 & p4;
        const uint32_t l0 = ~p3 & ~p2 & ~p1 & ~p0, l1 = ~p3 & ~p2 & ~p1 & p0, l3 = ~p3 & ~p2 & p1 & p0, l4 = ~p3 & p2 & ~p1 & ~p0;
        const uint32_t l5678 = (~p3 & p2 & (p1 | p0)) | (p3 & ~p2 & ~p1 & ~p0);
        const uint32_t l8 = p3 & ~p2 & ~p1 & ~p0, l9 = p3 & ~p2 & ~p1 & p0, lA = p3 & ~p2 & p1 & ~p0, lB = p3 & ~p2 & p1 & p0;
        const uint32_t lC = p3 & p2 & ~p1 & ~p0, lD = p3 & p2 & ~p1 & p0, lE = p3 & p2 & p1 & ~p0, lF = p3 & p2 & p1 & p0;
        c.F3 = hE & nc1 & nc2 & (l5678 | (l4 & n_geB8) | (l9 & ~n_BEBF) | lB | lC | (lA & n_geB0) | (lD & ~n_ge9E));
        c.F2 = nc1 & ((hC & l3 & ~n_x7) | (hD & l0) | (hD & l1 & ~n_ge90) | (hC & lE & n_B1BF) | (hC & lF & ~n_ge90) | (hD & l8 & n_geA0));
#undef NX
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// piece starts of the region (tools/flat_split_model.py flat_rules); also returns SPR / cont for the
// deferral test of the white-space run that reaches the region end
// ------------------------------------------------------------------------------------------
TK_DEV uint32_t tkf_rules(const TkfClass& m, uint32_t DS, int lane, uint32_t* SPR_out, uint32_t* cont_out) {
    const uint32_t nDS = ~DS & TKF_WM;
    const uint32_t DE = tkf_shr(DS, 1) | (lane == 63 ? TKF_TOPBIT : 0u);
    const uint32_t nDE = ~DE & TKF_WM;
#define P1(x) (tkf_shl((x), 1) & nDS)
#define N1(x) (tkf_shr((x), 1) & nDE)
    const uint32_t mL = m.L, mN = m.N, mS = m.S, NL = m.NL, SP = m.SP;
    const uint32_t U8C = m.U8C, CS = TKF_WM & ~U8C;  // first byte of every code point
    const bool u8 = tkf_any(m.HI);                   // wave-uniform: the region holds multi-byte code points
    const uint32_t mO = TKF_WM & ~(mL | mN | mS);
    const uint32_t pOS = P1(mO | SP);            // previous byte is class O or U+0020
    uint32_t CEND = 0;
    if (tkf_any(m.AP)) {                         // alt 1: fires only where a match starts at the apostrophe
        const uint32_t ok = m.AP & ~pOS;
        const uint32_t c2 = ok & N1(m.STMD);
        uint32_t three = (m.RV & N1(m.E)) | (m.LL & N1(m.LL));
        if (u8) three |= m.C5 & N1(m.BF);        // U+017F folds to 's' ((?i) of the pattern)
        const uint32_t c3 = ok & ~c2 & N1(three);
        CEND = tkf_shl(c2, 2) | tkf_shl(c3, 3);
    }
    const uint32_t L1 = P1(mL), O1 = P1(mO);
    const uint32_t Lst = mL & ~L1;
    // "the O char before me is not available as alt 2's prefix": it is itself preceded by O or U+0020.
    // X = such O chars, on all their bytes
    uint32_t X = CS & mO & pOS;
    if (u8) {
        X |= P1(X) & U8C;
        X |= P1(X) & U8C;
        X |= P1(X) & U8C;
    }
    const uint32_t psL = (mL & L1 & CEND) | (Lst & P1(mN | NL)) | (Lst & P1(X));
    const uint32_t psO = mO & ~O1 & ~P1(SP);
    // numbers: every 3rd char of a run (\p{N}{1,3})
    const uint32_t N1m = P1(mN);
    uint32_t psN = mN & ~N1m;
    if (!m.nmb) {
        // one byte per char: prefix doubling on byte positions
        const uint32_t N2m = P1(N1m);
        uint32_t M = mN & N1m & N2m & P1(N2m);
        int k = 3;
        while (tkf_any(M)) {
            psN |= tkf_shl_any(psN, k, lane) & M;
            M &= tkf_shl_any(M, k, lane);
            k *= 2;
        }
    } else {
        // multi-byte digits: walk the runs three CHARS at a time (next char start = where a carry through the
        // continuation bytes comes to rest)
        uint32_t cur = psN;
        while (tkf_any(cur)) {
            for (int q = 0; q < 3; ++q) cur = tkf_land(U8C, tkf_shl(cur, 1)) & mN & nDS;
            cur &= ~psN;
            psN |= cur;
        }
    }
    // white space
    const uint32_t seeds = NL & O1;
    uint32_t ABS = 0;
    if (tkf_any(seeds)) ABS = tkf_ripple(NL & nDS, seeds);
    const uint32_t SPR = mS & ~ABS;
    const uint32_t cont = SPR & P1(SPR);
    uint32_t Z = NL & SPR;
    {
        uint32_t C = tkf_shr(cont, 1);
        int k = 1;
        while (tkf_any(C) && tkf_any(Z)) {
            Z |= tkf_shr_any(Z, k, lane) & C;
            C &= tkf_shr_any(C, k, lane);
            k *= 2;
        }
    }
    uint32_t last = SPR & ~tkf_shr(cont, 1) & ~Z & nDE;   // last byte of a run that a non-space char follows
    if (u8) {                                              // -> the first byte of its char
        last = (last & CS) | N1(last & U8C);
        last = (last & CS) | N1(last & U8C);
        last = (last & CS) | N1(last & U8C);
        last &= CS;
    }
    const uint32_t psS = (SPR & ~cont) | (tkf_shl(Z, 1) & cont & ~Z) | last;
#undef P1
#undef N1
    *SPR_out = SPR;
    *cont_out = cont;
    return psL | psN | psO | psS | DS;
}

// ------------------------------------------------------------------------------------------
// Row f-3: the JSON pattern of Mistral's tekken.json in lane layout (tools/flat_split_model.py flat_rules_tekken).  Scope:
// chars of the classes upper (Lu | Lt), lower (Ll), N, \s, other; a region with a neutral letter (Lm / Lo) or a mark
// is handed back.  Against tkf_rules: no contraction alternative; inside a letter run a piece starts at an upper-case
// char that follows a lower-case one; every digit is a piece; the tail absorbed after a punctuation run is made of
// CR / LF / '/' and whatever follows it starts a piece.
// ------------------------------------------------------------------------------------------
TK_DEV uint32_t tkf_rules_json(const TkfClass& m, uint32_t DS, int lane, uint32_t* SPR_out, uint32_t* cont_out) {
    const uint32_t nDS = ~DS & TKF_WM;
    const uint32_t DE = tkf_shr(DS, 1) | (lane == 63 ? TKF_TOPBIT : 0u);
    const uint32_t nDE = ~DE & TKF_WM;
#define P1(x) (tkf_shl((x), 1) & nDS)
#define N1(x) (tkf_shr((x), 1) & nDE)
    const uint32_t mL = m.L, mN = m.N, mS = m.S, NL = m.NL, SP = m.SP, mX = m.X, mM = m.M;
    const uint32_t U8C = m.U8C, CS = TKF_WM & ~U8C;
    const bool u8 = tkf_any(m.HI);
    const bool marks = u8 && tkf_any(mM), neutral = u8 && tkf_any(mX | mM);
    const uint32_t mO0 = TKF_WM & ~(mL | mX | mM | mN | mS);          // punctuation / symbols / everything else
    // Two sets that feed each other, both decided by the char to the LEFT: T = the tail [\r\n/]* of a punctuation piece (a
    // CR / LF behind any punctuation char, behind a mark that the 4th alternative swallowed, or behind a tail char; a '/'
    // behind a tail char) and A = where the 4th alternative is running (a punctuation char -- not one of a tail -- behind
    // U+0020, behind a punctuation char that is not in a tail, or behind A; a mark behind A: that alternative's class
    // holds \p{M}).  Position i needs position i - 1 only: iterating both definitions from nothing settles chains of
    // length k after k rounds (the chains are runs of punctuation / marks / line ends: short).
    uint32_t T = 0u, A = 0u;
    {
        const uint32_t O01 = P1(mO0);
        if (tkf_any(NL & O01) || marks) {
            const uint32_t SP1 = P1(SP);
            for (;;) {
                const uint32_t T1 = P1(T);
                const uint32_t T2 = (NL & (O01 | T1 | P1(mM & A))) | (m.SL & T1);
                uint32_t A2 = 0u;
                if (marks) {
                    A2 = CS & ((mO0 & ~T2 & (SP1 | P1(mO0 & ~T2) | P1(A))) | (mM & P1(A)));   // decided at the char's first byte ...
                    A2 |= P1(A2) & U8C;                                                      // ... and valid for all its bytes
                    A2 |= P1(A2) & U8C;
                    A2 |= P1(A2) & U8C;
                }
                const bool changed = tkf_any((T2 ^ T) | (A2 ^ A));
                T = T2;
                A = A2;
                if (!changed) break;
            }
        }
    }
    const uint32_t ABS = T;
    const uint32_t Mabs = mM & A;                                      // marks that count as punctuation
    const uint32_t mO = mO0 | Mabs;
    const uint32_t neut = mX | (mM & ~Mabs);                           // upper-side AND lower-side word chars
    const uint32_t mWd = mL | neut;                                    // word chars
    const uint32_t after_abs = P1(ABS) & ~ABS & CS;                    // whatever follows the tail starts a piece
    const uint32_t Oe = mO & ~ABS;
    const uint32_t pOS = P1(Oe | SP);
    const uint32_t Wst = mWd & ~P1(mWd);                               // first byte of a word run
    uint32_t X = CS & Oe & pOS;                                        // an O char that is not available as a word's one-char prefix
    if (u8) {
        X |= P1(X) & U8C;
        X |= P1(X) & U8C;
        X |= P1(X) & U8C;
    }
    // inside a word run: an upper-case char starts a piece when the lower side has begun (a lower-case char, then any
    // neutral chars), and the all-upper tail of a run starts one behind a neutral char (the first alternative gives the
    // upper-side run back down to its last neutral char when nothing lower-side follows)
    uint32_t LW = mL & ~m.UP;
    uint32_t psC;
    if (neutral) {
        for (;;) {
            const uint32_t nxt = LW | (P1(LW) & neut);
            const bool grew = tkf_any(nxt & ~LW);
            LW = nxt;
            if (!grew) break;
        }
        uint32_t UE = m.UP & ~N1(mWd);                                 // upper-case chars with only upper-case chars up to the run end
        for (;;) {
            const uint32_t nxt = UE | (N1(UE) & m.UP);
            const bool grew = tkf_any(nxt & ~UE);
            UE = nxt;
            if (!grew) break;
        }
        psC = CS & ((m.UP & P1(LW)) | (UE & P1(neut)));
    } else {
        psC = CS & m.UP & P1(LW);
    }
    const uint32_t psL = (Wst & P1(mN | NL | X)) | psC;
    const uint32_t psO = Oe & ~P1(Oe) & ~P1(SP);
    const uint32_t psN = mN & CS;
    const uint32_t SPR = mS & ~ABS;
    const uint32_t cont = SPR & P1(SPR);
    uint32_t Z = NL & SPR;
    {
        uint32_t C = tkf_shr(cont, 1);
        int k = 1;
        while (tkf_any(C) && tkf_any(Z)) {
            Z |= tkf_shr_any(Z, k, lane) & C;
            C &= tkf_shr_any(C, k, lane);
            k *= 2;
        }
    }
    uint32_t last = SPR & ~tkf_shr(cont, 1) & ~Z & nDE;
    if (u8) {
        last = (last & CS) | N1(last & U8C);
        last = (last & CS) | N1(last & U8C);
        last = (last & CS) | N1(last & U8C);
        last &= CS;
    }
    const uint32_t psS = (SPR & ~cont) | (tkf_shl(Z, 1) & cont & ~Z) | last;
#undef P1
#undef N1
    *SPR_out = SPR;
    *cont_out = cont;
    return psL | psN | psO | psS | after_abs | DS;
}

// ------------------------------------------------------------------------------------------
// one chunk
// ------------------------------------------------------------------------------------------
// once per wave, before its first chunk: the key byte masks
TK_DEV void tk_flat_init_lds(const TkFlatArgs& a, uint32_t* lds, int lane) {
    if (lane == 63) {
        const uint64_t b8 = (uint64_t)reinterpret_cast<uintptr_t>(a.t.key8_tab), b16 = (uint64_t)reinterpret_cast<uintptr_t>(a.t.key_tab);
        lds[TKF_L_CONST + 0] = a.t.key8_mask;
        lds[TKF_L_CONST + 1] = a.t.key_mask;
        lds[TKF_L_CONST + 2] = (uint32_t)b8; lds[TKF_L_CONST + 3] = (uint32_t)(b8 >> 32);
        lds[TKF_L_CONST + 4] = (uint32_t)b16; lds[TKF_L_CONST + 5] = (uint32_t)(b16 >> 32);
        const uint64_t lc = (uint64_t)reinterpret_cast<uintptr_t>(a.long_ctl);
        lds[TKF_L_CONST + 6] = (uint32_t)lc; lds[TKF_L_CONST + 7] = (uint32_t)(lc >> 32);
    }
    if (lane >= 1 && lane <= 16) {
        for (int q = 0; q < 4; ++q) {
            const int keep = lane - 4 * q;
            lds[TKF_L_KM + 4 * lane + q] = keep >= 4 ? 0xFFFFFFFFu : keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u);
        }
    }
    if (lane == 0) {
        // row 0 of the key masks is never read (no piece has length 0): it holds the memo's constants -- table base, mask -- and the
        // wave's count of memo hits (tk_flat_flush_memo_hits); the kernel has no LDS granule and no scalar register to spare
        const uint64_t mb = a.memo_probe ? (uint64_t)reinterpret_cast<uintptr_t>(a.memo_tab) : 0ull;
        lds[TKF_L_MEMO + 0] = (uint32_t)mb; lds[TKF_L_MEMO + 1] = (uint32_t)(mb >> 32);
        lds[TKF_L_MEMO + 2] = a.memo_mask;
        lds[TKF_L_MEMO + 3] = 0u;
    }
    wv_lds_sync();
}

// once per wave, behind its last chunk: the wave's memo hits into the call's device counter
TK_DEV void tk_flat_flush_memo_hits(const TkFlatArgs& a, uint32_t* lds, int lane) {
    if (a.memo_tab && a.memo_hits) {
        wv_lds_sync();
        const uint32_t v = lds[TKF_L_MEMO + 3];
        if (lane == 0 && v != 0u) wv_atomic_add(a.memo_hits, v);
    }
}

TK_DEV uint32_t tkf_lowmask32(int n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

// DBG = 0: the production instantiation.  DBG = 1 adds the per-byte piece-start flags of tk_split_batch (and, in a `make ablate`
// build, the timing ablations of TK_DEBUG_ABLATE): every one of those tests costs scalar registers and VALU slots the kernel does not have.
// MODE = the key hash the tables were built with (TkTablesView::key_hash_mode), a compile-time constant here.
// PAT = 0: the reference's hard-coded pattern; 1: the JSON pattern of tekken.json (row f-3, opt-in).
// CUT = 0: the production instantiation; a chunk that holds a piece of more than 64 bytes (two neighbouring lanes without a
// piece start) is appended to a.cut_list and left untouched -- the function returns true.  CUT = 1 (tk_flat_cut_kernel, over
// that list): the same chunk work, and pieces of more than 64 bytes are cut into FRAGMENTS wherever the vocabulary rules out a
// part that spans the boundary (step 4b); fragments merge on their own, without the whole-piece look-up.
// timing-only ablations of the flat kernel (TK_DEBUG_ABLATE): compiled in by `make ablate` (-DTK_ABLATE) only -- the shipped
// library's DBG instantiation carries nothing but the per-byte piece-start flags of tk_split_batch
#ifdef TK_ABLATE
#define TKF_ABL(a, bit) (DBG && ((a).dbg_ablate & (bit)))
#else
#define TKF_ABL(a, bit) false
#endif

// MEMO = 1: the look-up in the memo of merged pieces compiled in (step 6).  The production kernel exists in both forms: a call that
// does not use the table (switched off, or paused by the policy: text with few unknown pieces) runs the MEMO = 0 instantiation, which
// is the kernel without a single instruction of it (13 M of 326 M VALU wave-instructions per C2 launch, two VGPRs and a spill).
template <int DBG, int MODE, int PAT = 0, int CUT = 0, int MEMO = 1>
TK_DEV bool tk_flat_chunk(const TkFlatArgs& a, uint64_t c, int lane, uint32_t* lds) {
    const TkTablesView& t = a.t;
    const int64_t n = (int64_t)a.n_bytes;
    const int64_t c0 = (int64_t)c * TKF_COMMIT, r0 = c0 - TKF_HL, r1 = r0 + TKF_REGION;
    const int64_t c1 = c0 + TKF_COMMIT < n ? c0 + TKF_COMMIT : n;
    const int ca = TKF_HL, cb = (int)(c1 - r0);  // commit range in region coordinates
    uint16_t* list = reinterpret_cast<uint16_t*>(lds + TKF_L_LIST);

    // ---- 1. load TKF_W bytes per lane, classify ---------------------------------------------------
    uint32_t x[TKF_W / 4];
    for (int k = 0; k < TKF_W / 4; ++k) x[k] = 0u;
    {
        const uint8_t* bytes = WV_KARGS(TkFlatArgs, a).bytes;
        const int64_t g = r0 + TKF_W * lane;
        if (g >= 0 && g + TKF_W <= n) {
            for (int k = 0; k < TKF_W / 16; ++k) wv_load16(bytes + g + 16 * k, x + 4 * k);
        } else if (g + TKF_W > 0 && g < n) {
            for (int k = 0; k < TKF_W; ++k) {
                const int64_t q = g + k;
                if (q >= 0 && q < n) x[k >> 2] |= (uint32_t)bytes[q] << (8 * (k & 3));
            }
        }
    }
    {
        uint32_t* txt = lds + TKF_L_TXT + (TKF_W / 4) * lane;   // bytes outside [0, n) are zero
        for (int k = 0; k < TKF_W / 4; ++k) txt[k] = x[k];
        if (lane == 0) for (int k = 0; k < 4; ++k) lds[TKF_L_TXT + TKF_REGION / 4 + k] = 0u;
    }
    uint32_t nextb = 0;                                     // CUT: the byte behind the region (the last lane's trigram reaches it)
    if (CUT && r1 < n) nextb = (uint32_t)a.bytes[r1];
    TkfClass m = tkf_classify(x);
    uint32_t walk = m.LEAD;                                 // lead bytes the per-char walk has to look up
    if (!PAT && tkf_any(m.F2 | m.F3)) {
        // the chars the range rules cover: letters, all their bytes (a char may reach into the next lane)
        m.L |= m.F2 | m.F3 | tkf_shl(m.F2 | m.F3, 1) | tkf_shl(m.F3, 2);
        walk &= ~(m.F2 | m.F3);
    }
    if (tkf_any(walk)) {
        // multi-byte code points: every lane walks the lead bytes among its own bytes, decodes the code point, looks its
        // class up in the trie and marks ALL bytes of the char (runs stay contiguous; a char may reach into the next lane)
        uint32_t* cl = lds + TKF_L_CL;
        cl[lane] = 0u; cl[64 + lane] = 0u; cl[128 + lane] = 0u;
        if (PAT) { cl[192 + lane] = 0u; cl[256 + lane] = 0u; cl[320 + lane] = 0u; }   // upper case, Lm | Lo, marks
        if (lane == 0) {
            // the code point of a lead byte in the region's last three bytes ends beyond it: the pad word takes the real bytes
            uint32_t pad = 0u;
            for (int k = 0; k < 4; ++k)
                if (r1 + k < n) pad |= (uint32_t)a.bytes[r1 + k] << (8 * k);
            lds[TKF_L_TXT + TKF_REGION / 4] = pad;
        }
        wv_lds_sync();
        bool nmb = false;
        uint32_t w = walk;
        // the char at byte i of this lane: its code point (0xFFFFFFFF: malformed) and byte length, from the LDS copy of the region
        // (bytes outside [0, n) are zero there)
        auto decode = [&](int i, uint32_t* clen_out) -> uint32_t {
            const uint32_t p = (uint32_t)TKF_W * (uint32_t)lane + (uint32_t)i;
            const uint32_t* tw = lds + TKF_L_TXT + (p >> 2);
            const uint32_t v4 = wv_alignbyte(tw[1], tw[0], p & 3u);
            const uint32_t b0 = v4 & 0xFFu, b1 = (v4 >> 8) & 0xFFu, b2 = (v4 >> 16) & 0xFFu, b3 = v4 >> 24;
            uint32_t cp = 0xFFFFFFFFu, clen = 1;
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
            *clen_out = clen;
            return cp;
        };
        auto mark = [&](int i, uint32_t clen, uint32_t cls, uint32_t extra) {
            if (PAT && extra) {
                const uint64_t xb = (uint64_t)((1u << clen) - 1u) << i;
                wv_lds_or(cl + (2u + extra) * 64u + (uint32_t)lane, (uint32_t)(xb & TKF_WM));
                if ((xb >> TKF_W) && lane < 63) wv_lds_or(cl + (2u + extra) * 64u + (uint32_t)lane + 1u, (uint32_t)(xb >> TKF_W));
            }
            if (cls != TK_CLS_O) {
                const uint64_t bits = (uint64_t)((1u << clen) - 1u) << i;   // may reach into the next lane's word
                wv_lds_or(cl + (cls - 1u) * 64u + (uint32_t)lane, (uint32_t)(bits & TKF_WM));
                if ((bits >> TKF_W) && lane < 63) wv_lds_or(cl + (cls - 1u) * 64u + (uint32_t)lane + 1u, (uint32_t)(bits >> TKF_W));
                if (cls == TK_CLS_N) nmb = true;
            }
        };
        // Default pattern: the lead bytes are dealt out over ALL lanes -- their positions go into an LDS list (the words of the
        // piece list behind the class words; step 5 builds its list later), one lane per char, 64 chars per round.  The chars
        // the range rules leave over sit in a few lanes (a Thai word, a run of symbols): with every lane walking its own
        // bytes the wave ran as many rounds -- each one a table load and its latency -- as its fullest lane had chars.
        bool dealt = false;
        if (!PAT) {
            uint32_t tot;
            const uint32_t before = tkf_scan_excl((uint32_t)__builtin_popcount(w), lane, &tot);
            if (tot <= TKF_WALKCAP) {
                dealt = true;
                uint16_t* wl = list + 2 * TKF_WALKOFF;
                uint32_t ww = w, idx = before;
                while (wv_ballot(ww != 0u)) {
                    if (ww) {
                        wl[idx++] = (uint16_t)(TKF_W * lane + __builtin_ctz(ww));
                        ww &= ww - 1u;
                    }
                }
                wv_lds_sync();
                for (uint32_t b0 = 0; b0 < tot; b0 += 64u) {
                    const uint32_t i = b0 + (uint32_t)lane;
                    if (i < tot) {
                        const uint32_t p = wl[i];
                        const uint32_t* tw = lds + TKF_L_TXT + (p >> 2);
                        const uint32_t v4 = wv_alignbyte(tw[1], tw[0], p & 3u);
                        const uint32_t b0b = v4 & 0xFFu, b1 = (v4 >> 8) & 0xFFu, b2 = (v4 >> 16) & 0xFFu, b3 = v4 >> 24;
                        uint32_t cp = 0xFFFFFFFFu, clen = 1;
                        if (b0b < 0xE0u) {
                            if ((b1 & 0xC0u) == 0x80u) { cp = ((b0b & 0x1Fu) << 6) | (b1 & 0x3Fu); clen = 2; }
                        } else if (b0b < 0xF0u) {
                            if ((b1 & 0xC0u) == 0x80u && (b2 & 0xC0u) == 0x80u) {
                                cp = ((b0b & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu); clen = 3;
                            }
                        } else if (b0b < 0xF8u) {
                            if ((b1 & 0xC0u) == 0x80u && (b2 & 0xC0u) == 0x80u && (b3 & 0xC0u) == 0x80u) {
                                cp = ((b0b & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu); clen = 4;
                            }
                        }
                        const uint32_t q = cp < 0x10000u ? cp : 0u;
                        uint32_t cw = t.uc_bmp[q >> 4];
                        uint32_t cls = (cw >> (2u * (q & 15u))) & 3u;
                        if (cp >= 0x10000u) cls = cp == 0xFFFFFFFFu ? (uint32_t)TK_CLS_O : tk_uc_class(t, cp);
                        if (cls != TK_CLS_O) {
                            const uint32_t wi = p >> TKF_LOGW, sh = p & (TKF_W - 1);
                            const uint64_t bits = (uint64_t)((1u << clen) - 1u) << sh;   // may reach into the next lane's word
                            wv_lds_or(cl + (cls - 1u) * 64u + wi, (uint32_t)(bits & TKF_WM));
                            if ((bits >> TKF_W) && wi < 63u) wv_lds_or(cl + (cls - 1u) * 64u + wi + 1u, (uint32_t)(bits >> TKF_W));
                            if (cls == TK_CLS_N) nmb = true;
                        }
                    }
                }
            }
        }
        while (!dealt && wv_ballot(w != 0u)) {
            if (w) {
                if (PAT) {
                    const int i = __builtin_ctz(w);
                    w &= w - 1u;
                    uint32_t clen;
                    const uint32_t cp = decode(i, &clen);
                    uint32_t cls = TK_CLS_O, extra = 0;   // JSON pattern: 1 upper case, 2 Lm | Lo, 3 mark -> a second mask word
                    if (cp != 0xFFFFFFFFu) {
                        // classes of unicode_tables2.h: 1 upper (Lu | Lt), 2 lower, 3 Lm | Lo, 4 mark, 5 N, 6 \s
                        const uint32_t c2 = tk_uc_class2(t, cp);
                        cls = c2 == 1u || c2 == 2u ? TK_CLS_L : c2 == 5u ? TK_CLS_N : c2 == 6u ? TK_CLS_S : TK_CLS_O;
                        extra = c2 == 1u ? 1u : c2 == 3u ? 2u : c2 == 4u ? 3u : 0u;
                    }
                    mark(i, clen, cls, extra);
                } else {
                    // two chars per round, their class words requested together: ONE load each from the trie flattened for the
                    // BMP (two dependent loads through the two-stage trie, one char at a time, were half of this kernel's time
                    // on mixed UTF-8 text); a char beyond the BMP walks the trie
                    const int i0 = __builtin_ctz(w);
                    w &= w - 1u;
                    const bool two = w != 0u;
                    const int i1 = two ? __builtin_ctz(w) : i0;
                    w &= w - 1u;                          // (w == 0 stays 0)
                    uint32_t l0, l1;
                    const uint32_t cp0 = decode(i0, &l0), cp1 = decode(i1, &l1);
                    const uint32_t q0 = cp0 < 0x10000u ? cp0 : 0u, q1 = cp1 < 0x10000u ? cp1 : 0u;
                    uint32_t w0 = t.uc_bmp[q0 >> 4], w1 = t.uc_bmp[q1 >> 4];
                    WV_PIN(w0); WV_PIN(w1);
                    uint32_t c0 = (w0 >> (2u * (q0 & 15u))) & 3u, c1 = (w1 >> (2u * (q1 & 15u))) & 3u;
                    if (cp0 >= 0x10000u) c0 = cp0 == 0xFFFFFFFFu ? (uint32_t)TK_CLS_O : tk_uc_class(t, cp0);
                    if (cp1 >= 0x10000u) c1 = cp1 == 0xFFFFFFFFu ? (uint32_t)TK_CLS_O : tk_uc_class(t, cp1);
                    mark(i0, l0, c0, 0u);
                    if (two) mark(i1, l1, c1, 0u);
                }
            }
        }
        wv_lds_sync();
        m.L |= cl[lane];
        m.N |= cl[64 + lane];
        m.S |= cl[128 + lane];
        m.nmb = wv_ballot(nmb) != 0ull;
        if (PAT) {
            m.UP |= cl[192 + lane];
            m.X = cl[256 + lane];
            m.M = cl[320 + lane];
        }
    }
    if (TKF_ABL(a, 16)) {
        // (keeps the work alive without producing slot counts: the downstream kernels must see an empty chunk)
        const uint32_t v = m.L + m.N + m.S + m.NL + m.SP + m.AP + m.HI + m.STMD + m.RV + m.E + m.LL;
        if (lane == 0) {
            a.kcount[c] = v == 0xFFFFFFFFu ? 1u : 0u;
            for (int k = 0; k < 4; ++k) a.miss_count[k * a.n_chunks + c] = 0u;
        }
        return false;
    }

    // ---- 2. document starts inside the region -> DS (lane layout, through LDS) ------------------
    lds[TKF_L_DS + lane] = 0u;
    lds[TKF_L_BAD + lane] = 0u;
    wv_lds_sync();
    const uint64_t fd = WV_KARGS(TkFlatArgs, a).first_doc[c];          // documents that start below max(r0, 0)
    bool starts_at_r1 = false;
    {
        // documents d >= fd start at or above the region start; d == n_docs stands for the end of the stream
        const auto& ka = WV_KARGS(TkFlatArgs, a);
        for (uint64_t base = fd;; base += 64) {
            const uint64_t d = base + (uint64_t)lane;
            int64_t s = INT64_MAX;
            if (d <= ka.n_docs) s = (int64_t)ka.doc_offs[d];
            const bool in = s < r1;
            if (in) wv_lds_or(lds + TKF_L_DS + ((s - r0) >> TKF_LOGW), 1u << ((s - r0) & (TKF_W - 1)));
            if (s == r1) starts_at_r1 = true;
            if (wv_ballot(in) != ~0ull) break;
        }
        starts_at_r1 = wv_ballot(starts_at_r1) != 0ull;
    }
    wv_lds_sync();
    const uint32_t DS = lds[TKF_L_DS + lane];

    // ---- 3. piece starts ---------------------------------------------------------------------------
    uint32_t SPR, cont;
    uint32_t PS = PAT ? tkf_rules_json(m, DS, lane, &SPR, &cont) : tkf_rules(m, DS, lane, &SPR, &cont);
    if (TKF_ABL(a, 8)) {
        if (lane == 0) {
            a.kcount[c] = (PS + SPR + cont) == 0xFFFFFFFFu ? 1u : 0u;
            for (int k = 0; k < 4; ++k) a.miss_count[k * a.n_chunks + c] = 0u;
        }
        return false;
    }
    if (!CUT && !PAT) {
        // A piece of more than 64 bytes (surely one of 96 and more, and any piece that begins or goes on beyond this region)
        // leaves two neighbouring lanes of the commit range / right halo without a piece start: the chunk is left to the CUT
        // instantiation.  Both chunks of a piece that crosses a commit boundary decide alike: the one it starts in sees no
        // start in its right halo (lanes 62, 63), the next one none in its first 64 commit bytes (lanes 2, 3).
        const uint64_t Zm = wv_ballot(PS == 0u) & ~(((uint64_t)1 << TKF_NHL) - 1ull);
        if (Zm & (Zm >> 1)) {
            const uint64_t lc = (uint64_t)wv_first(lds[TKF_L_CONST + 6]) | ((uint64_t)wv_first(lds[TKF_L_CONST + 7]) << 32);
            if (lc != 0ull) {
                const uint32_t* ctl = reinterpret_cast<const uint32_t*>(wv_global_ptr(lc));
                const uint64_t lp = (uint64_t)wv_first(ctl[3]) | ((uint64_t)wv_first(ctl[4]) << 32);
                if (lp != 0ull) {
                    if (lane == 0) {
                        const uint32_t q = wv_atomic_add(const_cast<uint32_t*>(ctl) - 4, 1u);
                        reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(wv_global_ptr(lp)))[q] = (uint32_t)c;
                    }
                    return true;
                }
            }
        }
    }
    // bits of the commit range [ca, cb): whole lanes 2..59 for every chunk but the last one of the stream
    uint32_t commit_mask = (uint32_t)(lane - TKF_NHL) < (uint32_t)(TKF_COMMIT / TKF_W) ? TKF_WM : 0u;
    if (cb != TKF_HL + TKF_COMMIT)
        commit_mask = (tkf_lowmask32(cb - TKF_W * lane < 0 ? 0 : cb - TKF_W * lane) & ~tkf_lowmask32(ca - TKF_W * lane < 0 ? 0 : ca - TKF_W * lane)) & TKF_WM;
    if (DBG && a.dbg_starts) {
        for (int k = 0; k < TKF_W; ++k)
            if ((commit_mask >> k) & 1u) a.dbg_starts[r0 + TKF_W * lane + k] = (PS >> k) & 1u;
    }

    // ---- 4. positions that make their document fall back ---------------------------------------
    uint32_t BAD = 0;
    {
        // (A) a digit / CR-LF run that comes from below the region and covers the whole left halo
        // (the region may begin inside a code point: its leading continuation bytes have no class and count as part of the run)
        const uint32_t d0 = wv_readlane(DS, 0);
        const uint32_t n0 = wv_readlane(m.N | (m.U8C & ~(m.U8C + 1u)), 0);
        // (the CR / LF run counts the orphan continuation bytes too: whether the char they belong to was punctuation -- the run is
        // then the tail its piece absorbs -- or white space is not known here.  Found by the GPU fuzz in round 4: U+3000, forty CRs,
        // eleven TABs, a region that began inside the U+3000 -> the CRs looked absorbed and a piece began at the first TAB)
        const uint32_t l0 = wv_readlane((PAT ? (m.NL | m.SL) : m.NL) | (m.U8C & ~(m.U8C + 1u)), 0);   // (JSON pattern: the absorbed tail runs through CR / LF / '/')
        const bool covered = n0 == TKF_WM || l0 == TKF_WM;
        if (r0 > 0 && d0 == 0u && covered && lane == TKF_NHL) BAD |= 1u;
        if (PAT) {
            // JSON pattern: two more runs whose state comes from below the region -- a word run (upper or lower side?) and a
            // punctuation / mark run (is the 4th alternative running?) must not cover the whole left halo either
            const uint32_t lead_cont = m.U8C & ~(m.U8C + 1u);
            const uint32_t wd = m.L | m.X | m.M, om = (TKF_WM & ~(m.L | m.X | m.N | m.S));
            // (the tail and the punctuation / mark state feed each other -- a mark behind a tail char is a word char, behind running
            // punctuation it is punctuation, a '/' behind a tail char is tail --, so a CHAIN of tail chars, punctuation and marks that
            // comes from below and covers the halo is as undecidable as a run of one kind.  Found by the model campaign of round 4:
            // "-----" + forty CRs + eleven '/' + thirty-one U+0301 + "'''''", the region beginning inside the CRs)
            bool cov2 = wv_readlane(wd | lead_cont, 0) == TKF_WM || wv_readlane(om | m.NL | m.SL | lead_cont, 0) == TKF_WM;
            if (r0 > 0 && d0 == 0u && cov2 && lane == TKF_NHL) BAD |= 1u;
        }
        // (B) a white-space run that reaches the region end, goes on in the same document and started inside
        //     the commit range
        const uint32_t top = wv_readlane(SPR, 63);
        if (r1 < n && (top & TKF_TOPBIT) && !starts_at_r1) {
            const uint32_t nz = ~cont & TKF_WM;
            const uint64_t NZ = wv_ballot(nz != 0u);
            int f = 0;
            if (NZ) {
                const int tl = tk_msb64(NZ);
                const uint32_t w = wv_readlane(nz, tl);
                f = TKF_W * tl + (31 - __builtin_clz(w));
            }
            if (f < cb) {
                const int at = f > ca ? f : ca;
                if (lane == (at >> TKF_LOGW)) BAD |= 1u << (at & (TKF_W - 1));
            }
        }
    }

    // ---- 4b. CUT instantiation: pieces of more than 64 bytes are cut into fragments -------------------------------
    // (tools/cut_model.py.)  The merge loop joins two parts only if their bytes are a vocabulary key, so a part that spans
    // the boundary between bytes i-1 and i is a key that contains b[i-1] b[i] there: the bigram itself, or a longer key
    // that holds the trigram b[i-2..i] or b[i-1..i+1].  Where the vocabulary has neither (cut_k2: two-byte keys; cut_g3:
    // trigrams inside tokens) no part ever spans the boundary and both sides merge on their own, in their own order:
    // position i becomes a piece start.  A piece with a cut is no key (its n-grams would occur in a token: itself), so its
    // FRAGMENTS skip the whole-piece look-up -- and must: what merging a fragment gives need not be the key it may happen
    // to be.  Which pieces: those of more than 64 bytes that start in the commit range (S64: 64 bytes without a start behind
    // them; that includes a piece whose end this region does not see) and the piece that comes in from the last chunk when
    // it covers the first 64 commit bytes -- exactly when that chunk saw no end of it.  A cut is a pure function of four
    // bytes, so the chunks of a piece agree on every fragment; a chunk owns the fragments that start in its commit range
    // and, when the piece ends in its right halo, those up to that end (the next chunk sees fewer than 64 foreign bytes
    // and leaves them alone).
    uint32_t own_mask = commit_mask;
    uint32_t FS = 0;                                        // cut-only piece starts
    if (CUT) {
        const uint32_t R1 = lane >= TKF_NHL ? (~PS & TKF_WM) : 0u;
        uint32_t F = R1 & tkf_shr(R1, 1);                   // F(i): no start at i .. i + 2^k - 1
        F &= tkf_shr(F, 2);
        F &= tkf_shr(F, 4);
        F &= tkf_shr(F, 8);
        F &= tkf_shr(F, 16);
        F &= wv_up1(F);                                     // ... i + 63 (the bit 32 positions on is the next lane's)
        const uint32_t S64 = PS & commit_mask & tkf_shr(F, 1);
        uint32_t LM = 0;                                    // bytes of the pieces that are cut (behind their first byte)
        if (tkf_any(S64)) LM = tkf_ripple(R1, tkf_shl(S64, 1) & R1);
        // the piece that comes in from the chunk before: up to the first start at or behind the commit start
        uint32_t f_lo = TKF_REGION;
        {
            const uint64_t Bm = wv_ballot(lane >= TKF_NHL && PS != 0u);
            if (Bm) {
                const int fl = tk_ctz64(Bm);
                f_lo = (uint32_t)TKF_W * (uint32_t)fl + (uint32_t)__builtin_ctz(wv_readlane(PS, fl));
            }
        }
        uint32_t LMc = 0;
        if (f_lo >= (uint32_t)ca + 64u) {
            const int lo = ca - TKF_W * lane, hi = (int)f_lo - TKF_W * lane;
            LMc = tkf_lowmask32(hi < 0 ? 0 : hi) & ~tkf_lowmask32(lo < 0 ? 0 : lo);
            LM |= LMc;
        }
        const uint32_t ge_cb = ~tkf_lowmask32(cb - TKF_W * lane < 0 ? 0 : cb - TKF_W * lane);
        const bool closed = tkf_any(PS & ge_cb);            // the piece that crosses the commit end ends inside the region
        // where a cut is wanted: T(j) = the trigram around byte j occurs inside a token, K(j) = bytes j-1, j are a key
        const uint32_t need = LM | tkf_shr(LM, 1);
        uint32_t TG = 0, K2 = 0;
        {
            const uint32_t prevb = wv_dn1(x[TKF_W / 4 - 1]) >> 24;
            uint32_t nb = wv_up1(x[0]) & 0xFFu;
            if (lane == 63) nb = nextb;
            if (need) {
                uint32_t b0 = prevb, b1 = x[0] & 0xFFu;
                for (int k = 0; k < TKF_W; ++k) {
                    const uint32_t b2 = k + 1 < TKF_W ? (x[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xFFu : nb;
                    if ((need >> k) & 1u) {
                        const uint32_t i3 = b0 | (b1 << 8) | (b2 << 16), i2 = b0 | (b1 << 8);
                        TG |= ((a.t.cut_g3[i3 >> 5] >> (i3 & 31u)) & 1u) << k;
                        K2 |= ((a.t.cut_k2[i2 >> 5] >> (i2 & 31u)) & 1u) << k;
                    }
                    b0 = b1; b1 = b2;
                }
            }
        }
        FS = LM & ~PS & ~K2 & ~TG & ~tkf_shl(TG, 1);
        // The piece that comes in is taken over only at a cut among the first 64 commit bytes -- the stretch the chunk before
        // sees as well (its right halo).  Without one that chunk saw no end of its last fragment: either it knows of no cut
        // at all and has left the WHOLE piece to a long-piece record (then every byte of it here is foreign, later cuts
        // included), or it has flagged the document.
        if (tkf_any(LMc)) {
            const uint32_t win = ((uint32_t)lane == (uint32_t)TKF_NHL || (uint32_t)lane == (uint32_t)TKF_NHL + 1u) ? TKF_WM : 0u;
            if (!tkf_any(FS & LMc & win)) { FS &= ~LMc; LM &= ~LMc; }
        }
        PS |= FS;
        if (closed) own_mask |= LM & ge_cb;
        lds[TKF_L_FS + lane] = FS;
    }

    // ---- 5. enumerate the pieces: positions of the set bits of PS from the commit start on -------
    // One pass over all lanes when the pieces fit the LDS list (always with 16 bytes per lane; with 32 bytes per lane
    // unless the region averages under two bytes per piece); otherwise two passes, lanes below 32 and lanes from 32 on.
    uint32_t PSown = PS & own_mask;
    uint32_t PSlist = lane >= TKF_NHL ? PS : 0u;            // commit range and right halo (ends of the last pieces)
    uint32_t np_all, np_own, pfx_all, pfx_own;
    {
        // one scan for both counts (each < 2^16)
        uint32_t tot;
        const uint32_t pf = tkf_scan_excl((uint32_t)__builtin_popcount(PSlist) | ((uint32_t)__builtin_popcount(PSown) << 16), lane, &tot);
        pfx_all = pf & 0xFFFFu; pfx_own = pf >> 16;
        np_all = tot & 0xFFFFu; np_own = tot >> 16;
    }
    const int npass = np_all > TKF_MAXPIECES ? 2 : 1;
    uint32_t* tmp = WV_KARGS(TkFlatArgs, a).tmp + c * TKF_STRIDE;
    uint32_t* mq = WV_KARGS(TkFlatArgs, a).miss_list + c * TKF_MISSCAP;
    uint32_t E = 0;                                         // slots beyond one per piece so far
    uint32_t base_own = 0;                                  // pieces of the earlier pass
    uint32_t nm0 = 0, nm1 = 0, nm2 = 0, nm3 = 0;
    uint32_t lane_lo = TKF_NHL, lane_hi = 64;               // lanes of the pass
    // slot of every document that starts inside the lanes of the pass (and, with want_flags, the hand-back flags)
    auto doc_outputs = [&](bool want_flags, bool anybad) {
        // documents that touch the commit range: fd - 1 (the one that contains the region start) onwards
        const uint64_t dfirst = fd > 0 ? fd - 1 : 0;
        const auto& ka = WV_KARGS(TkFlatArgs, a);           // (n_docs, doc_offs, lstart, flags: read here, not kept in scalar registers)
        for (uint64_t base = dfirst;; base += 64) {
            const uint64_t d = base + (uint64_t)lane;
            int64_t s = INT64_MAX, e = INT64_MAX;
            if (d < ka.n_docs) {
                s = (int64_t)ka.doc_offs[d];
                e = (int64_t)ka.doc_offs[d + 1];
            }
            const bool in = s < c1;
            if (in && s >= c0) {
                // id slots of this chunk before the document's first byte (a document start is a piece start)
                const uint32_t p = (uint32_t)(s - r0);
                const uint32_t pl = p >> TKF_LOGW;
                if (npass == 1 || (pl >= lane_lo && pl < lane_hi)) {
                    const uint32_t pi = lds[TKF_L_PFX + pl] + (uint32_t)__builtin_popcount(lds[TKF_L_PS + pl] & ((1u << (p & (TKF_W - 1))) - 1u));
                    ka.lstart[d] = pi < np_own ? list[pi] : base_own + np_own + E;
                }
            }
            if (want_flags && anybad && in && e > c0) {
                // any bad position inside [s, e) clipped to the region?
                const int64_t lo = s > r0 ? s - r0 : 0, hi = e < r1 ? e - r0 : TKF_REGION;
                if (hi > lo) {
                    const uint32_t pl = (uint32_t)lo, ph = (uint32_t)hi;
                    const uint32_t cl = lds[TKF_L_BPFX + (pl >> TKF_LOGW)] + (uint32_t)__builtin_popcount(lds[TKF_L_BAD + (pl >> TKF_LOGW)] & ((1u << (pl & (TKF_W - 1))) - 1u));
                    uint32_t ch;
                    if (ph >= TKF_REGION) ch = lds[TKF_L_BPFX + 63] + (uint32_t)__builtin_popcount(lds[TKF_L_BAD + 63]);
                    else ch = lds[TKF_L_BPFX + (ph >> TKF_LOGW)] + (uint32_t)__builtin_popcount(lds[TKF_L_BAD + (ph >> TKF_LOGW)] & ((1u << (ph & (TKF_W - 1))) - 1u));
                    if (ch > cl) ka.flags[d] = 1u;
                }
            }
            if (wv_ballot(in) != ~0ull) break;
        }
    };
  for (int pass = 0; pass < npass; ++pass) {
    uint32_t sentinel = TKF_REGION;                         // end of the pass's last piece: the region end ...
    if (npass == 2) {
        lane_lo = pass ? 32 : TKF_NHL;
        lane_hi = pass ? 64 : 32;
        const bool mine = (uint32_t)lane >= lane_lo && (uint32_t)lane < lane_hi;
        PSlist = mine ? PS : 0u;
        PSown = PSlist & own_mask;
        uint32_t tot;
        const uint32_t pf = tkf_scan_excl((uint32_t)__builtin_popcount(PSlist) | ((uint32_t)__builtin_popcount(PSown) << 16), lane, &tot);
        pfx_all = pf & 0xFFFFu; pfx_own = pf >> 16;
        np_all = tot & 0xFFFFu; np_own = tot >> 16;
        if (pass == 0) {                                    // ... or the first piece start of the second pass
            const uint64_t Bm = wv_ballot(lane >= 32 && PS != 0u);
            if (Bm) {
                const int fl = (int)__builtin_ctzll(Bm);
                sentinel = (uint32_t)TKF_W * (uint32_t)fl + (uint32_t)__builtin_ctz(wv_readlane(PS, fl));
            }
        }
    }
    {
        uint32_t w = PSlist, idx = pfx_all;
        while (wv_ballot(w != 0u)) {
            if (w) {
                list[idx++] = (uint16_t)(TKF_W * lane + __builtin_ctz(w));
                w &= w - 1u;
            }
        }
        if (lane == 0) list[np_all] = (uint16_t)sentinel;
    }
    lds[TKF_L_PS + lane] = PSown;
    lds[TKF_L_PFX + lane] = pfx_own;
    wv_lds_sync();

    // ---- 6. whole-piece lookup, one lane per piece; ids stored at once ---------------------------------
    // A piece that misses the vocabulary reserves `len` id slots (it cannot produce more ids than bytes) and is
    // queued for tk_merge_wave; the slots it does not fill stay TKF_HOLE and are squeezed out by the assembly.
    const uint32_t nbatch = (np_own + 63u) / 64u;
    for (uint32_t j = 0; j < nbatch; ++j) {
        // Straight-line for the common case (a piece of up to 16 bytes): every lane fetches bytes and hashes -- a lane
        // without a piece takes position 0, length 1 -- and only the table loads are predicated.
        const uint32_t idx = j * 64u + (uint32_t)lane;
        const bool act = idx < np_own;
        uint32_t pos, len;
        if (j + 1u < nbatch) {
            // every batch but the last one: all 64 lanes own a piece and idx + 1 is inside the list -- no clamping, no selects
            // (a scalar branch; six instructions less per batch)
            const uint32_t p0 = list[idx], p1 = list[idx + 1u];
            pos = p0;
            len = p1 - p0;
        } else {
            const uint32_t p0 = list[idx < np_all ? idx : np_all], p1 = list[idx < np_all ? idx + 1 : np_all];   // (never past the sentinel)
            pos = act ? p0 : 0u;
            len = act ? p1 - p0 : 1u;
        }
        // the piece's first 16 bytes from the LDS copy of the region (five aligned dwords, funnel-shifted), zeroed past the
        // piece (masks by length from LDS)
        uint32_t kk[4];
        {
            const uint32_t* tw = lds + TKF_L_TXT + (pos >> 2);
            const uint32_t t0 = tw[0], t1 = tw[1], t2 = tw[2], t3 = tw[3], t4 = tw[4];
            const uint32_t sh = pos & 3u;
            const uint32_t* km = lds + TKF_L_KM + 4u * (len < 16u ? len : 16u);
            kk[0] = wv_alignbyte(t1, t0, sh) & km[0]; kk[1] = wv_alignbyte(t2, t1, sh) & km[1];
            kk[2] = wv_alignbyte(t3, t2, sh) & km[2]; kk[3] = wv_alignbyte(t4, t3, sh) & km[3];
        }
        uint32_t r = kk[0];                                 // rank of a single byte is the byte (src/tekkenizer.rs:793-798)
        uint32_t h = 0;
        if (TKF_ABL(a, 1)) {
            r = 7u;
        } else if (TKF_ABL(a, 64)) {
            r = (kk[0] ^ kk[1] ^ kk[2] ^ kk[3]) & 0xFFFFu;                                            // timing: no hash, no table
        } else {
            h = tk_key_hash((uint32_t)MODE, kk[0], kk[1], kk[2], kk[3], len);
            if (TKF_ABL(a, 128)) {
                r = h & 0xFFFFu;                                                                      // timing: no table
            } else if (len - 2u <= 14u) {                   // 2..16 bytes: exact-key probe
                const uint32_t* kc = lds + TKF_L_CONST;
                const uint64_t b8 = (uint64_t)kc[2] | ((uint64_t)kc[3] << 32), b16 = (uint64_t)kc[4] | ((uint64_t)kc[5] << 32);
                // (timing only, 512: the probes stay inside the first 1 / 16 of either table -- what the look-up would cost if the tables fitted the L2)
                r = tk_probe_key_h(wv_global_ptr(b8), TKF_ABL(a, 512) ? kc[0] >> 4 : kc[0], wv_global_ptr(b16), TKF_ABL(a, 512) ? kc[1] >> 4 : kc[1],
                                   h, kk[0], kk[1], kk[2], kk[3], len);
            }
        }
        // CUT: a fragment (it begins or ends at a cut) takes no whole-piece look-up: a single byte is its own rank, anything
        // longer goes to the merge queues as it is
        bool frag = false;
        if (CUT) {
            const uint32_t pe = pos + len;
            frag = act && ((((lds[TKF_L_FS + (pos >> TKF_LOGW)] >> (pos & (TKF_W - 1))) & 1u) != 0u) ||
                           (pe < (uint32_t)TKF_REGION && ((lds[TKF_L_FS + (pe >> TKF_LOGW)] >> (pe & (TKF_W - 1))) & 1u) != 0u));
            if (frag && len >= 2u) r = TK_RANK_MAX;
        }
        bool toolong = false;
        if (wv_ballot(len > 16u)) {                         // rare: polynomial hash over the bytes, LONG table; > 64: see below
            uint32_t lres = 0;                              // a long piece that stays on the flat path: id slots reserved for it
            bool lopen = false;
            if (len > 64u) {
                // More than 64 bytes.  Up to TKF_LONGCAP the piece is left to tk_flat_long_kernel (whole-piece lookup, merge)
                // and the document stays here: `len` slots are reserved like for any missed piece.  The chunk's LAST piece
                // may end beyond the region (its length here is only "up to the region end"): it reserves LONGCAP slots and
                // the long kernel finds the end with the sequential matcher -- beyond LONGCAP it flags the document itself.
                lopen = idx + 1u == np_all && sentinel == TKF_REGION && r0 + (int64_t)TKF_REGION < (int64_t)a.n_bytes;
                const bool have_ctl = (lds[TKF_L_CONST + 6] | lds[TKF_L_CONST + 7]) != 0u;
                if (!PAT && have_ctl && (lopen || len <= TKF_LONGCAP) && !(CUT && frag && lopen)) lres = lopen ? TKF_LONGCAP : len;
                else toolong = true;                        // (a fragment whose end the region does not show: no cut in 64 bytes)
            } else if (len > 16u && !TKF_ABL(a, 1) && !(CUT && frag)) {
                // 17..64 bytes: KEY64, the piece's dwords from the LDS copy of the region (the polynomial byte hash of LONG,
                // two multiplies and a global load per byte, was 44 % of this kernel on mixed UTF-8 text, whose words are long).
                // First the pre-filter: the hash of (first 16 bytes, length) is at hand, and a clear bit says "no token"
                // without the fold over the piece's dwords and without a probe -- the answer for most long pieces of running text
                r = TK_RANK_MAX;
                const uint32_t pb = tk_k64_prebit(h);
                const uint32_t* pre = reinterpret_cast<const uint32_t*>(t.key64_tab + t.key64_mask + 1u);
                const bool maybe = TKF_ABL(a, 64) || TKF_ABL(a, 128) || ((pre[pb >> 5] >> (pb & 31u)) & 1u) != 0u;
                if (maybe) r = tk_probe_key64(t, lds + TKF_L_TXT + (pos >> 2), pos & 3u, len, kk[0], kk[1], kk[2], kk[3]);
            }
            if (toolong) wv_lds_or(lds + TKF_L_BAD + (pos >> TKF_LOGW), 1u << (pos & (TKF_W - 1)));
            if (wv_ballot(lres != 0u)) {
                // A batch with a long piece is finished HERE, in the general forms of the bookkeeping below, so that the
                // common path carries nothing of it (two more spilled scalars there were 1.5 % of the kernel on C2).
                const bool longp = lres != 0u;
                const bool lmiss = r == TK_RANK_MAX && !toolong && !longp && !TKF_ABL(a, 2);
                uint32_t lslot = base_own + idx + E;
                {
                    uint32_t tot;
                    lslot += tkf_scan_excl(lmiss ? len - 1u : longp ? lres - 1u : 0u, lane, &tot);
                    E += tot;
                }
                {
                    // (records, capacity and counter through the control words: see TkFlatArgs::long_ctl; ONE atomic for the
                    // batch's long pieces -- a counter every long piece of the batch bumps on its own serialises in the L2)
                    const uint64_t LB = wv_ballot(longp);
                    const int lfirst = tk_ctz64(LB);
                    const uint32_t* ctl = reinterpret_cast<const uint32_t*>(wv_global_ptr((uint64_t)lds[TKF_L_CONST + 6] | ((uint64_t)lds[TKF_L_CONST + 7] << 32)));
                    uint32_t qbase = 0;
                    if (lane == lfirst) qbase = wv_atomic_add(const_cast<uint32_t*>(ctl) - 5, (uint32_t)tk_popc64(LB));
                    qbase = wv_shfl(qbase, lfirst);
                    const uint32_t q = qbase + (uint32_t)tk_popc64(LB & tk_lowmask(lane));
                    TkFlatLongRec* recs = reinterpret_cast<TkFlatLongRec*>(const_cast<uint8_t*>(wv_global_ptr((uint64_t)ctl[0] | ((uint64_t)ctl[1] << 32))));
                    if (!longp) {
                    } else if (q < ctl[2]) {
                        TkFlatLongRec lr;
                        lr.pos = (uint64_t)(r0 + (int64_t)pos); lr.chunk = (uint32_t)c; lr.slot = lslot; lr.len = lopen ? 0u : (CUT && frag ? len | TKF_LREC_FRAG : len); lr.reserved = lres;
                        recs[q] = lr;
                    } else {                                // no room for the record: the document is handed back after all
                        wv_lds_or(lds + TKF_L_BAD + (pos >> TKF_LOGW), 1u << (pos & (TKF_W - 1)));
                    }
                }
                if (wv_ballot(lmiss)) {                     // the misses of the batch: queued by class
                    const uint32_t cls = (len > 8u ? 1u : 0u) + (len > 16u ? 1u : 0u) + (len > 32u ? 1u : 0u);
                    const uint32_t sh = cls * 8u;
                    uint32_t ctot;
                    const uint32_t before = (tkf_scan_excl(lmiss ? 1u << sh : 0u, lane, &ctot) >> sh) & 0xFFu;
                    if (lmiss) {
                        const uint32_t qb = cls == 0u ? TKF_MISSOFF0 + nm0 : cls == 1u ? TKF_MISSOFF1 + nm1 : cls == 2u ? TKF_MISSOFF2 + nm2 : TKF_MISSOFF3 + nm3;
                        mq[qb + before] = TKF_REC(pos, len, lslot);
                    }
                    nm0 += ctot & 0xFFu;
                    nm1 += (ctot >> 8) & 0xFFu;
                    nm2 += (ctot >> 16) & 0xFFu;
                    nm3 += ctot >> 24;
                }
                wv_lds_sync();                              // positions read before they are overwritten
                if (act) {
                    list[idx] = (uint16_t)lslot;
                    if (!lmiss && !longp && !TKF_ABL(a, 4)) tmp[lslot] = r + t.num_special;
                }
                continue;
            }
        }
        // (a lane without a piece holds a byte value in r: never TK_RANK_MAX)
        const bool miss = r == TK_RANK_MAX && !toolong && !TKF_ABL(a, 2);
        uint32_t slot = base_own + idx + E;
        const uint64_t MB = wv_ballot(miss);
        uint32_t res = len;                                 // id slots of a missed piece
        bool mhit = false;
        uint32_t mv0 = 0, mv1 = 0, mv2 = 0, mv4 = 0;        // the entry's rank words; mv4 = its fifth rank (kept across the slot scan: loading
                                                            // the entry again where the ids are stored was a second dependent load, +0.08 ms)
        if (MB) {
            // MEMO (tk_hash.h): a piece of 2..16 bytes that is no vocabulary key may have been merged in an earlier call -- its exact
            // key and the ids the merge gave it sit in a 32-byte entry (two loads, one round trip).  A hit reserves exactly its ids'
            // slots and stores them right here: no queue entry, no holes, nothing for the merge kernel to do.  The entry is a pure
            // function of the bytes (of a fragment's too: fragments take the pure merge), so a hit can never change an id.
            const uint64_t mbase = MEMO ? (uint64_t)wv_first(lds[TKF_L_MEMO + 0]) | ((uint64_t)wv_first(lds[TKF_L_MEMO + 1]) << 32) : 0ull;
            if (MEMO && mbase != 0ull) {
                if (miss && len <= 16u) {
                    const uint32_t ms = tk_memo_slot(h) & lds[TKF_L_MEMO + 2];
                    const tk_u32x4* me = reinterpret_cast<const tk_u32x4*>(wv_global_ptr(mbase) + ((uint64_t)ms << 5));
                    tk_u32x4 ek = me[0], ev = me[1];
                    WV_PIN(ek.x); WV_PIN(ev.x);                 // both loads before the compare
                    if ((((ek.x ^ kk[0]) | (ek.y ^ kk[1]) | (ek.z ^ kk[2]) | (ek.w ^ kk[3])) == 0u) && tk_memo_len(ev.w) == len) {
                        mhit = true;
                        res = tk_memo_n(ev.w);
                        mv0 = ev.y; mv1 = ev.z; mv2 = ev.w; mv4 = ev.x;
                    }
                }
                const uint64_t HB = wv_ballot(mhit);
                if (HB && lane == 0) lds[TKF_L_MEMO + 3] += (uint32_t)tk_popc64(HB);
            }
            // A miss reserves `len` id slots (it cannot produce more ids than bytes): one per piece + (len - 1) more for
            // every miss before it.  The misses are queued in the chunk's own region (no global atomics), records in
            // piece order, one sub-queue per length class (2..8, 9..16, 17..32, 33..64 bytes): the merge kernels run one
            // class per wave.  The two common cases are cheap: ONE miss in the batch (its extra slots shift the lanes
            // behind it: one readlane) and misses of the first class only (rank inside the class = misses before the
            // lane: mbcnt); otherwise a DPP scan of the extra slots and ONE scan over four packed 8-bit counters.
            if ((MB & (MB - 1ull)) == 0ull) {
                const int src = tk_ctz64(MB);
                const uint32_t extra = wv_readlane(res, src) - 1u;
                if (lane > src) slot += extra;
                E += extra;
            } else {
                uint32_t tot;
                slot += tkf_scan_excl(miss ? res - 1u : 0u, lane, &tot);
                E += tot;
            }
            const bool qmiss = MEMO ? miss && !mhit : miss; // what is left for the merge kernels
            const uint64_t QB = wv_ballot(qmiss);
            if (QB == 0ull) {
            } else if (wv_ballot(qmiss && len > 8u) == 0ull) {
                if (qmiss) mq[TKF_MISSOFF0 + nm0 + (uint32_t)tk_popc64(QB & tk_lowmask(lane))] = TKF_REC(pos, len, slot);
                nm0 += (uint32_t)tk_popc64(QB);
            } else {
                const uint32_t cls = (len > 8u ? 1u : 0u) + (len > 16u ? 1u : 0u) + (len > 32u ? 1u : 0u);
                const uint32_t sh = cls * 8u;
                uint32_t ctot;
                const uint32_t before = (tkf_scan_excl(qmiss ? 1u << sh : 0u, lane, &ctot) >> sh) & 0xFFu;
                if (qmiss) {
                    const uint32_t qb = cls == 0u ? TKF_MISSOFF0 + nm0 : cls == 1u ? TKF_MISSOFF1 + nm1 : cls == 2u ? TKF_MISSOFF2 + nm2 : TKF_MISSOFF3 + nm3;
                    mq[qb + before] = TKF_REC(pos, len, slot);
                }
                nm0 += ctot & 0xFFu;
                nm1 += (ctot >> 8) & 0xFFu;
                nm2 += (ctot >> 16) & 0xFFu;
                nm3 += ctot >> 24;
            }
        }
        wv_lds_sync();                                      // positions read before they are overwritten
        if (act) {
            list[idx] = (uint16_t)slot;                     // step 7 looks the slot of a document start up here
            if (!miss && !TKF_ABL(a, 4)) tmp[slot] = r + t.num_special;
        }
        if (MEMO && mhit) {                                 // the ids of the memo entry
            tmp[slot] = tk_memo_id(mv4, mv0, mv1, mv2, 0u) + t.num_special;
            if (res > 1u) tmp[slot + 1u] = tk_memo_id(mv4, mv0, mv1, mv2, 1u) + t.num_special;
            if (res > 2u) tmp[slot + 2u] = tk_memo_id(mv4, mv0, mv1, mv2, 2u) + t.num_special;
            if (res > 3u) tmp[slot + 3u] = tk_memo_id(mv4, mv0, mv1, mv2, 3u) + t.num_special;
            if (res > 4u) tmp[slot + 4u] = tk_memo_id(mv4, mv0, mv1, mv2, 4u) + t.num_special;
        }
    }
    if (pass + 1 < npass) {
        wv_lds_sync();                                      // the slots written by the last batch
        doc_outputs(false, false);                          // the slots of this pass's documents, while its list is there
        base_own += np_own;
        wv_lds_sync();                                      // the list is rebuilt by the next pass
    }
  }
    if (lane == 0) {
        const auto& ka = WV_KARGS(TkFlatArgs, a);
        ka.kcount[c] = base_own + np_own + E;
        ka.miss_count[c] = nm0;                      // class-major: class k of chunk c at [k * n_chunks + c]
        ka.miss_count[ka.n_chunks + c] = nm1;
        ka.miss_count[2 * ka.n_chunks + c] = nm2;
        ka.miss_count[3 * ka.n_chunks + c] = nm3;
    }

    // ---- 7. per-document outputs: slot of every document start, fall-back flags ---------------------
    BAD |= lds[TKF_L_BAD + lane];                           // piece-level marks of step 6
    const bool anybad = tkf_any(BAD);
    if (anybad) {
        uint32_t nb;
        const uint32_t bp = tkf_scan_excl((uint32_t)__builtin_popcount(BAD), lane, &nb);
        lds[TKF_L_BAD + lane] = BAD;
        lds[TKF_L_BPFX + lane] = bp;
    }
    wv_lds_sync();
    doc_outputs(true, anybad);
    wv_lds_sync();  // the next chunk reuses the LDS slice
    return false;
}

// ------------------------------------------------------------------------------------------
// One long-piece record (step 6 above), by one wave: what the per-document kernels would do with this piece, but
// only with this piece -- the rest of its document stays on the flat path.
// ------------------------------------------------------------------------------------------
TK_DEV void tk_flat_long_wave(const TkFlatArgs& a, const TkPolyPow& pw, uint32_t q, int lane, uint32_t* scratch) {
    const TkFlatLongRec lr = a.long_recs[q];
    const uint64_t g = lr.pos;
    // the piece's document: largest d with doc_offs[d] <= g
    uint64_t d = a.first_doc[lr.chunk];
    d = d > 0 ? d - 1 : 0;
    while (d + 1 < a.n_docs && a.doc_offs[d + 1] <= g) ++d;
    d = wv_first64(d);
    if (wv_first(a.flags[d]) != 0u) return;                 // handed back anyway: its slots are never read
    TkEncodeArgs ea;
    ea.bytes = a.bytes; ea.doc_offs = a.doc_offs; ea.n_docs = a.n_docs; ea.staging = nullptr; ea.counts = nullptr;
    ea.work_counter = nullptr; ea.defer_count = nullptr; ea.defer_list = nullptr; ea.todo_list = nullptr; ea.n_todo = 0; ea.n_todo_dev = nullptr;
    ea.scratch = scratch; ea.scratch_words_per_wave = 0; ea.add_bos = 0; ea.add_eos = 0; ea.split_only = 0; ea.pattern = 0; ea.dbg_ablate = 0;
    ea.dbg_starts = nullptr; ea.dbg_mark = nullptr; ea.long_list = nullptr; ea.long_count = nullptr; ea.long_min = 0; ea.long_lazy_mul = 0; ea.long_force = 0;
    ea.long_jobs = nullptr; ea.long_job_count = nullptr; ea.long_job_cap = 0; ea.t = a.t;
    const uint64_t s1 = wv_first64(a.doc_offs[d + 1]);
    const bool frag = (lr.len & TKF_LREC_FRAG) != 0u;       // a fragment of a cut piece: merged as it is, no whole-piece look-up
    const uint32_t rlen = lr.len & ~TKF_LREC_FRAG;
    const uint64_t e = rlen ? g + rlen : wv_first64(tk_match_end(a.t, a.bytes, g, s1));
    const uint32_t len = (uint32_t)(e - g);
    uint32_t* out = a.tmp + (uint64_t)lr.chunk * TKF_STRIDE + lr.slot;
    if (e - g > (uint64_t)lr.reserved) {                    // an open piece of more than LONGCAP bytes: the per-document kernels take the document
        if (lane == 0) {
            // (flagged AFTER the list of the handed-back documents was made -- it is made right behind the flat kernel, beside
            // the merge kernels --: the first record to flag a document puts it on the late list)
            if (a.late_list) {
                if (wv_atomic_exch(a.flags + d, 1u) == 0u) a.late_list[wv_atomic_add(a.late_count, 1u)] = (uint32_t)d;
            } else {
                a.flags[d] = 1u;
            }
        }
        return;
    }
    uint32_t cur = 0;
    const uint32_t r = frag ? TK_RANK_MAX : tk_piece_lookup(ea, pw, lane, g, e);
    if (r != TK_RANK_MAX) {
        if (lane == 0) out[0] = r + a.t.num_special;
        cur = 1;
    } else if (a.long_merge128) {
        // not a vocabulary key: the merge is left to the kernels that do nothing else -- up to 128 bytes one lane per piece
        // (tk_flat_long128_kernel: 64 chains per wave instead of one), beyond that the single-wave merge in a kernel of
        // its own (tk_flat_long_coop_kernel: this one's matcher and lookup cost it half its waves).  The record keeps
        // its length and a mark (no list, no counter: those kernels walk all records).
        if (lane == 0) {
            a.long_recs[q].len = len;
            a.long_recs[q].reserved = lr.reserved | (len <= 128u ? 0x80000000u : 0x40000000u);
        }
        return;
    } else {
        tk_piece_merge_coop(ea, lane, g, e, out, cur, scratch);
    }
    for (uint32_t k = cur + (uint32_t)lane; k < lr.reserved; k += 64u) out[k] = TKF_HOLE;
    if (lane == 0) wv_atomic_add(a.holes + d, lr.reserved - cur);
}

// a record of 129..TKF_LONGCAP bytes that tk_flat_long_wave marked: one wave merges it with the parts in registers
TK_DEV void tk_flat_long_coop_wave(const TkFlatArgs& a, uint32_t q, int lane, uint32_t* scratch) {
    TkFlatLongRec lr = a.long_recs[q];
    if (wv_first(lr.reserved & 0x40000000u) == 0u) return;
    lr.reserved &= 0x3FFFFFFFu;
    const uint64_t g = lr.pos;
    uint64_t d = a.first_doc[lr.chunk];
    d = d > 0 ? d - 1 : 0;
    while (d + 1 < a.n_docs && a.doc_offs[d + 1] <= g) ++d;
    d = wv_first64(d);
    uint32_t* out = a.tmp + (uint64_t)lr.chunk * TKF_STRIDE + lr.slot;
    uint32_t cur = 0;
    (void)scratch;
    tk_piece_merge_small(a.t, a.bytes, lane, g, lr.len, out, cur);   // (<= TKF_LONGCAP = 256 bytes: the parts in registers)
    for (uint32_t k = cur + (uint32_t)lane; k < lr.reserved; k += 64u) out[k] = TKF_HOLE;
    if (lane == 0) wv_atomic_add(a.holes + d, lr.reserved - cur);
}

// ------------------------------------------------------------------------------------------
// tk_merge_wave: byte-pair merge of 64 queued pieces, ONE LANE PER PIECE (tiktoken's _byte_pair_merge,
// repeatedly merge the leftmost minimum-rank adjacent pair).  Every lane runs its own
// chain of dependent PAIR probes, 64 chains per wave and several waves per SIMD hide their latency.
// Pieces of up to 32 bytes keep their parts in LDS columns of the lane's own (tk_merge_lds); the rare longer
// ones (<= 64 bytes) are merged one at a time, one lane per byte, with the segmented-min rounds.
// The ids go to the `len` slots the flat kernel reserved; unused slots become TKF_HOLE and the
// document's hole count is raised so that tk_flat_counts / assemble can squeeze them out.
// ------------------------------------------------------------------------------------------
#define TKM_SHORT 16

// the calling wave's own stretch of the memo log (tk_merge_lds): no counter is shared between waves -- one counter for all of them
// was up to 2 ms of the merge kernel (hundreds of thousands of atomics, and as many look-sees, on one address)
struct TkMemoLog {
    tk_memo_entry* base;   // this wave's records
    uint32_t n, cap;       // records written so far (wave-uniform), room
};
template <bool WIDE>
TK_DEV void tk_merge_items(const TkFlatArgs& a, bool have, uint32_t rec, uint32_t chunk, int lane, uint32_t* mlds, const uint32_t* filt, TkMemoLog* ml = nullptr);

// 64 queued pieces per wave.  The queue counts are laid out class-major ([k * n_chunks + c]); their exclusive prefix
// sums (total at [4 n_chunks]) order all queued pieces by class first, chunk second: item i lives in the sub-queue e
// with prefix[e] <= i < prefix[e + 1].  WIDE = false takes the items of the classes 2..8 / 9..16 bytes, WIDE = true
// those of 17..32 / 33..64 bytes (own kernel: its 32-entry LDS columns would cost the common case occupancy).
// mlds: TKM_LDS_WORDS(32 if WIDE, else 16) words of LDS of the wave's own; filt: the block's LDS copy of the PAIR
// filter (TK_PAIRF_WORDS words, tk_hash.h)
template <bool WIDE>
TK_DEV bool tk_merge_find(const TkFlatArgs& a, uint64_t wave_id, int lane, bool& have_out, uint32_t& rec_out, uint32_t& chunk_out) {
    const uint64_t* prefix = a.miss_prefix;
    const uint64_t n_e = 4 * a.n_chunks;
    // (WIDE takes the class 17..32 bytes; the class 33..64 bytes is merged one piece at a time, one lane per byte: eight
    // pieces per wave, tk_merge_wave_long3 below)
    const uint64_t first = WIDE ? prefix[2 * a.n_chunks] : 0, total = WIDE ? prefix[3 * a.n_chunks] : prefix[2 * a.n_chunks];
    const uint64_t item0 = first + wave_id * 64;
    have_out = false; rec_out = 0u; chunk_out = 0u;
    if (item0 >= total) return false;                 // wave-uniform
    const uint64_t item = item0 + (uint64_t)lane;
    const bool have = item < total;
    uint64_t cl = 0, pcl = 0;                         // the lane's sub-queue and its first item
    {
        // the wave's first sub-queue was written down by tk_merge_wavefirst_kernel (one table for the narrow classes, one
        // for the wide ones); the sub-queues of the 64 items follow from ONE coalesced load of the next 64 prefix sums and
        // a binary search across the lanes (bpermute)
        uint64_t base = (WIDE ? a.wave_first_wide : a.wave_first)[wave_id];        // prefix[base] <= item0
        uint64_t pbase = prefix[base];
        bool done = !have;
        int windows = 0;
        for (;;) {
            const uint64_t idx = base + 1 + (uint64_t)lane;
            const uint64_t pf = idx <= n_e ? prefix[idx] : ~0ull;
            // d = first item of sub-queue idx relative to item0, clamped to 0..65 (non-decreasing over the lanes)
            const uint32_t d = pf <= item0 ? 0u : (pf - item0 > 64 ? 65u : (uint32_t)(pf - item0));
            // cnt = number of lanes j with d_j <= lane: binary lifting over the lanes (bpermute), then the 64th
            uint32_t cnt = 0;
            for (uint32_t step = 32; step >= 1; step >>= 1) {
                const uint32_t dm = wv_shfl(d, (int)(cnt + step - 1u));
                if (dm <= (uint32_t)lane) cnt += step;
            }
            const uint32_t d63 = wv_shfl(d, 63);
            if (cnt == 63u && d63 <= (uint32_t)lane) cnt = 64u;
            const uint32_t dprev = wv_shfl(d, cnt ? (int)cnt - 1 : 0);
            if (!done && cnt < 64u) {
                cl = base + cnt;
                pcl = cnt ? item0 + dprev : pbase;       // (d is exact for every entry <= item0 + 63)
                done = true;
            }
            if (wv_ballot(!done) == 0ull) break;
            // all 64 sub-queues start at or below some lane's item (sparse queues: on the Zipf shape 64 wide items span
            // ~10^5 sub-queues, and walking them 64 at a time was 1.9 ms of one wave): two more windows, then every
            // lane left over searches the rest of its class by bisection (~20 dependent loads)
            pbase = ((uint64_t)wv_shfl((uint32_t)(pf >> 32), 63) << 32) | wv_shfl((uint32_t)pf, 63);
            base += 64;
            if (++windows < 3) continue;
            if (!done) {
                uint64_t lo = base, hi = WIDE ? 3 * a.n_chunks : 2 * a.n_chunks;      // prefix[lo] <= item < prefix[hi]
                while (hi - lo > 1) {
                    const uint64_t mid = (lo + hi) / 2;
                    if (prefix[mid] <= item) lo = mid; else hi = mid;
                }
                cl = lo;
                pcl = prefix[lo];
                done = true;
            }
            break;
        }
    }
    uint32_t rec = 0;
    uint64_t chunk = 0;
    if (have) {
        const uint64_t k = cl / a.n_chunks;
        chunk = cl - k * a.n_chunks;
        const uint32_t off = k == 0 ? TKF_MISSOFF0 : k == 1 ? TKF_MISSOFF1 : k == 2 ? TKF_MISSOFF2 : TKF_MISSOFF3;
        rec = a.miss_list[chunk * TKF_MISSCAP + off + (item - pcl)];
    }
    have_out = have; rec_out = rec; chunk_out = (uint32_t)chunk;
    return true;
}
template <bool WIDE>
TK_DEV void tk_merge_wave(const TkFlatArgs& a, uint64_t wave_id, int lane, uint32_t* mlds, const uint32_t* filt, TkMemoLog* ml = nullptr) {
    bool have; uint32_t rec, chunk;
    if (!tk_merge_find<WIDE>(a, wave_id, lane, have, rec, chunk)) return;
    tk_merge_items<WIDE>(a, have, rec, chunk, lane, mlds, filt, ml);
}

// The class 33..64 bytes: 64 pieces per wave, one lane each, in 64-entry LDS columns (tk_merge_lds<64>: 32 KB per wave, so
// every second wave of the wide kernel takes these, with its neighbour's columns).  The first form merged such a piece
// with the whole wave, one lane per byte, eight pieces per wave one after the other: fine while the class is rare, but
// text whose pieces are long by nature (CJK: a run of ideographs between two punctuation marks is ONE \p{L}+ piece of
// 3 bytes per character) puts most of its bytes here.  The lane's sub-queue by bisection over the class's prefix sums.
TK_DEV void tk_merge_items64(const TkFlatArgs& a, bool have, uint32_t rec, uint32_t chunk, int lane, uint32_t* mlds, const uint32_t* filt);
TK_DEV void tk_merge_wave_long3(const TkFlatArgs& a, uint64_t wave_id, int lane, uint32_t* mlds, const uint32_t* filt) {
    const uint64_t* prefix = a.miss_prefix;
    const uint64_t e0 = 3 * a.n_chunks, e1 = 4 * a.n_chunks;
    const uint64_t first = prefix[e0], total = prefix[e1];
    const uint64_t item0 = first + wave_id * 64;
    if (item0 >= total) return;                       // wave-uniform
    const uint64_t item = item0 + (uint64_t)lane;
    const bool have = item < total;
    uint32_t rec = 0;
    uint64_t chunk = 0;
    if (have) {
        uint64_t lo = e0, hi = e1;                    // prefix[lo] <= item < prefix[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) / 2;
            if (prefix[mid] <= item) lo = mid; else hi = mid;
        }
        chunk = lo - e0;
        rec = a.miss_list[chunk * TKF_MISSCAP + TKF_MISSOFF3 + (item - prefix[lo])];
    }
    tk_merge_items64(a, have, rec, (uint32_t)chunk, lane, mlds, filt);
}

// sequential merge of one piece of up to N bytes per lane (N = 8, 16, 32), parts in LDS: lane l owns column l of an
// N x 64 word array of KEYS and an N / 4 x 64 word array holding the piece's BYTES (word i * 64 + l, so whatever positions
// the lanes index, a wave's access is free of bank conflicts, and no lane ever touches another one's words: no barrier).
// Parts never move: bit i of `alive` says that a part starts at byte i and key[i] = (rank of the pair (part i, its
// successor) << PB) | i, or all ones -- so the leftmost smallest rank is one unsigned minimum over the column (v_min3: half
// an instruction per entry), and a merge is a handful of bit operations and a few LDS accesses.  The ID of a part needs no
// column of its own: a part of one byte is that byte, and a part of several bytes has a dead position right behind its first
// one -- key[i + 1], which no minimum may pick any more, holds 0x80000000 | id (anything from 0x80000000 up loses against
// every key; ids are below 2^21).  5 N bytes of LDS per lane instead of 8 N: the columns of a wave are what bounds the
// waves per CU of the merge kernels, and the waves are what hides their chains of dependent PAIR probes.  This COMPACT
// layout costs ~10 % more instructions per merge: it is used from 32-entry columns on (TKM_COMPACT), where it buys waves --
// tk_merge_wide_kernel runs 16 waves per CU (no PAIR filter in its LDS either) where it ran 8; the narrow classes keep a
// column of ids of their own (their block of 16 waves is the largest there is, and with twice the waves and no filter the
// kernel is slower: 2.76 against 2.49 ms on the mixed shape -- it is bound by its gathers, not by their latency).  (Parts in N-wide
// register arrays, shifted by predication -- the first form of this -- cost ~3x (N = 16) to ~4x (N = 32) the VALU issues
// per merge and 95 / 153 VGPRs.)
#ifndef TKM_COMPACT_MIN
#define TKM_COMPACT_MIN 32
#endif
#define TKM_COMPACT(N) ((N) >= TKM_COMPACT_MIN)           /* the classes whose waves per CU are bounded by their columns */
#define TKM_LDS_WORDS(N) (TKM_COMPACT(N) ? ((N) + (N) / 4) * 64 : 2 * (N) * 64)
#ifdef TKM_ABLATE   /* timing-only experiments on the merge kernels (never defined in the shipped build) */
#define TKM_AB(a, bit) (((a).dbg_ablate & (bit)) != 0)
#else
#define TKM_AB(a, bit) false
#endif
template <int N> struct TkmAlive { typedef uint32_t type; };
template <> struct TkmAlive<64> { typedef uint64_t type; };
template <> struct TkmAlive<128> { typedef unsigned __int128 type; };
// mask helpers: the low n bits (n up to the width), one bit, any bit set, lowest set bit cleared (the type is chosen by
// a null pointer argument)
template <typename T> TK_DEV T tkm_low(uint32_t n, const T*) { return n >= 8u * (uint32_t)sizeof(T) ? (T)~(T)0 : (T)(((T)1 << n) - (T)1); }
template <typename T> TK_DEV T tkm_bit(uint32_t n, const T*) { return (T)((T)1 << n); }
template <typename T> TK_DEV bool tkm_any(T v) { return v != (T)0; }
template <typename T> TK_DEV T tkm_clear_lowest(T v) { return (T)(v & (v - (T)1)); }
TK_DEV uint32_t tkm_ctz(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
TK_DEV uint32_t tkm_ctz(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
TK_DEV uint32_t tkm_msb(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
TK_DEV uint32_t tkm_msb(uint64_t v) { return 63u - (uint32_t)__builtin_clzll(v); }
TK_DEV uint32_t tkm_popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
TK_DEV uint32_t tkm_popc(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
TK_DEV uint32_t tkm_ctz(unsigned __int128 v) { const uint64_t lo = (uint64_t)v; return lo ? (uint32_t)__builtin_ctzll(lo) : 64u + (uint32_t)__builtin_ctzll((uint64_t)(v >> 64)); }
TK_DEV uint32_t tkm_msb(unsigned __int128 v) { const uint64_t hi = (uint64_t)(v >> 64); return hi ? 127u - (uint32_t)__builtin_clzll(hi) : 63u - (uint32_t)__builtin_clzll((uint64_t)v); }
TK_DEV uint32_t tkm_popc(unsigned __int128 v) { return (uint32_t)__builtin_popcountll((uint64_t)v) + (uint32_t)__builtin_popcountll((uint64_t)(v >> 64)); }

// N = 8, 16, 32, 64, 128 (64: the class 33..64 bytes, keys carry six position bits and the live mask is 64 bits wide;
// 128: the long-piece records of 65..128 bytes, seven position bits, a 128-bit mask)
template <int N>
TK_DEV uint32_t tk_merge_lds(const TkFlatArgs& a, const uint32_t* filt, bool mine, const uint32_t* kk, uint32_t len,
                             uint32_t* out, uint32_t* mlds, int lane, TkMemoLog* ml = nullptr) {
    typedef typename TkmAlive<N>::type alive_t;
    constexpr uint32_t PB = N > 64 ? 7u : N > 32 ? 6u : 5u, PM = (1u << PB) - 1u;
    constexpr uint32_t DEAD = 0x80000000u;                  // key[i] >= DEAD: no pair starts at i (all ones), or the id of the part that begins at i - 1
    const TkTablesView& t = a.t;
    constexpr bool C = TKM_COMPACT(N);
    // compact: keys, then the piece's bytes; otherwise ids, then keys (a column of ids of its own: fewer instructions per merge,
    // kept where the block cannot hold more waves anyway)
    uint32_t* keyc = C ? mlds + lane : mlds + N * 64 + lane;
    uint32_t* bytc = mlds + N * 64 + lane;
    uint32_t* tokc = mlds + lane;
    const uint32_t n = mine ? len : 0u;
    if (C) {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) bytc[q * 64] = kk[q];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t b = (kk[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const uint32_t b1 = i + 1 < N ? (kk[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xFFu : 0u;
        uint32_t key = 0xFFFFFFFFu;
        if ((uint32_t)(i + 1) < n && !TKM_AB(a, 2048)) {
            const uint32_t r = t.pair2[b | (b1 << 8)];
            if (r != TK_RANK_MAX) key = (r << PB) | (uint32_t)i;
        }
        if (!C) tokc[i * 64] = b;
        keyc[i * 64] = key;
    }
    alive_t alive = tkm_low(n, (const alive_t*)nullptr);
    // id of the live part at position x (see above)
    auto tok_at = [&](uint32_t x, alive_t al) -> uint32_t {
        if (!C) return tokc[x * 64u];
        const bool multi = x + 1u < n && !tkm_any((alive_t)(al & tkm_bit(x + 1u < (uint32_t)N ? x + 1u : 0u, (const alive_t*)nullptr)));
        const uint32_t v = multi ? keyc[(x + 1u) * 64u] : bytc[(x >> 2) * 64u];
        return multi ? v & ~DEAD : (v >> (8u * (x & 3u))) & 0xFFu;
    };
    bool active = mine && !TKM_AB(a, 1024);
    while (wv_ballot(active)) {
        if (active) {
            uint32_t best = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < N - 1; ++i) {
                const uint32_t k = keyc[i * 64];
                best = k < best ? k : best;
            }
            if (best >= DEAD) {
                active = false;
            } else {
                // the parts at bi and at the next live position j become one part (at bi) whose id is the rank
                const uint32_t bi = best & PM, rank = best >> PB;
                const alive_t above = alive & ~tkm_low(bi + 1u, (const alive_t*)nullptr);       // bi <= N - 2
                const uint32_t j = tkm_ctz(above);                            // exists: key[bi] was a pair
                const alive_t above2 = tkm_clear_lowest(above);
                const alive_t below = alive & tkm_low(bi, (const alive_t*)nullptr);
                const bool has_next = tkm_any(above2), has_prev = tkm_any(below);
                const uint32_t k = has_next ? tkm_ctz(above2) : 0u;
                const uint32_t p = has_prev ? tkm_msb(below) : 0u;
                const uint32_t tn = tok_at(k, alive), tp = tok_at(p, alive);   // (neither lives in a word written below: p + 1 <= bi, k + 1 > j)
                alive = alive & ~tkm_bit(j, (const alive_t*)nullptr);
                keyc[j * 64] = 0xFFFFFFFFu;                                   // j's own pair is gone ...
                if (C) keyc[(bi + 1u) * 64] = DEAD | rank;                    // ... and the word behind bi (dead: j or an older one) takes the new id
                else tokc[bi * 64] = rank;
                uint32_t r_next = TK_RANK_MAX, r_prev = TK_RANK_MAX;
                tk_probe_pair_x2f(t, filt, has_next, rank, tn, has_prev, tp, rank, r_next, r_prev);
                keyc[bi * 64] = r_next == TK_RANK_MAX ? 0xFFFFFFFFu : ((r_next << PB) | bi);
                if (has_prev) keyc[p * 64] = r_prev == TK_RANK_MAX ? 0xFFFFFFFFu : ((r_prev << PB) | p);
            }
        }
    }
    const uint32_t np = tkm_popc(alive);
    if (!C && ml != nullptr) {
        // MEMO (tk_hash.h): what this piece merged into, for the look-ups of the NEXT call -- exact key (the bytes, zero padded,
        // and the length), at most TK_MEMO_MAXIDS ranks.  The entry goes into this wave's stretch of a LOG (a full stretch drops
        // it, which only delays the entry by a call); tk_memo_commit_kernel puts the log into the table behind the merge kernels
        // -- a hot new word arrives in hundreds of lanes at once, and taking table slots with atomics from here would serialise
        // them.
        bool ins = mine && np <= TK_MEMO_MAXIDS;
        uint32_t k[4] = {0u, 0u, 0u, 0u};
        tk_memo_entry* slot = nullptr;
        if (wv_ballot(ins) && ml->n < ml->cap) {
            if (ins) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int keep = (int)len - 4 * q;
                    k[q] = (q < N / 4 && keep > 0) ? (kk[q < N / 4 ? q : 0] & (keep >= 4 ? 0xFFFFFFFFu : ((1u << (8 * keep)) - 1u))) : 0u;
                }
                slot = a.memo_tab + (tk_memo_slot(tk_key_hash(t.key_hash_mode, k[0], k[1], k[2], k[3], len)) & a.memo_mask);
                // a slot that already carries a claim of THIS call (a record index: no tag bit; the commit kernel leaves every claimed
                // slot tagged) is most often the same word merged somewhere else a moment ago -- a new word comes in hundreds of lanes:
                // no second record for it (a colliding word waits a call).  The load may be stale: then there is a duplicate, as before.
                // Only on the call that fills an empty table (no look-ups yet: EVERY merged piece comes here, and the hot words would fill
                // the log with copies of themselves -- with the check the second batch already runs warm, 2.32 -> 1.54 ms); later calls
                // log only what the table missed, and the extra gather would cost them 2 %.
                if (a.memo_probe == 0u) {
                    const uint32_t cw = slot->w4;
                    if (cw != 0u && (cw & TK_MEMO_TAG) == 0u) ins = false;
                }
            }
        }
        const uint64_t IB = wv_ballot(ins);
        if (IB && ml->n < ml->cap) {
            const uint32_t at = ml->n + (uint32_t)tk_popc64(IB & tk_lowmask(lane));
            if (ins && at < ml->cap) {
                uint32_t r5[5] = {0u, 0u, 0u, 0u, 0u};
                alive_t rem = alive;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    if ((uint32_t)q < np) r5[q] = tokc[tkm_ctz(rem) * 64u];
                    rem = tkm_clear_lowest(rem);
                }
                uint32_t v[3], w4;
                tk_memo_pack(r5, np, len, &w4, v);
                tk_memo_entry* w = ml->base + at;           // (a log record is the entry it will become)
                wv_store16(w->k, k[0], k[1], k[2], k[3]);
                wv_store16(&w->w4, w4, v[0], v[1], v[2]);
                // ... and CLAIMS its slot: the record's index into the slot's claim word, a plain store -- of all the records that want
                // a slot in this call one index stays, and tk_memo_commit_kernel lets that record write (nothing reads the table
                // while the merge and commit kernels run)
                slot->w4 = (uint32_t)(w - a.memo_log);
            }
            const uint32_t nn = ml->n + (uint32_t)tk_popc64(IB);
            ml->n = nn < ml->cap ? nn : ml->cap;
        }
    }
    if (mine && !TKM_AB(a, 256)) {
        // the piece's `len` slots: its np ids, then holes -- gathered into registers and stored four at a time (the cost
        // of a scattered store is per lane and instruction, not per byte), single words only for the last len % 4
        alive_t rem = alive;
#pragma unroll
        for (int q0 = 0; q0 < N; q0 += 4) {
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t pos = tkm_any(rem) ? tkm_ctz(rem) : 0u;
                const uint32_t id = tok_at(pos, alive) + t.num_special;
                v[q] = (uint32_t)(q0 + q) < np ? id : TKF_HOLE;
                rem = tkm_clear_lowest(rem);
            }
            if ((uint32_t)q0 + 4u <= len) {
                wv_store16(out + q0, v[0], v[1], v[2], v[3]);
            } else {
                if ((uint32_t)q0 < len) out[q0] = v[0];
                if ((uint32_t)q0 + 1u < len) out[q0 + 1] = v[1];
                if ((uint32_t)q0 + 2u < len) out[q0 + 2] = v[2];
            }
        }
    }
    return mine ? len - np : 0u;
}

// the document of a merged piece loses `holes` ids.  Queued pieces are in text order inside a sub-queue, so neighbouring
// lanes mostly share their document: the counts of a run of lanes with the same document are summed in the wave (one
// scan) and its last lane does the one atomic -- 64 atomics on two or three addresses serialise in the L2.
TK_DEV void tk_merge_holes(const TkFlatArgs& a, bool have, uint32_t holes, uint32_t chunk, int64_t g, int lane) {
    if (wv_ballot(have && holes != 0u) && !TKM_AB(a, 512)) {
        // largest d with doc_offs[d] <= g: the documents from first_doc[chunk] - 1 on start at or above the region
        uint64_t d = 0;
        if (have) {
            d = a.first_doc[chunk];
            d = d > 0 ? d - 1 : 0;
            // four offsets per round trip (one at a time this was a chain of up to four dependent loads per lane on 512-byte
            // documents -- a tenth of the merge kernel there); doc_offs[n_docs] = n_bytes > g ends the search by itself
            for (;;) {
                const uint64_t i1 = d + 1 < a.n_docs ? d + 1 : a.n_docs, i2 = d + 2 < a.n_docs ? d + 2 : a.n_docs;
                const uint64_t i3 = d + 3 < a.n_docs ? d + 3 : a.n_docs, i4 = d + 4 < a.n_docs ? d + 4 : a.n_docs;
                uint64_t o1 = a.doc_offs[i1], o2 = a.doc_offs[i2], o3 = a.doc_offs[i3], o4 = a.doc_offs[i4];
                const uint32_t cnt = ((int64_t)o1 <= g ? 1u : 0u) + ((int64_t)o2 <= g ? 1u : 0u) + ((int64_t)o3 <= g ? 1u : 0u) + ((int64_t)o4 <= g ? 1u : 0u);
                d += cnt;                                   // (the offsets do not decrease: the count is the number of steps)
                if (cnt < 4u) break;
            }
        }
        const uint32_t key = have ? (uint32_t)d : 0xFFFFFFFFu;             // (documents are numbered below 2^32 - 1)
        const uint32_t kprev = wv_shfl(key, lane > 0 ? lane - 1 : 0), knext = wv_shfl(key, lane < 63 ? lane + 1 : 63);
        const bool head = lane == 0 || kprev != key, tail = lane == 63 || knext != key;
        const uint64_t HEADS = wv_ballot(head);
        const uint32_t P = wv_scan_incl_u32(have ? holes : 0u);
        const int start = tk_msb64(HEADS & (tk_lowmask(lane) | (1ull << lane)));      // the head of this lane's run
        const uint32_t before = wv_shfl(P, start > 0 ? start - 1 : 0);
        const uint32_t sum = P - (start > 0 ? before : 0u);
        if (have && tail && sum) wv_atomic_add(a.holes + d, sum);
    }
}

// the class 33..64 bytes, one lane per piece like the shorter classes: 64-entry columns (32 KB of LDS per wave)
TK_DEV void tk_merge_items64(const TkFlatArgs& a, bool have, uint32_t rec, uint32_t chunk, int lane, uint32_t* mlds, const uint32_t* filt) {
    const uint32_t pos = TKF_REC_POS(rec), len = TKF_REC_LEN(rec), slot = TKF_REC_SLOT(rec);
    const int64_t g = (int64_t)chunk * TKF_COMMIT - TKF_HL + (int64_t)pos;   // first byte of the piece
    uint32_t* out = a.tmp + (uint64_t)chunk * TKF_STRIDE + slot;
    uint32_t kk[16];
    for (int q = 0; q < 16; ++q) kk[q] = 0u;
    if (have) {
        if (g + 64 <= (int64_t)a.n_bytes) {
            wv_load16(a.bytes + g, kk);
            wv_load16(a.bytes + g + 16, kk + 4);
            wv_load16(a.bytes + g + 32, kk + 8);
            wv_load16(a.bytes + g + 48, kk + 12);
        } else {
            for (uint32_t q = 0; q < len; ++q) kk[q >> 2] |= (uint32_t)a.bytes[g + q] << (8 * (q & 3));
        }
    }
    uint32_t holes = 0;
    if (wv_ballot(have)) holes = tk_merge_lds<64>(a, filt, have, kk, len, out, mlds, lane);
    tk_merge_holes(a, have, holes, chunk, g, lane);
}

// the long-piece records that are no vocabulary keys (marked by tk_flat_long_wave): 64 consecutive records per wave, one
// lane each, in N-entry LDS columns -- N = 128 for 65..128 bytes (64 KB per wave: two waves and the PAIR filter fill a CU's
// LDS) -- still 128 chains per CU where the single-wave merge runs a dozen
template <int N>
TK_DEV void tk_merge_long_wave(const TkFlatArgs& a, uint64_t wave_id, int lane, uint32_t* mlds, const uint32_t* filt) {
    const uint64_t n = *a.long_count < a.long_cap ? *a.long_count : a.long_cap;
    const uint64_t item = wave_id * 64 + (uint64_t)lane;
    if (wave_id * 64 >= n) return;                    // wave-uniform
    TkFlatLongRec lr;
    lr.pos = 0; lr.chunk = 0; lr.slot = 0; lr.len = 0; lr.reserved = 0;
    if (item < n) lr = a.long_recs[item];
    // marked by tk_flat_long_wave, and of this kernel's length class
    const bool have = (lr.reserved & 0x80000000u) != 0u && lr.len <= (uint32_t)N && lr.len > (uint32_t)N / 2u;
    lr.reserved &= 0x3FFFFFFFu;
    if (wv_ballot(have) == 0ull) return;
    const int64_t g = (int64_t)lr.pos;
    uint32_t* out = a.tmp + (uint64_t)lr.chunk * TKF_STRIDE + lr.slot;
    uint32_t kk[N / 4];
    for (int q = 0; q < N / 4; ++q) kk[q] = 0u;
    if (have) {
        if (g + N <= (int64_t)a.n_bytes) {
#pragma unroll
            for (int q = 0; q < N / 16; ++q) wv_load16(a.bytes + g + 16 * q, kk + 4 * q);
        } else {
            for (uint32_t q = 0; q < lr.len; ++q) kk[q >> 2] |= (uint32_t)a.bytes[g + q] << (8 * (q & 3));
        }
    }
    uint32_t holes = tk_merge_lds<N>(a, filt, have, kk, lr.len, out, mlds, lane);
    if (have) {
        // (a piece whose end the chunk did not see reserved TKF_LONGCAP slots: the rest are holes too)
        for (uint32_t k = lr.len; k < lr.reserved; ++k) out[k] = TKF_HOLE;
        holes += lr.reserved - lr.len;
    }
    tk_merge_holes(a, have, holes, lr.chunk, g, lane);
}

template <bool WIDE>
TK_DEV void tk_merge_items(const TkFlatArgs& a, bool have, uint32_t rec, uint32_t chunk, int lane, uint32_t* mlds, const uint32_t* filt, TkMemoLog* ml) {
    const TkTablesView& t = a.t;
    const uint32_t pos = TKF_REC_POS(rec), len = TKF_REC_LEN(rec), slot = TKF_REC_SLOT(rec);
    const int64_t g = (int64_t)chunk * TKF_COMMIT - TKF_HL + (int64_t)pos;   // first byte of the piece
    uint32_t* out = a.tmp + (uint64_t)chunk * TKF_STRIDE + slot;
    uint32_t holes = 0;

    // ---- pieces of up to 32 bytes: one lane each, parts in LDS columns (8 / 16 / 32 entries by class) ------
    {
        const uint32_t cap = WIDE ? 32u : TKM_SHORT;
        const bool inregs = have && len <= cap;
        uint32_t kk[WIDE ? 8 : 4];
        for (int q = 0; q < (WIDE ? 8 : 4); ++q) kk[q] = 0u;
        if (inregs) {
            if (g + (int64_t)cap <= (int64_t)a.n_bytes) {
                wv_load16(a.bytes + g, kk);
                if (WIDE) wv_load16(a.bytes + g + 16, kk + 4);
            } else {
                for (uint32_t q = 0; q < len; ++q) kk[q >> 2] |= (uint32_t)a.bytes[g + q] << (8 * (q & 3));
            }
        }
        if (!WIDE) {
            const bool s8 = inregs && len <= 8u, s16 = inregs && len > 8u;
            if (wv_ballot(s8)) holes += tk_merge_lds<8>(a, filt, s8, kk, len, out, mlds, lane, ml);
            if (wv_ballot(s16)) holes += tk_merge_lds<16>(a, filt, s16, kk, len, out, mlds, lane, ml);
        } else {
            if (wv_ballot(inregs)) holes += tk_merge_lds<32>(a, filt, inregs, kk, len, out, mlds, lane);
        }
    }

    // ---- longer pieces (33..64 bytes; 17..64 if they ever reach the narrow kernel): one at a time, one lane per byte
    uint64_t LONGM = wv_ballot(have && len > (WIDE ? 32u : TKM_SHORT));
    while (LONGM) {
        const int src = tk_ctz64(LONGM);
        LONGM &= LONGM - 1ull;
        const uint32_t plen = wv_shfl(len, src);
        const uint32_t glo = wv_shfl((uint32_t)g, src), ghi = wv_shfl((uint32_t)((uint64_t)g >> 32), src);
        const int64_t pg = (int64_t)(((uint64_t)ghi << 32) | glo);
        const uint32_t pchunk = wv_shfl(chunk, src), pslot = wv_shfl(slot, src);
        uint32_t* pout = a.tmp + (uint64_t)pchunk * TKF_STRIDE + pslot;
        const int pe = (int)plen;
        const bool inmiss = lane < pe;
        const uint32_t b0 = inmiss ? (uint32_t)a.bytes[pg + lane] : 0u;
        const uint32_t b1 = wv_up1(b0);
        uint64_t A = wv_ballot(inmiss);
        uint32_t tok = b0;
        uint32_t prank = TK_RANK_MAX;
        if (inmiss && lane + 1 < pe) prank = t.pair2[b0 | (b1 << 8)];
        for (int round = 0; round < 64; ++round) {
            const uint32_t key = prank == TK_RANK_MAX ? 0xFFFFFFFFu : ((prank << 6) | (uint32_t)lane);
            const uint32_t segmin = wv_min_u32(key);
            if (segmin == 0xFFFFFFFFu) break;
            const bool winner = key == segmin;
            const uint64_t Wm = wv_ballot(winner);
            const bool alive = tk_bit(A, lane);
            const uint64_t below = A & tk_lowmask(lane);
            const bool dead = alive && below && tk_bit(Wm, tk_msb64(below));
            const uint64_t Dm = wv_ballot(dead);
            A &= ~Dm;
            if (winner) tok = prank;
            if (dead) prank = TK_RANK_MAX;
            const uint64_t z = lane < 63 ? (A >> (lane + 1)) : 0ull;
            const int na = z ? lane + 1 + tk_ctz64(z) : 64;
            const bool has_next = na < pe;
            const uint32_t tn = wv_shfl(tok, na < 64 ? na : lane);
            const bool need = tk_bit(A, lane) && (winner || (has_next && tk_bit(Wm, na)));
            if (need) prank = has_next ? tk_probe_pair_f(t, filt, tok, tn) : TK_RANK_MAX;
        }
        const uint32_t k = (uint32_t)tk_popc64(A);
        if (inmiss && tk_bit(A, lane)) pout[tk_popc64(A & tk_lowmask(lane))] = tok + t.num_special;
        if (inmiss && (uint32_t)lane >= k) pout[lane] = TKF_HOLE;
        if (lane == src) holes = plen - k;
    }

    tk_merge_holes(a, have, holes, chunk, g, lane);
}

// The log of a call's new entries (tk_merge_lds) into the table, without a single atomic -- a word that is new in this call is new in
// hundreds of records, and that many read-modify-writes on one address serialise (measured: up to 1.4 ms for a million records).
// Record i = j-th record of merge wave w, i = w * per_wave + j, j < counts[w].  Step 1, in the merge kernel where the record is
// written: its INDEX goes into the claim word (w4) of its slot (plain stores: one of them stays).  Step 2, tk_memo_commit_kernel behind
// the merge kernels: the record whose index is the one that stayed owns the slot and writes key and ids.  One writer per slot and call
// by construction; nothing reads the table while the merge and commit kernels run.
TK_DEV bool tk_memo_log_live(const uint32_t* counts, uint32_t per_wave, uint32_t i) { return i % per_wave < counts[i / per_wave]; }
TK_DEV uint32_t tk_memo_slot_of(const tk_memo_entry& r, uint32_t key_hash_mode, uint32_t mask) {
    return tk_memo_slot(tk_key_hash(key_hash_mode, r.k[0], r.k[1], r.k[2], r.k[3], tk_memo_len(r.v[2]))) & mask;
}
TK_DEV void tk_memo_commit_one(tk_memo_entry* tab, const tk_memo_entry* log, uint32_t i, uint32_t key_hash_mode, uint32_t mask) {
    const tk_memo_entry r = log[i];
    tk_memo_entry* w = tab + tk_memo_slot_of(r, key_hash_mode, mask);
    if (w->w4 == i) {
        wv_store16(w->k, r.k[0], r.k[1], r.k[2], r.k[3]);
        wv_store16(&w->w4, r.w4, r.v[0], r.v[1], r.v[2]);
    }
}

#endif
