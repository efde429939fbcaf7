// tk_long_impl.h -- the ROUND-BASED byte-pair merges of one long piece by a whole workgroup (16 waves): compacting rounds
// (tkl_block_merge) and lazy rounds (tks_block_merge).  Device code of csrc/tk_long.hip, kept in a header -- like
// tk_flat_impl.h and tk_encode_impl.h -- so that the CPU emulator of the test-suite (tests/emu: 16 emulated waves, the
// workgroup barrier as a scheduling point, scratch and LDS between guard zones, AddressSanitizer / UBSan builds) runs this
// very source: the sel-word carry ripple across waves, the u16 occurrence lists, the Blr / Brr scratch indexing and the
// alive-bit updates are index arithmetic that no GPU sanitizer can check.
//
// The algorithm and its exactness argument: tk_long.hip's header and tools/batched_merge_model.py.
#ifndef TK_LONG_IMPL_H
#define TK_LONG_IMPL_H
#include <stdint.h>

#include "tk_encode_impl.h"

#define TKL_THREADS 1024
#define TKL_WAVES 16
#define TKL_MAXSTEPS 32                      /* 16 waves x 32 steps x 64 lanes = TK_LONG_MAX parts */
#define TKL_SELWORDS (TKL_WAVES * TKL_MAXSTEPS)

struct TklShared {
    uint64_t sel[TKL_SELWORDS + 2];          // [1 + g]: parts of step g (64 consecutive parts) that merge with their successor
    uint32_t wmin[TKL_WAVES];                // per wave: minimum pair rank of the parts it wrote
    uint32_t lead[TKL_WAVES];                // per wave: length of the candidate run that starts at its first part
    uint32_t full[TKL_WAVES];                // per wave: its whole range is one candidate run
    uint32_t tail_in[TKL_WAVES];             // per wave: its last part is a candidate
    uint32_t tail_par[TKL_WAVES];            // ... and the length of the run ending there is odd (fresh start assumed)
    uint32_t wunder[TKL_WAVES];              // per wave: leftmost occurrence that creates a pair below r*
    uint32_t wcnt[TKL_WAVES];                // per wave: parts that merge
    uint32_t doc;                            // the job ticket, thread 0 -> everybody
    uint16_t occ[TKL_WAVES][64 * TKL_MAXSTEPS / 2 + 64];   // per wave: its occurrences (part index relative to the wave's range), ascending
};

TK_DEV uint32_t tkl_wave_min(uint32_t v) { return wv_min_u32(v); }

// parts of a 64-bit candidate mask that sit at an EVEN offset of their run of consecutive candidates.  in_run: the run
// continues from the part before bit 0, par = parity of its length so far.
TK_DEV uint64_t tkl_even_offsets(uint64_t C, uint32_t in_run, uint32_t par) {
    const uint64_t E = 0x5555555555555555ull;
    uint64_t S = C & ~(C << 1);                       // run starts
    uint64_t seven, sodd;
    if (in_run && (C & 1ull)) {
        S &= ~1ull;                                   // bit 0 continues a run: its virtual start has the parity of `par`
        seven = (S & E) | (par ? 0ull : 1ull);
        sodd = (S & ~E) | (par ? 1ull : 0ull);
    } else {
        seven = S & E;
        sodd = S & ~E;
    }
    (void)sodd;
    const uint64_t reven = ((C + seven) ^ C) & C;     // the runs that start at an even position (carry ripple)
    const uint64_t rodd = C & ~reven;
    return (reven & E) | (rodd & ~E);
}

// The round-based merge of the piece bytes[0 .. n), 64 < n <= TK_LONG_MAX, by all 16 waves.  Final ids (shifted) go to
// out[0 ..); returns their number (block-uniform).  scratch: 4 n words (tok | rk | left ranks | right ranks).
TK_DEV uint32_t tkl_block_merge(const TkTablesView& t, const uint8_t* bytes, uint32_t n0, uint32_t* scratch, uint32_t* out,
                                    TklShared& L) {
    const int lane = wv_lane();
    const uint32_t wv = wv_first(wv_tid() >> 6);
    uint32_t* Atok = scratch;
    uint32_t* Ark = scratch + n0;
    uint32_t* Blr = scratch + 2 * (size_t)n0;
    uint32_t* Brr = scratch + 3 * (size_t)n0;
    uint32_t n = n0;

    // ---- start: one part per byte, pair ranks from PAIR2 ----
    {
        uint32_t m = TK_RANK_MAX;
        for (uint32_t i = wv_tid(); i < n; i += TKL_THREADS) {
            const uint32_t b0 = bytes[i];
            const uint32_t r = (i + 1u < n) ? t.pair2[b0 | ((uint32_t)bytes[i + 1u] << 8)] : TK_RANK_MAX;
            Atok[i] = b0;
            Ark[i] = r;
            m = r < m ? r : m;
        }
        m = tkl_wave_min(m);
        if (lane == 0) L.wmin[wv] = m;
        if (wv_tid() < 2) L.sel[wv_tid() ? TKL_SELWORDS + 1 : 0] = 0ull;
    }
    wv_block_sync();

    for (uint32_t round = 0; round <= n0; ++round) {              // (every round merges at least one pair: at most n0 - 1 rounds)
        // r* = the minimum over what every wave wrote last
        uint32_t rstar = lane < TKL_WAVES ? L.wmin[lane] : TK_RANK_MAX;
        rstar = tkl_wave_min(rstar);
        if (rstar == TK_RANK_MAX) break;                          // block-uniform
        const uint32_t per = ((n + TKL_THREADS - 1u) / TKL_THREADS) * 64u;   // parts per wave
        const uint32_t steps = per / 64u;                          // <= TKL_MAXSTEPS
        const uint32_t base = wv * per;
        const uint32_t g0 = wv * steps;                            // index of the wave's first step among all steps

        // ---- R0: the wave's range into registers ----
        uint32_t tok[TKL_MAXSTEPS], rk[TKL_MAXSTEPS];
#pragma unroll
        for (int s = 0; s < TKL_MAXSTEPS; ++s) {
            tok[s] = 0u; rk[s] = TK_RANK_MAX;
            if ((uint32_t)s < steps) {
                const uint32_t i = base + 64u * (uint32_t)s + (uint32_t)lane;
                if (i < n) { tok[s] = Atok[i]; rk[s] = Ark[i]; }
            }
        }

        // ---- R2: candidates, even offsets of their runs (fresh start at the wave's first part) ----
        {
            uint32_t in_run = 0, par = 0, lead = 0, lead_open = 1, full = 1;
#pragma unroll
            for (int s = 0; s < TKL_MAXSTEPS; ++s) {
                if ((uint32_t)s < steps) {
                    const uint64_t C = wv_ballot(rk[s] == rstar);
                    const uint64_t sel = tkl_even_offsets(C, in_run, par);
                    if (lane == 0) L.sel[1 + g0 + s] = sel;
                    if (lead_open) {
                        if (C == ~0ull) lead += 64u;
                        else { lead += (uint32_t)tk_ctz64(~C); lead_open = 0; }
                    }
                    if (C != ~0ull) full = 0;
                    in_run = (uint32_t)(C >> 63);
                    par = in_run ? (uint32_t)(sel >> 63) : 0u;     // last part at an even offset <=> odd length so far
                }
            }
            if (lane == 0) { L.lead[wv] = lead; L.full[wv] = full; L.tail_in[wv] = in_run; L.tail_par[wv] = par; }
            if (wv == 0 && lane == 0) L.sel[1 + TKL_WAVES * steps] = 0ull;   // the word behind the last step
        }
        wv_block_sync();
        {
            // the run that reaches this wave from the left: walk the waves before it (per is even: a full wave keeps the parity)
            uint32_t cin = 0, cpar = 0;
            for (uint32_t v = 0; v < wv; ++v) {
                if (L.full[v]) { if (!cin) { cin = 1; cpar = 0; } }
                else { cin = L.tail_in[v]; cpar = L.tail_par[v]; }
            }
            const uint32_t lead = L.lead[wv];
            if (cin && cpar && lead) {
                // an odd number of candidates before the wave's first part: inside its leading run the OTHER offsets merge
#pragma unroll
                for (int s = 0; s < TKL_MAXSTEPS; ++s) {
                    if ((uint32_t)s < steps && 64u * (uint32_t)s < lead && lane == 0) {
                        const uint32_t k = lead - 64u * (uint32_t)s;
                        L.sel[1 + g0 + s] ^= tk_lowmask(k >= 64u ? 64 : (int)k);
                    }
                }
            }
        }
        wv_block_sync();

        // ---- R3: the wave's occurrences as a list (ascending), then ONE LANE PER OCCURRENCE: the two pairs it creates, probed as
        // the sequential order would see them.  (A loop over the steps with the probes inside pays the chain neighbour loads
        // -> probes -> stores once per step that has an occurrence; the list pays it once per 64 occurrences.)
        uint16_t* occ = L.occ[wv];
        uint32_t cntw = 0;
#pragma unroll 1
        for (uint32_t s = 0; s < steps; ++s) {
            const uint64_t sel = L.sel[1 + g0 + s];
            if (sel) {                                             // wave-uniform
                if (tk_bit(sel, lane)) occ[cntw + (uint32_t)tk_popc64(sel & tk_lowmask(lane))] = (uint16_t)(64u * s + (uint32_t)lane);
                cntw += (uint32_t)tk_popc64(sel);
            }
        }
        wv_lds_sync();
        uint32_t umin = 0xFFFFFFFFu;
        uint32_t lr0 = TK_RANK_MAX, rr0 = TK_RANK_MAX;               // the first batch keeps its results in registers
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < cntw) {
                const uint32_t i = base + (uint32_t)occ[k];
                uint32_t lr = TK_RANK_MAX, rr = TK_RANK_MAX;
                uint32_t tl = 0, tr = 0;
                const bool hasl = i > 0u, hasr = i + 2u < n;
                if (hasl) {
                    // the part before: already merged if the occurrence two parts back merges
                    const bool lm = i >= 2u && tk_bit(L.sel[1 + ((i - 2u) >> 6)], (int)((i - 2u) & 63u));
                    tl = lm ? rstar : Atok[i - 1u];
                }
                if (hasr) tr = Atok[i + 2u];
                tk_probe_pair_x2(t, hasl ? tl : 0u, rstar, rstar, hasr ? tr : 0u, lr, rr);
                if (!hasl) lr = TK_RANK_MAX;
                if (!hasr) rr = TK_RANK_MAX;
                if (k0 == 0u) { lr0 = lr; rr0 = rr; }
                else { Blr[i] = lr; Brr[i] = rr; }
                if (lr < rstar || rr < rstar) umin = i < umin ? i : umin;
            }
        }
        umin = tkl_wave_min(umin);
        if (lane == 0) L.wunder[wv] = umin;
        wv_block_sync();
        uint32_t ustar = lane < TKL_WAVES ? L.wunder[lane] : 0xFFFFFFFFu;
        ustar = tkl_wave_min(ustar);

        // ---- R4: cut the round behind the first occurrence that creates a pair below r* (the lists are ascending: a prefix
        // of every list stays); count ----
        if (ustar != 0xFFFFFFFFu) {                                // block-uniform, rare
            uint32_t kept = 0;
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
                const uint32_t k = k0 + (uint32_t)lane;
                kept += (uint32_t)tk_popc64(wv_ballot(k < cntw && base + (uint32_t)occ[k < cntw ? k : 0u] <= ustar));
            }
            cntw = kept;
#pragma unroll 1
            for (uint32_t s = 0; s < steps; ++s) {
                const uint32_t first = base + 64u * s;
                const uint64_t keep = ustar < first ? 0ull : tk_lowmask(ustar - first >= 63u ? 64 : (int)(ustar - first + 1u));
                if (lane == 0) L.sel[1 + g0 + s] &= keep;
            }
        }
        if (lane == 0) L.wcnt[wv] = cntw;
        wv_block_sync();   // every old part and neighbour has been read: the arrays may be rewritten
        uint32_t before = 0, total = 0;
        for (uint32_t v = 0; v < TKL_WAVES; ++v) {
            const uint32_t c = L.wcnt[v];
            if (v < wv) before += c;
            total += c;
        }

        // ---- R5: commit + compact (new index = old index - occurrences before it), minimum of the new ranks.  Every slot of
        // the rank array is written exactly once: an unchanged pair by its part (a), the pair behind an occurrence and the
        // pair in front of it by the occurrence (b) -- the pair between two back-to-back occurrences by the second one. ----
        uint32_t mnew = TK_RANK_MAX;
        {
            uint32_t run = before;
#pragma unroll
            for (int s = 0; s < TKL_MAXSTEPS; ++s) {
                if ((uint32_t)s < steps) {
                    const uint32_t g = g0 + (uint32_t)s;
                    const uint64_t sel = L.sel[1 + g], prev = L.sel[g], next = L.sel[2 + g];
                    const uint64_t dead = (sel << 1) | (prev >> 63);           // the part after an occurrence is consumed
                    const uint64_t sel1 = (sel >> 1) | (next << 63);           // an occurrence right after me
                    const uint32_t i = base + 64u * (uint32_t)s + (uint32_t)lane;
                    if (i < n && !tk_bit(dead, lane)) {
                        const uint32_t ni = i - (run + (uint32_t)tk_popc64(sel & tk_lowmask(lane)));
                        const bool me = tk_bit(sel, lane);
                        Atok[ni] = me ? rstar : tok[s];
                        if (!me && !tk_bit(sel1, lane)) {
                            Ark[ni] = rk[s];
                            mnew = rk[s] < mnew ? rk[s] : mnew;
                        }
                    }
                    run += (uint32_t)tk_popc64(sel);
                }
            }
        }
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < cntw) {
                const uint32_t i = base + (uint32_t)occ[k];
                uint32_t lr = lr0, rr = rr0;
                if (k0 != 0u) { lr = Blr[i]; rr = Brr[i]; }
                const uint32_t ni = i - (before + k);
                const bool next2 = i + 2u < n && tk_bit(L.sel[1 + ((i + 2u) >> 6)], (int)((i + 2u) & 63u));
                if (!next2) { Ark[ni] = rr; mnew = rr < mnew ? rr : mnew; }
                if (i > 0u) { Ark[ni - 1u] = lr; mnew = lr < mnew ? lr : mnew; }
            }
        }
        mnew = tkl_wave_min(mnew);
        if (lane == 0) L.wmin[wv] = mnew;
        n -= total;
        wv_block_sync();   // the new arrays and minima are in place
        if (total == 0u) break;                                    // (cannot happen: the leftmost candidate always merges)
    }

    for (uint32_t i = wv_tid(); i < n; i += TKL_THREADS) out[i] = Atok[i] + t.num_special;
    wv_block_sync();   // the scratch is wave 0's again (the single-wave merge of the next ordinary piece uses it)
    return n;
}

// ------------------------------------------------------------------------------------------------------------------
// The LAZY form of the rounds, for long pieces with many distinct pairs (tools/batched_merge_model.py rounds_heads).
//
// On such a piece a rank has a dozen occurrences among 30 000 parts, and a round that compacts touches all of them: 25 us
// of instruction issue on one CU, more than the dozen merges cost one at a time.  Here nothing moves: part i keeps slot i,
// an alive bit says whether a part starts there, the pair ranks of all 32 K slots live in LDS (128 KB), and a round works
// only where the minimum rank occurs:
//   * every step of 64 slots keeps its minimum rank; r* = the minimum of the step minima; the steps whose minimum is r*
//     are the only ones looked at;
//   * of a run of consecutive candidates only the HEAD merges (no parity bookkeeping: neighbours are found by bit scans
//     over the alive words); exactness needs, besides the cut behind the first occurrence that creates a pair below r*, the
//     CHAIN cut: nothing to the right of the leftmost candidate whose two predecessors are candidates too (the sequential
//     order would merge that one in this "round" as well; it waits for the next);
//   * the heads become a per-wave list and are probed one lane per occurrence; the kept ones rewrite their slot, clear
//     the alive bit of the part they swallow and note which steps changed; those steps' minima are recomputed.
// Four workgroup barriers and two or three global round trips per round, whatever the length of the piece.
// ------------------------------------------------------------------------------------------------------------------
#define TKS_CAP 512                          /* occurrences a wave lists per round (more: the round is cut there) */
#define TKS_STEPS (TK_LONG_MAX / 64)         /* 512 */
#define TKS_WSTEPS (TKS_STEPS / TKL_WAVES)   /* steps owned by a wave: 32 */

struct TksShared {
    uint32_t rk[TK_LONG_MAX];                // rank of the pair (part at slot i, next alive part); MAX: none / dead slot
    uint64_t alive[TKS_STEPS];
    uint64_t hsel[TKS_STEPS];                // heads listed this round
    uint32_t smin[TKS_STEPS];
    uint32_t touched[TKL_WAVES];             // bit s of word w: step 32 w + s changed this round
    uint16_t occ[TKL_WAVES][TKS_CAP];        // per wave: slots of its heads relative to its range, ascending
    uint32_t wz[TKL_WAVES];                  // per wave: leftmost position nothing at or right of which may merge this round
    uint32_t wunder[TKL_WAVES];
    uint32_t wpop[TKL_WAVES];                // (output) alive parts per wave
    uint32_t doc;
};

TK_DEV int tks_next_alive(const uint64_t* alive, uint32_t G, uint32_t i) {
    uint32_t g = i >> 6;
    const uint32_t b = i & 63u;
    uint64_t w = b == 63u ? 0ull : (alive[g] & (~0ull << (b + 1u)));
    while (!w) {
        if (++g >= G) return -1;
        w = alive[g];
    }
    return (int)(64u * g + (uint32_t)__builtin_ctzll(w));
}
TK_DEV int tks_prev_alive(const uint64_t* alive, uint32_t i) {
    int g = (int)(i >> 6);
    const uint32_t b = i & 63u;
    uint64_t w = alive[g] & ((1ull << b) - 1ull);
    while (!w) {
        if (--g < 0) return -1;
        w = alive[g];
    }
    return 64 * g + 63 - __builtin_clzll(w);
}

TK_DEV uint32_t tks_block_merge(const TkTablesView& t, const uint8_t* bytes, uint32_t n0, uint32_t* scratch, uint32_t* out,
                                                    TksShared& L) {
    const int lane = wv_lane();
    const uint32_t wv = wv_first(wv_tid() >> 6);
    uint32_t* tok = scratch;                                       // [n0] token of the part at slot i
    uint32_t* Blr = scratch + n0;                                  // results of the occurrences beyond a wave's first 64
    uint32_t* Brr = scratch + 2 * (size_t)n0;
    const uint32_t G = (n0 + 63u) / 64u;                           // steps in use

    // ---- start: one part per byte, pair ranks from PAIR2; step minima ----
    for (uint32_t i = wv_tid(); i < TKS_STEPS * 64u; i += TKL_THREADS) {
        uint32_t r = TK_RANK_MAX;
        if (i < n0) {
            const uint32_t b0 = bytes[i];
            tok[i] = b0;
            if (i + 1u < n0) r = t.pair2[b0 | ((uint32_t)bytes[i + 1u] << 8)];
        }
        L.rk[i] = r;
    }
    for (uint32_t g = wv_tid(); g < TKS_STEPS; g += TKL_THREADS) {
        const uint32_t lo = 64u * g;
        L.alive[g] = lo >= n0 ? 0ull : (n0 - lo >= 64u ? ~0ull : ((1ull << (n0 - lo)) - 1ull));
        L.hsel[g] = 0ull;
    }
    if (wv_tid() < TKL_WAVES) L.touched[wv_tid()] = 0u;
    wv_block_sync();
    for (uint32_t s = 0; s < TKS_WSTEPS; ++s) {
        const uint32_t g = wv * TKS_WSTEPS + s;
        const uint32_t m = tkl_wave_min(L.rk[64u * g + (uint32_t)lane]);
        if (lane == 0) L.smin[g] = m;
    }
    wv_block_sync();

    for (uint32_t round = 0; round <= n0; ++round) {               // (every round merges at least one pair)
        // ---- a: r* ----
        uint32_t rstar = TK_RANK_MAX;
#pragma unroll
        for (int q = 0; q < TKS_STEPS / 64; ++q) {
            const uint32_t v = L.smin[64 * q + lane];
            rstar = v < rstar ? v : rstar;
        }
        rstar = tkl_wave_min(rstar);
        if (rstar == TK_RANK_MAX) break;                           // block-uniform

        // ---- b: the heads of the wave's steps, the chain cut ----
        uint16_t* occ = L.occ[wv];
        uint32_t cntw = 0, zw = 0xFFFFFFFFu;
        {
            const uint32_t gq = wv * TKS_WSTEPS + (uint32_t)lane;
            uint64_t act = wv_ballot(lane < TKS_WSTEPS && L.smin[lane < TKS_WSTEPS ? gq : 0u] == rstar);
            while (act) {                                          // wave-uniform
                const uint32_t s = (uint32_t)tk_ctz64(act);
                act &= act - 1ull;
                const uint32_t g = wv * TKS_WSTEPS + s;
                const uint32_t i = 64u * g + (uint32_t)lane;
                const bool c = L.rk[i] == rstar;                   // (a dead slot holds MAX)
                bool head = false;
                uint32_t zl = 0xFFFFFFFFu;
                if (c) {
                    const int p = tks_prev_alive(L.alive, i);
                    const bool pc = p >= 0 && L.rk[p] == rstar;
                    head = !pc;
                    if (pc) {
                        const int pp = tks_prev_alive(L.alive, (uint32_t)p);
                        if (pp >= 0 && L.rk[pp] == rstar) zl = i;  // third of a run: nothing from here on merges this round
                    }
                }
                zl = tkl_wave_min(zl);
                zw = zl < zw ? zl : zw;
                uint64_t H = wv_ballot(head);
                if (cntw + (uint32_t)tk_popc64(H) > TKS_CAP) {     // the list is full: cut the round at the first head left out
                    uint32_t keep = TKS_CAP - cntw;
                    uint64_t K = 0ull, R = H;
                    while (keep--) { K |= R & (~R + 1ull); R &= R - 1ull; }
                    const uint32_t first_out = 64u * g + (uint32_t)tk_ctz64(R);
                    zw = first_out < zw ? first_out : zw;
                    H = K;
                    act = 0ull;
                }
                if (tk_bit(H, lane)) occ[cntw + (uint32_t)tk_popc64(H & tk_lowmask(lane))] = (uint16_t)(64u * s + (uint32_t)lane);
                if (lane == 0) L.hsel[g] = H;
                cntw += (uint32_t)tk_popc64(H);
            }
            if (lane == 0) L.wz[wv] = zw;
        }
        wv_block_sync();
        uint32_t zcut = lane < TKL_WAVES ? L.wz[lane] : 0xFFFFFFFFu;
        zcut = tkl_wave_min(zcut);
        const uint32_t wbase = wv * TKS_WSTEPS * 64u;
        if (zcut != 0xFFFFFFFFu) {                                 // the lists are ascending: a prefix stays
            uint32_t kept = 0;
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
                const uint32_t k = k0 + (uint32_t)lane;
                kept += (uint32_t)tk_popc64(wv_ballot(k < cntw && wbase + (uint32_t)occ[k < cntw ? k : 0u] < zcut));
            }
            cntw = kept;
        }

        // ---- c: one lane per occurrence: neighbours by bit scans, the two created pairs probed ----
        uint32_t umin = 0xFFFFFFFFu;
        uint32_t lr0 = TK_RANK_MAX, rr0 = TK_RANK_MAX;
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < cntw) {
                const uint32_t i = wbase + (uint32_t)occ[k];
                const int j = tks_next_alive(L.alive, G, i);       // exists: rk[i] is a rank
                const int k2 = tks_next_alive(L.alive, G, (uint32_t)j);
                const int p = tks_prev_alive(L.alive, i);
                uint32_t tl = 0, tr = 0;
                if (p >= 0) {
                    const int pp = tks_prev_alive(L.alive, (uint32_t)p);
                    // the part before me is swallowed this round if the occurrence before it is a listed head left of the cut
                    const bool lm = pp >= 0 && tk_bit(L.hsel[pp >> 6], pp & 63) && (uint32_t)pp < zcut;
                    tl = lm ? rstar : tok[p];
                }
                if (k2 >= 0) tr = tok[k2];
                uint32_t lr = TK_RANK_MAX, rr = TK_RANK_MAX;
                tk_probe_pair_x2(t, p >= 0 ? tl : 0u, rstar, rstar, k2 >= 0 ? tr : 0u, lr, rr);
                if (p < 0) lr = TK_RANK_MAX;
                if (k2 < 0) rr = TK_RANK_MAX;
                if (k0 == 0u) { lr0 = lr; rr0 = rr; }
                else { Blr[i] = lr; Brr[i] = rr; }
                if (lr < rstar || rr < rstar) umin = i < umin ? i : umin;
            }
        }
        umin = tkl_wave_min(umin);
        if (lane == 0) L.wunder[wv] = umin;
        wv_block_sync();
        uint32_t ustar = lane < TKL_WAVES ? L.wunder[lane] : 0xFFFFFFFFu;
        ustar = tkl_wave_min(ustar);

        // ---- d: the kept occurrences rewrite their slots (every rank slot has one writer: see tkl_block_merge R5) ----
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < cntw) {
                const uint32_t i = wbase + (uint32_t)occ[k];
                if (i <= ustar) {
                    uint32_t lr = lr0, rr = rr0;
                    if (k0 != 0u) { lr = Blr[i]; rr = Brr[i]; }
                    const int j = tks_next_alive(L.alive, G, i);
                    const int k2 = tks_next_alive(L.alive, G, (uint32_t)j);
                    const int p = tks_prev_alive(L.alive, i);
                    int left = p;
                    if (p >= 0) {
                        const int pp = tks_prev_alive(L.alive, (uint32_t)p);
                        if (pp >= 0 && tk_bit(L.hsel[pp >> 6], pp & 63) && (uint32_t)pp < zcut) left = pp;   // (pp < i <= ustar)
                    }
                    const bool next_merges = k2 >= 0 && tk_bit(L.hsel[k2 >> 6], k2 & 63) && (uint32_t)k2 < zcut && (uint32_t)k2 <= ustar;
                    // (the alive words are read above by every lane of the batch before any lane clears a bit below:
                    //  the batch's reads and writes are separated by the wave's lockstep, the batches by wv_lds_sync)
                    tok[i] = rstar;
                    if (!next_merges) L.rk[i] = rr;
                    if (left >= 0) L.rk[left] = lr;
                    L.rk[j] = TK_RANK_MAX;
                    // which steps changed
                    wv_lds_or(&L.touched[(i >> 6) / TKS_WSTEPS], 1u << ((i >> 6) % TKS_WSTEPS));
                    wv_lds_or(&L.touched[((uint32_t)j >> 6) / TKS_WSTEPS], 1u << (((uint32_t)j >> 6) % TKS_WSTEPS));
                    if (left >= 0) wv_lds_or(&L.touched[((uint32_t)left >> 6) / TKS_WSTEPS], 1u << (((uint32_t)left >> 6) % TKS_WSTEPS));
                    // the swallowed part: its alive bit goes LAST, after every lane of the round has looked its neighbours up
                    Blr[i] = (uint32_t)j;                                      // (remembered for phase e)
                }
            }
        }
        wv_block_sync();   // all neighbour look-ups of the round are done
        // ---- e: clear the alive bits of the swallowed parts, recompute the minima of the steps that changed ----
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < cntw; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < cntw) {
                const uint32_t i = wbase + (uint32_t)occ[k];
                if (i <= ustar) {
                    const uint32_t j = Blr[i];
                    wv_lds_and64(&L.alive[j >> 6], ~(1ull << (j & 63u)));
                }
            }
        }
        {
            const uint32_t g = wv * TKS_WSTEPS + (uint32_t)lane;
            if (lane < TKS_WSTEPS) L.hsel[g] = 0ull;
        }
        wv_block_sync();
        {
            uint32_t T = L.touched[wv];
            while (T) {                                                // wave-uniform
                const uint32_t s = (uint32_t)__builtin_ctz(T);
                T &= T - 1u;
                const uint32_t g = wv * TKS_WSTEPS + s;
                const uint32_t m = tkl_wave_min(L.rk[64u * g + (uint32_t)lane]);
                if (lane == 0) L.smin[g] = m;
            }
            if (lane == 0) L.touched[wv] = 0u;
        }
        wv_block_sync();
    }

    // ---- the alive parts in order -> out ----
    {
        uint32_t cnt = 0;
        for (uint32_t s = 0; s < TKS_WSTEPS; ++s) cnt += (uint32_t)tk_popc64(L.alive[wv * TKS_WSTEPS + s]);
        if (lane == 0) L.wpop[wv] = cnt;
    }
    wv_block_sync();
    uint32_t at = 0, total = 0;
    for (uint32_t v = 0; v < TKL_WAVES; ++v) {
        const uint32_t c = L.wpop[v];
        if (v < wv) at += c;
        total += c;
    }
    for (uint32_t s = 0; s < TKS_WSTEPS; ++s) {
        const uint32_t g = wv * TKS_WSTEPS + s;
        const uint64_t A = L.alive[g];
        if (tk_bit(A, lane)) out[at + (uint32_t)tk_popc64(A & tk_lowmask(lane))] = tok[64u * g + (uint32_t)lane] + t.num_special;
        at += (uint32_t)tk_popc64(A);
    }
    wv_block_sync();   // the scratch and the LDS are the next job's
    return total;
}

#endif
