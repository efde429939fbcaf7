// emu_driver.cpp -- runs the device algorithm (tk_encode_impl.h) on the CPU wave emulator.
// TEST INFRASTRUCTURE ONLY; see tk_wave_emu.h.  Built into tests/emu/libtk_emu.so by
// tests/emu/Makefile and loaded with ctypes from tests/test_kernel_emu.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "tk_wave_emu.h"
#include "../../include/tekken_hip.h"
#include "../../tekken-rs_amd/csrc/tk_encode_impl.h"
#include "../../tekken-rs_amd/csrc/tk_flat_impl.h"
#include "../../tekken-rs_amd/csrc/tk_long_impl.h"

namespace tkemu {
Wave* g_wave = nullptr;
Block* g_block = nullptr;
}

static std::string g_err;

extern "C" const char* emu_last_error() { return g_err.c_str(); }

// Full pipeline on the emulator: pass 1, pass 2 (deferred documents, with scratch), host scan
// and compaction.  out_ids must hold n_bytes + 2*n_docs entries.
extern "C" int emu_encode_batch(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special,
                                uint32_t bos, uint32_t eos, const uint8_t* bytes, const uint64_t* doc_offs,
                                uint64_t n_docs, int add_bos, int add_eos, int split_only, uint32_t* out_ids,
                                uint64_t* out_offs, uint8_t* dbg_starts, uint64_t* n_deferred, uint64_t* n_ops, int pattern) {
    TkHostTables T;
    int rc = tk_build_tables(blob, offs, n_ranks, num_special, bos, eos, T, g_err);
    if (rc != TK_OK) return rc;
    const uint64_t n_bytes = doc_offs[n_docs];
    std::vector<uint32_t> staging(n_bytes + 2 * n_docs + 1, 0xDEADBEEFu);
    std::vector<uint32_t> counts(n_docs + 1, 0);
    std::vector<uint32_t> defer_list(n_docs + 1, 0);
    uint32_t work_counter = 0, defer_count = 0;
    uint64_t maxlen = 0;
    for (uint64_t d = 0; d < n_docs; ++d) maxlen = std::max<uint64_t>(maxlen, doc_offs[d + 1] - doc_offs[d]);

    TkEncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = bytes;
    a.doc_offs = doc_offs;
    a.n_docs = n_docs;
    a.staging = staging.data();
    a.counts = counts.data();
    a.work_counter = &work_counter;
    a.defer_list = defer_list.data();
    a.defer_count = &defer_count;
    a.dbg_starts = dbg_starts;
    a.add_bos = add_bos;
    a.add_eos = add_eos;
    a.split_only = split_only;
    a.t = T.host_view();

    uint64_t ops = 0;
    a.pattern = pattern;
    if (pattern) {
        // row f-3 (opt-in): every document takes the piece-by-piece path of pass 2 with the JSON pattern's matcher
        for (uint64_t d = 0; d < n_docs; ++d) defer_list[d] = (uint32_t)d;
        defer_count = (uint32_t)n_docs;
    } else {
        if (split_only) tkemu::run_wave([&](int lane) { tk_encode_wave<2>(a, lane, 0); });
        else tkemu::run_wave([&](int lane) { tk_encode_wave<0>(a, lane, 0); });
        ops += tkemu::g_wave->n_ops;
    }
    if (n_deferred) *n_deferred = defer_count;
    if (defer_count) {
        std::vector<uint32_t> scratch_raw(5 * maxlen + 2 * ((maxlen + 63) / 64) + 64 + 8, 0);
        uint32_t* scratch_al = scratch_raw.data();
        while (reinterpret_cast<uintptr_t>(scratch_al) % 16) ++scratch_al;
        a.todo_list = defer_list.data();
        a.n_todo = defer_count;
        a.scratch = scratch_al;
        a.scratch_words_per_wave = scratch_raw.size() - 8;
        work_counter = 0;
        uint32_t dc2 = 0;
        a.defer_count = &dc2;
        tkemu::run_wave([&](int lane) { tk_encode_wave<1>(a, lane, 0); });
        if (dc2 != 0) { g_err = "pass 2 deferred a document"; return TK_ERR_RUNTIME; }
    }
    if (n_ops) *n_ops = tkemu::g_wave->n_ops;
    uint64_t t = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        out_offs[d] = t;
        if (!split_only) {
            memcpy(out_ids + t, staging.data() + doc_offs[d] + 2 * d, sizeof(uint32_t) * counts[d]);
            t += counts[d];
        }
    }
    out_offs[n_docs] = t;
    return TK_OK;
}

// MEMO (tk_hash.h) on the emulator: a table the test owns, kept across emu_flat_encode_batch calls like the context's across
// tk_encode_batch calls (log2 = 0: off).  emu_memo_info: {calls that used it, hits of the last call, valid entries, records the last
// call logged, records of the last call that won their slot}.
static tk_memo_entry* g_memo = nullptr;
static uint32_t g_memo_mask = 0, g_memo_epoch = 0, g_memo_hits = 0, g_memo_log_cap = 0, g_memo_logged = 0, g_memo_won = 0;
extern "C" void emu_memo_set(uint32_t* table_words, uint32_t log2) {
    g_memo = log2 ? reinterpret_cast<tk_memo_entry*>(table_words) : nullptr;
    g_memo_mask = log2 ? (1u << log2) - 1u : 0u;
    g_memo_log_cap = log2 ? (log2 > 4 ? 1u << (log2 - 2) : 3u) : 0u;   // (a tiny log as well: dropped entries are part of the design)
    g_memo_epoch = 0; g_memo_hits = 0;
}
extern "C" void emu_memo_info(uint64_t* out) {
    out[0] = g_memo_epoch; out[1] = g_memo_hits; out[2] = 0; out[3] = g_memo_logged; out[4] = g_memo_won;
    if (g_memo) for (uint32_t i = 0; i <= g_memo_mask; ++i) out[2] += tk_memo_len(g_memo[i].v[2]) != 0u;
}

// the memo entry's packing (tk_hash.h): ranks in, ranks out; returns 0 if every field comes back
extern "C" int emu_memo_pack_roundtrip(const uint32_t* ranks5, uint32_t n, uint32_t len) {
    uint32_t w4 = 0, v[3] = {0, 0, 0};
    tk_memo_pack(ranks5, n, len, &w4, v);
    if (!(w4 & TK_MEMO_TAG) || tk_memo_n(v[2]) != n || tk_memo_len(v[2]) != len) return 1;
    for (uint32_t i = 0; i < 5; ++i)
        if (tk_memo_id(w4, v[0], v[1], v[2], i) != ranks5[i]) return 2 + (int)i;
    return 0;
}

// Flat path on the emulator: tk_flat_chunk for every chunk (one emulated wave), the flagged documents through
// the per-document algorithm (mode 3, then pass 2), and host restatements of the small bookkeeping kernels of
// tk_flat.hip (first_doc, todo list, chunk prefix sums, counts, assemble).
#ifndef TKF_LONG_SCRATCH_WORDS
#define TKF_LONG_SCRATCH_WORDS 2048u
#endif
extern "C" int emu_flat_encode_batch(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special,
                                     uint32_t bos, uint32_t eos, const uint8_t* bytes, const uint64_t* doc_offs,
                                     uint64_t n_docs, int add_bos, int add_eos, uint32_t* out_ids, uint64_t* out_offs,
                                     uint8_t* dbg_starts, uint8_t* out_flags, uint64_t* n_flagged, uint64_t* n_ops, int pattern) {
    TkHostTables T;
    int rc = tk_build_tables(blob, offs, n_ranks, num_special, bos, eos, T, g_err);
    if (rc != TK_OK) return rc;
    const uint64_t n_bytes = doc_offs[n_docs];
    const uint64_t n_chunks = (n_bytes + TKF_COMMIT - 1) / TKF_COMMIT;
    std::vector<uint32_t> first_doc(n_chunks + 1, 0), tmp(n_chunks * TKF_STRIDE + 1, 0xDEADBEEFu), kcount(n_chunks + 1, 0);
    std::vector<uint32_t> lstart(n_docs + 1, 0xDEADBEEFu), flags(n_docs + 1, 0), holes(n_docs + 1, 0);
    std::vector<uint32_t> miss_list(n_chunks * TKF_MISSCAP + 16, 0), miss_count(4 * n_chunks + 1, 0);
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const int64_t lo = (int64_t)c * TKF_COMMIT - TKF_HL;
        uint32_t k = 0;
        while (k < n_docs && (int64_t)doc_offs[k] < (lo > 0 ? lo : 0)) ++k;
        first_doc[c] = k;
    }
    TkFlatArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.bytes = bytes;
    fa.doc_offs = doc_offs;
    fa.n_docs = n_docs;
    fa.n_bytes = n_bytes;
    fa.n_chunks = n_chunks;
    fa.first_doc = first_doc.data();
    fa.tmp = tmp.data();
    fa.kcount = kcount.data();
    fa.lstart = lstart.data();
    fa.flags = flags.data();
    fa.miss_list = miss_list.data();
    fa.miss_count = miss_count.data();
    fa.holes = holes.data();
    fa.dbg_starts = dbg_starts;
    fa.pattern = pattern;
    fa.t = T.host_view();
    if (g_memo) {
        if (reinterpret_cast<uintptr_t>(g_memo) % 32) { g_err = "memo table must be 32-byte aligned"; return TK_ERR_INVALID_ARG; }
        g_memo_hits = 0;
        fa.memo_tab = g_memo; fa.memo_mask = g_memo_mask; fa.memo_epoch = ++g_memo_epoch; fa.memo_hits = &g_memo_hits;
        fa.memo_probe = g_memo_epoch > 1 ? 1 : 0;
    }
    // the log: three "waves" (the narrow groups are dealt out over them), each with its own stretch
    const uint32_t memo_waves = 3;
    std::vector<tk_memo_entry> memo_log((size_t)g_memo_log_cap * memo_waves + 1);
    std::vector<uint32_t> memo_log_counts(memo_waves, 0);
    if (g_memo) { fa.memo_log = memo_log.data(); fa.memo_log_counts = memo_log_counts.data(); fa.memo_log_per_wave = g_memo_log_cap; fa.memo_log_waves = memo_waves; }
    // pieces of 65..TKF_LONGCAP bytes stay on the flat path as records (TK_FLAT_LONG=0: they hand their documents back)
    std::vector<TkFlatLongRec> long_recs(n_bytes / 65 + 16);
    std::vector<uint32_t> ctlblk(24, 0);                    // the context's counter block: counter 11, control words 16..18
    uint32_t& long_count = ctlblk[11];
    if (!(getenv("TK_FLAT_LONG") && atoi(getenv("TK_FLAT_LONG")) == 0)) {
        fa.long_recs = long_recs.data();
        fa.long_count = &ctlblk[11];
        fa.long_cap = (uint32_t)long_recs.size();
        const uint64_t pv = (uint64_t)reinterpret_cast<uintptr_t>(long_recs.data());
        ctlblk[16] = (uint32_t)pv; ctlblk[17] = (uint32_t)(pv >> 32); ctlblk[18] = fa.long_cap;
        fa.long_ctl = &ctlblk[16];
    }
    // chunks with a piece of more than 64 bytes are left to the CUT instantiation (counter 12, control words 19..20; TK_FLAT_CUT=0: no cuts)
    std::vector<uint32_t> cut_list(n_chunks + 1, 0);
    if (fa.long_ctl && !pattern && !(getenv("TK_FLAT_CUT") && atoi(getenv("TK_FLAT_CUT")) == 0)) {
        const uint64_t pv = (uint64_t)reinterpret_cast<uintptr_t>(cut_list.data());
        ctlblk[19] = (uint32_t)pv; ctlblk[20] = (uint32_t)(pv >> 32);
        fa.cut_list = cut_list.data();
        fa.cut_count = &ctlblk[12];
    }
    std::vector<uint32_t> lds(TKF_LDS_WORDS_CUT, 0);
    uint64_t ops = 0;
    if (n_chunks) {
        tkemu::run_wave([&](int lane) {
            tk_flat_init_lds(fa, lds.data(), lane);
            for (uint64_t c = 0; c < n_chunks; ++c) {
                const bool m1 = fa.t.key_hash_mode != 0u;
                if (pattern) {   // row f-3: the JSON pattern's rules (the debug flags are compiled in: split checks use them)
                    if (m1) tk_flat_chunk<1, 1, 1>(fa, c, lane, lds.data()); else tk_flat_chunk<1, 0, 1>(fa, c, lane, lds.data());
                } else if (fa.dbg_starts) { if (m1) tk_flat_chunk<1, 1>(fa, c, lane, lds.data()); else tk_flat_chunk<1, 0>(fa, c, lane, lds.data()); }
                else { if (m1) tk_flat_chunk<0, 1>(fa, c, lane, lds.data()); else tk_flat_chunk<0, 0>(fa, c, lane, lds.data()); }   // production
            }
            tk_flat_flush_memo_hits(fa, lds.data(), lane);
        });
        ops += tkemu::g_wave->n_ops;
        if (fa.cut_list && ctlblk[12]) {                    // tk_flat_cut_kernel
            if (ctlblk[12] > n_chunks) { g_err = "cut list overflow"; return TK_ERR_RUNTIME; }
            tkemu::run_wave([&](int lane) {
                tk_flat_init_lds(fa, lds.data(), lane);
                for (uint32_t i = 0; i < ctlblk[12]; ++i) {
                    const uint64_t c = cut_list[i];
                    const bool m1 = fa.t.key_hash_mode != 0u;
                    if (fa.dbg_starts) { if (m1) tk_flat_chunk<1, 1, 0, 1>(fa, c, lane, lds.data()); else tk_flat_chunk<1, 0, 0, 1>(fa, c, lane, lds.data()); }
                    else { if (m1) tk_flat_chunk<0, 1, 0, 1>(fa, c, lane, lds.data()); else tk_flat_chunk<0, 0, 0, 1>(fa, c, lane, lds.data()); }
                }
                });
            ops += tkemu::g_wave->n_ops;
        }
        if (getenv("TK_EMU_LOG")) fprintf(stderr, "[emu] chunks %llu, cut chunks %u, long records %u\n", (unsigned long long)n_chunks, ctlblk[12], ctlblk[11]);
        std::vector<uint64_t> mpfx(4 * n_chunks + 1, 0);
        for (uint64_t e = 0; e < 4 * n_chunks; ++e) mpfx[e + 1] = mpfx[e] + miss_count[e];
        fa.miss_prefix = mpfx.data();
        const uint64_t n_narrow = mpfx[2 * n_chunks], n_wide = mpfx[4 * n_chunks] - n_narrow;
        std::vector<uint32_t> wave_first(n_narrow / 64 + 2, 0);
        for (uint64_t e = 0; e < 2 * n_chunks; ++e)
            for (uint64_t w = (mpfx[e] + 63) / 64; w * 64 < mpfx[e + 1]; ++w) wave_first[w] = (uint32_t)e;
        fa.wave_first = wave_first.data();
        std::vector<uint32_t> wave_first_wide(n_wide / 64 + 2, 0);
        for (uint64_t e = 2 * n_chunks; e < 4 * n_chunks; ++e)
            for (uint64_t w = (mpfx[e] - n_narrow + 63) / 64; w * 64 < mpfx[e + 1] - n_narrow; ++w) wave_first_wide[w] = (uint32_t)e;
        fa.wave_first_wide = wave_first_wide.data();
        // the wave's LDS columns sit between two guard zones: a write outside the kernel's share (8 KB narrow, 16 KB
        // wide) would be silent on the device
        const size_t G = 256;
        std::vector<uint32_t> mlds(G + TKM_LDS_WORDS(32) + G, 0xDEADBEEFu);
        auto guards_ok = [&](size_t used) {
            for (size_t i = 0; i < G; ++i)
                if (mlds[i] != 0xDEADBEEFu || mlds[G + used + i] != 0xDEADBEEFu) return false;
            return true;
        };
        for (uint64_t w = 0; w * 64 < n_narrow; ++w) {
            // (on the device the log position is a register of every lane, wave-uniform: one copy per emulated lane)
            TkMemoLog mls[64];
            const uint32_t mw = (uint32_t)(w % memo_waves);
            for (int l = 0; l < 64; ++l) {
                mls[l].base = fa.memo_log ? fa.memo_log + (size_t)mw * fa.memo_log_per_wave : nullptr;
                mls[l].n = fa.memo_log ? memo_log_counts[mw] : 0u;
                mls[l].cap = fa.memo_log ? fa.memo_log_per_wave : 0u;
            }
            tkemu::run_wave([&](int lane) { tk_merge_wave<false>(fa, w, lane, mlds.data() + G, fa.t.pair_filter, fa.memo_log ? &mls[lane] : nullptr); });
            if (fa.memo_log) {
                for (int l = 1; l < 64; ++l)
                    if (mls[l].n != mls[0].n) { g_err = "memo log position is not wave-uniform"; return TK_ERR_RUNTIME; }
                memo_log_counts[mw] = mls[0].n;
            }
            if (!guards_ok(TKM_LDS_WORDS(16))) { g_err = "tk_merge_wave<false> wrote outside its LDS columns"; return TK_ERR_RUNTIME; }
            ops += tkemu::g_wave->n_ops;
        }
        const uint64_t n_wide2 = mpfx[3 * n_chunks] - mpfx[2 * n_chunks], n_wide3 = mpfx[4 * n_chunks] - mpfx[3 * n_chunks];
        for (uint64_t w = 0; w * 64 < n_wide2; ++w) {
            std::fill(mlds.begin(), mlds.end(), 0xDEADBEEFu);
            tkemu::run_wave([&](int lane) { tk_merge_wave<true>(fa, w, lane, mlds.data() + G, fa.t.pair_filter); });
            if (!guards_ok(TKM_LDS_WORDS(32))) { g_err = "tk_merge_wave<true> wrote outside its LDS columns"; return TK_ERR_RUNTIME; }
            ops += tkemu::g_wave->n_ops;
        }
        std::vector<uint32_t> mlds64(G + TKM_LDS_WORDS(64) + G, 0xDEADBEEFu);
        for (uint64_t w = 0; w * 64 < n_wide3; ++w) {
            std::fill(mlds64.begin(), mlds64.end(), 0xDEADBEEFu);
            tkemu::run_wave([&](int lane) { tk_merge_wave_long3(fa, w, lane, mlds64.data() + G, fa.t.pair_filter); });
            for (size_t i = 0; i < G; ++i)
                if (mlds64[i] != 0xDEADBEEFu || mlds64[G + TKM_LDS_WORDS(64) + i] != 0xDEADBEEFu) { g_err = "tk_merge_wave_long3 wrote outside its LDS columns"; return TK_ERR_RUNTIME; }
            ops += tkemu::g_wave->n_ops;
        }
    }
    if (fa.memo_tab) {                                      // tk_memo_commit_kernel
        const uint32_t nl = fa.memo_log_per_wave * fa.memo_log_waves;
        g_memo_logged = g_memo_won = 0;
        for (uint32_t i = 0; i < nl; ++i) {
            if (!tk_memo_log_live(memo_log_counts.data(), fa.memo_log_per_wave, i)) continue;
            ++g_memo_logged;
            g_memo_won += fa.memo_tab[tk_memo_slot_of(memo_log[i], fa.t.key_hash_mode, fa.memo_mask)].w4 == i;
            if (!(memo_log[i].w4 & TK_MEMO_TAG) || tk_memo_n(memo_log[i].v[2]) == 0u || tk_memo_n(memo_log[i].v[2]) > TK_MEMO_MAXIDS) { g_err = "memo log: malformed record " + std::to_string(i) + " w4=" + std::to_string(memo_log[i].w4) + " v2=" + std::to_string(memo_log[i].v[2]) + " counts=" + std::to_string(memo_log_counts[i / fa.memo_log_per_wave]); return TK_ERR_RUNTIME; }
        }
        for (uint32_t i = 0; i < nl; ++i)
            if (tk_memo_log_live(memo_log_counts.data(), fa.memo_log_per_wave, i)) tk_memo_commit_one(fa.memo_tab, memo_log.data(), i, fa.t.key_hash_mode, fa.memo_mask);
    }
    // the long-piece records (tk_flat_long_kernel): one wave each; a piece beyond TKF_LONGCAP flags its document; those of up
    // to 128 bytes that are no vocabulary keys go on to the lane-per-piece merge (tk_flat_long128_kernel)
    {
        std::vector<uint32_t> lscratch(TKF_LONG_SCRATCH_WORDS + 64, 0xDEADBEEFu);
        const uint32_t nl = long_count < fa.long_cap ? long_count : fa.long_cap;
        fa.long_merge128 = (getenv("TK_FLAT_LONG128") && atoi(getenv("TK_FLAT_LONG128")) == 0) ? 0 : 1;
        for (uint32_t q = 0; q < nl; ++q) {
            tkemu::run_wave([&](int lane) {
                const TkPolyPow pw = tk_poly_pow(fa.t, lane);
                tk_flat_long_wave(fa, pw, q, lane, lscratch.data());
            });
            ops += tkemu::g_wave->n_ops;
            for (size_t i = 0; i < 64; ++i)
                if (lscratch[TKF_LONG_SCRATCH_WORDS + i] != 0xDEADBEEFu) { g_err = "tk_flat_long_wave wrote past its scratch"; return TK_ERR_RUNTIME; }
        }
        const size_t G2 = 256;
        std::vector<uint32_t> mldsN(G2 + TKM_LDS_WORDS(128) + G2, 0xDEADBEEFu);
        for (uint64_t w = 0; fa.long_merge128 && w * 64 < nl; ++w) {
            std::fill(mldsN.begin(), mldsN.end(), 0xDEADBEEFu);
            tkemu::run_wave([&](int lane) { tk_merge_long_wave<128>(fa, w, lane, mldsN.data() + G2, fa.t.pair_filter); });
            ops += tkemu::g_wave->n_ops;
            for (size_t i = 0; i < G2; ++i)
                if (mldsN[i] != 0xDEADBEEFu || mldsN[G2 + TKM_LDS_WORDS(128) + i] != 0xDEADBEEFu) { g_err = "tk_merge_long_wave wrote outside its LDS columns"; return TK_ERR_RUNTIME; }
        }
        for (uint32_t q = 0; fa.long_merge128 && q < nl; ++q) {
            tkemu::run_wave([&](int lane) { tk_flat_long_coop_wave(fa, q, lane, lscratch.data()); });
            ops += tkemu::g_wave->n_ops;
            for (size_t i = 0; i < 64; ++i)
                if (lscratch[TKF_LONG_SCRATCH_WORDS + i] != 0xDEADBEEFu) { g_err = "tk_flat_long_coop_wave wrote past its scratch"; return TK_ERR_RUNTIME; }
        }
    }
    // flagged documents -> per-document algorithm
    std::vector<uint32_t> todo;
    for (uint64_t d = 0; d < n_docs; ++d)
        if (flags[d]) todo.push_back((uint32_t)d);
    if (n_flagged) *n_flagged = todo.size();
    if (out_flags) for (uint64_t d = 0; d < n_docs; ++d) out_flags[d] = (uint8_t)flags[d];
    std::vector<uint32_t> staging(n_bytes + 2 * n_docs + 1, 0xDEADBEEFu), counts(n_docs + 1, 0), defer_list(n_docs + 1, 0);
    if (!todo.empty()) {
        uint32_t work_counter = 0, defer_count = 0;
        TkEncodeArgs a;
        memset(&a, 0, sizeof(a));
        a.bytes = bytes;
        a.doc_offs = doc_offs;
        a.n_docs = n_docs;
        a.staging = staging.data();
        a.counts = counts.data();
        a.work_counter = &work_counter;
        a.defer_list = defer_list.data();
        a.defer_count = &defer_count;
        a.todo_list = todo.data();
        a.n_todo = (uint32_t)todo.size();
        a.add_bos = add_bos;
        a.add_eos = add_eos;
        a.t = T.host_view();
        a.pattern = pattern;
        if (pattern) {   // JSON pattern: the handed-back documents go straight to the piece-by-piece path
            for (size_t i = 0; i < todo.size(); ++i) defer_list[i] = todo[i];
            defer_count = (uint32_t)todo.size();
        } else {
            tkemu::run_wave([&](int lane) { tk_encode_wave<3>(a, lane, 0); });
            ops += tkemu::g_wave->n_ops;
        }
        if (defer_count) {
            uint64_t maxlen = 0;
            for (uint32_t i = 0; i < defer_count; ++i)
                maxlen = std::max<uint64_t>(maxlen, doc_offs[defer_list[i] + 1] - doc_offs[defer_list[i]]);
            std::vector<uint32_t> scratch_raw(5 * maxlen + 2 * ((maxlen + 63) / 64) + 64 + 8, 0);
            uint32_t* scratch_al = scratch_raw.data();
            while (reinterpret_cast<uintptr_t>(scratch_al) % 16) ++scratch_al;
            std::vector<uint32_t> todo2(defer_list.begin(), defer_list.begin() + defer_count);
            a.todo_list = todo2.data();
            a.n_todo = defer_count;
            a.scratch = scratch_al;
            a.scratch_words_per_wave = scratch_raw.size() - 8;
            work_counter = 0;
            uint32_t dc2 = 0;
            a.defer_count = &dc2;
            tkemu::run_wave([&](int lane) { tk_encode_wave<1>(a, lane, 0); });
            ops += tkemu::g_wave->n_ops;
            if (dc2 != 0) { g_err = "pass 2 deferred a document"; return TK_ERR_RUNTIME; }
        }
    }
    if (n_ops) *n_ops = ops;
    // chunk prefix sums, counts, assemble (host restatement of tk_flat.hip's bookkeeping kernels)
    std::vector<uint64_t> P(n_chunks + 1, 0);
    for (uint64_t c = 0; c < n_chunks; ++c) P[c + 1] = P[c] + kcount[c];
    auto G = [&](uint64_t i) -> uint64_t {
        const uint64_t p = doc_offs[i];
        if (p >= n_bytes) return P[n_chunks];
        return P[p / TKF_COMMIT] + lstart[i];
    };
    uint64_t t = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        out_offs[d] = t;
        if (flags[d]) {
            memcpy(out_ids + t, staging.data() + doc_offs[d] + 2 * d, sizeof(uint32_t) * counts[d]);
            t += counts[d];
            continue;
        }
        const uint64_t g0 = G(d), g1 = G(d + 1);
        if (add_bos) out_ids[t++] = T.bos_id;
        uint64_t g = g0, c = doc_offs[d] / TKF_COMMIT;
        while (g < g1) {
            const uint64_t hi = g1 < P[c + 1] ? g1 : P[c + 1];
            for (; g < hi; ++g) {
                const uint32_t v = tmp[c * TKF_STRIDE + (g - P[c])];
                if (v != TKF_HOLE) out_ids[t++] = v;
            }
            ++c;
        }
        if (add_eos) out_ids[t++] = T.eos_id;
    }
    out_offs[n_docs] = t;
    return TK_OK;
}

// table facts for tests: out = {key_hash_mode, KEY8 slots, KEY16 slots, keys stored in their second slot, flagged slots, PAIR buckets}
extern "C" int emu_table_info(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, uint64_t* out) {
    TkHostTables T;
    int rc = tk_build_tables(blob, offs, n_ranks, num_special, 1, 2, T, g_err);
    if (rc != TK_OK) return rc;
    out[0] = T.key_hash_mode; out[1] = (uint64_t)T.key8_mask + 1; out[2] = (uint64_t)T.key_mask + 1;
    out[3] = T.n_key_second; out[4] = T.n_key_spill_slots; out[5] = (uint64_t)T.pair_mask + 1;
    // PAIR filter (tk_hash.h): every stored pair must have its bit set (a clear bit is taken as proof of absence);
    // out[6] = pairs, out[7] = set bits, out[8] = filter bits
    if (T.pair_filter.size() != TK_PAIRF_WORDS) { g_err = "pair filter size"; return TK_ERR_RUNTIME; }
    uint64_t n_pairs = 0, n_set = 0;
    for (uint64_t e : T.pair_tab) {
        if (e == TK_PAIR_EMPTY) continue;
        ++n_pairs;
        const uint64_t key = tk_pair_key(e);
        const uint32_t b = tk_pair_fbit(tk_pair_hash((uint32_t)(key >> TK_ID_BITS), (uint32_t)(key & ((1u << TK_ID_BITS) - 1u))));
        if (b >= (1u << TK_PAIRF_LOG2) || !((T.pair_filter[b >> 5] >> (b & 31u)) & 1u)) { g_err = "a stored pair is missing from the PAIR filter"; return TK_ERR_RUNTIME; }
    }
    for (uint32_t w : T.pair_filter) n_set += (uint64_t)__builtin_popcount(w);
    if (n_pairs != T.n_pairs || n_set > n_pairs) { g_err = "pair filter counts"; return TK_ERR_RUNTIME; }
    out[6] = n_pairs; out[7] = n_set; out[8] = 1ull << TK_PAIRF_LOG2;
    return TK_OK;
}

// table cache (row f-2): build, save, load, compare field by field; out = {build seconds, save seconds, load seconds, file bytes}
#include <algorithm>
#include <chrono>
extern "C" int emu_table_cache_roundtrip(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special,
                                         const char* path, double* out) {
    using clk = std::chrono::steady_clock;
    TkHostTables A, B;
    auto t0 = clk::now();
    int rc = tk_build_tables(blob, offs, n_ranks, num_special, 1, 2, A, g_err);
    if (rc != TK_OK) return rc;
    auto t1 = clk::now();
    const uint64_t key = tk_tables_key(blob, offs, n_ranks, num_special, 1, 2);
    if (!tk_tables_save(A, key, path)) { g_err = "save failed"; return TK_ERR_RUNTIME; }
    auto t2 = clk::now();
    if (tk_tables_load(B, key + 1, path)) { g_err = "a wrong key was accepted"; return TK_ERR_RUNTIME; }
    auto t3 = clk::now();
    if (!tk_tables_load(B, key, path)) { g_err = "load failed"; return TK_ERR_RUNTIME; }
    auto t4 = clk::now();
    auto same = [](const auto& x, const auto& y) { return x.size() == y.size() && (x.empty() || memcmp(x.data(), y.data(), x.size() * sizeof(x[0])) == 0); };
    if (!(same(A.blob, B.blob) && same(A.offs, B.offs) && same(A.uc_stage1, B.uc_stage1) && same(A.uc_stage2, B.uc_stage2) &&
          same(A.key8_tab, B.key8_tab) && same(A.key_tab, B.key_tab) && same(A.long_tab, B.long_tab) && same(A.pair_tab, B.pair_tab) &&
          same(A.pair2, B.pair2) && same(A.pair_filter, B.pair_filter) && A.key8_mask == B.key8_mask && A.key_mask == B.key_mask && A.long_mask == B.long_mask &&
          A.pair_mask == B.pair_mask && A.key_hash_mode == B.key_hash_mode && A.n_ranks == B.n_ranks && A.num_special == B.num_special &&
          A.bos_id == B.bos_id && A.eos_id == B.eos_id && A.p1inv == B.p1inv && A.p2inv == B.p2inv && A.n_pairs == B.n_pairs)) {
        g_err = "loaded tables differ from the built ones";
        return TK_ERR_RUNTIME;
    }
    out[0] = std::chrono::duration<double>(t1 - t0).count();
    out[1] = std::chrono::duration<double>(t2 - t1).count();
    out[2] = std::chrono::duration<double>(t4 - t3).count();
    FILE* f = fopen(path, "rb");
    out[3] = 0;
    if (f) { fseek(f, 0, SEEK_END); out[3] = (double)ftell(f); fclose(f); }
    return TK_OK;
}

extern "C" int emu_table_cache_load(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, const char* path) {
    TkHostTables B;
    return tk_tables_load(B, tk_tables_key(blob, offs, n_ranks, num_special, 1, 2), path) ? TK_OK : TK_ERR_RUNTIME;
}


// The round-based workgroup merges of ONE long piece (tk_long_impl.h) on 16 emulated waves: kind 0 = compacting rounds
// (tkl_block_merge), 1 = lazy rounds (tks_block_merge).  The scratch (4 n words) and the LDS image sit between guard words
// (and under ASan between red zones); out_ids must hold n entries; returns the number of ids or a negative code.
extern "C" int64_t emu_long_merge(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, int kind,
                                  const uint8_t* bytes, uint32_t n, uint32_t* out_ids, uint64_t* n_barriers) {
    if (n <= 64u || n > TK_LONG_MAX) { g_err = "piece length out of range for the workgroup merges"; return TK_ERR_INVALID_ARG; }
    TkHostTables T;
    int rc = tk_build_tables(blob, offs, n_ranks, num_special, 1, 2, T, g_err);
    if (rc != TK_OK) return rc;
    const TkTablesView t = T.host_view();
    const size_t G = 64;
    std::vector<uint32_t> scratch(G + 4 * (size_t)n + G, 0xDEADBEEFu), out(G + n + G, 0xDEADBEEFu);
    std::vector<uint8_t> text(bytes, bytes + n);            // (its own allocation: reads past the piece are caught)
    TksShared* LS = new TksShared();                         // heap: the sanitizer sees its bounds
    static_assert(sizeof(TklShared) <= sizeof(TksShared), "one LDS image serves both forms");
    std::vector<uint32_t> result(TKL_WAVES, 0);
    tkemu::run_block(TKL_WAVES, [&](int lane) {
        uint32_t r;
        if (kind == 0) r = tkl_block_merge(t, text.data(), n, scratch.data() + G, out.data() + G, *reinterpret_cast<TklShared*>(LS));
        else r = tks_block_merge(t, text.data(), n, scratch.data() + G, out.data() + G, *LS);
        if (lane == 0) result[wv_tid() >> 6] = r;
    });
    if (n_barriers) *n_barriers = 0;
    delete LS;
    for (size_t i = 0; i < G; ++i)
        if (scratch[i] != 0xDEADBEEFu || scratch[G + 4 * (size_t)n + i] != 0xDEADBEEFu || out[i] != 0xDEADBEEFu || out[G + n + i] != 0xDEADBEEFu) {
            g_err = "the workgroup merge wrote outside its scratch / output";
            return TK_ERR_RUNTIME;
        }
    for (int v = 1; v < TKL_WAVES; ++v)
        if (result[v] != result[0]) { g_err = "the waves disagree on the number of ids"; return TK_ERR_RUNTIME; }
    if (result[0] > n) { g_err = "more ids than bytes"; return TK_ERR_RUNTIME; }
    memcpy(out_ids, out.data() + G, sizeof(uint32_t) * result[0]);
    return (int64_t)result[0];
}
