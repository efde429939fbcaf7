// tk_capi.cpp -- engine-level C ABI (include/tekken_hip.h): context, device tables, the
// batch pipeline  encode(pass 1) -> scan -> compact [-> pass 2 -> scan -> compact].
//
// Replaces CoreBPE::new / CoreBPE::encode at reference src/tekkenizer.rs:125 and :384-386 and
// fuses the id shift / BOS / EOS of :390-402.  There is NO CPU fallback in this file: without
// a HIP device every entry point fails with TK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tekken_hip.h"
#include "tekkenizer.hpp"
#include "tk_engine.h"
#include "tk_kernels.h"
#include "tk_tables.h"

static thread_local std::string g_tls_err;

void tk_set_tls_error(const std::string& e) { g_tls_err = e; }
const std::string& tk_get_tls_error() { return g_tls_err; }

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};

// Pinned result buffers of tk_encode_batch / tk_decode_batch: a process-wide pool.  hipHostMalloc / hipHostFree cost
// hundreds of microseconds each (they map the pages into every device); a caller that encodes batch after batch gets the
// same blocks back from tk_free_result instead.  Bounded: at most 8 free blocks / 1 GiB are kept.
struct PinPool {
    std::mutex mu;
    std::unordered_map<void*, size_t> live;            // handed out
    std::vector<std::pair<void*, size_t>> free_blocks;
    size_t free_bytes = 0;
    void* get(size_t bytes) {
        if (bytes == 0) bytes = 1;
        {
            std::lock_guard<std::mutex> lock(mu);
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].second >= bytes && free_blocks[i].second <= 4 * bytes + (1u << 16) &&
                    (best == free_blocks.size() || free_blocks[i].second < free_blocks[best].second))
                    best = i;
            if (best != free_blocks.size()) {
                std::pair<void*, size_t> b = free_blocks[best];
                free_blocks.erase(free_blocks.begin() + (long)best);
                free_bytes -= b.second;
                live[b.first] = b.second;
                return b.first;
            }
        }
        const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 8;
        void* p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        live[p] = want;
        return p;
    }
    void put(void* p) {
        if (!p) return;
        size_t sz = 0;
        {
            std::lock_guard<std::mutex> lock(mu);
            auto it = live.find(p);
            if (it == live.end()) { sz = 0; }
            else {
                sz = it->second;
                live.erase(it);
                if (free_blocks.size() < 8 && free_bytes + sz <= (1ull << 30)) {
                    free_blocks.push_back({p, sz});
                    free_bytes += sz;
                    return;
                }
            }
        }
        (void)hipHostFree(p);
    }
};
PinPool g_pin_pool;

}  // namespace

void* tk_pinned_get(size_t bytes) { return g_pin_pool.get(bytes); }
void tk_pinned_put(void* p) { g_pin_pool.put(p); }

struct tk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    TkHostTables host;
    TkTablesView dview;
    DevBuf t_uc1, t_uc2, t_key8, t_key, t_long, t_pair, t_pair2, t_pairf, t_blob, t_offs, t_spblob, t_spoffs, t_uc2a, t_uc2b;
    DevBuf t_key64;                // whole pieces of 17..64 bytes by the flat kernel's dword hash
    DevBuf t_cutk2, t_cutg3, t_ucbmp;   // the cut rule's bit maps, the class trie flattened for the BMP (tk_tables.cpp make_cut_tables)
    DevBuf f_cut;                  // flat path: chunks left to the CUT instantiation (tk_flat_cut_kernel)
    void* cut_ctl_ptr = nullptr;   // what the control words at counters + 19 describe
    bool no_flat_cut = false;      // TK_FLAT_CUT=0: no cut decomposition (pieces of more than 256 bytes hand their documents back; A / B and tests)
    int pattern = 0;               // tk_ctx_set_pattern: 0 the reference's hard-coded pattern, 1 the JSON pattern (row f-3)
    bool have_specials = false;
    DevBuf dec_lens, dec_bytes, dec_offs, dec_bits, dec_err, dec_in_ids, dec_in_offs, dec_hi, dec_glens, dec_goffs;
    bool no_decode_groups = false;   // TK_DECODE_GROUPS=0: the per-document length pass (A / B and tests of the fall-back)
    uint32_t decode_group_limit = 0x7FFFFF00u;   // ids / text bytes of a group from which the call falls back (TK_DECODE_GROUP_LIMIT: tests)
    DevBuf t_inline, t_len8;   // decode: 16-byte inline entries and one-byte lengths by rank (built at the first decode call)
    DevBuf staging, counts, out_ids, out_offs, block_sums, defer_list, scratch, counters, in_bytes, in_offs, dbg;
    DevBuf f_long;             // flat path: records of the pieces of 65..TKF_LONGCAP bytes
    DevBuf long_jobs;              // tk_long.hip: the long pieces of the long-list documents
    DevBuf long_list;              // pass 2 -> tk_long.hip: documents with a long piece that is not a vocabulary key
    uint32_t long_lazy_mul = 0;    // TK_LONG_LAZY_MUL (0 = the default of tk_piece_is_long)
    uint32_t long_min = 256;       // shortest piece (bytes) merged in rounds by a workgroup (TK_LONG_MIN; 0 = never)
    uint32_t long_force = 0;       // TK_LONG_FORCE (tests): rounds for every long piece, not only the repetitive ones
    uint64_t n_round_docs = 0;     // documents the round-based kernel took in the last call
    DevBuf f_first, f_tmp, f_lstart, f_flags, f_todo, f_miss, f_mcnt, f_mpfx, f_wfirst, f_info;  // flat path (tk_flat.hip)
    bool use_flat = true;
    int pipeline_forced = 0;       // TK_PIPELINE: 0 / 1 flat (default), 2 per-document kernels only
    uint64_t n_flagged = 0;
    uint64_t n_long_recs = 0;      // pieces of 65..TKF_LONGCAP bytes the flat path kept (last call)
    uint64_t n_cut_chunks = 0;     // regions that went through the CUT instantiation (last call)
    void* long_ctl_ptr = nullptr;  // what the control words at counters + 16 describe
    uint32_t long_ctl_cap = 0;
    bool no_flat_long128 = false;  // TK_FLAT_LONG128=0: every long-piece record takes the single-wave merge
    bool no_flat_long = false;     // TK_FLAT_LONG=0: such pieces hand their documents back (the round-1 behaviour; A / B and tests)
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [4]: behind the merge kernels (tk_last_merge_ms)
    // the tail of a batch (documents handed back by the flat kernel) runs on a second stream beside the merge kernels
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_b[3] = {nullptr, nullptr, nullptr};   // flat kernel done (A) | list of handed-back documents on the host (B) | tail done (B)
    DevBuf scratch_rec;            // scratch of the long-piece record kernels (stream A; c->scratch belongs to the tail on stream B)
    DevBuf f_late;                 // documents a long-piece record flagged after the list of handed-back documents was made
    bool serial_tail = false;      // TK_TAIL=serial: the tail behind the merge kernels on the one stream (A / B, tests)
    uint32_t host_syncs = 0;       // host waits of the last flat-pipeline call (diagnostics)
    float pipeline_ms = 0.f, encode_ms = 0.f, merge_ms = 0.f;
    uint64_t n_long_docs = 0;
    uint32_t* dbg_mark = nullptr;  // pinned host memory, only with TK_DEBUG_MARKS
    uint32_t* h_pin = nullptr;     // pinned host words: the per-batch device counters land here with ONE copy
    // pipelined ingestion (tk_encode_batch_pipelined): copy streams, events, the second set of staging buffers
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    DevBuf in_bytes2, in_offs2, out_ids2, out_offs2;
    uint64_t* h_offs_stage[2] = {nullptr, nullptr};   // pinned: slice-relative document offsets going up
    uint64_t h_offs_cap = 0;
    // small batches in one launch (tk_small_kernel): mapped pinned host buffers the kernel reads / writes directly
    uint8_t* hs_in = nullptr;      // [TK_SMALL_MAX_BYTES] text | [TK_SMALL_MAX_DOCS + 1] u64 offsets
    uint32_t* hs_out = nullptr;    // [TK_SMALL_MAX_BYTES + 2 * TK_SMALL_MAX_DOCS] ids | [TK_SMALL_MAX_DOCS + 1] u64 offsets | [4] status
    void* ds_in = nullptr;         // the same buffers as the device sees them
    void* ds_out = nullptr;
    DevBuf s_offs;
    uint64_t n_small_calls = 0;    // calls served by the one-launch path (tk_last_stats_ex)
    bool small_ready = false;      // small_prepare() went through completely
    // memo of merged pieces (tk_hash.h MEMO; include/tekken_hip.h tk_ctx_set_memo)
    DevBuf t_memo, t_memo_log;
    uint32_t memo_log2 = 24;       // entries = 2^memo_log2 (32 bytes each: 512 MB of a 288 GB part), 0 = off; TK_MEMO_LOG2
    uint32_t memo_have_log2 = 0;   // size of the table that is allocated (0: none yet)
    uint32_t memo_epoch = 0;       // calls that used the table so far
    int memo_policy = 0;           // 0 adaptive (pause while the hit rate is low), 1 always on; TK_MEMO_POLICY=always
    uint32_t memo_low_streak = 0, memo_pause = 0;
    bool memo_active_last = false;
    uint64_t memo_hits_last = 0, memo_lookups_last = 0, memo_hits_total = 0, memo_lookups_total = 0;
};

#define TK_SMALL_IDS_CAP (TK_SMALL_MAX_BYTES + 2 * TK_SMALL_MAX_DOCS)
#define TK_SMALL_OUT_OFFS_WORD (TK_SMALL_IDS_CAP)                          /* u32 index of the u64 offsets in hs_out (8-byte aligned) */
#define TK_SMALL_STATUS_WORD (TK_SMALL_OUT_OFFS_WORD + 2 * (TK_SMALL_MAX_DOCS + 1) + 2)

#define TK_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                        \
            return TK_ERR_RUNTIME;                                                                 \
        }                                                                                          \
    } while (0)

static int upload(tk_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    TK_HIP(c, b.reserve(bytes ? bytes : 16));
    if (bytes) TK_HIP(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return TK_OK;
}

extern "C" int tk_ctx_create(const uint8_t* token_bytes, const uint32_t* token_offsets, uint32_t n_ranks,
                             uint32_t num_special_tokens, uint32_t bos_id, uint32_t eos_id, int device_id,
                             tk_ctx** out_ctx) {
    if (!out_ctx) { g_tls_err = "out_ctx is NULL"; return TK_ERR_INVALID_ARG; }
    *out_ctx = nullptr;
    tk_ctx* c = new tk_ctx();
    std::string err;
    bool from_cache = false;   // TK_TABLE_CACHE_DIR: derived tables from a side file keyed by the rank table (row f-2)
    int rc = tk_build_tables_cached(token_bytes, token_offsets, n_ranks, num_special_tokens, bos_id, eos_id, c->host, err, &from_cache);
    if (rc != TK_OK) { g_tls_err = err; delete c; return rc; }

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_tls_err = "no HIP device available (the tokenization path has no CPU fallback)";
        delete c;
        return TK_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) {
        g_tls_err = "device_id out of range";
        delete c;
        return TK_ERR_INVALID_ARG;
    }
    c->device = device_id;
    auto fail = [&](int code) {
        g_tls_err = c->err;
        tk_ctx_destroy(c);
        return code;
    };
    if (hipSetDevice(device_id) != hipSuccess) { c->err = "hipSetDevice failed"; return fail(TK_ERR_RUNTIME); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->err = "hipStreamCreate failed";
        return fail(TK_ERR_RUNTIME);
    }
    for (int i = 0; i < 5; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(TK_ERR_RUNTIME); }
    if (hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(TK_ERR_RUNTIME); }
    for (int i = 0; i < 3; ++i)
        if (hipEventCreateWithFlags(&c->ev_b[i], hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(TK_ERR_RUNTIME); }
    if (const char* tl = getenv("TK_TAIL")) c->serial_tail = strcmp(tl, "serial") == 0;
    if (const char* dg = getenv("TK_DECODE_GROUPS")) c->no_decode_groups = strcmp(dg, "0") == 0;
    if (const char* gl = getenv("TK_DECODE_GROUP_LIMIT")) { const long long v = atoll(gl); if (v > 0 && v < 0x7FFFFF00ll) c->decode_group_limit = (uint32_t)v; }
    if (hipHostMalloc((void**)&c->h_pin, 256, hipHostMallocDefault) != hipSuccess) { c->err = "hipHostMalloc failed"; return fail(TK_ERR_RUNTIME); }

    const TkHostTables& h = c->host;
    if (getenv("TK_DEBUG_LOG"))
        fprintf(stderr, "[tk] tables%s: KEY8 %u slots, KEY16 %u slots (hash mode %u, %llu keys, %llu in their second slot, %llu slots flagged), PAIR %u buckets (%llu pairs)\n",
                from_cache ? " (from cache)" : "", h.key8_mask + 1, h.key_mask + 1, h.key_hash_mode, (unsigned long long)h.n_key, (unsigned long long)h.n_key_second,
                (unsigned long long)h.n_key_spill_slots, h.pair_mask + 1, (unsigned long long)h.n_pairs);
    if ((rc = upload(c, c->t_uc1, h.uc_stage1.data(), h.uc_stage1.size() * 2)) ||
        (rc = upload(c, c->t_uc2, h.uc_stage2.data(), h.uc_stage2.size() * 4)) ||
        (rc = upload(c, c->t_uc2a, h.uc2_stage1.data(), h.uc2_stage1.size() * 2)) ||
        (rc = upload(c, c->t_uc2b, h.uc2_stage2.data(), h.uc2_stage2.size() * 4)) ||
        (rc = upload(c, c->t_key8, h.key8_tab.data(), h.key8_tab.size() * sizeof(tk_key8_entry))) ||
        (rc = upload(c, c->t_key, h.key_tab.data(), h.key_tab.size() * sizeof(tk_key_entry))) ||
        (rc = upload(c, c->t_long, h.long_tab.data(), h.long_tab.size() * sizeof(tk_long_entry))) ||
        (rc = upload(c, c->t_pair, h.pair_tab.data(), h.pair_tab.size() * 8)) ||
        (rc = upload(c, c->t_pair2, h.pair2.data(), h.pair2.size() * 4)) ||
        (rc = upload(c, c->t_pairf, h.pair_filter.data(), h.pair_filter.size() * 4)) ||
        (rc = upload(c, c->t_ucbmp, h.uc_bmp.data(), h.uc_bmp.size() * 4)) ||
        (rc = upload(c, c->t_key64, h.key64_tab.data(), h.key64_tab.size() * sizeof(tk_long_entry))) ||
        (rc = upload(c, c->t_cutk2, h.cut_k2.data(), h.cut_k2.size() * 4)) ||
        (rc = upload(c, c->t_cutg3, h.cut_g3.data(), h.cut_g3.size() * 4)) ||
        (rc = upload(c, c->t_blob, h.blob.data(), h.blob.size())) ||
        (rc = upload(c, c->t_offs, h.offs.data(), h.offs.size() * 4)))
        return fail(rc);
    c->dview = h.host_view();
    c->dview.uc_stage1 = (const uint16_t*)c->t_uc1.p;
    c->dview.uc_stage2 = (const uint32_t*)c->t_uc2.p;
    c->dview.uc2_stage1 = (const uint16_t*)c->t_uc2a.p;
    c->dview.uc2_stage2 = (const uint32_t*)c->t_uc2b.p;
    c->dview.key8_tab = (const tk_key8_entry*)c->t_key8.p;
    c->dview.key_tab = (const tk_key_entry*)c->t_key.p;
    c->dview.long_tab = (const tk_long_entry*)c->t_long.p;
    c->dview.pair_tab = (const uint64_t*)c->t_pair.p;
    c->dview.pair2 = (const uint32_t*)c->t_pair2.p;
    c->dview.pair_filter = (const uint32_t*)c->t_pairf.p;
    c->dview.uc_bmp = (const uint32_t*)c->t_ucbmp.p;
    c->dview.key64_tab = (const tk_long_entry*)c->t_key64.p;
    c->dview.cut_k2 = (const uint32_t*)c->t_cutk2.p;
    c->dview.cut_g3 = (const uint32_t*)c->t_cutg3.p;
    c->dview.blob = (const uint8_t*)c->t_blob.p;

    if (c->counters.reserve(128) != hipSuccess) { c->err = "hipMalloc(counters) failed"; return fail(TK_ERR_RUNTIME); }
    // the wave primitives (DPP wave shifts, bpermute) are checked once on the real device
    uint32_t bad = 1;
    if (hipMemsetAsync(c->counters.p, 0, 128, c->stream) != hipSuccess ||   // (counters 0..15 and the control words behind them)
        tk_launch_wave_selftest((uint32_t*)c->counters.p, c->stream) != hipSuccess ||
        hipMemcpyAsync(&bad, c->counters.p, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        c->err = std::string("wave self-test launch failed: ") + hipGetErrorString(hipGetLastError());
        return fail(TK_ERR_RUNTIME);
    }
    if (bad != 0) {
        c->err = "wave primitive self-test failed on this device (mask " + std::to_string(bad) + ")";
        return fail(TK_ERR_RUNTIME);
    }
    if (const char* fl = getenv("TK_FLAT_LONG")) c->no_flat_long = atoi(fl) == 0;
    if (const char* fl = getenv("TK_FLAT_LONG128")) c->no_flat_long128 = atoi(fl) == 0;
    if (const char* fl = getenv("TK_FLAT_CUT")) c->no_flat_cut = atoi(fl) == 0;
    if (const char* lm = getenv("TK_LONG_MIN")) c->long_min = (uint32_t)atoi(lm);
    if (const char* lz = getenv("TK_LONG_LAZY_MUL")) c->long_lazy_mul = (uint32_t)atoi(lz);
    if (const char* lf = getenv("TK_LONG_FORCE")) c->long_force = (uint32_t)atoi(lf);   // tests: 1 = every long piece through the compacting rounds, 2 = through the lazy rounds
    if (const char* ml = getenv("TK_MEMO_LOG2")) { const int v = atoi(ml); c->memo_log2 = v <= 0 ? 0u : (uint32_t)(v < 10 ? 10 : v > 26 ? 26 : v); }
    if (const char* mp = getenv("TK_MEMO_POLICY")) c->memo_policy = strcmp(mp, "always") == 0 ? 1 : 0;
    if (const char* pl = getenv("TK_PIPELINE"))  // "doc": per-document kernels only, "flat": chunk-per-wave kernel always
        c->pipeline_forced = strcmp(pl, "doc") == 0 ? 2 : strcmp(pl, "flat") == 0 ? 1 : 0;
    *out_ctx = c;
    return TK_OK;
}

extern "C" void tk_ctx_destroy(tk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    DevBuf* bufs[] = {&c->t_key64, &c->t_ucbmp, &c->t_cutk2, &c->t_cutg3, &c->f_cut, &c->t_uc2a, &c->t_uc2b, &c->t_uc1, &c->t_uc2, &c->t_key8, &c->t_key, &c->t_long, &c->t_pair, &c->t_pair2, &c->t_pairf, &c->t_blob, &c->t_offs,
                      &c->t_spblob, &c->t_spoffs, &c->dec_lens, &c->dec_bytes, &c->dec_offs, &c->dec_bits,
                      &c->dec_err, &c->dec_in_ids, &c->dec_in_offs, &c->dec_hi, &c->dec_glens, &c->dec_goffs, &c->t_inline, &c->t_len8,
                      &c->staging, &c->counts, &c->out_ids, &c->out_offs, &c->block_sums, &c->defer_list,
                      &c->scratch, &c->long_list, &c->long_jobs, &c->f_long, &c->counters, &c->in_bytes, &c->in_offs, &c->dbg,
                      &c->f_first, &c->f_tmp, &c->f_lstart, &c->f_flags, &c->f_todo, &c->f_miss, &c->f_mcnt, &c->f_mpfx, &c->f_wfirst, &c->f_info};
    for (DevBuf* b : bufs) b->release();
    for (int i = 0; i < 5; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 3; ++i)
        if (c->ev_b[i]) (void)hipEventDestroy(c->ev_b[i]);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    c->scratch_rec.release();
    c->f_late.release();
    c->t_memo.release();
    c->t_memo_log.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    DevBuf* pbufs[] = {&c->in_bytes2, &c->in_offs2, &c->out_ids2, &c->out_offs2};
    for (DevBuf* b : pbufs) b->release();
    for (int i = 0; i < 2; ++i) {
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
        if (c->h_offs_stage[i]) (void)hipHostFree(c->h_offs_stage[i]);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->hs_in) (void)hipHostFree(c->hs_in);
    if (c->hs_out) (void)hipHostFree(c->hs_out);
    c->s_offs.release();
    delete c;
}

extern "C" int tk_ctx_set_pattern(tk_ctx* c, int mode) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (mode != 0 && mode != 1) { c->err = "pattern mode must be 0 (hard-coded pattern) or 1 (tekken.json pattern)"; return TK_ERR_INVALID_ARG; }
    c->pattern = mode;
    return TK_OK;
}

extern "C" const char* tk_last_error(const tk_ctx* c) { return c ? c->err.c_str() : g_tls_err.c_str(); }

const TkHostTables* tk_ctx_host_tables(const tk_ctx* c) { return c ? &c->host : nullptr; }

// Pass 2 over the n_def documents of c->defer_list: documents with a long piece that missed the vocabulary need the
// scratch-backed cooperative merge.  The scratch is sized from the longest deferred document.
static int run_pass2(tk_ctx* c, TkEncodeArgs& a, const uint64_t* d_offs, uint32_t n_def, hipStream_t s, uint64_t max_waves = 1024) {
    const bool dbg = getenv("TK_DEBUG_LOG") != nullptr;
    uint32_t maxlen32 = 0;
    TK_HIP(c, hipMemsetAsync((uint32_t*)c->counters.p + 3, 0, 4, s));
    TK_HIP(c, tk_launch_defer_maxlen((const uint32_t*)c->defer_list.p, n_def, d_offs, (uint32_t*)c->counters.p + 3, s));
    TK_HIP(c, hipMemcpyAsync(&maxlen32, (uint32_t*)c->counters.p + 3, 4, hipMemcpyDeviceToHost, s));
    TK_HIP(c, hipStreamSynchronize(s));
    const uint64_t maxlen = maxlen32;
    if (dbg) fprintf(stderr, "[tk] pass2: n_def=%u maxlen=%llu\n", n_def, (unsigned long long)maxlen);
    // nodes (4 words per byte) | block minima | successor tokens (1 word per byte); 16-byte aligned slices
    const uint64_t words = ((5 * maxlen + 2 * ((maxlen + 63) / 64) + 64 + 3) / 4) * 4;
    // the grid is launched in blocks of 4 waves and EVERY launched wave owns a scratch slice
    uint64_t waves2 = n_def < max_waves ? n_def : max_waves;
    const uint64_t budget_words = (8ull << 30) / 4;
    if (waves2 * words > budget_words) waves2 = budget_words / words;
    waves2 = ((waves2 + 3) / 4) * 4;
    if (waves2 == 0) waves2 = 4;
    TK_HIP(c, c->scratch.reserve(waves2 * words * 4));
    a.todo_list = (const uint32_t*)c->defer_list.p;
    a.n_todo = n_def;
    a.scratch = (uint32_t*)c->scratch.p;
    a.scratch_words_per_wave = words;
    // documents with a LONG piece that is not a vocabulary key are handed on to the workgroup-per-document kernel
    // (tk_long.hip: the piece is merged in rounds by 16 waves instead of step by step by one)
    uint32_t* d_long_count = (uint32_t*)c->counters.p + 9;
    if (c->long_min) {
        TK_HIP(c, c->long_list.reserve(((size_t)n_def + 1) * 4));
        a.long_list = (uint32_t*)c->long_list.p;
        a.long_count = d_long_count;
        a.long_min = c->long_min < 65u ? 65u : c->long_min;
        a.long_lazy_mul = c->long_lazy_mul;
        a.long_force = c->long_force;
    }
    TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 8, s));
    TK_HIP(c, hipMemsetAsync(d_long_count, 0, 4, s));
    TK_HIP(c, tk_launch_encode(a, 1, (uint32_t)waves2, s));
    if (dbg) { TK_HIP(c, hipStreamSynchronize(s)); fprintf(stderr, "[tk] pass2 kernel done\n"); }
    if (c->long_min) {
        TK_HIP(c, hipMemcpyAsync(c->h_pin + 9, d_long_count, 4, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipStreamSynchronize(s));
        const uint32_t n_long = c->h_pin[9];
        c->n_round_docs += n_long;
        if (n_long) {
            // walk (one wave per document; long pieces become jobs) -> merge the jobs in rounds (one workgroup each) ->
            // squeeze the holes out.  No host sync in between: the merge grid is persistent and reads the job count itself.
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
            const uint64_t fit = budget_words / words >= 4 ? budget_words / words : 4;   // scratch slices the budget allows
            uint64_t walk_waves = ((n_long < 1024u ? n_long : 1024u) + 3) / 4 * 4;
            uint64_t blocks = (uint64_t)cus;                                       // one 16-wave block per CU (158 KB of LDS)
            if (walk_waves > fit) walk_waves = fit / 4 * 4;
            if (blocks > fit) blocks = fit;
            TK_HIP(c, c->scratch.reserve((walk_waves > blocks ? walk_waves : blocks) * words * 4));   // (pass 2 is complete: its slices are free)
            const uint64_t job_cap = maxlen / (a.long_min ? a.long_min : 1) * (uint64_t)n_long + n_long + 16;
            TK_HIP(c, c->long_jobs.reserve(job_cap * sizeof(TkLongJob)));
            uint32_t* d_job_count = (uint32_t*)c->counters.p + 10;
            TkEncodeArgs b = a;
            b.todo_list = (const uint32_t*)c->long_list.p;
            b.n_todo = n_long;
            b.long_list = nullptr;
            b.long_jobs = (TkLongJob*)c->long_jobs.p;
            b.long_job_count = d_job_count;
            b.long_job_cap = (uint32_t)(job_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : job_cap);
            b.scratch = (uint32_t*)c->scratch.p;
            TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 4, s));
            TK_HIP(c, hipMemsetAsync(d_job_count, 0, 4, s));
            TK_HIP(c, tk_launch_encode_long(b, (uint32_t)walk_waves, 0, s));
            TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 4, s));                   // the job queue's ticket counter
            const uint64_t cblocks = (n_long + 3) / 4 < 4096 ? (n_long + 3) / 4 : 4096;
            TK_HIP(c, tk_launch_encode_long_merge(b, (uint32_t)blocks, (uint32_t)cblocks, s));
            if (dbg) { TK_HIP(c, hipStreamSynchronize(s)); fprintf(stderr, "[tk] round-based kernels done: %u documents\n", n_long); }
        }
    }
    return TK_OK;
}

// The same passes WITHOUT a host sync, for the flat pipeline's tail: the documents are a.todo_list = `list`, their number lives
// in device memory (count_dev; NULL: n_bound is exact), n_bound and maxlen bound it and every document's length from above.
// Walk (one wave per document, piece by piece; long pieces become jobs), round-based merges of the jobs, compaction
// (tk_long.hip); TK_LONG_MIN=0: pass 2 alone.
static int enqueue_pass2(tk_ctx* c, TkEncodeArgs a, const uint32_t* list, const uint32_t* count_dev, uint32_t n_bound,
                         uint64_t maxlen, uint64_t n_bytes, hipStream_t s, uint64_t max_waves = 1024) {
    const uint64_t words = ((5 * maxlen + 2 * ((maxlen + 63) / 64) + 64 + 3) / 4) * 4;
    const uint64_t budget_words = (8ull << 30) / 4;
    const uint64_t fit = budget_words / words >= 4 ? budget_words / words : 4;   // scratch slices the budget allows
    uint64_t waves2 = n_bound < max_waves ? n_bound : max_waves;
    if (waves2 > fit) waves2 = fit;
    waves2 = ((waves2 + 3) / 4) * 4;
    if (waves2 == 0) waves2 = 4;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    uint64_t walk_waves = ((n_bound < 1024u ? n_bound : 1024u) + 3) / 4 * 4;
    uint64_t blocks = (uint64_t)cus;                                       // one 16-wave block per CU (158 KB of LDS)
    if (walk_waves > fit) walk_waves = fit / 4 * 4;
    if (blocks > fit) blocks = fit;
    // (with the round-based merges on -- the default -- the mode-1 pass-2 kernel is never launched: only the walk's waves and the
    // merging workgroups own a slice.  Sizing for waves2 as well allocated up to 5.4 GB on the JSON-pattern path, where max_waves
    // is 8192, for a batch with many handed-back 32 KiB documents)
    const uint64_t slices = c->long_min ? std::max(walk_waves, blocks) : waves2;
    TK_HIP(c, c->scratch.reserve(slices * words * 4));
    a.todo_list = list;
    a.n_todo = n_bound;
    a.n_todo_dev = count_dev;
    a.defer_count = (uint32_t*)c->counters.p + 5;                          // (pass 2 defers nothing; count_dev may be counter 1)
    a.scratch = (uint32_t*)c->scratch.p;
    a.scratch_words_per_wave = words;
    if (c->long_min) {
        a.long_min = c->long_min < 65u ? 65u : c->long_min;
        a.long_lazy_mul = c->long_lazy_mul;
        a.long_force = c->long_force;
    }
    TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 4, s));
    TK_HIP(c, hipMemsetAsync((uint32_t*)c->counters.p + 9, 0, 8, s));     // (9: unused on this path) job count
    if (!c->long_min) {
        TK_HIP(c, tk_launch_encode(a, 1, (uint32_t)waves2, s));             // (TK_LONG_MIN=0: no round-based merges -- pass 2 does it all)
    } else {
        // a job is a piece of at least long_min bytes: no more of them than the text holds, nor than every document's share
        uint64_t job_cap = maxlen / a.long_min * (uint64_t)n_bound + n_bound + 16;
        if (job_cap > n_bytes / a.long_min + n_bound + 16) job_cap = n_bytes / a.long_min + n_bound + 16;
        TK_HIP(c, c->long_jobs.reserve(job_cap * sizeof(TkLongJob)));
        // The walk takes EVERY document of the list (the first form ran pass 2 first and walked only the documents in which it
        // met a long piece: two kernels in a row, each as long as its slowest document, the second redoing what the first
        // had done of its documents)
        TkEncodeArgs b = a;
        b.long_list = nullptr;
        b.long_jobs = (TkLongJob*)c->long_jobs.p;
        b.long_job_count = (uint32_t*)c->counters.p + 10;
        b.long_job_cap = (uint32_t)(job_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : job_cap);
        TK_HIP(c, tk_launch_encode_long(b, (uint32_t)walk_waves, 0, s));
        TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 4, s));                   // the job queue's ticket counter
        const uint64_t cblocks = ((uint64_t)n_bound + 3) / 4 < 4096 ? ((uint64_t)n_bound + 3) / 4 : 4096;
        TK_HIP(c, tk_launch_encode_long_merge(b, (uint32_t)blocks, (uint32_t)cblocks, s));
    }
    return TK_OK;
}

// counters layout (u32): [0] work queue head, [1] deferred documents, [2] invalid docs, [3] max deferred length,
// [4] documents the flat path handed back (counted by the counts kernel), [5] scratch, [6..7] total ids, [9] long list, [10] long
// jobs, [11] long-piece records, [12] cut chunks, [13] handed-back documents as listed (tk_flat_todo_kernel), [14] the longest of
// them, [15] documents flagged late by a long-piece record; [16..20] control words of the flat kernel
static int run_pipeline_doc(tk_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offs, uint64_t n_docs, uint64_t n_bytes,
                        int add_bos, int add_eos, hipStream_t s, uint64_t* n_ids) {
    const uint64_t cap = n_bytes + 2 * n_docs + 64;
    TK_HIP(c, c->staging.reserve(cap * 4));
    TK_HIP(c, c->out_ids.reserve(cap * 4));
    TK_HIP(c, c->counts.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->out_offs.reserve((n_docs + 1) * 8));
    TK_HIP(c, c->block_sums.reserve((n_docs / 2048 + 4) * 8));
    TK_HIP(c, c->defer_list.reserve((n_docs + 1) * 4));

    TkEncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = d_bytes;
    a.doc_offs = d_offs;
    a.n_docs = n_docs;
    a.staging = (uint32_t*)c->staging.p;
    a.counts = (uint32_t*)c->counts.p;
    a.work_counter = (uint32_t*)c->counters.p;
    a.defer_count = (uint32_t*)c->counters.p + 1;
    a.defer_list = (uint32_t*)c->defer_list.p;
    a.add_bos = add_bos;
    a.add_eos = add_eos;
    a.t = c->dview;
#ifdef TK_ABLATE   /* `make ablate` builds only */
    if (const char* ab = getenv("TK_DEBUG_ABLATE")) a.dbg_ablate = atoi(ab);  // timing-only experiments
#endif
    if (getenv("TK_DEBUG_MARKS")) {
        if (!c->dbg_mark) {
            TK_HIP(c, hipHostMalloc((void**)&c->dbg_mark, 256, hipHostMallocMapped));
            memset(c->dbg_mark, 0, 256);
        }
        a.dbg_mark = c->dbg_mark;
    }

    TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 64, s));
    TK_HIP(c, hipEventRecord(c->ev[0], s));
    uint64_t want = (n_docs + 7) / 8;
    uint32_t n_waves = (uint32_t)(want < 8192 ? (want ? want : 1) : 8192);
    TK_HIP(c, tk_launch_encode(a, 0, n_waves, s));
    TK_HIP(c, hipEventRecord(c->ev[1], s));
    TK_HIP(c, tk_launch_scan(a.counts, n_docs, (uint64_t*)c->out_offs.p, (uint64_t*)c->block_sums.p, s));
    TK_HIP(c, tk_launch_compact(a.staging, d_offs, a.counts, (const uint64_t*)c->out_offs.p, n_docs,
                                (uint32_t*)c->out_ids.p, s));
    TK_HIP(c, hipEventRecord(c->ev[2], s));
    uint32_t ctr[4] = {0, 0, 0, 0};
    uint64_t total = 0;
    TK_HIP(c, hipMemcpyAsync(ctr, c->counters.p, 16, hipMemcpyDeviceToHost, s));
    TK_HIP(c, hipMemcpyAsync(&total, (uint64_t*)c->out_offs.p + n_docs, 8, hipMemcpyDeviceToHost, s));
    TK_HIP(c, hipStreamSynchronize(s));
    c->n_long_docs = ctr[1];
    const bool dbg = getenv("TK_DEBUG_LOG") != nullptr;
    if (dbg) fprintf(stderr, "[tk] pass1 done: docs=%llu deferred=%u total=%llu\n", (unsigned long long)n_docs, ctr[1], (unsigned long long)total);
    if (ctr[1] != 0 && getenv("TK_DEBUG_SKIP_PASS2") == nullptr) {
        int rc2 = run_pass2(c, a, d_offs, ctr[1], s);
        if (rc2 != TK_OK) return rc2;
        TK_HIP(c, tk_launch_scan(a.counts, n_docs, (uint64_t*)c->out_offs.p, (uint64_t*)c->block_sums.p, s));
        TK_HIP(c, tk_launch_compact(a.staging, d_offs, a.counts, (const uint64_t*)c->out_offs.p, n_docs,
                                    (uint32_t*)c->out_ids.p, s));
        TK_HIP(c, hipEventRecord(c->ev[2], s));
        TK_HIP(c, hipMemcpyAsync(&total, (uint64_t*)c->out_offs.p + n_docs, 8, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipStreamSynchronize(s));
    }
    (void)hipEventElapsedTime(&c->encode_ms, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&c->pipeline_ms, c->ev[0], c->ev[2]);
    *n_ids = total;
    return TK_OK;
}

// MEMO (tk_hash.h): the table is allocated at the first flat-pipeline call that wants it.  Adaptive policy (memo_account): a call's
// hit rate = hits / (hits + pieces of 2..16 bytes the narrow merge kernel still had to merge); on text whose unknown pieces do not
// come back (random code points: BASELINE configs[2]) or that has few of them (a vocabulary fitted to the text), the look-ups cost
// more than the hits return, so after two such calls in a row the table is left alone for 30 calls, then tried again.
// Whatever the policy does, ids never depend on it: an entry is the exact key and the pure merge of its bytes.
static int memo_prepare(tk_ctx* c, TkFlatArgs& fa, hipStream_t s, uint64_t n_bytes) {
    fa.memo_tab = nullptr; fa.memo_mask = 0; fa.memo_epoch = 0; fa.memo_hits = nullptr;
    fa.memo_log = nullptr; fa.memo_log_counts = nullptr; fa.memo_log_per_wave = 0; fa.memo_log_waves = 0;
    c->memo_active_last = false;
    if (c->memo_log2 == 0) return TK_OK;
    // adaptive policy: a call of under 1 MB leaves the table alone (and a context that only ever sees such calls never allocates
    // its 576 MB): memo_account cannot judge a call that small, and its two extra launches are a tenth of its time
    if (c->memo_policy == 0 && n_bytes < (1u << 20)) return TK_OK;
    if (c->memo_policy == 0 && c->memo_pause) { --c->memo_pause; return TK_OK; }
    const size_t bytes = ((size_t)1 << c->memo_log2) * sizeof(tk_memo_entry);
    if (c->memo_have_log2 != c->memo_log2) {
        c->t_memo.release();
        if (c->t_memo.reserve(bytes) != hipSuccess) {      // no room: the memo is an optimisation, the call goes on without it
            (void)hipGetLastError();
            c->memo_log2 = 0; c->memo_have_log2 = 0;
            return TK_OK;
        }
        TK_HIP(c, hipMemsetAsync(c->t_memo.p, 0, bytes, s));
        c->memo_have_log2 = c->memo_log2;
        c->memo_epoch = 0;
    }
    if (c->memo_epoch >= 0xFFFFFFF0u) {                     // the claim word would wrap: start over
        TK_HIP(c, hipMemsetAsync(c->t_memo.p, 0, bytes, s));
        c->memo_epoch = 0;
    }
    // the log of a call's new entries: one stretch per wave of the narrow merge kernel's grid (at most 16 waves on each CU), a
    // quarter of the table in all, at most 2^21 records (what does not fit is dropped and comes again)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    const uint32_t log_waves = (uint32_t)cus * 16u;
    // (TK_MEMO_LOG_LOG2: records of the log, all waves together; measured on the held-out shape: what bounds the hit rate of a table
    // of 2^22 entries is how many new entries a call can log, not the table)
    uint32_t log_log2 = c->memo_log2 > 23 ? 21 : c->memo_log2 - 2;
    if (const char* ll = getenv("TK_MEMO_LOG_LOG2")) { const int v = atoi(ll); if (v >= 8 && v <= 24) log_log2 = (uint32_t)v; }
    const uint32_t log_cap = 1u << log_log2;
    const uint32_t per_wave = log_cap / log_waves > 0 ? log_cap / log_waves : 1u;
    if (c->t_memo_log.reserve((size_t)per_wave * log_waves * sizeof(tk_memo_entry) + (size_t)log_waves * 4) != hipSuccess) {
        (void)hipGetLastError();
        c->t_memo.release();
        c->memo_log2 = 0; c->memo_have_log2 = 0;
        return TK_OK;
    }
    fa.memo_tab = (tk_memo_entry*)c->t_memo.p;
    fa.memo_mask = (1u << c->memo_log2) - 1u;
    fa.memo_log = (tk_memo_entry*)c->t_memo_log.p;
    fa.memo_log_counts = (uint32_t*)(fa.memo_log + (size_t)per_wave * log_waves);
    fa.memo_log_per_wave = per_wave;
    fa.memo_log_waves = log_waves;
    TK_HIP(c, hipMemsetAsync(fa.memo_log_counts, 0, (size_t)log_waves * 4, s));   // (waves the grid does not launch log nothing)
    fa.memo_epoch = ++c->memo_epoch;
    fa.memo_probe = c->memo_epoch > 1 ? 1 : 0;            // (the first call on an empty table: nothing to find, only to fill)
    fa.memo_hits = (uint32_t*)c->counters.p + 24;
    c->memo_active_last = true;
    return TK_OK;
}
static void memo_account(tk_ctx* c, const uint32_t* ctr28, uint64_t n_bytes) {
    c->memo_hits_last = c->memo_lookups_last = 0;
    if (!c->memo_active_last) return;
    c->memo_hits_last = ctr28[24];
    c->memo_lookups_last = (uint64_t)ctr28[24] + ctr28[25];
    c->memo_hits_total += c->memo_hits_last;
    c->memo_lookups_total += c->memo_lookups_last;
    if (c->memo_policy != 0 || c->memo_epoch < 2 || n_bytes < (1u << 20)) return;   // (the first call fills an empty table)
    // does it pay?  A look-up is one more dependent load in the flat kernel's miss path (measured on the 1 M x 512-byte shapes:
    // +0.15 .. 0.18 ms whatever the number of look-ups), a hit saves a merge (~0.075 ms per million): under one hit per 160 bytes of
    // text, or under three hits in ten look-ups (the mixed UTF-8 shape at 28 %: no gain, no loss), the table is left alone for 30 calls.
    const bool pays = c->memo_hits_last * 10 >= c->memo_lookups_last * 3 && c->memo_hits_last * 160 >= n_bytes;
    if (!pays) {
        if (++c->memo_low_streak >= 2) { c->memo_pause = 30; c->memo_low_streak = 0; }
    } else {
        c->memo_low_streak = 0;
    }
}

// The flat pipeline (tk_flat.hip): one wave per 2048-byte region of the packed stream, documents the
// fast path cannot take (non-ASCII, very long runs / pieces) redone by the per-document kernels.
static int run_pipeline_flat(tk_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offs, uint64_t n_docs, uint64_t n_bytes,
                             int add_bos, int add_eos, hipStream_t s, uint64_t* n_ids) {
    const bool dbg = getenv("TK_DEBUG_LOG") != nullptr;
    const uint64_t n_chunks = (n_bytes + TKF_COMMIT - 1) / TKF_COMMIT;
    TK_HIP(c, c->f_first.reserve((n_chunks + 1) * 4));
    TK_HIP(c, c->f_tmp.reserve((n_chunks * TKF_STRIDE + 64) * 4));
    TK_HIP(c, c->f_lstart.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->f_flags.reserve(2 * (n_docs + 1) * 4));   // flags | holes (one memset)
    TK_HIP(c, c->f_todo.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->f_miss.reserve((n_chunks * TKF_MISSCAP + 64) * 4));  // worst case; only the used records are ever touched
    TK_HIP(c, c->f_mcnt.reserve((5 * n_chunks + 1) * 4));   // 4 C miss counts (class-major) | C slot counts (one scan)
    TK_HIP(c, c->f_mpfx.reserve((5 * n_chunks + 2) * 8));
    TK_HIP(c, c->f_info.reserve((n_docs + 1) * 16));
    // one entry per 64 queued pieces: the narrow classes (2..16 bytes), then the wide ones (17..64 bytes)
    const uint64_t wf_narrow = n_chunks * (TKF_MISSOFF2 / 64 + 1) + 64, wf_wide = n_chunks * ((TKF_MISSCAP - TKF_MISSOFF2) / 64 + 1) + 64;
    TK_HIP(c, c->f_wfirst.reserve((wf_narrow + wf_wide) * 4));
    TK_HIP(c, c->counts.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->out_offs.reserve((n_docs + 1) * 8));
    const uint64_t scan_n = n_docs > 5 * n_chunks ? n_docs : 5 * n_chunks;
    TK_HIP(c, c->block_sums.reserve((scan_n / 2048 + 4) * 8));

    TkFlatArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.bytes = d_bytes;
    fa.doc_offs = d_offs;
    fa.n_docs = n_docs;
    fa.n_bytes = n_bytes;
    fa.n_chunks = n_chunks;
    fa.first_doc = (const uint32_t*)c->f_first.p;
    fa.tmp = (uint32_t*)c->f_tmp.p;
    fa.kcount = (uint32_t*)c->f_mcnt.p + 4 * n_chunks;
    fa.lstart = (uint32_t*)c->f_lstart.p;
    fa.flags = (uint32_t*)c->f_flags.p;
    fa.miss_list = (uint32_t*)c->f_miss.p;
    fa.miss_count = (uint32_t*)c->f_mcnt.p;
    fa.miss_prefix = (const uint64_t*)c->f_mpfx.p;
    fa.holes = (uint32_t*)c->f_flags.p + (n_docs + 1);
    fa.wave_first = (uint32_t*)c->f_wfirst.p;
    fa.wave_first_wide = (uint32_t*)c->f_wfirst.p + wf_narrow;
    fa.t = c->dview;
    fa.pattern = c->pattern;
#ifdef TK_ABLATE   /* `make ablate` builds only */
    if (const char* ab = getenv("TK_DEBUG_ABLATE")) fa.dbg_ablate = atoi(ab);  // timing-only experiments
#endif
    // pieces of 65..TKF_LONGCAP bytes stay on the flat path as records (counter 11); TK_FLAT_LONG=0: they hand their documents back
    const uint64_t long_cap = n_bytes / 65 + 1024;
    if (!c->no_flat_long) {
        TK_HIP(c, c->f_long.reserve(long_cap * sizeof(TkFlatLongRec)));
        fa.long_recs = (TkFlatLongRec*)c->f_long.p;
        fa.long_count = (uint32_t*)c->counters.p + 11;
        fa.long_cap = (uint32_t)(c->f_long.cap / sizeof(TkFlatLongRec) > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : c->f_long.cap / sizeof(TkFlatLongRec));
        fa.long_ctl = (const uint32_t*)c->counters.p + 16;
        // the control words behind the counters (not touched by the pre-pass, which clears words 0..15): written when the
        // record buffer changes, i.e. a handful of times in a context's life
        if (c->long_ctl_ptr != c->f_long.p || c->long_ctl_cap != fa.long_cap) {
            const uint64_t pv = (uint64_t)reinterpret_cast<uintptr_t>(c->f_long.p);
            const uint32_t ctl[3] = {(uint32_t)pv, (uint32_t)(pv >> 32), fa.long_cap};
            TK_HIP(c, hipMemcpyAsync((uint32_t*)c->counters.p + 16, ctl, sizeof(ctl), hipMemcpyHostToDevice, s));
            TK_HIP(c, hipStreamSynchronize(s));
            c->long_ctl_ptr = c->f_long.p;
            c->long_ctl_cap = fa.long_cap;
        }
        // the list of the chunks that hold a piece of more than 64 bytes (counter 12, control words 19..20): the flat kernel
        // leaves them to tk_flat_cut_kernel (TK_FLAT_CUT=0: a null list, no cuts)
        void* want_cut = nullptr;
        if (!c->no_flat_cut && c->pattern == 0) {
            TK_HIP(c, c->f_cut.reserve((n_chunks + 1) * 4));
            want_cut = c->f_cut.p;
            fa.cut_list = (uint32_t*)c->f_cut.p;
            fa.cut_count = (const uint32_t*)c->counters.p + 12;
        }
        if (c->cut_ctl_ptr != want_cut) {
            const uint64_t pv = (uint64_t)reinterpret_cast<uintptr_t>(want_cut);
            const uint32_t ctl[2] = {(uint32_t)pv, (uint32_t)(pv >> 32)};
            TK_HIP(c, hipMemcpyAsync((uint32_t*)c->counters.p + 19, ctl, sizeof(ctl), hipMemcpyHostToDevice, s));
            TK_HIP(c, hipStreamSynchronize(s));
            c->cut_ctl_ptr = want_cut;
        }
    }

    // One host sync per batch in the common case.  Everything that depends on device-side counts stays on the device:
    // the merge kernels are persistent, the output buffer takes its upper bound (a document cannot produce more ids
    // than bytes + 2), and the handed-back documents are only COUNTED at first -- if there are any, the per-document
    // kernels run afterwards and counts / scan / assembly are redone.
    const uint32_t extra = (uint32_t)((add_bos ? 1 : 0) + (add_eos ? 1 : 0));
    uint64_t* d_pfx = (uint64_t*)c->f_mpfx.p;                  // prefix sums over [4 C miss counts | C slot counts]
    const uint64_t* d_P = d_pfx + 4 * n_chunks;                // chunk slot prefix sums (offset by the miss total: only differences are used)
    TK_HIP(c, c->out_ids.reserve((n_bytes + 2 * n_docs + 64) * 4));
    { int rcm = memo_prepare(c, fa, s, n_bytes); if (rcm != TK_OK) return rcm; }
    TK_HIP(c, hipEventRecord(c->ev[3], s));
    // (the pre-pass also clears the per-document flags / holes and the 16 counter words: no memset launches)
    TK_HIP(c, tk_launch_flat_firstdoc(d_offs, n_docs, n_chunks, (uint32_t*)c->f_first.p, fa.flags, fa.holes, (uint32_t*)c->counters.p, s));
    TK_HIP(c, hipEventRecord(c->ev[0], s));
    TK_HIP(c, tk_launch_flat(fa, s));
    TK_HIP(c, hipEventRecord(c->ev[1], s));
    if (fa.dbg_ablate & 24) {  // timing-only runs that stop inside the flat kernel: nothing downstream has valid input
        TK_HIP(c, hipEventRecord(c->ev[2], s));
        TK_HIP(c, hipStreamSynchronize(s));
        (void)hipEventElapsedTime(&c->encode_ms, c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&c->pipeline_ms, c->ev[3], c->ev[2]);
        c->n_flagged = 0;
        *n_ids = 0;
        return TK_OK;
    }
    // The list of the handed-back documents is made on a second stream (B) right behind the flat kernel, beside the merge
    // kernels, and comes to the host first: if there are such documents, the per-document passes over them run on B while
    // stream A is still merging -- the tail of a batch (Zipf shape: pass 2, walk, round-based merges, compaction; each as long
    // as its longest document) is hidden instead of appended.  Two host waits per call: the list, the result.
    uint32_t* ctr = (uint32_t*)c->counters.p;
    const bool serial = c->serial_tail;
    hipStream_t sb = serial ? s : c->stream_b;
    c->host_syncs = 0;
    // From here on work is queued on TWO streams.  Whatever way this function is left on an error -- a failed reserve, a failed
    // launch --, both have drained before the caller sees it: the next call's pre-pass runs on `s` alone and would otherwise race
    // the tail of this batch (pass 1, the walk, the merges) for counts, staging and the counters.
    struct JoinStreams {
        hipStream_t a, b;
        bool armed = true;
        ~JoinStreams() {
            if (!armed) return;
            (void)hipStreamSynchronize(b);
            (void)hipStreamSynchronize(a);
        }
    } join_guard{s, sb};
    if (!serial) {
        TK_HIP(c, hipEventRecord(c->ev_b[0], s));
        TK_HIP(c, hipStreamWaitEvent(sb, c->ev_b[0], 0));
    }
    TK_HIP(c, tk_launch_flat_todo(fa.flags, d_offs, n_docs, (uint32_t*)c->f_todo.p, ctr + 13, ctr + 14, sb));
    if (!serial) {
        TK_HIP(c, hipMemcpyAsync(c->h_pin + 32, ctr, 64, hipMemcpyDeviceToHost, sb));
        TK_HIP(c, hipEventRecord(c->ev_b[1], sb));
    }
    TK_HIP(c, tk_launch_scan(fa.miss_count, 5 * n_chunks, d_pfx, (uint64_t*)c->block_sums.p, s));
    TK_HIP(c, tk_launch_merge(fa, (uint32_t*)c->counters.p + 25, s));
    TK_HIP(c, hipEventRecord(c->ev[4], s));
    uint64_t total = 0;
    auto finish = [&](int final_pass, bool wait) -> int {
        TK_HIP(c, tk_launch_flat_counts(d_offs, n_docs, n_bytes, n_chunks, d_P, fa.lstart, fa.flags, fa.holes, extra,
                                        (uint32_t*)c->counts.p, c->f_info.p, final_pass, ctr + 4, s));
        TK_HIP(c, tk_launch_scan((const uint32_t*)c->counts.p, n_docs, (uint64_t*)c->out_offs.p, (uint64_t*)c->block_sums.p, s));
        TK_HIP(c, tk_launch_flat_assemble(n_docs, c->f_info.p, fa.kcount, (const uint64_t*)c->out_offs.p, fa.tmp,
                                          (const uint32_t*)c->staging.p, (uint32_t*)c->out_ids.p, c->host.bos_id,
                                          c->host.eos_id, add_bos, add_eos, (uint64_t*)(ctr + 6),
                                          final_pass ? nullptr : (const uint32_t*)ctr + 4, s));
        TK_HIP(c, hipEventRecord(c->ev[2], s));
        // every counter of the batch with one copy into pinned memory (6..7: the total, left there by the assembly)
        TK_HIP(c, hipMemcpyAsync(c->h_pin, ctr, 112, hipMemcpyDeviceToHost, s));   // (24, 25: memo hits, narrow pieces left to merge)
        if (wait) {
            TK_HIP(c, hipStreamSynchronize(s));
            ++c->host_syncs;
            memcpy(&total, c->h_pin + 6, 8);
        }
        return TK_OK;
    };
    // optimistic: if nothing was handed back and no long-piece record waits, this IS the result (the assembly copies nothing otherwise)
    int rc = finish(0, serial);
    if (rc != TK_OK) return rc;
    const uint32_t* early = c->h_pin;
    if (!serial) {
        TK_HIP(c, hipEventSynchronize(c->ev_b[1]));
        ++c->host_syncs;
        early = c->h_pin + 32;
    }
    const uint32_t n_todo = early[13];
    uint32_t n_lrec = early[11];
    const uint64_t maxlen = early[14];
    c->n_cut_chunks = early[12];
    c->n_flagged = n_todo;
    c->n_long_docs = 0;
    c->n_long_recs = 0;
    if (dbg) fprintf(stderr, "[tk] flat: docs=%llu chunks=%llu handed back=%u (longest %llu bytes) long-piece records=%u cut chunks=%u\n",
                     (unsigned long long)n_docs, (unsigned long long)n_chunks, n_todo, (unsigned long long)maxlen, n_lrec, early[12]);
    if (n_todo == 0 && n_lrec == 0) {
        if (!serial) {
            TK_HIP(c, hipStreamSynchronize(s));
            ++c->host_syncs;
            memcpy(&total, c->h_pin + 6, 8);
        }
    } else {
        TK_HIP(c, c->staging.reserve((n_bytes + 2 * n_docs + 64) * 4));
        TK_HIP(c, c->defer_list.reserve((n_docs + 1) * 4));
        TkEncodeArgs a;
        memset(&a, 0, sizeof(a));
        a.bytes = d_bytes;
        a.doc_offs = d_offs;
        a.n_docs = n_docs;
        a.staging = (uint32_t*)c->staging.p;
        a.counts = (uint32_t*)c->counts.p;
        a.work_counter = ctr;
        a.defer_count = ctr + 1;
        a.defer_list = (uint32_t*)c->defer_list.p;
        a.add_bos = add_bos;
        a.add_eos = add_eos;
        a.t = c->dview;
        a.pattern = c->pattern;
        if (n_lrec) {
            // stream A: the long-piece records, one wave each (lookup / merge into the reserved slots); a piece that turns out
            // longer than TKF_LONGCAP flags its document and puts it on the late list (counter 15)
            if (n_lrec > fa.long_cap) n_lrec = fa.long_cap;
            c->n_long_recs = n_lrec;
            fa.long_merge128 = c->no_flat_long128 ? 0 : 1;
            const uint32_t lwaves = ((n_lrec < 8192u ? n_lrec : 8192u) + 3u) / 4u * 4u;
            TK_HIP(c, c->scratch_rec.reserve((size_t)lwaves * TKF_LONG_SCRATCH_WORDS * 4));
            TK_HIP(c, c->f_late.reserve((n_docs + 1) * 4));
            fa.late_list = (uint32_t*)c->f_late.p;
            fa.late_count = ctr + 15;
            TK_HIP(c, tk_launch_flat_long(fa, ctr, (uint32_t*)c->scratch_rec.p, TKF_LONG_SCRATCH_WORDS, lwaves, s));
        }
        if (n_todo) {
            // stream B: the per-document path over the handed-back documents -- pass 1 (mode 3), then, for what it defers (the
            // count stays on the device), pass 2 and the round-based kernels
            a.todo_list = (const uint32_t*)c->f_todo.p;
            a.n_todo = n_todo;
            if (c->pattern == 1) {
                // JSON pattern: the handed-back documents go straight to the piece-by-piece path with its sequential matcher
                int rc2 = enqueue_pass2(c, a, (const uint32_t*)c->f_todo.p, nullptr, n_todo, maxlen, n_bytes, sb, 8192);
                if (rc2 != TK_OK) return rc2;
            } else {
                TK_HIP(c, hipMemsetAsync(ctr, 0, 8, sb));
                const uint64_t want = ((uint64_t)n_todo + 7) / 8;
                TK_HIP(c, tk_launch_encode(a, 3, (uint32_t)(want < 8192 ? want : 8192), sb));
                int rc2 = enqueue_pass2(c, a, (const uint32_t*)c->defer_list.p, ctr + 1, n_todo, maxlen, n_bytes, sb);
                if (rc2 != TK_OK) return rc2;
            }
            if (!serial) {
                TK_HIP(c, hipEventRecord(c->ev_b[2], sb));
                TK_HIP(c, hipStreamWaitEvent(s, c->ev_b[2], 0));
            }
        }
        rc = finish(1, true);
        if (rc != TK_OK) return rc;
        if (c->h_pin[5] == 0xDEADu) { c->err = "internal: the long-piece job list overflowed"; return TK_ERR_RUNTIME; }   // (set by tk_long_walk_kernel)
        c->n_long_docs = c->pattern == 1 ? n_todo : c->h_pin[1];
        c->n_round_docs += n_todo && c->long_min ? c->h_pin[10] : 0;   // (here: long pieces merged in rounds)
        const uint32_t n_late = n_lrec ? c->h_pin[15] : 0;
        if (n_late) {
            // rare: documents that a long-piece record flagged (an open piece of more than TKF_LONGCAP bytes without a cut) after
            // the list was made -- the same passes over the late list, in sequence, and the result is assembled again
            c->n_flagged += n_late;
            a.todo_list = (const uint32_t*)c->f_late.p;
            a.n_todo = n_late;
            a.n_todo_dev = nullptr;
            a.defer_count = ctr + 1;
            if (c->pattern == 1) {
                TK_HIP(c, hipMemcpyAsync(c->defer_list.p, c->f_late.p, (size_t)n_late * 4, hipMemcpyDeviceToDevice, s));
                int rc2 = run_pass2(c, a, d_offs, n_late, s, 8192);
                if (rc2 != TK_OK) return rc2;
                c->n_long_docs += n_late;
            } else {
                TK_HIP(c, hipMemsetAsync(ctr, 0, 8, s));
                const uint64_t want = ((uint64_t)n_late + 7) / 8;
                TK_HIP(c, tk_launch_encode(a, 3, (uint32_t)(want < 8192 ? want : 8192), s));
                uint32_t n_def = 0;
                TK_HIP(c, hipMemcpyAsync(&n_def, ctr + 1, 4, hipMemcpyDeviceToHost, s));
                TK_HIP(c, hipStreamSynchronize(s));
                c->n_long_docs += n_def;
                if (n_def) {
                    int rc2 = run_pass2(c, a, d_offs, n_def, s);
                    if (rc2 != TK_OK) return rc2;
                }
            }
            rc = finish(1, true);
            if (rc != TK_OK) return rc;
        }
    }
    (void)hipEventElapsedTime(&c->encode_ms, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&c->pipeline_ms, c->ev[3], c->ev[2]);
    (void)hipEventElapsedTime(&c->merge_ms, c->ev[1], c->ev[4]);
    memo_account(c, c->h_pin, n_bytes);
    *n_ids = total;
    join_guard.armed = false;      // (every path to here has waited for both streams already)
    return TK_OK;
}

// Row f-3, opt-in (tk_ctx_set_pattern(ctx, 1)) with TK_PIPELINE=doc: EVERY document takes the piece-by-piece path of
// pass 2 with the sequential matcher tk_match_end2 (one wave per document).  The default route for the JSON pattern is
// the flat pipeline with tk_flat_json_kernel; this is what its handed-back documents use, and the A / B form.
static int run_pipeline_seq(tk_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offs, uint64_t n_docs, uint64_t n_bytes,
                            int add_bos, int add_eos, hipStream_t s, uint64_t* n_ids) {
    const uint64_t cap = n_bytes + 2 * n_docs + 64;
    TK_HIP(c, c->staging.reserve(cap * 4));
    TK_HIP(c, c->out_ids.reserve(cap * 4));
    TK_HIP(c, c->counts.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->out_offs.reserve((n_docs + 1) * 8));
    TK_HIP(c, c->block_sums.reserve((n_docs / 2048 + 4) * 8));
    TK_HIP(c, c->defer_list.reserve((n_docs + 1) * 4));
    TkEncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = d_bytes;
    a.doc_offs = d_offs;
    a.n_docs = n_docs;
    a.staging = (uint32_t*)c->staging.p;
    a.counts = (uint32_t*)c->counts.p;
    a.work_counter = (uint32_t*)c->counters.p;
    a.defer_count = (uint32_t*)c->counters.p + 1;
    a.defer_list = (uint32_t*)c->defer_list.p;
    a.add_bos = add_bos;
    a.add_eos = add_eos;
    a.pattern = 1;
    a.t = c->dview;
    c->n_flagged = 0;
    c->n_long_docs = n_docs;
    uint64_t total = 0;
    TK_HIP(c, hipEventRecord(c->ev[0], s));
    TK_HIP(c, hipEventRecord(c->ev[3], s));
    if (n_docs) {
        TK_HIP(c, tk_launch_iota((uint32_t*)c->defer_list.p, n_docs, s));
        int rc = run_pass2(c, a, d_offs, (uint32_t)n_docs, s, 8192);
        if (rc != TK_OK) return rc;
    }
    TK_HIP(c, hipEventRecord(c->ev[1], s));
    TK_HIP(c, tk_launch_scan(a.counts, n_docs, (uint64_t*)c->out_offs.p, (uint64_t*)c->block_sums.p, s));
    TK_HIP(c, tk_launch_compact(a.staging, d_offs, a.counts, (const uint64_t*)c->out_offs.p, n_docs, (uint32_t*)c->out_ids.p, s));
    TK_HIP(c, hipEventRecord(c->ev[2], s));
    TK_HIP(c, hipMemcpyAsync(&total, (uint64_t*)c->out_offs.p + n_docs, 8, hipMemcpyDeviceToHost, s));
    TK_HIP(c, hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&c->encode_ms, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&c->pipeline_ms, c->ev[0], c->ev[2]);
    *n_ids = total;
    return TK_OK;
}

// Pipeline choice: the flat pipeline, unless TK_PIPELINE=doc asks for the per-document kernels alone (tests / A-B runs).
static int run_pipeline(tk_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offs, uint64_t n_docs, uint64_t n_bytes,
                        int add_bos, int add_eos, hipStream_t s, uint64_t* n_ids) {
    // (row f-3: TK_PIPELINE=doc selects the purely sequential form of the opt-in; the per-document window kernels only
    // know the hard-coded pattern)
    if (c->pattern == 1 && c->pipeline_forced == 2) return run_pipeline_seq(c, d_bytes, d_offs, n_docs, n_bytes, add_bos, add_eos, s, n_ids);
    c->use_flat = c->pipeline_forced != 2;
    if (c->use_flat) return run_pipeline_flat(c, d_bytes, d_offs, n_docs, n_bytes, add_bos, add_eos, s, n_ids);
    c->n_flagged = 0;
    return run_pipeline_doc(c, d_bytes, d_offs, n_docs, n_bytes, add_bos, add_eos, s, n_ids);
}

extern "C" int tk_encode_batch_device(tk_ctx* c, const void* d_bytes, const void* d_doc_offsets, uint64_t n_docs,
                                      uint64_t n_bytes, int add_bos, int add_eos, void* hip_stream, void** d_ids,
                                      void** d_out_offsets, uint64_t* n_ids) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!d_doc_offsets || (!d_bytes && n_bytes) || !d_ids || !d_out_offsets || !n_ids) {
        c->err = "null argument";
        return TK_ERR_INVALID_ARG;
    }
    if (n_docs >= 0xFFFFFFF0ull) { c->err = "too many documents in one batch"; return TK_ERR_INVALID_ARG; }
    TK_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;  // NULL = HIP's null stream: ordered after the caller's own work on it
    int rc = run_pipeline(c, (const uint8_t*)d_bytes, (const uint64_t*)d_doc_offsets, n_docs, n_bytes, add_bos,
                          add_eos, s, n_ids);
    if (rc != TK_OK) return rc;
    *d_ids = c->out_ids.p;
    *d_out_offsets = c->out_offs.p;
    return TK_OK;
}

// The same entry with the checks a host caller gets from tk_encode_batch ("C callers get a `validate` flag"):
// TK_CHECK_OFFSETS -- d_doc_offsets[0] == 0, non-decreasing, [n_docs] == n_bytes, or TK_ERR_INVALID_ARG (without it a bad offset
// array is out-of-bounds indexing on the device); TK_CHECK_UTF8 -- every document is well-formed UTF-8 on its own (which includes:
// no document starts inside a code point), or TK_ERR_INVALID_UTF8; implies the offsets check.  One small kernel and one host wait
// each, before anything else runs.
extern "C" int tk_encode_batch_device_ex(tk_ctx* c, const void* d_bytes, const void* d_doc_offsets, uint64_t n_docs,
                                         uint64_t n_bytes, int add_bos, int add_eos, int checks, void* hip_stream, void** d_ids,
                                         void** d_out_offsets, uint64_t* n_ids) {
    if (!c) return TK_ERR_INVALID_ARG;
    if (checks & ~(TK_CHECK_OFFSETS | TK_CHECK_UTF8)) { std::lock_guard<std::mutex> lock(c->mu); c->err = "unknown check flag"; return TK_ERR_INVALID_ARG; }
    if (checks) {
        std::lock_guard<std::mutex> lock(c->mu);
        if (!d_doc_offsets || (!d_bytes && n_bytes)) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
        if (n_docs >= 0xFFFFFFF0ull) { c->err = "too many documents in one batch"; return TK_ERR_INVALID_ARG; }
        TK_HIP(c, hipSetDevice(c->device));
        hipStream_t s = (hipStream_t)hip_stream;
        uint32_t* d_bad = (uint32_t*)c->counters.p + 2;
        uint32_t bad = 0;
        TK_HIP(c, hipMemsetAsync(d_bad, 0, 4, s));
        TK_HIP(c, tk_launch_check_offsets((const uint64_t*)d_doc_offsets, n_docs, n_bytes, d_bad, s));
        TK_HIP(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipStreamSynchronize(s));
        if (bad) { c->err = "doc_offsets must start at 0, be non-decreasing and end at n_bytes (" + std::to_string(bad) + " violation(s))"; return TK_ERR_INVALID_ARG; }
        if (checks & TK_CHECK_UTF8) {
            TK_HIP(c, hipMemsetAsync(d_bad, 0, 4, s));
            TK_HIP(c, tk_launch_validate((const uint8_t*)d_bytes, (const uint64_t*)d_doc_offsets, n_docs, d_bad, s));
            TK_HIP(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
            TK_HIP(c, hipStreamSynchronize(s));
            if (bad) { c->err = std::to_string(bad) + " document(s) are not valid UTF-8"; return TK_ERR_INVALID_UTF8; }
        }
    }
    return tk_encode_batch_device(c, d_bytes, d_doc_offsets, n_docs, n_bytes, add_bos, add_eos, hip_stream, d_ids, d_out_offsets, n_ids);
}

static int check_offsets(tk_ctx* c, const uint64_t* doc_offsets, uint64_t n_docs) {
    if (doc_offsets[0] != 0) { c->err = "doc_offsets[0] must be 0"; return TK_ERR_INVALID_ARG; }
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_offsets[d + 1] < doc_offsets[d]) { c->err = "doc_offsets must be non-decreasing"; return TK_ERR_INVALID_ARG; }
    return TK_OK;
}

static int stage_input(tk_ctx* c, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs) {
    const uint64_t n_bytes = doc_offsets[n_docs];
    TK_HIP(c, c->in_bytes.reserve(n_bytes + 64));
    TK_HIP(c, c->in_offs.reserve((n_docs + 1) * 8));
    if (n_bytes) TK_HIP(c, hipMemcpyAsync(c->in_bytes.p, bytes, n_bytes, hipMemcpyHostToDevice, c->stream));
    TK_HIP(c, hipMemcpyAsync(c->in_offs.p, doc_offsets, (n_docs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// small batches in ONE launch (tk_small_kernel).  The reference's own signature is one &str per call
// (src/tekkenizer.rs:378-405): through the batch pipeline that is about ten launches, two copies and a host sync.
// ------------------------------------------------------------------------------------------
static bool small_eligible(const tk_ctx* c, uint64_t n_docs, uint64_t n_bytes) {
    static const bool off = getenv("TK_NO_SMALL_PATH") != nullptr;
    return !off && c->pattern == 0 && c->pipeline_forced == 0 && n_docs >= 1 && n_docs <= TK_SMALL_MAX_DOCS && n_bytes <= TK_SMALL_MAX_BYTES;
}

static int small_prepare(tk_ctx* c) {
    if (c->small_ready) return TK_OK;
    // (ready only once EVERY step below went through: a call that fails half-way leaves the flag clear, and the next call
    // starts over with what is still missing instead of running on null pointers)
    const size_t in_bytes = TK_SMALL_MAX_BYTES + (TK_SMALL_MAX_DOCS + 1) * 8;
    const size_t out_bytes = (size_t)(TK_SMALL_STATUS_WORD + 4) * 4;
    if (!c->hs_in) TK_HIP(c, hipHostMalloc((void**)&c->hs_in, in_bytes, hipHostMallocMapped));
    if (!c->hs_out) TK_HIP(c, hipHostMalloc((void**)&c->hs_out, out_bytes, hipHostMallocMapped));
    TK_HIP(c, hipHostGetDevicePointer(&c->ds_in, c->hs_in, 0));
    TK_HIP(c, hipHostGetDevicePointer(&c->ds_out, c->hs_out, 0));
    TK_HIP(c, c->staging.reserve((size_t)(TK_SMALL_IDS_CAP + 64) * 4));
    TK_HIP(c, c->counts.reserve((TK_SMALL_MAX_DOCS + 1) * 4));
    TK_HIP(c, c->in_bytes.reserve(TK_SMALL_MAX_BYTES + 64));
    TK_HIP(c, c->s_offs.reserve((TK_SMALL_MAX_DOCS + 1) * 8));
    c->small_ready = true;
    return TK_OK;
}

// hs_in holds the text and (behind it) the document offsets.  *fallback = true: a document needs pass 2 (a piece that
// does not fit a window) -- nothing was produced, the caller takes the batch pipeline.  Otherwise the ids are in
// hs_out[0 .. *n_ids) and the id offsets at hs_out + TK_SMALL_OUT_OFFS_WORD when this returns.
static int run_small(tk_ctx* c, uint64_t n_docs, uint64_t n_bytes, int add_bos, int add_eos, uint64_t* n_ids, bool* fallback) {
    uint64_t* h_offs = (uint64_t*)(c->hs_in + TK_SMALL_MAX_BYTES);
    volatile uint32_t* status = (volatile uint32_t*)(c->hs_out + TK_SMALL_STATUS_WORD);
    TkEncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_docs = n_docs;
    // A few short strings are read by the kernel straight from pinned host memory (one PCIe round trip per window); beyond
    // that the copy engine is the better reader.
    if (n_bytes <= 4096 && n_docs <= 16) {
        a.bytes = (const uint8_t*)c->ds_in;
        a.doc_offs = (const uint64_t*)((const uint8_t*)c->ds_in + TK_SMALL_MAX_BYTES);
    } else {
        if (n_bytes) TK_HIP(c, hipMemcpyAsync(c->in_bytes.p, c->hs_in, n_bytes, hipMemcpyHostToDevice, c->stream));
        TK_HIP(c, hipMemcpyAsync(c->s_offs.p, h_offs, (n_docs + 1) * 8, hipMemcpyHostToDevice, c->stream));
        a.bytes = (const uint8_t*)c->in_bytes.p;
        a.doc_offs = (const uint64_t*)c->s_offs.p;
    }
    a.staging = (uint32_t*)c->staging.p;
    a.counts = (uint32_t*)c->counts.p;
    a.add_bos = add_bos;
    a.add_eos = add_eos;
    a.t = c->dview;
    status[0] = 0xFFFFFFFFu;
    uint32_t* d_out = (uint32_t*)c->ds_out;
    TK_HIP(c, tk_launch_small(a, d_out, (uint64_t*)(d_out + TK_SMALL_OUT_OFFS_WORD), d_out + TK_SMALL_STATUS_WORD, c->stream));
    TK_HIP(c, hipStreamSynchronize(c->stream));
    if (status[0] == 0xFFFFFFFFu) { c->err = "the small-batch kernel did not report"; return TK_ERR_RUNTIME; }
    *fallback = status[0] != 0u;
    *n_ids = status[1];
    if (!*fallback) {
        c->n_small_calls++;
        c->n_flagged = 0; c->n_long_docs = 0; c->pipeline_ms = 0.f; c->encode_ms = 0.f;
    }
    return TK_OK;
}

/* Tekkenizer::encode for ONE &str with a caller-owned output (the reference's own call shape): no allocation, and for
 * texts of up to 64 KiB one kernel launch.  ids_capacity >= len + 2 always suffices. */
extern "C" int tk_encode_one(tk_ctx* c, const uint8_t* text, uint64_t len, int add_bos, int add_eos, uint32_t* ids_out,
                             uint64_t ids_capacity, uint64_t* n_ids_out) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if ((!text && len) || !n_ids_out || (!ids_out && ids_capacity)) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    *n_ids_out = 0;
    TK_HIP(c, hipSetDevice(c->device));
    uint64_t n_ids = 0;
    if (small_eligible(c, 1, len)) {
        int rc = small_prepare(c);
        if (rc != TK_OK) return rc;
        if (len) memcpy(c->hs_in, text, len);
        uint64_t* h_offs = (uint64_t*)(c->hs_in + TK_SMALL_MAX_BYTES);
        h_offs[0] = 0; h_offs[1] = len;
        bool fallback = false;
        if ((rc = run_small(c, 1, len, add_bos, add_eos, &n_ids, &fallback)) != TK_OK) return rc;
        if (!fallback) {
            *n_ids_out = n_ids;
            if (n_ids > ids_capacity) { c->err = "ids_out is too small"; return TK_ERR_INVALID_ARG; }
            if (n_ids) memcpy(ids_out, c->hs_out, n_ids * 4);
            return TK_OK;
        }
    }
    const uint64_t offs[2] = {0, len};
    int rc = stage_input(c, text, offs, 1);
    if (rc != TK_OK) return rc;
    rc = run_pipeline(c, (const uint8_t*)c->in_bytes.p, (const uint64_t*)c->in_offs.p, 1, len, add_bos, add_eos, c->stream, &n_ids);
    if (rc != TK_OK) return rc;
    *n_ids_out = n_ids;
    if (n_ids > ids_capacity) { c->err = "ids_out is too small"; return TK_ERR_INVALID_ARG; }
    if (n_ids) TK_HIP(c, hipMemcpy(ids_out, c->out_ids.p, n_ids * 4, hipMemcpyDeviceToHost));
    return TK_OK;
}

extern "C" int tk_encode_batch(tk_ctx* c, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs,
                               int add_bos, int add_eos, int validate_utf8, tk_result* out) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!doc_offsets || !out || (!bytes && doc_offsets[n_docs])) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    if (n_docs >= 0xFFFFFFF0ull) { c->err = "too many documents in one batch"; return TK_ERR_INVALID_ARG; }
    memset(out, 0, sizeof(*out));
    int rc = check_offsets(c, doc_offsets, n_docs);
    if (rc != TK_OK) return rc;
    TK_HIP(c, hipSetDevice(c->device));
    const uint64_t n_bytes = doc_offsets[n_docs];
    if (small_eligible(c, n_docs, n_bytes)) {
        // one launch for the whole batch; UTF-8 is validated on the host (same RFC 3629 rules as tk_validate_kernel)
        if (validate_utf8) {
            uint64_t bad = 0;
            for (uint64_t d = 0; d < n_docs; ++d)
                if (!tekken::utf8_valid(bytes + doc_offsets[d], doc_offsets[d + 1] - doc_offsets[d])) ++bad;
            if (bad) { c->err = std::to_string(bad) + " document(s) are not valid UTF-8"; return TK_ERR_INVALID_UTF8; }
        }
        if ((rc = small_prepare(c)) != TK_OK) return rc;
        if (n_bytes) memcpy(c->hs_in, bytes, n_bytes);
        memcpy(c->hs_in + TK_SMALL_MAX_BYTES, doc_offsets, (n_docs + 1) * 8);
        uint64_t n_ids = 0;
        bool fallback = false;
        if ((rc = run_small(c, n_docs, n_bytes, add_bos, add_eos, &n_ids, &fallback)) != TK_OK) return rc;
        if (!fallback) {
            uint32_t* h_ids = (uint32_t*)tk_pinned_get((n_ids ? n_ids : 1) * 4);
            uint64_t* h_offs = (uint64_t*)tk_pinned_get((n_docs + 1) * 8);
            if (!h_ids || !h_offs) { tk_pinned_put(h_ids); tk_pinned_put(h_offs); c->err = "hipHostMalloc failed"; return TK_ERR_RUNTIME; }
            if (n_ids) memcpy(h_ids, c->hs_out, n_ids * 4);
            memcpy(h_offs, c->hs_out + TK_SMALL_OUT_OFFS_WORD, (n_docs + 1) * 8);
            out->ids = h_ids; out->offsets = h_offs; out->n_ids = n_ids; out->n_docs = n_docs;
            return TK_OK;
        }
    }
    if ((rc = stage_input(c, bytes, doc_offsets, n_docs)) != TK_OK) return rc;
    if (validate_utf8) {
        uint32_t bad = 0;
        TK_HIP(c, hipMemsetAsync((uint32_t*)c->counters.p + 2, 0, 4, c->stream));
        TK_HIP(c, tk_launch_validate((const uint8_t*)c->in_bytes.p, (const uint64_t*)c->in_offs.p, n_docs,
                                     (uint32_t*)c->counters.p + 2, c->stream));
        TK_HIP(c, hipMemcpyAsync(&bad, (uint32_t*)c->counters.p + 2, 4, hipMemcpyDeviceToHost, c->stream));
        TK_HIP(c, hipStreamSynchronize(c->stream));
        if (bad) {
            c->err = std::to_string(bad) + " document(s) are not valid UTF-8";
            return TK_ERR_INVALID_UTF8;
        }
    }
    uint64_t n_ids = 0;
    rc = run_pipeline(c, (const uint8_t*)c->in_bytes.p, (const uint64_t*)c->in_offs.p, n_docs, n_bytes, add_bos,
                      add_eos, c->stream, &n_ids);
    if (rc != TK_OK) return rc;
    // (pinned buffers from the process-wide pool: no hipHostMalloc per call once the pool is warm)
    uint32_t* h_ids = (uint32_t*)tk_pinned_get((n_ids ? n_ids : 1) * 4);
    uint64_t* h_offs = (uint64_t*)tk_pinned_get((n_docs + 1) * 8);
    if (!h_ids || !h_offs) { tk_pinned_put(h_ids); tk_pinned_put(h_offs); c->err = "hipHostMalloc failed"; return TK_ERR_RUNTIME; }
    hipError_t e = hipSuccess;
    if (n_ids) e = hipMemcpyAsync(h_ids, c->out_ids.p, n_ids * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_offs, c->out_offs.p, (n_docs + 1) * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        tk_pinned_put(h_ids); tk_pinned_put(h_offs);
        c->err = std::string("result copy failed: ") + hipGetErrorString(e);
        return TK_ERR_RUNTIME;
    }
    out->ids = h_ids;
    out->offsets = h_offs;
    out->n_ids = n_ids;
    out->n_docs = n_docs;
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// pipelined ingestion (row f-4)
// ------------------------------------------------------------------------------------------
extern "C" void* tk_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void tk_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int tk_encode_batch_pipelined(tk_ctx* c, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs,
                                         int add_bos, int add_eos, uint64_t slice_bytes, uint32_t* ids_out, uint64_t ids_capacity,
                                         uint64_t* offsets_out, uint64_t* n_ids_out) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!doc_offsets || !offsets_out || !n_ids_out || (!ids_out && ids_capacity) || (!bytes && doc_offsets[n_docs])) {
        c->err = "null argument";
        return TK_ERR_INVALID_ARG;
    }
    if (n_docs >= 0xFFFFFFF0ull) { c->err = "too many documents in one batch"; return TK_ERR_INVALID_ARG; }
    *n_ids_out = 0;
    int rc = check_offsets(c, doc_offsets, n_docs);
    if (rc != TK_OK) return rc;
    TK_HIP(c, hipSetDevice(c->device));
    if (!c->s_in) {
        TK_HIP(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
        TK_HIP(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            TK_HIP(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            TK_HIP(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
    }
    if (slice_bytes == 0) slice_bytes = 32ull << 20;
    // slices of whole documents: [cut[k], cut[k + 1])
    std::vector<uint64_t> cut(1, 0);
    uint64_t max_bytes = 0, max_docs = 0;
    for (uint64_t d = 0; d < n_docs;) {
        const uint64_t b0 = doc_offsets[d];
        uint64_t e = d + 1;                               // at least one document per slice, however long it is
        while (e < n_docs && doc_offsets[e + 1] - b0 <= slice_bytes && e - d < (1ull << 22)) ++e;
        cut.push_back(e);
        if (doc_offsets[e] - b0 > max_bytes) max_bytes = doc_offsets[e] - b0;
        if (e - d > max_docs) max_docs = e - d;
        d = e;
    }
    const size_t n_slices = cut.size() - 1;
    offsets_out[0] = 0;
    if (n_slices == 0) return TK_OK;
    // staging: two input sets, two output sets (run_pipeline writes c->out_ids / c->out_offs: the sets are swapped per slice)
    DevBuf* inb[2] = {&c->in_bytes, &c->in_bytes2};
    DevBuf* ino[2] = {&c->in_offs, &c->in_offs2};
    for (int i = 0; i < 2; ++i) {
        TK_HIP(c, inb[i]->reserve(max_bytes + 64));
        TK_HIP(c, ino[i]->reserve((max_docs + 1) * 8));
    }
    TK_HIP(c, c->out_ids.reserve((max_bytes + 2 * max_docs + 64) * 4));
    TK_HIP(c, c->out_ids2.reserve((max_bytes + 2 * max_docs + 64) * 4));
    TK_HIP(c, c->out_offs.reserve((max_docs + 1) * 8));
    TK_HIP(c, c->out_offs2.reserve((max_docs + 1) * 8));
    if (c->h_offs_cap < max_docs + 1) {
        for (int i = 0; i < 2; ++i) {
            if (c->h_offs_stage[i]) (void)hipHostFree(c->h_offs_stage[i]);
            c->h_offs_stage[i] = nullptr;
            TK_HIP(c, hipHostMalloc((void**)&c->h_offs_stage[i], (max_docs + 1) * 8, hipHostMallocDefault));
        }
        c->h_offs_cap = max_docs + 1;
    }
    auto upload_slice = [&](size_t k) -> int {            // host -> device of slice k on the input stream
        const int b = (int)(k & 1);
        const uint64_t d0 = cut[k], d1 = cut[k + 1], b0 = doc_offsets[d0], nb = doc_offsets[d1] - b0;
        uint64_t* ho = c->h_offs_stage[b];
        for (uint64_t d = d0; d <= d1; ++d) ho[d - d0] = doc_offsets[d] - b0;
        if (nb) TK_HIP(c, hipMemcpyAsync(inb[b]->p, bytes + b0, nb, hipMemcpyHostToDevice, c->s_in));
        TK_HIP(c, hipMemcpyAsync(ino[b]->p, ho, (d1 - d0 + 1) * 8, hipMemcpyHostToDevice, c->s_in));
        TK_HIP(c, hipEventRecord(c->ev_in[b], c->s_in));
        return TK_OK;
    };
    uint64_t id_base = 0;
    std::vector<uint64_t> slice_ids(n_slices, 0);
    float pipe_ms = 0.f, enc_ms = 0.f;
    uint64_t flagged = 0, longd = 0;
    // the offsets staging of slice k is rewritten by upload_slice(k + 2): that copy must have been consumed -- it has, the
    // kernels of slice k (which waited for it) are complete when run_pipeline returns
    // (inside the loop a failing HIP call sets rc and leaves the loop: the drain below must run whatever happened)
#define TK_HIP_BRK(call)                                                                           \
    {                                                                                              \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            c->err = std::string(#call) + ": " + hipGetErrorString(_e);                            \
            rc = TK_ERR_RUNTIME;                                                                   \
            break;                                                                                 \
        }                                                                                          \
    }
    rc = upload_slice(0);
    for (size_t k = 0; rc == TK_OK && k < n_slices; ++k) {
        const int b = (int)(k & 1);
        if (k + 1 < n_slices && (rc = upload_slice(k + 1)) != TK_OK) break;
        const uint64_t d0 = cut[k], d1 = cut[k + 1], nb = doc_offsets[d1] - doc_offsets[d0];
        TK_HIP_BRK(hipStreamWaitEvent(c->stream, c->ev_in[b], 0));
        if (k >= 2) TK_HIP_BRK(hipStreamWaitEvent(c->stream, c->ev_out[b], 0));   // the ids of slice k - 2 have left this output set
        uint64_t n_ids = 0;
        rc = run_pipeline(c, (const uint8_t*)inb[b]->p, (const uint64_t*)ino[b]->p, d1 - d0, nb, add_bos, add_eos, c->stream, &n_ids);
        if (rc != TK_OK) break;
        pipe_ms += c->pipeline_ms; enc_ms += c->encode_ms; flagged += c->n_flagged; longd += c->n_long_docs;
        slice_ids[k] = n_ids;
        if (id_base + n_ids > ids_capacity) {
            *n_ids_out = id_base + n_ids;
            c->err = "ids_out is too small";
            rc = TK_ERR_INVALID_ARG;
            break;
        }
        // device -> host on the output stream (run_pipeline returned after its stream drained: the ids are complete)
        if (n_ids) TK_HIP_BRK(hipMemcpyAsync(ids_out + id_base, c->out_ids.p, n_ids * 4, hipMemcpyDeviceToHost, c->s_out));
        TK_HIP_BRK(hipMemcpyAsync(offsets_out + d0 + 1, (const uint64_t*)c->out_offs.p + 1, (d1 - d0) * 8, hipMemcpyDeviceToHost, c->s_out));
        TK_HIP_BRK(hipEventRecord(c->ev_out[b], c->s_out));
        std::swap(c->out_ids, c->out_ids2);
        std::swap(c->out_offs, c->out_offs2);
        id_base += n_ids;
    }
#undef TK_HIP_BRK
    // drain the copy streams whatever happened (buffers must not be in flight when the call returns)
    (void)hipStreamSynchronize(c->s_in);
    hipError_t e = hipStreamSynchronize(c->s_out);
    if (rc != TK_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
    if (e != hipSuccess) { c->err = std::string("result copy failed: ") + hipGetErrorString(e); return TK_ERR_RUNTIME; }
    // slice-relative id offsets -> batch offsets
    uint64_t base = 0;
    for (size_t k = 0; k < n_slices; ++k) {
        if (base)
            for (uint64_t d = cut[k] + 1; d <= cut[k + 1]; ++d) offsets_out[d] += base;
        base += slice_ids[k];
    }
    c->pipeline_ms = pipe_ms; c->encode_ms = enc_ms; c->n_flagged = flagged; c->n_long_docs = longd;
    *n_ids_out = id_base;
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// 18-bit wire format of ids (multi-GPU gather)
// ------------------------------------------------------------------------------------------
extern "C" uint64_t tk_ids18_bytes(uint64_t n_ids) { return ((2 * n_ids + 3) & ~3ull) + 4 * ((n_ids + 15) / 16); }

extern "C" int tk_pack_ids18_device(tk_ctx* c, const void* d_ids, uint64_t n_ids, void* d_packed, void* hip_stream) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if ((!d_ids || !d_packed) && n_ids) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    TK_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;
    uint32_t* d_bad = (uint32_t*)c->counters.p + 8;
    TK_HIP(c, hipMemsetAsync(d_bad, 0, 4, s));
    TK_HIP(c, tk_launch_pack18((const uint32_t*)d_ids, n_ids, d_packed, d_bad, s));
    TK_HIP(c, hipMemcpyAsync(c->h_pin + 8, d_bad, 4, hipMemcpyDeviceToHost, s));
    TK_HIP(c, hipStreamSynchronize(s));
    if (c->h_pin[8]) { c->err = "an id does not fit 18 bits"; return TK_ERR_INVALID_ARG; }
    return TK_OK;
}

extern "C" int tk_unpack_ids18_device(tk_ctx* c, const void* d_packed, uint64_t n_ids, void* d_ids, void* hip_stream) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if ((!d_ids || !d_packed) && n_ids) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    TK_HIP(c, hipSetDevice(c->device));
    TK_HIP(c, tk_launch_unpack18(d_packed, n_ids, (uint32_t*)d_ids, (hipStream_t)hip_stream));
    return TK_OK;
}

extern "C" void tk_free_result(tk_result* r) {
    if (!r) return;
    tk_pinned_put(r->ids);
    tk_pinned_put(r->offsets);
    memset(r, 0, sizeof(*r));
}

extern "C" const uint32_t* tk_debug_marks(const tk_ctx* c) { return c ? c->dbg_mark : nullptr; }

extern "C" float tk_last_merge_ms(const tk_ctx* c) { return c ? c->merge_ms : 0.f; }

extern "C" int tk_last_timing(const tk_ctx* c, float* pipeline_ms, float* encode_kernel_ms) {
    if (!c) return TK_ERR_INVALID_ARG;
    if (pipeline_ms) *pipeline_ms = c->pipeline_ms;
    if (encode_kernel_ms) *encode_kernel_ms = c->encode_ms;
    return TK_OK;
}

extern "C" int tk_ctx_set_memo(tk_ctx* c, int log2_entries, int policy) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (log2_entries != 0 && (log2_entries < 10 || log2_entries > 26)) { c->err = "memo size: log2_entries must be 0 (off) or 10..26"; return TK_ERR_INVALID_ARG; }
    if (policy != 0 && policy != 1) { c->err = "memo policy must be 0 (adaptive) or 1 (always)"; return TK_ERR_INVALID_ARG; }
    c->memo_log2 = (uint32_t)log2_entries;
    c->memo_policy = policy;
    c->memo_pause = 0; c->memo_low_streak = 0;
    if (log2_entries == 0) {
        TK_HIP(c, hipSetDevice(c->device));
        TK_HIP(c, hipDeviceSynchronize());
        c->t_memo.release();
        c->t_memo_log.release();
        c->memo_have_log2 = 0;
    }
    return TK_OK;
}
extern "C" int tk_ctx_memo_clear(tk_ctx* c) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    c->memo_have_log2 = 0;         // (the next call that wants the table clears it on its own stream)
    c->memo_pause = 0; c->memo_low_streak = 0;
    return TK_OK;
}
extern "C" int tk_memo_stats(const tk_ctx* c, uint64_t* lookups_last, uint64_t* hits_last, uint64_t* lookups_total, uint64_t* hits_total, int* active_last) {
    if (!c) return TK_ERR_INVALID_ARG;
    if (lookups_last) *lookups_last = c->memo_lookups_last;
    if (hits_last) *hits_last = c->memo_hits_last;
    if (lookups_total) *lookups_total = c->memo_lookups_total;
    if (hits_total) *hits_total = c->memo_hits_total;
    if (active_last) *active_last = c->memo_active_last ? 1 : 0;
    return TK_OK;
}

extern "C" uint64_t tk_small_path_calls(const tk_ctx* c) { return c ? c->n_small_calls : 0; }
extern "C" uint64_t tk_round_path_docs(const tk_ctx* c) { return c ? c->n_round_docs : 0; }
extern "C" uint64_t tk_long_piece_records(const tk_ctx* c) { return c ? c->n_long_recs : 0; }
extern "C" uint64_t tk_cut_chunks(const tk_ctx* c) { return c ? c->n_cut_chunks : 0; }
extern "C" uint64_t tk_last_host_syncs(const tk_ctx* c) { return c ? c->host_syncs : 0; }

extern "C" int tk_last_stats(const tk_ctx* c, uint64_t* n_long_docs, uint64_t* reserved) {
    if (!c) return TK_ERR_INVALID_ARG;
    if (n_long_docs) *n_long_docs = c->n_long_docs;
    if (reserved) *reserved = c->n_flagged;  // documents the flat path handed back to the per-document kernels
    return TK_OK;
}

extern "C" int tk_split_batch(tk_ctx* c, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs,
                              uint8_t* out_is_start) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!doc_offsets || (!bytes && doc_offsets[n_docs]) || !out_is_start) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    int rc = check_offsets(c, doc_offsets, n_docs);
    if (rc != TK_OK) return rc;
    TK_HIP(c, hipSetDevice(c->device));
    const uint64_t n_bytes = doc_offsets[n_docs];
    if ((rc = stage_input(c, bytes, doc_offsets, n_docs)) != TK_OK) return rc;
    TK_HIP(c, c->dbg.reserve(n_bytes + 64));
    TK_HIP(c, c->staging.reserve((n_bytes + 2 * n_docs + 64) * 4));
    TK_HIP(c, c->counts.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->defer_list.reserve((n_docs + 1) * 4));
    TkEncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = (const uint8_t*)c->in_bytes.p;
    a.doc_offs = (const uint64_t*)c->in_offs.p;
    a.n_docs = n_docs;
    a.staging = (uint32_t*)c->staging.p;
    a.counts = (uint32_t*)c->counts.p;
    a.work_counter = (uint32_t*)c->counters.p;
    a.defer_count = (uint32_t*)c->counters.p + 1;
    a.defer_list = (uint32_t*)c->defer_list.p;
    a.dbg_starts = (uint8_t*)c->dbg.p;
    a.split_only = 1;
    a.t = c->dview;
    TK_HIP(c, hipMemsetAsync(c->counters.p, 0, 64, c->stream));
    TK_HIP(c, hipMemsetAsync(c->dbg.p, 0, n_bytes + 64, c->stream));
    uint64_t want = (n_docs + 7) / 8;
    TK_HIP(c, tk_launch_encode(a, 2, (uint32_t)(want < 8192 ? (want ? want : 1) : 8192), c->stream));
    if (n_bytes) TK_HIP(c, hipMemcpyAsync(out_is_start, c->dbg.p, n_bytes, hipMemcpyDeviceToHost, c->stream));
    TK_HIP(c, hipStreamSynchronize(c->stream));
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------
extern "C" int tk_ctx_set_special_tokens(tk_ctx* c, const uint8_t* blob, const uint32_t* offs, uint32_t n) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!offs || n != c->host.num_special || (!blob && n && offs[n])) {
        c->err = "special token strings: need exactly num_special_tokens entries";
        return TK_ERR_INVALID_ARG;
    }
    TK_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = upload(c, c->t_spblob, blob, n ? offs[n] : 0)) || (rc = upload(c, c->t_spoffs, offs, ((size_t)n + 1) * 4))) return rc;
    c->have_specials = true;
    return TK_OK;
}

static int run_decode(tk_ctx* c, const uint32_t* d_ids, const uint64_t* d_id_offs, uint64_t n_docs, uint64_t n_ids, int policy,
                      hipStream_t s, uint64_t* n_bytes, uint64_t* bad_doc) {
    if (policy < TK_POLICY_IGNORE || policy > TK_POLICY_RAISE) { c->err = "invalid policy"; return TK_ERR_INVALID_ARG; }
    if (policy == TK_POLICY_KEEP && !c->have_specials) {
        c->err = "TK_POLICY_KEEP needs tk_ctx_set_special_tokens first";
        return TK_ERR_INVALID_ARG;
    }
    TK_HIP(c, c->dec_lens.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->dec_offs.reserve((n_docs + 1) * 8));
    TK_HIP(c, c->dec_err.reserve(64));
    TK_HIP(c, c->dec_hi.reserve((n_docs + 1) * 4));
    TK_HIP(c, c->block_sums.reserve((n_docs / 2048 + 4) * 8));
    if (!c->t_inline.p) {
        // by rank: the token's bytes and length in ONE 16-byte entry (tokens of up to 15 bytes), and the length alone in a byte
        const TkHostTables& h = c->host;
        std::vector<uint8_t> inl((size_t)h.n_ranks * 16 + 16, 0), l8((size_t)h.n_ranks + 16, 0);
        for (uint32_t r = 0; r < h.n_ranks; ++r) {
            const uint32_t len = h.offs[r + 1] - h.offs[r];
            l8[r] = (uint8_t)(len < 255u ? len : 255u);
            if (len <= 15u) {
                memcpy(&inl[(size_t)r * 16], h.blob.data() + h.offs[r], len);
                inl[(size_t)r * 16 + 15] = (uint8_t)len;
            } else {
                inl[(size_t)r * 16 + 15] = 0xFFu;
            }
        }
        int rcu;
        if ((rcu = upload(c, c->t_inline, inl.data(), inl.size())) || (rcu = upload(c, c->t_len8, l8.data(), l8.size()))) return rcu;
    }
    TkDecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.ids = d_ids;
    a.id_offs = d_id_offs;
    a.n_ids = n_ids;
    a.n_docs = n_docs;
    a.lens = (uint32_t*)c->dec_lens.p;
    a.out_offs = (uint64_t*)c->dec_offs.p;
    a.err = (unsigned long long*)c->dec_err.p;
    a.doc_hi = (uint32_t*)c->dec_hi.p;
    a.tok_blob = (const uint8_t*)c->t_blob.p;
    a.tok_offs = (const uint32_t*)c->t_offs.p;
    a.tok_inline = (const uint8_t*)c->t_inline.p;
    a.tok_len8 = (const uint8_t*)c->t_len8.p;
    a.sp_blob = (const uint8_t*)c->t_spblob.p;
    a.sp_offs = (const uint32_t*)c->t_spoffs.p;
    a.n_ranks = c->host.n_ranks;
    a.num_special = c->host.num_special;
    a.policy = policy;
    // Lengths by GROUPS of 16 documents (tk_decode_grouplen_kernel): the emit kernel only needs to know where a group's text begins
    // and writes the documents' offsets itself.  A group whose text reaches 4 GiB (err[3]) sends the call through the per-document
    // length pass instead.
    const uint64_t n_groups = (n_docs + TK_DECODE_GROUP_DOCS - 1) / TK_DECODE_GROUP_DOCS;
    TK_HIP(c, c->dec_glens.reserve((n_groups + 1) * 4));
    TK_HIP(c, c->dec_goffs.reserve((n_groups + 2) * 8));
    a.glens = (uint32_t*)c->dec_glens.p;
    a.group_limit = c->decode_group_limit;
    TK_HIP(c, hipMemsetAsync(c->dec_err.p, 0xFF, 32, s));
    TK_HIP(c, hipEventRecord(c->ev[0], s));
    uint64_t total = 0;
    unsigned long long err[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    const bool by_groups = !c->no_decode_groups;
    if (by_groups) {
        TK_HIP(c, tk_launch_decode_grouplen(a, s));
        TK_HIP(c, tk_launch_scan(a.glens, n_groups, (uint64_t*)c->dec_goffs.p, (uint64_t*)c->block_sums.p, s));
        TK_HIP(c, hipMemcpyAsync(&total, (uint64_t*)c->dec_goffs.p + n_groups, 8, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipMemcpyAsync(err, c->dec_err.p, 32, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipStreamSynchronize(s));
        if (n_docs == 0) total = 0;
    }
    if (by_groups && err[3] == ~0ull) {
        a.goffs = (const uint64_t*)c->dec_goffs.p;
    } else {
        TK_HIP(c, tk_launch_decode_doclen(a, s));
        TK_HIP(c, tk_launch_scan(a.lens, n_docs, (uint64_t*)c->dec_offs.p, (uint64_t*)c->block_sums.p, s));
        TK_HIP(c, hipMemcpyAsync(&total, (uint64_t*)c->dec_offs.p + n_docs, 8, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipMemcpyAsync(err, c->dec_err.p, 16, hipMemcpyDeviceToHost, s));
        TK_HIP(c, hipStreamSynchronize(s));
    }
    auto doc_of = [&](uint64_t id_index, uint64_t* out) -> int {
        // first document whose id range contains id_index: binary search on the device offsets (rare path)
        uint64_t lo = 0, hi = n_docs;
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2, v = 0;
            TK_HIP(c, hipMemcpy(&v, d_id_offs + mid + 1, 8, hipMemcpyDeviceToHost));
            if (v <= id_index) lo = mid + 1; else hi = mid;
        }
        *out = lo;
        return TK_OK;
    };
    TK_HIP(c, c->dec_bytes.reserve(total + 64));
    TK_HIP(c, c->dec_bits.reserve((total / 32 + 4) * 4));
    a.out_bytes = (uint8_t*)c->dec_bytes.p;
    a.run_bits = (uint32_t*)c->dec_bits.p;
    TK_HIP(c, hipMemsetAsync(c->dec_bits.p, 0, (total / 32 + 4) * 4, s));
    TK_HIP(c, tk_launch_decode_emit(a, s));
    TK_HIP(c, tk_launch_decode_validate(a, s));
    TK_HIP(c, hipEventRecord(c->ev[2], s));
    TK_HIP(c, hipMemcpyAsync(err, c->dec_err.p, 24, hipMemcpyDeviceToHost, s));
    TK_HIP(c, hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&c->pipeline_ms, c->ev[0], c->ev[2]);
    c->encode_ms = 0.f;
    if (err[0] != ~0ull || err[1] != ~0ull || err[2] != ~0ull) {
        // Some document makes the reference return Err.  The GPU found WHICH documents; the class of the
        // error of the first one is decided by walking that single document's groups in the reference's
        // order (src/tekkenizer.rs:463-560) -- error classification only, no result is computed here.
        uint64_t first = err[2];
        for (int k = 0; k < 2; ++k) {
            if (err[k] == ~0ull) continue;
            uint64_t d = 0;
            int rc = doc_of(err[k], &d);
            if (rc != TK_OK) return rc;
            if (d < first) first = d;
        }
        if (bad_doc) *bad_doc = first;
        uint64_t range[2] = {0, 0};
        TK_HIP(c, hipMemcpy(range, d_id_offs + first, 16, hipMemcpyDeviceToHost));
        std::vector<uint32_t> hid((size_t)(range[1] - range[0]));
        if (!hid.empty()) TK_HIP(c, hipMemcpy(hid.data(), d_ids + range[0], hid.size() * 4, hipMemcpyDeviceToHost));
        const TkHostTables& h = c->host;
        size_t g0 = 0;
        while (g0 < hid.size()) {
            const bool sp = hid[g0] < h.num_special;
            size_t g1 = g0 + 1;
            while (g1 < hid.size() && (hid[g1] < h.num_special) == sp) ++g1;
            if (sp) {
                if (policy == TK_POLICY_RAISE) {
                    c->err = "Decoding tokens that contain special tokens is not allowed (document " + std::to_string(first) + ")";
                    return TK_ERR_SPECIAL_POLICY;
                }
            } else {
                std::string run;
                for (size_t k = g0; k < g1; ++k) {
                    const uint32_t r = hid[k] - h.num_special;
                    if (r >= h.n_ranks) {
                        c->err = "DecodeKeyError: invalid token for decoding: " + std::to_string(r) + " (document " + std::to_string(first) + ")";
                        return TK_ERR_RUNTIME;
                    }
                    run.append((const char*)h.blob.data() + h.offs[r], h.offs[r + 1] - h.offs[r]);
                }
                if (!tekken::utf8_valid((const uint8_t*)run.data(), run.size())) {
                    c->err = "FromUtf8Error: invalid utf-8 sequence (document " + std::to_string(first) + ")";
                    return TK_ERR_RUNTIME;
                }
            }
            g0 = g1;
        }
        c->err = "decode: device flagged document " + std::to_string(first) + " but the host walk found no error";
        return TK_ERR_RUNTIME;
    }
    *n_bytes = total;
    return TK_OK;
}

extern "C" int tk_decode_batch_device(tk_ctx* c, const void* d_ids, const void* d_id_offsets, uint64_t n_docs, uint64_t n_ids,
                                      int policy, void* hip_stream, void** d_bytes, void** d_out_offsets, uint64_t* n_bytes,
                                      uint64_t* bad_doc) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!d_id_offsets || (!d_ids && n_ids) || !d_bytes || !d_out_offsets || !n_bytes) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    TK_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;  // NULL = HIP's null stream: ordered after the caller's own work on it
    int rc = run_decode(c, (const uint32_t*)d_ids, (const uint64_t*)d_id_offsets, n_docs, n_ids, policy, s, n_bytes, bad_doc);
    if (rc != TK_OK) return rc;
    *d_bytes = c->dec_bytes.p;
    *d_out_offsets = c->dec_offs.p;
    return TK_OK;
}

extern "C" int tk_decode_batch(tk_ctx* c, const uint32_t* ids, const uint64_t* id_offsets, uint64_t n_docs, int policy,
                               tk_text_result* out, uint64_t* bad_doc) {
    if (!c) return TK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!id_offsets || !out || (!ids && id_offsets[n_docs])) { c->err = "null argument"; return TK_ERR_INVALID_ARG; }
    memset(out, 0, sizeof(*out));
    int rc = check_offsets(c, id_offsets, n_docs);
    if (rc != TK_OK) return rc;
    TK_HIP(c, hipSetDevice(c->device));
    const uint64_t n_ids = id_offsets[n_docs];
    TK_HIP(c, c->dec_in_ids.reserve((n_ids + 1) * 4));
    TK_HIP(c, c->dec_in_offs.reserve((n_docs + 1) * 8));
    if (n_ids) TK_HIP(c, hipMemcpyAsync(c->dec_in_ids.p, ids, n_ids * 4, hipMemcpyHostToDevice, c->stream));
    TK_HIP(c, hipMemcpyAsync(c->dec_in_offs.p, id_offsets, (n_docs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    uint64_t n_bytes = 0;
    rc = run_decode(c, (const uint32_t*)c->dec_in_ids.p, (const uint64_t*)c->dec_in_offs.p, n_docs, n_ids, policy, c->stream,
                    &n_bytes, bad_doc);
    if (rc != TK_OK) return rc;
    uint8_t* hb = (uint8_t*)tk_pinned_get(n_bytes ? n_bytes : 1);
    uint64_t* ho = (uint64_t*)tk_pinned_get((n_docs + 1) * 8);
    if (!hb || !ho) { tk_pinned_put(hb); tk_pinned_put(ho); c->err = "hipHostMalloc failed"; return TK_ERR_RUNTIME; }
    hipError_t e = hipSuccess;
    if (n_bytes) e = hipMemcpyAsync(hb, c->dec_bytes.p, n_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ho, c->dec_offs.p, (n_docs + 1) * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        tk_pinned_put(hb); tk_pinned_put(ho);
        c->err = std::string("result copy failed: ") + hipGetErrorString(e);
        return TK_ERR_RUNTIME;
    }
    out->bytes = hb;
    out->offsets = ho;
    out->n_bytes = n_bytes;
    out->n_docs = n_docs;
    return TK_OK;
}

extern "C" void tk_free_text_result(tk_text_result* r) {
    if (!r) return;
    tk_pinned_put(r->bytes);
    tk_pinned_put(r->offsets);
    memset(r, 0, sizeof(*r));
}
