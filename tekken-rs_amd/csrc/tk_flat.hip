// tk_flat.hip -- gfx950 kernels of the flat (chunk-per-wave) tokenization path, see tk_flat_impl.h.
//
//   tk_flat_firstdoc_kernel   per chunk: how many documents start below its loaded region
//   tk_flat_kernel            split + lookup of one region (64 x TKF_W bytes) per wave, ids chunk-dense
//   tk_flat_mode1_kernel      the same for tables built with the strong key hash; tk_flat_json_kernel: the split rules of the
//                             JSON pattern of tekken.json (opt-in); tk_flat_split_kernel: the per-byte
//                             piece-start flags of tk_split_batch compiled in (`make ablate`: the timing ablations too)
//   tk_merge_kernel           byte-pair merge of the queued pieces (2..16 bytes) that missed the vocabulary, one lane per piece
//   tk_merge_wide_kernel      the same for pieces of 17..64 bytes (32- and 64-entry LDS columns)
//   tk_flat_long_kernel       the records of pieces of 65..256 bytes: end of the piece where the chunk did not see it, whole-piece
//                             lookup; tk_flat_long128_kernel: 65..128 bytes, one lane per piece; tk_flat_long_coop_kernel:
//                             129..256 bytes, one wave per piece with the parts in registers
//   tk_flat_todo_kernel       flagged documents -> list for the per-document kernel
//   tk_flat_counts_kernel     ids per document from the chunk prefix sums and the document-start ranks
//   tk_flat_assemble_kernel   chunk-dense ids -> packed ids in document order with BOS / EOS
//                             (reference src/tekkenizer.rs:390-402)
//
// Integer / byte work, no MFMA; roofline: HBM by the byte accounting of DESIGN.md, in practice VALU issue (section 6).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "tk_kernels.h"
#include "tk_wave_hip.h"
#include "tk_flat_impl.h"

#define TKF_BLOCK 256
#ifndef TKF_OCC
#define TKF_OCC __attribute__((amdgpu_waves_per_eu(7, 7)))  /* the LDS slices of 28 waves fill the CU: 7 waves per SIMD, <= 72 VGPRs */
#endif

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_firstdoc_kernel(const uint64_t* __restrict__ doc_offs, uint64_t n_docs,
                                                                      uint64_t n_chunks, uint32_t* __restrict__ first_doc,
                                                                      uint32_t* __restrict__ flags, uint32_t* __restrict__ holes,
                                                                      uint32_t* __restrict__ counters16) {
    // first_doc[c] = number of documents d with doc_offs[d] < lo(c), lo(c) = max(c * COMMIT - HL, 0);
    // document d owns the chunks whose lo lies in (doc_offs[d], doc_offs[d + 1]]  (the last document: everything above)
    const uint64_t d = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    if (d == 0 && n_chunks) first_doc[0] = 0u;
    if (d < 16 || d == 24) counters16[d] = 0u;      // the batch's device counters (24: memo hits)
    if (d <= n_docs) { flags[d] = 0u; holes[d] = 0u; }
    if (d >= n_docs) return;
    const uint64_t s = doc_offs[d], e = doc_offs[d + 1];
    const uint64_t c_lo = (s + TKF_HL) / TKF_COMMIT + 1;
    uint64_t c_hi = d + 1 == n_docs ? n_chunks - 1 : (e + TKF_HL) / TKF_COMMIT;
    if (n_chunks == 0) return;
    if (c_hi > n_chunks - 1) c_hi = n_chunks - 1;
    for (uint64_t c = c_lo; c <= c_hi; ++c) first_doc[c] = (uint32_t)(d + 1);
}

template <int DBG, int MODE, int PAT = 0, int MEMO = 1>
__device__ __forceinline__ void tk_flat_kernel_body(const TkFlatArgs& a, uint32_t* lds_all) {
    const int lane = wv_lane();
    // the wave number is wave-uniform: say so, and the chunk index and everything addressed by it stay in scalar registers
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* lds = lds_all + wv * TKF_LDS_WORDS;
    tk_flat_init_lds(a, lds, lane);
    uint64_t c_begin = (uint64_t)blockIdx.x * (TKF_BLOCK / 64) + wv, c_end = a.n_chunks, c_step = (uint64_t)gridDim.x * (TKF_BLOCK / 64);
    if (gridDim.x >= 8) {
        // XCD-aware: blocks b and b + 8 share an XCD (and its L2), so the blocks of one residue class take ONE contiguous
        // eighth of the chunks -- neighbouring chunks share their halo bytes and the cache lines of the per-chunk /
        // per-document arrays they read and write.  (Placement is a speed matter only.)
        const uint64_t label = blockIdx.x & 7u, nb = (gridDim.x - label + 7u) / 8u;
        c_begin = a.n_chunks * label / 8 + (uint64_t)(blockIdx.x >> 3) * (TKF_BLOCK / 64) + wv;
        c_end = a.n_chunks * (label + 1) / 8;
        c_step = nb * (TKF_BLOCK / 64);
    }
    for (uint64_t c = c_begin; c < c_end; c += c_step) tk_flat_chunk<DBG, MODE, PAT, 0, MEMO>(a, c, lane, lds);
    if (MEMO) tk_flat_flush_memo_hits(a, lds, lane);
}

__global__ __launch_bounds__(TKF_BLOCK) TKF_OCC void tk_flat_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS];
    tk_flat_kernel_body<0, 0, 0, 0>(a, lds_all);
}
// the same with the look-up in the memo of merged pieces (launched when the call uses the table: a.memo_tab != NULL)
__global__ __launch_bounds__(TKF_BLOCK) TKF_OCC void tk_flat_memo_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS];
    tk_flat_kernel_body<0, 0, 0, 1>(a, lds_all);
}

// the tables were built with the strong key hash (mode 1: the cheap one could not place the vocabulary)
__global__ __launch_bounds__(TKF_BLOCK) TKF_OCC void tk_flat_mode1_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS];
    tk_flat_kernel_body<0, 1>(a, lds_all);
}

// opt-in (row f-3): the split rules of the JSON pattern of Mistral's tekken.json
__global__ __launch_bounds__(TKF_BLOCK) TKF_OCC void tk_flat_json_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS];
    if (a.t.key_hash_mode == 0u) tk_flat_kernel_body<0, 0, 1>(a, lds_all);
    else tk_flat_kernel_body<0, 1, 1>(a, lds_all);
}

// the same kernels with the per-byte piece-start flags of tk_split_batch compiled in (and, in a `make ablate` build, TK_DEBUG_ABLATE)
__global__ __launch_bounds__(TKF_BLOCK) TKF_OCC void tk_flat_split_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS];
    if (a.t.key_hash_mode == 0u) tk_flat_kernel_body<1, 0>(a, lds_all);
    else tk_flat_kernel_body<1, 1>(a, lds_all);
}

// The chunks tk_flat_kernel left alone because they hold a piece of more than 64 bytes (a.cut_list): the same chunk work
// with the CUT step -- such pieces are cut into fragments wherever no vocabulary token can span the boundary
// (tk_flat_impl.h step 4b), so that a 32 KiB letter run becomes thousands of independent short merges instead of one chain
// and its document stays on the flat path.  Persistent waves over the list; the count never leaves the device.
template <int DBG>
__device__ __forceinline__ void tk_flat_cut_body(const TkFlatArgs& a, uint32_t* lds_all) {
    const uint32_t n = *a.cut_count;
    if (n == 0u) return;                                     // (grid-uniform)
    const int lane = wv_lane();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* lds = lds_all + wv * TKF_LDS_WORDS_CUT;
    tk_flat_init_lds(a, lds, lane);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKF_BLOCK / 64);
    for (uint64_t i = (uint64_t)blockIdx.x * (TKF_BLOCK / 64) + wv; i < n; i += n_waves) {
        const uint64_t c = (uint64_t)__builtin_amdgcn_readfirstlane((int)a.cut_list[i]);
        if (a.t.key_hash_mode == 0u) tk_flat_chunk<DBG, 0, 0, 1>(a, c, lane, lds);
        else tk_flat_chunk<DBG, 1, 0, 1>(a, c, lane, lds);
    }
    tk_flat_flush_memo_hits(a, lds, lane);
}
__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_cut_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS_CUT];
    tk_flat_cut_body<0>(a, lds_all);
}
__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_cut_split_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS_CUT];
    tk_flat_cut_body<1>(a, lds_all);
}

// wave w of tk_merge_kernel starts with item 64 w of the narrow classes, wave w of tk_merge_wide_kernel with item 64 w of
// the wide ones: note down which sub-queue holds it (thread e owns the waves whose first item falls into sub-queue e), so
// that the merge waves do not have to search the prefix sums
__global__ __launch_bounds__(TKF_BLOCK) void tk_merge_wavefirst_kernel(const uint64_t* __restrict__ prefix, uint64_t n_chunks,
                                                                        uint32_t* __restrict__ wave_first,
                                                                        uint32_t* __restrict__ wave_first_wide,
                                                                        uint32_t* __restrict__ narrow_left_out) {
    const uint64_t e = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    if (e >= 4 * n_chunks) return;
    if (e == 0 && narrow_left_out) {   // pieces of 2..16 bytes left to the merge kernel: the memo's misses of this call (the host's hit-rate policy)
        const uint64_t nl = prefix[2 * n_chunks];
        *narrow_left_out = nl > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nl;
    }
    const bool wide = e >= 2 * n_chunks;
    const uint64_t first = wide ? prefix[2 * n_chunks] : 0;
    const uint64_t lo = prefix[e] - first, hi = prefix[e + 1] - first;
    uint32_t* out = wide ? wave_first_wide : wave_first;
    for (uint64_t w = (lo + 63) / 64; w * 64 < hi; ++w) out[w] = (uint32_t)e;
}

// persistent waves: groups of 64 queued pieces strided over the grid (the host does not know how many there are).
// Every block keeps the PAIR filter (tk_hash.h) in LDS: a probe whose filter bit is clear issues no gather.
template <int THREADS, uint32_t WORDS = TK_PAIRF_WORDS>
TK_DEV void tk_merge_load_filter(const TkFlatArgs& a, uint32_t* filt) {
    const tk_u32x4* src = reinterpret_cast<const tk_u32x4*>(a.t.pair_filter);
    for (uint32_t i = threadIdx.x; i < WORDS / 4; i += THREADS) reinterpret_cast<tk_u32x4*>(filt)[i] = src[i];
    __syncthreads();
}

// LDS per block: the narrow classes keep the PAIR filter (32 KB) in front of every wave's columns (tk_merge_lds: 8 KB); the wide
// classes spend all of the LDS on columns (10 KB per wave in the compact layout): 16 waves without the filter beat 12 with it
// (mixed shape, same box: 2.47 against 2.63 ms; 8 waves in the old layout: 3.08).  One block per CU.
#ifndef TKM_BLOCK
#define TKM_BLOCK 1024       /* 16 waves (the largest block there is): 32 KB + 16 x 8 KB = 160 KB */
#endif
#ifndef TKM_WIDE_BLOCK
#define TKM_WIDE_BLOCK 1024  /* 16 waves x 10 KB = 160 KB */
#endif
#ifndef TKM_FILTER
#define TKM_FILTER 1
#endif
#define TKM_FWORDS (TKM_FILTER ? TK_PAIRF_WORDS : 0u)
#define TKM_LDS_BYTES ((TKM_FWORDS + (TKM_BLOCK / 64) * TKM_LDS_WORDS(16)) * 4)
#ifndef TKM_WIDE_FILTER
#define TKM_WIDE_FILTER 0
#endif
#define TKM_WIDE_FWORDS (TKM_WIDE_FILTER ? TK_PAIRF_WORDS : 0u)
#define TKM_WIDE_LDS_BYTES ((TKM_WIDE_FWORDS + (TKM_WIDE_BLOCK / 64) * TKM_LDS_WORDS(32)) * 4)
__global__ __launch_bounds__(TKM_BLOCK) void tk_merge_kernel(TkFlatArgs a) {   // pieces of 2..16 bytes
    extern __shared__ __attribute__((aligned(16))) uint32_t wlds[];
    if (TKM_FILTER) tk_merge_load_filter<TKM_BLOCK, TK_PAIRF_WORDS>(a, wlds);
    const uint64_t wave = (uint64_t)blockIdx.x * (TKM_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKM_BLOCK / 64);
    const uint64_t total = a.miss_prefix[2 * a.n_chunks];
    uint32_t* mlds = wlds + TKM_FWORDS + (threadIdx.x >> 6) * TKM_LDS_WORDS(16);
    TkMemoLog ml;
    ml.base = a.memo_log ? a.memo_log + wave * a.memo_log_per_wave : nullptr;
    ml.n = 0u;
    ml.cap = a.memo_log && wave < a.memo_log_waves ? a.memo_log_per_wave : 0u;
    for (uint64_t w = wave; w * 64 < total; w += n_waves) tk_merge_wave<false>(a, w, wv_lane(), mlds, TKM_FILTER ? wlds : nullptr, a.memo_log ? &ml : nullptr);
    if (a.memo_log && wave < a.memo_log_waves && wv_lane() == 0) a.memo_log_counts[wave] = ml.n;
}

// the log of the merge kernel's new memo entries into the table (tk_memo_commit_one; the slots were claimed where the records were written)
__global__ __launch_bounds__(TKF_BLOCK) void tk_memo_commit_kernel(tk_memo_entry* __restrict__ tab, const tk_memo_entry* __restrict__ log,
                                                                    const uint32_t* __restrict__ counts, uint32_t per_wave, uint32_t n_waves,
                                                                    uint32_t key_hash_mode, uint32_t mask) {
    const uint32_t n = per_wave * n_waves;
    for (uint32_t i = blockIdx.x * TKF_BLOCK + threadIdx.x; i < n; i += gridDim.x * TKF_BLOCK)
        if (tk_memo_log_live(counts, per_wave, i)) tk_memo_commit_one(tab, log, i, key_hash_mode, mask);
}

__global__ __launch_bounds__(TKM_WIDE_BLOCK) void tk_merge_wide_kernel(TkFlatArgs a) {   // pieces of 17..64 bytes
    extern __shared__ __attribute__((aligned(16))) uint32_t wlds[];
    if (TKM_WIDE_FILTER) tk_merge_load_filter<TKM_WIDE_BLOCK, TKM_WIDE_FWORDS>(a, wlds);
    const uint32_t* wfilt = TKM_WIDE_FILTER ? wlds : nullptr;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKM_WIDE_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKM_WIDE_BLOCK / 64);
    const uint64_t n2 = a.miss_prefix[3 * a.n_chunks] - a.miss_prefix[2 * a.n_chunks];   // pieces of 17..32 bytes: 64 per wave
    const uint64_t n3 = a.miss_prefix[4 * a.n_chunks] - a.miss_prefix[3 * a.n_chunks];   // pieces of 33..64 bytes: 64 per wave
    const uint64_t w2 = (n2 + 63) / 64, w3 = (n3 + 63) / 64;
    uint32_t* mlds = wlds + TKM_WIDE_FWORDS + (threadIdx.x >> 6) * TKM_LDS_WORDS(32);
    for (uint64_t w = wave; w < w2; w += n_waves) tk_merge_wave<true>(a, w, wv_lane(), mlds, wfilt);
    if (w3 == 0) return;                                     // (grid-uniform)
    // the class 33..64 bytes needs 64-entry columns: every second wave takes it, with its neighbour's LDS
    __syncthreads();
    if (((threadIdx.x >> 6) & 1u) == 0u) {
        const uint64_t ew = wave >> 1, n_ew = n_waves >> 1;
        for (uint64_t w = ew; w < w3; w += n_ew) tk_merge_wave_long3(a, w, wv_lane(), mlds, wfilt);
    }
}

// flagged documents -> list; the longest of them (it sizes the scratch of the piece-by-piece pass) -> *maxlen
__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_todo_kernel(const uint32_t* __restrict__ flags, const uint64_t* __restrict__ doc_offs,
                                                                  uint64_t n_docs, uint32_t* __restrict__ todo,
                                                                  uint32_t* __restrict__ n_todo, uint32_t* __restrict__ maxlen) {
    const uint64_t d = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    const bool f = d < n_docs && flags[d] != 0u;
    const uint64_t m = __ballot(f);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(n_todo, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, (int)__builtin_ctzll(m));
    if (f) {
        todo[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)d;
        const uint64_t len = doc_offs[d + 1] - doc_offs[d];
        atomicMax(maxlen, (uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len));
    }
}

// ids of the stream before byte doc_offs[i]
__device__ __forceinline__ uint64_t tkf_G(const uint64_t* doc_offs, uint64_t i, uint64_t n_bytes, uint64_t n_chunks,
                                          const uint64_t* P, const uint32_t* lstart) {
    const uint64_t p = doc_offs[i];
    if (p >= n_bytes) return P[n_chunks];
    return P[p / TKF_COMMIT] + lstart[i];
}

// per document: where its id slots start in the chunk-dense buffer, how many there are (holes included) and how many of
// them lie in the first chunk -- everything tk_flat_assemble_kernel needs in one 16-byte load
struct alignas(16) TkFlatDocInfo {
    uint64_t src;      // (chunk << 32) | first slot inside the chunk's row of tmp -- no division by the row stride in the
                       // assembly; for a handed-back document the index into the per-document kernel's staging
    uint32_t n_slots;  // slots to walk (handed-back document: ids to copy)
    uint32_t n_first;  // slots in the first chunk; bit 31: not eligible for the two-segment fast copy; 0xFFFFFFFF marks a
                       // handed-back document
};

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_counts_kernel(const uint64_t* __restrict__ doc_offs, uint64_t n_docs,
                                                                    uint64_t n_bytes, uint64_t n_chunks,
                                                                    const uint64_t* __restrict__ P,
                                                                    const uint32_t* __restrict__ lstart,
                                                                    const uint32_t* __restrict__ flags,
                                                                    const uint32_t* __restrict__ holes, uint32_t extra,
                                                                    uint32_t* __restrict__ counts,
                                                                    TkFlatDocInfo* __restrict__ info, int final_pass,
                                                                    uint32_t* __restrict__ n_flagged) {
    const uint64_t d = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    const bool flagged = d < n_docs && flags[d] != 0u;
    if (!final_pass) {
        // first (optimistic) pass: count the handed-back documents; the host redoes counts / scan / assembly after the
        // per-document kernels if there are any.  Until then they stand in as empty documents.
        const uint64_t m = __ballot(flagged);
        if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(n_flagged, (uint32_t)__builtin_popcountll(m));
    }
    if (d >= n_docs) return;
    TkFlatDocInfo di;
    if (flagged) {
        if (final_pass) {  // the document keeps the count of the per-document kernel, the assembly copies it from staging
            // (a document that a long-piece record flagged late has not been through those kernels yet and holds a stale count: no
            // document has more ids than bytes + 2, which keeps this pass inside the buffers; the host then redoes it)
            const uint64_t most = doc_offs[d + 1] - doc_offs[d] + 2;
            if ((uint64_t)counts[d] > most) counts[d] = (uint32_t)most;
            di.src = doc_offs[d] + 2 * d;
            di.n_slots = counts[d];
            di.n_first = 0xFFFFFFFFu;
        } else {
            // (counts[d] is left alone: the per-document kernels may be writing it right now, on the second stream -- whatever
            // this pass computes for a batch with flagged documents is thrown away)
            di.src = 0; di.n_slots = 0; di.n_first = 0;
        }
        info[d] = di;
        return;
    }
    const uint64_t g0 = tkf_G(doc_offs, d, n_bytes, n_chunks, P, lstart);
    const uint64_t g1 = tkf_G(doc_offs, d + 1, n_bytes, n_chunks, P, lstart);
    counts[d] = (uint32_t)(g1 - g0) - holes[d] + extra;
    const uint64_t p = doc_offs[d];
    di.src = 0;
    di.n_slots = (uint32_t)(g1 - g0);
    di.n_first = 0;
    if (p < n_bytes) {
        const uint64_t c = p / TKF_COMMIT;
        const uint64_t in_chunk = P[c + 1] - g0;  // slots of chunk c from the document start on
        di.src = (c << 32) | (uint64_t)lstart[d];
        di.n_first = (uint32_t)(in_chunk < (g1 - g0) ? in_chunk : (g1 - g0));
        // the assembly's prefetching copy takes documents of <= 128 slots that lie in at most two chunks
        const uint64_t rest = (g1 - g0) - di.n_first;
        if ((g1 - g0) > 128 || (rest && rest > P[c + 2] - P[c + 1])) di.n_first |= 0x80000000u;
    }
    info[d] = di;
}

struct TkFlatAssembleArgs {
    uint64_t n_docs;
    const TkFlatDocInfo* info;
    const uint32_t* kcount;
    const uint64_t* out_offs;
    const uint32_t* tmp;
    const uint32_t* staging;  // per-document kernel output (document d at doc_offs[d] + 2 d), flagged documents only
    uint32_t* out_ids;
    uint32_t bos_id, eos_id;
    int add_bos, add_eos;
    uint64_t* total_out;      // receives out_offs[n_docs] (the host reads it with the other counters)
    const uint32_t* skip_if;  // optimistic first pass (counters + 4): nothing is copied when skip_if[0] != 0 (documents were handed back)
                              // or skip_if[7] != 0 (long-piece records wait for tk_flat_long_kernel): the host
                              // runs the per-document kernels and assembles again); NULL for the final pass
};

// generic copy of one document (any number of chunks / slots, or a handed-back document)
__device__ __forceinline__ void tkf_assemble_doc(const TkFlatAssembleArgs& a, const TkFlatDocInfo& di0, uint32_t* dst, int lane) {
    TkFlatDocInfo di = di0;
    if (di.n_first == 0xFFFFFFFFu) {
        const uint32_t* src = a.staging + di.src;
        for (uint32_t k = (uint32_t)lane; k < di.n_slots; k += 64u) dst[k] = src[k];
        return;
    }
    di.n_first &= 0x7FFFFFFFu;
    if (a.add_bos) {
        if (lane == 0) dst[0] = a.bos_id;
        dst += 1;
    }
    // the document's slots: n_first in its first chunk, then whole chunks (slot 0 on) until n_slots are walked;
    // holes (slots a missed piece reserved and did not need) are skipped
    uint32_t left = di.n_slots, nn = di.n_first;
    uint64_t c = di.src >> 32;
    const uint32_t* src = a.tmp + c * TKF_STRIDE + (uint32_t)di.src;
    // four groups of 64 slots are requested together (a long document is a chain of load -> ballot -> store steps: one
    // group at a time leaves the wave waiting a memory round trip per 64 ids), and the next row's slot count with them
    const uint64_t below = (1ull << lane) - 1ull;
    while (left) {
        uint32_t kc_next = 0;
        if (left > nn) kc_next = a.kcount[c + 1];
        for (uint32_t k0 = 0; k0 < nn; k0 += 256u) {
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t k = k0 + 64u * (uint32_t)q + (uint32_t)lane;
                v[q] = k < nn ? src[k] : TKF_HOLE;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t keep = __ballot(v[q] != TKF_HOLE);
                if (v[q] != TKF_HOLE) dst[__builtin_popcountll(keep & below)] = v[q];
                dst += __builtin_popcountll(keep);
            }
        }
        left -= nn;
        if (left == 0) break;
        ++c;
        nn = left < kc_next ? left : kc_next;
        src = a.tmp + c * TKF_STRIDE;
    }
    if (a.add_eos && lane == 0) dst[0] = a.eos_id;
}

__device__ __forceinline__ uint32_t tkf_rl(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }

// One wave takes 64 consecutive documents: their 16-byte records and output offsets are fetched with one coalesced
// load each (lane = document).  The documents are then copied EIGHT at a time: the 16 loads of a group (slots 0..63
// and 64..127 of each document, across its chunk boundary) are issued back to back, so eight documents' worth of
// HBM latency overlap; then each is squeezed (holes out) and stored with all 64 lanes.
#ifndef TKA_GROUP
#define TKA_GROUP 8
#endif
__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_assemble_kernel(TkFlatAssembleArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKF_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKF_BLOCK / 64);
    if (wave == 0 && lane == 0) *a.total_out = a.out_offs[a.n_docs];
    if (a.skip_if && (a.skip_if[0] != 0u || a.skip_if[7] != 0u)) return;   // (grid-uniform; counters 4 and 11)
    for (uint64_t d0 = wave * 64; d0 < a.n_docs; d0 += n_waves * 64) {
        const uint64_t dm = d0 + (uint64_t)lane;
        TkFlatDocInfo mine;
        mine.src = 0; mine.n_slots = 0; mine.n_first = 0x80000000u;
        uint64_t oo = 0;
        if (dm < a.n_docs) {
            mine = a.info[dm];
            oo = a.out_offs[dm];
        }
        const int nd = (int)(a.n_docs - d0 < 64 ? a.n_docs - d0 : 64);
        const uint32_t src_lo = (uint32_t)mine.src, src_hi = (uint32_t)(mine.src >> 32);
        const uint32_t oo_lo = (uint32_t)oo, oo_hi = (uint32_t)(oo >> 32);
        for (int j0 = 0; j0 < nd; j0 += TKA_GROUP) {
            uint32_t v0[TKA_GROUP], v1[TKA_GROUP];
#pragma unroll
            for (int g = 0; g < TKA_GROUP; ++g) {
                const int j = j0 + g;                     // lanes beyond nd hold n_first = bit 31: skipped
                v0[g] = TKF_HOLE; v1[g] = TKF_HOLE;
                const uint32_t nf = tkf_rl(mine.n_first, j & 63);
                if (!(nf & 0x80000000u)) {                // wave-uniform
                    const uint32_t ns = tkf_rl(mine.n_slots, j & 63);
                    // (chunk, slot) -> the chunk's row of tmp (a scalar base) + a 32-bit slot offset per lane; slots past
                    // the document's n_first continue at slot 0 of the next row
                    const uint32_t* rowp = a.tmp + (uint64_t)tkf_rl(src_hi, j & 63) * TKF_STRIDE;
                    const uint32_t slot = tkf_rl(src_lo, j & 63);
                    const uint32_t q0 = (uint32_t)lane, q1 = 64u + (uint32_t)lane;
                    const uint32_t o0 = q0 < nf ? slot + q0 : (uint32_t)TKF_STRIDE + (q0 - nf);
                    const uint32_t o1 = q1 < nf ? slot + q1 : (uint32_t)TKF_STRIDE + (q1 - nf);
                    if (q0 < ns) v0[g] = rowp[o0];
                    if (q1 < ns) v1[g] = rowp[o1];
                }
            }
#pragma unroll
            for (int g = 0; g < TKA_GROUP; ++g) {
                const int j = j0 + g;
                if (j >= nd) break;
                const uint32_t nf = tkf_rl(mine.n_first, j);
                uint32_t* dst = a.out_ids + (((uint64_t)tkf_rl(oo_hi, j) << 32) | tkf_rl(oo_lo, j));
                if (nf & 0x80000000u) {
                    TkFlatDocInfo di;
                    di.src = ((uint64_t)tkf_rl(src_hi, j) << 32) | tkf_rl(src_lo, j);
                    di.n_slots = tkf_rl(mine.n_slots, j);
                    di.n_first = nf;
                    tkf_assemble_doc(a, di, dst, lane);
                    continue;
                }
                if (a.add_bos) {
                    if (lane == 0) dst[0] = a.bos_id;
                    dst += 1;
                }
                const uint32_t c0 = v0[g], c1 = v1[g];
                const uint64_t k0 = __ballot(c0 != TKF_HOLE), k1 = __ballot(c1 != TKF_HOLE);
                const uint64_t below = (1ull << lane) - 1ull;
                const uint32_t n0 = (uint32_t)__builtin_popcountll(k0);
                if (c0 != TKF_HOLE) dst[(uint32_t)__builtin_popcountll(k0 & below)] = c0;
                if (c1 != TKF_HOLE) dst[n0 + (uint32_t)__builtin_popcountll(k1 & below)] = c1;
                if (a.add_eos && lane == 0) dst[n0 + (uint32_t)__builtin_popcountll(k1)] = a.eos_id;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static uint32_t tkf_blocks(uint64_t n_threads) { return (uint32_t)((n_threads + TKF_BLOCK - 1) / TKF_BLOCK); }

hipError_t tk_launch_flat_firstdoc(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_chunks, uint32_t* first_doc,
                                   uint32_t* flags, uint32_t* holes, uint32_t* counters16, hipStream_t s) {
    // (always launched: it also clears flags / holes [n_docs + 1] and the counters)
    hipLaunchKernelGGL(tk_flat_firstdoc_kernel, dim3(tkf_blocks(n_docs + 32)), dim3(TKF_BLOCK), 0, s, doc_offs, n_docs,
                       n_chunks, first_doc, flags, holes, counters16);
    return hipGetLastError();
}

hipError_t tk_launch_flat(const TkFlatArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return hipSuccess;
    uint64_t blocks = (a.n_chunks + (TKF_BLOCK / 64) - 1) / (TKF_BLOCK / 64);
    // persistent waves, chunks strided over them: exactly the blocks that are resident at once, so that no partially
    // filled last round of blocks trails behind
    // (per device, like the merge grids below: a process may hold contexts on several GPUs)
    static uint64_t cap_dev[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint64_t& cap = cap_dev[dev & 63];
    if (cap == 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int per_cu = 0;   // resident blocks per CU: 8 (16 bytes per lane, 8 waves per SIMD) or 7 (32 bytes per lane: LDS)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tk_flat_kernel, TKF_BLOCK, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
        cap = (uint64_t)cus * (uint64_t)per_cu;
        if (const char* e = getenv("TK_FLAT_BLOCKS")) cap = (uint64_t)atoll(e);
        if (cap == 0) cap = 1280;
    }
    if (blocks > cap) blocks = cap;
    if (a.pattern == 1) hipLaunchKernelGGL(tk_flat_json_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    else if (a.dbg_ablate || a.dbg_starts) hipLaunchKernelGGL(tk_flat_split_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    else if (a.t.key_hash_mode == 0u && a.memo_tab && a.memo_probe) hipLaunchKernelGGL(tk_flat_memo_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    else if (a.t.key_hash_mode == 0u) hipLaunchKernelGGL(tk_flat_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(tk_flat_mode1_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    if (a.cut_list && a.pattern == 0) {
        // the chunks with a piece of more than 64 bytes (none on ordinary text: the blocks read a zero and leave -- two blocks
        // per CU, not the seven of the flat kernel: dispatching 1 792 blocks that do nothing took 10 us of every C2 step; the
        // 14 k listed regions of the 4 M-document Zipf shape are seven per wave instead of two)
        const uint64_t cblocks = blocks > cap / 7 * 2 && cap >= 7 ? cap / 7 * 2 : blocks;
        if (a.dbg_ablate || a.dbg_starts) hipLaunchKernelGGL(tk_flat_cut_split_kernel, dim3((uint32_t)cblocks), dim3(TKF_BLOCK), 0, s, a);
        else hipLaunchKernelGGL(tk_flat_cut_kernel, dim3((uint32_t)cblocks), dim3(TKF_BLOCK), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t tk_launch_merge(const TkFlatArgs& a, uint32_t* narrow_left_out, hipStream_t s) {
    if (a.n_chunks == 0) return hipSuccess;
    // persistent grids (the number of queued pieces stays on the device): the blocks that are resident at once -- each
    // one copies the PAIR filter into its LDS first -- and never more than the sub-queues could fill
    // (per device: a process may hold contexts on several GPUs, and the LDS opt-in is a per-device function attribute;
    // two threads racing through the first call on a device compute the same values)
    static uint64_t res1_dev[64] = {0}, res2_dev[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint64_t& res1 = res1_dev[dev & 63];
    uint64_t& res2 = res2_dev[dev & 63];
    if (res1 == 0) {
        int cus = 256, p1 = 0, p2 = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tk_merge_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TKM_WIDE_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(tk_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TKM_LDS_BYTES) != hipSuccess)
            return hipErrorInvalidValue;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p1, tk_merge_kernel, TKM_BLOCK, TKM_LDS_BYTES) != hipSuccess || p1 <= 0) p1 = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, tk_merge_wide_kernel, TKM_WIDE_BLOCK, TKM_WIDE_LDS_BYTES) != hipSuccess || p2 <= 0) p2 = 1;
        res2 = (uint64_t)cus * (uint64_t)p2;
        res1 = (uint64_t)cus * (uint64_t)p1;
    }
    uint64_t b1 = (a.n_chunks * 4 + TKM_BLOCK / 64 - 1) / (TKM_BLOCK / 64), b2 = (a.n_chunks + TKM_WIDE_BLOCK / 64 - 1) / (TKM_WIDE_BLOCK / 64);
    if (b1 > res1) b1 = res1;
    if (b2 > res2) b2 = res2;
    hipLaunchKernelGGL(tk_merge_wavefirst_kernel, dim3(tkf_blocks(4 * a.n_chunks)), dim3(TKF_BLOCK), 0, s, a.miss_prefix,
                       a.n_chunks, a.wave_first, a.wave_first_wide, narrow_left_out);
    // (the memo log is cut into one stretch per wave of THIS grid; a grid with more waves than the log was sized for logs nothing
    // from the surplus waves)
    hipLaunchKernelGGL(tk_merge_kernel, dim3((uint32_t)b1), dim3(TKM_BLOCK), TKM_LDS_BYTES, s, a);
    hipLaunchKernelGGL(tk_merge_wide_kernel, dim3((uint32_t)b2), dim3(TKM_WIDE_BLOCK), TKM_WIDE_LDS_BYTES, s, a);
    if (a.memo_tab && a.memo_log) {
        hipLaunchKernelGGL(tk_memo_commit_kernel, dim3(1024), dim3(TKF_BLOCK), 0, s, a.memo_tab, a.memo_log, a.memo_log_counts, a.memo_log_per_wave, a.memo_log_waves, a.t.key_hash_mode, a.memo_mask);
    }
    return hipGetLastError();
}

// Pieces of 65..TKF_LONGCAP bytes (tk_flat_impl.h step 6): one wave per record -- the end of the piece where the chunk did
// not see it (sequential matcher), whole-piece lookup, the single-wave merge; ids + holes into the slots the chunk
// reserved, the document's hole count like the merge kernels.  Launched only when the flat kernel wrote records.
__global__ __launch_bounds__(256) void tk_flat_long_kernel(TkFlatArgs a, uint32_t* work_counter, uint32_t* scratch, uint32_t scratch_words) {
    const int lane = wv_lane();
    const TkPolyPow pw = tk_poly_pow(a.t, lane);
    const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t* my = scratch + wave_id * scratch_words;
    const uint32_t n = *a.long_count < a.long_cap ? *a.long_count : a.long_cap;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    (void)work_counter;                                      // (records strided over the waves: a ticket per record is an atomic per record)
    for (uint64_t q = wave_id; q < n; q += n_waves) tk_flat_long_wave(a, pw, (uint32_t)q, lane, my);
}

// stage two for the records of 65..128 bytes that are no vocabulary keys: one lane per piece (tk_merge_long_wave<128>), three
// waves per block (32 KB of PAIR filter + 3 x 40 KB of columns in the compact layout), one block per CU.  (The same with 256-entry columns for
// 129..256 bytes -- 128 KB per wave, ONE wave per CU -- was measured and dropped: 4.4 ms where the single-wave merge of stage
// one takes 4.7 ms for the same pieces; 64 chains per CU at ~3 us a round are no better than a dozen at 1.1 us a merge.)
#define TKM_L128_BLOCK 192
#define TKM_L128_LDS_BYTES ((TK_PAIRF_WORDS + (TKM_L128_BLOCK / 64) * TKM_LDS_WORDS(128)) * 4)
template <int N, int BLOCK>
__device__ __forceinline__ void tk_flat_longN_body(const TkFlatArgs& a, uint32_t* wlds) {
    const uint64_t n = *a.long_count < a.long_cap ? *a.long_count : a.long_cap;
    if (n == 0) return;                                      // (grid-uniform)
    tk_merge_load_filter<BLOCK>(a, wlds);
    const uint64_t wave = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (BLOCK / 64);
    uint32_t* mlds = wlds + TK_PAIRF_WORDS + (threadIdx.x >> 6) * TKM_LDS_WORDS(N);
    for (uint64_t w = wave; w * 64 < n; w += n_waves) tk_merge_long_wave<N>(a, w, wv_lane(), mlds, wlds);
}
__global__ __launch_bounds__(TKM_L128_BLOCK) void tk_flat_long128_kernel(TkFlatArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t wlds[];
    tk_flat_longN_body<128, TKM_L128_BLOCK>(a, wlds);
}

// stage two for the marked records of 129..TKF_LONGCAP bytes: the single-wave merge alone, records strided over the waves
__global__ __launch_bounds__(256) void tk_flat_long_coop_kernel(TkFlatArgs a, uint32_t* scratch, uint32_t scratch_words) {
    const int lane = wv_lane();
    const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t* my = scratch + wave_id * scratch_words;
    const uint32_t n = *a.long_count < a.long_cap ? *a.long_count : a.long_cap;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t q = wave_id; q < n; q += n_waves) tk_flat_long_coop_wave(a, (uint32_t)q, lane, my);
}

hipError_t tk_launch_flat_long(const TkFlatArgs& a, uint32_t* work_counter, uint32_t* scratch, uint32_t scratch_words, uint32_t n_waves,
                               hipStream_t s) {
    hipLaunchKernelGGL(tk_flat_long_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, s, a, work_counter, scratch, scratch_words);
    if (a.long_merge128) {
        static bool attr_set[64] = {false};        // (per device: the LDS opt-in is a per-device function attribute)
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        if (!attr_set[dev & 63]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(tk_flat_long128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    TKM_L128_LDS_BYTES) != hipSuccess)
                return hipErrorInvalidValue;
            attr_set[dev & 63] = true;
        }
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        hipLaunchKernelGGL(tk_flat_long128_kernel, dim3((uint32_t)cus), dim3(TKM_L128_BLOCK), TKM_L128_LDS_BYTES, s, a);
        hipLaunchKernelGGL(tk_flat_long_coop_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, s, a, scratch, scratch_words);
    }
    return hipGetLastError();
}

hipError_t tk_launch_flat_todo(const uint32_t* flags, const uint64_t* doc_offs, uint64_t n_docs, uint32_t* todo, uint32_t* n_todo,
                               uint32_t* maxlen, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_flat_todo_kernel, dim3(tkf_blocks(n_docs)), dim3(TKF_BLOCK), 0, s, flags, doc_offs, n_docs, todo, n_todo, maxlen);
    return hipGetLastError();
}

hipError_t tk_launch_flat_counts(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_bytes, uint64_t n_chunks,
                                 const uint64_t* P, const uint32_t* lstart, const uint32_t* flags, const uint32_t* holes,
                                 uint32_t extra, uint32_t* counts, void* doc_info, int final_pass, uint32_t* n_flagged, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_flat_counts_kernel, dim3(tkf_blocks(n_docs)), dim3(TKF_BLOCK), 0, s, doc_offs, n_docs, n_bytes,
                       n_chunks, P, lstart, flags, holes, extra, counts, (TkFlatDocInfo*)doc_info, final_pass, n_flagged);
    return hipGetLastError();
}

hipError_t tk_launch_flat_assemble(uint64_t n_docs, const void* doc_info, const uint32_t* kcount, const uint64_t* out_offs,
                                   const uint32_t* tmp, const uint32_t* staging, uint32_t* out_ids, uint32_t bos_id,
                                   uint32_t eos_id, int add_bos, int add_eos, uint64_t* total_out, const uint32_t* skip_if, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    TkFlatAssembleArgs a;
    a.total_out = total_out;
    a.skip_if = skip_if;
    a.n_docs = n_docs; a.info = (const TkFlatDocInfo*)doc_info; a.kcount = kcount; a.out_offs = out_offs;
    a.tmp = tmp; a.staging = staging; a.out_ids = out_ids;
    a.bos_id = bos_id; a.eos_id = eos_id; a.add_bos = add_bos; a.add_eos = add_eos;
    uint64_t blocks = ((n_docs + 63) / 64 + (TKF_BLOCK / 64) - 1) / (TKF_BLOCK / 64);   // 64 documents per wave
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(tk_flat_assemble_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    return hipGetLastError();
}
