// tk_flat.hip -- gfx950 kernels of the flat (chunk-per-wave) tokenization path, see tk_flat_impl.h.
//
//   tk_flat_firstdoc_kernel   per chunk: how many documents start below its loaded region
//   tk_flat_kernel            split + lookup + merge of one 1024-byte region per wave, ids chunk-dense
//   tk_merge_kernel           byte-pair merge of the queued pieces that missed the vocabulary, one lane per piece
//   tk_flat_todo_kernel       flagged documents -> list for the per-document kernel
//   tk_flat_counts_kernel     ids per document from the chunk prefix sums and the document-start ranks
//   tk_flat_assemble_kernel   chunk-dense ids -> packed ids in document order with BOS / EOS
//                             (reference src/tekkenizer.rs:390-402)
//
// Integer / byte work, no MFMA; bound: HBM (DESIGN.md has the byte accounting).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tk_kernels.h"
#include "tk_wave_hip.h"
#include "tk_flat_impl.h"

#define TKF_BLOCK 256

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_firstdoc_kernel(const uint64_t* __restrict__ doc_offs, uint64_t n_docs,
                                                                      uint64_t n_chunks, uint32_t* __restrict__ first_doc) {
    // first_doc[c] = number of documents d with doc_offs[d] < lo(c), lo(c) = max(c * COMMIT - HL, 0);
    // document d owns the chunks whose lo lies in (doc_offs[d], doc_offs[d + 1]]  (the last document: everything above)
    const uint64_t d = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    if (d == 0 && n_chunks) first_doc[0] = 0u;
    if (d >= n_docs) return;
    const uint64_t s = doc_offs[d], e = doc_offs[d + 1];
    const uint64_t c_lo = (s + TKF_HL) / TKF_COMMIT + 1;
    uint64_t c_hi = d + 1 == n_docs ? n_chunks - 1 : (e + TKF_HL) / TKF_COMMIT;
    if (n_chunks == 0) return;
    if (c_hi > n_chunks - 1) c_hi = n_chunks - 1;
    for (uint64_t c = c_lo; c <= c_hi; ++c) first_doc[c] = (uint32_t)(d + 1);
}

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_kernel(TkFlatArgs a) {
    __shared__ uint32_t lds_all[(TKF_BLOCK / 64) * TKF_LDS_WORDS];
    const int lane = wv_lane();
    uint32_t* lds = lds_all + (threadIdx.x >> 6) * TKF_LDS_WORDS;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKF_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKF_BLOCK / 64);
    TkPolyPow pw;
    pw.pw1 = pw.ipw1 = pw.pw2 = pw.ipw2 = 1u;
    for (uint64_t c = wave; c < a.n_chunks; c += n_waves) tk_flat_chunk(a, c, lane, lds, pw);
}

__global__ __launch_bounds__(TKF_BLOCK) void tk_merge_kernel(TkFlatArgs a) {
    const uint64_t wave = (uint64_t)blockIdx.x * (TKF_BLOCK / 64) + (threadIdx.x >> 6);
    tk_merge_wave(a, wave, wv_lane());
}

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_todo_kernel(const uint32_t* __restrict__ flags, uint64_t n_docs,
                                                                  uint32_t* __restrict__ todo, uint32_t* __restrict__ n_todo) {
    const uint64_t d = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    const bool f = d < n_docs && flags[d] != 0u;
    const uint64_t m = __ballot(f);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(n_todo, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, (int)__builtin_ctzll(m));
    if (f) todo[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)d;
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
    uint64_t src;      // index into tmp (or into the per-document kernel's staging for a handed-back document)
    uint32_t n_slots;  // slots to walk (handed-back document: ids to copy)
    uint32_t n_first;  // slots in the first chunk; 0xFFFFFFFF marks a handed-back document
};

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_counts_kernel(const uint64_t* __restrict__ doc_offs, uint64_t n_docs,
                                                                    uint64_t n_bytes, uint64_t n_chunks,
                                                                    const uint64_t* __restrict__ P,
                                                                    const uint32_t* __restrict__ lstart,
                                                                    const uint32_t* __restrict__ flags,
                                                                    const uint32_t* __restrict__ holes, uint32_t extra,
                                                                    uint32_t* __restrict__ counts,
                                                                    TkFlatDocInfo* __restrict__ info) {
    const uint64_t d = (uint64_t)blockIdx.x * TKF_BLOCK + threadIdx.x;
    if (d >= n_docs) return;
    TkFlatDocInfo di;
    if (flags[d]) {  // a flagged document keeps the count of the per-document kernel
        di.src = doc_offs[d] + 2 * d;
        di.n_slots = counts[d];
        di.n_first = 0xFFFFFFFFu;
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
        di.src = c * TKF_STRIDE + lstart[d];
        di.n_first = (uint32_t)(in_chunk < (g1 - g0) ? in_chunk : (g1 - g0));
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
};

__global__ __launch_bounds__(TKF_BLOCK) void tk_flat_assemble_kernel(TkFlatAssembleArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKF_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKF_BLOCK / 64);
    for (uint64_t d = wave; d < a.n_docs; d += n_waves) {
        const TkFlatDocInfo di = a.info[d];
        uint32_t* dst = a.out_ids + a.out_offs[d];
        if (di.n_first == 0xFFFFFFFFu) {
            const uint32_t* src = a.staging + di.src;
            for (uint32_t k = (uint32_t)lane; k < di.n_slots; k += 64u) dst[k] = src[k];
            continue;
        }
        if (a.add_bos) {
            if (lane == 0) dst[0] = a.bos_id;
            dst += 1;
        }
        // the document's slots: n_first in its first chunk, then whole chunks (slot 0 on) until n_slots are walked;
        // holes (slots a missed piece reserved and did not need) are skipped
        uint32_t left = di.n_slots, nn = di.n_first;
        uint64_t c = di.src / TKF_STRIDE;
        const uint32_t* src = a.tmp + di.src;
        while (left) {
            for (uint32_t k0 = 0; k0 < nn; k0 += 64u) {
                const uint32_t k = k0 + (uint32_t)lane;
                const uint32_t v = k < nn ? src[k] : TKF_HOLE;
                const uint64_t keep = __ballot(v != TKF_HOLE);
                if (v != TKF_HOLE) dst[__builtin_popcountll(keep & ((1ull << lane) - 1ull))] = v;
                dst += __builtin_popcountll(keep);
            }
            left -= nn;
            if (left == 0) break;
            ++c;
            const uint32_t kc = a.kcount[c];
            nn = left < kc ? left : kc;
            src = a.tmp + c * TKF_STRIDE;
        }
        if (a.add_eos && lane == 0) dst[0] = a.eos_id;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static uint32_t tkf_blocks(uint64_t n_threads) { return (uint32_t)((n_threads + TKF_BLOCK - 1) / TKF_BLOCK); }

hipError_t tk_launch_flat_firstdoc(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_chunks, uint32_t* first_doc,
                                   hipStream_t s) {
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_flat_firstdoc_kernel, dim3(tkf_blocks(n_docs ? n_docs : 1)), dim3(TKF_BLOCK), 0, s, doc_offs, n_docs,
                       n_chunks, first_doc);
    return hipGetLastError();
}

hipError_t tk_launch_flat(const TkFlatArgs& a, hipStream_t s) {
    if (a.n_chunks == 0) return hipSuccess;
    uint64_t blocks = (a.n_chunks + (TKF_BLOCK / 64) - 1) / (TKF_BLOCK / 64);
    const uint64_t cap = 256ull * 16ull;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(tk_flat_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_merge(const TkFlatArgs& a, uint64_t n_miss, hipStream_t s) {
    if (n_miss == 0) return hipSuccess;
    const uint64_t waves = (n_miss + 63) / 64;
    hipLaunchKernelGGL(tk_merge_kernel, dim3((uint32_t)((waves + (TKF_BLOCK / 64) - 1) / (TKF_BLOCK / 64))), dim3(TKF_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_flat_todo(const uint32_t* flags, uint64_t n_docs, uint32_t* todo, uint32_t* n_todo, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_flat_todo_kernel, dim3(tkf_blocks(n_docs)), dim3(TKF_BLOCK), 0, s, flags, n_docs, todo, n_todo);
    return hipGetLastError();
}

hipError_t tk_launch_flat_counts(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_bytes, uint64_t n_chunks,
                                 const uint64_t* P, const uint32_t* lstart, const uint32_t* flags, const uint32_t* holes,
                                 uint32_t extra, uint32_t* counts, void* doc_info, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_flat_counts_kernel, dim3(tkf_blocks(n_docs)), dim3(TKF_BLOCK), 0, s, doc_offs, n_docs, n_bytes,
                       n_chunks, P, lstart, flags, holes, extra, counts, (TkFlatDocInfo*)doc_info);
    return hipGetLastError();
}

hipError_t tk_launch_flat_assemble(uint64_t n_docs, const void* doc_info, const uint32_t* kcount, const uint64_t* out_offs,
                                   const uint32_t* tmp, const uint32_t* staging, uint32_t* out_ids, uint32_t bos_id,
                                   uint32_t eos_id, int add_bos, int add_eos, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    TkFlatAssembleArgs a;
    a.n_docs = n_docs; a.info = (const TkFlatDocInfo*)doc_info; a.kcount = kcount; a.out_offs = out_offs;
    a.tmp = tmp; a.staging = staging; a.out_ids = out_ids;
    a.bos_id = bos_id; a.eos_id = eos_id; a.add_bos = add_bos; a.add_eos = add_eos;
    uint64_t blocks = (n_docs + (TKF_BLOCK / 64) - 1) / (TKF_BLOCK / 64);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(tk_flat_assemble_kernel, dim3((uint32_t)blocks), dim3(TKF_BLOCK), 0, s, a);
    return hipGetLastError();
}
