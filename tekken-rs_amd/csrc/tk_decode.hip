// tk_decode.hip -- gfx950 kernels of the batch DECODE path.
//
// Reference: Tekkenizer::decode / decode_all / decode_group (src/tekkenizer.rs:436-560):
//   ids are grouped into maximal runs of special (id < num_special_tokens) / non-special ids;
//   a special run is dropped (Ignore), replaced by the token strings (Keep) or an error (Raise);
//   a non-special run is the concatenation of its token bytes and must be valid UTF-8 ON ITS OWN
//   (CoreBPE::decode -> String::from_utf8, :552-555); the document text is the join of the runs.
//
// GPU formulation (byte/index work, HBM-bound):
//   tk_decode_len_kernel     one thread per id: byte length of what the id contributes + error flags
//   (tk_scan_*)              exclusive scan of the lengths -> byte offset of every id (u64)
//   tk_decode_copy_kernel    one lane per id: token bytes are scattered into an LDS image of the workgroup's
//                            output span and streamed out with aligned coalesced dword stores; ids that
//                            start a run mark a bit in a run-start bitmap
//   tk_decode_docoffs_kernel out_offsets[d] = byte offset of the document's first id
//   tk_decode_validate_kernel one lane per output byte: UTF-8 well-formedness where run starts and
//                            document boundaries are hard boundaries (a code point may not span them)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tk_kernels.h"

#define TKD_BLOCK 256

// error word layout: [0] first id index with a Raise-policy special token (or ~0), [1] first id index
// that is out of the vocabulary, [2] first document with an invalid UTF-8 run
__global__ __launch_bounds__(TKD_BLOCK) void tk_decode_len_kernel(TkDecodeArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * TKD_BLOCK + threadIdx.x;
    if (i >= a.n_ids) return;
    const uint32_t id = a.ids[i];
    uint32_t len = 0;
    if (id < a.num_special) {
        if (a.policy == TK_POLICY_RAISE) atomicMin(a.err + 0, (unsigned long long)i);
        else if (a.policy == TK_POLICY_KEEP) len = a.sp_offs[id + 1] - a.sp_offs[id];
    } else {
        const uint32_t r = id - a.num_special;
        if (r >= a.n_ranks) atomicMin(a.err + 1, (unsigned long long)i);
        else len = a.tok_offs[r + 1] - a.tok_offs[r];
    }
    a.lens[i] = len;
}

// One workgroup = 256 consecutive ids = one contiguous span of output bytes (about 1 KB).  Lanes scatter
// their token bytes into an LDS image of that span (byte writes stay on chip), then the workgroup
// streams the image to HBM with aligned, fully coalesced dword stores; only the first and last partial
// dword of the span are written byte-wise (their other bytes belong to the neighbouring workgroups).
#define TKD_LDS_BYTES 8192u
__global__ __launch_bounds__(TKD_BLOCK) void tk_decode_copy_kernel(TkDecodeArgs a) {
    __shared__ uint32_t img[TKD_LDS_BYTES / 4 + 2];
    const uint64_t i0 = (uint64_t)blockIdx.x * TKD_BLOCK;
    const uint64_t i = i0 + threadIdx.x;
    const uint64_t i1 = i0 + TKD_BLOCK < a.n_ids ? i0 + TKD_BLOCK : a.n_ids;
    const uint64_t base = a.boff[i0], end = a.boff[i1];  // block-uniform
    const uint8_t* src = nullptr;
    uint32_t len = 0;
    uint64_t dst = 0;
    if (i < a.n_ids) {
        const uint32_t id = a.ids[i];
        dst = a.boff[i];
        const bool special = id < a.num_special;
        // hard boundary for UTF-8 validation: the first id of every run.  A conservative superset is marked:
        // every special id and every id that follows a special one (document starts are boundaries anyway).
        const bool prev_special = i > 0 && a.ids[i - 1] < a.num_special;
        if (special || prev_special) atomicOr(a.run_bits + (dst >> 5), 1u << (dst & 31));
        if (special) {
            if (a.policy == TK_POLICY_KEEP) { src = a.sp_blob + a.sp_offs[id]; len = a.sp_offs[id + 1] - a.sp_offs[id]; }
        } else {
            const uint32_t r = id - a.num_special;
            if (r < a.n_ranks) { src = a.tok_blob + a.tok_offs[r]; len = a.tok_offs[r + 1] - a.tok_offs[r]; }
        }
    }
    const uint64_t g0 = base & ~3ull;  // the image starts at an aligned global address
    if (end - g0 <= TKD_LDS_BYTES) {
        uint8_t* img8 = reinterpret_cast<uint8_t*>(img);
        const uint32_t off = (uint32_t)(dst - g0);
        for (uint32_t k = 0; k < len; ++k) img8[off + k] = src[k];
        __syncthreads();
        const uint32_t nwords = (uint32_t)((end - g0 + 3) / 4);
        for (uint32_t w = threadIdx.x; w < nwords; w += TKD_BLOCK) {
            const uint64_t ga = g0 + 4ull * w;
            if (ga >= base && ga + 4 <= end) {
                *reinterpret_cast<uint32_t*>(a.out_bytes + ga) = img[w];
            } else {
                const uint64_t lo = ga > base ? ga : base, hi = ga + 4 < end ? ga + 4 : end;
                for (uint64_t q = lo; q < hi; ++q) a.out_bytes[q] = img8[q - g0];
            }
        }
    } else {
        uint8_t* out = a.out_bytes + dst;  // unusually long tokens: direct byte stores
        for (uint32_t k = 0; k < len; ++k) out[k] = src[k];
    }
}

__global__ __launch_bounds__(TKD_BLOCK) void tk_decode_docoffs_kernel(TkDecodeArgs a) {
    const uint64_t d = (uint64_t)blockIdx.x * TKD_BLOCK + threadIdx.x;
    if (d > a.n_docs) return;
    a.out_offs[d] = a.boff[a.id_offs[d]];  // boff has n_ids + 1 entries; id_offs[n_docs] == n_ids
}

__device__ __forceinline__ bool tkd_is_boundary(const TkDecodeArgs& a, uint64_t p) {
    return (a.run_bits[p >> 5] >> (p & 31)) & 1u;
}

// One wave per document, one lane per byte.  Same checks as tk_validate_kernel (RFC 3629) with the
// extra rule that positions flagged in run_bits cut sequences.
__global__ __launch_bounds__(TKD_BLOCK) void tk_decode_validate_kernel(TkDecodeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKD_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKD_BLOCK / 64);
    const uint8_t* b = a.out_bytes;
    for (uint64_t d = wave; d < a.n_docs; d += n_waves) {
        const uint64_t s0 = a.out_offs[d], s1 = a.out_offs[d + 1];
        bool err = false;
        for (uint64_t p = s0 + (uint64_t)lane; p < s1; p += 64) {
            const uint32_t b0 = b[p];
            if (b0 < 0x80u) continue;
            if ((b0 & 0xC0u) == 0x80u) {
                // continuation: a lead that covers it must sit within 3 bytes, with no boundary in (lead, p]
                bool ok = false;
                for (uint32_t k = 1; k <= 3 && p >= s0 + k; ++k) {
                    if (tkd_is_boundary(a, p - k + 1)) break;
                    const uint32_t q = b[p - k];
                    if ((q & 0xC0u) == 0x80u) continue;
                    const uint32_t need = q >= 0xF0u ? 3u : q >= 0xE0u ? 2u : q >= 0xC0u ? 1u : 0u;
                    ok = need >= k;
                    break;
                }
                if (!ok) err = true;
                continue;
            }
            const uint32_t need = b0 >= 0xF8u ? 99u : b0 >= 0xF0u ? 3u : b0 >= 0xE0u ? 2u : b0 >= 0xC2u ? 1u : 99u;
            if (need == 99u || p + need >= s1) { err = true; continue; }
            bool cut = false;
            for (uint32_t k = 1; k <= need; ++k) cut |= tkd_is_boundary(a, p + k) || (b[p + k] & 0xC0u) != 0x80u;
            if (cut) { err = true; continue; }
            const uint32_t b1 = b[p + 1];
            if (b0 == 0xE0u && b1 < 0xA0u) err = true;   // overlong 3-byte
            if (b0 == 0xEDu && b1 >= 0xA0u) err = true;  // surrogates
            if (b0 == 0xF0u && b1 < 0x90u) err = true;   // overlong 4-byte
            if (b0 == 0xF4u && b1 >= 0x90u) err = true;  // > U+10FFFF
            if (b0 > 0xF4u) err = true;
        }
        if (__ballot(err) && lane == 0) atomicMin(a.err + 2, (unsigned long long)d);
    }
}

hipError_t tk_launch_decode_len(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_ids == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_decode_len_kernel, dim3((uint32_t)((a.n_ids + TKD_BLOCK - 1) / TKD_BLOCK)), dim3(TKD_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_decode_copy(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_ids) {
        hipLaunchKernelGGL(tk_decode_copy_kernel, dim3((uint32_t)((a.n_ids + TKD_BLOCK - 1) / TKD_BLOCK)), dim3(TKD_BLOCK), 0, s, a);
    }
    hipLaunchKernelGGL(tk_decode_docoffs_kernel, dim3((uint32_t)((a.n_docs + 1 + TKD_BLOCK - 1) / TKD_BLOCK)), dim3(TKD_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_decode_validate(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_docs == 0) return hipSuccess;
    uint64_t blocks = (a.n_docs + (TKD_BLOCK / 64) - 1) / (TKD_BLOCK / 64);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tk_decode_validate_kernel, dim3((uint32_t)blocks), dim3(TKD_BLOCK), 0, s, a);
    return hipGetLastError();
}
