// tk_decode.hip -- gfx950 kernels of the batch DECODE path.
//
// Reference: Tekkenizer::decode / decode_all / decode_group (src/tekkenizer.rs:436-560):
//   ids are grouped into maximal runs of special (id < num_special_tokens) / non-special ids;
//   a special run is dropped (Ignore), replaced by the token strings (Keep) or an error (Raise);
//   a non-special run is the concatenation of its token bytes and must be valid UTF-8 ON ITS OWN
//   (CoreBPE::decode -> String::from_utf8, :552-555); the document text is the join of the runs.
//
// GPU formulation (byte/index work, HBM-bound):
//   tk_decode_doclen_kernel  one wave per document, one lane per id: text length of the document + error flags
//   (tk_scan_*)              exclusive scan of the document lengths -> out_offsets (u64)
//   tk_decode_emit_kernel    one wave per 16 consecutive documents, their ids as one stream, 64 at a time: in-register prefix sum, token bytes
//                            scattered into a per-wave LDS image of the chunk's span, streamed out with
//                            aligned coalesced dword stores; ids that start a run mark a bit in a bitmap
//   tk_decode_validate_kernel one lane per output byte: UTF-8 well-formedness where run starts and
//                            document boundaries are hard boundaries (a code point may not span them)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tk_kernels.h"
#include "tk_utf8_swar.h"

#define TKD_BLOCK 256

// error word layout: [0] first id index with a Raise-policy special token (or ~0), [1] first id index
// that is out of the vocabulary, [2] first document with an invalid UTF-8 run

// what one id contributes: source pointer + length (0 for ignored / erroneous ids)
__device__ __forceinline__ uint32_t tkd_piece(const TkDecodeArgs& a, uint32_t id, const uint8_t** src) {
    if (id < a.num_special) {
        if (a.policy != TK_POLICY_KEEP) return 0u;
        *src = a.sp_blob + a.sp_offs[id];
        return a.sp_offs[id + 1] - a.sp_offs[id];
    }
    const uint32_t r = id - a.num_special;
    if (r >= a.n_ranks) return 0u;
    // offs[r] and offs[r + 1] with ONE 8-byte load (4-byte aligned): scattered loads are what this path pays for
    typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) u32x2_a4;
    const u32x2_a4 o = *reinterpret_cast<const u32x2_a4*>(a.tok_offs + r);
    *src = a.tok_blob + o.x;
    return o.y - o.x;
}

__device__ __forceinline__ uint32_t tkd_wave_sum(uint32_t v) {
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts: VALU only; six ds_bpermute round trips of
// __shfl_up were a third of a step's latency)
__device__ __forceinline__ uint32_t tkd_scan_incl(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

#define TKD_DOCS TK_DECODE_GROUP_DOCS          /* consecutive documents a wave takes as one stream of ids */
#define TKD_L8_LDS 32768u     /* one-byte lengths of the ranks below this live in LDS (99 % of the ids of running text) */

// pass A: the text length of every document + error flags: one wave per document, one lane per id; the length of a token
// comes from an LDS copy of the one-byte length table (a gather from LDS costs a twelfth of one from the L2, and the
// gathers were all this pass paid for).  (Sixteen documents per wave as one id stream with segmented sums -- the emit
// kernel's form -- was slower here: 0.52 ms against 0.34, the bookkeeping outweighs the idle lanes.)
#define TKD_LEN_BLOCK 1024     /* 16 waves share one LDS copy of the table: two blocks fill a CU's wave slots */
__global__ __launch_bounds__(TKD_LEN_BLOCK) void tk_decode_doclen_kernel(TkDecodeArgs a) {
    __shared__ uint32_t l8w[TKD_L8_LDS / 4];
    const uint32_t n_lds = a.n_ranks < TKD_L8_LDS ? a.n_ranks : TKD_L8_LDS;
    for (uint32_t q = threadIdx.x; q < (n_lds + 3u) / 4u; q += TKD_LEN_BLOCK) l8w[q] = reinterpret_cast<const uint32_t*>(a.tok_len8)[q];
    __syncthreads();
    const uint8_t* l8 = reinterpret_cast<const uint8_t*>(l8w);
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKD_LEN_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKD_LEN_BLOCK / 64);
    for (uint64_t d = wave; d < a.n_docs; d += n_waves) {
        const uint64_t i0 = a.id_offs[d], i1 = a.id_offs[d + 1];
        uint32_t acc = 0;
        for (uint64_t i = i0 + (uint64_t)lane; i < i1; i += 64) {
            const uint32_t id = a.ids[i];
            if (id < a.num_special) {
                if (a.policy == TK_POLICY_RAISE) atomicMin(a.err + 0, (unsigned long long)i);
                if (a.policy == TK_POLICY_KEEP) acc += a.sp_offs[id + 1] - a.sp_offs[id];
            } else {
                const uint32_t r = id - a.num_special;
                if (r >= a.n_ranks) {
                    atomicMin(a.err + 1, (unsigned long long)i);
                } else {
                    uint32_t len = r < TKD_L8_LDS ? (uint32_t)l8[r] : (uint32_t)a.tok_len8[r];
                    if (len == 0xFFu) len = a.tok_offs[r + 1] - a.tok_offs[r];
                    acc += len;
                }
            }
        }
        acc = (uint32_t)__builtin_amdgcn_readlane((int)tkd_scan_incl(acc), 63);
        if (lane == 0) a.lens[d] = acc;
    }
}

// pass A, the form the pipeline uses: the text length of every GROUP of TKD_DOCS consecutive documents -- all the emit kernel needs
// to know where a group's text begins (it places the documents inside the group itself and writes their offsets).  One wave per
// group, the group's ids as one coalesced stream, four loads in flight per lane, ONE reduction per group: 0.5 instructions per id
// where the per-document pass above spends 2 (its ~150 instructions of overhead per document are 98 ids' worth on the C2 shape).
// Error flags as above; err[3] is set when a group's text reaches 4 GiB (the scan takes 32-bit lengths): the host then falls
// back to the per-document pass.
__global__ __launch_bounds__(TKD_LEN_BLOCK) void tk_decode_grouplen_kernel(TkDecodeArgs a) {
    __shared__ uint32_t l8w[TKD_L8_LDS / 4];
    const uint32_t n_lds = a.n_ranks < TKD_L8_LDS ? a.n_ranks : TKD_L8_LDS;
    for (uint32_t q = threadIdx.x; q < (n_lds + 3u) / 4u; q += TKD_LEN_BLOCK) l8w[q] = reinterpret_cast<const uint32_t*>(a.tok_len8)[q];
    __syncthreads();
    const uint8_t* l8 = reinterpret_cast<const uint8_t*>(l8w);
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKD_LEN_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKD_LEN_BLOCK / 64);
    const uint64_t n_groups = (a.n_docs + TKD_DOCS - 1) / TKD_DOCS;
    auto id_len = [&](uint32_t id, uint64_t i) -> uint32_t {
        if (id < a.num_special) {
            if (a.policy == TK_POLICY_RAISE) atomicMin(a.err + 0, (unsigned long long)i);
            return a.policy == TK_POLICY_KEEP ? a.sp_offs[id + 1] - a.sp_offs[id] : 0u;
        }
        const uint32_t r = id - a.num_special;
        if (r >= a.n_ranks) {
            atomicMin(a.err + 1, (unsigned long long)i);
            return 0u;
        }
        uint32_t len = r < TKD_L8_LDS ? (uint32_t)l8[r] : (uint32_t)a.tok_len8[r];
        if (len == 0xFFu) len = a.tok_offs[r + 1] - a.tok_offs[r];
        return len;
    };
    for (uint64_t g = wave; g < n_groups; g += n_waves) {
        const uint64_t dA = g * TKD_DOCS, dB = dA + TKD_DOCS < a.n_docs ? dA + TKD_DOCS : a.n_docs;
        const uint64_t i0 = a.id_offs[dA], i1 = a.id_offs[dB];
        uint64_t acc = 0;
        uint64_t i = i0 + (uint64_t)lane;
        for (; i + 192 < i1; i += 256) {
            const uint32_t v0 = a.ids[i], v1 = a.ids[i + 64], v2 = a.ids[i + 128], v3 = a.ids[i + 192];
            acc += (uint64_t)id_len(v0, i) + id_len(v1, i + 64) + id_len(v2, i + 128) + id_len(v3, i + 192);
        }
        for (; i < i1; i += 64) acc += id_len(a.ids[i], i);
        // 64-bit sum over the lanes (butterfly)
        for (int sh = 32; sh >= 1; sh >>= 1) {
            const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)acc, sh), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(acc >> 32), sh);
            acc += ((uint64_t)ohi << 32) | olo;
        }
        const uint64_t tot = acc;
        if (lane == 0) {
            a.glens[g] = (uint32_t)tot;
            if (tot >= (uint64_t)a.group_limit || i1 - i0 >= (uint64_t)a.group_limit) atomicMin(a.err + 3, (unsigned long long)g);
        }
    }
}

// pass B: one wave per document.  64 ids at a time: an in-register prefix sum places every token inside the
// chunk's byte span, lanes scatter their token bytes into a per-wave LDS image of that span (byte writes
// stay on chip), and the wave streams the image to HBM with aligned, coalesced dword stores; only the
// first / last partial dword of a span is written byte-wise.  ids that start a run mark a bit for the
// UTF-8 pass.
#define TKD_IMG_BYTES 4096u
typedef uint32_t __attribute__((ext_vector_type(4))) tkd_u32x4;

// the inline entry of id (zeros for a special / out-of-range id: such an id takes the slow path of its step)
__device__ __forceinline__ tkd_u32x4 tkd_entry(const TkDecodeArgs& a, uint32_t id, bool have) {
    tkd_u32x4 v = {0u, 0u, 0u, 0xFF000000u};
    const uint32_t r = id - a.num_special;
    if (have && id >= a.num_special && r < a.n_ranks) v = *reinterpret_cast<const tkd_u32x4*>(a.tok_inline + 16ull * r);
    return v;
}

template <bool GROUPS>
__global__ __launch_bounds__(TKD_BLOCK) void tk_decode_emit_kernel(TkDecodeArgs a) {
    __shared__ uint32_t img_all[(TKD_BLOCK / 64) * (TKD_IMG_BYTES / 4 + 2)];
    const int lane = threadIdx.x & 63;
    uint32_t* img = img_all + (threadIdx.x >> 6) * (TKD_IMG_BYTES / 4 + 2);
    uint8_t* img8 = reinterpret_cast<uint8_t*>(img);
    const uint64_t wave = (uint64_t)blockIdx.x * (TKD_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKD_BLOCK / 64);
    // A wave takes TKD_DOCS consecutive documents as ONE stream of ids: the text of consecutive documents is contiguous
    // (out_offs is the running sum of the lengths), so the 64-id steps run across the document boundaries -- full lanes
    // (a 98-id document alone fills 77 % of two steps) and no per-document loads.  The loads of a step are issued two
    // steps (ids) and one step (entries) ahead: a step waits for nothing that was not requested a whole step earlier.
    for (uint64_t dA = wave * TKD_DOCS; dA < a.n_docs; dA += n_waves * TKD_DOCS) {
        const uint64_t dB = dA + TKD_DOCS < a.n_docs ? dA + TKD_DOCS : a.n_docs;
        const uint64_t i0 = a.id_offs[dA], i1 = a.id_offs[dB];
        uint64_t cursor = GROUPS ? a.goffs[dA / TKD_DOCS] : a.out_offs[dA];
        // Group mode (goffs): the documents' own offsets come from HERE -- a document begins where the step's prefix sum puts its
        // first id.  Lane j < 16 keeps ONE word for document dA + j: the distance of its first id from the group's first one (31
        // bits: the length pass sends a group of 2^31 ids or bytes down the other path), then -- bit 31 set -- where its text begins
        // in the group's text; jn / nrel: the next document whose start has not been seen, wave-uniform (the starts are in order).
        // One coalesced store of the 16 offsets per group.  (Costs the kernel its eighth wave per SIMD: 72 VGPRs against 60, 0.75
        // against 0.68 ms on the C2 shape -- and saves 0.13 of the length pass's 0.26.)
        const uint32_t ndg = (uint32_t)(dB - dA);
        uint32_t dword = 0x7FFFFFFFu;               // lanes past the group's documents: "never"
        const uint64_t gstart = cursor;
        if (GROUPS && (uint32_t)lane < ndg) {
            const uint64_t ds = a.id_offs[dA + (uint64_t)lane];
            dword = ds < i1 ? (uint32_t)(ds - i0) : 0x7FFFFFFEu;     // (0x7FFFFFFE: an empty document at the group's very end)
        }
        uint32_t jn = 0, nrel = GROUPS ? (uint32_t)__builtin_amdgcn_readlane((int)dword, 0) : 0x7FFFFFFFu;
        bool hi_any = false;                       // some byte >= 0x80 among the token bytes this lane emitted
        uint32_t id0 = i0 + (uint64_t)lane < i1 ? a.ids[i0 + lane] : 0u;
        uint32_t id1 = i0 + 64 + (uint64_t)lane < i1 ? a.ids[i0 + 64 + lane] : 0u;
        tkd_u32x4 e0 = tkd_entry(a, id0, i0 + (uint64_t)lane < i1);
        uint32_t prev_last = a.num_special;        // the id before the step's first one (none before the group: not special)
        for (uint64_t c0 = i0; c0 < i1; c0 += 64) {
            const uint64_t i = c0 + (uint64_t)lane;
            const uint32_t id2 = i + 128 < i1 ? a.ids[i + 128] : 0u;            // two steps ahead
            const tkd_u32x4 e1 = tkd_entry(a, id1, i + 64 < i1);                 // one step ahead
            const uint32_t id = id0;
            // the id in front of this lane's (lane 0: the last id of the step before) -- all lanes take part
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)prev_last, (int)id, 0x138, 0xF, 0xF, false);  // wave_shr:1
            const uint8_t* src = nullptr;
            uint32_t len = 0;
            bool mark = false;
            uint32_t w[4] = {0u, 0u, 0u, 0u};          // the token's first 16 bytes
            if (i < i1) {
                const uint32_t l = e0.w >> 24;
                if (l != 0xFFu) {
                    // ONE 16-byte gather gave the bytes and the length of a token of up to 15 bytes (all but a few hundred
                    // of a 130 k vocabulary): offsets pair -> bytes was two dependent gathers
                    len = l;
                    w[0] = e0.x; w[1] = e0.y; w[2] = e0.z; w[3] = e0.w & 0x00FFFFFFu;
                } else {
                    len = tkd_piece(a, id, &src);      // special ids, long tokens, ids out of range
                    if (len && len <= 16u) {
                        typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_u;
                        const u32x4_u v = *reinterpret_cast<const u32x4_u*>(src);
                        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
                    }
                }
                // run starts (superset): every special id and every id that follows one
                mark = id < a.num_special || (i > i0 && before < a.num_special);
            }
            prev_last = (uint32_t)__builtin_amdgcn_readlane((int)id, 63);
            const uint32_t incl = tkd_scan_incl(len);
            const uint32_t span = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint64_t dst = cursor + (incl - len);
            if (mark) atomicOr(a.run_bits + (dst >> 5), 1u << (dst & 31));
            const uint64_t base = cursor, end = cursor + span, g0 = base & ~3ull;
            if (end - g0 <= TKD_IMG_BYTES) {
                const uint32_t off = (uint32_t)(dst - g0);
                if (len <= 16u) {
                    // the token bytes are in registers.  They go into the LDS image by the bits of the length -- 8, 4, 2, 1
                    // bytes, unaligned LDS stores (gfx950 takes them) -- four predicated stores instead of sixteen byte
                    // stores with their extraction
                    typedef uint32_t __attribute__((aligned(1))) u32_u;
                    typedef uint16_t __attribute__((aligned(1))) u16_u;
                    typedef uint64_t __attribute__((aligned(1))) u64_u;
                    if (src && len < 16u) {       // (not an inline entry: the bytes past the token are somebody else's)
#pragma unroll
                        for (uint32_t q = 0; q < 4u; ++q) {
                            const uint32_t have = len > 4u * q ? len - 4u * q : 0u;
                            w[q] &= have >= 4u ? 0xFFFFFFFFu : ((1u << (8u * have)) - 1u);
                        }
                    }
                    hi_any |= ((w[0] | w[1] | w[2] | w[3]) & 0x80808080u) != 0u;
                    uint8_t* q8 = img8 + off;
                    uint32_t a0 = w[0], a1 = w[1], a2 = w[2], a3 = w[3];
                    if (len & 16u) {
                        *reinterpret_cast<u64_u*>(q8) = (uint64_t)a0 | ((uint64_t)a1 << 32);
                        *reinterpret_cast<u64_u*>(q8 + 8) = (uint64_t)a2 | ((uint64_t)a3 << 32);
                    }
                    if (len & 8u) {
                        *reinterpret_cast<u64_u*>(q8) = (uint64_t)a0 | ((uint64_t)a1 << 32);
                        a0 = a2; a1 = a3; q8 += 8;
                    }
                    if (len & 4u) {
                        *reinterpret_cast<u32_u*>(q8) = a0;
                        a0 = a1; q8 += 4;
                    }
                    if (len & 2u) {
                        *reinterpret_cast<u16_u*>(q8) = (uint16_t)a0;
                        a0 >>= 16; q8 += 2;
                    }
                    if (len & 1u) *q8 = (uint8_t)a0;
                } else {
                    for (uint32_t k = 0; k < len; ++k) {
                        const uint8_t bk = src[k];
                        img8[off + k] = bk;
                        hi_any |= bk >= 0x80u;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t nwords = (uint32_t)((end - g0 + 3) / 4);
                for (uint32_t wq = (uint32_t)lane; wq < nwords; wq += 64u) {
                    const uint64_t ga = g0 + 4ull * wq;
                    if (ga >= base && ga + 4 <= end) {
                        *reinterpret_cast<uint32_t*>(a.out_bytes + ga) = img[wq];
                    } else {
                        const uint64_t lo = ga > base ? ga : base, hi = ga + 4 < end ? ga + 4 : end;
                        for (uint64_t q = lo; q < hi; ++q) a.out_bytes[q] = img8[q - g0];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // image reads done before the next chunk overwrites it
                __builtin_amdgcn_wave_barrier();
            } else {
                uint8_t* out = a.out_bytes + dst;  // unusually long tokens: direct byte stores
                for (uint32_t k = 0; k < len; ++k) {
                    const uint8_t bk = src ? src[k] : (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));   // (no src: an inline entry, <= 15 bytes)
                    out[k] = bk;
                    hi_any |= bk >= 0x80u;
                }
            }
            if (GROUPS) {
                // documents whose first id is one of this step's 64 (scalar loop: jn, nrel are wave-uniform)
                const uint32_t srel = (uint32_t)(c0 - i0);
                while (nrel - srel < 64u) {
                    // (a lane past the group's last id has len 0 and the whole span in front of it)
                    const uint32_t at = (uint32_t)(cursor - gstart) + (uint32_t)__builtin_amdgcn_readlane((int)(incl - len), (int)(nrel - srel));
                    if ((uint32_t)lane == jn) dword = 0x80000000u | at;
                    ++jn;
                    nrel = jn < ndg ? (uint32_t)__builtin_amdgcn_readlane((int)dword, (int)(jn & 63u)) : 0x7FFFFFFFu;
                }
            }
            cursor = end;
            id0 = id1; id1 = id2; e0 = e1;
        }
        if (GROUPS) {
            // empty documents at the end of the group begin where its text ends; the last group also writes the grand total
            // (what was not seen in the loop: empty documents at the group's very end -- they begin where its text ends)
            const uint32_t myoff = (dword & 0x80000000u) ? (dword & 0x7FFFFFFFu) : (uint32_t)(cursor - gstart);
            if ((uint32_t)lane < ndg) a.out_offs[dA + (uint64_t)lane] = gstart + myoff;
            if (dB == a.n_docs && lane == 0) a.out_offs[a.n_docs] = cursor;
        }
        const bool doc_hi = __ballot(hi_any) != 0ull;   // ASCII documents need no UTF-8 validation pass (decided per group)
        if (dA + (uint64_t)lane < dB) a.doc_hi[dA + lane] = doc_hi ? 1u : 0u;
    }
}

// One wave per document, FOUR BYTES PER LANE in one register: 256 bytes per step, one aligned dword per lane, all four bytes
// judged at once by tku8_err4 (tk_utf8_swar.h: the three-table look-up of Keiser & Lemire with byte permutes, run starts as hard
// boundaries -- the same checks as tk_validate_kernel, RFC 3629, with the extra rule that positions flagged in run_bits cut
// sequences).  A lane needs the dword in front of its own (DPP shift; lane 0: the last dword of the step before, a scalar
// carried along) and the run-start bits of the six positions around its dword (two words of run_bits, requested with the dword
// one step ahead).  Bytes outside the document read as NUL; the dword that holds the position BEHIND the document is judged too
// (a NUL behind an open sequence is "too short").
// (Round 2: one lane per byte, 58 bytes per step, ~100 instructions a step -- 5.8 of the 10.7 ms of a C3 decode.  Four bytes per
// lane with a lead / continuation branch per byte was no better, 4.7 ms: both branches run for every byte of non-ASCII text,
// ~300 instructions per 256 bytes.  This form: ~70.)
__global__ __launch_bounds__(TKD_BLOCK) void tk_decode_validate_kernel(TkDecodeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TKD_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TKD_BLOCK / 64);
    const uint8_t* b = a.out_bytes;
    for (uint64_t d = wave; d < a.n_docs; d += n_waves) {
        if (a.doc_hi[d] == 0u) continue;            // the emit kernel saw only ASCII bytes: always valid
        const uint64_t s0 = a.out_offs[d], s1 = a.out_offs[d + 1];
        if (s1 == s0) continue;
        bool err = false;
        const uint64_t A0 = s0 & ~3ull;
        // the dword at q, bytes outside [s0, s1) as NUL (out_bytes is allocated in whole dwords, with slack)
        auto load = [&](uint64_t q) -> uint32_t {
            if (q >= s1 || q + 4 <= s0) return 0u;
            uint32_t w = *reinterpret_cast<const uint32_t*>(b + q);
            if (q < s0) w &= 0xFFFFFFFFu << (8u * (uint32_t)(s0 - q));
            if (q + 4 > s1) w &= 0xFFFFFFFFu >> (8u * (uint32_t)(q + 4 - s1));
            return w;
        };
        // run-start bits of the positions q - 4 .. q + 27 (bit k: a run starts at q - 4 + k)
        auto load_rb = [&](uint64_t q) -> uint32_t {
            if (q > s1) return 0u;
            if (q < 4) return a.run_bits[0] << 4;
            const uint64_t rb = q - 4;
            const uint32_t r0 = a.run_bits[rb >> 5], r1 = a.run_bits[(rb >> 5) + 1];
            const uint32_t sh = (uint32_t)(rb & 31u);
            return sh ? ((r0 >> sh) | (r1 << (32u - sh))) : r0;
        };
        uint32_t nw = load(A0 + 4ull * (uint64_t)lane);
        uint32_t nrb = load_rb(A0 + 4ull * (uint64_t)lane);
        uint32_t carry = 0u;                        // the dword in front of the step's first one (in front of the document: NUL)
        for (uint64_t p0 = A0; p0 <= s1; p0 += 256) {
            const uint64_t q = p0 + 4ull * (uint64_t)lane;
            const uint32_t w = nw, RB = nrb;
            nw = load(q + 256);
            nrb = load_rb(q + 256);
            uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xF, 0xF, false);   // wave_shr:1: lane l <- lane l - 1
            if (lane == 0) pw = carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
            if (__builtin_amdgcn_ballot_w64(((w | pw) & 0x80808080u) != 0u) == 0ull) continue;        // an ASCII step (wave-uniform)
            if (q <= s1 && tku8_err4(pw, w, RB) != 0u) err = true;
        }
        if (__ballot(err) && lane == 0) atomicMin(a.err + 2, (unsigned long long)d);
    }
}

static uint32_t tkd_doc_grid(uint64_t n_docs) {
    uint64_t blocks = (n_docs + (TKD_BLOCK / 64) - 1) / (TKD_BLOCK / 64);
    return (uint32_t)(blocks > 256 * 16 ? 256 * 16 : blocks);
}

hipError_t tk_launch_decode_doclen(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_docs == 0) return hipSuccess;
    uint64_t blocks = (a.n_docs + TKD_LEN_BLOCK / 64 - 1) / (TKD_LEN_BLOCK / 64);
    if (blocks > 256u * 2u) blocks = 256u * 2u;             // every block copies 32 KB into its LDS first: no more than are resident
    hipLaunchKernelGGL(tk_decode_doclen_kernel, dim3((uint32_t)blocks), dim3(TKD_LEN_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_decode_grouplen(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_docs == 0) return hipSuccess;
    const uint64_t n_groups = (a.n_docs + TKD_DOCS - 1) / TKD_DOCS;
    uint64_t blocks = (n_groups + TKD_LEN_BLOCK / 64 - 1) / (TKD_LEN_BLOCK / 64);
    if (blocks > 256u * 2u) blocks = 256u * 2u;             // every block copies 32 KB into its LDS first: no more than are resident
    hipLaunchKernelGGL(tk_decode_grouplen_kernel, dim3((uint32_t)blocks), dim3(TKD_LEN_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_decode_emit(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_docs == 0) return hipSuccess;
    if (a.goffs) hipLaunchKernelGGL(tk_decode_emit_kernel<true>, dim3(tkd_doc_grid((a.n_docs + TKD_DOCS - 1) / TKD_DOCS)), dim3(TKD_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(tk_decode_emit_kernel<false>, dim3(tkd_doc_grid((a.n_docs + TKD_DOCS - 1) / TKD_DOCS)), dim3(TKD_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t tk_launch_decode_validate(const TkDecodeArgs& a, hipStream_t s) {
    if (a.n_docs == 0) return hipSuccess;
    uint64_t blocks = (a.n_docs + (TKD_BLOCK / 64) - 1) / (TKD_BLOCK / 64);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tk_decode_validate_kernel, dim3((uint32_t)blocks), dim3(TKD_BLOCK), 0, s, a);
    return hipGetLastError();
}
