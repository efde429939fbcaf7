// tk_kernels.hip -- gfx950 kernels of the batch tokenization path.
//
//   tk_encode_kernel<0>       pass 1: every document, one wave per document at a time
//   tk_encode_kernel<1>       pass 2: the few documents with a piece that does not fit a window
//   tk_encode_kernel<2>       split only (tk_split_batch)
//   tk_encode_kernel<3>       pass 1 over a list of documents (those the flat path handed back)
//   tk_scan_*                 per-document id counts -> output offsets (exclusive scan, u64)
//   tk_compact_kernel         staging -> packed ids (the reference's Vec<u32> per document,
//                             concatenated; ids already carry +num_special and BOS/EOS:
//                             reference src/tekkenizer.rs:390-402)
//   tk_validate_kernel        UTF-8 well-formedness per document (the &str invariant of
//                             src/tekkenizer.rs:380 for callers that are not Rust)
//
// All integer / byte work: no MFMA.  Bound: HBM (see DESIGN.md for the byte accounting).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tk_kernels.h"
#include "tk_wave_hip.h"
#include "tk_encode_impl.h"

#define TK_BLOCK 256

template <int MODE>
__global__ __launch_bounds__(TK_BLOCK) void tk_encode_kernel(TkEncodeArgs a) {
    const int lane = wv_lane();
    const uint64_t wave_id = (uint64_t)blockIdx.x * (TK_BLOCK / 64) + (threadIdx.x >> 6);
    tk_encode_wave<MODE>(a, lane, wave_id);
}

hipError_t tk_launch_encode(const TkEncodeArgs& args, int mode, uint32_t n_waves, hipStream_t s) {
    const uint32_t blocks = (n_waves + (TK_BLOCK / 64) - 1) / (TK_BLOCK / 64);
    if (blocks == 0) return hipSuccess;
    if (mode == 1) hipLaunchKernelGGL(tk_encode_kernel<1>, dim3(blocks), dim3(TK_BLOCK), 0, s, args);
    else if (mode == 2) hipLaunchKernelGGL(tk_encode_kernel<2>, dim3(blocks), dim3(TK_BLOCK), 0, s, args);
    else if (mode == 3) hipLaunchKernelGGL(tk_encode_kernel<3>, dim3(blocks), dim3(TK_BLOCK), 0, s, args);
    else hipLaunchKernelGGL(tk_encode_kernel<0>, dim3(blocks), dim3(TK_BLOCK), 0, s, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Small batches in ONE launch (the reference's own call shape is one &str per call, src/tekkenizer.rs:378-405): a
// single workgroup encodes up to TK_SMALL_MAX_DOCS documents (wave w takes documents w, w + n_waves, ...: the
// per-document pass 1), scans their id counts and packs the ids in document order -- what the batch pipeline does with
// about ten launches.  `bytes` may be mapped pinned host memory (a 64-byte string is read over PCIe by the kernel itself:
// no copy engine in the path) and so may out_ids / out_offs / status; staging and counts are device memory.
// status[0] = 1 if a document has to go to pass 2 (a piece that does not fit a window): the host then takes the batch
// pipeline; status[1] = total ids.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TK_SMALL_THREADS) void tk_small_kernel(TkEncodeArgs a, uint32_t* __restrict__ out_ids,
                                                                    uint64_t* __restrict__ out_offs, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_cnt[TK_SMALL_MAX_DOCS];
    __shared__ uint32_t s_off[TK_SMALL_MAX_DOCS + 1];
    __shared__ uint32_t s_defer;
    const int lane = wv_lane();
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_waves = TK_SMALL_THREADS / 64;
    const uint32_t n_docs = (uint32_t)a.n_docs;
    if (threadIdx.x == 0) s_defer = 0u;
    __syncthreads();
    const TkPolyPow pw = tk_poly_pow(a.t, lane);
    for (uint32_t d = wv; d < n_docs; d += n_waves) {
        const bool ok = tk_encode_doc<0>(a, (uint64_t)d, lane, pw);
        if (!ok && lane == 0) s_defer = 1u;
    }
    __threadfence_block();
    __syncthreads();
    if (s_defer) {                                   // block-uniform
        if (threadIdx.x == 0) { status[0] = 1u; status[1] = 0u; }
        return;
    }
    for (uint32_t d = threadIdx.x; d < n_docs; d += TK_SMALL_THREADS) s_cnt[d] = a.counts[d];
    __syncthreads();
    if (wv == 0) {                                   // exclusive scan of up to TK_SMALL_MAX_DOCS counts by one wave
        uint32_t carry = 0;
        for (uint32_t base = 0; base < n_docs; base += 64u) {
            const uint32_t d = base + (uint32_t)lane;
            const uint32_t c = d < n_docs ? s_cnt[d] : 0u;
            const uint32_t incl = wv_scan_incl_u32(c);
            if (d < n_docs) s_off[d] = carry + incl - c;
            carry += wv_readlane(incl, 63);
        }
        if (lane == 0) s_off[n_docs] = carry;
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d <= n_docs; d += TK_SMALL_THREADS) out_offs[d] = (uint64_t)s_off[d];
    for (uint32_t d = wv; d < n_docs; d += n_waves) {
        const uint32_t* src = a.staging + a.doc_offs[d] + 2ull * d;
        uint32_t* dst = out_ids + s_off[d];
        const uint32_t cnt = s_cnt[d];
        for (uint32_t k = (uint32_t)lane; k < cnt; k += 64u) dst[k] = src[k];
    }
    if (threadIdx.x == 0) { status[0] = 0u; status[1] = s_off[n_docs]; }
}

hipError_t tk_launch_small(const TkEncodeArgs& args, uint32_t* out_ids, uint64_t* out_offs, uint32_t* status, hipStream_t s) {
    if (args.n_docs > TK_SMALL_MAX_DOCS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tk_small_kernel, dim3(1), dim3(TK_SMALL_THREADS), 0, s, args, out_ids, out_offs, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// exclusive scan of per-document counts (u32) into u64 offsets; 2048 counts per block
// ------------------------------------------------------------------------------------------
#define TK_SCAN_PER_THREAD 8
#define TK_SCAN_TILE (TK_BLOCK * TK_SCAN_PER_THREAD)

__device__ __forceinline__ uint64_t tk_block_exclusive_scan(uint64_t v, uint64_t* lds, uint64_t* total) {
    // wave-level inclusive scan with DPP-free shuffles, then a scan over the 4 wave totals
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint64_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(x, d);
        if (lane >= d) x += o;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int w = 0; w < TK_BLOCK / 64; ++w) {
        uint64_t s = lds[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(TK_BLOCK) void tk_scan_block_sums(const uint32_t* counts, uint64_t n, uint64_t* block_sums) {
    __shared__ uint64_t lds[TK_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * TK_SCAN_TILE + (uint64_t)threadIdx.x * TK_SCAN_PER_THREAD;
    uint64_t v = 0;
    for (int k = 0; k < TK_SCAN_PER_THREAD; ++k)
        if (base + k < n) v += counts[base + k];
    uint64_t tot;
    (void)tk_block_exclusive_scan(v, lds, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(TK_BLOCK) void tk_scan_top(uint64_t* block_sums, uint64_t n_blocks) {
    // single block: exclusive scan of block_sums in place; total at block_sums[n_blocks]
    __shared__ uint64_t lds[TK_BLOCK / 64];
    uint64_t carry = 0;
    for (uint64_t base = 0; base < n_blocks; base += TK_BLOCK) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n_blocks ? block_sums[i] : 0;
        uint64_t tot;
        const uint64_t ex = tk_block_exclusive_scan(v, lds, &tot);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) block_sums[n_blocks] = carry;
}

__global__ __launch_bounds__(TK_BLOCK) void tk_scan_apply(const uint32_t* counts, uint64_t n, const uint64_t* block_sums,
                                                          uint64_t n_blocks, uint64_t* offs) {
    __shared__ uint64_t lds[TK_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * TK_SCAN_TILE + (uint64_t)threadIdx.x * TK_SCAN_PER_THREAD;
    uint32_t c[TK_SCAN_PER_THREAD];
    uint64_t v = 0;
    for (int k = 0; k < TK_SCAN_PER_THREAD; ++k) {
        c[k] = (base + k < n) ? counts[base + k] : 0u;
        v += c[k];
    }
    uint64_t tot;
    uint64_t run = block_sums[blockIdx.x] + tk_block_exclusive_scan(v, lds, &tot);
    for (int k = 0; k < TK_SCAN_PER_THREAD; ++k) {
        if (base + k < n) offs[base + k] = run;
        run += c[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offs[n] = block_sums[n_blocks];
}

hipError_t tk_launch_scan(const uint32_t* counts, uint64_t n, uint64_t* offs, uint64_t* block_sums, hipStream_t s) {
    const uint64_t n_blocks = (n + TK_SCAN_TILE - 1) / TK_SCAN_TILE;
    if (n_blocks == 0) {
        return hipMemsetAsync(offs, 0, sizeof(uint64_t), s);
    }
    hipLaunchKernelGGL(tk_scan_block_sums, dim3((uint32_t)n_blocks), dim3(TK_BLOCK), 0, s, counts, n, block_sums);
    hipLaunchKernelGGL(tk_scan_top, dim3(1), dim3(TK_BLOCK), 0, s, block_sums, n_blocks);
    hipLaunchKernelGGL(tk_scan_apply, dim3((uint32_t)n_blocks), dim3(TK_BLOCK), 0, s, counts, n, block_sums, n_blocks,
                       offs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// compaction: one wave per document, grid-stride over documents
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TK_BLOCK) void tk_compact_kernel(const uint32_t* __restrict__ staging,
                                                              const uint64_t* __restrict__ doc_offs,
                                                              const uint32_t* __restrict__ counts,
                                                              const uint64_t* __restrict__ out_offs, uint64_t n_docs,
                                                              uint32_t* __restrict__ out_ids) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TK_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TK_BLOCK / 64);
    for (uint64_t d = wave; d < n_docs; d += n_waves) {
        const uint32_t cnt = counts[d];
        const uint32_t* src = staging + doc_offs[d] + 2 * d;
        uint32_t* dst = out_ids + out_offs[d];
        for (uint32_t k = (uint32_t)lane; k < cnt; k += 64u) dst[k] = src[k];
    }
}

hipError_t tk_launch_compact(const uint32_t* staging, const uint64_t* doc_offs, const uint32_t* counts,
                             const uint64_t* out_offs, uint64_t n_docs, uint32_t* out_ids, hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    uint64_t blocks = (n_docs + (TK_BLOCK / 64) - 1) / (TK_BLOCK / 64);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tk_compact_kernel, dim3((uint32_t)blocks), dim3(TK_BLOCK), 0, s, staging, doc_offs, counts,
                       out_offs, n_docs, out_ids);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// UTF-8 validation (RFC 3629 well-formedness: no overlongs, no surrogates, <= U+10FFFF)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TK_BLOCK) void tk_validate_kernel(const uint8_t* __restrict__ bytes,
                                                               const uint64_t* __restrict__ doc_offs, uint64_t n_docs,
                                                               uint32_t* bad) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (TK_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (TK_BLOCK / 64);
    for (uint64_t d = wave; d < n_docs; d += n_waves) {
        const uint64_t s0 = doc_offs[d], s1 = doc_offs[d + 1];
        bool err = false;
        for (uint64_t p = s0 + (uint64_t)lane; p < s1; p += 64) {
            const uint32_t b0 = bytes[p];
            if (b0 < 0x80u || (b0 & 0xC0u) == 0x80u) {
                // a continuation byte must be preceded (within 3) by a lead that covers it
                if ((b0 & 0xC0u) == 0x80u) {
                    bool ok = false;
                    for (uint32_t k = 1; k <= 3 && p >= s0 + k; ++k) {
                        const uint32_t q = bytes[p - k];
                        if ((q & 0xC0u) == 0x80u) continue;
                        const uint32_t need = q >= 0xF0u ? 3u : q >= 0xE0u ? 2u : q >= 0xC0u ? 1u : 0u;
                        ok = need >= k;
                        break;
                    }
                    if (!ok) err = true;
                }
                continue;
            }
            const uint32_t need = b0 >= 0xF8u ? 99u : b0 >= 0xF0u ? 3u : b0 >= 0xE0u ? 2u : b0 >= 0xC2u ? 1u : 99u;
            if (need == 99u || p + need >= s1) { err = true; continue; }  // bad lead or truncated
            const uint32_t b1 = bytes[p + 1];
            if ((b1 & 0xC0u) != 0x80u) { err = true; continue; }
            if (need >= 2u && (bytes[p + 2] & 0xC0u) != 0x80u) { err = true; continue; }
            if (need == 3u && (bytes[p + 3] & 0xC0u) != 0x80u) { err = true; continue; }
            if (b0 == 0xE0u && b1 < 0xA0u) err = true;   // overlong 3-byte
            if (b0 == 0xEDu && b1 >= 0xA0u) err = true;  // surrogates
            if (b0 == 0xF0u && b1 < 0x90u) err = true;   // overlong 4-byte
            if (b0 == 0xF4u && b1 >= 0x90u) err = true;  // > U+10FFFF
            if (b0 > 0xF4u) err = true;
        }
        if (__ballot(err) && lane == 0) atomicAdd(bad, 1u);
    }
}

// the document offsets of a device-resident batch (tk_encode_batch_device_ex): [0] == 0, non-decreasing, [n_docs] == n_bytes --
// what every kernel of the pipeline indexes with.  *bad counts the violations.
__global__ __launch_bounds__(TK_BLOCK) void tk_check_offsets_kernel(const uint64_t* __restrict__ doc_offs, uint64_t n_docs, uint64_t n_bytes,
                                                                     uint32_t* bad) {
    const uint64_t d = (uint64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    bool err = false;
    if (d == 0) err = doc_offs[0] != 0ull || doc_offs[n_docs] != n_bytes;
    if (d < n_docs) err = err || doc_offs[d + 1] < doc_offs[d] || doc_offs[d + 1] > n_bytes;
    const uint64_t m = __ballot(err);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(bad, (uint32_t)__builtin_popcountll(m));
}
hipError_t tk_launch_check_offsets(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_bytes, uint32_t* d_bad, hipStream_t s) {
    const uint64_t blocks = (n_docs + 1 + TK_BLOCK - 1) / TK_BLOCK;
    hipLaunchKernelGGL(tk_check_offsets_kernel, dim3((uint32_t)blocks), dim3(TK_BLOCK), 0, s, doc_offs, n_docs, n_bytes, d_bad);
    return hipGetLastError();
}

hipError_t tk_launch_validate(const uint8_t* bytes, const uint64_t* doc_offs, uint64_t n_docs, uint32_t* d_bad,
                              hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    uint64_t blocks = (n_docs + (TK_BLOCK / 64) - 1) / (TK_BLOCK / 64);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tk_validate_kernel, dim3((uint32_t)blocks), dim3(TK_BLOCK), 0, s, bytes, doc_offs, n_docs, d_bad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// longest deferred document (sizes the pass-2 scratch)
// ------------------------------------------------------------------------------------------
__global__ void tk_defer_maxlen_kernel(const uint32_t* defer_list, uint32_t n, const uint64_t* doc_offs, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t d = defer_list[i];
    const uint64_t len = doc_offs[d + 1] - doc_offs[d];
    atomicMax(out, (uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len));
}

hipError_t tk_launch_defer_maxlen(const uint32_t* defer_list, uint32_t n, const uint64_t* doc_offs, uint32_t* d_out,
                                  hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_defer_maxlen_kernel, dim3((n + 255) / 256), dim3(256), 0, s, defer_list, n, doc_offs, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// self-test of the wave primitives (run once per context)
// ------------------------------------------------------------------------------------------
__global__ void tk_wave_selftest_kernel(uint32_t* fail) {
    const int lane = wv_lane();
    uint32_t v = 1000u + (uint32_t)lane;
    uint32_t bad = 0;
    const uint32_t up = wv_up1(v), dn = wv_dn1(v);
    if (up != (lane < 63 ? v + 1u : 0u)) bad |= 1u;
    if (dn != (lane > 0 ? v - 1u : 0u)) bad |= 2u;
    if (wv_shfl(v, 63 - lane) != 1000u + (uint32_t)(63 - lane)) bad |= 4u;
    if (wv_ballot((lane & 1) == 0) != 0x5555555555555555ull) bad |= 8u;
    if (tk_wave_sum((uint32_t)lane, lane) != 2016u) bad |= 16u;
    if (tk_wave_min64(((uint64_t)(100 - lane) << 32) | (uint32_t)lane, lane) != (((uint64_t)37 << 32) | 63u)) bad |= 32u;
    // DPP min reduction: minimum placed in every row / at both ends in turn
    for (int at = 0; at < 64; at += 7) {
        if (wv_min_u32(lane == at ? 5u : 1000u + (uint32_t)lane) != 5u) bad |= 64u;
    }
    if (wv_min_u32(0xFFFFFFFFu) != 0xFFFFFFFFu) bad |= 128u;
    if (wv_readlane(v, 17) != 1017u) bad |= 256u;
    if (wv_scan_incl_u32((uint32_t)lane * 3u + 1u) != (uint32_t)(3 * (lane * (lane + 1)) / 2 + lane + 1)) bad |= 512u;
    if (wv_perm(0x07060504u, 0x03020100u, 0x05010400u) != 0x05010400u || wv_inverse_ballot(0xF0F0F0F0F0F0F0F1ull) != ((lane & 4) != 0 || lane == 0)) bad |= 1024u;
    if (bad) atomicOr(fail, bad);
}

hipError_t tk_launch_wave_selftest(uint32_t* d_fail, hipStream_t s) {
    hipLaunchKernelGGL(tk_wave_selftest_kernel, dim3(1), dim3(64), 0, s, d_fail);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 18-bit wire format of token ids for the multi-GPU gather: the xGMI link into rank 0 is what
// bounds N > 1, and an id of a Tekken vocabulary (< 2^18) travels as 2.25 bytes instead of 4:
//   [ n x u16 low halves | padded to 4 bytes | ceil(n / 16) x u32, two high bits of 16 ids each ]
// One thread per 16 ids (64 contiguous bytes in, 32 + 4 out); an id >= 2^18 raises *d_bad.
// ------------------------------------------------------------------------------------------
// The id buffers are only word-aligned in general (a rank's ids land at an arbitrary id index of the gathered buffer) and the
// wire buffer is a caller's pointer: 16-byte accesses through vector types that promise 4-byte alignment only.
typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) tk_u32x4_w;

__global__ __launch_bounds__(256) void tk_pack18_kernel(const uint32_t* __restrict__ ids, uint64_t n, uint16_t* __restrict__ lows,
                                                        uint32_t* __restrict__ highs, uint32_t* __restrict__ d_bad) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, i0 = g * 16;
    if (i0 >= n) return;
    uint32_t v[16];
    if (i0 + 16 <= n) {
        const tk_u32x4_w* src = reinterpret_cast<const tk_u32x4_w*>(ids + i0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const tk_u32x4_w x = src[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = i0 + q < n ? ids[i0 + q] : 0u;
    }
    uint32_t hi = 0, over = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { hi |= ((v[q] >> 16) & 3u) << (2 * q); over |= v[q] >> 18; }
    if (over) atomicOr(d_bad, 1u);
    highs[g] = hi;
    if (i0 + 16 <= n) {
        tk_u32x4_w* dst = reinterpret_cast<tk_u32x4_w*>(lows + i0);   // 32 i0 bytes into a word-aligned buffer
        tk_u32x4_w a, b;
        a.x = (v[0] & 0xFFFFu) | (v[1] << 16); a.y = (v[2] & 0xFFFFu) | (v[3] << 16); a.z = (v[4] & 0xFFFFu) | (v[5] << 16); a.w = (v[6] & 0xFFFFu) | (v[7] << 16);
        b.x = (v[8] & 0xFFFFu) | (v[9] << 16); b.y = (v[10] & 0xFFFFu) | (v[11] << 16); b.z = (v[12] & 0xFFFFu) | (v[13] << 16); b.w = (v[14] & 0xFFFFu) | (v[15] << 16);
        dst[0] = a; dst[1] = b;
    } else {
        for (int q = 0; q < 16 && i0 + q < n; ++q) lows[i0 + q] = (uint16_t)v[q];
    }
}

__global__ __launch_bounds__(256) void tk_unpack18_kernel(const uint16_t* __restrict__ lows, const uint32_t* __restrict__ highs, uint64_t n,
                                                          uint32_t* __restrict__ ids) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x, i0 = g * 16;
    if (i0 >= n) return;
    const uint32_t hi = highs[g];
    if (i0 + 16 <= n) {
        const tk_u32x4_w* src = reinterpret_cast<const tk_u32x4_w*>(lows + i0);
        const tk_u32x4_w a = src[0], b = src[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        tk_u32x4_w* dst = reinterpret_cast<tk_u32x4_w*>(ids + i0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            tk_u32x4_w o;
            o.x = (w[2 * q] & 0xFFFFu) | (((hi >> (8 * q)) & 3u) << 16);
            o.y = (w[2 * q] >> 16) | (((hi >> (8 * q + 2)) & 3u) << 16);
            o.z = (w[2 * q + 1] & 0xFFFFu) | (((hi >> (8 * q + 4)) & 3u) << 16);
            o.w = (w[2 * q + 1] >> 16) | (((hi >> (8 * q + 6)) & 3u) << 16);
            dst[q] = o;
        }
    } else {
        for (int q = 0; q < 16 && i0 + q < n; ++q) ids[i0 + q] = (uint32_t)lows[i0 + q] | (((hi >> (2 * q)) & 3u) << 16);
    }
}

hipError_t tk_launch_pack18(const uint32_t* ids, uint64_t n, void* packed, uint32_t* d_bad, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint16_t* lows = reinterpret_cast<uint16_t*>(packed);
    uint32_t* highs = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(packed) + ((2 * n + 3) & ~3ull));
    const uint64_t groups = (n + 15) / 16;
    hipLaunchKernelGGL(tk_pack18_kernel, dim3((uint32_t)((groups + 255) / 256)), dim3(256), 0, s, ids, n, lows, highs, d_bad);
    return hipGetLastError();
}

hipError_t tk_launch_unpack18(const void* packed, uint64_t n, uint32_t* ids, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint16_t* lows = reinterpret_cast<const uint16_t*>(packed);
    const uint32_t* highs = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(packed) + ((2 * n + 3) & ~3ull));
    const uint64_t groups = (n + 15) / 16;
    hipLaunchKernelGGL(tk_unpack18_kernel, dim3((uint32_t)((groups + 255) / 256)), dim3(256), 0, s, lows, highs, n, ids);
    return hipGetLastError();
}

// p[i] += add (the node-level gather rebases a run's id offsets to batch offsets)
__global__ __launch_bounds__(256) void tk_add_u64_kernel(uint64_t* __restrict__ p, uint64_t n, uint64_t add) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] += add;
}
hipError_t tk_launch_add_u64(uint64_t* p, uint64_t n, uint64_t add, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_add_u64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, p, n, add);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void tk_iota_kernel(uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}
hipError_t tk_launch_iota(uint32_t* out, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(tk_iota_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, out, n);
    return hipGetLastError();
}
