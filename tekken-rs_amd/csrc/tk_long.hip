// tk_long.hip -- documents with a LONG piece that is not a vocabulary key: one WORKGROUP per document, the long piece
// merged in ROUNDS.
//
// The byte-pair merge (tiktoken's _byte_pair_merge behind CoreBPE::encode, reference src/tekkenizer.rs:384-386; SURVEY
// App. A.2) is sequential: merge the leftmost pair of minimum rank, re-probe its two neighbours, repeat.  A 32 KiB run of
// letters is one piece with ~10^4 such steps -- one wave needs ~1.1 us per step (tk_piece_merge_coop), and that single
// chain was the whole tail of the Zipf shape (BASELINE configs[4]).  All occurrences of the current minimum rank r* can
// be merged in one parallel sweep (a ROUND) as long as the result stays what the sequential order gives
// (tools/batched_merge_model.py is the executable statement, checked against the sequential algorithm on adversarial
// vocabularies):
//   * candidates = positions whose pair has rank r*; inside a run of CONSECUTIVE candidates only the even offsets merge;
//   * every merge is probed as the sequential algorithm would see it (left neighbour already merged if the previous
//     occurrence ends right before it, right neighbour not yet merged); if a created pair ranks BELOW r* the round is cut
//     after that occurrence -- the sequential algorithm would turn to the new pair first;
//   * the parts are compacted, the probed ranks become the new pair ranks, the minimum of the new ranks is the next r*.
// A random-letter piece of 32 KiB takes under a thousand rounds.
//
// Layout: the parts (token, rank of the pair with the successor) live in two u32 arrays in global scratch (L2-resident,
// coalesced); wave w owns a contiguous range, lane l of step s the part w * per + 64 s + l, and holds its range in
// registers across the round (read once, written once).  Which parts merge is a 64-bit mask per step, derived from the
// candidate ballot with the carry-ripple trick (runs that start at an even / odd position); the run that crosses a wave
// boundary is settled with one exchange through LDS.  Six workgroup barriers per round.
//
// How the documents get here and back: the three kernels at the end of this file.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tk_kernels.h"
#include "tk_wave_hip.h"
#include "tk_encode_impl.h"
#include "tk_long_impl.h"

// ------------------------------------------------------------------------------------------------------------------
// The documents of the long list take three launches:
//   tk_long_walk_kernel     one wave per document walks it piece by piece (sequential matcher, whole-piece lookup, the
//                           single-wave merge for ordinary pieces) and writes the ids of the piece at byte offset p to
//                           staging slot p (a piece of L bytes has at most L ids; the slots it does not use become
//                           holes); a long piece that is not a vocabulary key becomes a JOB instead;
//   tk_long_merge_kernel    one workgroup per job: the round-based merge above (compacting rounds for repetitive pieces, lazy
//                           rounds for the others), ids (+ holes) into the piece's slots;
//   tk_long_compact_kernel  one wave per document squeezes the holes out, appends EOS and writes the id count.
// (One fused kernel -- wave 0 walking, all waves joining for a long piece -- was the first form: the walker's 136 registers
// under the 128 a 1024-thread block leaves a wave spilled 237 of them and dragged the round loop's arrays into scratch.)
// ------------------------------------------------------------------------------------------------------------------
#define TKL_HOLE 0xFFFFFFFFu

__global__ __launch_bounds__(256) void tk_long_walk_kernel(TkEncodeArgs a) {
    const int lane = wv_lane();
    const TkTablesView& t = a.t;
    const TkPolyPow pw = tk_poly_pow(t, lane);
    const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t* scratch = a.scratch + wave_id * a.scratch_words_per_wave;
    const uint32_t n_todo = a.n_todo_dev ? wv_first(*a.n_todo_dev) : a.n_todo;
    for (;;) {
        const uint32_t ticket = wv_first(wv_atomic_add_all(a.work_counter, 1u));   // (all lanes take part: see tk_encode_wave)
        const uint32_t q = ticket / 64u;
        if (q >= n_todo) break;
        const uint64_t d = (uint64_t)wv_first(a.todo_list[q]);
        const uint64_t s0 = wv_first64(a.doc_offs[d]), s1 = wv_first64(a.doc_offs[d + 1]);
        uint32_t* out = a.staging + s0 + 2 * d;
        const uint32_t base = a.add_bos ? 1u : 0u;
        if (a.add_bos && lane == 0) out[0] = t.bos_id;
        uint64_t w0 = s0;
        while (w0 < s1) {
            const uint64_t e = wv_first64(a.pattern ? tk_match_end2(t, a.bytes, w0, s1) : tk_match_end(t, a.bytes, w0, s1));
            const uint32_t len = (uint32_t)(e - w0);
            uint32_t* dst = out + base + (w0 - s0);
            const uint32_t r = tk_piece_lookup(a, pw, lane, w0, e);
            uint32_t cur = 0;
            if (r != TK_RANK_MAX) {
                if (lane == 0) dst[0] = r + t.num_special;
                cur = 1;
            } else if (const uint32_t how = tk_piece_is_long(a, lane, w0, e)) {
                // 1: repetitive (few distinct pairs: a rank has thousands of occurrences) -> the compacting rounds (kind 0);
                // 2: many distinct pairs (a dozen occurrences per rank) and long enough -> the lazy rounds (kind 1)
                const uint32_t kind = how - 1u;
                if (lane == 0) {
                    const uint32_t slot = wv_atomic_add(a.long_job_count, 1u);
                    if (slot < a.long_job_cap) {
                        TkLongJob j;
                        j.doc = (uint32_t)d; j.off = (uint32_t)(w0 - s0); j.len = len; j.pad = kind;
                        a.long_jobs[slot] = j;
                    } else if (a.defer_count) {
                        // cannot happen with the host's sizing (a job is at least long_min bytes of text); if it ever does the
                        // piece's slots stay unwritten, so the call must fail loudly: an error word the host checks
                        *a.defer_count = 0xDEADu;
                    }
                }
                cur = len;                                         // (its slots are written by tk_long_merge_kernel)
            } else {
                tk_piece_merge_coop(a, lane, w0, e, dst, cur, scratch);
            }
            for (uint32_t k = cur + (uint32_t)lane; k < len; k += 64u) dst[k] = TKL_HOLE;
            w0 = e;
        }
    }
}

// one kernel for both kinds of job (a workgroup takes whatever comes next: the long jobs of both kinds run side by side);
// the two forms share the block's LDS
__global__ __launch_bounds__(TKL_THREADS) void tk_long_merge_kernel(TkEncodeArgs a) {
    __shared__ TksShared LS;
    TklShared& L = *reinterpret_cast<TklShared*>(&LS);
    static_assert(sizeof(TklShared) <= sizeof(TksShared), "the compacting form's LDS fits inside the lazy form's");
    uint32_t* scratch = a.scratch + (size_t)blockIdx.x * a.scratch_words_per_wave;
    const uint32_t n_jobs = *a.long_job_count < a.long_job_cap ? *a.long_job_count : a.long_job_cap;
    const uint32_t base = a.add_bos ? 1u : 0u;
    for (;;) {
        if (threadIdx.x == 0) LS.doc = atomicAdd(a.work_counter, 1u);
        __syncthreads();
        const uint32_t q = LS.doc;
        __syncthreads();                                           // everybody has the ticket before it is overwritten
        if (q >= n_jobs) break;                                    // block-uniform
        const TkLongJob j = a.long_jobs[q];
        const uint64_t s0 = a.doc_offs[j.doc];
        uint32_t* out = a.staging + s0 + 2ull * j.doc + base + j.off;
        uint32_t n;
        if (j.pad == 0u) n = tkl_block_merge(a.t, a.bytes + s0 + j.off, j.len, scratch, out, L);     // repetitive: compacting rounds
        else n = tks_block_merge(a.t, a.bytes + s0 + j.off, j.len, scratch, out, LS);               // many distinct pairs: lazy rounds
        for (uint32_t i = n + threadIdx.x; i < j.len; i += TKL_THREADS) out[i] = TKL_HOLE;
    }
}

__global__ __launch_bounds__(256) void tk_long_compact_kernel(TkEncodeArgs a) {
    const int lane = wv_lane();
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    const uint32_t n_todo = a.n_todo_dev ? *a.n_todo_dev : a.n_todo;
    for (uint64_t q = wave; q < n_todo; q += n_waves) {
        const uint64_t d = a.todo_list[q];
        const uint64_t s0 = a.doc_offs[d], s1 = a.doc_offs[d + 1];
        uint32_t* out = a.staging + s0 + 2 * d + (a.add_bos ? 1u : 0u);
        const uint32_t len = (uint32_t)(s1 - s0);
        uint32_t wr = 0;
        // four groups of 64 slots per step, their loads issued together (a 32 KiB document is a chain of 128 such steps).  The
        // slots written lie at or below the ones just read and the next step reads above them: no load ever meets a store of
        // its own wave that it could overtake; the ballots order a step's loads before its stores.
        for (uint32_t k0 = 0; k0 < len; k0 += 256u) {
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t k = k0 + 64u * (uint32_t)q + (uint32_t)lane;
                v[q] = k < len ? out[k] : TKL_HOLE;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t keep = wv_ballot(v[q] != TKL_HOLE);
                if (v[q] != TKL_HOLE) out[wr + (uint32_t)tk_popc64(keep & tk_lowmask(lane))] = v[q];
                wr += (uint32_t)tk_popc64(keep);
            }
        }
        if (lane == 0) {
            if (a.add_eos) out[wr] = a.t.eos_id;
            a.counts[d] = wr + (a.add_bos ? 1u : 0u) + (a.add_eos ? 1u : 0u);
        }
    }
}

// the long list (documents pass 2 handed on): walk, merge the jobs, compact.  scratch: args.scratch_words_per_wave words
// per walking wave / per merging workgroup.
hipError_t tk_launch_encode_long(const TkEncodeArgs& args, uint32_t n_walk_waves, uint32_t n_merge_blocks, hipStream_t s) {
    if (args.n_todo == 0 && !args.n_todo_dev) return hipSuccess;
    hipLaunchKernelGGL(tk_long_walk_kernel, dim3((n_walk_waves + 3) / 4), dim3(256), 0, s, args);
    return hipGetLastError();
}
hipError_t tk_launch_encode_long_merge(const TkEncodeArgs& args, uint32_t n_merge_blocks, uint32_t n_compact_blocks, hipStream_t s) {
    if (args.n_todo == 0 && !args.n_todo_dev) return hipSuccess;
    hipLaunchKernelGGL(tk_long_merge_kernel, dim3(n_merge_blocks), dim3(TKL_THREADS), 0, s, args);
    hipLaunchKernelGGL(tk_long_compact_kernel, dim3(n_compact_blocks), dim3(256), 0, s, args);
    return hipGetLastError();
}
