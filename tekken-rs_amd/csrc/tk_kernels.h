// tk_kernels.h -- host-callable launchers of the gfx950 kernels (tk_kernels.hip).
#ifndef TK_KERNELS_H
#define TK_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tekken_hip.h"
#include "tk_encode_impl_args.h"
#include "tk_flat_args.h"

// mode 0: pass 1; mode 1: pass 2 (scratch-backed, every launched wave owns a scratch slice);
// mode 2: split only; mode 3: pass 1 over args.todo_list.  n_waves = waves launched (rounded up to whole 4-wave blocks)
hipError_t tk_launch_encode(const TkEncodeArgs& args, int mode, uint32_t n_waves, hipStream_t s);

// One launch for a small batch (<= TK_SMALL_MAX_DOCS documents): encode + scan + pack by a single workgroup.  bytes /
// out_ids / out_offs / status may be mapped pinned host memory.  status[0] != 0: a document needs pass 2, nothing usable
// was written; status[1] = total ids.
#define TK_SMALL_MAX_DOCS 1024
#define TK_SMALL_MAX_BYTES (64u << 10)
#define TK_SMALL_THREADS 1024
hipError_t tk_launch_small(const TkEncodeArgs& args, uint32_t* out_ids, uint64_t* out_offs, uint32_t* status, hipStream_t s);

// Workgroup-per-document pass over args.todo_list (documents with a long piece that is not a vocabulary key, handed on by
// pass 2): the long piece is merged in rounds by 16 waves (tk_long.hip).  Every block owns a scratch slice of
// args.scratch_words_per_wave words.
hipError_t tk_launch_encode_long(const TkEncodeArgs& args, uint32_t n_walk_waves, uint32_t n_merge_blocks, hipStream_t s);
hipError_t tk_launch_encode_long_merge(const TkEncodeArgs& args, uint32_t n_merge_blocks, uint32_t n_compact_blocks, hipStream_t s);

// counts[n] (u32) -> offs[n+1] (u64, exclusive prefix sum); block_sums: workspace of
// ceil(n/2048)+1 u64.  offs[n] (= total) is also what the host reads back.
hipError_t tk_launch_scan(const uint32_t* counts, uint64_t n, uint64_t* offs, uint64_t* block_sums, hipStream_t s);

// out_ids[out_offs[d] + k] = staging[doc_offs[d] + 2*d + k] for k < counts[d]
hipError_t tk_launch_compact(const uint32_t* staging, const uint64_t* doc_offs, const uint32_t* counts,
                             const uint64_t* out_offs, uint64_t n_docs, uint32_t* out_ids, hipStream_t s);

// UTF-8 validation of every document; *d_bad receives the number of invalid documents
hipError_t tk_launch_check_offsets(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_bytes, uint32_t* d_bad, hipStream_t s);
hipError_t tk_launch_validate(const uint8_t* bytes, const uint64_t* doc_offs, uint64_t n_docs, uint32_t* d_bad,
                              hipStream_t s);

// ---- flat path (tk_flat.hip, tk_flat_impl.h): one wave per 2048-byte region of the packed stream ----
hipError_t tk_launch_flat_firstdoc(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_chunks, uint32_t* first_doc,
                                   uint32_t* flags, uint32_t* holes, uint32_t* counters16, hipStream_t s);
hipError_t tk_launch_flat(const TkFlatArgs& a, hipStream_t s);
// the long-piece records of the flat kernel (65..TKF_LONGCAP bytes): n_waves persistent waves, scratch_words words of scratch each
#define TKF_LONG_SCRATCH_WORDS 2048u
hipError_t tk_launch_flat_long(const TkFlatArgs& a, uint32_t* work_counter, uint32_t* scratch, uint32_t scratch_words, uint32_t n_waves,
                               hipStream_t s);
// flagged documents -> todo list (count in *n_todo), the longest of them in *maxlen (atomicMax: zero it first)
hipError_t tk_launch_flat_todo(const uint32_t* flags, const uint64_t* doc_offs, uint64_t n_docs, uint32_t* todo, uint32_t* n_todo,
                               uint32_t* maxlen, hipStream_t s);
// doc_info: [n_docs] 16-byte records (TkFlatDocInfo, tk_flat.hip) written by counts, read by assemble
hipError_t tk_launch_flat_counts(const uint64_t* doc_offs, uint64_t n_docs, uint64_t n_bytes, uint64_t n_chunks,
                                 const uint64_t* P, const uint32_t* lstart, const uint32_t* flags, const uint32_t* holes,
                                 uint32_t extra, uint32_t* counts, void* doc_info, int final_pass, uint32_t* n_flagged, hipStream_t s);
hipError_t tk_launch_flat_assemble(uint64_t n_docs, const void* doc_info, const uint32_t* kcount, const uint64_t* out_offs,
                                   const uint32_t* tmp, const uint32_t* staging, uint32_t* out_ids, uint32_t bos_id,
                                   uint32_t eos_id, int add_bos, int add_eos, uint64_t* total_out, const uint32_t* skip_if, hipStream_t s);
hipError_t tk_launch_merge(const TkFlatArgs& a, uint32_t* narrow_left_out, hipStream_t s);  // both merge kernels, persistent grids

hipError_t tk_launch_iota(uint32_t* out, uint64_t n, hipStream_t s);   // out[i] = i
hipError_t tk_launch_add_u64(uint64_t* p, uint64_t n, uint64_t add, hipStream_t s);   // p[i] += add

// ---- 18-bit wire format of ids for the multi-GPU gather (tk_kernels.hip) ----
hipError_t tk_launch_pack18(const uint32_t* ids, uint64_t n, void* packed, uint32_t* d_bad, hipStream_t s);
hipError_t tk_launch_unpack18(const void* packed, uint64_t n, uint32_t* ids, hipStream_t s);

// ---- decode path (tk_decode.hip) ----
#define TK_DECODE_GROUP_DOCS 16u   /* consecutive documents the decode kernels take as one stream of ids */
struct TkDecodeArgs {
    const uint32_t* ids;       // [n_ids] packed token ids of all documents
    const uint64_t* id_offs;   // [n_docs + 1]
    uint64_t n_ids, n_docs;
    uint32_t* lens;            // [n_docs] text bytes of every document (per-document length pass: the fall-back form)
    uint32_t* glens;           // [ceil(n_docs / 16)] text bytes of every group of 16 consecutive documents (tk_decode_grouplen_kernel)
    const uint64_t* goffs;     // [groups + 1] exclusive scan of glens; non-NULL: the emit kernel starts a group there and writes out_offs itself
    uint32_t group_limit;      // a group with this many ids or text bytes (< 2^31) raises err[3]: the call takes the per-document pass
    uint8_t* out_bytes;        // [total bytes]
    uint64_t* out_offs;        // [n_docs + 1] exclusive scan of lens
    uint32_t* run_bits;        // bitmap over output bytes: 1 = a run starts here (hard UTF-8 boundary)
    uint32_t* doc_hi;          // [n_docs] written by emit: the document's text holds a byte >= 0x80 (only those are validated)
    unsigned long long* err;   // [3] see tk_decode.hip
    const uint8_t* tok_blob;   // token bytes by rank
    const uint32_t* tok_offs;  // [n_ranks + 1]
    const uint8_t* tok_inline; // [n_ranks] 16-byte entries: the token's bytes (<= 15) and its length in byte 15; 0xFF there = longer, see tok_offs
    const uint8_t* tok_len8;   // [n_ranks] length of the token, 0xFF = 255 bytes or more (see tok_offs)
    const uint8_t* sp_blob;    // special token strings by POSITION (reference src/tekkenizer.rs:536-540)
    const uint32_t* sp_offs;   // [num_special + 1]
    uint32_t n_ranks, num_special;
    int policy;                // TK_POLICY_*
};
hipError_t tk_launch_decode_doclen(const TkDecodeArgs& a, hipStream_t s);
hipError_t tk_launch_decode_grouplen(const TkDecodeArgs& a, hipStream_t s);   // err[3] != ~0: a group's text reaches 4 GiB, use the per-document pass
hipError_t tk_launch_decode_emit(const TkDecodeArgs& a, hipStream_t s);
hipError_t tk_launch_decode_validate(const TkDecodeArgs& a, hipStream_t s);

// max document length over the deferred documents (atomicMax into *d_out, which must be zeroed)
hipError_t tk_launch_defer_maxlen(const uint32_t* defer_list, uint32_t n, const uint64_t* doc_offs, uint32_t* d_out,
                                  hipStream_t s);

// self-test of the wave primitives (DPP shifts, bpermute); writes 0 to *d_fail when all pass
hipError_t tk_launch_wave_selftest(uint32_t* d_fail, hipStream_t s);

#endif
