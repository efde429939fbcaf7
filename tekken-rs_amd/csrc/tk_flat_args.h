// tk_flat_args.h -- geometry and argument block of the flat path (host + device view), see tk_flat_impl.h.
#ifndef TK_FLAT_ARGS_H
#define TK_FLAT_ARGS_H
#include <stdint.h>

#include "tk_tables.h"

#define TKF_W 32                                     /* bytes per lane = bits of a lane-layout mask word */
#define TKF_REGION (64 * TKF_W)                      /* bytes loaded per chunk: 64 lanes x TKF_W bytes */
#define TKF_HL 32                                    /* left halo (look-behind context) */
#define TKF_HR 64                                    /* right halo (look-ahead, ends of the last pieces) */
#define TKF_COMMIT (TKF_REGION - TKF_HL - TKF_HR)    /* bytes committed per chunk: 1952 */
#define TKF_LONGCAP 256u                             /* a piece of 65..LONGCAP bytes keeps its document on the flat path (tk_flat_long_kernel) */
#define TKF_STRIDE (TKF_COMMIT + 64 + TKF_LONGCAP)   /* id slots per chunk: a piece may reach 63 bytes past the commit range, and the
                                                        chunk's last piece, when its end is not in the region, reserves LONGCAP slots */
/* queue records per chunk, one sub-queue per length class (2..8, 9..16, 17..32, 33..64 bytes).  A chunk owns the pieces that
   start in its commit range and, where a long piece that it cuts into fragments ends in its right halo, the fragments up to
   that end (tk_flat_impl.h, CUT instantiation): TKF_OWN bytes at most */
#define TKF_OWN (TKF_COMMIT + TKF_HR)
#define TKF_MISSOFF0 0u
#define TKF_MISSOFF1 (TKF_OWN / 2u)                                 /* at most OWN / 2 pieces of >= 2 bytes */
#define TKF_MISSOFF2 (TKF_MISSOFF1 + (TKF_OWN + 8u) / 9u)
#define TKF_MISSOFF3 (TKF_MISSOFF2 + (TKF_OWN + 16u) / 17u)
#define TKF_MISSCAP ((TKF_MISSOFF3 + (TKF_OWN + 32u) / 33u + 7u) & ~7u)
/* queue record: position in the region | length << POSBITS | id slot << (POSBITS + 7) */
#define TKF_POSBITS 11
#define TKF_REC(pos, len, slot) ((pos) | ((len) << TKF_POSBITS) | ((slot) << (TKF_POSBITS + 7)))
#define TKF_REC_POS(rec) ((rec) & ((1u << TKF_POSBITS) - 1u))
#define TKF_REC_LEN(rec) (((rec) >> TKF_POSBITS) & 127u)
#define TKF_REC_SLOT(rec) ((rec) >> (TKF_POSBITS + 7))
#define TKF_HOLE 0xFFFFFFFFu                         /* id slot reserved by a missed piece and not used */

// a piece of more than 64 bytes that the flat kernel leaves to tk_flat_long_kernel (one wave per record)
struct TkFlatLongRec {
    uint64_t pos;        // first byte of the piece in the packed stream
    uint32_t chunk;      // owning chunk: its ids go to tmp[chunk * TKF_STRIDE + slot ..]
    uint32_t slot;
    uint32_t len;        // bytes of the piece; 0: the end lies beyond the chunk's region (the sequential matcher finds it);
                         // bit 31 (TKF_LREC_FRAG): a FRAGMENT of a cut piece -- merged without the whole-piece look-up
    uint32_t reserved;   // id slots reserved for it: len, or TKF_LONGCAP when the end was not seen
};

#define TKF_LREC_FRAG 0x80000000u

struct TkFlatArgs {
    const uint8_t* bytes;        // packed text of all documents
    const uint64_t* doc_offs;    // [n_docs + 1]
    uint64_t n_docs, n_bytes, n_chunks;
    const uint32_t* first_doc;   // [n_chunks] documents that start below the chunk's loaded region
    uint32_t* tmp;               // [n_chunks * TKF_STRIDE] chunk-dense ids
    uint32_t* kcount;            // [n_chunks] id slots of the chunk (holes included)
    uint32_t* lstart;            // [n_docs] id slots of the chunk before the document's first byte
    uint32_t* miss_list;         // [n_chunks * TKF_MISSCAP] per chunk, the pieces that missed the vocabulary: TKF_REC(pos, len, slot),
                                 // sub-queue of length class k at TKF_MISSOFFk
    uint32_t* miss_count;        // [4 * n_chunks] class-major: queued pieces of class k of chunk c at [k * n_chunks + c]
    const uint64_t* miss_prefix; // [4 * n_chunks + 1] exclusive prefix sums of miss_count (the merge kernels' item order)
    uint32_t* wave_first;        // [narrow items / 64 + 1] sub-queue that holds item 64 w of the narrow classes (tk_merge_wavefirst_kernel)
    uint32_t* wave_first_wide;   // [wide items / 64 + 1] the same for the wide classes (items counted from the first wide one)
    uint32_t* holes;             // [n_docs] reserved id slots the document's missed pieces did not use
    uint32_t* flags;             // [n_docs] 1 = the document is redone by the per-document kernel
    TkFlatLongRec* long_recs;    // [long_cap] pieces of 65..TKF_LONGCAP bytes (what tk_flat_long_kernel reads)
    uint32_t* long_count;        // records appended (may exceed long_cap: the surplus pieces hand their documents back)
    uint32_t long_cap;
    int long_merge128;           // records of 65..128 bytes that are no vocabulary keys are marked for tk_flat_long128_kernel (one lane per
                                 // piece); 0: the single-wave merge takes them too
    const uint32_t* long_ctl;    // what the flat kernel reads, in its rare path only, so that the three values above cost it no
                                 // scalar registers: device words {long_recs lo, hi, long_cap, cut_list lo, hi}, the record counter
                                 // five words BELOW them and the cut-chunk counter four words below (the context's counter block:
                                 // counters 11 and 12, control words at 16..20).  NULL: a piece of more than 64 bytes hands its
                                 // document back
    uint32_t* cut_list;          // [n_chunks] chunks with a piece of more than 64 bytes: left by tk_flat_kernel to the CUT instantiation
                                 // (tk_flat_cut_kernel), which cuts such pieces into fragments where no token can span (NULL: no cuts)
    const uint32_t* cut_count;   // entries of cut_list (device counter 12)
    uint32_t* late_list;         // [n_docs] documents that a long-piece record flags (an open piece beyond TKF_LONGCAP) after the list of
                                 // the handed-back documents was made; NULL: the flag alone (the list is made afterwards)
    uint32_t* late_count;
    tk_memo_entry* memo_tab;     // memo of merged pieces (tk_hash.h MEMO; NULL: off), memo_mask + 1 entries: read by the flat kernel for a piece
                                 // of 2..16 bytes that is no vocabulary key -- a hit reserves exactly its ids' slots and stores them: no queue
                                 // entry, no holes
    uint32_t memo_mask;          // entries - 1
    int memo_probe;              // 0: this call only FILLS the table (its first call: an empty table answers nothing, and a look-up is a
                                 // dependent load in the flat kernel's miss path) -- the flat kernel runs without the look-up
    uint32_t memo_epoch;         // number of this call on the context (> 0, rising)
    uint32_t* memo_hits;         // device counter: memo hits of the call (the host's hit-rate policy); may be NULL
    tk_memo_entry* memo_log;     // what the merge kernel merged into <= TK_MEMO_MAXIDS ids in this call (every record is the entry it will become): merge wave w
                                 // owns records [w * memo_log_per_wave, (w + 1) * memo_log_per_wave) and leaves their number in
                                 // memo_log_counts[w]; tk_memo_commit_kernel moves them into the table
    uint32_t* memo_log_counts;   // [memo_log_waves]
    uint32_t memo_log_per_wave, memo_log_waves;
    uint8_t* dbg_starts;         // optional: per-byte piece-start flags
    int pattern;                 // 0: the reference's hard-coded pattern; 1: the JSON pattern of tekken.json (row f-3, opt-in)
    int dbg_ablate;              // timing-only ablation bits (TK_DEBUG_ABLATE): 1 no probes, 2 no merges, 4 no id stores,
                                 // 8 stop after the split rules, 16 stop after the classification
    TkTablesView t;
};

#endif
