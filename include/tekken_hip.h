/*
 * tekken_hip.h -- C ABI of the MI355X-native batch tokenization path for tekken-rs.
 *
 * The reference has no FFI seam today; this boundary is inserted at its single call into
 * the third-party BPE engine:
 *
 *     let (tokens, _) = self.tekkenizer.encode(text, &HashSet::new());   src/tekkenizer.rs:384-386
 *     CoreBPE::new(mergeable_ranks.clone(), special_tokens, pattern)      src/tekkenizer.rs:125
 *
 * plus the id shift and BOS/EOS insertion of src/tekkenizer.rs:390-402, which are fused
 * into the device emit.  Everything is plain pointers and sizes; no C++ or torch types.
 * The Rust binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: a context serialises its calls internally (one HIP stream + mutex), so it can
 * back the reference's `&self` / `Sync` Tekkenizer (tests/test_tokenizer_output.rs:5-12).
 */
#ifndef TEKKEN_HIP_H
#define TEKKEN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes; the Rust shim maps them onto TokenizerError (src/errors.rs:23-59) ---- */
#define TK_OK 0
#define TK_ERR_INVALID_CONFIG (-1)   /* -> TokenizerError::InvalidConfig   (errors.rs:45-46) */
#define TK_ERR_RUNTIME (-2)          /* -> TokenizerError::Tokenizers(msg) (errors.rs:37-38) */
#define TK_ERR_INVALID_UTF8 (-3)     /* -> TokenizerError::Tokenizers(msg); &str can never trigger it */
#define TK_ERR_NO_DEVICE (-4)        /* -> TokenizerError::Tokenizers(msg): no gfx950 device / HIP missing */
#define TK_ERR_INVALID_ARG (-5)      /* -> TokenizerError::InvalidConfig */
#define TK_ERR_IO (-6)               /* -> TokenizerError::Io              (errors.rs:25-26) */
#define TK_ERR_JSON (-7)             /* -> TokenizerError::Json            (errors.rs:29-30) */
#define TK_ERR_BASE64 (-8)           /* -> TokenizerError::Base64          (errors.rs:33-34) */
#define TK_ERR_TOKEN_NOT_FOUND (-9)  /* -> TokenizerError::TokenNotFound   (errors.rs:49-50) */
#define TK_ERR_SPECIAL_POLICY (-10)  /* -> TokenizerError::SpecialTokenPolicy (errors.rs:53-54) */

#define TK_POLICY_IGNORE 0  /* SpecialTokenPolicy::Ignore  (src/special_tokens.rs:128-136) */
#define TK_POLICY_KEEP 1    /* SpecialTokenPolicy::Keep  */
#define TK_POLICY_RAISE 2   /* SpecialTokenPolicy::Raise */

typedef struct tk_ctx tk_ctx;

/* ------------------------------------------------------------------------------------------
 * Engine level: replaces CoreBPE::new / CoreBPE::encode.
 * ---------------------------------------------------------------------------------------- */

/* Replaces `CoreBPE::new(mergeable_ranks, {}, pattern)` (src/tekkenizer.rs:122-126).
 * token_bytes/token_offsets: rank i has bytes token_bytes[token_offsets[i] .. token_offsets[i+1]);
 * the table must already satisfy the checks of reload_mergeable_ranks (src/tekkenizer.rs:776-816)
 * -- the checks are repeated and TK_ERR_INVALID_CONFIG is returned if they fail.
 * bos_id/eos_id are the FINAL ids of "<s>" / "</s>" (src/tekkenizer.rs:286-297).
 * The split pattern is the literal of src/tekkenizer.rs:123 (the JSON `pattern` is ignored by the
 * reference, :74).  Inputs are copied.  device_id: HIP device ordinal. */
int tk_ctx_create(const uint8_t* token_bytes, const uint32_t* token_offsets, uint32_t n_ranks,
                  uint32_t num_special_tokens, uint32_t bos_id, uint32_t eos_id, int device_id,
                  tk_ctx** out_ctx);

/* Frees device and host state.  NULL is a no-op. */
void tk_ctx_destroy(tk_ctx* ctx);

/* Message of the last failing call on this context (or of the last failing tk_ctx_create /
 * tk_tokenizer_* constructor on this thread when ctx == NULL).  Never NULL. */
const char* tk_last_error(const tk_ctx* ctx);

typedef struct tk_result {
    uint32_t* ids;      /* n_ids final token ids (shifted, BOS/EOS in place), document order */
    uint64_t* offsets;  /* n_docs + 1 entries: document d owns ids[offsets[d] .. offsets[d+1]) */
    uint64_t n_ids;
    uint64_t n_docs;
} tk_result;

/* Batch form of `Tekkenizer::encode(text, add_bos, add_eos)` (src/tekkenizer.rs:378-405):
 * document d = bytes[doc_offsets[d] .. doc_offsets[d+1]), for every d exactly what the reference
 * returns for that &str.  Host buffers in, host buffers out (pinned, owned by the library until
 * tk_free_result).  Each document must be valid UTF-8 (a Rust &str always is); pass
 * validate_utf8 != 0 to have that checked (TK_ERR_INVALID_UTF8). */
int tk_encode_batch(tk_ctx* ctx, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs,
                    int add_bos, int add_eos, int validate_utf8, tk_result* out);
void tk_free_result(tk_result* r);

/* `Tekkenizer::encode(text, add_bos, add_eos)` for ONE document with a caller-owned output -- the reference's own call shape
 * (src/tekkenizer.rs:378-405: one &str in, one Vec<u32> out).  No allocation; a text of up to 64 KiB is ONE kernel launch
 * (a single workgroup splits, looks up, merges, and packs; the kernel reads the text from, and writes the ids to, mapped
 * pinned host memory), longer ones take the batch pipeline.  ids_capacity >= len + 2 always suffices;
 * TK_ERR_INVALID_ARG if the ids do not fit (*n_ids then holds the count needed).  Batches of up to 1024 documents / 64 KiB
 * handed to tk_encode_batch take the same one-launch path. */
int tk_encode_one(tk_ctx* ctx, const uint8_t* text, uint64_t len, int add_bos, int add_eos, uint32_t* ids_out,
                  uint64_t ids_capacity, uint64_t* n_ids);
/* Calls on this context served by the one-launch path so far (diagnostics / tests). */
uint64_t tk_small_path_calls(const tk_ctx* ctx);
/* Documents so far whose long piece (>= 1 KiB, not a vocabulary key) was merged in rounds by a whole workgroup
 * (csrc/tk_long.hip) instead of step by step by one wave (diagnostics / tests; TK_LONG_MIN overrides the threshold). */
uint64_t tk_round_path_docs(const tk_ctx* ctx);
/* Pieces of 65..256 bytes of the LAST batch that stayed on the flat path as records (csrc/tk_flat_impl.h step 6) instead of
 * handing their documents to the per-document kernels (diagnostics / tests; TK_FLAT_LONG=0 at context creation switches
 * the path off, TK_FLAT_LONG128=0 its one-lane-per-piece merge). */
uint64_t tk_long_piece_records(const tk_ctx* ctx);
/* 2048-byte regions of the LAST batch that held a piece of more than 64 bytes and went through the CUT instantiation of the
 * flat kernel (csrc/tk_flat_impl.h step 4b: such a piece is cut into fragments wherever no vocabulary token can span the
 * boundary, and the fragments merge independently -- exact; diagnostics / tests; TK_FLAT_CUT=0 switches the cuts off). */
uint64_t tk_cut_chunks(const tk_ctx* ctx);
/* Host waits of the LAST batch on the flat pipeline: 2 -- the list of the documents the flat kernel handed back (made on a
 * second stream beside the merge kernels), then the result; 3 only when a long-piece record flagged a document late
 * (diagnostics / tests; TK_TAIL=serial at context creation runs the tail behind the merge kernels on the one stream). */
uint64_t tk_last_host_syncs(const tk_ctx* ctx);

/* Memo of merged pieces (no reference equivalent; `encode` is a pure function of the text, src/tekkenizer.rs:384-386, and stays
 * one): a device table {piece of 2..16 bytes that is no vocabulary key -> the <= 5 ids the byte-pair merge gives it}, read by the
 * split + look-up kernel, filled behind the merge kernel, visible from the NEXT call on the context.  Text repeats its unknown
 * words; a hit costs one 32-byte gather instead of a chain of ~20 dependent PAIR probes.  An entry holds the exact key and the
 * exact result: the table can change how long a call takes, never an id (tests/test_gpu_parity.py::test_memo_*).
 *   log2_entries  0 = off (the table is freed); 10..26: 2^n entries of 32 bytes (default 24 = 512 MB of a 288 GB part, plus 64 MB for the log of a call's new entries; TK_MEMO_LOG2 --
 *                 measured on the held-out shape: 2^20 entries 0.66 of the look-ups hit, 2^22 0.77, 2^24 0.85: the table is direct-mapped)
 *   policy        0 = adaptive: after two calls in a row that hit less than once per 160 bytes of text or less than three times in
 *                 ten look-ups (text with few unknown pieces, or whose unknown pieces never come back) the table is left
 *                 alone for 30 calls; a call of under 1 MB of text never uses it (a context that only sees such calls never
 *                 allocates it).  1 = always on (TK_MEMO_POLICY=always)
 * tk_ctx_memo_clear empties the table.  tk_memo_stats: look-ups (pieces of 2..16 bytes that missed the vocabulary) and hits of
 * the last call and since the context was created, and whether the last call used the table. */
int tk_ctx_set_memo(tk_ctx* ctx, int log2_entries, int policy);
int tk_ctx_memo_clear(tk_ctx* ctx);
int tk_memo_stats(const tk_ctx* ctx, uint64_t* lookups_last, uint64_t* hits_last, uint64_t* lookups_total, uint64_t* hits_total,
                  int* active_last);

/* Opt-in: honour the `pattern` of Mistral's tekken.json -- case-aware words
 * (`HelloWorld` -> `Hello`, `World`), single digits, `/` absorbed after punctuation; literal in reference
 * tests/test_small_vocab.rs:62 -- instead of the pattern the reference hard-codes and always uses
 * (src/tekkenizer.rs:74,123).  mode 0 (default) = the reference's behaviour, 1 = the JSON pattern, on its own
 * instantiation of the same kernels (ASCII text at the default pipeline's rate). */
int tk_ctx_set_pattern(tk_ctx* ctx, int mode);

/* Streaming / pipelined ingestion (same results as tk_encode_batch, which is what the reference's
 * `encode` returns per document).  The batch is cut into slices of whole documents (about slice_bytes of text each,
 * 0 = default 32 MiB) that go through a three-stage pipeline on three HIP streams: host->device copy of slice k+1, the
 * kernels of slice k, device->host copy of the ids of slice k-1 (double-buffered device staging).  The caller owns all
 * four host buffers; when they come from tk_host_alloc (pinned memory) every copy is an asynchronous DMA and the three
 * stages overlap -- pageable buffers work too, at the runtime's staged-copy rate.
 *   ids_out       capacity ids_capacity; doc_offsets[n_docs] + 2 * n_docs always suffices (a document produces at most
 *                 one id per byte, plus BOS / EOS)
 *   offsets_out   n_docs + 1 entries
 * TK_ERR_INVALID_ARG if the ids do not fit (*n_ids then holds the count reached when it was noticed). */
void* tk_host_alloc(size_t bytes);      /* pinned host memory (NULL on failure) */
void tk_host_free(void* p);
int tk_encode_batch_pipelined(tk_ctx* ctx, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs,
                              int add_bos, int add_eos, uint64_t slice_bytes, uint32_t* ids_out, uint64_t ids_capacity,
                              uint64_t* offsets_out, uint64_t* n_ids);

/* Same computation with inputs already resident in HBM (hipMalloc'ed on the context's device):
 * d_bytes = n_bytes packed text bytes, d_doc_offsets = n_docs+1 uint64 (non-decreasing, [0] = 0, [n_docs] = n_bytes:
 * not checked on this entry -- tk_encode_batch_device_ex checks).  Work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = HIP's null stream, so the work is ordered after whatever the
 * caller already enqueued there) and the call returns after the stream has drained.  *d_ids / *d_out_offsets are device buffers owned by the context, valid
 * until the next call on it; *n_ids = total ids. */
int tk_encode_batch_device(tk_ctx* ctx, const void* d_bytes, const void* d_doc_offsets, uint64_t n_docs,
                           uint64_t n_bytes, int add_bos, int add_eos, void* hip_stream, void** d_ids,
                           void** d_out_offsets, uint64_t* n_ids);

/* The same with the checks tk_encode_batch makes for host callers, on the device (one small kernel and one host wait each, before
 * anything else runs): TK_CHECK_OFFSETS -- d_doc_offsets[0] == 0, non-decreasing, [n_docs] == n_bytes, else TK_ERR_INVALID_ARG;
 * TK_CHECK_UTF8 -- every document is well-formed UTF-8 on its own (a Rust &str always is; this includes "no document starts inside
 * a code point"), else TK_ERR_INVALID_UTF8; implies the offsets check.  checks = 0 is tk_encode_batch_device. */
#define TK_CHECK_OFFSETS 1
#define TK_CHECK_UTF8 2
int tk_encode_batch_device_ex(tk_ctx* ctx, const void* d_bytes, const void* d_doc_offsets, uint64_t n_docs, uint64_t n_bytes,
                              int add_bos, int add_eos, int checks, void* hip_stream, void** d_ids, void** d_out_offsets,
                              uint64_t* n_ids);

/* ---- decode: batch form of Tekkenizer::decode (src/tekkenizer.rs:436-560) ----
 * The engine needs the special-token strings for TK_POLICY_KEEP: entry i is the string of the special token
 * at POSITION i of the reference's all_special_tokens vector (src/tekkenizer.rs:108-116, 536-540);
 * n must equal num_special_tokens.  Copied. */
int tk_ctx_set_special_tokens(tk_ctx* ctx, const uint8_t* strings_blob, const uint32_t* string_offsets, uint32_t n);

typedef struct tk_text_result {
    uint8_t* bytes;     /* concatenated UTF-8 text of all documents */
    uint64_t* offsets;  /* n_docs + 1: document d is bytes[offsets[d] .. offsets[d+1]) */
    uint64_t n_bytes;
    uint64_t n_docs;
} tk_text_result;

/* For every document d (ids[id_offsets[d] .. id_offsets[d+1])) exactly the String that
 * Tekkenizer::decode(ids_d, policy) returns.  If ANY document would make the reference return Err, the call
 * fails as a whole and *bad_doc (optional) receives the first such document:
 *   TK_ERR_SPECIAL_POLICY  a special id under TK_POLICY_RAISE          (src/tekkenizer.rs:531-535)
 *   TK_ERR_RUNTIME         an id outside the vocabulary, or a non-special run that is not valid UTF-8
 *                          (CoreBPE::decode -> TokenizerError::Tokenizers, src/tekkenizer.rs:552-555) */
int tk_decode_batch(tk_ctx* ctx, const uint32_t* ids, const uint64_t* id_offsets, uint64_t n_docs, int policy,
                    tk_text_result* out, uint64_t* bad_doc);
void tk_free_text_result(tk_text_result* r);
/* Same with ids resident in HBM; outputs are context-owned device buffers valid until the next call. */
int tk_decode_batch_device(tk_ctx* ctx, const void* d_ids, const void* d_id_offsets, uint64_t n_docs, uint64_t n_ids,
                           int policy, void* hip_stream, void** d_bytes, void** d_out_offsets, uint64_t* n_bytes,
                           uint64_t* bad_doc);

/* 18-bit wire format of token ids for the multi-GPU gather (no reference equivalent: the reference is one process on a
 * CPU; BASELINE north_star asks for "a single RCCL gather of token-id buffers over xGMI").  The link into the
 * gathering GPU bounds the job, and an id below 2^18 -- every Tekken vocabulary -- travels as 2.25 bytes instead of 4.
 * All pointers are device memory of the context's device; the work is enqueued on hip_stream and NOT waited for
 * (tk_pack_ids18_device returns after the stream has drained only because it has to report an id >= 2^18 as
 * TK_ERR_INVALID_ARG).  tk_ids18_bytes(n) = size of the packed form of n ids. */
uint64_t tk_ids18_bytes(uint64_t n_ids);
int tk_pack_ids18_device(tk_ctx* ctx, const void* d_ids, uint64_t n_ids, void* d_packed, void* hip_stream);
int tk_unpack_ids18_device(tk_ctx* ctx, const void* d_packed, uint64_t n_ids, void* d_ids, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Node level: every GPU of one node behind ONE call (no reference equivalent: the reference is one thread on a CPU;
 * BASELINE north_star "shards ... across the 8xMI355X node with a single RCCL gather of token-id buffers over xGMI";
 * SURVEY section 8b `ctx_create(.., device_ids[], n_devices, ..)`).  One process; per device one context (tables
 * replicated), one stream, one host thread for the life of the node.  tk_node_encode_batch cuts the batch into contiguous runs of WHOLE
 * documents with balanced bytes, every device tokenizes its run, and the id buffers are gathered on device_ids[0] with
 * direct peer -> root RCCL transfers inside one ncclGroupStart / ncclGroupEnd (18 bits per id on the wire when every id
 * fits) -- for every document exactly what Tekkenizer::encode returns (src/tekkenizer.rs:378-405), in document order.
 * device_ids must be distinct (TK_ERR_INVALID_ARG otherwise).  RCCL is opened with dlopen only when n_devices > 1.
 * The result is released with tk_free_result. */
typedef struct tk_node tk_node;
int tk_node_create(const uint8_t* token_bytes, const uint32_t* token_offsets, uint32_t n_ranks, uint32_t num_special_tokens,
                   uint32_t bos_id, uint32_t eos_id, const int* device_ids, int n_devices, tk_node** out_node);
void tk_node_destroy(tk_node* node);
/* node == NULL: the last failing tk_node_create on this thread */
const char* tk_node_last_error(const tk_node* node);
int tk_node_encode_batch(tk_node* node, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, int add_bos,
                         int add_eos, tk_result* out);
/* The same with CALLER-OWNED host buffers: ids_out[ids_capacity], offsets_out[n_docs + 1]; *n_ids_out = ids written (also set
 * when ids_capacity is too small: TK_ERR_INVALID_ARG, nothing written).  With every buffer from tk_host_alloc (pinned) the
 * copies up and down are asynchronous DMAs and nothing is allocated, pinned or copied on the host per call -- the form a
 * host that encodes batch after batch should use (the Rust shim's encode_batch_into). */
int tk_node_encode_batch_pinned(tk_node* node, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, int add_bos,
                                int add_eos, uint32_t* ids_out, uint64_t ids_capacity, uint64_t* offsets_out, uint64_t* n_ids_out);
int tk_node_n_devices(const tk_node* node);
/* Device timings of the last tk_node_encode_batch: the slowest device's tokenization pipeline, and the exchange on the root
 * (first receive posted .. offsets rebased), milliseconds. */
int tk_node_last_timing(const tk_node* node, float* kernels_ms_max, float* gather_ms);
/* How the last batch was cut: text bytes and ids of every device's run (up to `cap` entries each; either pointer may be NULL).
 * Returns the number of devices.  Runs are contiguous whole documents with balanced BYTES: no run exceeds the mean by more than
 * one document. */
int tk_node_last_shards(const tk_node* node, uint64_t* shard_bytes, uint64_t* shard_ids, int cap);

/* Device timings of the last tk_encode_batch* call, from HIP events on the stream the kernels
 * ran on: whole pipeline and the dominant encode kernel alone (milliseconds). */
int tk_last_timing(const tk_ctx* ctx, float* pipeline_ms, float* encode_kernel_ms);
/* ... and the span from the end of that kernel to the end of the merge kernels (the scans in between included): on text with
 * many pieces outside the vocabulary (BASELINE configs[2]) the merge kernels, not the encode kernel, are the longest part. */
float tk_last_merge_ms(const tk_ctx* ctx);

/* Counters of the last call: documents handled by the long-piece path (pass 2), and documents the flat
 * chunk-per-wave kernel handed back to the per-document kernels (non-ASCII, very long runs / pieces). */
int tk_last_stats(const tk_ctx* ctx, uint64_t* n_long_docs, uint64_t* n_handed_back);

/* Pre-tokenization split only (vocab-free): out_is_start[i] = 1 iff a piece starts at byte i of
 * the packed buffer (host in / host out).  Debug / parity entry for the split rules. */
int tk_split_batch(tk_ctx* ctx, const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs,
                   uint8_t* out_is_start);

/* ------------------------------------------------------------------------------------------
 * Tokenizer level: the host-side mirror of tekken::tekkenizer::Tekkenizer, exported so that
 * non-C++ hosts (the Rust shim, Python tests) can drive loader + encode + decode.
 * ---------------------------------------------------------------------------------------- */
typedef struct tk_tokenizer tk_tokenizer;

/* Tekkenizer::from_file (src/tekkenizer.rs:222-248).  device_id < 0 => host-only object
 * (loader, decode and accessors work; encode returns TK_ERR_NO_DEVICE). */
int tk_tokenizer_from_file(const char* path, int device_id, tk_tokenizer** out);
/* Same, from an in-memory tekken.json document. */
int tk_tokenizer_from_json(const char* json, size_t json_len, int device_id, tk_tokenizer** out);
void tk_tokenizer_destroy(tk_tokenizer* t);
/* Error text of the last failing call on t (t == NULL: last failing constructor on this thread). */
const char* tk_tokenizer_last_error(const tk_tokenizer* t);

/* Tekkenizer::encode (src/tekkenizer.rs:378-405) for one document; *ids is malloc'ed, free with
 * tk_free_ids. */
int tk_tokenizer_encode(tk_tokenizer* t, const char* text, size_t len, int add_bos, int add_eos,
                        uint32_t** ids, size_t* n_ids);
/* Batch addition (no reference equivalent; see tk_encode_batch). */
int tk_tokenizer_encode_batch(tk_tokenizer* t, const uint8_t* bytes, const uint64_t* doc_offsets,
                              uint64_t n_docs, int add_bos, int add_eos, tk_result* out);
void tk_free_ids(uint32_t* ids);
/* Opt-in (row f-3): honour the `pattern` of the loaded tekken.json instead of ignoring it like the reference does
 * (src/tekkenizer.rs:74).  Only Mistral's pattern string is known; any other is refused with TK_ERR_INVALID_CONFIG.
 * See tk_ctx_set_pattern. */
int tk_tokenizer_set_honour_pattern(tk_tokenizer* t, int honour);

/* Batch decode on the GPU (needs a device-backed tokenizer); see tk_decode_batch. */
int tk_tokenizer_decode_batch(tk_tokenizer* t, const uint32_t* ids, const uint64_t* id_offsets, uint64_t n_docs,
                              int policy, tk_text_result* out, uint64_t* bad_doc);

/* Tekkenizer::decode (src/tekkenizer.rs:436-443); *text is malloc'ed (not NUL terminated beyond
 * *len, but a trailing NUL is added for convenience), free with tk_free_text. */
int tk_tokenizer_decode(tk_tokenizer* t, const uint32_t* ids, size_t n_ids, int policy, char** text,
                        size_t* len);
void tk_free_text(char* text);
/* Tekkenizer::decode_all (src/tekkenizer.rs:463-560): the segments decode() joins, one per run of special / non-special
 * ids -- *text holds them back to back (tk_free_text), seg_ends[i] (tk_free_offsets) is the END of segment i in *text. */
int tk_tokenizer_decode_all(tk_tokenizer* t, const uint32_t* ids, size_t n_ids, int policy, char** text, uint64_t** seg_ends,
                            size_t* n_segments);
void tk_free_offsets(uint64_t* offsets);
/* Tekkenizer::vocab (src/tekkenizer.rs:348-350): the piece string of every id, specials included, back to back in *text;
 * ends[id] is where the piece of `id` ends (vocab_size entries). */
int tk_tokenizer_vocab(tk_tokenizer* t, char** text, uint64_t** ends, size_t* n_pieces);

/* Accessors (src/tekkenizer.rs:260-350, 574-600, 617-695). */
uint32_t tk_tokenizer_vocab_size(const tk_tokenizer* t);
uint32_t tk_tokenizer_num_special_tokens(const tk_tokenizer* t);
const char* tk_tokenizer_version(const tk_tokenizer* t); /* "v3" | "v7" | "v11" | "v13" */
int tk_tokenizer_control_token(tk_tokenizer* t, const char* name, uint32_t* id); /* get_control_token */
int tk_tokenizer_is_special(const tk_tokenizer* t, uint32_t id);
int tk_tokenizer_is_byte(const tk_tokenizer* t, uint32_t id);
int tk_tokenizer_id_to_piece(tk_tokenizer* t, uint32_t id, char** text, size_t* len);
int tk_tokenizer_id_to_byte_piece(tk_tokenizer* t, uint32_t id, int policy, uint8_t** bytes, size_t* len);
/* config.pattern of the loaded tekken.json (src/config.rs:38-49; the reference parses and ignores it, src/tekkenizer.rs:74),
 * and whether the object came from a TK_TABLE_CACHE_DIR side file (row f-2) rather than from parsing the JSON. */
const char* tk_tokenizer_json_pattern(const tk_tokenizer* t);
int tk_tokenizer_from_cache(const tk_tokenizer* t);
/* The engine context behind the tokenizer (NULL for host-only objects). */
tk_ctx* tk_tokenizer_ctx(tk_tokenizer* t);
/* The validated rank table (for building an oracle beside it in tests). */
int tk_tokenizer_rank_table(const tk_tokenizer* t, const uint8_t** blob, const uint32_t** offsets,
                            uint32_t* n_ranks);

#ifdef __cplusplus
}
#endif
#endif
