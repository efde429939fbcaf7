/*
 * tk_oracle.h -- CPU oracle for the tekken-rs `Tekkenizer::encode` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (tekken-rs_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" at token-id level.  The arithmetic of this path lives
 * in the third-party crate tiktoken-rs ^0.7.0 (reference Cargo.toml:40; no Cargo.lock,
 * source not under /root/reference), the reference cannot be built here (no Rust
 * toolchain) and the only asset that pins its golden id vectors
 * (tests/assets/tekken.json) is absent from the mount.  What IS pinned: the split
 * behaviour against an independent engine (Python `regex` on the literal pattern of
 * src/tekkenizer.rs:123), the vocab-free facts of the reference's golden vectors
 *  and the hand-derivable small-vocab known answer (App. B.3).
 */
#ifndef TK_ORACLE_H
#define TK_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_oracle tk_oracle;

/* Build the rank table: rank i has bytes blob[offs[i] .. offs[i+1]).
 * Mirrors the FxHashMap<Vec<u8>,u32> handed to CoreBPE::new (src/tekkenizer.rs:118-126).
 * bos_id / eos_id are the final ids of "<s>" / "</s>" (src/tekkenizer.rs:286-297). */
tk_oracle* tk_oracle_new(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks,
                         uint32_t num_special, uint32_t bos_id, uint32_t eos_id);
void tk_oracle_free(tk_oracle* o);

/* Pre-tokenization split (pattern literal src/tekkenizer.rs:123).
 * Writes the byte offset of every piece start into starts[0..cap) and returns the
 * number of pieces (which may exceed cap; only cap are written).  Vocab-free. */
size_t tk_oracle_split(const uint8_t* text, size_t n, uint32_t* starts, size_t cap);

/* SURVEY section 8 row f-3 (groundwork): the split under the `pattern` string of Mistral's tekken.json (literal in
 * reference tests/test_small_vocab.rs:62), which the reference ignores (src/tekkenizer.rs:74).  Pinned against
 * Python `regex` (tests/golden/split_vectors_tekken.json).  tk_oracle_set_pattern(o, 1) makes tk_oracle_encode* use
 * it (0, the default, is the reference's behaviour).  4-bit class: 0=O 1=U(Lu|Lt) 2=W(Ll) 3=X(Lm|Lo) 4=M 5=N 6=S. */
size_t tk_oracle_split_tekken(const uint8_t* text, size_t n, uint32_t* starts, size_t cap);
void tk_oracle_set_pattern(tk_oracle* o, int pattern);
int tk_oracle_class2(uint32_t cp);

/* Tekkenizer::encode(text, add_bos, add_eos) (src/tekkenizer.rs:378-405) for ONE document.
 * Returns the number of ids (may exceed cap; only cap are written). */
size_t tk_oracle_encode(const tk_oracle* o, const uint8_t* text, size_t n, int add_bos,
                        int add_eos, uint32_t* out, size_t cap);

/* Host loop over a packed batch: doc d = bytes[offs[d] .. offs[d+1]).  out_ids must hold
 * n_bytes + 2*n_docs ids; out_offs has n_docs+1 entries.  Returns total ids.
 * n_threads <= 1 -> the single-thread loop used as the CPU baseline. */
uint64_t tk_oracle_encode_batch(const tk_oracle* o, const uint8_t* bytes, const uint64_t* offs,
                                uint64_t n_docs, int add_bos, int add_eos, uint32_t* out_ids,
                                uint64_t* out_offs, int n_threads);

/* Wall time (seconds) spent inside the last tk_oracle_encode_batch call of this process: what bench.py reports as the
 * CPU baseline (1 thread) and as cpu_baseline_nt (n_threads > 1). */
double tk_oracle_last_batch_seconds(void);

/* out[4] = {pieces, pieces that miss the vocabulary, bytes of those, ids they produce} over a packed batch. */
void tk_oracle_miss_stats(const tk_oracle* o, const uint8_t* bytes, const uint64_t* offs, uint64_t n_docs, uint64_t* out);

/* one record {hash32 of the bytes, bytes, ids} per missed piece, for tools/miss_analysis.py; returns their number */
uint64_t tk_oracle_miss_records(const tk_oracle* o, const uint8_t* bytes, const uint64_t* offs, uint64_t n_docs, uint32_t* rec, uint64_t cap);

/* 2-bit class of a code point: 0=O 1=L 2=N 3=S (tables generated from Python `regex`). */
int tk_oracle_class(uint32_t cp);

/* FNV-1a 64 over a u32 id stream (the checksum BASELINE.md names for bit-exact compares). */
uint64_t tk_oracle_fnv1a(const uint32_t* ids, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
