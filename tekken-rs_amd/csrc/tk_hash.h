// tk_hash.h -- hash functions and table-entry layouts shared by the host table builder
// (tk_tables.cpp) and the gfx950 kernels (tk_encode_impl.h).  Integer-only.
//
// Three lookup structures replace the FxHashMap<Vec<u8>,u32> that the reference hands to
// CoreBPE::new (reference src/tekkenizer.rs:118-126, built at :776-816):
//
//   KEY    piece bytes (2..16 bytes, exact 128-bit key + length)        -> rank
//   LONG   piece bytes (>= 17 bytes, polynomial hash, verified vs blob)  -> rank
//   PAIR   (id(A), id(B))  ->  rank(bytes(A) ++ bytes(B))                
//   PAIR2  direct 64K table for two single bytes                         -> rank
//
// KEY/LONG serve the whole-piece shortcut, PAIR/PAIR2 serve the merge loop.
//
// KEY8 / KEY16 and PAIR are CUCKOO tables: every key has exactly two candidate locations (slot / bucket h & mask
// and rotl(h, 16) & mask) -- no probe loop (with linear probing a wave waited for the longest chain of its 64 lanes).
// What a wave-wide probe costs on CDNA is the NUMBER OF SCATTERED LOAD INSTRUCTIONS (the CU's L1 takes about one lane
// per clock), so the layouts minimise that: KEY8 (pieces of 2..8 bytes, most of the text) has 16-byte entries
// {k0, k1, rank, len} -- ONE 16-byte load fetches key, rank and length; KEY16 (9..16 bytes) has 32-byte entries
// (16-byte + 8-byte load).  The builder fills first choices first (load <= 1/3), the second location is fetched only
// by the lanes whose first one mismatched.  PAIR: buckets of two 8-byte entries (one 16-byte load per bucket), both
// buckets fetched together (the merge kernel is latency-, not issue-bound), load <= 1/2.
#ifndef TK_HASH_H
#define TK_HASH_H
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TK_HD __host__ __device__ __forceinline__
#else
#define TK_HD inline
#endif

#define TK_RANK_MAX 0xFFFFFFFFu
#define TK_ID_BITS 21u                   /* ids < 2^21 so that (a,b,rank) packs into 63 bits */
#define TK_MAX_RANKS (1u << TK_ID_BITS)
#define TK_PAIR_EMPTY 0xFFFFFFFFFFFFFFFFull
#define TK_POLY_P1 0x01000193u           /* odd => invertible mod 2^32 */
#define TK_POLY_P2 0x9E3779B1u
#define TK_KEY_SPILL 0x80000000u         /* in the len word of a KEY8 / KEY16 slot: some key whose FIRST choice is this slot lives in its second
                                            choice; clear => a probe that does not match here is a definite miss (no second fetch) */

struct alignas(32) tk_key_entry {        /* 32 B, len == 0 <=> empty: pieces of 9..16 bytes, exact 128-bit key.  The first 16 bytes
                                            read like a KEY8 entry (a piece of up to 8 bytes has k2 = k3 = 0): one compare
                                            sequence serves both tables, no per-lane selects */
    uint32_t k01[2];                     /* piece bytes 0..7 little-endian */
    uint32_t rank;
    uint32_t len;                        /* 9..16 | TK_KEY_SPILL */
    uint32_t k23[2];                     /* piece bytes 8..15, zero padded */
    uint32_t pad[2];
};

struct alignas(16) tk_key8_entry {       /* 16 B, len == 0 <=> empty: pieces of 2..8 bytes */
    uint32_t k[2];                       /* piece bytes little-endian, zero padded */
    uint32_t rank;
    uint32_t len;                        /* 2..8 */
};

struct alignas(16) tk_long_entry {       /* 16 B, len == 0 <=> empty */
    uint32_t tag;                        /* second polynomial hash */
    uint32_t rank;
    uint32_t len;                        /* >= 17 */
    uint32_t blob_off;                   /* token bytes start in the blob */
};

TK_HD uint32_t tk_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

TK_HD uint32_t tk_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// Hash of a KEY piece (zero-padded 16 bytes + length).  mode 0 (default): the upper 8 bytes are folded into the lower
// ones before two multiplies + one finalizer multiply -- 32-bit multiplies are quarter rate on CDNA and this hash runs
// once per piece of text.  mode 1: one multiply per dword and the full finalizer; the table builder falls back to it
// if a vocabulary cannot be placed with mode 0 (the fold makes structured collisions possible, the cuckoo placement
// tolerates two keys per hash value but not three).
TK_HD uint32_t tk_key_hash(uint32_t mode, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len) {
    if (mode == 0u) {
        const uint32_t x = (k0 ^ tk_rotl32(k2, 13)) * 0x9E3779B1u;
        const uint32_t y = (k1 ^ tk_rotl32(k3, 17) ^ (len << 24)) * 0x85EBCA77u;
        uint32_t h = x ^ tk_rotl32(y, 15);
        h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15;
        return h;
    }
    uint32_t h = k0 * 0x9E3779B1u + k1 * 0x85EBCA77u + k2 * 0xC2B2AE3Du + k3 * 0x27D4EB2Fu + len * 0x165667B1u;
    return tk_fmix32(h);
}

TK_HD uint32_t tk_long_hash(uint32_t h1, uint32_t len) {
    return tk_fmix32(h1 + len * 0x9E3779B1u);
}

TK_HD uint32_t tk_pair_hash(uint32_t a, uint32_t b) {
    return tk_fmix32(a * 0x9E3779B1u + b * 0x85EBCA77u + 0x165667B1u);
}

/* PAIR filter: one bit per tk_pair_hash value of a stored pair (bit tk_pair_fbit(h) of a 2^TK_PAIRF_LOG2-bit map, 32 KB:
   the merge kernels keep it in LDS).  A clear bit proves that the pair is in no bucket, so the probe's gather is not
   issued; a set bit proves nothing.  (Measured, DESIGN.md section 6: 64 KB halves the false positives but costs the
   merge kernels four waves per CU, and the waves are worth more.) */
#ifndef TK_PAIRF_LOG2
#define TK_PAIRF_LOG2 18
#endif
#define TK_PAIRF_WORDS (1u << (TK_PAIRF_LOG2 - 5))
TK_HD uint32_t tk_pair_fbit(uint32_t h) { return h >> (32 - TK_PAIRF_LOG2); }

/* KEY64: whole pieces of 17..64 bytes as the FLAT kernel looks them up -- one lane per piece, the bytes as little-endian
   dwords (the last one zero padded).  The hash folds the dwords with a rotate and an exclusive or each (no multiply: 32-bit
   multiplies are quarter rate, and on text whose words are long this hash was a third of the flat kernel), starting from the
   first four dwords -- which the kernel holds in registers anyway --, and is scrambled ONCE at the end (tk_k64_slot / tag).
   A weak fold only costs false candidates: every tag match is verified against the blob.
   Entries like LONG: {tag, rank, len, blob_off}, linear probing from tk_k64_slot(ha, len), tag = tk_k64_tag(ha). */
/* In front of KEY64: one bit per tk_key_hash(first 16 bytes, length) of every token of 17..64 bytes (2^18 bits = 32 KB, stored right
   behind the table's entries).  The flat kernel has that hash of every piece anyway; a clear bit proves that the piece is no token
   -- no dword fold, no table probe -- and most long pieces of running text are none. */
#define TK_K64PRE_LOG2 18
#define TK_K64PRE_WORDS (1u << (TK_K64PRE_LOG2 - 5))
TK_HD uint32_t tk_k64_prebit(uint32_t h) { return h >> (32 - TK_K64PRE_LOG2); }
TK_HD uint32_t tk_k64_start(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
    return k0 ^ tk_rotl32(k1, 7) ^ tk_rotl32(k2, 14) ^ tk_rotl32(k3, 21);
}
TK_HD void tk_k64_step(uint32_t& ha, uint32_t w) { ha = tk_rotl32(ha, 5) ^ w; }
TK_HD uint32_t tk_k64_slot(uint32_t ha, uint32_t len) { return tk_fmix32(ha + len * 0x165667B1u); }
TK_HD uint32_t tk_k64_tag(uint32_t ha) { return tk_fmix32(ha ^ 0x85EBCA77u); }   /* other bits of the same hash: the blob compare decides */

/* MEMO: merged pieces of 2..16 bytes that are no vocabulary key -> the ids the byte-pair merge gives them (at most TK_MEMO_MAXIDS = 5:
   97 % of the missed pieces of running text; four would be 90 %).  A pure function of the bytes (reference src/tekkenizer.rs:384-386:
   encode is a function of the text), so an entry can change how long a call takes, never an id.  Direct-mapped, slot
   tk_memo_slot(tk_key_hash(key, len)) & mask, 32-byte entries:
     k[4]   the piece's bytes, zero padded (exact key)
     w4     TK_MEMO_TAG | rank4.  Between the two commit kernels of a call this word is the CLAIM word (the index of a log record:
            tk_merge_lds / tk_memo_commit_one, ONE writer per slot and call); a record index never has the tag bit
     v[3]   rank0 | rank1 << 21 | rank2 << 42 (v[0], v[1]);  v[2] = rank3 | len << 21 | n << 26   (ranks, not final ids)
   An empty entry is all zero: its length 0 matches no piece.  Readers (the flat kernels) and the writers (the merge kernel's claim stores and the commit kernel)
   never run at the same time: merge and commit kernels follow a call's flat and merge kernels on the stream, the next call's flat kernel
   follows them, and a slot has at most one writer per call -- no reader ever sees a torn entry.  A log record (tk_merge_lds) IS the
   entry it will become; the commit kernels find its slot from its key. */
#define TK_MEMO_MAXIDS 5u
#define TK_MEMO_TAG 0x80000000u
struct alignas(32) tk_memo_entry {
    uint32_t k[4];
    uint32_t w4;
    uint32_t v[3];
};
TK_HD uint32_t tk_memo_slot(uint32_t h) { return tk_rotl32(h, 11) ^ (h >> 3); }
TK_HD uint32_t tk_memo_len(uint32_t v2) { return (v2 >> 21) & 31u; }
TK_HD uint32_t tk_memo_n(uint32_t v2) { return (v2 >> 26) & 7u; }
TK_HD uint32_t tk_memo_id(uint32_t w4, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t i) {      /* i = 0..4 */
    const uint32_t m = (1u << TK_ID_BITS) - 1u;
    return i == 0u ? (v0 & m) : i == 1u ? (((v0 >> 21) | (v1 << 11)) & m) : i == 2u ? ((v1 >> 10) & m) : i == 3u ? (v2 & m) : (w4 & m);
}
TK_HD void tk_memo_pack(const uint32_t* r, uint32_t n, uint32_t len, uint32_t* w4, uint32_t* v) {  /* r[0..5): ranks < 2^21, unused ones 0 */
    const uint64_t lo = (uint64_t)r[0] | ((uint64_t)r[1] << 21) | ((uint64_t)r[2] << 42);
    v[0] = (uint32_t)lo; v[1] = (uint32_t)(lo >> 32);
    v[2] = r[3] | (len << 21) | (n << 26);
    *w4 = TK_MEMO_TAG | r[4];
}

/* cut rule (tk_tables.cpp make_cut_tables): exact bit maps over byte bigrams / trigrams */
#define TK_CUT_K2_WORDS (1u << 11)       /* 2^16 bits */
#define TK_CUT_G3_WORDS (1u << 19)       /* 2^24 bits = 2 MB */

TK_HD uint32_t tk_hash_alt(uint32_t h) { return (h << 16) | (h >> 16); }  /* second location: the other half of the bits */

TK_HD uint64_t tk_pair_pack(uint32_t a, uint32_t b, uint32_t rank) {
    return ((uint64_t)a << (2 * TK_ID_BITS)) | ((uint64_t)b << TK_ID_BITS) | (uint64_t)rank;
}
#define TK_PAIR_SPILL 0x8000000000000000ull  /* in the FIRST entry of a PAIR bucket: some pair whose first choice is this bucket lives
                                                 in its second one; clear => a probe that does not match here is a definite miss */
TK_HD uint64_t tk_pair_key(uint64_t e) { return (e >> TK_ID_BITS) & ((1ull << (2 * TK_ID_BITS)) - 1ull); }  /* an EMPTY entry reads
                                                 as the pair (2^21 - 1, 2^21 - 1): ids are < 2^21 - 1, it matches nothing */
TK_HD uint32_t tk_pair_rank(uint64_t e) { return (uint32_t)(e & ((1u << TK_ID_BITS) - 1u)); }

#endif
