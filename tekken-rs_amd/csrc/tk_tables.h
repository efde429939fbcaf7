// tk_tables.h -- host-side builder of the lookup tables the kernels read.
//
// Input is the validated rank table (rank i <-> token bytes i), i.e. the contents of the
// FxHashMap built by reload_mergeable_ranks (reference src/tekkenizer.rs:776-816) and handed
// to CoreBPE::new (:122-126).  Output is the flat, pointer-free image uploaded to HBM once
// per context (tk_capi.cpp) -- and the same image is what the CPU wave emulator of the
// test-suite runs the kernel source against.
#ifndef TK_TABLES_H
#define TK_TABLES_H
#include <stdint.h>
#include <string>
#include <vector>

#include "tk_hash.h"

// Raw-pointer view consumed by the kernels (device pointers in the product).
struct TkTablesView {
    const uint16_t* uc_stage1;       // Unicode class trie, stage 1 (cp >> 7 -> block)
    const uint32_t* uc_stage2;       // stage 2: 16 x 2-bit classes per word
    const uint16_t* uc2_stage1;      // class trie of the opt-in JSON pattern (row f-3): 4-bit classes O U W X M N S
    const uint32_t* uc2_stage2;      // stage 2: 8 x 4-bit classes per word
    const tk_key8_entry* key8_tab;   // whole pieces of 2..8 bytes (exact key), cuckoo: slots h & key8_mask, alt(h) & key8_mask
    const tk_key_entry* key_tab;     // whole pieces of 9..16 bytes (exact key), cuckoo: slots h & key_mask, alt(h) & key_mask
    const tk_long_entry* long_tab;   // whole pieces of >= 17 bytes
    const uint64_t* pair_tab;        // (idA,idB) -> rank, packed 21/21/21; cuckoo buckets of 2 entries, pair_mask = buckets - 1
    const uint32_t* pair2;           // [65536] (b0 | b1<<8) -> rank or TK_RANK_MAX
    const uint32_t* pair_filter;     // [TK_PAIRF_WORDS] bit tk_pair_fbit(hash) set for every pair of pair_tab (tk_hash.h)
    const tk_long_entry* key64_tab;  // whole pieces of 17..64 bytes, hashed by dwords (tk_hash.h KEY64): the flat kernel's look-up; behind the
                                     // key64_mask + 1 entries: TK_K64PRE_WORDS bit words of the pre-filter
    uint32_t key64_mask;
    const uint32_t* uc_bmp;          // [4096] the class trie flattened for the BMP: 16 x 2-bit classes per word, word cp >> 4 (ONE load per char)
    const uint32_t* cut_k2;          // [TK_CUT_K2_WORDS] bit (b0 | b1 << 8): the two bytes are a vocabulary KEY (cut rule, tk_hash.h)
    const uint32_t* cut_g3;          // [TK_CUT_G3_WORDS] bit (b0 | b1 << 8 | b2 << 16): the trigram occurs inside some token
    const uint8_t* blob;             // token bytes, for verifying LONG hits
    uint32_t key8_mask, key_mask, long_mask, pair_mask;
    uint32_t key_hash_mode;          // tk_key_hash mode the KEY table was built with
    uint32_t n_ranks, num_special, bos_id, eos_id;
    uint32_t p1inv, p2inv;           // inverses of the polynomial bases mod 2^32
};

struct TkHostTables {
    std::vector<uint8_t> blob;
    std::vector<uint32_t> offs;
    std::vector<uint16_t> uc_stage1;
    std::vector<uint32_t> uc_stage2;
    std::vector<uint16_t> uc2_stage1;   // constant tables (unicode_tables2.h), not part of the table cache
    std::vector<uint32_t> uc2_stage2;
    std::vector<tk_key8_entry> key8_tab;
    std::vector<tk_key_entry> key_tab;
    std::vector<tk_long_entry> long_tab;
    std::vector<uint64_t> pair_tab;
    std::vector<uint32_t> pair2;
    std::vector<uint32_t> pair_filter;  // derived from pair_tab (make_pair_filter), not part of the table cache
    std::vector<uint32_t> cut_k2, cut_g3;   // the cut rule's bit maps, derived from blob / offs (make_cut_tables), not part of the table cache
    std::vector<uint32_t> uc_bmp;           // derived from the trie (make_cut_tables)
    std::vector<tk_long_entry> key64_tab;   // derived from blob / offs (make_cut_tables)
    uint32_t key64_mask = 0;
    uint32_t key8_mask = 0, key_mask = 0, long_mask = 0, pair_mask = 0, key_hash_mode = 0;
    uint32_t n_ranks = 0, num_special = 0, bos_id = 0, eos_id = 0;
    uint32_t p1inv = 0, p2inv = 0;
    uint64_t n_pairs = 0, n_key = 0, n_long = 0, n_key_second = 0, n_key_spill_slots = 0;

    TkTablesView host_view() const;
    void make_pair_filter();
    void make_cut_tables();
};

// Returns 0 on success; on failure returns a negative TK_ERR_* code and fills `err`.
int tk_build_tables(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special,
                    uint32_t bos_id, uint32_t eos_id, TkHostTables& out, std::string& err);

uint32_t tk_inverse_u32(uint32_t odd);

// ---- table cache (loader / table-build acceleration, reference src/tekkenizer.rs:222-248,776-816) ----
// The derived tables depend only on the validated rank table; tk_tables_key hashes exactly the inputs of
// tk_build_tables (FNV-1a 64 over a format tag, the scalar arguments, the offsets and the token bytes).
// tk_build_tables_cached: with TK_TABLE_CACHE_DIR set, loads `<dir>/tk_tables_<key>.bin` if it is present and intact,
// otherwise builds and writes it (atomic rename).  A cache file never changes a result: every field of TkHostTables
// is stored, the loader checks magic, version, key and sizes and falls back to building on any mismatch.
uint64_t tk_tables_key(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, uint32_t bos_id,
                       uint32_t eos_id);
bool tk_tables_save(const TkHostTables& t, uint64_t key, const std::string& path);
bool tk_tables_load(TkHostTables& t, uint64_t key, const std::string& path);
int tk_build_tables_cached(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, uint32_t bos_id,
                           uint32_t eos_id, TkHostTables& out, std::string& err, bool* from_cache);

// 64-bit checksum of a byte range, chained through `seed` (four independent multiply-mix lanes: memory speed).  The side
// files of the loader cache end with the checksum of their payload: a flipped bit anywhere makes the file "not check out",
// so it is ignored and rewritten -- a cache can change load time, never a result.
inline uint64_t tk_sum64(uint64_t seed, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    uint64_t h[4] = {seed ^ 0xCBF29CE484222325ull, seed + 0x9E3779B97F4A7C15ull, seed ^ 0xD6E8FEB86659FD93ull, seed + 0xA0761D6478BD642Full};
    size_t i = 0;
    for (; i + 32 <= n; i += 32)
        for (int l = 0; l < 4; ++l) {
            uint64_t w;
            __builtin_memcpy(&w, b + i + 8 * l, 8);
            h[l] = (h[l] ^ w) * 0x100000001B3ull;
            h[l] ^= h[l] >> 29;
        }
    for (; i < n; ++i) { h[0] = (h[0] ^ b[i]) * 0x100000001B3ull; h[0] ^= h[0] >> 29; }
    uint64_t x = h[0] ^ (h[1] * 0x9E3779B97F4A7C15ull) ^ (h[2] >> 7) ^ (h[3] << 9) ^ (uint64_t)n;
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 33;
    return x;
}

#endif
