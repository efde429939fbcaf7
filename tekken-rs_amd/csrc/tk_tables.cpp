// tk_tables.cpp -- builds SHORT / LONG / PAIR / PAIR2 and the Unicode class trie image.
// See tk_hash.h for the layouts and tk_tables.h for provenance.
#include "tk_tables.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <utility>

#include "../../include/tekken_hip.h"
#include "unicode_tables.h"
#include "unicode_tables2.h"

uint32_t tk_inverse_u32(uint32_t a) {
    // Newton iteration for the inverse of an odd number modulo 2^32
    uint32_t x = a;  // correct to 3 bits
    for (int i = 0; i < 5; ++i) x *= 2u - a * x;
    return x;
}

static uint32_t pow2_at_least(uint64_t n) {
    uint32_t c = 1024;
    while (c < n) c <<= 1;
    return c;
}

void TkHostTables::make_pair_filter() {
    pair_filter.assign(TK_PAIRF_WORDS, 0u);
    for (uint64_t e : pair_tab) {
        if (e == TK_PAIR_EMPTY) continue;
        const uint64_t key = tk_pair_key(e);
        const uint32_t b = tk_pair_fbit(tk_pair_hash((uint32_t)(key >> TK_ID_BITS), (uint32_t)(key & ((1u << TK_ID_BITS) - 1u))));
        pair_filter[b >> 5] |= 1u << (b & 31u);
    }
}

// The cut rule (tk_flat_impl.h, CUT instantiation; model: tools/cut_model.py): a part of the merge loop that spans the
// boundary between bytes i-1 and i is a vocabulary key -- of two bytes (then it is the bigram itself: cut_k2) or of more
// (then it contains b[i-2..i] or b[i-1..i+1]: cut_g3 holds every trigram that occurs inside a token).
void TkHostTables::make_cut_tables() {
    // (also: the two-stage class trie flattened for the BMP -- the flat kernel classifies a multi-byte char with one load)
    uc_bmp.assign(4096, 0u);
    for (uint32_t cp = 0; cp < 0x10000u; ++cp) {
        const uint32_t blk = uc_stage1[cp >> 7];
        const uint32_t w = uc_stage2[blk * 8u + ((cp & 127u) >> 4)];
        uc_bmp[cp >> 4] |= ((w >> (2u * (cp & 15u))) & 3u) << (2u * (cp & 15u));
    }
    // KEY64: the tokens of 17..64 bytes by the dword hash of the flat kernel (tk_hash.h)
    {
        uint64_t n64 = 0;
        for (uint32_t r = 0; r < n_ranks; ++r) {
            const uint32_t len = offs[r + 1] - offs[r];
            if (len >= 17u && len <= 64u) ++n64;
        }
        const uint32_t cap = pow2_at_least(2 * n64 + 1);
        key64_mask = cap - 1;
        key64_tab.assign(cap + TK_K64PRE_WORDS / 4, tk_long_entry{0, 0, 0, 0});   // entries, then the pre-filter's bit words (tk_hash.h)
        for (uint32_t r = 0; r < n_ranks; ++r) {
            const uint8_t* p = blob.data() + offs[r];
            const uint32_t len = offs[r + 1] - offs[r];
            if (len < 17u || len > 64u) continue;
            uint32_t ha = 0;
            for (uint32_t j = 0; j < len; j += 4) {
                uint32_t w = 0;
                for (uint32_t k = 0; k < 4 && j + k < len; ++k) w |= (uint32_t)p[j + k] << (8 * k);
                if (j == 0) ha = w;                                       // tk_k64_start, dword by dword
                else if (j == 4) ha ^= tk_rotl32(w, 7);
                else if (j == 8) ha ^= tk_rotl32(w, 14);
                else if (j == 12) ha ^= tk_rotl32(w, 21);
                else tk_k64_step(ha, w);
            }
            uint32_t sl = tk_k64_slot(ha, len) & key64_mask;
            while (key64_tab[sl].len) sl = (sl + 1) & key64_mask;
            key64_tab[sl] = tk_long_entry{tk_k64_tag(ha), r, len, offs[r]};
            uint32_t k[4];
            memcpy(k, p, 16);                                             // (little-endian host, like the byte order of every key here)
            const uint32_t pb = tk_k64_prebit(tk_key_hash(key_hash_mode, k[0], k[1], k[2], k[3], len));
            reinterpret_cast<uint32_t*>(key64_tab.data() + cap)[pb >> 5] |= 1u << (pb & 31u);
        }
    }
    cut_k2.assign(TK_CUT_K2_WORDS, 0u);
    cut_g3.assign(TK_CUT_G3_WORDS, 0u);
    for (uint32_t r = 0; r < n_ranks; ++r) {
        const uint8_t* p = blob.data() + offs[r];
        const uint32_t len = offs[r + 1] - offs[r];
        if (len == 2u) {
            const uint32_t b = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
            cut_k2[b >> 5] |= 1u << (b & 31u);
        }
        for (uint32_t i = 0; i + 3u <= len; ++i) {
            const uint32_t b = (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16);
            cut_g3[b >> 5] |= 1u << (b & 31u);
        }
    }
}

TkTablesView TkHostTables::host_view() const {
    TkTablesView v;
    v.uc_stage1 = uc_stage1.data();
    v.uc_stage2 = uc_stage2.data();
    v.uc2_stage1 = uc2_stage1.data();
    v.uc2_stage2 = uc2_stage2.data();
    v.key8_tab = key8_tab.data();
    v.key_tab = key_tab.data();
    v.long_tab = long_tab.data();
    v.pair_tab = pair_tab.data();
    v.pair2 = pair2.data();
    v.pair_filter = pair_filter.data();
    v.uc_bmp = uc_bmp.data();
    v.key64_tab = key64_tab.data();
    v.key64_mask = key64_mask;
    v.cut_k2 = cut_k2.data();
    v.cut_g3 = cut_g3.data();
    v.blob = blob.data();
    v.key8_mask = key8_mask;
    v.key_mask = key_mask;
    v.long_mask = long_mask;
    v.pair_mask = pair_mask;
    v.key_hash_mode = key_hash_mode;
    v.n_ranks = n_ranks;
    v.num_special = num_special;
    v.bos_id = bos_id;
    v.eos_id = eos_id;
    v.p1inv = p1inv;
    v.p2inv = p2inv;
    return v;
}

int tk_build_tables(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special,
                    uint32_t bos_id, uint32_t eos_id, TkHostTables& out, std::string& err) {
    if (!blob || !offs) { err = "null rank table"; return TK_ERR_INVALID_CONFIG; }
    if (n_ranks < 256) {
        err = "rank table must contain the 256 single-byte tokens (ranks 0..255)";
        return TK_ERR_INVALID_CONFIG;
    }
    if (n_ranks >= TK_MAX_RANKS) {
        err = "rank table too large: ids must fit in 21 bits";
        return TK_ERR_INVALID_CONFIG;
    }
    if ((uint64_t)n_ranks + num_special >= 0xFFFFFFF0ull) { err = "id space overflow"; return TK_ERR_INVALID_CONFIG; }
    for (uint32_t r = 0; r < n_ranks; ++r) {
        if (offs[r + 1] < offs[r]) {
            err = "token offsets must be non-decreasing (rank " + std::to_string(r) + ")";
            return TK_ERR_INVALID_CONFIG;
        }
    }
    // ranks < 256 must be the byte itself (reference src/tekkenizer.rs:793-798)
    for (uint32_t r = 0; r < 256; ++r) {
        if (offs[r + 1] - offs[r] != 1 || blob[offs[r]] != (uint8_t)r) {
            err = "Expected byte token at rank " + std::to_string(r) + " to be [" + std::to_string(r) + "]";
            return TK_ERR_INVALID_CONFIG;
        }
    }

    out = TkHostTables();
    out.n_ranks = n_ranks;
    out.num_special = num_special;
    out.bos_id = bos_id;
    out.eos_id = eos_id;
    out.p1inv = tk_inverse_u32(TK_POLY_P1);
    out.p2inv = tk_inverse_u32(TK_POLY_P2);
    out.offs.assign(offs, offs + n_ranks + 1);
    out.blob.assign(blob, blob + offs[n_ranks]);
    out.blob.resize(out.blob.size() + 16, 0);  // slack so verify loops never end on the last byte
    out.uc_stage1.assign(TK_UC_STAGE1, TK_UC_STAGE1 + TK_UC_STAGE1_LEN);
    out.uc_stage2.assign(TK_UC_STAGE2, TK_UC_STAGE2 + TK_UC_STAGE2_LEN);
    out.uc2_stage1.assign(TK_UC2_STAGE1, TK_UC2_STAGE1 + TK_UC2_STAGE1_LEN);
    out.uc2_stage2.assign(TK_UC2_STAGE2, TK_UC2_STAGE2 + TK_UC2_STAGE2_LEN);

    // bytes -> rank; duplicate keys collapse in the reference's map and then fail its
    // contiguity check (src/tekkenizer.rs:801-813) => reject here as well
    std::unordered_map<std::string, uint32_t> by_bytes;
    by_bytes.reserve((size_t)n_ranks * 2);
    for (uint32_t r = 0; r < n_ranks; ++r) {
        std::string k((const char*)blob + offs[r], offs[r + 1] - offs[r]);
        if (!by_bytes.emplace(std::move(k), r).second) {
            err = "Vocabulary ranks are not contiguous (duplicate token bytes at rank " + std::to_string(r) + ")";
            return TK_ERR_INVALID_CONFIG;
        }
    }

    uint64_t n_key = 0, n_long = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        uint32_t len = offs[r + 1] - offs[r];
        if (len >= 2 && len <= 16) ++n_key;
        else if (len >= 17) ++n_long;
    }
    out.n_key = n_key;
    out.n_long = n_long;
    uint32_t lcap = pow2_at_least(2 * n_long + 1);
    out.long_mask = lcap - 1;
    struct KeyEnt { uint32_t k[4], rank, len; };   // placement works on this; the device layouts are written at the end
    std::vector<KeyEnt> key_entries;
    key_entries.reserve(n_key);
    out.long_tab.assign(lcap, tk_long_entry{0, 0, 0, 0});
    out.pair2.assign(65536, TK_RANK_MAX);

    for (uint32_t r = 0; r < n_ranks; ++r) {
        const uint8_t* p = blob + offs[r];
        uint32_t len = offs[r + 1] - offs[r];
        if (len == 2) out.pair2[p[0] | ((uint32_t)p[1] << 8)] = r;
        if (len >= 2 && len <= 16) {
            uint32_t k[4] = {0, 0, 0, 0};
            for (uint32_t j = 0; j < len; ++j) k[j >> 2] |= (uint32_t)p[j] << (8 * (j & 3));
            key_entries.push_back(KeyEnt{{k[0], k[1], k[2], k[3]}, r, len});
        } else if (len >= 17) {
            uint32_t h1 = 0, h2 = 0;
            for (uint32_t k = 0; k < len; ++k) {
                h1 = h1 * TK_POLY_P1 + p[k];
                h2 = h2 * TK_POLY_P2 + p[k];
            }
            uint32_t s = tk_long_hash(h1, len) & out.long_mask;
            while (out.long_tab[s].len) s = (s + 1) & out.long_mask;
            out.long_tab[s] = tk_long_entry{h2, r, len, offs[r]};
        }
    }

    // KEY8 (2..8 bytes, 16-byte entries) and KEY16 (9..16 bytes, 32-byte entries): cuckoo placement, one entry per slot,
    // two candidate slots per key, load <= 1/3, first choices filled first.  The cheap hash (mode 0) is tried at the
    // initial capacity and at twice that; if the vocabulary cannot be placed (structured collisions of the fold), the
    // strong hash (mode 1) is used and the tables grown until they fit.
    {
        std::vector<KeyEnt> e8, e16;
        for (const KeyEnt& e : key_entries) (e.len <= 8 ? e8 : e16).push_back(e);
        // Placement is frequency-aware: `ents` come in rank order (merge order = how common a token is).  First every key
        // whose FIRST choice is still free takes it, lower ranks first; only then the others take their second choice or
        // kick -- and a kick moves the RARER of the two residents.  So the tokens that make up most of any text sit in their
        // first choice, and a wave of 64 probes seldom has a lane that needs the second fetch (the second-fetch code is
        // per wave, not per lane: before this ~0.9 of the batches of the flat kernel ran it).
        auto place = [&](const std::vector<KeyEnt>& ents, uint32_t mode, uint32_t cap, std::vector<KeyEnt>& tab) -> bool {
            const uint32_t mask = cap - 1;
            tab.assign(cap, KeyEnt{{0, 0, 0, 0}, 0, 0});
            std::vector<const KeyEnt*> later;
            for (const KeyEnt& e : ents) {
                const uint32_t s1 = tk_key_hash(mode, e.k[0], e.k[1], e.k[2], e.k[3], e.len) & mask;
                if (tab[s1].len == 0) tab[s1] = e; else later.push_back(&e);
            }
            uint32_t rnd = 0x9E3779B9u;
            for (const KeyEnt* e0 : later) {
                KeyEnt e = *e0;
                uint32_t avoid = 0xFFFFFFFFu;
                for (int kicks = 0;; ++kicks) {
                    const uint32_t h = tk_key_hash(mode, e.k[0], e.k[1], e.k[2], e.k[3], e.len);
                    const uint32_t s1 = h & mask, s2 = tk_hash_alt(h) & mask;
                    if (tab[s1].len == 0) { tab[s1] = e; break; }
                    if (tab[s2].len == 0) { tab[s2] = e; break; }
                    if (kicks >= 500) return false;
                    rnd = rnd * 1664525u + 1013904223u;
                    // the rarer resident goes (ties and the no-bounce rule: the other one); now and then a random one, so that
                    // a cycle of kicks cannot repeat forever
                    uint32_t victim = tab[s1].rank > tab[s2].rank ? s1 : s2;
                    if (((rnd >> 16) & 7u) == 0u) victim = (rnd >> 20) & 1u ? s2 : s1;
                    if (victim == avoid) victim = victim == s1 ? s2 : s1;   // do not bounce straight back
                    std::swap(e, tab[victim]);
                    avoid = victim;
                }
            }
            return true;
        };
        const uint32_t cap8_0 = pow2_at_least(3 * e8.size() + 1), cap16_0 = pow2_at_least(3 * e16.size() + 1);
        std::vector<KeyEnt> t8, t16;
        bool done = false;
        for (int attempt = 0; !done; ++attempt) {
            const uint32_t mode = attempt < 2 ? 0u : 1u;
            const int grow = attempt < 2 ? attempt : attempt - 2;
            if (((uint64_t)cap8_0 << grow) > (1ull << 27) || ((uint64_t)cap16_0 << grow) > (1ull << 26)) {   // byte offsets stay below 2^31
                err = "could not place the vocabulary in the KEY tables";
                return TK_ERR_INVALID_CONFIG;
            }
            out.key_hash_mode = mode;
            out.key8_mask = (cap8_0 << grow) - 1;
            out.key_mask = (cap16_0 << grow) - 1;
            done = place(e8, mode, cap8_0 << grow, t8) && place(e16, mode, cap16_0 << grow, t16);
        }
        // spill flags: slot s is flagged iff a key whose first choice is s was placed in its second choice
        auto flag = [&](std::vector<KeyEnt>& tab, uint32_t mask) {
            std::vector<uint32_t> spill;
            for (uint32_t sl = 0; sl <= mask; ++sl) {
                const KeyEnt& e = tab[sl];
                if (e.len == 0) continue;
                const uint32_t s1 = tk_key_hash(out.key_hash_mode, e.k[0], e.k[1], e.k[2], e.k[3], e.len) & mask;
                if (s1 != sl) { spill.push_back(s1); ++out.n_key_second; }
            }
            for (uint32_t s1 : spill) {
                if (!(tab[s1].len & TK_KEY_SPILL)) ++out.n_key_spill_slots;
                tab[s1].len |= TK_KEY_SPILL;
            }
        };
        flag(t8, out.key8_mask);
        flag(t16, out.key_mask);
        out.key8_tab.resize(t8.size());
        for (size_t i = 0; i < t8.size(); ++i) out.key8_tab[i] = tk_key8_entry{{t8[i].k[0], t8[i].k[1]}, t8[i].rank, t8[i].len};
        out.key_tab.resize(t16.size());
        for (size_t i = 0; i < t16.size(); ++i)
            out.key_tab[i] = tk_key_entry{{t16[i].k[0], t16[i].k[1]}, t16[i].rank, t16[i].len, {t16[i].k[2], t16[i].k[3]}, {0, 0}};
    }

    // PAIR: every split of every token whose halves are both tokens
    std::vector<uint64_t> pairs;
    pairs.reserve((size_t)n_ranks * 4);
    std::string a, b;
    for (uint32_t r = 256; r < n_ranks; ++r) {
        const char* p = (const char*)blob + offs[r];
        uint32_t len = offs[r + 1] - offs[r];
        for (uint32_t k = 1; k < len; ++k) {
            a.assign(p, k);
            auto ia = by_bytes.find(a);
            if (ia == by_bytes.end()) continue;
            b.assign(p + k, len - k);
            auto ib = by_bytes.find(b);
            if (ib == by_bytes.end()) continue;
            pairs.push_back(tk_pair_pack(ia->second, ib->second, r));
        }
    }
    out.n_pairs = pairs.size();
    // cuckoo placement: buckets of two entries, two candidate buckets per pair, load <= 1/2 (grown on failure)
    for (uint32_t nb = pow2_at_least(pairs.size() + 1);; nb <<= 1) {
        out.pair_mask = nb - 1;
        out.pair_tab.assign((size_t)nb * 2, TK_PAIR_EMPTY);
        bool ok = true;
        uint32_t rnd = 0x85EBCA6Bu;
        for (uint64_t e0 : pairs) {
            uint64_t e = e0;
            uint64_t avoid = ~0ull;
            int kicks = 0;
            for (;; ++kicks) {
                const uint64_t key = tk_pair_key(e);
                const uint32_t ida = (uint32_t)(key >> TK_ID_BITS), idb = (uint32_t)(key & ((1u << TK_ID_BITS) - 1u));
                const uint32_t h = tk_pair_hash(ida, idb);
                const uint64_t b1 = 2ull * (h & out.pair_mask), b2 = 2ull * (tk_hash_alt(h) & out.pair_mask);
                const uint64_t cand[4] = {b1, b1 + 1, b2, b2 + 1};
                bool placed = false;
                for (uint64_t sl : cand)
                    if (out.pair_tab[sl] == TK_PAIR_EMPTY) { out.pair_tab[sl] = e; placed = true; break; }
                if (placed) break;
                if (kicks >= 500) { ok = false; break; }
                rnd = rnd * 1664525u + 1013904223u;
                uint64_t victim = cand[(rnd >> 16) & 3u];
                if (victim == avoid) victim = cand[((rnd >> 16) + 1u) & 3u];
                std::swap(e, out.pair_tab[victim]);
                avoid = victim;
            }
            if (!ok) break;
        }
        if (ok) break;
        if (nb >= (1u << 30)) { err = "could not place the vocabulary in the PAIR table"; return TK_ERR_INVALID_CONFIG; }
    }
    // spill flags: bucket b is flagged (bit 63 of its first entry) iff a pair whose first choice is b was placed in its
    // second choice -- only then can a probe that does not match in b find its pair elsewhere
    {
        std::vector<uint64_t> spill;
        for (uint64_t sl = 0; sl < out.pair_tab.size(); ++sl) {
            const uint64_t e = out.pair_tab[sl];
            if (e == TK_PAIR_EMPTY) continue;
            const uint64_t key = tk_pair_key(e);
            const uint32_t ida = (uint32_t)(key >> TK_ID_BITS), idb = (uint32_t)(key & ((1u << TK_ID_BITS) - 1u));
            const uint64_t b1 = tk_pair_hash(ida, idb) & out.pair_mask;
            if (b1 != sl / 2) spill.push_back(b1);
        }
        for (uint64_t b1 : spill) out.pair_tab[2 * b1] |= TK_PAIR_SPILL;   // (a bucket that spilled is full: its first entry is real)
    }
    out.make_pair_filter();
    out.make_cut_tables();
    return TK_OK;
}

// ------------------------------------------------------------------------------------------
// table cache (row f-2)
// ------------------------------------------------------------------------------------------
#define TK_CACHE_MAGIC 0x42544B54u /* "TKTB" */
#define TK_CACHE_VERSION 6u        /* bump whenever a table layout or a hash function changes */

static uint64_t fnv1a64(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}

uint64_t tk_tables_key(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, uint32_t bos_id,
                       uint32_t eos_id) {
    const uint32_t head[8] = {TK_CACHE_MAGIC, TK_CACHE_VERSION, n_ranks, num_special, bos_id, eos_id, TK_UC_STAGE1_LEN, TK_UC_STAGE2_LEN};
    uint64_t h = fnv1a64(0xCBF29CE484222325ull, head, sizeof(head));
    h = fnv1a64(h, offs, ((size_t)n_ranks + 1) * sizeof(uint32_t));
    return fnv1a64(h, blob, offs[n_ranks]);
}

namespace {
// (every vector goes through the running payload checksum `sum`, which is the file's tail)
template <class T>
bool put_vec(FILE* f, const std::vector<T>& v, uint64_t& sum) {
    const uint64_t n = v.size();
    sum = tk_sum64(tk_sum64(sum, &n, 8), v.data(), n * sizeof(T));
    return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T>
bool get_vec(FILE* f, std::vector<T>& v, uint64_t max_elems, uint64_t& sum) {
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1 || n > max_elems) return false;
    v.resize(n);
    if (!(n == 0 || fread(v.data(), sizeof(T), n, f) == n)) return false;
    sum = tk_sum64(tk_sum64(sum, &n, 8), v.data(), n * sizeof(T));
    return true;
}
struct CacheHead {
    uint32_t magic, version;
    uint64_t key;
    uint32_t key8_mask, key_mask, long_mask, pair_mask, key_hash_mode, n_ranks, num_special, bos_id, eos_id, p1inv, p2inv, pad;
    uint64_t n_pairs, n_key, n_long, n_key_second, n_key_spill_slots;
};
}  // namespace

bool tk_tables_save(const TkHostTables& t, uint64_t key, const std::string& path) {
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)key & 0xFFFF);
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    CacheHead h;
    memset(&h, 0, sizeof(h));
    h.magic = TK_CACHE_MAGIC; h.version = TK_CACHE_VERSION; h.key = key;
    h.key8_mask = t.key8_mask; h.key_mask = t.key_mask; h.long_mask = t.long_mask; h.pair_mask = t.pair_mask;
    h.key_hash_mode = t.key_hash_mode; h.n_ranks = t.n_ranks; h.num_special = t.num_special; h.bos_id = t.bos_id; h.eos_id = t.eos_id;
    h.p1inv = t.p1inv; h.p2inv = t.p2inv;
    h.n_pairs = t.n_pairs; h.n_key = t.n_key; h.n_long = t.n_long; h.n_key_second = t.n_key_second; h.n_key_spill_slots = t.n_key_spill_slots;
    uint64_t sum = tk_sum64(key, &h, sizeof(h));
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && put_vec(f, t.blob, sum) && put_vec(f, t.offs, sum) && put_vec(f, t.uc_stage1, sum) &&
              put_vec(f, t.uc_stage2, sum) && put_vec(f, t.key8_tab, sum) && put_vec(f, t.key_tab, sum) && put_vec(f, t.long_tab, sum) &&
              put_vec(f, t.pair_tab, sum) && put_vec(f, t.pair2, sum);
    const uint32_t tail = TK_CACHE_MAGIC;   // a truncated file has no tail; a damaged one a wrong checksum
    ok = ok && fwrite(&sum, 8, 1, f) == 1 && fwrite(&tail, 4, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) remove(tmp.c_str());
    return ok;
}

bool tk_tables_load(TkHostTables& t, uint64_t key, const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    CacheHead h;
    TkHostTables x;
    const uint64_t lim = 1ull << 31;
    uint32_t tail = 0;
    uint64_t sum = 0, want = 0;
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && h.magic == TK_CACHE_MAGIC && h.version == TK_CACHE_VERSION && h.key == key;
    if (ok) sum = tk_sum64(key, &h, sizeof(h));
    ok = ok && get_vec(f, x.blob, lim, sum) && get_vec(f, x.offs, lim, sum) && get_vec(f, x.uc_stage1, lim, sum) && get_vec(f, x.uc_stage2, lim, sum) &&
         get_vec(f, x.key8_tab, lim, sum) && get_vec(f, x.key_tab, lim, sum) && get_vec(f, x.long_tab, lim, sum) && get_vec(f, x.pair_tab, lim, sum) &&
         get_vec(f, x.pair2, lim, sum) && fread(&want, 8, 1, f) == 1 && want == sum && fread(&tail, 4, 1, f) == 1 && tail == TK_CACHE_MAGIC;
    fclose(f);
    if (!ok) return false;
    // sizes must agree with the masks the kernels index with
    if (x.key8_tab.size() != (size_t)h.key8_mask + 1 || x.key_tab.size() != (size_t)h.key_mask + 1 ||
        x.long_tab.size() != (size_t)h.long_mask + 1 || x.pair_tab.size() != 2 * ((size_t)h.pair_mask + 1) || x.pair2.size() != 65536 ||
        x.offs.size() != (size_t)h.n_ranks + 1 || x.uc_stage1.size() != TK_UC_STAGE1_LEN || x.uc_stage2.size() != TK_UC_STAGE2_LEN)
        return false;
    x.key8_mask = h.key8_mask; x.key_mask = h.key_mask; x.long_mask = h.long_mask; x.pair_mask = h.pair_mask;
    x.key_hash_mode = h.key_hash_mode; x.n_ranks = h.n_ranks; x.num_special = h.num_special; x.bos_id = h.bos_id; x.eos_id = h.eos_id;
    x.p1inv = h.p1inv; x.p2inv = h.p2inv;
    x.n_pairs = h.n_pairs; x.n_key = h.n_key; x.n_long = h.n_long; x.n_key_second = h.n_key_second; x.n_key_spill_slots = h.n_key_spill_slots;
    x.uc2_stage1.assign(TK_UC2_STAGE1, TK_UC2_STAGE1 + TK_UC2_STAGE1_LEN);   // constant tables, not cached
    x.uc2_stage2.assign(TK_UC2_STAGE2, TK_UC2_STAGE2 + TK_UC2_STAGE2_LEN);
    x.make_pair_filter();
    x.make_cut_tables();
    t = std::move(x);
    return true;
}

int tk_build_tables_cached(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special, uint32_t bos_id,
                           uint32_t eos_id, TkHostTables& out, std::string& err, bool* from_cache) {
    if (from_cache) *from_cache = false;
    const char* dir = getenv("TK_TABLE_CACHE_DIR");
    if (!dir || !*dir || !blob || !offs || n_ranks < 256) return tk_build_tables(blob, offs, n_ranks, num_special, bos_id, eos_id, out, err);
    for (uint32_t r = 0; r < n_ranks; ++r)   // (the key walks offs[n_ranks] bytes: the offsets must be sane first)
        if (offs[r + 1] < offs[r]) return tk_build_tables(blob, offs, n_ranks, num_special, bos_id, eos_id, out, err);
    const uint64_t key = tk_tables_key(blob, offs, n_ranks, num_special, bos_id, eos_id);
    char name[64];
    snprintf(name, sizeof(name), "/tk_tables_%016llx.bin", (unsigned long long)key);
    const std::string path = std::string(dir) + name;
    if (tk_tables_load(out, key, path)) {
        if (from_cache) *from_cache = true;
        return TK_OK;
    }
    const int rc = tk_build_tables(blob, offs, n_ranks, num_special, bos_id, eos_id, out, err);
    if (rc == TK_OK) (void)tk_tables_save(out, key, path);   // best effort
    return rc;
}
