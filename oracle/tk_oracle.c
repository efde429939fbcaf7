#define _POSIX_C_SOURCE 200809L
/*
 * tk_oracle.c -- CPU restatement of the tekken-rs text-encode hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see tk_oracle.h).  "parity unpinned" at token-id level: the
 * algorithm below restates tiktoken-rs ^0.7.0 (reference Cargo.toml:40, source absent)
 * from its published behaviour; see SURVEY.md App. A.  Each function cites the reference
 * line it follows.
 *
 *   reference call site                       here
 *   src/tekkenizer.rs:123 (pattern literal)   match_at()  -- the 7 alternatives, leftmost-first
 *   src/tekkenizer.rs:384-386 CoreBPE::encode tk_oracle_encode(): find_iter + lookup + merge
 *   (tiktoken-rs byte_pair_encode/_merge)     bpe_piece()
 *   src/tekkenizer.rs:390-392 id shift        tk_oracle_encode()
 *   src/tekkenizer.rs:394-402 BOS/EOS         tk_oracle_encode()
 */
#include "tk_oracle.h"
#include "tk_unicode_tables.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tk_unicode_tables2.h"

#define CLS_O 0
#define CLS_L 1
#define CLS_N 2
#define CLS_S 3
#define RANK_MAX 0xFFFFFFFFu

int tk_oracle_class(uint32_t cp) {
    if (cp >= 0x110000u) return CLS_O;
    uint32_t blk = TK_UC_STAGE1[cp >> 7];
    uint32_t w = TK_UC_STAGE2[blk * 8 + ((cp & 127) >> 4)];
    return (int)((w >> (2 * (cp & 15))) & 3u);
}

/* Decode one UTF-8 scalar at t[p] (p < n).  Input is a Rust &str in the reference
 * (src/tekkenizer.rs:380) so it is valid UTF-8; malformed bytes are treated as one-byte
 * class-O characters so that the oracle never reads out of bounds. */
static uint32_t decode(const uint8_t* t, size_t n, size_t p, size_t* len) {
    uint8_t b0 = t[p];
    if (b0 < 0x80) { *len = 1; return b0; }
    if ((b0 & 0xE0) == 0xC0 && p + 1 < n && (t[p + 1] & 0xC0) == 0x80) {
        *len = 2; return ((uint32_t)(b0 & 0x1F) << 6) | (t[p + 1] & 0x3F);
    }
    if ((b0 & 0xF0) == 0xE0 && p + 2 < n && (t[p + 1] & 0xC0) == 0x80 && (t[p + 2] & 0xC0) == 0x80) {
        *len = 3;
        return ((uint32_t)(b0 & 0x0F) << 12) | ((uint32_t)(t[p + 1] & 0x3F) << 6) | (t[p + 2] & 0x3F);
    }
    if ((b0 & 0xF8) == 0xF0 && p + 3 < n && (t[p + 1] & 0xC0) == 0x80 && (t[p + 2] & 0xC0) == 0x80 &&
        (t[p + 3] & 0xC0) == 0x80) {
        *len = 4;
        return ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(t[p + 1] & 0x3F) << 12) |
               ((uint32_t)(t[p + 2] & 0x3F) << 6) | (t[p + 3] & 0x3F);
    }
    *len = 1;
    return 0xFFFFFFFFu; /* malformed -> class O */
}

static int cls_at(const uint8_t* t, size_t n, size_t p, size_t* len, uint32_t* cp) {
    uint32_t c = decode(t, n, p, len);
    *cp = c;
    return tk_oracle_class(c);
}

static int is_crlf(uint32_t cp) { return cp == '\r' || cp == '\n'; }

/* Unicode simple case folding restricted to what (?i:'s|'t|'re|'ve|'m|'ll|'d) can see:
 * ASCII upper/lower, plus U+017F LATIN SMALL LETTER LONG S folding to 's'
 * (verified against Python `regex`: no other code point folds to s,t,r,e,v,m,l,d). */
static uint32_t fold(uint32_t cp) {
    if (cp >= 'A' && cp <= 'Z') return cp + 32;
    if (cp == 0x17F) return 's';
    return cp;
}

/* One leftmost-first match attempt at byte position pos (a char boundary, pos < n).
 * Returns the end of the match (> pos).  Alternatives are tried in pattern order and the
 * first that matches wins (Perl semantics of fancy-regex), src/tekkenizer.rs:123. */
static size_t match_at(const uint8_t* t, size_t n, size_t pos) {
    size_t l0, l1, l2;
    uint32_t c0, c1, c2;
    int k0 = cls_at(t, n, pos, &l0, &c0);

    /* alt 1: (?i:'s|'t|'re|'ve|'m|'ll|'d) */
    if (c0 == '\'' && pos + 1 < n) {
        (void)cls_at(t, n, pos + 1, &l1, &c1);
        uint32_t f1 = fold(c1);
        if (f1 == 's' || f1 == 't') return pos + 1 + l1;
        if ((f1 == 'r' || f1 == 'v') && pos + 1 + l1 < n) {
            (void)cls_at(t, n, pos + 1 + l1, &l2, &c2);
            if (fold(c2) == 'e') return pos + 1 + l1 + l2;
        }
        if (f1 == 'm') return pos + 1 + l1;
        if (f1 == 'l' && pos + 1 + l1 < n) {
            (void)cls_at(t, n, pos + 1 + l1, &l2, &c2);
            if (fold(c2) == 'l') return pos + 1 + l1 + l2;
        }
        if (f1 == 'd') return pos + 1 + l1;
    }

    /* alt 2: [^\r\n\p{L}\p{N}]?\p{L}+   (greedy '?', backtrack to the empty prefix) */
    {
        int with_prefix = !(is_crlf(c0) || k0 == CLS_L || k0 == CLS_N);
        for (int attempt = 0; attempt < 2; ++attempt) {
            size_t p;
            if (attempt == 0) { if (!with_prefix) continue; p = pos + l0; }
            else p = pos;
            size_t q = p;
            while (q < n) {
                size_t l; uint32_t c;
                if (cls_at(t, n, q, &l, &c) != CLS_L) break;
                q += l;
            }
            if (q > p) return q;
        }
    }

    /* alt 3: \p{N}{1,3} */
    if (k0 == CLS_N) {
        size_t q = pos + l0;
        for (int cnt = 1; cnt < 3 && q < n; ++cnt) {
            size_t l; uint32_t c;
            if (cls_at(t, n, q, &l, &c) != CLS_N) break;
            q += l;
        }
        return q;
    }

    /* alt 4:  ?[^\s\p{L}\p{N}]+[\r\n]*   (greedy ' ?', backtrack to no space) */
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t p;
        if (attempt == 0) { if (c0 != ' ') continue; p = pos + 1; }
        else p = pos;
        size_t q = p;
        while (q < n) {
            size_t l; uint32_t c;
            if (cls_at(t, n, q, &l, &c) != CLS_O) break;
            q += l;
        }
        if (q > p) {
            while (q < n && (t[q] == '\r' || t[q] == '\n')) ++q;
            return q;
        }
    }

    /* from here on the char at pos is white space (L, N, O were all consumed above) */
    {
        /* maximal \s run [pos, e) ; remember the position after the last CR/LF and the start
         * of the last char of the run */
        size_t e = pos, after_last_nl = 0, last_char = pos;
        int has_nl = 0;
        while (e < n) {
            size_t l; uint32_t c;
            if (cls_at(t, n, e, &l, &c) != CLS_S) break;
            last_char = e;
            e += l;
            if (is_crlf(c)) { has_nl = 1; after_last_nl = e; }
        }
        /* alt 5: \s*[\r\n]+  -- greedy \s*, backtracked until [\r\n]+ can close the match:
         * ends right after the LAST CR/LF of the run */
        if (has_nl) return after_last_nl;
        /* alt 6: \s+(?!\S) -- whole run at end of text, else the run minus its last char */
        if (e == n) return e;
        if (last_char > pos) return last_char;
        /* alt 7: \s+ */
        return e;
    }
}

/* ---------------------------------------------------------------------------------------
 * SURVEY section 8 row f-3 (groundwork, NOT on the device yet): the `pattern` that Mistral's tekken.json carries and the
 * reference ignores (src/tekkenizer.rs:74; literal in tests/test_small_vocab.rs:62):
 *   [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+
 *  |[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*
 *  |\p{N}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+
 * Classes (tk_unicode_tables2.h): U = Lu|Lt, W = Ll, X = Lm|Lo, M = marks, N, S, O.  A = U|X|M, B = W|X|M.
 * Leftmost-first alternation, greedy quantifiers with backtracking -- pinned against Python `regex`
 * (tests/golden/split_vectors_tekken.json).
 * ------------------------------------------------------------------------------------- */
#define C2_O 0
#define C2_U 1
#define C2_W 2
#define C2_X 3
#define C2_M 4
#define C2_N 5
#define C2_S 6

int tk_oracle_class2(uint32_t cp) {
    if (cp >= 0x110000u) return C2_O;
    uint32_t blk = TK_UC2_STAGE1[cp >> 7];
    uint32_t w = TK_UC2_STAGE2[blk * 16 + ((cp & 127) >> 3)];
    return (int)((w >> (4 * (cp & 7))) & 15u);
}
static int cls2_at(const uint8_t* t, size_t n, size_t p, size_t* len, uint32_t* cp) {
    uint32_t c = decode(t, n, p, len);
    *cp = c;
    return tk_oracle_class2(c);
}
static int in_A(int k) { return k == C2_U || k == C2_X || k == C2_M; }
static int in_B(int k) { return k == C2_W || k == C2_X || k == C2_M; }
static int is_letter2(int k) { return k == C2_U || k == C2_W || k == C2_X; }

/* [A]*[B]+ at p with backtracking: returns the end of the match, or p if it fails */
static size_t word1_at(const uint8_t* t, size_t n, size_t p) {
    /* the maximal run of A from p, remembering the last position inside it whose char is also in B (X or M) */
    size_t q = p, last_b = (size_t)-1;
    while (q < n) {
        size_t l; uint32_t c;
        int k = cls2_at(t, n, q, &l, &c);
        if (!in_A(k)) break;
        if (in_B(k)) last_b = q;
        q += l;
    }
    size_t k0 = (size_t)-1;
    if (q < n) {                       /* greedy A* took everything: does B+ start right behind it? */
        size_t l; uint32_t c;
        if (in_B(cls2_at(t, n, q, &l, &c))) k0 = q;
    }
    if (k0 == (size_t)-1) k0 = last_b; /* give A chars back until one of them can open B+ */
    if (k0 == (size_t)-1) return p;
    size_t e = k0;
    while (e < n) {
        size_t l; uint32_t c;
        if (!in_B(cls2_at(t, n, e, &l, &c))) break;
        e += l;
    }
    return e;
}
/* [A]+[B]* at p: end of the match, or p if it fails */
static size_t word2_at(const uint8_t* t, size_t n, size_t p) {
    size_t q = p;
    while (q < n) {
        size_t l; uint32_t c;
        if (!in_A(cls2_at(t, n, q, &l, &c))) break;
        q += l;
    }
    if (q == p) return p;
    while (q < n) {
        size_t l; uint32_t c;
        if (!in_B(cls2_at(t, n, q, &l, &c))) break;
        q += l;
    }
    return q;
}

static size_t match2_at(const uint8_t* t, size_t n, size_t pos) {
    size_t l0;
    uint32_t c0;
    const int k0 = cls2_at(t, n, pos, &l0, &c0);
    const int prefix_ok = !(is_crlf(c0) || is_letter2(k0) || k0 == C2_N);
    /* alt 1, alt 2: optional one-char prefix (greedy: with it first), then the word */
    for (int alt = 1; alt <= 2; ++alt) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            size_t p;
            if (attempt == 0) { if (!prefix_ok || pos + l0 >= n) continue; p = pos + l0; }
            else p = pos;
            const size_t e = alt == 1 ? word1_at(t, n, p) : word2_at(t, n, p);
            if (e > p) return e;
        }
    }
    /* alt 3: \p{N} */
    if (k0 == C2_N) return pos + l0;
    /* alt 4:  ?[^\s\p{L}\p{N}]+[\r\n/]*   (the class is O or M) */
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t p;
        if (attempt == 0) { if (c0 != ' ') continue; p = pos + 1; }
        else p = pos;
        size_t q = p;
        while (q < n) {
            size_t l; uint32_t c;
            int k = cls2_at(t, n, q, &l, &c);
            if (!(k == C2_O || k == C2_M)) break;
            q += l;
        }
        if (q > p) {
            while (q < n && (t[q] == '\r' || t[q] == '\n' || t[q] == '/')) ++q;
            return q;
        }
    }
    /* white space: the same three alternatives as the hard-coded pattern */
    {
        size_t e = pos, after_last_nl = 0, last_char = pos;
        int has_nl = 0;
        while (e < n) {
            size_t l; uint32_t c;
            if (cls2_at(t, n, e, &l, &c) != C2_S) break;
            last_char = e;
            e += l;
            if (is_crlf(c)) { has_nl = 1; after_last_nl = e; }
        }
        if (e == pos) return pos + l0;       /* (unreachable for valid UTF-8: every class is covered above) */
        if (has_nl) return after_last_nl;
        if (e == n) return e;
        if (last_char > pos) return last_char;
        return e;
    }
}

size_t tk_oracle_split_tekken(const uint8_t* text, size_t n, uint32_t* starts, size_t cap) {
    size_t pos = 0, k = 0;
    while (pos < n) {
        if (k < cap) starts[k] = (uint32_t)pos;
        ++k;
        pos = match2_at(text, n, pos);
    }
    return k;
}

size_t tk_oracle_split(const uint8_t* text, size_t n, uint32_t* starts, size_t cap) {
    size_t pos = 0, k = 0;
    while (pos < n) {
        if (k < cap) starts[k] = (uint32_t)pos;
        ++k;
        pos = match_at(text, n, pos);
    }
    return k;
}

/* ---------------------------------------------------------------------------------------
 * rank table: bytes -> rank (the FxHashMap<Vec<u8>,u32> of src/tekkenizer.rs:776-816)
 * ------------------------------------------------------------------------------------- */
struct tk_oracle {
    uint8_t* blob;
    uint32_t* offs;
    uint32_t n_ranks;
    uint32_t num_special, bos_id, eos_id;
    int pattern;     /* 0 = the hard-coded pattern (reference behaviour), 1 = the JSON pattern of row f-3 */
    uint32_t* slots; /* rank+1, 0 = empty */
    uint32_t mask;
    uint32_t byte_rank[256];
};

static uint32_t hash_bytes(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    h ^= h >> 29;
    return (uint32_t)(h ^ (h >> 32));
}

static uint32_t rank_of(const tk_oracle* o, const uint8_t* p, size_t n) {
    uint32_t s = hash_bytes(p, n) & o->mask;
    for (;;) {
        uint32_t v = o->slots[s];
        if (v == 0) return RANK_MAX;
        uint32_t r = v - 1;
        uint32_t a = o->offs[r], b = o->offs[r + 1];
        if ((size_t)(b - a) == n && memcmp(o->blob + a, p, n) == 0) return r;
        s = (s + 1) & o->mask;
    }
}

tk_oracle* tk_oracle_new(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks,
                         uint32_t num_special, uint32_t bos_id, uint32_t eos_id) {
    tk_oracle* o = (tk_oracle*)calloc(1, sizeof(*o));
    if (!o) return NULL;
    o->n_ranks = n_ranks;
    o->num_special = num_special;
    o->bos_id = bos_id;
    o->eos_id = eos_id;
    o->offs = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n_ranks + 1));
    memcpy(o->offs, offs, sizeof(uint32_t) * ((size_t)n_ranks + 1));
    o->blob = (uint8_t*)malloc(offs[n_ranks] ? offs[n_ranks] : 1);
    memcpy(o->blob, blob, offs[n_ranks]);
    uint32_t cap = 1024;
    while (cap < 4u * n_ranks) cap <<= 1;
    o->mask = cap - 1;
    o->slots = (uint32_t*)calloc(cap, sizeof(uint32_t));
    for (int b = 0; b < 256; ++b) o->byte_rank[b] = RANK_MAX;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        const uint8_t* p = o->blob + offs[r];
        size_t n = offs[r + 1] - offs[r];
        uint32_t s = hash_bytes(p, n) & o->mask;
        /* a later duplicate key replaces the earlier one, like HashMap::insert
         * (src/tekkenizer.rs:801); the loader rejects such tables anyway (:804-813) */
        int replaced = 0;
        while (o->slots[s]) {
            uint32_t q = o->slots[s] - 1;
            if ((size_t)(offs[q + 1] - offs[q]) == n && memcmp(o->blob + offs[q], p, n) == 0) {
                o->slots[s] = r + 1; replaced = 1; break;
            }
            s = (s + 1) & o->mask;
        }
        if (!replaced) o->slots[s] = r + 1;
        if (n == 1) o->byte_rank[p[0]] = r;
    }
    return o;
}

void tk_oracle_set_pattern(tk_oracle* o, int pattern) { if (o) o->pattern = pattern ? 1 : 0; }

void tk_oracle_free(tk_oracle* o) {
    if (!o) return;
    free(o->blob); free(o->offs); free(o->slots); free(o);
}

/* byte_pair_encode / _byte_pair_merge of tiktoken-rs: parts are
 * (start, rank of the pair starting here); repeatedly merge the leftmost minimum-rank
 * pair; ranks are looked up by the concatenated BYTES. */
typedef struct { uint32_t start, rank; } part_t;

static size_t bpe_piece(const tk_oracle* o, const uint8_t* p, size_t n, uint32_t* out, size_t cap,
                        size_t k, part_t** scratch, size_t* scratch_cap) {
    if (n == 1) {
        if (k < cap) out[k] = o->byte_rank[p[0]];
        return k + 1;
    }
    if (*scratch_cap < n + 2) {
        *scratch_cap = 2 * (n + 2);
        *scratch = (part_t*)realloc(*scratch, *scratch_cap * sizeof(part_t));
    }
    part_t* parts = *scratch;
    size_t np = 0;
    uint32_t min_rank = RANK_MAX;
    size_t min_i = (size_t)-1;
    for (size_t i = 0; i + 1 < n; ++i) {
        uint32_t r = rank_of(o, p + i, 2);
        if (r < min_rank) { min_rank = r; min_i = i; }
        parts[np].start = (uint32_t)i; parts[np].rank = r; ++np;
    }
    parts[np].start = (uint32_t)(n - 1); parts[np].rank = RANK_MAX; ++np;
    parts[np].start = (uint32_t)n; parts[np].rank = RANK_MAX; ++np;

    while (min_rank != RANK_MAX) {
        size_t i = min_i;
        /* get_rank(parts, j) = rank(piece[parts[j].start .. parts[j+3].start)) if j+3 < len */
        if (i > 0) {
            size_t j = i - 1;
            parts[j].rank = (j + 3 < np) ? rank_of(o, p + parts[j].start, parts[j + 3].start - parts[j].start) : RANK_MAX;
        }
        parts[i].rank = (i + 3 < np) ? rank_of(o, p + parts[i].start, parts[i + 3].start - parts[i].start) : RANK_MAX;
        memmove(parts + i + 1, parts + i + 2, (np - i - 2) * sizeof(part_t));
        --np;
        min_rank = RANK_MAX; min_i = (size_t)-1;
        for (size_t j = 0; j + 1 < np; ++j)
            if (parts[j].rank < min_rank) { min_rank = parts[j].rank; min_i = j; }
    }
    for (size_t j = 0; j + 1 < np; ++j) {
        uint32_t r = rank_of(o, p + parts[j].start, parts[j + 1].start - parts[j].start);
        if (k < cap) out[k] = r;
        ++k;
    }
    return k;
}

static size_t encode_doc(const tk_oracle* o, const uint8_t* text, size_t n, int add_bos, int add_eos,
                         uint32_t* out, size_t cap, part_t** scratch, size_t* scratch_cap) {
    size_t k = 0;
    if (add_bos) { if (k < cap) out[k] = o->bos_id; ++k; }          /* src/tekkenizer.rs:394-397 */
    size_t first = k;
    size_t pos = 0;
    while (pos < n) {                                                 /* regex.find_iter(text) */
        size_t end = o->pattern ? match2_at(text, n, pos) : match_at(text, n, pos);
        uint32_t r = rank_of(o, text + pos, end - pos);               /* whole-piece shortcut */
        if (r != RANK_MAX) { if (k < cap) out[k] = r; ++k; }
        else k = bpe_piece(o, text + pos, end - pos, out, cap, k, scratch, scratch_cap);
        pos = end;
    }
    for (size_t j = first; j < k && j < cap; ++j) out[j] += o->num_special;   /* :390-392 */
    if (add_eos) { if (k < cap) out[k] = o->eos_id; ++k; }           /* :399-402 */
    return k;
}

size_t tk_oracle_encode(const tk_oracle* o, const uint8_t* text, size_t n, int add_bos, int add_eos,
                        uint32_t* out, size_t cap) {
    part_t* scratch = NULL; size_t sc = 0;
    size_t k = encode_doc(o, text, n, add_bos, add_eos, out, cap, &scratch, &sc);
    free(scratch);
    return k;
}

typedef struct {
    const tk_oracle* o; const uint8_t* bytes; const uint64_t* offs; uint64_t d0, d1;
    int add_bos, add_eos; uint32_t* stage; uint32_t* counts;
    uint32_t* out_ids; const uint64_t* out_offs;   /* second phase: pack the documents of [d0, d1) */
} job_t;

/* each doc d writes into stage[offs[d] + 2d ...] (capacity len+2), counts[d] = #ids */
static void* job_main(void* arg) {
    job_t* j = (job_t*)arg;
    part_t* scratch = NULL; size_t sc = 0;
    for (uint64_t d = j->d0; d < j->d1; ++d) {
        uint64_t a = j->offs[d], b = j->offs[d + 1];
        j->counts[d] = (uint32_t)encode_doc(j->o, j->bytes + a, (size_t)(b - a), j->add_bos, j->add_eos,
                                             j->stage + a + 2 * d, (size_t)(b - a) + 2, &scratch, &sc);
    }
    free(scratch);
    return NULL;
}

static void* pack_main(void* arg) {
    job_t* j = (job_t*)arg;
    for (uint64_t d = j->d0; d < j->d1; ++d)
        memcpy(j->out_ids + j->out_offs[d], j->stage + j->offs[d] + 2 * d, sizeof(uint32_t) * j->counts[d]);
    return NULL;
}

static double g_last_batch_seconds = 0.0;
static double now_seconds(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
/* wall time spent inside the last tk_oracle_encode_batch call of this process (what bench.py reports as the CPU
   baseline: the binding's own buffer handling is not part of it) */
double tk_oracle_last_batch_seconds(void) { return g_last_batch_seconds; }

uint64_t tk_oracle_encode_batch(const tk_oracle* o, const uint8_t* bytes, const uint64_t* offs,
                                uint64_t n_docs, int add_bos, int add_eos, uint32_t* out_ids,
                                uint64_t* out_offs, int n_threads) {
    const double t_begin = now_seconds();
    if (n_threads <= 1) {
        /* the plain host loop of BASELINE.md section 2 (one encode() per document) */
        part_t* scratch = NULL; size_t sc = 0;
        uint64_t t = 0;
        for (uint64_t d = 0; d < n_docs; ++d) {
            out_offs[d] = t;
            uint64_t a = offs[d], b = offs[d + 1];
            t += encode_doc(o, bytes + a, (size_t)(b - a), add_bos, add_eos, out_ids + t,
                            (size_t)(b - a) + 2, &scratch, &sc);
        }
        out_offs[n_docs] = t;
        free(scratch);
        g_last_batch_seconds = now_seconds() - t_begin;
        return t;
    }
    /* N threads: contiguous document ranges; every document is encoded into its own slot of a staging buffer, a prefix
       sum over the counts gives the offsets, and the same threads pack their ranges (no serial pass over the ids) */
    uint64_t n_bytes = offs[n_docs];
    /* the staging buffer is kept between calls (grown on demand): a fresh 2 GB malloc per call is 500 k page faults that
       256 threads take on one address space -- the second pass of a timing run then measures the encode, not the kernel's mm */
    /* (one call at a time uses the kept buffers: the bindings release the GIL, two N-thread calls of one process are serialised here) */
    static pthread_mutex_t g_stage_mu = PTHREAD_MUTEX_INITIALIZER;
    static uint32_t* g_stage = NULL; static uint64_t g_stage_cap = 0;
    static uint32_t* g_counts = NULL; static uint64_t g_counts_cap = 0;
    pthread_mutex_lock(&g_stage_mu);
    if (g_stage_cap < n_bytes + 2 * n_docs + 1) { free(g_stage); g_stage_cap = n_bytes + 2 * n_docs + 1; g_stage = (uint32_t*)malloc(sizeof(uint32_t) * g_stage_cap); }
    if (g_counts_cap < n_docs + 1) { free(g_counts); g_counts_cap = n_docs + 1; g_counts = (uint32_t*)malloc(sizeof(uint32_t) * g_counts_cap); }
    uint32_t* stage = g_stage;
    uint32_t* counts = g_counts;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * n_threads);
    /* contiguous document ranges cut by BYTES, like the GPU side shards (by document count the thread that draws the 32 KiB
       documents of the Zipf shape finishes long after the others and the N-thread baseline is understated) */
    uint64_t d_lo = 0;
    for (int i = 0; i < n_threads; ++i) {
        uint64_t d_hi = n_docs;
        if (i + 1 < n_threads) {
            const uint64_t want = offs[0] + (n_bytes - offs[0]) / (uint64_t)n_threads * (uint64_t)(i + 1);
            uint64_t lo = d_lo, hi = n_docs;              /* first document that starts at or beyond `want` */
            while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (offs[mid] < want) lo = mid + 1; else hi = mid; }
            d_hi = lo;
        }
        jobs[i] = (job_t){o, bytes, offs, d_lo, d_hi, add_bos, add_eos, stage, counts, out_ids, out_offs};
        d_lo = d_hi;
        pthread_create(&th[i], NULL, job_main, &jobs[i]);
    }
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    uint64_t t = 0;
    for (uint64_t d = 0; d < n_docs; ++d) { out_offs[d] = t; t += counts[d]; }
    out_offs[n_docs] = t;
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, pack_main, &jobs[i]);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th); free(jobs);
    pthread_mutex_unlock(&g_stage_mu);
    g_last_batch_seconds = now_seconds() - t_begin;
    return t;
}

/* out[0] = pieces, out[1] = pieces that are not a vocabulary key (they go through the merge loop), out[2] = bytes of those
   pieces, out[3] = ids those pieces produce -- what bench.py states beside a throughput figure (a vocabulary that misses
   30 % of the pieces and one that misses 3 % are different workloads for the merge kernels) */
void tk_oracle_miss_stats(const tk_oracle* o, const uint8_t* bytes, const uint64_t* offs, uint64_t n_docs, uint64_t* out) {
    part_t* scratch = NULL; size_t sc = 0;
    uint32_t* tmp = NULL; size_t tcap = 0;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        const uint8_t* text = bytes + offs[d];
        const size_t n = (size_t)(offs[d + 1] - offs[d]);
        size_t pos = 0;
        while (pos < n) {
            size_t end = o->pattern ? match2_at(text, n, pos) : match_at(text, n, pos);
            ++out[0];
            if (rank_of(o, text + pos, end - pos) == RANK_MAX) {
                ++out[1];
                out[2] += end - pos;
                if (end - pos > tcap) { tcap = 2 * (end - pos); tmp = (uint32_t*)realloc(tmp, tcap * sizeof(uint32_t)); }
                out[3] += bpe_piece(o, text + pos, end - pos, tmp, tcap, 0, &scratch, &sc);
            }
            pos = end;
        }
    }
    free(scratch); free(tmp);
}

/* One record per missed piece (for the analyses of tools/miss_analysis.py: memo hit rates, hole reservations): rec[3 i] = FNV-1a
   64 of the piece's bytes folded to 32 bits ^ its upper half, rec[3 i + 1] = bytes, rec[3 i + 2] = ids it produces.  Returns the
   number of missed pieces (only cap records are written). */
uint64_t tk_oracle_miss_records(const tk_oracle* o, const uint8_t* bytes, const uint64_t* offs, uint64_t n_docs, uint32_t* rec, uint64_t cap) {
    part_t* scratch = NULL; size_t sc = 0;
    uint32_t* tmp = NULL; size_t tcap = 0;
    uint64_t nrec = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        const uint8_t* text = bytes + offs[d];
        const size_t n = (size_t)(offs[d + 1] - offs[d]);
        size_t pos = 0;
        while (pos < n) {
            size_t end = o->pattern ? match2_at(text, n, pos) : match_at(text, n, pos);
            if (rank_of(o, text + pos, end - pos) == RANK_MAX) {
                if (end - pos > tcap) { tcap = 2 * (end - pos); tmp = (uint32_t*)realloc(tmp, tcap * sizeof(uint32_t)); }
                const size_t k = bpe_piece(o, text + pos, end - pos, tmp, tcap, 0, &scratch, &sc);
                if (nrec < cap) {
                    uint64_t h = 1469598103934665603ull;
                    for (size_t i = pos; i < end; ++i) { h ^= text[i]; h *= 1099511628211ull; }
                    rec[3 * nrec] = (uint32_t)h ^ (uint32_t)(h >> 32);
                    rec[3 * nrec + 1] = (uint32_t)(end - pos);
                    rec[3 * nrec + 2] = (uint32_t)k;
                }
                ++nrec;
            }
            pos = end;
        }
    }
    free(scratch); free(tmp);
    return nrec;
}

uint64_t tk_oracle_fnv1a(const uint32_t* ids, uint64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t v = ids[i];
        for (int k = 0; k < 4; ++k) { h ^= (v >> (8 * k)) & 0xFF; h *= 1099511628211ull; }
    }
    return h;
}
