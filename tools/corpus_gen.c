/*
 * corpus_gen.c -- seeded synthetic UTF-8 corpora of the shapes BASELINE.json names
 * (SURVEY.md section 8d).  Bench / test infrastructure; nothing here is reference code.
 *
 *   kind 0  G-ascii : English-like word stream, exactly doc_len bytes per document
 *   kind 1  G-mixed : multi-script valid UTF-8 (2/3/4-byte code points, Unicode white space,
 *                     non-ASCII digits, contractions), exactly doc_len bytes per document
 *   kind 2  G-zipf  : G-ascii content with lengths drawn from P(len) ~ len^-1.2 on [16, 32768];
 *                     0.1 % of documents are ONE letter run, 0.1 % ONE white-space run
 *
 * Every document is generated from its own PRNG state (xoshiro256** seeded with
 * splitmix64(seed, doc index)), so any rank can generate any shard independently and a
 * sample of the corpus is a prefix of it.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_WORDS 4096
#define MAX_WORD 16

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix64(uint64_t* x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t next(rng_t* r) {
    uint64_t* s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static void seed_rng(rng_t* r, uint64_t seed, uint64_t idx) {
    uint64_t x = seed ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&x);
}
static uint32_t below(rng_t* r, uint32_t n) { return (uint32_t)((next(r) >> 32) * (uint64_t)n >> 32); }
static double unit(rng_t* r) { return (double)(next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* ---- the embedded word list: 4096 pseudo-English lowercase words from a syllable model ---- */
static char g_words[N_WORDS][MAX_WORD];
static uint8_t g_wlen[N_WORDS];
static double g_zipf_cdf[N_WORDS];
static int g_init = 0;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static void init_words(void) {
    pthread_mutex_lock(&g_mu);
    if (g_init) { pthread_mutex_unlock(&g_mu); return; }
    static const char* onset[] = {"", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w",
                                  "y", "z", "th", "sh", "ch", "st", "tr", "pr", "br", "cl", "gr", "pl", "fr", "wh", "qu",
                                  "sp", "sl", "dr"};
    static const char* nucleus[] = {"a", "e", "i", "o", "u", "ea", "ou", "io", "ai", "ee", "oo", "ie", "au"};
    static const char* coda[] = {"", "", "", "n", "r", "s", "t", "l", "d", "m", "ng", "nt", "st", "ck", "ll", "rd", "ss",
                                 "nd", "ly", "er", "ed"};
    static const char* common[] = {"the", "of", "and", "to", "a", "in", "is", "that", "it", "was", "for", "on", "are",
                                   "as", "with", "his", "they", "at", "be", "this", "have", "from", "or", "one", "had",
                                   "by", "word", "but", "not", "what", "all", "were", "we", "when", "your", "can",
                                   "said", "there", "use", "an", "each", "which", "she", "do", "how", "their", "if",
                                   "will", "up", "other", "about", "out", "many", "then", "them", "these", "so", "some",
                                   "her", "would", "make", "like", "him", "into", "time", "has", "look", "two", "more",
                                   "write", "go", "see", "number", "no", "way", "could", "people", "my", "than", "first",
                                   "water", "been", "call", "who", "oil", "its", "now", "find", "long", "down", "day",
                                   "did", "get", "come", "made", "may", "part"};
    const int n_common = (int)(sizeof(common) / sizeof(common[0]));
    rng_t r;
    seed_rng(&r, 0x7E44E2ull, 0xC0FFEEull);
    int n = 0;
    for (; n < n_common && n < N_WORDS; ++n) {
        strncpy(g_words[n], common[n], MAX_WORD - 1);
        g_wlen[n] = (uint8_t)strlen(g_words[n]);
    }
    while (n < N_WORDS) {
        char w[64];
        int len = 0;
        int syl = 1 + (int)below(&r, 3) + (n > 1024 ? (int)below(&r, 2) : 0);
        for (int s = 0; s < syl; ++s) {
            const char* a = onset[below(&r, sizeof(onset) / sizeof(onset[0]))];
            const char* b = nucleus[below(&r, sizeof(nucleus) / sizeof(nucleus[0]))];
            const char* c = coda[below(&r, sizeof(coda) / sizeof(coda[0]))];
            len += snprintf(w + len, sizeof(w) - (size_t)len, "%s%s%s", a, b, c);
        }
        if (len < 2 || len >= MAX_WORD) continue;
        int dup = 0;
        for (int k = 0; k < n && !dup; ++k) dup = (strcmp(g_words[k], w) == 0);
        if (dup) continue;
        strcpy(g_words[n], w);
        g_wlen[n] = (uint8_t)len;
        ++n;
    }
    double tot = 0;
    for (int i = 0; i < N_WORDS; ++i) tot += 1.0 / (double)(i + 1);
    double acc = 0;
    for (int i = 0; i < N_WORDS; ++i) { acc += 1.0 / (double)(i + 1) / tot; g_zipf_cdf[i] = acc; }
    g_zipf_cdf[N_WORDS - 1] = 1.0;
    g_init = 1;
    pthread_mutex_unlock(&g_mu);
}

static int zipf_word(rng_t* r) {
    double u = unit(r);
    int lo = 0, hi = N_WORDS - 1;
    while (lo < hi) { int mid = (lo + hi) / 2; if (g_zipf_cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

int tkc_n_words(void) { init_words(); return N_WORDS; }
const char* tkc_word(int i) { init_words(); return (i >= 0 && i < N_WORDS) ? g_words[i] : ""; }

/* bounded writer */
typedef struct { uint8_t* p; uint64_t n, cap; } wr_t;
static void put(wr_t* w, const void* s, size_t len) {
    for (size_t i = 0; i < len && w->n < w->cap; ++i) w->p[w->n++] = ((const uint8_t*)s)[i];
}
static void putc_(wr_t* w, uint8_t c) { if (w->n < w->cap) w->p[w->n++] = c; }
static size_t utf8_put(uint8_t* b, uint32_t cp) {
    if (cp < 0x80) { b[0] = (uint8_t)cp; return 1; }
    if (cp < 0x800) { b[0] = (uint8_t)(0xC0 | (cp >> 6)); b[1] = (uint8_t)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) {
        b[0] = (uint8_t)(0xE0 | (cp >> 12)); b[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); b[2] = (uint8_t)(0x80 | (cp & 0x3F));
        return 3;
    }
    b[0] = (uint8_t)(0xF0 | (cp >> 18)); b[1] = (uint8_t)(0x80 | ((cp >> 12) & 0x3F));
    b[2] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); b[3] = (uint8_t)(0x80 | (cp & 0x3F));
    return 4;
}
/* writes cp only if it fits entirely (never splits a code point) */
static int put_cp(wr_t* w, uint32_t cp) {
    uint8_t b[4];
    size_t l = utf8_put(b, cp);
    if (w->n + l > w->cap) return 0;
    memcpy(w->p + w->n, b, l);
    w->n += l;
    return 1;
}

static const char SYMS[] = "!@#$%^&*()_+-={}[]|\\:;\"'<>,.?/";
static const char* CONTR[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};

static void ascii_word(rng_t* r, wr_t* w) {
    int wi = zipf_word(r);
    uint32_t c = below(r, 100);
    char buf[MAX_WORD];
    int len = g_wlen[wi];
    memcpy(buf, g_words[wi], (size_t)len);
    if (c < 2) for (int i = 0; i < len; ++i) buf[i] = (char)(buf[i] - 32);
    else if (c < 12) buf[0] = (char)(buf[0] - 32);
    put(w, buf, (size_t)len);
}

static void ascii_sep(rng_t* r, wr_t* w) {
    uint32_t c = below(r, 100);
    if (c < 82) putc_(w, ' ');
    else if (c < 87) put(w, ", ", 2);
    else if (c < 92) put(w, ". ", 2);
    else if (c < 95) putc_(w, '\n');
    else if (c < 96) put(w, "\n\n", 2);
    else if (c < 98) {
        putc_(w, ' ');
        int nd = 1 + (int)below(r, 6);
        for (int i = 0; i < nd; ++i) putc_(w, (uint8_t)('0' + below(r, 10)));
        putc_(w, ' ');
    } else if (c < 99) {
        putc_(w, ' ');
        int ns = 1 + (int)below(r, 8);
        for (int i = 0; i < ns; ++i) putc_(w, (uint8_t)SYMS[below(r, sizeof(SYMS) - 1)]);
        putc_(w, ' ');
    } else {
        const char* s = CONTR[below(r, 7)];
        put(w, s, strlen(s));
        putc_(w, ' ');
    }
}

static void gen_ascii(rng_t* r, uint8_t* out, uint64_t len) {
    wr_t w = {out, 0, len};
    while (w.n < w.cap) { ascii_word(r, &w); ascii_sep(r, &w); }
}

static uint32_t pick_range(rng_t* r, uint32_t lo, uint32_t hi) { return lo + below(r, hi - lo + 1); }

static void gen_mixed(rng_t* r, uint8_t* out, uint64_t len) {
    wr_t w = {out, 0, len};
    /* per-document script weights: ASCII is always the largest share (~40 % of bytes overall) */
    uint32_t wt[5];
    wt[0] = 30 + below(r, 40);  /* ascii words */
    wt[1] = below(r, 25);       /* latin + accents (2-byte letters mixed in) */
    wt[2] = below(r, 25);       /* cyrillic / greek / arabic (2-byte) */
    wt[3] = below(r, 25);       /* cjk / hangul / thai (3-byte) */
    wt[4] = below(r, 12);       /* emoji + math symbols */
    uint32_t tot = wt[0] + wt[1] + wt[2] + wt[3] + wt[4];
    int guard = 0;
    while (w.n < w.cap && guard < 8) {
        uint64_t before = w.n;
        uint32_t c = below(r, tot);
        if (c < wt[0]) {
            ascii_word(r, &w);
        } else if ((c -= wt[0]) < wt[1]) {
            int wi = zipf_word(r);
            for (int i = 0; i < g_wlen[wi]; ++i) {
                char ch = g_words[wi][i];
                if ((ch == 'a' || ch == 'e' || ch == 'o' || ch == 'u' || ch == 'n') && below(r, 3) == 0) {
                    static const uint32_t acc[] = {0xE0, 0xE1, 0xE4, 0xE8, 0xE9, 0xEA, 0xF1, 0xF3, 0xF6, 0xFC, 0xFA, 0xC9, 0xD6};
                    if (!put_cp(&w, acc[below(r, 13)])) break;
                } else putc_(&w, (uint8_t)ch);
            }
        } else if ((c -= wt[1]) < wt[2]) {
            uint32_t script = below(r, 3);
            int n = 2 + (int)below(r, 9);
            for (int i = 0; i < n; ++i) {
                uint32_t cp = script == 0 ? pick_range(r, 0x0430, 0x044F) : script == 1 ? pick_range(r, 0x03B1, 0x03C9)
                                                                                      : pick_range(r, 0x0627, 0x063A);
                if (i == 0 && script == 0 && below(r, 5) == 0) cp -= 0x20; /* capital cyrillic */
                if (!put_cp(&w, cp)) break;
            }
        } else if ((c -= wt[2]) < wt[3]) {
            uint32_t script = below(r, 3);
            int n = 1 + (int)below(r, 10);
            for (int i = 0; i < n; ++i) {
                uint32_t cp = script == 0 ? pick_range(r, 0x4E00, 0x9FA5) : script == 1 ? pick_range(r, 0xAC00, 0xD7A3)
                                                                                      : pick_range(r, 0x0E01, 0x0E2E);
                if (!put_cp(&w, cp)) break;
            }
        } else {
            int n = 1 + (int)below(r, 3);
            for (int i = 0; i < n; ++i) {
                uint32_t cp = below(r, 2) ? pick_range(r, 0x1F600, 0x1F64F) : pick_range(r, 0x2200, 0x22FF);
                if (!put_cp(&w, cp)) break;
            }
        }
        /* separator */
        uint32_t s = below(r, 100);
        if (s < 70) putc_(&w, ' ');
        else if (s < 75) put(&w, ", ", 2);
        else if (s < 79) put(&w, ". ", 2);
        else if (s < 82) putc_(&w, '\n');
        else if (s < 84) put(&w, "\r\n", 2);
        else if (s < 86) (void)put_cp(&w, 0x00A0);
        else if (s < 87) (void)put_cp(&w, 0x2028);
        else if (s < 89) (void)put_cp(&w, 0x3000);
        else if (s < 90) (void)put_cp(&w, 0x3002);           /* ideographic full stop */
        else if (s < 91) put(&w, "\t", 1);
        else if (s < 94) {                                    /* non-ASCII numbers */
            putc_(&w, ' ');
            int nd = 1 + (int)below(r, 5);
            uint32_t kind = below(r, 4);
            for (int i = 0; i < nd; ++i) {
                uint32_t cp = kind == 0 ? pick_range(r, 0x0660, 0x0669) : kind == 1 ? pick_range(r, 0xFF10, 0xFF19)
                              : kind == 2 ? (uint32_t)('0' + below(r, 10))
                                          : (below(r, 3) == 0 ? 0x00B2u : below(r, 2) ? 0x2167u : 0x00BDu);
                if (!put_cp(&w, cp)) break;
            }
            putc_(&w, ' ');
        } else if (s < 96) {                                  /* contractions incl. U+017F and capitals */
            uint32_t k = below(r, 9);
            if (k == 7) { putc_(&w, '\''); (void)put_cp(&w, 0x017F); }
            else if (k == 8) put(&w, "'S", 2);
            else put(&w, CONTR[k], strlen(CONTR[k]));
            putc_(&w, ' ');
        } else if (s < 98) {
            putc_(&w, ' ');
            int ns = 1 + (int)below(r, 4);
            for (int i = 0; i < ns; ++i) putc_(&w, (uint8_t)SYMS[below(r, sizeof(SYMS) - 1)]);
            if (below(r, 3) == 0) putc_(&w, '\n');
        } else {
            (void)put_cp(&w, 0x0301);                         /* combining acute: a mark, class "other" */
            putc_(&w, ' ');
        }
        guard = (w.n == before) ? guard + 1 : 0;
    }
    while (w.n < w.cap) w.p[w.n++] = ' ';  /* pad with spaces, never split a code point */
}

static uint64_t zipf_len(rng_t* r) {
    /* inverse CDF of p(x) ~ x^-1.2 on [16, 32768] */
    const double a = pow(16.0, -0.2), b = pow(32768.0, -0.2);
    double u = unit(r);
    double x = pow(a - u * (a - b), -5.0);
    uint64_t l = (uint64_t)x;
    if (l < 16) l = 16;
    if (l > 32768) l = 32768;
    return l;
}

static void gen_zipf_doc(rng_t* r, uint8_t* out, uint64_t len) {
    uint32_t special = below(r, 1000);
    if (special == 0) {
        for (uint64_t i = 0; i < len; ++i) out[i] = (uint8_t)('a' + below(r, 26));
    } else if (special == 1) {
        for (uint64_t i = 0; i < len; ++i) { uint32_t c = below(r, 40); out[i] = c == 0 ? '\n' : c == 1 ? '\t' : ' '; }
    } else gen_ascii(r, out, len);
}

uint64_t tkc_doc_len(int kind, uint64_t seed, uint64_t doc, uint64_t doc_len) {
    if (kind != 2) return doc_len;
    rng_t r;
    seed_rng(&r, seed, doc);
    return zipf_len(&r);
}

typedef struct { int kind; uint64_t seed, first, d0, d1, doc_len; uint8_t* out; const uint64_t* offs; } job_t;

static void* job_main(void* arg) {
    job_t* j = (job_t*)arg;
    for (uint64_t d = j->d0; d < j->d1; ++d) {
        rng_t r;
        seed_rng(&r, j->seed, j->first + d);
        uint8_t* o = j->out + j->offs[d];
        uint64_t len = j->offs[d + 1] - j->offs[d];
        if (j->kind == 0) gen_ascii(&r, o, len);
        else if (j->kind == 1) gen_mixed(&r, o, len);
        else { (void)zipf_len(&r); gen_zipf_doc(&r, o, len); }
    }
    return NULL;
}

/* offs[0..n_docs] must be filled by tkc_fill_offsets first; out holds offs[n_docs] bytes */
void tkc_fill_offsets(int kind, uint64_t seed, uint64_t first_doc, uint64_t n_docs, uint64_t doc_len, uint64_t* offs) {
    init_words();
    offs[0] = 0;
    for (uint64_t d = 0; d < n_docs; ++d) offs[d + 1] = offs[d] + tkc_doc_len(kind, seed, first_doc + d, doc_len);
}

void tkc_gen_docs(int kind, uint64_t seed, uint64_t first_doc, uint64_t n_docs, uint64_t doc_len, const uint64_t* offs,
                  uint8_t* out, int n_threads) {
    init_words();
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_docs) n_threads = n_docs ? (int)n_docs : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)n_threads);
    for (int i = 0; i < n_threads; ++i) {
        jobs[i] = (job_t){kind, seed, first_doc, n_docs * (uint64_t)i / (uint64_t)n_threads,
                          n_docs * (uint64_t)(i + 1) / (uint64_t)n_threads, doc_len, out, offs};
        pthread_create(&th[i], NULL, job_main, &jobs[i]);
    }
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}
