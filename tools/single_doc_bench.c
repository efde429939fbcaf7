/* single_doc_bench.c -- bench.py's single_doc leg: N sequential tk_encode_one calls (one document each) timed in C so that
 * no Python sits in the timed region.  The entry point is passed in as a function pointer (no link dependency on the
 * library; the Python process has it loaded already).  Bench infrastructure, not product. */
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef int (*encode_one_fn)(void* ctx, const uint8_t* text, uint64_t len, int add_bos, int add_eos, uint32_t* ids_out,
                             uint64_t ids_capacity, uint64_t* n_ids);

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* `passes` timed passes over the documents after one warm-up pass; per_call[d] = the fastest of the passes for document d
 * (seconds); *fnv / *total = FNV-1a and count of all ids of the last pass (the oracle's tk_oracle_fnv1a) */
int tkb_single_doc_loop(void* fn_, void* ctx, const uint8_t* bytes, const uint64_t* offs, uint64_t n_docs, int passes,
                        double* per_call, uint64_t* fnv, uint64_t* total) {
    encode_one_fn fn = (encode_one_fn)fn_;
    uint64_t cap = 0;
    for (uint64_t d = 0; d < n_docs; ++d)
        if (offs[d + 1] - offs[d] + 2 > cap) cap = offs[d + 1] - offs[d] + 2;
    uint32_t* ids = (uint32_t*)malloc(cap * 4);
    if (!ids) return -100;
    for (uint64_t d = 0; d < n_docs; ++d) per_call[d] = 1e9;
    for (int p = 0; p <= passes; ++p) {
        uint64_t h = 1469598103934665603ull, tot = 0;
        for (uint64_t d = 0; d < n_docs; ++d) {
            uint64_t n = 0;
            const double t0 = now_s();
            const int rc = fn(ctx, bytes + offs[d], offs[d + 1] - offs[d], 1, 1, ids, cap, &n);
            const double dt = now_s() - t0;
            if (rc != 0) { free(ids); return rc; }
            if (p > 0 && dt < per_call[d]) per_call[d] = dt;
            for (uint64_t i = 0; i < n; ++i)
                for (int k = 0; k < 4; ++k) { h ^= (ids[i] >> (8 * k)) & 0xFF; h *= 1099511628211ull; }
            tot += n;
        }
        *fnv = h;
        *total = tot;
    }
    free(ids);
    return 0;
}
