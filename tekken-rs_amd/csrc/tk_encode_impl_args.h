// tk_encode_impl_args.h -- argument block of the encode kernels (host + device view).
#ifndef TK_ENCODE_IMPL_ARGS_H
#define TK_ENCODE_IMPL_ARGS_H
#include <stdint.h>

#include "tk_hash.h"
#include "tk_tables.h"

struct TkLongJob { uint32_t doc, off, len, pad; };   // a long piece that is not a vocabulary key: bytes [off, off + len) of document doc

struct TkEncodeArgs {
    const uint8_t* bytes;       // packed text of all documents
    const uint64_t* doc_offs;   // [n_docs + 1]
    uint64_t n_docs;
    uint32_t* staging;          // [n_bytes + 2*n_docs]: doc d writes from doc_offs[d] + 2*d
    uint32_t* counts;           // [n_docs] ids produced per document
    uint32_t* work_counter;     // dynamic work queue head
    uint32_t* defer_list;       // pass 1: documents handed to pass 2
    uint32_t* defer_count;
    const uint32_t* todo_list;  // pass 2: the documents to process
    uint32_t n_todo;
    const uint32_t* n_todo_dev; // not NULL: the length of todo_list is read from here (a count that never left the device) instead of n_todo
    uint8_t* dbg_starts;        // optional: per-byte piece-start flags (tk_split_batch)
    volatile uint32_t* dbg_mark; // optional: progress marks of the long-piece merge (debug builds of the host)
    uint32_t* long_list;        // pass 2: documents with a long piece that misses the vocabulary, handed on to tk_long.hip (NULL: merged here)
    uint32_t* long_count;
    uint32_t long_min;          // shortest piece (bytes) that counts as long
    uint32_t long_lazy_mul;     // a piece with many distinct pairs takes the lazy rounds from long_lazy_mul * long_min bytes on (0 = 2)
    uint32_t long_force;        // tests: every long piece takes the round-based merge, repetitive or not
    TkLongJob* long_jobs;       // tk_long.hip: the long pieces of the documents of the long list
    uint32_t* long_job_count;
    uint32_t long_job_cap;
    uint32_t* scratch;          // pass 2: per-wave scratch for the cooperative merge
    uint64_t scratch_words_per_wave;
    int add_bos, add_eos;
    int split_only;
    int pattern;                // 0: the hard-coded pattern (reference behaviour); 1: the JSON pattern of tekken.json (row f-3, pass 2 only)
    int dbg_ablate;             // timing-only ablation bits (TK_DEBUG_ABLATE): 1 skip probes, 2 skip merges, 4 skip id stores
    TkTablesView t;
};

#endif
