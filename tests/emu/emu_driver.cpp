// emu_driver.cpp -- runs the device algorithm (tk_encode_impl.h) on the CPU wave emulator.
// TEST INFRASTRUCTURE ONLY; see tk_wave_emu.h.  Built into tests/emu/libtk_emu.so by
// tests/emu/Makefile and loaded with ctypes from tests/test_kernel_emu.py.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "tk_wave_emu.h"
#include "../../include/tekken_hip.h"
#include "../../tekken-rs_amd/csrc/tk_encode_impl.h"

namespace tkemu {
Wave* g_wave = nullptr;
}

static std::string g_err;

extern "C" const char* emu_last_error() { return g_err.c_str(); }

// Full pipeline on the emulator: pass 1, pass 2 (deferred documents, with scratch), host scan
// and compaction.  out_ids must hold n_bytes + 2*n_docs entries.
extern "C" int emu_encode_batch(const uint8_t* blob, const uint32_t* offs, uint32_t n_ranks, uint32_t num_special,
                                uint32_t bos, uint32_t eos, const uint8_t* bytes, const uint64_t* doc_offs,
                                uint64_t n_docs, int add_bos, int add_eos, int split_only, uint32_t* out_ids,
                                uint64_t* out_offs, uint8_t* dbg_starts, uint64_t* n_deferred, uint64_t* n_ops) {
    TkHostTables T;
    int rc = tk_build_tables(blob, offs, n_ranks, num_special, bos, eos, T, g_err);
    if (rc != TK_OK) return rc;
    const uint64_t n_bytes = doc_offs[n_docs];
    std::vector<uint32_t> staging(n_bytes + 2 * n_docs + 1, 0xDEADBEEFu);
    std::vector<uint32_t> counts(n_docs + 1, 0);
    std::vector<uint32_t> defer_list(n_docs + 1, 0);
    uint32_t work_counter = 0, defer_count = 0;
    uint64_t maxlen = 0;
    for (uint64_t d = 0; d < n_docs; ++d) maxlen = std::max<uint64_t>(maxlen, doc_offs[d + 1] - doc_offs[d]);

    TkEncodeArgs a;
    memset(&a, 0, sizeof(a));
    a.bytes = bytes;
    a.doc_offs = doc_offs;
    a.n_docs = n_docs;
    a.staging = staging.data();
    a.counts = counts.data();
    a.work_counter = &work_counter;
    a.defer_list = defer_list.data();
    a.defer_count = &defer_count;
    a.dbg_starts = dbg_starts;
    a.add_bos = add_bos;
    a.add_eos = add_eos;
    a.split_only = split_only;
    a.t = T.host_view();

    uint64_t ops = 0;
    if (split_only) tkemu::run_wave([&](int lane) { tk_encode_wave<2>(a, lane, 0); });
    else tkemu::run_wave([&](int lane) { tk_encode_wave<0>(a, lane, 0); });
    ops += tkemu::g_wave->n_ops;
    if (n_deferred) *n_deferred = defer_count;
    if (defer_count) {
        std::vector<uint32_t> scratch_raw(4 * maxlen + 2 * ((maxlen + 63) / 64) + 64 + 8, 0);
        uint32_t* scratch_al = scratch_raw.data();
        while (reinterpret_cast<uintptr_t>(scratch_al) % 16) ++scratch_al;
        a.todo_list = defer_list.data();
        a.n_todo = defer_count;
        a.scratch = scratch_al;
        a.scratch_words_per_wave = scratch_raw.size() - 8;
        work_counter = 0;
        uint32_t dc2 = 0;
        a.defer_count = &dc2;
        tkemu::run_wave([&](int lane) { tk_encode_wave<1>(a, lane, 0); });
        if (dc2 != 0) { g_err = "pass 2 deferred a document"; return TK_ERR_RUNTIME; }
    }
    if (n_ops) *n_ops = tkemu::g_wave->n_ops;
    uint64_t t = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        out_offs[d] = t;
        if (!split_only) {
            memcpy(out_ids + t, staging.data() + doc_offs[d] + 2 * d, sizeof(uint32_t) * counts[d]);
            t += counts[d];
        }
    }
    out_offs[n_docs] = t;
    return TK_OK;
}
