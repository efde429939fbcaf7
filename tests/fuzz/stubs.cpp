// Host-only build of the loader for the fuzzer (tests/fuzz/fuzz_loaders.cpp): the engine entry points the host-side
// Tekkenizer mirror links against, answering "no device".  Test infrastructure -- the product library has no such stubs
// (tk_capi.cpp fails loudly without a GPU); the fuzzer only ever builds host-only objects (device = -1).
#include <string>

#include "../../include/tekken_hip.h"

static thread_local std::string g_err;
void tk_set_tls_error(const std::string& e) { g_err = e; }
const std::string& tk_get_tls_error() { return g_err; }
extern "C" {
int tk_ctx_create(const uint8_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, int, tk_ctx**) { return TK_ERR_NO_DEVICE; }
void tk_ctx_destroy(tk_ctx*) {}
const char* tk_last_error(const tk_ctx*) { return g_err.c_str(); }
int tk_encode_batch(tk_ctx*, const uint8_t*, const uint64_t*, uint64_t, int, int, int, tk_result*) { return TK_ERR_NO_DEVICE; }
int tk_encode_one(tk_ctx*, const uint8_t*, uint64_t, int, int, uint32_t*, uint64_t, uint64_t*) { return TK_ERR_NO_DEVICE; }
int tk_decode_batch(tk_ctx*, const uint32_t*, const uint64_t*, uint64_t, int, tk_text_result*, uint64_t*) { return TK_ERR_NO_DEVICE; }
int tk_ctx_set_special_tokens(tk_ctx*, const uint8_t*, const uint32_t*, uint32_t) { return TK_ERR_NO_DEVICE; }
int tk_ctx_set_pattern(tk_ctx*, int) { return TK_ERR_NO_DEVICE; }
}
