// tk_engine.h -- internal glue between the engine-level C ABI (tk_capi.cpp) and the host-side
// Tekkenizer mirror (tekkenizer.cpp).  Not part of the public interface.
#ifndef TK_ENGINE_H
#define TK_ENGINE_H
#include <string>

#include "../../include/tekken_hip.h"
#include "tk_tables.h"

void tk_set_tls_error(const std::string& e);
const std::string& tk_get_tls_error();
const TkHostTables* tk_ctx_host_tables(const tk_ctx* c);
// pinned host blocks of the result structs (tk_result / tk_text_result): a process-wide pool, see tk_capi.cpp
void* tk_pinned_get(size_t bytes);
void tk_pinned_put(void* p);

#endif
