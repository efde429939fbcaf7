// tekkenizer.cpp -- host-side mirror of the reference's Tekkenizer (see tekkenizer.hpp) and
// the tokenizer-level C ABI of include/tekken_hip.h.
#include "tekkenizer.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <unordered_set>

#include "tk_engine.h"

namespace tekken {

static TokenizerError mk(int code, const std::string& m) {
    TokenizerError e;
    e.code = code;
    e.message = m;
    return e;
}

// ------------------------------------------------------------------------------------------
// base64, STANDARD alphabet, canonical padding required (base64 0.22 general_purpose::STANDARD,
// used at reference src/tekkenizer.rs:789)
// ------------------------------------------------------------------------------------------
bool base64_decode_standard(const std::string& in, std::string& out, std::string& err) {
    static int8_t T[256];
    static bool init = false;
    if (!init) {
        memset(T, -1, sizeof(T));
        const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) T[(uint8_t)A[i]] = (int8_t)i;
        init = true;
    }
    out.clear();
    const size_t n = in.size();
    if (n % 4 != 0) { err = "Invalid input length " + std::to_string(n); return false; }
    for (size_t i = 0; i < n; i += 4) {
        int v[4];
        int pad = 0;
        for (int k = 0; k < 4; ++k) {
            const uint8_t c = (uint8_t)in[i + k];
            if (c == '=') {
                if (i + 4 != n || k < 2) { err = "Invalid padding"; return false; }
                v[k] = 0;
                ++pad;
            } else {
                if (pad) { err = "Invalid padding"; return false; }
                if (T[c] < 0) { err = "Invalid symbol " + std::to_string((int)c) + ", offset " + std::to_string(i + k); return false; }
                v[k] = T[c];
            }
        }
        const uint32_t w = ((uint32_t)v[0] << 18) | ((uint32_t)v[1] << 12) | ((uint32_t)v[2] << 6) | (uint32_t)v[3];
        out.push_back((char)(w >> 16));
        if (pad < 2) out.push_back((char)((w >> 8) & 0xFF));
        if (pad < 1) out.push_back((char)(w & 0xFF));
        if ((pad == 2 && (v[1] & 0xF)) || (pad == 1 && (v[2] & 0x3))) { err = "Invalid last symbol"; return false; }
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// UTF-8 helpers with Rust's semantics (String::from_utf8 / from_utf8_lossy)
// ------------------------------------------------------------------------------------------
// length of the well-formed sequence at p (0 if ill-formed); *bad_len = maximal ill-formed subpart
static int utf8_seq(const uint8_t* p, size_t n, int* bad_len) {
    const uint8_t b0 = p[0];
    *bad_len = 1;
    if (b0 < 0x80) return 1;
    if (b0 < 0xC2) return 0;
    if (b0 < 0xE0) {
        if (n >= 2 && (p[1] & 0xC0) == 0x80) return 2;
        return 0;
    }
    if (b0 < 0xF0) {
        if (n < 2) return 0;
        const uint8_t lo = b0 == 0xE0 ? 0xA0 : 0x80, hi = b0 == 0xED ? 0x9F : 0xBF;
        if (p[1] < lo || p[1] > hi) return 0;
        *bad_len = 2;
        if (n >= 3 && (p[2] & 0xC0) == 0x80) return 3;
        return 0;
    }
    if (b0 < 0xF5) {
        if (n < 2) return 0;
        const uint8_t lo = b0 == 0xF0 ? 0x90 : 0x80, hi = b0 == 0xF4 ? 0x8F : 0xBF;
        if (p[1] < lo || p[1] > hi) return 0;
        *bad_len = 2;
        if (n < 3 || (p[2] & 0xC0) != 0x80) return 0;
        *bad_len = 3;
        if (n >= 4 && (p[3] & 0xC0) == 0x80) return 4;
        return 0;
    }
    return 0;
}

bool utf8_valid(const uint8_t* p, size_t n) {
    size_t i = 0;
    while (i < n) {
        int bad;
        int l = utf8_seq(p + i, n - i, &bad);
        if (!l) return false;
        i += (size_t)l;
    }
    return true;
}

std::string utf8_lossy(const uint8_t* p, size_t n) {
    std::string out;
    size_t i = 0;
    while (i < n) {
        int bad;
        int l = utf8_seq(p + i, n - i, &bad);
        if (l) { out.append((const char*)p + i, (size_t)l); i += (size_t)l; }
        else { out.append("\xEF\xBF\xBD"); i += (size_t)bad; }
    }
    return out;
}

// ------------------------------------------------------------------------------------------
// minimal JSON reader (the subset serde_json needs for ModelData, reference src/config.rs)
// ------------------------------------------------------------------------------------------
namespace {

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    bool is_int = false, neg = false;
    uint64_t u = 0;
    double d = 0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal* get(const char* key) const {
        const JVal* r = nullptr;
        for (auto& kv : o)
            if (kv.first == key) r = &kv.second;  // serde keeps the last duplicate
        return r;
    }
};

struct JParser {
    const char* p;
    const char* end;
    std::string err;
    int depth = 0;

    bool fail(const std::string& m) {
        if (err.empty()) err = m + " at offset " + std::to_string((size_t)(p - start));
        return false;
    }
    const char* start;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }

    static void put_utf8(std::string& s, uint32_t cp) {
        if (cp < 0x80) s.push_back((char)cp);
        else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t& v) {
        if (end - p < 4) return fail("EOF in \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("invalid \\u escape");
        }
        return true;
    }
    bool str(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        out.clear();
        for (;;) {
            if (p >= end) return fail("EOF while parsing a string");
            unsigned char c = (unsigned char)*p++;
            if (c == '"') break;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') { out.push_back((char)c); continue; }
            if (p >= end) return fail("EOF in escape");
            char e = *p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t v;
                    if (!hex4(v)) return false;
                    if (v >= 0xD800 && v < 0xDC00) {
                        if (end - p < 2 || p[0] != '\\' || p[1] != 'u') return fail("unexpected end of hex escape");
                        p += 2;
                        uint32_t lo;
                        if (!hex4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) return fail("lone leading surrogate in hex escape");
                        v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (v >= 0xDC00 && v <= 0xDFFF) {
                        return fail("lone trailing surrogate in hex escape");
                    }
                    put_utf8(out, v);
                    break;
                }
                default: return fail("invalid escape");
            }
        }
        if (!utf8_valid((const uint8_t*)out.data(), out.size())) return fail("invalid unicode code point");
        return true;
    }
    bool num(JVal& v) {
        const char* s = p;
        v.kind = JVal::Num;
        v.neg = false;
        if (p < end && *p == '-') { v.neg = true; ++p; }
        if (p >= end || *p < '0' || *p > '9') return fail("invalid number");
        if (*p == '0') ++p;
        else while (p < end && *p >= '0' && *p <= '9') ++p;
        bool is_int = true;
        if (p < end && *p == '.') {
            is_int = false; ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("invalid number");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            is_int = false; ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("invalid number");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        std::string t(s, (size_t)(p - s));
        v.d = strtod(t.c_str(), nullptr);
        v.is_int = false;
        if (is_int) {
            const char* q = t.c_str() + (v.neg ? 1 : 0);
            if (strlen(q) <= 19) { v.u = strtoull(q, nullptr, 10); v.is_int = true; }
        }
        return true;
    }
    bool value(JVal& v) {
        if (++depth > 128) return fail("recursion limit exceeded");
        ws();
        if (p >= end) return fail("EOF while parsing a value");
        bool ok = true;
        char c = *p;
        if (c == '{') {
            v.kind = JVal::Obj; ++p; ws();
            if (p < end && *p == '}') { ++p; }
            else for (;;) {
                ws();
                std::string k;
                if (!str(k)) { ok = false; break; }
                ws();
                if (p >= end || *p != ':') { ok = fail("expected `:`"); break; }
                ++p;
                v.o.emplace_back(std::move(k), JVal());
                if (!value(v.o.back().second)) { ok = false; break; }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = fail("expected `,` or `}`"); break;
            }
        } else if (c == '[') {
            v.kind = JVal::Arr; ++p; ws();
            if (p < end && *p == ']') { ++p; }
            else for (;;) {
                v.a.emplace_back();
                if (!value(v.a.back())) { ok = false; break; }
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = fail("expected `,` or `]`"); break;
            }
        } else if (c == '"') {
            v.kind = JVal::Str;
            ok = str(v.s);
        } else if (c == 't' && end - p >= 4 && !memcmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; }
        else if (c == 'f' && end - p >= 5 && !memcmp(p, "false", 5)) { v.kind = JVal::Bool; v.b = false; p += 5; }
        else if (c == 'n' && end - p >= 4 && !memcmp(p, "null", 4)) { v.kind = JVal::Null; p += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) ok = num(v);
        else ok = fail("expected value");
        --depth;
        return ok;
    }
};

bool get_usize(const JVal* v, uint64_t& out) {
    if (!v || v->kind != JVal::Num || !v->is_int || v->neg) return false;
    out = v->u;
    return true;
}
bool get_str(const JVal* v, std::string& out) {
    if (!v || v->kind != JVal::Str) return false;
    out = v->s;
    return true;
}
bool get_f64(const JVal* v) { return v && v->kind == JVal::Num; }

}  // namespace

TokenizerError parse_model_data(const char* json, size_t len, ModelData& out) {
    JParser P;
    P.p = P.start = json;
    P.end = json + len;
    JVal root;
    if (!P.value(root)) return mk(TK_ERR_JSON, P.err);
    P.ws();
    if (P.p != P.end) return mk(TK_ERR_JSON, "trailing characters");
    if (root.kind != JVal::Obj) return mk(TK_ERR_JSON, "invalid type: expected struct ModelData");

    const JVal* vocab = root.get("vocab");
    if (!vocab) return mk(TK_ERR_JSON, "missing field `vocab`");
    if (vocab->kind != JVal::Arr) return mk(TK_ERR_JSON, "invalid type for `vocab`: expected a sequence");
    out.vocab.clear();
    out.vocab.reserve(vocab->a.size());
    for (const JVal& e : vocab->a) {
        if (e.kind != JVal::Obj) return mk(TK_ERR_JSON, "invalid type: expected struct TokenInfo");
        TokenInfo ti;
        if (!e.get("rank")) return mk(TK_ERR_JSON, "missing field `rank`");
        if (!get_usize(e.get("rank"), ti.rank)) return mk(TK_ERR_JSON, "invalid type for `rank`: expected usize");
        if (!e.get("token_bytes")) return mk(TK_ERR_JSON, "missing field `token_bytes`");
        if (!get_str(e.get("token_bytes"), ti.token_bytes)) return mk(TK_ERR_JSON, "invalid type for `token_bytes`: expected a string");
        const JVal* ts = e.get("token_str");
        if (ts && ts->kind != JVal::Null) {
            if (!get_str(ts, ti.token_str)) return mk(TK_ERR_JSON, "invalid type for `token_str`: expected a string");
            ti.has_token_str = true;
        }
        out.vocab.push_back(std::move(ti));
    }

    const JVal* sp = root.get("special_tokens");
    out.has_special_tokens = false;
    out.special_tokens.clear();
    if (sp && sp->kind != JVal::Null) {
        if (sp->kind != JVal::Arr) return mk(TK_ERR_JSON, "invalid type for `special_tokens`: expected a sequence");
        out.has_special_tokens = true;
        for (const JVal& e : sp->a) {
            if (e.kind != JVal::Obj) return mk(TK_ERR_JSON, "invalid type: expected struct SpecialTokenInfo");
            SpecialTokenInfo si;
            if (!e.get("rank")) return mk(TK_ERR_JSON, "missing field `rank`");
            if (!get_usize(e.get("rank"), si.rank)) return mk(TK_ERR_JSON, "invalid type for `rank`: expected usize");
            if (!e.get("token_str")) return mk(TK_ERR_JSON, "missing field `token_str`");
            if (!get_str(e.get("token_str"), si.token_str)) return mk(TK_ERR_JSON, "invalid type for `token_str`: expected a string");
            const JVal* ic = e.get("is_control");
            if (!ic) return mk(TK_ERR_JSON, "missing field `is_control`");
            if (ic->kind != JVal::Bool) return mk(TK_ERR_JSON, "invalid type for `is_control`: expected a boolean");
            si.is_control = ic->b;
            out.special_tokens.push_back(std::move(si));
        }
    }

    const JVal* cfg = root.get("config");
    if (!cfg) return mk(TK_ERR_JSON, "missing field `config`");
    if (cfg->kind != JVal::Obj) return mk(TK_ERR_JSON, "invalid type: expected struct TekkenConfig");
    if (!cfg->get("pattern")) return mk(TK_ERR_JSON, "missing field `pattern`");
    if (!get_str(cfg->get("pattern"), out.config.pattern)) return mk(TK_ERR_JSON, "invalid type for `pattern`: expected a string");
    const char* nums[3] = {"num_vocab_tokens", "default_vocab_size", "default_num_special_tokens"};
    uint64_t* dst[3] = {&out.config.num_vocab_tokens, &out.config.default_vocab_size,
                        &out.config.default_num_special_tokens};
    for (int i = 0; i < 3; ++i) {
        if (!cfg->get(nums[i])) return mk(TK_ERR_JSON, std::string("missing field `") + nums[i] + "`");
        if (!get_usize(cfg->get(nums[i]), *dst[i])) return mk(TK_ERR_JSON, std::string("invalid type for `") + nums[i] + "`: expected usize");
    }
    if (!cfg->get("version")) return mk(TK_ERR_JSON, "missing field `version`");
    if (!get_str(cfg->get("version"), out.config.version)) return mk(TK_ERR_JSON, "invalid type for `version`: expected a string");

    const JVal* au = root.get("audio");
    out.has_audio = false;
    if (au && au->kind != JVal::Null) {
        // AudioConfig / AudioSpectrogramConfig (reference src/audio.rs:17-22, 85-91): shape only
        if (au->kind != JVal::Obj) return mk(TK_ERR_JSON, "invalid type: expected struct AudioConfig");
        uint64_t tmp;
        if (!get_usize(au->get("sampling_rate"), tmp)) return mk(TK_ERR_JSON, "missing or invalid field `sampling_rate`");
        if (!get_f64(au->get("frame_rate"))) return mk(TK_ERR_JSON, "missing or invalid field `frame_rate`");
        const JVal* ec = au->get("audio_encoding_config");
        if (!ec || ec->kind != JVal::Obj) return mk(TK_ERR_JSON, "missing or invalid field `audio_encoding_config`");
        if (!get_usize(ec->get("num_mel_bins"), tmp) || !get_usize(ec->get("hop_length"), tmp) ||
            !get_usize(ec->get("window_size"), tmp))
            return mk(TK_ERR_JSON, "missing or invalid field in `audio_encoding_config`");
        const JVal* cl = au->get("chunk_length_s");
        if (cl && cl->kind != JVal::Null && !get_f64(cl)) return mk(TK_ERR_JSON, "invalid type for `chunk_length_s`");
        out.has_audio = true;
    }
    return TokenizerError();
}

// ------------------------------------------------------------------------------------------
// the 20 legacy special tokens (reference src/tekkenizer.rs:827-930)
// ------------------------------------------------------------------------------------------
static std::vector<SpecialTokenInfo> deprecated_special_tokens() {
    static const char* names[20] = {"<unk>", "<s>", "</s>", "[INST]", "[/INST]", "[AVAILABLE_TOOLS]",
                                    "[/AVAILABLE_TOOLS]", "[TOOL_RESULTS]", "[/TOOL_RESULTS]", "[TOOL_CALLS]",
                                    "[IMG]", "<pad>", "[IMG_BREAK]", "[IMG_END]", "[PREFIX]", "[MIDDLE]", "[SUFFIX]",
                                    "[SYSTEM_PROMPT]", "[/SYSTEM_PROMPT]", "[TOOL_CONTENT]"};
    std::vector<SpecialTokenInfo> v;
    for (uint64_t i = 0; i < 20; ++i) v.push_back(SpecialTokenInfo{i, names[i], true});
    return v;
}

Tekkenizer::~Tekkenizer() {
    if (ctx_) tk_ctx_destroy(ctx_);
}

// Tekkenizer::new, reference src/tekkenizer.rs:71-191 (checks in the same order)
Tekkenizer* Tekkenizer::create(const std::vector<TokenInfo>& vocab_in, const std::vector<SpecialTokenInfo>& special_tokens,
                               const std::string& /*pattern: ignored, src/tekkenizer.rs:74*/, uint64_t vocab_size,
                               uint64_t num_special_tokens, const std::string& version, bool has_audio, int device_id,
                               TokenizerError& err) {
    err = TokenizerError();
    if (vocab_size > vocab_in.size() + num_special_tokens) {  // :80-87
        err = mk(TK_ERR_INVALID_CONFIG, "vocab_size (" + std::to_string(vocab_size) + ") must be <= vocab.len() (" +
                                            std::to_string(vocab_in.size()) + ") + num_special_tokens (" +
                                            std::to_string(num_special_tokens) + ")");
        return nullptr;
    }
    std::unordered_set<std::string> seen;  // :90-98
    for (const auto& t : special_tokens) {
        if (!seen.insert(t.token_str).second) {
            err = mk(TK_ERR_INVALID_CONFIG, "Duplicate special token: " + t.token_str);
            return nullptr;
        }
    }
    if (special_tokens.size() > num_special_tokens) {  // :100-106
        err = mk(TK_ERR_INVALID_CONFIG, "special_tokens.len() (" + std::to_string(special_tokens.size()) +
                                            ") must be <= num_special_tokens (" + std::to_string(num_special_tokens) + ")");
        return nullptr;
    }
    std::vector<SpecialTokenInfo> all = special_tokens;  // :108-116
    for (uint64_t i = special_tokens.size(); i < num_special_tokens; ++i)
        all.push_back(SpecialTokenInfo{i, "<SPECIAL_" + std::to_string(i) + ">", true});

    if (vocab_size < num_special_tokens) {
        // the reference computes `vocab_size - num_special_tokens` on usize (:118): a debug build
        // panics, a release build wraps; both are configuration errors
        err = mk(TK_ERR_INVALID_CONFIG, "vocab_size must be >= num_special_tokens");
        return nullptr;
    }
    const uint64_t inner = vocab_size - num_special_tokens;

    // reload_mergeable_ranks, :776-816
    const size_t n_take = vocab_in.size() > inner ? (size_t)inner : vocab_in.size();
    std::unordered_map<std::string, uint64_t> ranks;
    ranks.reserve(n_take * 2);
    std::string bytes, b64err;
    for (size_t i = 0; i < n_take; ++i) {
        const TokenInfo& t = vocab_in[i];
        if (!base64_decode_standard(t.token_bytes, bytes, b64err)) {
            err = mk(TK_ERR_BASE64, b64err);
            return nullptr;
        }
        if (t.rank < 256 && !(bytes.size() == 1 && (uint8_t)bytes[0] == (uint8_t)t.rank)) {  // :793-798
            err = mk(TK_ERR_INVALID_CONFIG, "Expected byte token at rank " + std::to_string(t.rank) + " to be [" +
                                                std::to_string(t.rank) + "]");
            return nullptr;
        }
        ranks[bytes] = t.rank;  // later duplicates replace earlier ones, like HashMap::insert (:801)
    }
    {  // contiguity, :804-813
        std::vector<uint8_t> hit(ranks.size(), 0);
        bool ok = true;
        for (const auto& kv : ranks) {
            if (kv.second >= ranks.size() || hit[kv.second]) { ok = false; break; }
            hit[kv.second] = 1;
        }
        if (!ok) {
            err = mk(TK_ERR_INVALID_CONFIG, "Vocabulary ranks are not contiguous");
            return nullptr;
        }
    }
    // rank-indexed table
    const size_t n_ranks = ranks.size();
    std::vector<const std::string*> by_rank(n_ranks, nullptr);
    size_t total = 0;
    for (const auto& kv : ranks) { by_rank[kv.second] = &kv.first; total += kv.first.size(); }
    std::vector<uint32_t> offs(n_ranks + 1, 0);
    std::vector<uint8_t> blob;
    blob.reserve(total + 1);
    for (size_t r = 0; r < n_ranks; ++r) {
        blob.insert(blob.end(), by_rank[r]->begin(), by_rank[r]->end());
        offs[r + 1] = (uint32_t)blob.size();
    }
    return assemble(std::move(all), std::move(blob), std::move(offs), vocab_size, num_special_tokens, version, has_audio, device_id, err);
}

Tekkenizer* Tekkenizer::assemble(std::vector<SpecialTokenInfo>&& all, std::vector<uint8_t>&& blob, std::vector<uint32_t>&& offs,
                                 uint64_t vocab_size, uint64_t num_special_tokens, const std::string& version, bool has_audio,
                                 int device_id, TokenizerError& err) {
    Tekkenizer* t = new Tekkenizer();
    t->vocab_size_ = vocab_size;
    t->num_special_tokens_ = num_special_tokens;
    t->version_ = version;
    t->special_tokens_ = std::move(all);
    t->has_audio_ = has_audio;
    t->blob_ = std::move(blob);
    t->offs_ = std::move(offs);
    const std::vector<SpecialTokenInfo>& all_ref = t->special_tokens_;
    for (const auto& s : all_ref) t->special_tokens_map_[s.token_str] = s.rank;  // :129-132 (later wins)
    const size_t n_ranks = t->offs_.size() - 1;
    // vocabulary strings, :135-155
    t->vocab_.resize((size_t)vocab_size);
    for (uint64_t i = 0; i < vocab_size; ++i) {
        if (i < num_special_tokens) t->vocab_[i] = all_ref[i].token_str;
        else {
            const uint64_t r = i - num_special_tokens;
            if (r < n_ranks) t->vocab_[i] = utf8_lossy(t->blob_.data() + t->offs_[r], t->offs_[r + 1] - t->offs_[r]);
            else t->vocab_[i] = "<?>";
        }
    }
    if (has_audio) {  // :158-178
        if (!t->special_tokens_map_.count("[AUDIO]")) {
            err = mk(TK_ERR_TOKEN_NOT_FOUND, "Audio token not found");
            delete t;
            return nullptr;
        }
        if (!t->special_tokens_map_.count("[BEGIN_AUDIO]")) {
            err = mk(TK_ERR_TOKEN_NOT_FOUND, "BeginAudio token not found");
            delete t;
            return nullptr;
        }
    }
    if (device_id >= 0) {
        uint32_t bos = 0, eos = 0;
        (void)t->bos_id(bos);
        (void)t->eos_id(eos);
        if (t->blob_.empty()) t->blob_.push_back(0);
        int rc = tk_ctx_create(t->blob_.data(), t->offs_.data(), (uint32_t)n_ranks, (uint32_t)num_special_tokens, bos,
                               eos, device_id, &t->ctx_);
        if (rc != TK_OK) {
            err = mk(rc, std::string("Failed to create CoreBPE: ") + tk_last_error(nullptr));
            delete t;
            return nullptr;
        }
        // special-token strings BY POSITION, for the device decode path (Keep policy, :536-540)
        std::string sblob;
        std::vector<uint32_t> soffs(1, 0);
        for (const auto& s : all_ref) { sblob += s.token_str; soffs.push_back((uint32_t)sblob.size()); }
        rc = tk_ctx_set_special_tokens(t->ctx_, (const uint8_t*)sblob.data(), soffs.data(), (uint32_t)all_ref.size());
        if (rc != TK_OK) {
            err = mk(rc, std::string("special tokens upload failed: ") + tk_last_error(t->ctx_));
            delete t;
            return nullptr;
        }
    }
    return t;
}

Tekkenizer* Tekkenizer::from_json(const char* json, size_t len, int device_id, TokenizerError& err) {
    ModelData md;
    err = parse_model_data(json, len, md);
    if (!err.ok()) return nullptr;
    const std::string& v = md.config.version;  // :226-232
    if (!(v == "v3" || v == "v7" || v == "v11" || v == "v13")) {
        err = mk(TK_ERR_INVALID_CONFIG, "Unknown version: " + v);
        return nullptr;
    }
    const std::vector<SpecialTokenInfo> sp = md.has_special_tokens ? md.special_tokens : deprecated_special_tokens();
    Tekkenizer* t = create(md.vocab, sp, md.config.pattern, md.config.default_vocab_size, md.config.default_num_special_tokens, v,
                           md.has_audio, device_id, err);
    if (t) t->pattern_ = md.config.pattern;
    return t;
}

// ---- model cache ----
// With TK_TABLE_CACHE_DIR set, from_file keeps what it derived from a tekken.json -- the validated rank table, the
// special tokens and the scalars -- in `<dir>/tk_model_<hash of the file's bytes>.bin`; the next from_file of a file
// with the same bytes skips the JSON parse, base64 and the rank-map checks (and, through tk_build_tables_cached, the
// device-table build).  Only a load that SUCCEEDED is ever written, so error behaviour is that of the uncached path; the
// key is 128 bits over length + content, the file carries magic, version, key, sizes and a tail marker, and anything
// that does not check out falls back to the JSON.
#define TK_MODEL_MAGIC 0x4D4B5454u /* "TTKM" */
#define TK_MODEL_VERSION 3u   /* 2: config.pattern is kept (set_honour_pattern after a cached load); 3: payload checksum */

static void content_key(const std::string& c, uint64_t key[2]) {
    uint64_t a = 0xCBF29CE484222325ull ^ c.size(), b = 0x9E3779B97F4A7C15ull + c.size();
    const size_t n8 = c.size() / 8;
    const char* p = c.data();
    for (size_t i = 0; i < n8; ++i) {
        uint64_t w;
        memcpy(&w, p + 8 * i, 8);
        a = (a ^ w) * 0x100000001B3ull;
        a ^= a >> 29;
        b = (b + w) * 0xD6E8FEB86659FD93ull;
        b ^= b >> 32;
    }
    uint64_t w = 0;
    memcpy(&w, p + 8 * n8, c.size() - 8 * n8);
    a = (a ^ w) * 0x100000001B3ull;
    b = (b + w) * 0xD6E8FEB86659FD93ull;
    key[0] = a ^ (a >> 31);
    key[1] = b ^ (b >> 33);
}

namespace {
struct ModelHead {
    uint32_t magic, version;
    uint64_t key[2];
    uint64_t vocab_size, num_special_tokens, n_special, n_ranks, blob_bytes, strings_bytes;
    uint32_t has_audio, version_len;
    uint32_t pattern_len, pad;
};
}  // namespace

bool Tekkenizer::save_model_cache(const std::string& path, const uint64_t key[2]) const {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    std::string strings;                     // version, then per special token: u64 rank, u8 is_control, u32 len, bytes
    strings += version_;
    for (const auto& s : special_tokens_) {
        const uint64_t r = s.rank;
        const uint8_t ic = s.is_control ? 1 : 0;
        const uint32_t l = (uint32_t)s.token_str.size();
        strings.append((const char*)&r, 8);
        strings.append((const char*)&ic, 1);
        strings.append((const char*)&l, 4);
        strings += s.token_str;
    }
    strings += pattern_;                     // config.pattern, last (the reference ignores it; the opt-in of row f-3 needs it)
    ModelHead h;
    memset(&h, 0, sizeof(h));
    h.magic = TK_MODEL_MAGIC; h.version = TK_MODEL_VERSION; h.key[0] = key[0]; h.key[1] = key[1];
    h.vocab_size = vocab_size_; h.num_special_tokens = num_special_tokens_; h.n_special = special_tokens_.size();
    h.n_ranks = offs_.size() - 1; h.blob_bytes = offs_.back(); h.strings_bytes = strings.size();
    h.has_audio = has_audio_ ? 1 : 0; h.version_len = (uint32_t)version_.size();
    h.pattern_len = (uint32_t)pattern_.size();
    const uint32_t tail = TK_MODEL_MAGIC;
    // the payload checksum goes in front of the tail marker: a flipped bit anywhere makes the file "not check out"
    uint64_t sum = tk_sum64(key[0] ^ key[1], &h, sizeof(h));
    sum = tk_sum64(sum, offs_.data(), 4 * offs_.size());
    sum = tk_sum64(sum, blob_.data(), h.blob_bytes);
    sum = tk_sum64(sum, strings.data(), strings.size());
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(offs_.data(), 4, offs_.size(), f) == offs_.size() &&
              (h.blob_bytes == 0 || fwrite(blob_.data(), 1, h.blob_bytes, f) == h.blob_bytes) &&
              (strings.empty() || fwrite(strings.data(), 1, strings.size(), f) == strings.size()) && fwrite(&sum, 8, 1, f) == 1 &&
              fwrite(&tail, 4, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) remove(tmp.c_str());
    return ok;
}

Tekkenizer* Tekkenizer::load_model_cache(const std::string& path, const uint64_t key[2], int device_id, TokenizerError& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    ModelHead h;
    std::vector<uint32_t> offs;
    std::vector<uint8_t> blob;
    std::string strings;
    uint32_t tail = 0;
    uint64_t want = 0;
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && h.magic == TK_MODEL_MAGIC && h.version == TK_MODEL_VERSION && h.key[0] == key[0] &&
              h.key[1] == key[1] && h.n_ranks < (1ull << 31) && h.blob_bytes < (1ull << 32) && h.strings_bytes < (1ull << 31) &&
              h.n_special <= h.num_special_tokens && h.n_special < (1ull << 24) && h.version_len <= h.strings_bytes &&
              (uint64_t)h.version_len + h.pattern_len <= h.strings_bytes;
    if (ok) {
        offs.resize(h.n_ranks + 1);
        blob.resize(h.blob_bytes);
        strings.resize(h.strings_bytes);
        ok = fread(offs.data(), 4, offs.size(), f) == offs.size() && (blob.empty() || fread(blob.data(), 1, blob.size(), f) == blob.size()) &&
             (strings.empty() || fread(&strings[0], 1, strings.size(), f) == strings.size()) && fread(&want, 8, 1, f) == 1 &&
             fread(&tail, 4, 1, f) == 1 && tail == TK_MODEL_MAGIC && offs[0] == 0 && offs.back() == h.blob_bytes;
        if (ok) {
            uint64_t sum = tk_sum64(key[0] ^ key[1], &h, sizeof(h));
            sum = tk_sum64(sum, offs.data(), 4 * offs.size());
            sum = tk_sum64(sum, blob.data(), blob.size());
            sum = tk_sum64(sum, strings.data(), strings.size());
            ok = sum == want;
        }
    }
    fclose(f);
    for (size_t r = 0; ok && r < h.n_ranks; ++r) ok = offs[r + 1] >= offs[r];
    if (!ok) return nullptr;
    std::vector<SpecialTokenInfo> all;
    size_t q = h.version_len;
    const std::string version = strings.substr(0, q);
    for (uint64_t i = 0; i < h.n_special; ++i) {
        if (q + 13 > strings.size()) return nullptr;
        uint64_t r; uint8_t ic; uint32_t l;
        memcpy(&r, &strings[q], 8); ic = (uint8_t)strings[q + 8]; memcpy(&l, &strings[q + 9], 4);
        q += 13;
        if (q + l > strings.size()) return nullptr;
        all.push_back(SpecialTokenInfo{r, strings.substr(q, l), ic != 0});
        q += l;
    }
    if (q + h.pattern_len != strings.size() || all.size() != h.num_special_tokens) return nullptr;
    const std::string pattern = strings.substr(q, h.pattern_len);
    Tekkenizer* t = assemble(std::move(all), std::move(blob), std::move(offs), h.vocab_size, h.num_special_tokens, version, h.has_audio != 0,
                             device_id, err);
    if (t) { t->pattern_ = pattern; t->from_cache_ = true; }
    return t;
}

Tekkenizer* Tekkenizer::from_file(const std::string& path, int device_id, TokenizerError& err) {
    std::string content;
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) {
            err = mk(TK_ERR_IO, "No such file or directory (os error 2): " + path);
            return nullptr;
        }
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) content.append(buf, got);
        fclose(f);
    }
    const char* dir = getenv("TK_TABLE_CACHE_DIR");
    uint64_t key[2] = {0, 0};
    std::string side;
    if (dir && *dir) {
        content_key(content, key);
        char name[80];
        snprintf(name, sizeof(name), "/tk_model_%016llx%016llx.bin", (unsigned long long)key[0], (unsigned long long)key[1]);
        side = std::string(dir) + name;
        err = TokenizerError();
        Tekkenizer* t = load_model_cache(side, key, device_id, err);
        if (t) return t;
        if (!err.ok()) return nullptr;       // the cache was good, the device side failed: same error as the uncached path
    }
    if (!utf8_valid((const uint8_t*)content.data(), content.size())) {  // read_to_string (:223)
        err = mk(TK_ERR_IO, "stream did not contain valid UTF-8");
        return nullptr;
    }
    Tekkenizer* t = from_json(content.data(), content.size(), device_id, err);
    if (t && !side.empty()) (void)t->save_model_cache(side, key);   // best effort
    return t;
}

// the `pattern` of Mistral's tekken.json (literal in reference tests/test_small_vocab.rs:62)
static const char* kTekkenJsonPattern =
    "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+|"
    "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*|"
    "\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";

TokenizerError Tekkenizer::set_honour_pattern(bool honour) {
    if (!ctx_) return mk(TK_ERR_NO_DEVICE, "no device context");
    if (honour && pattern_ != kTekkenJsonPattern)
        return mk(TK_ERR_INVALID_CONFIG, "unsupported pattern: only the pattern of Mistral's tekken.json can be honoured");
    const int rc = tk_ctx_set_pattern(ctx_, honour ? 1 : 0);
    if (rc != TK_OK) return mk(rc, tk_last_error(ctx_));
    return TokenizerError();
}

TokenizerError Tekkenizer::get_control_token(const std::string& name, uint32_t& id) const {  // :331-341
    auto it = special_tokens_map_.find(name);
    if (it == special_tokens_map_.end()) return mk(TK_ERR_TOKEN_NOT_FOUND, "Unknown control token: '" + name + "'");
    id = (uint32_t)it->second;
    return TokenizerError();
}

TokenizerError Tekkenizer::encode_batch(const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, bool add_bos,
                                        bool add_eos, tk_result* out) {
    uint32_t id;
    if (add_bos) { TokenizerError e = bos_id(id); if (!e.ok()) return e; }  // :394-397
    if (add_eos) { TokenizerError e = eos_id(id); if (!e.ok()) return e; }  // :399-402
    if (!ctx_) return mk(TK_ERR_NO_DEVICE, "tokenizer was created without a device (host-only object)");
    int rc = tk_encode_batch(ctx_, bytes, doc_offsets, n_docs, add_bos, add_eos, 0, out);
    if (rc != TK_OK) return mk(rc, tk_last_error(ctx_));
    return TokenizerError();
}

TokenizerError Tekkenizer::encode(const char* text, size_t len, bool add_bos, bool add_eos, std::vector<uint32_t>& out) {
    uint32_t id;
    if (add_bos) { TokenizerError e = bos_id(id); if (!e.ok()) return e; }  // :394-397
    if (add_eos) { TokenizerError e = eos_id(id); if (!e.ok()) return e; }  // :399-402
    if (!ctx_) return mk(TK_ERR_NO_DEVICE, "tokenizer was created without a device (host-only object)");
    // one document, caller-owned output: no allocation, one launch for texts of up to 64 KiB (tk_encode_one)
    out.resize(len + 2);
    uint64_t n = 0;
    const int rc = tk_encode_one(ctx_, (const uint8_t*)text, len, add_bos, add_eos, out.data(), out.size(), &n);
    if (rc != TK_OK) { out.clear(); return mk(rc, tk_last_error(ctx_)); }
    out.resize(n);
    return TokenizerError();
}

// CoreBPE::decode: concatenate token bytes, then String::from_utf8 (call sites :552-555, :680)
TokenizerError Tekkenizer::core_decode(const uint32_t* ranks, size_t n, std::string& out) const {
    out.clear();
    const size_t n_ranks = offs_.size() - 1;
    for (size_t i = 0; i < n; ++i) {
        if (ranks[i] >= n_ranks) return mk(TK_ERR_RUNTIME, "DecodeKeyError: Invalid token for decoding: " + std::to_string(ranks[i]));
        out.append((const char*)blob_.data() + offs_[ranks[i]], offs_[ranks[i] + 1] - offs_[ranks[i]]);
    }
    if (!utf8_valid((const uint8_t*)out.data(), out.size())) return mk(TK_ERR_RUNTIME, "FromUtf8Error: invalid utf-8 sequence");
    return TokenizerError();
}

TokenizerError Tekkenizer::decode_group(const uint32_t* ids, size_t n, bool is_special, SpecialTokenPolicy policy,
                                        std::vector<std::string>& out) const {  // :522-560
    if (is_special) {
        if (policy == SpecialTokenPolicy::Raise) {
            std::string l = "[";
            for (size_t i = 0; i < n; ++i) l += (i ? ", " : "") + std::to_string(ids[i]);
            return mk(TK_ERR_SPECIAL_POLICY, "Decoding tokens that contain special tokens (" + l + "]) is not allowed");
        }
        if (policy == SpecialTokenPolicy::Keep)
            for (size_t i = 0; i < n; ++i) out.push_back(special_tokens_[ids[i]].token_str);
        return TokenizerError();
    }
    std::vector<uint32_t> shifted(n);
    for (size_t i = 0; i < n; ++i) shifted[i] = ids[i] - (uint32_t)num_special_tokens_;
    std::string text;
    TokenizerError e = core_decode(shifted.data(), n, text);
    if (!e.ok()) return e;
    out.push_back(std::move(text));
    return TokenizerError();
}

TokenizerError Tekkenizer::decode_all(const uint32_t* ids, size_t n, SpecialTokenPolicy policy,
                                      std::vector<std::string>& out) const {  // :463-511
    out.clear();
    size_t g0 = 0;
    while (g0 < n) {
        const bool sp = ids[g0] < num_special_tokens_;
        size_t g1 = g0 + 1;
        while (g1 < n && (ids[g1] < num_special_tokens_) == sp) ++g1;
        TokenizerError e = decode_group(ids + g0, g1 - g0, sp, policy, out);
        if (!e.ok()) return e;
        g0 = g1;
    }
    return TokenizerError();
}

TokenizerError Tekkenizer::decode(const uint32_t* ids, size_t n, SpecialTokenPolicy policy, std::string& out) const {
    std::vector<std::string> parts;
    TokenizerError e = decode_all(ids, n, policy, parts);
    if (!e.ok()) return e;
    out.clear();
    for (auto& p : parts) out += p;  // :442
    return TokenizerError();
}

TokenizerError Tekkenizer::id_to_piece(uint32_t id, std::string& out) const {  // :617-628
    if (id >= vocab_size_)
        return mk(TK_ERR_INVALID_CONFIG, "Token ID " + std::to_string(id) + " is out of vocabulary range (0-" +
                                             std::to_string(vocab_size_ - 1) + ")");
    return decode(&id, 1, SpecialTokenPolicy::Keep, out);
}

TokenizerError Tekkenizer::id_to_byte_piece(uint32_t id, SpecialTokenPolicy policy, std::string& out) const {  // :648-695
    if (id >= vocab_size_)
        return mk(TK_ERR_INVALID_CONFIG, "Token ID " + std::to_string(id) + " is out of vocabulary range (0-" +
                                             std::to_string(vocab_size_ - 1) + ")");
    out.clear();
    if (id < num_special_tokens_) {
        if (policy == SpecialTokenPolicy::Keep) { out = special_tokens_[id].token_str; return TokenizerError(); }
        if (policy == SpecialTokenPolicy::Raise)
            return mk(TK_ERR_SPECIAL_POLICY, "Token ID " + std::to_string(id) + " is a special token (" +
                                                 special_tokens_[id].token_str +
                                                 "), cannot convert to byte piece with Raise policy");
        return TokenizerError();
    }
    const uint32_t shifted = id - (uint32_t)num_special_tokens_;
    TokenizerError e = core_decode(&shifted, 1, out);
    if (e.ok()) return e;
    out = vocab_[id];  // the reference falls back to the (lossy) vocabulary string, :683-686
    return TokenizerError();
}

}  // namespace tekken

// ------------------------------------------------------------------------------------------
// tokenizer-level C ABI
// ------------------------------------------------------------------------------------------
struct tk_tokenizer {
    tekken::Tekkenizer* t = nullptr;
    std::string err;
};

static int finish(tk_tokenizer* h, const tekken::TokenizerError& e) {
    if (!e.ok()) h->err = e.message;
    return e.code;
}

static int wrap_new(tekken::Tekkenizer* t, const tekken::TokenizerError& e, tk_tokenizer** out) {
    if (!t) { tk_set_tls_error(e.message); return e.code ? e.code : TK_ERR_RUNTIME; }
    tk_tokenizer* h = new tk_tokenizer();
    h->t = t;
    *out = h;
    return TK_OK;
}

extern "C" int tk_tokenizer_from_file(const char* path, int device_id, tk_tokenizer** out) {
    if (!path || !out) { tk_set_tls_error("null argument"); return TK_ERR_INVALID_ARG; }
    *out = nullptr;
    tekken::TokenizerError e;
    return wrap_new(tekken::Tekkenizer::from_file(path, device_id, e), e, out);
}

extern "C" int tk_tokenizer_from_json(const char* json, size_t json_len, int device_id, tk_tokenizer** out) {
    if (!json || !out) { tk_set_tls_error("null argument"); return TK_ERR_INVALID_ARG; }
    *out = nullptr;
    tekken::TokenizerError e;
    return wrap_new(tekken::Tekkenizer::from_json(json, json_len, device_id, e), e, out);
}

extern "C" void tk_tokenizer_destroy(tk_tokenizer* h) {
    if (!h) return;
    delete h->t;
    delete h;
}

extern "C" const char* tk_tokenizer_last_error(const tk_tokenizer* h) {
    return h ? h->err.c_str() : tk_get_tls_error().c_str();
}

extern "C" int tk_tokenizer_encode(tk_tokenizer* h, const char* text, size_t len, int add_bos, int add_eos,
                                   uint32_t** ids, size_t* n_ids) {
    if (!h || !ids || !n_ids || (!text && len)) return TK_ERR_INVALID_ARG;
    std::vector<uint32_t> v;
    tekken::TokenizerError e = h->t->encode(text ? text : "", len, add_bos != 0, add_eos != 0, v);
    if (!e.ok()) return finish(h, e);
    *ids = (uint32_t*)malloc((v.size() ? v.size() : 1) * sizeof(uint32_t));
    if (!*ids) { h->err = "out of memory"; return TK_ERR_RUNTIME; }
    memcpy(*ids, v.data(), v.size() * sizeof(uint32_t));
    *n_ids = v.size();
    return TK_OK;
}

extern "C" int tk_tokenizer_encode_batch(tk_tokenizer* h, const uint8_t* bytes, const uint64_t* doc_offsets,
                                         uint64_t n_docs, int add_bos, int add_eos, tk_result* out) {
    if (!h || !doc_offsets || !out) return TK_ERR_INVALID_ARG;
    return finish(h, h->t->encode_batch(bytes, doc_offsets, n_docs, add_bos != 0, add_eos != 0, out));
}

extern "C" int tk_tokenizer_set_honour_pattern(tk_tokenizer* h, int honour) {
    if (!h) return TK_ERR_INVALID_ARG;
    return finish(h, h->t->set_honour_pattern(honour != 0));
}

extern "C" void tk_free_ids(uint32_t* ids) { free(ids); }

static int export_str(tk_tokenizer* h, const std::string& s, char** text, size_t* len) {
    *text = (char*)malloc(s.size() + 1);
    if (!*text) { h->err = "out of memory"; return TK_ERR_RUNTIME; }
    memcpy(*text, s.data(), s.size());
    (*text)[s.size()] = 0;
    *len = s.size();
    return TK_OK;
}

extern "C" int tk_tokenizer_decode(tk_tokenizer* h, const uint32_t* ids, size_t n_ids, int policy, char** text,
                                   size_t* len) {
    if (!h || !text || !len || (!ids && n_ids) || policy < 0 || policy > 2) return TK_ERR_INVALID_ARG;
    std::string s;
    tekken::TokenizerError e = h->t->decode(ids, n_ids, (tekken::SpecialTokenPolicy)policy, s);
    if (!e.ok()) return finish(h, e);
    return export_str(h, s, text, len);
}

extern "C" void tk_free_text(char* text) { free(text); }
extern "C" void tk_free_offsets(uint64_t* offsets) { free(offsets); }

static int export_strs(tk_tokenizer* h, const std::vector<std::string>& parts, char** text, uint64_t** ends, size_t* n) {
    std::string all;
    uint64_t* e = (uint64_t*)malloc(sizeof(uint64_t) * (parts.size() + 1));
    if (!e) { h->err = "out of memory"; return TK_ERR_RUNTIME; }
    for (size_t i = 0; i < parts.size(); ++i) {
        all += parts[i];
        e[i] = all.size();
    }
    size_t len = 0;
    int rc = export_str(h, all, text, &len);
    if (rc != TK_OK) { free(e); return rc; }
    *ends = e;
    *n = parts.size();
    return TK_OK;
}

extern "C" int tk_tokenizer_decode_all(tk_tokenizer* h, const uint32_t* ids, size_t n_ids, int policy, char** text,
                                       uint64_t** seg_ends, size_t* n_segments) {
    if (!h || !text || !seg_ends || !n_segments || (!ids && n_ids) || policy < 0 || policy > 2) return TK_ERR_INVALID_ARG;
    std::vector<std::string> parts;
    tekken::TokenizerError e = h->t->decode_all(ids, n_ids, (tekken::SpecialTokenPolicy)policy, parts);
    if (!e.ok()) return finish(h, e);
    return export_strs(h, parts, text, seg_ends, n_segments);
}

extern "C" int tk_tokenizer_vocab(tk_tokenizer* h, char** text, uint64_t** ends, size_t* n_pieces) {
    if (!h || !text || !ends || !n_pieces) return TK_ERR_INVALID_ARG;
    return export_strs(h, h->t->vocab(), text, ends, n_pieces);
}

extern "C" int tk_tokenizer_decode_batch(tk_tokenizer* h, const uint32_t* ids, const uint64_t* id_offsets, uint64_t n_docs,
                                         int policy, tk_text_result* out, uint64_t* bad_doc) {
    if (!h || !id_offsets || !out) return TK_ERR_INVALID_ARG;
    if (!h->t->ctx()) { h->err = "tokenizer was created without a device (host-only object)"; return TK_ERR_NO_DEVICE; }
    int rc = tk_decode_batch(h->t->ctx(), ids, id_offsets, n_docs, policy, out, bad_doc);
    if (rc != TK_OK) h->err = tk_last_error(h->t->ctx());
    return rc;
}

extern "C" uint32_t tk_tokenizer_vocab_size(const tk_tokenizer* h) { return h ? h->t->vocab_size() : 0; }
extern "C" uint32_t tk_tokenizer_num_special_tokens(const tk_tokenizer* h) { return h ? h->t->num_special_tokens() : 0; }
extern "C" const char* tk_tokenizer_version(const tk_tokenizer* h) { return h ? h->t->version().c_str() : ""; }

extern "C" int tk_tokenizer_control_token(tk_tokenizer* h, const char* name, uint32_t* id) {
    if (!h || !name || !id) return TK_ERR_INVALID_ARG;
    return finish(h, h->t->get_control_token(name, *id));
}

extern "C" int tk_tokenizer_is_special(const tk_tokenizer* h, uint32_t id) { return h && h->t->is_special_token(id); }
extern "C" int tk_tokenizer_is_byte(const tk_tokenizer* h, uint32_t id) { return h && h->t->is_byte(id); }

extern "C" int tk_tokenizer_id_to_piece(tk_tokenizer* h, uint32_t id, char** text, size_t* len) {
    if (!h || !text || !len) return TK_ERR_INVALID_ARG;
    std::string s;
    tekken::TokenizerError e = h->t->id_to_piece(id, s);
    if (!e.ok()) return finish(h, e);
    return export_str(h, s, text, len);
}

extern "C" int tk_tokenizer_id_to_byte_piece(tk_tokenizer* h, uint32_t id, int policy, uint8_t** bytes, size_t* len) {
    if (!h || !bytes || !len || policy < 0 || policy > 2) return TK_ERR_INVALID_ARG;
    std::string s;
    tekken::TokenizerError e = h->t->id_to_byte_piece(id, (tekken::SpecialTokenPolicy)policy, s);
    if (!e.ok()) return finish(h, e);
    return export_str(h, s, (char**)bytes, len);
}

extern "C" tk_ctx* tk_tokenizer_ctx(tk_tokenizer* h) { return h ? h->t->ctx() : nullptr; }
extern "C" const char* tk_tokenizer_json_pattern(const tk_tokenizer* h) { return h ? h->t->json_pattern().c_str() : ""; }
extern "C" int tk_tokenizer_from_cache(const tk_tokenizer* h) { return h && h->t->from_cache(); }

extern "C" int tk_tokenizer_rank_table(const tk_tokenizer* h, const uint8_t** blob, const uint32_t** offsets,
                                       uint32_t* n_ranks) {
    if (!h || !blob || !offsets || !n_ranks) return TK_ERR_INVALID_ARG;
    *blob = h->t->rank_blob().data();
    *offsets = h->t->rank_offsets().data();
    *n_ranks = (uint32_t)(h->t->rank_offsets().size() - 1);
    return TK_OK;
}
