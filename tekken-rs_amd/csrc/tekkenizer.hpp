// tekkenizer.hpp -- host-side mirror of tekken::tekkenizer::Tekkenizer (reference
// src/tekkenizer.rs:26-760): loader, construction checks, encode/decode surface and the
// SpecialTokenPolicy behaviour, with the text-encode hot path delegated to the gfx950 engine
// (tk_ctx).  The reference is Rust; no Rust toolchain exists in the build image, so the host
// side above the C ABI is written in C++ and mirrors the reference's names, argument meaning
// and error classes (TokenizerError, reference src/errors.rs:23-59).
#ifndef TK_TEKKENIZER_HPP
#define TK_TEKKENIZER_HPP
#include <stdint.h>

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tekken_hip.h"

namespace tekken {

// reference src/errors.rs:23-59; `code` is the TK_ERR_* the C ABI reports
struct TokenizerError {
    int code = TK_OK;
    std::string message;
    bool ok() const { return code == TK_OK; }
};

// reference src/special_tokens.rs:128-136
enum class SpecialTokenPolicy { Ignore = TK_POLICY_IGNORE, Keep = TK_POLICY_KEEP, Raise = TK_POLICY_RAISE };

// reference src/config.rs:16-23
struct TokenInfo {
    uint64_t rank = 0;
    std::string token_bytes;  // base64
    bool has_token_str = false;
    std::string token_str;
};

// reference src/special_tokens.rs:160-168
struct SpecialTokenInfo {
    uint64_t rank = 0;
    std::string token_str;
    bool is_control = true;
};

// reference src/config.rs:38-49
struct TekkenConfig {
    std::string pattern;  // parsed and IGNORED, like the reference (src/tekkenizer.rs:74,123)
    uint64_t num_vocab_tokens = 0, default_vocab_size = 0, default_num_special_tokens = 0;
    std::string version;
};

// reference src/config.rs:73-82 (audio is only checked for presence / shape: out of scope)
struct ModelData {
    std::vector<TokenInfo> vocab;
    bool has_special_tokens = false;
    std::vector<SpecialTokenInfo> special_tokens;
    TekkenConfig config;
    bool has_audio = false;
};

class Tekkenizer {
public:
    ~Tekkenizer();
    // Tekkenizer::from_file (src/tekkenizer.rs:222-248); device_id < 0 => host-only object
    static Tekkenizer* from_file(const std::string& path, int device_id, TokenizerError& err);
    static Tekkenizer* from_json(const char* json, size_t len, int device_id, TokenizerError& err);
    // Tekkenizer::new (src/tekkenizer.rs:71-191)
    static Tekkenizer* create(const std::vector<TokenInfo>& vocab, const std::vector<SpecialTokenInfo>& special_tokens,
                              const std::string& pattern, uint64_t vocab_size, uint64_t num_special_tokens,
                              const std::string& version, bool has_audio, int device_id, TokenizerError& err);

    // src/tekkenizer.rs:378-405
    TokenizerError encode(const char* text, size_t len, bool add_bos, bool add_eos, std::vector<uint32_t>& out);
    TokenizerError encode_batch(const uint8_t* bytes, const uint64_t* doc_offsets, uint64_t n_docs, bool add_bos,
                                bool add_eos, tk_result* out);
    // src/tekkenizer.rs:436-560
    TokenizerError decode(const uint32_t* ids, size_t n, SpecialTokenPolicy policy, std::string& out) const;
    TokenizerError decode_all(const uint32_t* ids, size_t n, SpecialTokenPolicy policy,
                              std::vector<std::string>& out) const;
    // src/tekkenizer.rs:617-695
    TokenizerError id_to_piece(uint32_t id, std::string& out) const;
    TokenizerError id_to_byte_piece(uint32_t id, SpecialTokenPolicy policy, std::string& out) const;
    // src/tekkenizer.rs:574-600
    bool is_special_token(uint32_t id) const { return id < num_special_tokens_; }
    bool is_byte(uint32_t id) const { return id >= num_special_tokens_ && id - num_special_tokens_ < 256; }
    // src/tekkenizer.rs:260-350
    uint32_t vocab_size() const { return (uint32_t)vocab_size_; }
    uint32_t num_special_tokens() const { return (uint32_t)num_special_tokens_; }
    const std::string& version() const { return version_; }
    TokenizerError get_control_token(const std::string& name, uint32_t& id) const;
    TokenizerError bos_id(uint32_t& id) const { return get_control_token("<s>", id); }
    TokenizerError eos_id(uint32_t& id) const { return get_control_token("</s>", id); }
    TokenizerError pad_id(uint32_t& id) const { return get_control_token("<pad>", id); }
    TokenizerError unk_id(uint32_t& id) const { return get_control_token("<unk>", id); }
    const std::vector<std::string>& vocab() const { return vocab_; }

    tk_ctx* ctx() { return ctx_; }
    // row f-3 (opt-in): honour the JSON `pattern` the file carried.  Only the pattern of Mistral's tekken.json is known
    // to the matcher; any other string is refused.  honour = false restores the reference's behaviour (pattern ignored).
    TokenizerError set_honour_pattern(bool honour);
    const std::string& json_pattern() const { return pattern_; }
    bool from_cache() const { return from_cache_; }              // loaded from a TK_TABLE_CACHE_DIR side file (row f-2)
    const std::vector<uint8_t>& rank_blob() const { return blob_; }
    const std::vector<uint32_t>& rank_offsets() const { return offs_; }
    std::string last_error;

private:
    Tekkenizer() = default;
    // second half of `create`: everything after the rank table is validated (also the entry of the model cache)
    static Tekkenizer* assemble(std::vector<SpecialTokenInfo>&& all, std::vector<uint8_t>&& blob, std::vector<uint32_t>&& offs,
                                uint64_t vocab_size, uint64_t num_special_tokens, const std::string& version, bool has_audio,
                                int device_id, TokenizerError& err);
    bool save_model_cache(const std::string& path, const uint64_t key[2]) const;
    static Tekkenizer* load_model_cache(const std::string& path, const uint64_t key[2], int device_id, TokenizerError& err);
    TokenizerError decode_group(const uint32_t* ids, size_t n, bool is_special, SpecialTokenPolicy policy,
                                std::vector<std::string>& out) const;
    TokenizerError core_decode(const uint32_t* ranks, size_t n, std::string& out) const;

    tk_ctx* ctx_ = nullptr;
    uint64_t vocab_size_ = 0, num_special_tokens_ = 0;
    std::string version_;
    std::vector<SpecialTokenInfo> special_tokens_;               // by position (src/tekkenizer.rs:108-116)
    std::unordered_map<std::string, uint64_t> special_tokens_map_;  // token_str -> rank (:129-132)
    std::vector<std::string> vocab_;                             // lossy strings (:135-155)
    std::vector<uint8_t> blob_;                                  // rank table: bytes of rank i
    std::vector<uint32_t> offs_;
    bool has_audio_ = false;
    std::string pattern_;                                        // config.pattern as loaded (ignored unless opted in)
    bool from_cache_ = false;
};

// helpers exposed for tests
bool base64_decode_standard(const std::string& in, std::string& out, std::string& err);
bool utf8_valid(const uint8_t* p, size_t n);
std::string utf8_lossy(const uint8_t* p, size_t n);
TokenizerError parse_model_data(const char* json, size_t len, ModelData& out);

}  // namespace tekken
#endif
