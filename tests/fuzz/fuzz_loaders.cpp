// fuzz_loaders.cpp -- mutation fuzzer of the three file parsers of the host side, built with -fsanitize=address,undefined:
//   * tk_tokenizer_from_json        the hand-written JSON / base64 reader + the construction checks (csrc/tekkenizer.cpp)
//   * the model side file           TK_TABLE_CACHE_DIR/tk_model_<key>.bin   (Tekkenizer::load_model_cache)
//   * the tables side file          TK_TABLE_CACHE_DIR/tk_tables_<key>.bin  (tk_build_tables_cached, csrc/tk_tables.cpp)
// Inputs are truncated, bit-flipped, byte-replaced and spliced copies of a valid seed.  A mutant may load or fail; it must
// never crash, read out of bounds or (side files) change what a load returns.
//   fuzz_loaders <seed tekken.json> <scratch dir> <iterations> <seed>
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <string>
#include <vector>

#include "../../include/tekken_hip.h"
#include "../../tekken-rs_amd/csrc/tk_tables.h"

static uint64_t g_s = 88172645463325252ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }

static std::string slurp(const std::string& p) {
    std::string s;
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}
static void spit(const std::string& p, const std::string& s) {
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) { perror(p.c_str()); exit(2); }
    fwrite(s.data(), 1, s.size(), f);
    fclose(f);
}

static std::string mutate(const std::string& in) {
    std::string s = in;
    const int kind = (int)(rnd() % 6);
    if (s.empty()) return s;
    if (kind == 0) s.resize(rnd() % s.size());                                  // truncate
    else if (kind == 1) for (int k = 0, n = 1 + (int)(rnd() % 4); k < n; ++k) s[rnd() % s.size()] ^= (char)(1u << (rnd() % 8));   // bit flips
    else if (kind == 2) for (int k = 0, n = 1 + (int)(rnd() % 8); k < n; ++k) s[rnd() % s.size()] = (char)rnd();                   // bytes
    else if (kind == 3) { const size_t a = rnd() % s.size(), l = rnd() % 64; s.erase(a, l); }                                      // delete a span
    else if (kind == 4) { const size_t a = rnd() % s.size(), b = rnd() % s.size(), l = rnd() % 64; s.insert(a, s.substr(b, l)); }  // splice
    else { static const char* toks[] = {"{", "}", "[", "]", "\"", "\\", ",", ":", "null", "-1", "1e999", "18446744073709551616", "\\ud800", "=", "\x80"};
           s.insert(rnd() % s.size(), toks[rnd() % 15]); }
    return s;
}

// what a successfully loaded tokenizer answers (compared between the plain load and loads with damaged side files)
static std::string fingerprint(tk_tokenizer* t) {
    std::string fp = std::to_string(tk_tokenizer_vocab_size(t)) + "/" + std::to_string(tk_tokenizer_num_special_tokens(t)) + "/" + tk_tokenizer_version(t) + "/" + tk_tokenizer_json_pattern(t);
    const uint32_t ns = tk_tokenizer_num_special_tokens(t), vs = tk_tokenizer_vocab_size(t);
    uint32_t ids[6] = {ns + 72, ns + 105, 1, ns + 255, vs ? vs - 1 : 0, 2};
    for (int pol = 0; pol < 3; ++pol) {
        char* text = nullptr;
        size_t len = 0;
        const int rc = tk_tokenizer_decode(t, ids, 6, pol, &text, &len);
        fp += "|" + std::to_string(rc);
        if (rc == TK_OK) { fp.append(text, len); tk_free_text(text); }
    }
    for (uint32_t id : {0u, 1u, ns, ns + 65, vs - 1, vs, vs + 7}) {
        char* text = nullptr;
        size_t len = 0;
        if (tk_tokenizer_id_to_piece(t, id, &text, &len) == TK_OK) { fp.append(text, len); tk_free_text(text); } else fp += "!";
    }
    const uint8_t* blob; const uint32_t* offs; uint32_t n;
    if (tk_tokenizer_rank_table(t, &blob, &offs, &n) == TK_OK) {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t i = 0; i < offs[n]; ++i) { h ^= blob[i]; h *= 1099511628211ull; }
        fp += "#" + std::to_string(n) + ":" + std::to_string(h);
    }
    return fp;
}

static std::vector<std::string> list_dir(const std::string& d, const char* prefix) {
    std::vector<std::string> out;
    DIR* dir = opendir(d.c_str());
    if (!dir) return out;
    while (dirent* e = readdir(dir))
        if (strncmp(e->d_name, prefix, strlen(prefix)) == 0 && !strstr(e->d_name, ".tmp")) out.push_back(d + "/" + e->d_name);
    closedir(dir);
    return out;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: fuzz_loaders seed.json scratch_dir iterations seed\n"); return 2; }
    const std::string seed_json = slurp(argv[1]), dir = argv[2];
    const long iters = atol(argv[3]);
    g_s ^= (uint64_t)atoll(argv[4]) * 0x9E3779B97F4A7C15ull;
    if (seed_json.empty()) { fprintf(stderr, "empty seed\n"); return 2; }
    mkdir(dir.c_str(), 0777);
    long loaded = 0, failed = 0;

    // ---- 1. the JSON reader ---------------------------------------------------------------------------------------
    unsetenv("TK_TABLE_CACHE_DIR");
    tk_tokenizer* t = nullptr;
    if (tk_tokenizer_from_json(seed_json.data(), seed_json.size(), -1, &t) != TK_OK) { fprintf(stderr, "the seed does not load: %s\n", tk_tokenizer_last_error(nullptr)); return 3; }
    const std::string want = fingerprint(t);
    tk_tokenizer_destroy(t);
    for (long i = 0; i < iters; ++i) {
        std::string m = mutate(seed_json);
        if (rnd() % 4 == 0) m = mutate(m);
        t = nullptr;
        const int rc = tk_tokenizer_from_json(m.data(), m.size(), -1, &t);
        if (rc == TK_OK) { ++loaded; (void)fingerprint(t); tk_tokenizer_destroy(t); }
        else { ++failed; if (t != nullptr || !tk_tokenizer_last_error(nullptr)[0]) { fprintf(stderr, "failure without an error text (rc %d)\n", rc); return 4; } }
    }
    printf("json: %ld mutants, %ld loaded, %ld refused\n", iters, loaded, failed);

    // ---- 2. the model side file -----------------------------------------------------------------------------------
    const std::string cache = dir + "/cache", jpath = dir + "/seed.json";
    mkdir(cache.c_str(), 0777);
    spit(jpath, seed_json);
    setenv("TK_TABLE_CACHE_DIR", cache.c_str(), 1);
    if (tk_tokenizer_from_file(jpath.c_str(), -1, &t) != TK_OK) return 5;        // parses, writes the side file
    tk_tokenizer_destroy(t);
    std::vector<std::string> side = list_dir(cache, "tk_model_");
    if (side.size() != 1) { fprintf(stderr, "expected one model side file, found %zu\n", side.size()); return 5; }
    const std::string good = slurp(side[0]);
    if (tk_tokenizer_from_file(jpath.c_str(), -1, &t) != TK_OK || !tk_tokenizer_from_cache(t) || fingerprint(t) != want) { fprintf(stderr, "cached load differs\n"); return 6; }
    tk_tokenizer_destroy(t);
    long used = 0;
    for (long i = 0; i < iters / 4 + 8; ++i) {
        spit(side[0], mutate(good));
        t = nullptr;
        if (tk_tokenizer_from_file(jpath.c_str(), -1, &t) != TK_OK) { fprintf(stderr, "a damaged side file made the load fail\n"); return 7; }
        used += tk_tokenizer_from_cache(t);
        if (fingerprint(t) != want) { fprintf(stderr, "a damaged side file changed the result (mutant %ld)\n", i); tk_tokenizer_destroy(t); return 8; }
        tk_tokenizer_destroy(t);
    }
    printf("model side file: %ld mutants, %ld still accepted (payload bytes that the checks do not cover must not matter)\n", iters / 4 + 8, used);

    // ---- 3. the tables side file ----------------------------------------------------------------------------------
    if (tk_tokenizer_from_file(jpath.c_str(), -1, &t) != TK_OK) return 9;
    const uint8_t* blob; const uint32_t* offs; uint32_t n;
    tk_tokenizer_rank_table(t, &blob, &offs, &n);
    const uint32_t ns = tk_tokenizer_num_special_tokens(t);
    TkHostTables ref;
    std::string err;
    if (tk_build_tables(blob, offs, n, ns, 1, 2, ref, err) != TK_OK) { fprintf(stderr, "tables: %s\n", err.c_str()); return 9; }
    bool from_cache = false;
    TkHostTables a;
    if (tk_build_tables_cached(blob, offs, n, ns, 1, 2, a, err, &from_cache) != TK_OK) return 9;      // writes
    std::vector<std::string> ts = list_dir(cache, "tk_tables_");
    if (ts.size() != 1) { fprintf(stderr, "expected one tables side file, found %zu\n", ts.size()); return 9; }
    const std::string tgood = slurp(ts[0]);
    long tused = 0;
    for (long i = 0; i < iters / 4 + 8; ++i) {
        spit(ts[0], mutate(tgood));       // (the payload checksum catches damage anywhere, not only in the structure)
        TkHostTables b;
        from_cache = false;
        if (tk_build_tables_cached(blob, offs, n, ns, 1, 2, b, err, &from_cache) != TK_OK) { fprintf(stderr, "a damaged tables file made the build fail\n"); return 10; }
        tused += from_cache;
        if (b.key8_tab.size() != ref.key8_tab.size() || b.pair_tab != ref.pair_tab || b.pair2 != ref.pair2 || b.offs != ref.offs || b.blob != ref.blob ||
            memcmp(b.key8_tab.data(), ref.key8_tab.data(), ref.key8_tab.size() * sizeof(ref.key8_tab[0])) != 0) {
            fprintf(stderr, "a damaged tables file changed the tables (mutant %ld)\n", i);
            return 11;
        }
    }
    tk_tokenizer_destroy(t);
    printf("tables side file: %ld mutants, %ld accepted\n", iters / 4 + 8, tused);
    return 0;
}
