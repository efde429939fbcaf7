/* CPU driver of csrc/tk_utf8_swar.h for tests/test_utf8_swar.py (TEST INFRASTRUCTURE): walks a document the way
   tk_decode_validate_kernel does -- aligned dwords, bytes outside [s0, s1) read as NUL, the run-start bits of the six positions
   around every dword, every dword up to the one that holds position s1 -- and says whether any position is in error. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../tekken-rs_amd/csrc/tk_utf8_swar.h"

static uint32_t load_masked(const uint8_t* b, uint64_t q, uint64_t s0, uint64_t s1) {
    if (q >= s1 || q + 4 <= s0) return 0u;
    uint32_t w;
    memcpy(&w, b + q, 4);
    if (q < s0) w &= 0xFFFFFFFFu << (8u * (uint32_t)(s0 - q));
    if (q + 4 > s1) w &= 0xFFFFFFFFu >> (8u * (uint32_t)(q + 4 - s1));
    return w;
}

static uint32_t load_rb(const uint32_t* run_bits, uint64_t q) {   /* bit k: a run starts at q - 4 + k */
    if (q < 4) return run_bits[0] << 4;
    const uint64_t rb = q - 4;
    const uint32_t r0 = run_bits[rb >> 5], r1 = run_bits[(rb >> 5) + 1];
    const uint32_t sh = (uint32_t)(rb & 31u);
    return sh ? ((r0 >> sh) | (r1 << (32u - sh))) : r0;
}

/* buf must be readable up to the dword that holds s1 (the caller pads); run_bits: one bit per byte position, two words of slack */
int tku8_check_doc(const uint8_t* buf, uint64_t s0, uint64_t s1, const uint32_t* run_bits) {
    if (s1 == s0) return 0;
    uint32_t pw = 0;
    for (uint64_t q = s0 & ~3ull; q <= s1; q += 4) {
        const uint32_t w = load_masked(buf, q, s0, s1);
        if (tku8_err4(pw, w, load_rb(run_bits, q))) return 1;
        pw = w;
    }
    return 0;
}
