// tk_utf8_swar.h -- UTF-8 validation of FOUR bytes at a time in one 32-bit register, with hard boundaries.
//
// What tk_decode_validate_kernel (tk_decode.hip) runs per lane and step; kept in a header of plain C so that the CPU test
// (tests/test_utf8_swar.py, tests/utf8_swar_check.c) drives the very same function over random byte strings with random run
// boundaries against a scalar RFC 3629 validator.
//
// The reference validates every non-special run of a document on its own (CoreBPE::decode -> String::from_utf8, reference
// src/tekkenizer.rs:552-555): a code point may not span a run start, and a document is a sequence of runs.
//
// Method: the three-table look-up of Keiser & Lemire ("Validating UTF-8 in less than one instruction per byte", the simdjson
// validator): for every byte, the high nibble of the byte before it, the low nibble of the byte before it and its own high
// nibble index three 16-entry tables of error classes; the AND of the three entries is non-zero exactly for the two-byte
// windows that are malformed (too short, too long, overlong 2 / 3 / 4-byte forms, surrogates, beyond U+10FFFF), with one bit
// (0x80) that says "a continuation byte behind a continuation byte" -- legal exactly where a 3- or 4-byte lead stands two or
// three bytes back.  Four bytes are looked up at once: a 16-entry byte table is two v_perm_b32 (eight entries each) and a
// bit-field insert.  Boundaries: a byte in front of a run start counts as NUL for the bytes behind the start (so a
// continuation byte that opens a run is "too long", a lead that closes one is caught by the explicit test below), and a
// sequence that is still open at a run start is an error of its own.  The end of the document needs nothing special: the
// bytes behind it read as NUL, and a NUL behind an open sequence is "too short".
#ifndef TK_UTF8_SWAR_H
#define TK_UTF8_SWAR_H
#include <stdint.h>

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define TK_U8_FN __device__ __forceinline__
TK_U8_FN uint32_t tku8_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
#if defined(__HIPCC__)
#define TK_U8_FN __host__ __device__ inline
#else
#define TK_U8_FN static inline
#endif
TK_U8_FN uint32_t tku8_perm(uint32_t hi, uint32_t lo, uint32_t sel) {   // v_perm_b32 for selectors 0..7
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8u * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return r;
}
#endif

// entry idx[i] (0..15, one per byte of idx) of the 16-byte table {t3:t2:t1:t0} (t0 = entries 0..3, little-endian)
TK_U8_FN uint32_t tku8_lookup16(uint32_t idx, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t lo = tku8_perm(t1, t0, sel), hi = tku8_perm(t3, t2, sel);
    const uint32_t m = ((idx >> 3) & 0x01010101u) * 0xFFu;     // 0xFF where the index is 8..15
    return (lo & ~m) | (hi & m);
}

// bits 0..3 of x -> 0xFF in bytes 0..3
TK_U8_FN uint32_t tku8_spread4(uint32_t x) { return (((x & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu; }

// pw = the four bytes in front of w, w = the four bytes judged (positions q .. q + 3), rb = run-start bits: bit 2 + j = a run
// starts at position q - 2 + j (j = 0 .. 5).  Bytes outside the document must read as NUL in pw / w.  Non-zero: one of the
// four positions is in error.
TK_U8_FN uint32_t tku8_err4(uint32_t pw, uint32_t w, uint32_t rb) {
    // error classes (Keiser & Lemire)
    //   TOO_SHORT 01  TOO_LONG 02  OVERLONG_3 04  TOO_LARGE 08  SURROGATE 10  OVERLONG_2 20  TOO_LARGE_1000 / OVERLONG_4 40  TWO_CONTS 80
    const uint64_t back = ((uint64_t)w << 32) | pw;
    const uint32_t p1 = (uint32_t)(back >> 24), p2 = (uint32_t)(back >> 16), p3 = (uint32_t)(back >> 8);   // the bytes 1 / 2 / 3 in front of each
    const uint32_t z1 = tku8_spread4(rb >> 4);                    // a run starts AT the byte
    const uint32_t z2 = z1 | tku8_spread4(rb >> 3);               // ... or at the one before
    const uint32_t z3 = z2 | tku8_spread4(rb >> 2);
    const uint32_t s1 = p1 & ~z1, s2 = p2 & ~z2, s3 = p3 & ~z3;   // what is in front of a run start does not count behind it
    const uint32_t t1 = tku8_lookup16((s1 >> 4) & 0x0F0F0F0Fu, 0x02020202u, 0x02020202u, 0x80808080u, 0x49150121u);
    const uint32_t t2 = tku8_lookup16(s1 & 0x0F0F0F0Fu, 0x8383A3E7u, 0xCBCBCB8Bu, 0xCBCBCBCBu, 0xCBCBDBCBu);
    const uint32_t t3 = tku8_lookup16((w >> 4) & 0x0F0F0F0Fu, 0x01010101u, 0x01010101u, 0xBABAAEE6u, 0x01010101u);
    const uint32_t sc = t1 & t2 & t3;
    // a 3- / 4-byte lead two / three bytes back: the byte must be a continuation behind a continuation (top three / four bits set)
    const uint32_t m2 = s2 & (s2 << 1) & (s2 << 2) & 0x80808080u;
    const uint32_t m3 = s3 & (s3 << 1) & (s3 << 2) & (s3 << 3) & 0x80808080u;
    uint32_t err = sc ^ (m2 | m3);
    // a sequence still open where a run starts (the REAL bytes in front): a lead right before, a 3- / 4-byte lead two back, a 4-byte
    // lead three back
    const uint32_t o1 = p1 & (p1 << 1) & 0x80808080u;
    const uint32_t o2 = p2 & (p2 << 1) & (p2 << 2) & 0x80808080u;
    const uint32_t o3 = p3 & (p3 << 1) & (p3 << 2) & (p3 << 3) & 0x80808080u;
    err |= (o1 | o2 | o3) & z1;
    return err;
}

#endif
