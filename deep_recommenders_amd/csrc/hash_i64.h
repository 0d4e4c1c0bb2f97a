// FarmHash Fingerprint64 of an int64 key's decimal text, register-resident (K1's arithmetic; see csrc/hash_bucket.hip for the
// provenance).  A header since round 6: the slot plan's front kernel (csrc/emb_plan.hip, dr_hash_sort_slots) hashes the raw keys itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drhash {

constexpr uint64_t K0 = 0xc3a5c85c97cb3127ULL;
constexpr uint64_t K1 = 0xb492b66fbe98f273ULL;
constexpr uint64_t K2 = 0x9ae16a3b2f90404fULL;

__device__ __forceinline__ uint64_t rot64(uint64_t v, int s) { return (v >> s) | (v << (64 - s)); }
__device__ __forceinline__ uint64_t smix(uint64_t v) { return v ^ (v >> 47); }
__device__ __forceinline__ uint64_t hl16(uint64_t u, uint64_t v, uint64_t mul) {
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    return b * mul;
}
// unaligned 8-byte window starting `off` bytes (0..8) into the 16-byte little-endian pair (lo, hi)
__device__ __forceinline__ uint64_t window64(uint64_t lo, uint64_t hi, int off) {
    if (off == 0) return lo;
    if (off == 8) return hi;
    return (lo >> (off * 8)) | (hi << (64 - off * 8));
}

// Fingerprint64 of a string of `len` (1..24) bytes held in w0 (bytes 0-7), w1 (8-15), w2 (16-23).
__device__ __forceinline__ uint64_t fp64_words(uint64_t w0, uint64_t w1, uint64_t w2, int len) {
    if (len <= 3) {
        uint32_t a = (uint32_t)(w0 & 0xff);
        uint32_t b = (uint32_t)((w0 >> ((len >> 1) * 8)) & 0xff);
        uint32_t c = (uint32_t)((w0 >> ((len - 1) * 8)) & 0xff);
        uint32_t y = a + (b << 8);
        uint32_t z = (uint32_t)len + (c << 2);
        return smix((uint64_t)y * K2 ^ (uint64_t)z * K0) * K2;
    }
    const uint64_t mul = K2 + (uint64_t)len * 2;
    if (len <= 7) {
        uint64_t a = w0 & 0xffffffffULL;
        uint64_t b = (w0 >> ((len - 4) * 8)) & 0xffffffffULL;
        return hl16((uint64_t)len + (a << 3), b, mul);
    }
    if (len <= 16) {
        uint64_t a = w0 + K2;
        uint64_t b = window64(w0, w1, len - 8);
        uint64_t c = rot64(b, 37) * mul + a;
        uint64_t d = (rot64(a, 25) + b) * mul;
        return hl16(c, d, mul);
    }
    // 17..24 bytes (HashLen17to32)
    uint64_t a = w0 * K1;
    uint64_t b = w1;
    uint64_t c = window64(w1, w2, len - 16) * mul;   // fetch64(s + len - 8)
    uint64_t d = window64(w0, w1, len - 16) * K2;    // fetch64(s + len - 16)
    return hl16(rot64(a + b, 43) + rot64(c, 30) + d, a + rot64(b + K2, 18) + c, mul);
}

// [TF] as_string(int64): plain decimal, leading '-' for negatives, no padding.  Characters are
// pushed least-significant first into a 192-bit little-endian shift register so byte 0 ends up
// holding the first character of the text.
// Non-negative keys, fast path: the 20-digit zero-padded decimal text is built RIGHT-aligned in the 24-byte register
// (w0 | w1 | w2) with every digit at a compile-time byte position (32-bit digit extraction from three chunks of 4 / 8 / 8
// digits), then left-aligned with ONE 192-bit byte shift -- instead of a 192-bit shift per digit and a 64-bit division per
// digit.  Byte b of the text lives in word b / 8, bits 8 * (b % 8).
__device__ __forceinline__ void put_digits8(uint32_t v, int first_byte, uint64_t& w0, uint64_t& w1, uint64_t& w2, int& top,
                                            int digit_base) {
    // v < 1e8: its 8 digits, least significant first, go to bytes first_byte + 7 ... first_byte
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t q = v / 10u;
        const uint32_t d = v - q * 10u;
        v = q;
        const int b = first_byte + 7 - j;                 // compile-time after unrolling
        const uint64_t ch = (uint64_t)(0x30u + d) << (8 * (b & 7));
        if (b < 8) w0 |= ch; else if (b < 16) w1 |= ch; else w2 |= ch;
        if (d != 0) top = digit_base + j + 1;             // number of significant digits so far
    }
}

__device__ __forceinline__ void text_u64_fast(uint64_t mag, uint64_t& o0, uint64_t& o1, uint64_t& o2, int& olen) {
    const uint64_t q8 = mag / 100000000ull;               // mag = q8 * 1e8 + lo
    const uint32_t lo = (uint32_t)(mag - q8 * 100000000ull);
    const uint64_t q16 = q8 / 100000000ull;               // q8 = q16 * 1e8 + mid,  q16 < 1845
    const uint32_t mid = (uint32_t)(q8 - q16 * 100000000ull);
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    int top = 1;                                          // "0" has one digit
    put_digits8(lo, 16, w0, w1, w2, top, 0);              // text bytes 16..23
    if (q8 != 0) put_digits8(mid, 8, w0, w1, w2, top, 8); // text bytes 8..15
    else { w1 = 0x3030303030303030ull; }
    uint32_t hi = (uint32_t)q16;                          // up to 4 digits: text bytes 4..7
    uint64_t hw = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t q = hi / 10u;
        const uint32_t d = hi - q * 10u;
        hi = q;
        hw |= (uint64_t)(0x30u + d) << (8 * (7 - j));
        if (d != 0) top = 16 + j + 1;
    }
    w0 |= hw;                                             // bytes 0..3 stay 0 and are shifted out below
    // left-align: drop the first (24 - top) bytes
    const int drop = 24 - top;                            // 4 .. 23
    uint64_t a = w0, b = w1, c = w2;
    if (drop >= 16) { a = c; b = 0; c = 0; }
    else if (drop >= 8) { a = b; b = c; c = 0; }
    const int sh = (drop & 7) * 8;
    if (sh != 0) {
        a = (a >> sh) | (b << (64 - sh));
        b = (b >> sh) | (c << (64 - sh));
        c = c >> sh;
    }
    o0 = a; o1 = b; o2 = c; olen = top;
}

__device__ __forceinline__ uint64_t hash_i64_key(int64_t key) {
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    int len = 0;
    if (key >= 0) {
        text_u64_fast((uint64_t)key, w0, w1, w2, len);
    } else {                                               // negative keys (rare): the simple digit-at-a-time form
        uint64_t mag = (uint64_t)0 - (uint64_t)key;
        do {
            uint64_t q = mag / 10;
            uint64_t ch = (uint64_t)'0' + (mag - q * 10);
            w2 = (w2 << 8) | (w1 >> 56);
            w1 = (w1 << 8) | (w0 >> 56);
            w0 = (w0 << 8) | ch;
            mag = q;
            ++len;
        } while (mag != 0);
        w2 = (w2 << 8) | (w1 >> 56);
        w1 = (w1 << 8) | (w0 >> 56);
        w0 = (w0 << 8) | (uint64_t)'-';
        ++len;
    }
    return fp64_words(w0, w1, w2, len);
}

// key -> bucket id exactly as hash_bucket_i64_kernel does it: -1 stays -1 ([TF] dense int input: -1 entries are dropped), nb == 0 passes
// the key through, otherwise Fingerprint64(text(key)) mod nb by Barrett reduction with inv = floor((2^64 - 1) / nb)
__device__ __forceinline__ int64_t bucket_of_key(int64_t key, uint64_t nb, uint64_t inv) {
    if (key == -1) return -1;
    if (nb == 0) return key;
    const uint64_t h = hash_i64_key(key);
    uint64_t r = h - __umul64hi(h, inv) * nb;              // Barrett: quotient estimate is low by at most 2
    if (r >= nb) r -= nb;
    if (r >= nb) r -= nb;
    return (int64_t)r;
}

}  // namespace drhash
