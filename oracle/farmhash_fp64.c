/*
 * ORACLE — test infrastructure only.  Nothing under oracle/ is ever imported, linked
 * or executed by the product path (deep_recommenders_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * CPU restatement (plain C) of the integer id path the reference reaches through
 * TensorFlow (third-party, NOT vendored under /root/reference, version unpinned:
 * requirements.txt:1-4 does not list it; CI matrix 1.15..2.6):
 *
 *   tf.feature_column.categorical_column_with_hash_bucket(key, N)
 *     call sites: examples/train_fm_on_movielens_estimator.py:12-13,20-21
 *                 tests/keras/test_fm.py:70-73, tests/keras/test_deepfm.py:19-22
 *   = string_to_hash_bucket_fast(as_string(x), N)
 *   = int64( Fingerprint64(utf8 bytes) mod uint64(N) ),
 *   where Fingerprint64 is FarmHash `farmhashna::Hash64` (Google FarmHash 1.1,
 *   published algorithm restated below from its specification).
 *
 * Pinning status (oracle/README.md): the 1-3 / 4-7 / 8-16 byte branches are pinned by the upstream
 * TensorFlow vectors in tests/golden/reference_kats.json; the whole 0..32-byte range (so every
 * decimal rendering of an int64 key) by third-party CityHash64 answers in
 * tests/golden/cityhash64_le32_vectors.json (farmhashna::Hash64 == CityHash64 v1.1 up to 32 bytes).
 * The 33-64 and >64 byte branches have NO external vector available in this container: they are
 * cross-checked against an independent Python restatement (tests/test_oracle_pinning.py) only --
 * "parity unpinned" for byte strings longer than 32 bytes.
 */
#include <stdint.h>
#include <string.h>
#include <stdio.h>

#define K0 0xc3a5c85c97cb3127ULL
#define K1 0xb492b66fbe98f273ULL
#define K2 0x9ae16a3b2f90404fULL

static uint64_t fetch64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint64_t fetch32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rot(uint64_t v, int s) { return s == 0 ? v : (v >> s) | (v << (64 - s)); }
static uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

static uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    b *= mul;
    return b;
}

static uint64_t hash_len0to16(const uint8_t *s, size_t len) {
    if (len >= 8) {
        uint64_t mul = K2 + len * 2;
        uint64_t a = fetch64(s) + K2;
        uint64_t b = fetch64(s + len - 8);
        uint64_t c = rot(b, 37) * mul + a;
        uint64_t d = (rot(a, 25) + b) * mul;
        return hash_len16(c, d, mul);
    }
    if (len >= 4) {
        uint64_t mul = K2 + len * 2;
        uint64_t a = fetch32(s);
        return hash_len16(len + (a << 3), fetch32(s + len - 4), mul);
    }
    if (len > 0) {
        uint8_t a = s[0], b = s[len >> 1], c = s[len - 1];
        uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
        uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
        return shift_mix(y * K2 ^ z * K0) * K2;
    }
    return K2;
}

static uint64_t hash_len17to32(const uint8_t *s, size_t len) {
    uint64_t mul = K2 + len * 2;
    uint64_t a = fetch64(s) * K1;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * K2;
    return hash_len16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + K2, 18) + c, mul);
}

static void weak32(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b,
                   uint64_t *o1, uint64_t *o2) {
    a += w;
    b = rot(b + a + z, 21);
    uint64_t c = a;
    a += x;
    a += y;
    b += rot(a, 44);
    *o1 = a + z;
    *o2 = b + c;
}
static void weak32s(const uint8_t *s, uint64_t a, uint64_t b, uint64_t *o1, uint64_t *o2) {
    weak32(fetch64(s), fetch64(s + 8), fetch64(s + 16), fetch64(s + 24), a, b, o1, o2);
}

static uint64_t hash_len33to64(const uint8_t *s, size_t len) {
    uint64_t mul = K2 + len * 2;
    uint64_t a = fetch64(s) * K2;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * K2;
    uint64_t y = rot(a + b, 43) + rot(c, 30) + d;
    uint64_t z = hash_len16(y, a + rot(b + K2, 18) + c, mul);
    uint64_t e = fetch64(s + 16) * mul;
    uint64_t f = fetch64(s + 24);
    uint64_t g = (y + fetch64(s + len - 32)) * mul;
    uint64_t h = (z + fetch64(s + len - 24)) * mul;
    return hash_len16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
}

uint64_t oracle_fingerprint64(const uint8_t *s, size_t len) {
    const uint64_t seed = 81;
    if (len <= 32) return len <= 16 ? hash_len0to16(s, len) : hash_len17to32(s, len);
    if (len <= 64) return hash_len33to64(s, len);
    uint64_t x = seed, y = seed * K1 + 113, z = shift_mix(y * K2 + 113) * K2;
    uint64_t v1 = 0, v2 = 0, w1 = 0, w2 = 0, t;
    x = x * K2 + fetch64(s);
    const uint8_t *end = s + ((len - 1) / 64) * 64;
    const uint8_t *last64 = end + ((len - 1) & 63) - 63;
    do {
        x = rot(x + y + v1 + fetch64(s + 8), 37) * K1;
        y = rot(y + v2 + fetch64(s + 48), 42) * K1;
        x ^= w2;
        y += v1 + fetch64(s + 40);
        z = rot(z + w1, 33) * K1;
        weak32s(s, v2 * K1, x + w1, &v1, &v2);
        weak32s(s + 32, z + w2, y + fetch64(s + 16), &w1, &w2);
        t = z; z = x; x = t;
        s += 64;
    } while (s != end);
    uint64_t mul = K1 + ((z & 0xff) << 1);
    s = last64;
    w1 += ((len - 1) & 63);
    v1 += w1;
    w1 += v1;
    x = rot(x + y + v1 + fetch64(s + 8), 37) * mul;
    y = rot(y + v2 + fetch64(s + 48), 42) * mul;
    x ^= w2 * 9;
    y += v1 * 9 + fetch64(s + 40);
    z = rot(z + w1, 33) * mul;
    weak32s(s, v2 * mul, x + w1, &v1, &v2);
    weak32s(s + 32, z + w2, y + fetch64(s + 16), &w1, &w2);
    t = z; z = x; x = t;
    return hash_len16(hash_len16(v1, w1, mul) + shift_mix(y) * K0 + z,
                      hash_len16(v2, w2, mul) + x, mul);
}

/* [TF] as_string on int64: plain decimal, '-' sign, no padding. */
static size_t i64_to_dec(int64_t v, uint8_t *buf) {
    char tmp[24];
    int n = snprintf(tmp, sizeof tmp, "%lld", (long long)v);
    memcpy(buf, tmp, (size_t)n);
    return (size_t)n;
}

/* hash-bucket column over int64 keys.  [TF] dense int input: entries == -1 are dropped
 * before hashing (B1) -> reported here as id -1 ("missing"). */
void oracle_hash_bucket_i64(const int64_t *keys, int64_t n, uint64_t num_buckets, int64_t *out) {
    uint8_t buf[24];
    for (int64_t i = 0; i < n; ++i) {
        if (keys[i] == -1) { out[i] = -1; continue; }
        size_t len = i64_to_dec(keys[i], buf);
        out[i] = (int64_t)(oracle_fingerprint64(buf, len) % num_buckets);
    }
}

/* hash-bucket column over byte strings in CSR form.  [TF] dense string input: "" dropped. */
void oracle_hash_bucket_bytes(const uint8_t *bytes, const int64_t *offsets, int64_t n,
                              uint64_t num_buckets, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) {
        size_t len = (size_t)(offsets[i + 1] - offsets[i]);
        if (len == 0) { out[i] = -1; continue; }
        out[i] = (int64_t)(oracle_fingerprint64(bytes + offsets[i], len) % num_buckets);
    }
}

void oracle_fingerprint64_bytes(const uint8_t *bytes, const int64_t *offsets, int64_t n, uint64_t *out) {
    for (int64_t i = 0; i < n; ++i)
        out[i] = oracle_fingerprint64(bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
}
