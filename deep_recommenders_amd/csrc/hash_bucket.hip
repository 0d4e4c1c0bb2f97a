// K1 / K2 — integer id path on gfx950: decimal rendering + FarmHash Fingerprint64 + mod (bit-exact),
// vocabulary-list lookup.  HBM-bound element-wise integer work: 8 B in, 8 B out per id; the hash
// itself is ~150 integer VALU ops per key and hides under the memory stream.
//
// Replaces [TF] categorical_column_with_hash_bucket / _with_vocabulary_list reached from
// examples/train_fm_on_movielens_estimator.py:12-23 (reference root).  FarmHash (farmhashna::Hash64,
// FarmHash 1.1) is restated here from its published algorithm, register-resident: an int64 key's
// decimal text is at most 20 bytes, built directly into three little-endian 64-bit words.
#include "dr_common.h"
#include "hash_i64.h"

namespace {
using namespace drhash;

__global__ __launch_bounds__(256) void hash_bucket_i64_kernel(const int64_t* __restrict__ keys, int64_t n,
                                                              int32_t C,
                                                              const uint64_t* __restrict__ col_buckets,
                                                              int64_t* __restrict__ out) {
    // per-column Barrett constants: one 64-bit division per column per block instead of one per key
    extern __shared__ uint64_t cb[];                       // [C] buckets, then [C] floor((2^64 - 1) / buckets)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const uint64_t nb = col_buckets[c];
        cb[c] = nb;
        cb[C + c] = nb != 0 ? ~0ull / nb : 0;
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int col = (int)(i0 % C);
    const int col_step = (int)(stride % C);
    for (int64_t i = i0; i < n; i += stride) {
        out[i] = bucket_of_key(keys[i], cb[col], cb[C + col]);      // (-1 stays -1; buckets == 0: pass-through column)
        col += col_step;
        if (col >= C) col -= C;
    }
}

// ---- general byte strings (any length), bytes in HBM -------------------------------------
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
__device__ __forceinline__ uint64_t ld32(const uint8_t* p) {
    return (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
}
__device__ __forceinline__ void weak32(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b,
                                       uint64_t& o1, uint64_t& o2) {
    a += w;
    b = rot64(b + a + z, 21);
    const uint64_t c = a;
    a += x;
    a += y;
    b += rot64(a, 44);
    o1 = a + z;
    o2 = b + c;
}
__device__ __forceinline__ void weak32s(const uint8_t* s, uint64_t a, uint64_t b, uint64_t& o1, uint64_t& o2) {
    weak32(ld64(s), ld64(s + 8), ld64(s + 16), ld64(s + 24), a, b, o1, o2);
}

__device__ uint64_t fp64_bytes(const uint8_t* s, int64_t len) {
    if (len == 0) return K2;
    if (len <= 3) {
        uint32_t y = (uint32_t)s[0] + ((uint32_t)s[len >> 1] << 8);
        uint32_t z = (uint32_t)len + ((uint32_t)s[len - 1] << 2);
        return smix((uint64_t)y * K2 ^ (uint64_t)z * K0) * K2;
    }
    uint64_t mul = K2 + (uint64_t)len * 2;
    if (len <= 7) return hl16((uint64_t)len + (ld32(s) << 3), ld32(s + len - 4), mul);
    if (len <= 16) {
        uint64_t a = ld64(s) + K2, b = ld64(s + len - 8);
        return hl16(rot64(b, 37) * mul + a, (rot64(a, 25) + b) * mul, mul);
    }
    if (len <= 32) {
        uint64_t a = ld64(s) * K1, b = ld64(s + 8), c = ld64(s + len - 8) * mul, d = ld64(s + len - 16) * K2;
        return hl16(rot64(a + b, 43) + rot64(c, 30) + d, a + rot64(b + K2, 18) + c, mul);
    }
    if (len <= 64) {
        uint64_t a = ld64(s) * K2, b = ld64(s + 8), c = ld64(s + len - 8) * mul, d = ld64(s + len - 16) * K2;
        uint64_t y = rot64(a + b, 43) + rot64(c, 30) + d;
        uint64_t z = hl16(y, a + rot64(b + K2, 18) + c, mul);
        uint64_t e = ld64(s + 16) * mul, f = ld64(s + 24);
        uint64_t g = (y + ld64(s + len - 32)) * mul, h = (z + ld64(s + len - 24)) * mul;
        return hl16(rot64(e + f, 43) + rot64(g, 30) + h, e + rot64(f + a, 18) + g, mul);
    }
    uint64_t x = 81, y = 81 * K1 + 113, z = smix(y * K2 + 113) * K2;
    uint64_t v1 = 0, v2 = 0, w1 = 0, w2 = 0, t;
    x = x * K2 + ld64(s);
    const uint8_t* end = s + ((len - 1) / 64) * 64;
    const uint8_t* last64 = end + ((len - 1) & 63) - 63;
    do {
        x = rot64(x + y + v1 + ld64(s + 8), 37) * K1;
        y = rot64(y + v2 + ld64(s + 48), 42) * K1;
        x ^= w2;
        y += v1 + ld64(s + 40);
        z = rot64(z + w1, 33) * K1;
        weak32s(s, v2 * K1, x + w1, v1, v2);
        weak32s(s + 32, z + w2, y + ld64(s + 16), w1, w2);
        t = z; z = x; x = t;
        s += 64;
    } while (s != end);
    mul = K1 + ((z & 0xff) << 1);
    s = last64;
    w1 += ((len - 1) & 63);
    v1 += w1;
    w1 += v1;
    x = rot64(x + y + v1 + ld64(s + 8), 37) * mul;
    y = rot64(y + v2 + ld64(s + 48), 42) * mul;
    x ^= w2 * 9;
    y += v1 * 9 + ld64(s + 40);
    z = rot64(z + w1, 33) * mul;
    weak32s(s, v2 * mul, x + w1, v1, v2);
    weak32s(s + 32, z + w2, y + ld64(s + 16), w1, w2);
    t = z; z = x; x = t;
    return hl16(hl16(v1, w1, mul) + smix(y) * K0 + z, hl16(v2, w2, mul) + x, mul);
}

__global__ __launch_bounds__(256) void hash_bucket_bytes_kernel(const uint8_t* __restrict__ bytes,
                                                                const int64_t* __restrict__ offsets, int64_t n,
                                                                uint64_t nb, int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t o = offsets[i], len = offsets[i + 1] - o;
        out[i] = len == 0 ? -1 : (int64_t)(fp64_bytes(bytes + o, len) % nb);   // [TF] "" is dropped
    }
}

// ---- K2 vocabulary lookup: the vocab (2 / 7 / 21 entries in the reference) is staged in LDS ---
constexpr int VOCAB_LDS = 4096;

__global__ __launch_bounds__(256) void vocab_lookup_i64_kernel(const int64_t* __restrict__ keys, int64_t n,
                                                               const int64_t* __restrict__ vocab, int32_t m,
                                                               int64_t* __restrict__ out) {
    __shared__ int64_t sv[VOCAB_LDS];
    const int ml = m < VOCAB_LDS ? m : VOCAB_LDS;
    for (int j = threadIdx.x; j < ml; j += blockDim.x) sv[j] = vocab[j];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t key = keys[i];
        int64_t id = -1;
        if (key != -1) {                           // -1 entries are dropped before lookup
            for (int j = 0; j < ml; ++j)
                if (sv[j] == key) { id = j; break; }           // first match wins
            if (id < 0)
                for (int j = ml; j < m; ++j)
                    if (vocab[j] == key) { id = j; break; }
        }
        out[i] = id;
    }
}

__global__ __launch_bounds__(256) void vocab_lookup_bytes_kernel(const uint8_t* __restrict__ bytes,
                                                                 const int64_t* __restrict__ offsets, int64_t n,
                                                                 const uint8_t* __restrict__ vbytes,
                                                                 const int64_t* __restrict__ voffsets, int32_t m,
                                                                 int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t o = offsets[i], len = offsets[i + 1] - o;
        int64_t id = -1;
        if (len > 0) {
            for (int j = 0; j < m && id < 0; ++j) {
                const int64_t vo = voffsets[j], vl = voffsets[j + 1] - vo;
                if (vl != len) continue;
                bool eq = true;
                for (int64_t t = 0; t < len; ++t)
                    if (bytes[o + t] != vbytes[vo + t]) { eq = false; break; }
                if (eq) id = j;
            }
        }
        out[i] = id;
    }
}

}  // namespace

extern "C" int dr_hash_bucket_i64(const int64_t* keys, int64_t B, int32_t C, const uint64_t* col_buckets,
                                  int64_t* ids_out, dr_stream_t stream) {
    if (B < 0 || C <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!keys || !col_buckets || !ids_out) return DR_EINVAL;
    const int64_t n = B * C;
    hipLaunchKernelGGL(hash_bucket_i64_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 2 * C * sizeof(uint64_t), dr_s(stream), keys, n, C,
                       col_buckets, ids_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_hash_bucket_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n, uint64_t num_buckets,
                                    int64_t* ids_out, dr_stream_t stream) {
    if (n < 0 || num_buckets == 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!bytes || !offsets || !ids_out) return DR_EINVAL;
    hipLaunchKernelGGL(hash_bucket_bytes_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), bytes, offsets,
                       n, num_buckets, ids_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_vocab_lookup_i64(const int64_t* keys, int64_t n, const int64_t* vocab, int32_t vocab_len,
                                   int64_t* ids_out, dr_stream_t stream) {
    if (n < 0 || vocab_len < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!keys || !ids_out || (vocab_len > 0 && !vocab)) return DR_EINVAL;
    hipLaunchKernelGGL(vocab_lookup_i64_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), keys, n, vocab,
                       vocab_len, ids_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_vocab_lookup_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n,
                                     const uint8_t* vocab_bytes, const int64_t* vocab_offsets, int32_t vocab_len,
                                     int64_t* ids_out, dr_stream_t stream) {
    if (n < 0 || vocab_len < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!bytes || !offsets || !ids_out || (vocab_len > 0 && (!vocab_bytes || !vocab_offsets))) return DR_EINVAL;
    hipLaunchKernelGGL(vocab_lookup_bytes_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), bytes,
                       offsets, n, vocab_bytes, vocab_offsets, vocab_len, ids_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// ids [B, F] int64 -> out [F][B] int32 (field-major): the row indices of one field for consecutive examples become one 128-byte
// line per 32 examples -- what the gather form of the first-layer wgrad (dr_bf3_wgrad_emb) reads per k-tile.  -1 stays -1.
namespace {
__global__ __launch_bounds__(256) void ids_transpose_i32_kernel(const int64_t* __restrict__ ids, int64_t B, int32_t F,
                                                                int32_t* __restrict__ out) {
    __shared__ int32_t tile[64][65];
    // a block moves 64 examples x up to 64 fields through LDS so that both sides are coalesced
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int f0 = 0; f0 < F; f0 += 64) {
        for (int r = ty; r < 64; r += 4) {                               // rows = examples, tx = field
            const int64_t b = b0 + r;
            const int f = f0 + tx;
            tile[r][tx] = (b < B && f < F) ? (int32_t)ids[b * F + f] : -1;
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {                               // rows = fields, tx = example
            const int f = f0 + r;
            const int64_t b = b0 + tx;
            if (f < F && b < B) out[(int64_t)f * B + b] = tile[tx][r];
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int dr_ids_transpose_i32(const int64_t* ids, int64_t B, int32_t F, int32_t* out, dr_stream_t stream) {
    if (B < 0 || F <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!ids || !out) return DR_EINVAL;
    hipLaunchKernelGGL(ids_transpose_i32_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, dr_s(stream), ids, B, F, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
