// K3 / K4 — fused embedding gather + segment-mean pool (+ first-order + FM second-order) forward and
// its transposed scatter-add backward, for gfx950.  Both are HBM-bound (random 4*D-byte row reads /
// read-modify-writes against a table slab far larger than L2+MALL) — no MFMA here.
//
// Replaces the per-column [TF] safe_embedding_lookup_sparse + tf.stack + tf.concat + indicator/Dense(1)
// + FM reduce_sum/pow/subtract op chain of keras/models/ranking/fm.py:23-37,54-64 and deepfm.py:36-47
// (reference root), and its autodiff.
//
// Work decomposition (wave64): one wavefront owns one example at a time (grid-stride).  A table row of
// D floats is read as D/4 float4's by LPR = D/4 adjacent lanes (one global_load_dwordx4 per lane, a
// row is one contiguous 4*D-byte segment), so a wave serves NS = 64/LPR fields per load instruction and
// keeps U such instructions in flight before consuming any of them (memory-level parallelism for the
// ~2 us HBM miss latency).  Ids of the example are loaded once, coalesced, one per lane, and broadcast
// with wave shuffles.  FM partial sums live in registers; the cross-field reduction is a butterfly of
// __shfl_xor over the slot bits; the D-reduction for the second-order scalar is a butterfly over the
// sub-lane bits.  The pooled rows are written once, directly in the [B, F*D] "concat" layout (which
// is byte-identical to the [B, F, D] "stack" layout), so tf.stack / tf.concat cost nothing.
#include "dr_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma2(float4 a, float4 acc) {
    return make_float4(fmaf(a.x, a.x, acc.x), fmaf(a.y, a.y, acc.y), fmaf(a.z, a.z, acc.z), fmaf(a.w, a.w, acc.w));
}
// streaming (non-temporal) 16-byte store: the pooled rows are consumed much later by the GEMM, from HBM
typedef float ep_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void f4store_nt(float* p, float4 v) {
    const ep_f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<ep_f4v*>(p));
}
// ... and the table rows are read once: nontemporal too, so that they do not wash the first-order lines out of the caches (round 4,
// the same finding as K4's: emb_sorted.hip)
__device__ __forceinline__ float4 f4load_nt(const float* p) {
    const ep_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ep_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 f4shfl_xor(float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64));
}

// ------------------------------------------------------------------------------------------------
// Forward, fast path: every field single-valued (C == F), F <= 64.  Branch-free: every memory instruction of the
// loop body is UNCONDITIONAL:
//  * a conditional load/store always gets an s_cbranch_execz around it, and across those branches the
//    compiler's wait-count bookkeeping turns pessimistic (s_waitcnt vmcnt(0) before every access), which
//    serialised the first-order loads and the concat stores into one HBM round trip each;
//  * so out-of-range work is clamped instead of masked: a slot past the last field repeats field F-1 (same
//    row, same destination, same value: a benign duplicate store), a missing id (-1) reads row 0 and its
//    value is replaced by zeros in registers, lanes past D/4 repeat the last float4 of the row;
//  * the next example's ids are fetched before this example's rows, so the loop-top wait is already
//    satisfied (vmcnt retires in order) and never covers this example's stores.
// ------------------------------------------------------------------------------------------------
template <int LPR, int U>
__global__ __launch_bounds__(256) void emb_pool_fwd_sv_kernel(
    const int64_t* __restrict__ ids, int64_t B, int32_t F, const int64_t* __restrict__ row_base,
    const float* __restrict__ table, int32_t D, const float* __restrict__ lin_w, const float* __restrict__ lin_bias,
    float* __restrict__ concat, int64_t ld, float* __restrict__ sum_x, float* __restrict__ fm_logit, float so) {
    constexpr int NS = DR_WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPR;
    const int sub = lane % LPR;
    const int nq = D >> 2;
    const bool dvalid = sub < nq;
    const int subc = dvalid ? sub : nq - 1;
    const float* __restrict__ lsrc = lin_w != nullptr ? lin_w : table;     // value discarded when lin_w == NULL
    const bool use_lin = lin_w != nullptr && sub == 0;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave0 >= B) return;
    const int lanec = lane < F ? lane : F - 1;
    const int64_t my_base = row_base[lanec];
    int64_t my_row;
    {
        const int64_t id = ids[wave0 * F + lanec];
        my_row = (lane < F && id >= 0) ? my_base + id : -1;
    }
    for (int64_t b = wave0; b < B; b += nwaves) {
        const int64_t bn = b + nwaves;
        const int64_t next_id = ids[(bn < B ? bn : b) * F + lanec];           // prefetch: consumed at the loop bottom
        float4 S = f4zero(), SS = f4zero();
        float lin = 0.f;
        float* out_row = concat + b * ld;
        for (int f0 = 0; f0 < F; f0 += NS * U) {
            float4 v[U];
            float lw[U];
            bool present[U], counted[U];
            int fc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * NS + slot;
                fc[u] = f < F ? f : F - 1;
                const int64_t r = __shfl(my_row, fc[u], 64);
                present[u] = r >= 0;
                counted[u] = f < F && dvalid;
                const int64_t rc = present[u] ? r : 0;
                v[u] = f4load_nt(table + rc * D + subc * 4);
                lw[u] = lsrc[lin_w != nullptr ? rc : 0];      // (no first-order table: one dummy address, not a random gather)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 x = present[u] ? v[u] : f4zero();
                f4store_nt(out_row + fc[u] * D + subc * 4, x);
                const float4 xs = counted[u] ? x : f4zero();
                S = f4add(S, xs);
                SS = f4fma2(xs, SS);
                lin += (counted[u] && present[u] && use_lin) ? lw[u] : 0.f;
            }
        }
        my_row = (bn < B && lane < F && next_id >= 0) ? my_base + next_id : -1;
        if (sum_x == nullptr && fm_logit == nullptr) continue;
#pragma unroll
        for (int m = LPR; m < DR_WAVE; m <<= 1) {
            S = f4add(S, f4shfl_xor(S, m));
            SS = f4add(SS, f4shfl_xor(SS, m));
            lin += __shfl_xor(lin, m, 64);
        }
        if (sum_x != nullptr && slot == 0 && dvalid) *reinterpret_cast<float4*>(sum_x + b * D + sub * 4) = S;
        if (fm_logit != nullptr) {
            float t = (S.x * S.x - SS.x) + (S.y * S.y - SS.y) + (S.z * S.z - SS.z) + (S.w * S.w - SS.w);
#pragma unroll
            for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m, 64);
            if (lane == 0) fm_logit[b] = (lin_bias != nullptr ? lin_bias[0] : 0.f) + lin + so * t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forward, general path: ragged bags (field f owns columns [col_start[f], col_start[f+1])).
// [TF] B5: drop ids < 0, sum rows in id order, one divide by the count, empty bag -> zeros.
// ------------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void emb_pool_fwd_bag_kernel(
    const int64_t* __restrict__ ids, int64_t B, int32_t F, int32_t C, const int32_t* __restrict__ col_start,
    const int64_t* __restrict__ row_base, const float* __restrict__ table, int32_t D,
    const float* __restrict__ lin_w, const float* __restrict__ lin_bias, float* __restrict__ concat, int64_t ld,
    float* __restrict__ sum_x, float* __restrict__ fm_logit, float so) {
    constexpr int NS = DR_WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPR;
    const int sub = lane % LPR;
    const bool dvalid = sub * 4 < D;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

    for (int64_t b = wave0; b < B; b += nwaves) {
        float4 S = f4zero(), SS = f4zero();
        float lin = 0.f;
        float* out_row = concat + b * ld;
        const int64_t* id_row = ids + b * C;
        for (int f = slot; f < F; f += NS) {
            const int c0 = col_start[f], c1 = col_start[f + 1];
            const int64_t base = row_base[f];
            float4 acc = f4zero();
            int cnt = 0;
            for (int c = c0; c < c1; ++c) {
                const int64_t id = id_row[c];
                if (id >= 0) {
                    const int64_t row = base + id;
                    if (dvalid) acc = f4add(acc, *reinterpret_cast<const float4*>(table + row * D + sub * 4));
                    if (lin_w != nullptr && sub == 0) lin += lin_w[row];   // count vector: duplicates add
                    ++cnt;
                }
            }
            if (cnt > 1) {
                const float n = (float)cnt;
                acc = make_float4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
            }
            if (dvalid) *reinterpret_cast<float4*>(out_row + f * D + sub * 4) = acc;
            S = f4add(S, acc);
            SS = f4fma2(acc, SS);
        }
        if (sum_x == nullptr && fm_logit == nullptr) continue;
#pragma unroll
        for (int m = LPR; m < DR_WAVE; m <<= 1) {
            S = f4add(S, f4shfl_xor(S, m));
            SS = f4add(SS, f4shfl_xor(SS, m));
            lin += __shfl_xor(lin, m, 64);
        }
        if (sum_x != nullptr && slot == 0 && dvalid) *reinterpret_cast<float4*>(sum_x + b * D + sub * 4) = S;
        if (fm_logit != nullptr) {
            float t = (S.x * S.x - SS.x) + (S.y * S.y - SS.y) + (S.z * S.z - SS.z) + (S.w * S.w - SS.w);
#pragma unroll
            for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m, 64);
            if (lane == 0) fm_logit[b] = (lin_bias != nullptr ? lin_bias[0] : 0.f) + lin + so * t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward: g = d_concat + d_fm_logit * (sum_x - x); dst[row] += scale * g / count (hardware fp32
// atomics at L2, fire-and-forget).  Lane -> d mapping is STRIDED (d = j*LPR*... see below) so that each
// atomic instruction of a slot covers one contiguous 4*LPR-byte span of the row (fewer, fuller L2
// atomic requests than a float4-per-lane mapping, where each instruction touches every 4th dword).
// ------------------------------------------------------------------------------------------------
template <int LPR, bool STRIDED>
__global__ __launch_bounds__(256) void emb_pool_bwd_kernel(
    const int64_t* __restrict__ ids, int64_t B, int32_t F, int32_t C, const int32_t* __restrict__ col_start,
    const int64_t* __restrict__ row_base, int32_t D, const float* __restrict__ d_concat, int64_t ld_dc,
    const float* __restrict__ concat, int64_t ld_c, const float* __restrict__ sum_x,
    const float* __restrict__ d_fm_logit, float scale, float* __restrict__ dst_table,
    float* __restrict__ dst_lin, float* __restrict__ dst_bias) {
    constexpr int NS = DR_WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPR;
    const int sub = lane % LPR;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // element offsets handled by this lane within a row
    int doff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) doff[j] = STRIDED ? (j * LPR + sub) : (sub * 4 + j);

    if (blockIdx.x == 0 && dst_bias != nullptr && d_fm_logit != nullptr) dr_block_sum_axpy(d_fm_logit, B, scale, dst_bias);
    for (int64_t b = wave0; b < B; b += nwaves) {
        const float dl = d_fm_logit != nullptr ? d_fm_logit[b] : 0.f;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (d_fm_logit != nullptr && sum_x != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (doff[j] < D) s[j] = sum_x[b * D + doff[j]];
        }
        const int64_t* id_row = ids + b * C;
        for (int f = slot; f < F; f += NS) {
            const int c0 = col_start[f], c1 = col_start[f + 1];
            int cnt = 0;
            for (int c = c0; c < c1; ++c) cnt += id_row[c] >= 0 ? 1 : 0;
            if (cnt == 0) continue;
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = 0.f;
                if (doff[j] < D) {
                    if (d_concat != nullptr) g[j] = d_concat[b * ld_dc + f * D + doff[j]];
                    if (d_fm_logit != nullptr && concat != nullptr) g[j] += dl * (s[j] - concat[b * ld_c + f * D + doff[j]]);
                    g[j] *= scale;
                    if (cnt > 1) g[j] /= (float)cnt;
                }
            }
            const int64_t base = row_base[f];
            for (int c = c0; c < c1; ++c) {
                const int64_t id = id_row[c];
                if (id < 0) continue;
                float* dst = dst_table + (base + id) * D;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (doff[j] < D) unsafeAtomicAdd(dst + doff[j], g[j]);
                if (dst_lin != nullptr && d_fm_logit != nullptr && sub == 0) unsafeAtomicAdd(dst_lin + base + id, scale * dl);
            }
        }
    }
}

int lpr_for(int D) {
    int l = 1;
    while (l * 4 < D) l <<= 1;
    return l;
}

}  // namespace

#define DR_DISPATCH_LPR(lpr, CALL)            \
    switch (lpr) {                            \
        case 1: { CALL(1) } break;            \
        case 2: { CALL(2) } break;            \
        case 4: { CALL(4) } break;            \
        case 8: { CALL(8) } break;            \
        case 16: { CALL(16) } break;          \
        case 32: { CALL(32) } break;          \
        case 64: { CALL(64) } break;          \
        default: return DR_EINVAL;            \
    }

extern "C" int dr_emb_pool_fwd_ex(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                                  const int64_t* row_base, const float* table, int32_t D, const float* lin_w,
                                  const float* lin_bias, float* concat, int64_t ld_concat, float* sum_x, float* fm_logit,
                                  int32_t flags, dr_stream_t stream) {
    const float so = (flags & 1) ? 0.f : 0.5f;          // DR_POOL_FIRST_ORDER_ONLY: fm_logit = bias + sum w (the "wide" logit)
    if (B < 0 || F <= 0 || C < F || D < 4 || D > 256 || (D & 3) || ld_concat < (int64_t)F * D || (ld_concat & 3))
        return DR_EINVAL;
    if (col_start == nullptr && C != F) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!ids || !row_base || !table || !concat) return DR_EINVAL;
    const int lpr = lpr_for(D);
    const int grid = dr_grid_for(B, 4);
    if (col_start == nullptr && F <= 64) {
#define CALL(L)                                                                                                    \
    {                                                                                                              \
        constexpr int NS_ = 64 / L;                                                                                \
        constexpr int U_ = NS_ >= 16 ? 2 : (NS_ >= 4 ? 8 : 4);                                                     \
        hipLaunchKernelGGL((emb_pool_fwd_sv_kernel<L, U_>), dim3(grid), dim3(256), 0, dr_s(stream), ids, B, F,     \
                           row_base, table, D, lin_w, lin_bias, concat, ld_concat, sum_x, fm_logit, so);           \
    }
        DR_DISPATCH_LPR(lpr, CALL)
#undef CALL
    } else {
        if (col_start == nullptr) return DR_EINVAL;   // F > 64 single-valued: pass an explicit col_start
#define CALL(L)                                                                                                    \
    hipLaunchKernelGGL((emb_pool_fwd_bag_kernel<L>), dim3(grid), dim3(256), 0, dr_s(stream), ids, B, F, C,         \
                       col_start, row_base, table, D, lin_w, lin_bias, concat, ld_concat, sum_x, fm_logit, so);
        DR_DISPATCH_LPR(lpr, CALL)
#undef CALL
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_emb_pool_fwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                               const int64_t* row_base, const float* table, int32_t D, const float* lin_w,
                               const float* lin_bias, float* concat, int64_t ld_concat, float* sum_x, float* fm_logit,
                               dr_stream_t stream) {
    return dr_emb_pool_fwd_ex(ids, B, F, C, col_start, row_base, table, D, lin_w, lin_bias, concat, ld_concat, sum_x, fm_logit,
                              0, stream);
}

// ---- per-field first-order outputs (FNN, estimator/models/ranking/fnn.py:53-64 of the reference: one Dense(1, no bias)
// over each indicator column's multi-hot input, then tf.concat -> [B, F]):  out[b][f] = sum_{c in bag f} w[row_base[f] + id]
namespace {
__global__ __launch_bounds__(256) void lin_fields_fwd_kernel(const int64_t* __restrict__ ids, int64_t B, int32_t F, int32_t C,
                                                             const int32_t* __restrict__ col_start,
                                                             const int64_t* __restrict__ row_base,
                                                             const float* __restrict__ lin_w, float* __restrict__ out,
                                                             int64_t ld) {
    const int64_t n = B * F, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int c0 = col_start != nullptr ? col_start[f] : f, c1 = col_start != nullptr ? col_start[f + 1] : f + 1;
        const int64_t base = row_base[f];
        float acc = 0.f;
        for (int c = c0; c < c1; ++c) {
            const int64_t id = ids[b * C + c];
            if (id >= 0) acc += lin_w[base + id];
        }
        out[b * ld + f] = acc;
    }
}
__global__ __launch_bounds__(256) void lin_fields_bwd_kernel(const int64_t* __restrict__ ids, int64_t B, int32_t F, int32_t C,
                                                             const int32_t* __restrict__ col_start,
                                                             const int64_t* __restrict__ row_base,
                                                             const float* __restrict__ d_out, int64_t ld, float scale,
                                                             float* __restrict__ dst_lin) {
    const int64_t n = B * F, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int c0 = col_start != nullptr ? col_start[f] : f, c1 = col_start != nullptr ? col_start[f + 1] : f + 1;
        const int64_t base = row_base[f];
        const float g = scale * d_out[b * ld + f];
        for (int c = c0; c < c1; ++c) {
            const int64_t id = ids[b * C + c];
            if (id >= 0) unsafeAtomicAdd(dst_lin + base + id, g);
        }
    }
}
}  // namespace

extern "C" int dr_lin_fields_fwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                                 const int64_t* row_base, const float* lin_w, float* out, int64_t ld_out, dr_stream_t stream) {
    if (B < 0 || F <= 0 || C < F || ld_out < F) return DR_EINVAL;
    if (col_start == nullptr && C != F) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!ids || !row_base || !lin_w || !out) return DR_EINVAL;
    hipLaunchKernelGGL(lin_fields_fwd_kernel, dim3(dr_grid_for(B * F, 256)), dim3(256), 0, dr_s(stream), ids, B, F, C, col_start,
                       row_base, lin_w, out, ld_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_lin_fields_bwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                                 const int64_t* row_base, const float* d_out, int64_t ld_dout, float scale, float* dst_lin,
                                 dr_stream_t stream) {
    if (B < 0 || F <= 0 || C < F || ld_dout < F) return DR_EINVAL;
    if (col_start == nullptr && C != F) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!ids || !row_base || !d_out || !dst_lin) return DR_EINVAL;
    hipLaunchKernelGGL(lin_fields_bwd_kernel, dim3(dr_grid_for(B * F, 256)), dim3(256), 0, dr_s(stream), ids, B, F, C, col_start,
                       row_base, d_out, ld_dout, scale, dst_lin);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_emb_pool_bwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                               const int64_t* row_base, int32_t D, const float* d_concat, int64_t ld_dconcat,
                               const float* concat, int64_t ld_concat, const float* sum_x,
                               const float* d_fm_logit, float scale, float* dst_table, float* dst_lin,
                               float* dst_bias, dr_stream_t stream) {
    if (B < 0 || F <= 0 || C < F || D < 4 || D > 256 || (D & 3)) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!ids || !row_base || !dst_table || !col_start) return DR_EINVAL;
    if (d_concat == nullptr && d_fm_logit == nullptr) return DR_EINVAL;
    if (d_concat != nullptr && ld_dconcat < (int64_t)F * D) return DR_EINVAL;
    // d_fm_logit with concat == sum_x == NULL: the logit was first-order only (DR_POOL_FIRST_ORDER_ONLY): its gradient
    // reaches dst_lin / dst_bias, not the rows
    if (d_fm_logit != nullptr && (concat == nullptr) != (sum_x == nullptr)) return DR_EINVAL;
    if (d_fm_logit != nullptr && concat != nullptr && ld_concat < (int64_t)F * D) return DR_EINVAL;
    const int lpr = lpr_for(D);
    const int grid = dr_grid_for(B, 4);
    // lane -> element map strided (each atomic instruction covers a contiguous 64-byte span): 3.2x faster than a float4 per lane
#define CALL(L)                                                                                                    \
    hipLaunchKernelGGL((emb_pool_bwd_kernel<L, true>), dim3(grid), dim3(256), 0, dr_s(stream), ids, B, F, C,       \
                       col_start, row_base, D, d_concat, ld_dconcat, concat, ld_concat, sum_x, d_fm_logit, scale,   \
                       dst_table, dst_lin, dst_bias);
    DR_DISPATCH_LPR(lpr, CALL)
#undef CALL
    DR_CHECK_LAUNCH();
    return DR_OK;
}
