// Fused backward of a NARROW dense layer  y = act(x W + b),  x:[M,K]  W:[K,N]  N <= 32  (the last hidden layer of the
// reference's DNN towers: Dense(32) in keras/models/ranking/deepfm.py:30-34 / estimator/models/feature_interaction/dnn.py:17-29
// of the reference, followed by Dense(1)).
//
//   dx[m][k]  = (sum_n dy[m][n] W[k][n]) * (x[m][k] > 0 if relu_mask)          -- gradient for the layer below
//   W[k][n]  += scale * sum_m x[m][k] dy[m][n]                                  -- fused SGD step (scale = -lr)
//   b[n]     += scale * sum_m dy[m][n]
//
// The generic path runs this as two GEMM launches plus a split-K reduce (60 + 80 us at M=65536, K=256, N=32 on
// MI355X, both far from any roofline: the shapes have one k-tile (dx) or two output tiles (dW)).  Here x is read from
// HBM exactly once: a block walks 32-row chunks; each wave owns K/4 columns of x for BOTH products, so x never touches
// LDS -- a lane's 16 x-values per column are at once (a) the A operand of the dW MFMAs and (b) the ReLU mask of the
// dx tile it just computed, because the reduction slot -> row assignment of the dW product is chosen to coincide with
// the MFMA accumulator row layout  row(j, lane) = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5).
// Only the 32 x N chunk of dy goes through LDS (double-buffered, one barrier per chunk).  Every global load/store in
// the loop is unconditional (clamped addresses): conditional memory ops get s_cbranch_execz and a pessimistic
// s_waitcnt vmcnt(0) each (see emb_pool.hip).
//
// Bounded by HBM: algorithmic bytes = 4 M (2 K + N).
#include "dr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TT_ROWS = 32;     // rows per chunk (one MFMA tile)
constexpr int TT_P = 33;        // LDS pitch of the dy chunk

template <int KT> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };

template <int KT> __device__ __forceinline__ float vget(const typename VecT<KT>::type& v, int i);
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float vget<2>(const float2& v, int i) { return i == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float vget<4>(const float4& v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}
template <int KT> __device__ __forceinline__ void vset(typename VecT<KT>::type& v, int i, float x);
template <> __device__ __forceinline__ void vset<1>(float& v, int, float x) { v = x; }
template <> __device__ __forceinline__ void vset<2>(float2& v, int i, float x) { if (i == 0) v.x = x; else v.y = x; }
template <> __device__ __forceinline__ void vset<4>(float4& v, int i, float x) {
    if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x;
}

__device__ __forceinline__ int tt_row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

// KT = K / 128: every wave owns 32*KT columns; lane c holds the KT adjacent columns  col0 + KT*c + t  (one KT-wide
// vector load per row), so "tile t" is the column set { col0 + KT*c + t : c < 32 }.
template <int KT>
__global__ __launch_bounds__(256, 2) void linear_bwd_narrow_kernel(const float* __restrict__ x, int64_t ldx,
                                                                   const float* __restrict__ dy, int64_t lddy,
                                                                   const float* __restrict__ W, int64_t ldw, int64_t M,
                                                                   int32_t K, int32_t N, int32_t relu_mask,
                                                                   float* __restrict__ dx, int64_t lddx,
                                                                   float* __restrict__ partial, uint32_t* __restrict__ dx_amax) {
    typedef typename VecT<KT>::type vec_t;
    __shared__ float dys[2][TT_ROWS * TT_P];
    __shared__ uint32_t amax_w[4];
    float dx_max = 0.f;                                       // largest |dx| this lane stored (dx_amax != NULL: the f16x2 GEMMs' record)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int col0 = wave * 32 * KT + KT * c;                 // first of this lane's KT columns
    const int64_t chunks = M / TT_ROWS;

    // W fragments (B operand of the dx product): wf[t][s] = W[col0 + t][2 s + h], zero past N
    float wf[KT][16];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int n = 2 * s + h;
            const float v = W[(int64_t)(col0 + t) * ldw + (n < N ? n : N - 1)];
            wf[t][s] = n < N ? v : 0.f;
        }

    f32x16 accw[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) accw[t][j] = 0.f;
    float bias_acc = 0.f;

    // staging coordinates of the dy chunk: thread -> (row = tid >> 3, 4 columns from 4 * (tid & 7))
    const int sr = tid >> 3, sc = (tid & 7) * 4;
    auto load_dy = [&](float (&r)[4], int64_t chunk) {
        const float* p = dy + (chunk * TT_ROWS + sr) * lddy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = sc + j;
            const float v = p[n < N ? n : N - 1];
            r[j] = n < N ? v : 0.f;
        }
    };
    // row addresses = (uniform base of row (j & 3) + 8 * (j >> 2) of the chunk, in SGPRs) + ONE 32-bit per-lane offset
    const unsigned x_lane = (unsigned)(4 * h * ldx + col0), dx_lane = (unsigned)(4 * h * lddx + col0);
    auto load_x = [&](vec_t (&v)[16], int64_t chunk) {
        const float* base = x + chunk * TT_ROWS * ldx;                       // wave-uniform
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = *reinterpret_cast<const vec_t*>(base + (int64_t)tt_row(s, 0) * ldx + x_lane);
    };

    int64_t chunk = blockIdx.x;
    if (chunk >= chunks) chunk = chunks - 1;                  // surplus blocks redo the last chunk with zero weight
    const bool live_block = (int64_t)blockIdx.x < chunks;
    vec_t xv[16], xn[16];
    float dyr[4];
    load_dy(dyr, chunk);
    load_x(xv, chunk);
#pragma unroll
    for (int j = 0; j < 4; ++j) dys[0][sr * TT_P + sc + j] = live_block ? dyr[j] : 0.f;
    __syncthreads();
    int p = 0;
    for (; chunk < chunks; chunk += gridDim.x) {
        const int64_t next = chunk + gridDim.x;
        const bool has_next = next < chunks;
        const int64_t nc = has_next ? next : chunk;
        load_dy(dyr, nc);                                     // prefetch (clamped: the last iteration re-reads its own chunk)
        load_x(xn, nc);
        __builtin_amdgcn_sched_barrier(0);

        const float* cur = dys[p];
        float dxa[16], dwb[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            dxa[s] = cur[c * TT_P + 2 * s + h];               // dy[m = c][n = 2 s + h]
            dwb[s] = cur[tt_row(s, h) * TT_P + c];            // dy[m = row(s, h)][n = c]
        }
        if (wave == 0 && h == 0) {
#pragma unroll 8
            for (int r = 0; r < TT_ROWS; ++r) bias_acc += cur[r * TT_P + c];
        }
        // ---- dW tiles first (they consume xv as the A operand): accw[t][k][n] += sum_m x[m][k] dy[m][n] ------------
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget<KT>(xv[s], t), dwb[s], accw[t], 0, 0, 0);
        // ---- dx tiles; the masked result overwrites the x value it was masked with ---------------------------------
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dxa[s], wf[t][s], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = acc[j];
                if (relu_mask && !(vget<KT>(xv[j], t) > 0.f)) v = 0.f;
                dx_max = fmaxf(dx_max, fabsf(v));
                vset<KT>(xv[j], t, v);
            }
        }
        // ---- next dy chunk into the other LDS buffer, then the dx stores ------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) dys[p ^ 1][sr * TT_P + sc + j] = has_next ? dyr[j] : 0.f;
        {
            float* base = dx + chunk * TT_ROWS * lddx;                        // wave-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j)
                *reinterpret_cast<vec_t*>(base + (int64_t)tt_row(j, 0) * lddx + dx_lane) = xv[j];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) xv[s] = xn[s];
        p ^= 1;
    }
    // ---- partial results: partial[block][k][n] (k < K), row K = bias column sums -----------------------------------
    float* pp = partial + (int64_t)blockIdx.x * (K + 1) * 32;
    const float live = live_block ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = wave * 32 * KT + KT * tt_row(j, h) + t;      // accumulator row i <-> column col(i, t)
            pp[(int64_t)k * 32 + c] = live * accw[t][j];
        }
    if (wave == 0 && h == 0) pp[(int64_t)K * 32 + c] = live * bias_acc;
    if (dx_amax != nullptr) {                                 // (kernel-uniform) one atomic per block, and only if it would raise the record
        uint32_t m = live_block ? __float_as_uint(dx_max) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0) amax_w[wave] = m;
        __syncthreads();
        if (tid == 0) {
            m = max(max(amax_w[0], amax_w[1]), max(amax_w[2], amax_w[3]));
            if (m > __hip_atomic_load(dx_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dx_amax, m);
        }
    }
}

// dstW[k][n] += scale * sum_p partial[p][k][n] ; k == K -> dstb[n].  One block per k, fixed summation order.
__global__ __launch_bounds__(256) void linear_bwd_narrow_reduce_kernel(const float* __restrict__ partial, int32_t nparts,
                                                                       int32_t K, int32_t N, float scale,
                                                                       float* __restrict__ dstW, int64_t ldw,
                                                                       float* __restrict__ dstb) {
    __shared__ float red[8][32];
    const int k = blockIdx.x, n = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int64_t stride = (int64_t)(K + 1) * 32;
    const float* p = partial + (int64_t)k * 32 + n;
    float acc = 0.f;
    for (int q0 = pg; q0 < nparts; q0 += 8 * 8) {            // 8 loads in flight, not a chain of dependent round trips
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + 8 * u;
            v[u] = p[(int64_t)(q < nparts ? q : q0) * stride];
            if (q >= nparts) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    red[pg][n] = acc;
    __syncthreads();
    if (pg == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][n];
        if (k < K) dstW[(int64_t)k * ldw + n] = fmaf(scale, s, dstW[(int64_t)k * ldw + n]);
        else if (dstb != nullptr) dstb[n] = fmaf(scale, s, dstb[n]);
    }
}

int tt_grid(int64_t M) {
    const int64_t chunks = M / TT_ROWS;
    return (int)(chunks < 512 ? chunks : 512);                 // 2 resident blocks per CU
}

}  // namespace

extern "C" int64_t dr_linear_bwd_narrow_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    (void)N;
    if (M < TT_ROWS) return 256;
    return (int64_t)tt_grid(M) * (K + 1) * 32 * (int64_t)sizeof(float);
}

// Returns DR_ESHAPE when the shape is outside the fused kernel's domain (the caller then uses dr_linear_bwd_dx +
// dr_linear_bwd_dw): needs N <= 32, K in {128, 256, 512}, M a positive multiple of 32, 4*KT-byte aligned rows.
static int bwd_narrow_impl(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W,
                           int64_t ld_w, int64_t M, int32_t K, int32_t N, int32_t relu_mask, float scale,
                           float* dstW, int64_t ld_dstw, float* dstb, float* dx, int64_t ld_dx, void* workspace,
                           int64_t workspace_bytes, int32_t parts, dr_stream_t stream, uint32_t* dx_amax = nullptr) {
    if (M <= 0 || K <= 0 || N <= 0) return DR_EINVAL;
    if (!x || !dy || !W || !dstW || !dx || !workspace) return DR_EINVAL;
    if (N > 32 || (K != 128 && K != 256 && K != 512) || (M % TT_ROWS) != 0) return DR_ESHAPE;
    const int kt = K / 128;
    const uintptr_t al = (uintptr_t)(4 * kt - 1);
    if ((reinterpret_cast<uintptr_t>(x) & al) || (reinterpret_cast<uintptr_t>(dx) & al) || ((ld_x * 4) & al) ||
        ((ld_dx * 4) & al))
        return DR_ESHAPE;
    if (ld_x < K || ld_dx < K || ld_dy < N || ld_w < N || ld_dstw < N) return DR_EINVAL;
    if (workspace_bytes < dr_linear_bwd_narrow_workspace_bytes(M, K, N)) return DR_EINVAL;
    const int grid = tt_grid(M);
    float* partial = static_cast<float*>(workspace);
#define TT_CALL(KT)                                                                                                     \
    hipLaunchKernelGGL((linear_bwd_narrow_kernel<KT>), dim3(grid), dim3(256), 0, dr_s(stream), x, ld_x, dy, ld_dy, W,   \
                       ld_w, M, K, N, relu_mask, dx, ld_dx, partial, dx_amax)
    if (parts & 1) {
        if (dx_amax != nullptr && hipMemsetAsync(dx_amax, 0, sizeof(uint32_t), dr_s(stream)) != hipSuccess) return DR_ELAUNCH;
        if (kt == 1) TT_CALL(1);
        else if (kt == 2) TT_CALL(2);
        else TT_CALL(4);
    }
#undef TT_CALL
    if (parts & 2)
        hipLaunchKernelGGL(linear_bwd_narrow_reduce_kernel, dim3(K + 1), dim3(256), 0, dr_s(stream), partial, grid, K, N, scale,
                           dstW, ld_dstw, dstb);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_linear_bwd_narrow(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W,
                                    int64_t ld_w, int64_t M, int32_t K, int32_t N, int32_t relu_mask, float scale,
                                    float* dstW, int64_t ld_dstw, float* dstb, float* dx, int64_t ld_dx, void* workspace,
                                    int64_t workspace_bytes, dr_stream_t stream) {
    return bwd_narrow_impl(x, ld_x, dy, ld_dy, W, ld_w, M, K, N, relu_mask, scale, dstW, ld_dstw, dstb, dx, ld_dx, workspace,
                           workspace_bytes, 3, stream);
}

// In two halves: parts = 1 the one-pass kernel (dx and the per-block partials of dW / db), parts = 2 the reduce that applies them
// (dstW += scale * sum, dstb likewise), 3 = both.  Part 2 may run on another stream; it must finish before anything reads dstW / dstb
// and before the next part 1 over the same workspace.
extern "C" int dr_linear_bwd_narrow_parts(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W,
                                          int64_t ld_w, int64_t M, int32_t K, int32_t N, int32_t relu_mask, float scale,
                                          float* dstW, int64_t ld_dstw, float* dstb, float* dx, int64_t ld_dx, void* workspace,
                                          int64_t workspace_bytes, int32_t parts, dr_stream_t stream) {
    if (parts < 1 || parts > 3) return DR_EINVAL;
    return bwd_narrow_impl(x, ld_x, dy, ld_dy, W, ld_w, M, K, N, relu_mask, scale, dstW, ld_dstw, dstb, dx, ld_dx, workspace,
                           workspace_bytes, parts, stream);
}

// dr_linear_bwd_narrow_parts that also leaves max |dx| (float bits) in dx_amax[0] -- the amax record the f16x2 GEMMs (dr_h2_linear_nt,
// dr_h2_wgrad_emb) want for dx as their operand; the record is reset and rebuilt by part 1.
extern "C" int dr_linear_bwd_narrow_amax(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W,
                                         int64_t ld_w, int64_t M, int32_t K, int32_t N, int32_t relu_mask, float scale,
                                         float* dstW, int64_t ld_dstw, float* dstb, float* dx, int64_t ld_dx, void* workspace,
                                         int64_t workspace_bytes, int32_t parts, uint32_t* dx_amax, dr_stream_t stream) {
    if (parts < 1 || parts > 3 || !dx_amax) return DR_EINVAL;
    return bwd_narrow_impl(x, ld_x, dy, ld_dy, W, ld_w, M, K, N, relu_mask, scale, dstW, ld_dstw, dstb, dx, ld_dx, workspace,
                           workspace_bytes, parts, stream, dx_amax);
}
