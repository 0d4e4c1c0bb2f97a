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

// s_barrier behind the wave's own LDS traffic only.  __syncthreads() also waits vmcnt(0): with it every barrier of the loop below would
// sit out the NEXT chunk's prefetch (and this chunk's dx stores) -- three times per chunk.  The hazards the barriers order are all
// LDS ones; registers loaded from HBM are waited for by the compiler where they are used.
__device__ __forceinline__ void tail_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
    tail_lds_barrier();
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
        tail_lds_barrier();
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

// =====================================================================================================================
// The whole tower tail in ONE pass over x = h0 (round 5): dr_tower_head_fwd_bwd followed by
// dr_linear_bwd_narrow read the same [M, K] activations twice from HBM (67 MB each at config 3) in two latency-bound launches
// (51 + 60 us at 0.19 / 0.29 of the HBM roofline).  Here a block walks 32-row chunks like the narrow kernel above and, per chunk,
//   (1) stages the chunk of x through the LDS in row-major order and multiplies it with W1 on the fp32 MFMA -- every wave its
//       K / 4 columns: lane (row m, half h) reads 16 KT consecutive floats of row m as the A operand of 16 KT
//       v_mfma_f32_32x32x2_f32, W1's rows sit in registers as the B operand -- and the four partial [32, 32] tiles are summed
//       through the LDS by wave 0;
//   (2) wave 0 runs the head epilogue of dense.hip (bias, ReLU, Dense(1) as a 32-lane butterfly, + extra logit, BCE terms,
//       d logit, d h = d logit (x) w2 * ReLU') and leaves d h in the LDS where the narrow backward expects its dy chunk;
//   (3) all four waves run the narrow backward of the kernel above on the x values they still hold in registers.
// x is read once, d x written once; prob / d_logit / (optionally) d_h go out as before; the Dense(1) gradient, its bias gradient
// and the loss leave as per-block partials in dr_tower_head's layout ([block][34]), the narrow layer's as in the kernel above.
// K in {128, 256} (the staged chunk is 32 x (K + 4) floats: 33 KB at K = 256, two blocks per CU).
// =====================================================================================================================
// experiment switches of the one-pass tail (tools/exp/tail_variants.sh builds one library per value; 0 = shipped):
//   1 two blocks per CU (launch bounds 512 x 2: <= 128 VGPRs)    2 no register prefetch of the next chunk (loaded behind the dx stores)
//   16 no head MFMAs   32 no epilogue math   64 no narrow-backward MFMAs   128 no dx stores   (16..128: timing only, wrong results)
#ifndef DR_TAIL_DBG
#define DR_TAIL_DBG 0
#endif
constexpr int TAIL_HEAD_PART = 34;          // == HEAD_PART of dense.hip: dw2[32], db2, loss

struct TailArgs {
    const float* x; int64_t ldx;
    const float* W1; int64_t ldw1; const float* b1;
    int64_t M; int32_t K; int32_t N;             // N = H <= 32
    const float* w2; int64_t ld_w2; const float* b2;
    const float* extra; const float* labels; int32_t loss_mode; float inv_n;
    float* prob; float* d_logit; float* d_h; int64_t ld_dh;      // d_h may be null
    float* dx; int64_t lddx;
    float* partial;                              // narrow layer: [grid][(K + 1) * 32]
    float* head_partial;                         // Dense(1) + loss: [grid][34]
    uint32_t* dx_amax;                           // (may be null) record raised with atomicMax -- only when amax_part is null
    uint32_t* amax_part;                         // (may be null) [grid]: every block's max |dx| as float bits, reduced by tower_tail_reduce_kernel
};

// NWV waves per block, wave w owns columns 32 w .. 32 w + 31 of x (K = 32 NWV): lane (c, h) keeps x[tt_row(s, h)][32 w + c], s < 16.
template <int NWV>
__global__ __launch_bounds__(64 * NWV, (DR_TAIL_DBG & 1) ? 4 : 2) void tower_tail_fused_kernel(TailArgs a) {   // (HIP: second argument = min waves per SIMD)
    constexpr int K = 32 * NWV, XP = K + 4;                   // staged chunk: [32][XP] floats (pitch = 4 mod 32: b128 row reads conflict-free)
    constexpr int RPW = 16 / NWV;                             // accumulator registers (= 2 rows each) of the head tile a wave finishes
    __shared__ __attribute__((aligned(16))) float xs[TT_ROWS * XP];
    __shared__ float red[NWV][TT_ROWS * TT_P];
    __shared__ float dys[TT_ROWS * TT_P];
    __shared__ float hsum[NWV][TAIL_HEAD_PART];
    __shared__ uint32_t amax_w[NWV];
    const float* __restrict__ x = a.x;
    const int64_t ldx = a.ldx, lddx = a.lddx;
    const int N = a.N;
    float dx_max = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int col0 = wave * 32 + c;                           // this lane's column (narrow-backward layout)
    const int64_t chunks = a.M / TT_ROWS;

    // B operand of the dx product (as in linear_bwd_narrow_kernel): wf[s] = W1[col0][2 s + h], zero past N
    float wf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int n = 2 * s + h;
        const float v = a.W1[(int64_t)col0 * a.ldw1 + (n < N ? n : N - 1)];
        wf[s] = n < N ? v : 0.f;
    }
    // B operand of the HEAD product: k-step j of this wave multiplies column cb + j (cb = 32 wave + 16 h), output column n = c
    const int cb = wave * 32 + 16 * h;
    float w1f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float v = a.W1[(int64_t)(cb + j) * a.ldw1 + (c < N ? c : N - 1)];
        w1f[j] = c < N ? v : 0.f;
    }
    const bool cv = c < N;
    const float b1j = a.b1 != nullptr ? a.b1[cv ? c : N - 1] : 0.f;
    const float w2j = cv ? a.w2[(int64_t)c * a.ld_w2] : 0.f;
    const float b2v = a.b2 != nullptr ? a.b2[0] : 0.f;
    float dw2_acc = 0.f, db2_acc = 0.f, loss_acc = 0.f;

    f32x16 accw;
#pragma unroll
    for (int j = 0; j < 16; ++j) accw[j] = 0.f;
    float bias_acc = 0.f;

    const unsigned x_lane = (unsigned)(4 * h * ldx + col0), dx_lane = (unsigned)(4 * h * lddx + col0);
    auto load_x = [&](float (&v)[16], int64_t chunk) {
        const float* base = x + chunk * TT_ROWS * ldx;                       // wave-uniform
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = base[(int64_t)tt_row(s, 0) * ldx + x_lane];
    };

    int64_t chunk = blockIdx.x;
    if (chunk >= chunks) chunk = chunks - 1;                  // surplus blocks redo the last chunk with zero weight, no stores
    const bool live_block = (int64_t)blockIdx.x < chunks;
    float xv[16], xn[16];
    load_x(xv, chunk);
    for (; chunk < chunks; chunk += gridDim.x) {
        const int64_t next = chunk + gridDim.x;
        if constexpr (!(DR_TAIL_DBG & 2)) load_x(xn, next < chunks ? next : chunk);             // prefetch (clamped)
        // labels / extra logit of the rows this wave's share of the head epilogue covers: loaded HERE, two barriers ahead of their use
        float labv[RPW], extv[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int reg = wave * RPW + rr;
            const int64_t row = chunk * TT_ROWS + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            labv[rr] = a.labels[row];
            extv[rr] = a.extra != nullptr ? a.extra[row] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- (1) the chunk of x, row-major, into the LDS; every wave multiplies its 32 columns with W1 -------------------------------
#pragma unroll
        for (int s = 0; s < 16; ++s) xs[tt_row(s, h) * XP + col0] = xv[s];
        tail_lds_barrier();
        {
            f32x16 acc1;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[j] = 0.f;
            const float* xr = &xs[c * XP + cb];               // row m = c, this wave-half's 16 columns
#pragma unroll
            for (int q = 0; q < ((DR_TAIL_DBG & 16) ? 1 : 4); ++q) {
                const float4 v = *reinterpret_cast<const float4*>(xr + 4 * q);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, w1f[4 * q + 0], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, w1f[4 * q + 1], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, w1f[4 * q + 2], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, w1f[4 * q + 3], acc1, 0, 0, 0);
            }
            // this wave's partial [32 rows][32 n] -> LDS (C/D layout: column n = c, row = tt_row(reg, h))
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) red[wave][tt_row(reg, h) * TT_P + c] = acc1[reg];
        }
        tail_lds_barrier();
        // ---- (2) head epilogue (dense.hip EPI_HEAD) of the rows of accumulator registers RPW wave .. + RPW - 1; d h -> dys ------------
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int reg = wave * RPW + rr;                  // (wave-uniform)
            const int ro = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int64_t row = chunk * TT_ROWS + ro;
            const int li = ro * TT_P + c;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) v += red[w][li];   // fixed order
            v = fmaxf(v + b1j, 0.f);
            if (!cv) v = 0.f;
            float dot = v * w2j;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) dot += __shfl_xor(dot, o, 64);
            const float lg = (dot + b2v) + extv[rr];
            float p, l, gr;
            if constexpr (DR_TAIL_DBG & 32) { p = lg; l = lg; gr = lg - labv[rr]; }
            else dr_bce_terms(lg, labv[rr], a.loss_mode, p, l, gr);
            float gs = gr * a.inv_n;
            if (!live_block) { l = 0.f; gs = 0.f; }
            const float dh = !(v > 0.f) ? 0.f : gs * w2j;
            dys[li] = dh;
            if (live_block) {
                if (c == 0) {
                    if (a.prob != nullptr) a.prob[row] = p;
                    if (a.d_logit != nullptr) a.d_logit[row] = gs;
                }
                if (cv && a.d_h != nullptr) a.d_h[row * a.ld_dh + c] = dh;
            }
            dw2_acc = fmaf(v, gs, dw2_acc);
            if (c == 0) { db2_acc += gs; loss_acc += l; }
        }
        tail_lds_barrier();
        // ---- (3) the narrow backward of this chunk (linear_bwd_narrow_kernel's body, one column per lane) ----------------------------
        float dxa[16], dwb[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            dxa[s] = dys[c * TT_P + 2 * s + h];               // dy[m = c][n = 2 s + h]
            dwb[s] = dys[tt_row(s, h) * TT_P + c];            // dy[m = row(s, h)][n = c]
        }
        if (wave == 0 && h == 0) {
#pragma unroll 8
            for (int r = 0; r < TT_ROWS; ++r) bias_acc += dys[r * TT_P + c];
        }
#pragma unroll
        for (int s = 0; s < ((DR_TAIL_DBG & 64) ? 1 : 16); ++s) accw = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s], dwb[s], accw, 0, 0, 0);
        {
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
            for (int s = 0; s < ((DR_TAIL_DBG & 64) ? 1 : 16); ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dxa[s], wf[s], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = acc[j];
                if (!(xv[j] > 0.f)) v = 0.f;
                dx_max = fmaxf(dx_max, fabsf(v));
                xv[j] = v;
            }
        }
        if (live_block && (!(DR_TAIL_DBG & 128) || chunk < 0)) {
            float* base = a.dx + chunk * TT_ROWS * lddx;                      // wave-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) base[(int64_t)tt_row(j, 0) * lddx + dx_lane] = xv[j];
        }
        if constexpr (DR_TAIL_DBG & 2) {
            load_x(xv, next < chunks ? next : chunk);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) xv[s] = xn[s];
        }
        // (the next iteration's first barrier -- behind its xs writes -- also orders this iteration's dys / red reads before their rewrite)
    }
    // ---- partial results -------------------------------------------------------------------------------------------------------------
    float* pp = a.partial + (int64_t)blockIdx.x * (K + 1) * 32;
    const float live = live_block ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) pp[(int64_t)(wave * 32 + tt_row(j, h)) * 32 + c] = live * accw[j];   // accumulator row i <-> column 32 wave + i
    if (wave == 0 && h == 0) pp[(int64_t)K * 32 + c] = live * bias_acc;
    dw2_acc += __shfl_xor(dw2_acc, 32, 64);
    db2_acc += __shfl_xor(db2_acc, 32, 64);
    loss_acc += __shfl_xor(loss_acc, 32, 64);
    if (h == 0) {
        hsum[wave][c] = dw2_acc;
        if (c == 0) { hsum[wave][32] = db2_acc; hsum[wave][33] = loss_acc; }
    }
    uint32_t m = live_block ? __float_as_uint(dx_max) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if (lane == 0) amax_w[wave] = m;
    tail_lds_barrier();
    if (tid < TAIL_HEAD_PART) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) sacc += hsum[w][tid];
        a.head_partial[(int64_t)blockIdx.x * TAIL_HEAD_PART + tid] = live * sacc;
    }
    if (tid == 0 && (a.dx_amax != nullptr || a.amax_part != nullptr)) {
        uint32_t mm = 0u;
#pragma unroll
        for (int w = 0; w < NWV; ++w) mm = max(mm, amax_w[w]);
        if (a.amax_part != nullptr) a.amax_part[blockIdx.x] = mm;                       // reduced (and the record STORED) by the reduce launch
        else if (mm > __hip_atomic_load(a.dx_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.dx_amax, mm);
    }
}

// Both fixed-order reduces of the one-pass tail in ONE launch: blocks 0 .. K apply the narrow layer's partials (as
// linear_bwd_narrow_reduce_kernel), block K + 1 the head's (as head_finish_kernel of dense.hip: dst_w2 / dst_b2 / loss).
__global__ __launch_bounds__(256) void tower_tail_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ head_partial,
                                                                int32_t nparts, int32_t K, int32_t N, float scale, float inv_n,
                                                                float* __restrict__ dstW, int64_t ldw, float* __restrict__ dstb,
                                                                float* dst_w2, int64_t ld_w2, float* dst_b2, float* __restrict__ loss_out,
                                                                const uint32_t* __restrict__ amax_part, uint32_t* __restrict__ dx_amax) {
    __shared__ float red[8][TAIL_HEAD_PART];
    __shared__ uint32_t amx[4];
    if ((int)blockIdx.x <= K) {
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
        return;
    }
    constexpr int NG = 7;
    const int grp = threadIdx.x / TAIL_HEAD_PART, c = threadIdx.x % TAIL_HEAD_PART;
    if (grp < NG) {
        float acc = 0.f;
        for (int b0 = grp; b0 < nparts; b0 += NG * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * NG;
                v[u] = head_partial[(int64_t)(b < nparts ? b : b0) * TAIL_HEAD_PART + c];
                if (b >= nparts) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        red[grp][c] = acc;
    }
    if (amax_part != nullptr) {                                         // (kernel-uniform) the record of dx: max over the blocks' maxima, STORED
        uint32_t m = 0u;
        for (int b = threadIdx.x; b < nparts; b += 256) m = max(m, amax_part[b]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0) amx[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (amax_part != nullptr && threadIdx.x == 0) dx_amax[0] = max(max(amx[0], amx[1]), max(amx[2], amx[3]));
    if (threadIdx.x < TAIL_HEAD_PART) {
        float sacc = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) sacc += red[g2][c];
        if (c < 32) {
            if (c < N && dst_w2 != nullptr && scale != 0.f) dst_w2[(int64_t)c * ld_w2] = fmaf(scale, sacc, dst_w2[(int64_t)c * ld_w2]);
        } else if (c == 32) {
            if (dst_b2 != nullptr && scale != 0.f) dst_b2[0] = fmaf(scale, sacc, dst_b2[0]);
        } else if (loss_out != nullptr) {
            loss_out[0] = sacc * inv_n;
        }
    }
}

int tail_grid(int64_t M, int K) {
    const int64_t chunks = M / TT_ROWS;
    const int64_t cap = (K == 256 && !(DR_TAIL_DBG & 1)) ? 256 : 512;                 // 8-wave blocks: one per CU; 4-wave blocks: two
    return (int)(chunks < cap ? chunks : cap);
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

extern "C" int64_t dr_tower_tail_workspace_bytes(int64_t M, int32_t K) {
    if (M < TT_ROWS) return 512;
    return (int64_t)tail_grid(M, K) * ((int64_t)(K + 1) * 32 + TAIL_HEAD_PART + 1) * (int64_t)sizeof(float);
}

// dr_tower_head_fwd_bwd (act = relu) followed by dr_linear_bwd_narrow (relu_mask = 1) of the SAME layer W1 -- the last hidden layer
// of the tower is both the head's first factor and the narrow backward's layer -- in one pass over x (tower_tail_fused_kernel).
extern "C" int dr_tower_tail_fused(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1, int64_t M,
                                   int64_t n_total, int32_t K, int32_t H, const float* w2, int64_t ld_w2, const float* b2,
                                   const float* extra_logit, const float* labels, int32_t loss_mode, float scale, float* dst_w1,
                                   int64_t ld_dst_w1, float* dst_b1, float* dst_w2, int64_t ld_dst_w2, float* dst_b2, float* prob,
                                   float* d_logit, float* d_h, int64_t ld_dh, float* dx, int64_t ld_dx, float* loss_out,
                                   void* workspace, int64_t workspace_bytes, int32_t parts, uint32_t* dx_amax, dr_stream_t stream) {
    if (M <= 0 || K <= 0 || H <= 0 || parts < 1 || parts > 3) return DR_EINVAL;
    if (!x || !W1 || !w2 || !labels || !dst_w1 || !dx || !workspace || loss_mode < 0 || loss_mode > 2) return DR_EINVAL;
    if (H > 32 || (K != 128 && K != 256) || (M % TT_ROWS) != 0) return DR_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(dx) & 3)) return DR_ESHAPE;
    if (ld_x < K || ld_dx < K || ld_w1 < H || ld_dst_w1 < H || ld_w2 < 1 || (dst_w2 && ld_dst_w2 < 1) || (d_h && ld_dh < H)) return DR_EINVAL;
    if (workspace_bytes < dr_tower_tail_workspace_bytes(M, K)) return DR_EINVAL;
    const int grid = tail_grid(M, K);
    float* partial = static_cast<float*>(workspace);
    float* head_partial = partial + (int64_t)grid * (K + 1) * 32;
    const float inv_n = 1.f / (float)(n_total > 0 ? n_total : M);
    // the record of dx: with both parts in this call the blocks leave their maxima in the workspace and the reduce launch STORES the
    // record -- no reset, no atomics, one launch boundary less than a memset in front of the kernel; with part 1 alone (the reduce may
    // run on another stream, later than the record's first reader) the record is reset here and raised with atomicMax
    uint32_t* amax_part = (dx_amax != nullptr && parts == 3) ? reinterpret_cast<uint32_t*>(head_partial + (int64_t)grid * TAIL_HEAD_PART) : nullptr;
    if (parts & 1) {
        if (dx_amax != nullptr && amax_part == nullptr && hipMemsetAsync(dx_amax, 0, sizeof(uint32_t), dr_s(stream)) != hipSuccess) return DR_ELAUNCH;
        TailArgs a{};
        a.x = x; a.ldx = ld_x; a.W1 = W1; a.ldw1 = ld_w1; a.b1 = b1; a.M = M; a.K = K; a.N = H;
        a.w2 = w2; a.ld_w2 = ld_w2; a.b2 = b2; a.extra = extra_logit; a.labels = labels; a.loss_mode = loss_mode; a.inv_n = inv_n;
        a.prob = prob; a.d_logit = d_logit; a.d_h = d_h; a.ld_dh = ld_dh; a.dx = dx; a.lddx = ld_dx;
        a.partial = partial; a.head_partial = head_partial; a.dx_amax = dx_amax; a.amax_part = amax_part;
        if (K == 128) hipLaunchKernelGGL((tower_tail_fused_kernel<4>), dim3(grid), dim3(256), 0, dr_s(stream), a);
        else hipLaunchKernelGGL((tower_tail_fused_kernel<8>), dim3(grid), dim3(512), 0, dr_s(stream), a);
    }
    if (parts & 2) {
        hipLaunchKernelGGL(tower_tail_reduce_kernel, dim3(K + 2), dim3(256), 0, dr_s(stream), partial, head_partial, grid, K, H, scale, inv_n,
                           dst_w1, ld_dst_w1, dst_b1, dst_w2, ld_dst_w2, dst_b2, loss_out, amax_part, dx_amax);
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}
