// Shared helpers for the gfx950 kernels behind include/dr_hotpath.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dr_hotpath.h"

#define DR_WAVE 64

#define DR_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return DR_ELAUNCH;           \
    } while (0)

static inline hipStream_t dr_s(dr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// memory-bound kernels: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
static inline int dr_grid_for(int64_t work_items, int items_per_block, int max_blocks = 2048) {
    int64_t g = (work_items + items_per_block - 1) / items_per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

__device__ __forceinline__ float dr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Binary cross-entropy terms of one example, logit x, label z (shared by the stand-alone loss kernel and the fused tower
// head).  mode 0: [TF] sigmoid_cross_entropy_with_logits  max(x,0) - x z + log1p(exp(-|x|)) ; mode 1: tf.losses.log_loss
// on p = sigmoid(x) (eps 1e-7) ; mode 2: keras binary_crossentropy (p clipped to [eps, 1-eps] first).  Outputs the
// probability, the loss term and d(loss term)/dx.
__device__ __forceinline__ void dr_bce_terms(float x, float z, int mode, float& p, float& l, float& g) {
    p = 1.f / (1.f + expf(-x));
    if (mode == 0) {
        l = fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
        g = p - z;
    } else {
        const float eps = 1e-7f;
        float pc = p;
        float dclip = 1.f;
        if (mode == 2) {
            if (p < eps) { pc = eps; dclip = 0.f; }
            else if (p > 1.f - eps) { pc = 1.f - eps; dclip = 0.f; }
        }
        l = -z * logf(pc + eps) - (1.f - z) * logf(1.f - pc + eps);
        const float dl_dp = -z / (pc + eps) + (1.f - z) / (1.f - pc + eps);
        g = dl_dp * dclip * p * (1.f - p);
    }
}

// dst[0] += alpha * sum_{i<n} x[i], computed by ONE 256-thread block in a fixed order (deterministic) and written
// with a plain read-modify-write: the caller guarantees this block is the only writer of dst during the kernel.
// Replaces "one same-address atomic per wave": tens of thousands of those serialise on a single L2 channel
// (~88 per microsecond on MI355X) and were costing more than the scatter kernel they decorated.
__device__ __forceinline__ void dr_block_sum_axpy(const float* __restrict__ x, int64_t n, float alpha,
                                                  float* __restrict__ dst) {
    __shared__ float dr_wsum_[4];
    float acc = 0.f;
    const int64_t nv = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? (n >> 2) : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = threadIdx.x; i < nv; i += 256 * 4) {
        float4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t q = i + j * 256;
            a[j] = x4[q < nv ? q : i];
            if (q >= nv) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += (a[j].x + a[j].y) + (a[j].z + a[j].w);
    }
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += 256) acc += x[i];
    acc = dr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) dr_wsum_[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dst[0] = fmaf(alpha, (dr_wsum_[0] + dr_wsum_[1]) + (dr_wsum_[2] + dr_wsum_[3]), dst[0]);
}
