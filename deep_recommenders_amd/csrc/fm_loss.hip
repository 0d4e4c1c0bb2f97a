// K6 — stand-alone FM second-order term on a caller-provided [B, F, D] tensor (forward + backward);
// K11 — fused sigmoid + binary cross-entropy + gradient wrt the logit.  Both HBM-bound streaming kernels.
//
// K6 replaces keras FM.call (keras/models/ranking/fm.py:28-35) / estimator fm(x)
// (estimator/models/feature_interaction/fm.py:10-26); K11 replaces tf.nn.sigmoid + the three losses used
// by the reference's examples (train_fm_on_movielens_estimator.py:46, train_deepfm_on_movielens_estimator.py:47,
// train_deepfm_on_movielens_keras.py:43).
#include "dr_common.h"

namespace {

// One wave per example; lanes stride over d, loop over f.  x row of an example is F*D contiguous floats.
__global__ __launch_bounds__(256) void fm2_fwd_kernel(const float* __restrict__ x, int64_t B, int32_t F, int32_t D,
                                                      float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int64_t b = wave0; b < B; b += nwaves) {
        const float* xb = x + b * (int64_t)F * D;
        float t = 0.f;
        for (int d = lane; d < D; d += 64) {
            float s = 0.f, ss = 0.f;
            for (int f = 0; f < F; ++f) {
                const float v = xb[f * D + d];
                s += v;
                ss = fmaf(v, v, ss);
            }
            t += s * s - ss;
        }
        t = dr_wave_sum(t);
        if (lane == 0) out[b] = 0.5f * t;
    }
}

__global__ __launch_bounds__(256) void fm2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ d_out,
                                                      int64_t B, int32_t F, int32_t D, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int64_t b = wave0; b < B; b += nwaves) {
        const float* xb = x + b * (int64_t)F * D;
        float* dxb = dx + b * (int64_t)F * D;
        const float g = d_out[b];
        for (int d = lane; d < D; d += 64) {
            float s = 0.f;
            for (int f = 0; f < F; ++f) s += xb[f * D + d];
            for (int f = 0; f < F; ++f) dxb[f * D + d] = g * (s - xb[f * D + d]);
        }
    }
}

// ---- K11 ---------------------------------------------------------------------------------
constexpr int BCE_BLOCKS = 512;   // partial sums; stage 2 is one block

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void bce_stage1(const float* __restrict__ logits, const float* __restrict__ logits_b,
                                                  int64_t ldb, const float* __restrict__ labels,
                                                  int64_t n, int mode, float* __restrict__ prob,
                                                  float* __restrict__ d_logit, float* __restrict__ partial) {
    __shared__ float red[4];
    const float inv_n = 1.f / (float)n;
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = logits[i] + (logits_b != nullptr ? logits_b[i * ldb] : 0.f), z = labels[i];
        float p, l, g;
        dr_bce_terms(x, z, mode, p, l, g);      // [TF] B9 / B10 / B11
        acc += l;
        if (prob != nullptr) prob[i] = p;
        if (d_logit != nullptr) d_logit[i] = g * inv_n;
    }
    acc = dr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bce_stage2(const float* __restrict__ partial, int nparts, int64_t n,
                                                  float* __restrict__ loss_out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
}


__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = sigmoidf_(x[i]);
}
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          int64_t n, float* __restrict__ dx) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float p = y[i];
        dx[i] = dy[i] * p * (1.f - p);
    }
}

// losses on PROBABILITIES (the reference applies them to the model's sigmoid output):
// mode 1 = tf.losses.log_loss ([TF] B10), mode 2 = keras binary_crossentropy ([TF] B11)
__global__ __launch_bounds__(256) void bce_prob_stage1(const float* __restrict__ prob, const float* __restrict__ labels,
                                                       int64_t n, int mode, float* __restrict__ d_prob,
                                                       float* __restrict__ partial) {
    __shared__ float red[4];
    const float inv_n = 1.f / (float)n;
    const float eps = 1e-7f;
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float p = prob[i], z = labels[i];
        float pc = p, dclip = 1.f;
        if (mode == 2) {
            if (p < eps) { pc = eps; dclip = 0.f; }
            else if (p > 1.f - eps) { pc = 1.f - eps; dclip = 0.f; }
        }
        acc += -z * logf(pc + eps) - (1.f - z) * logf(1.f - pc + eps);
        if (d_prob != nullptr) d_prob[i] = (-z / (pc + eps) + (1.f - z) / (1.f - pc + eps)) * dclip * inv_n;
    }
    acc = dr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int dr_fm2_fwd(const float* x, int64_t B, int32_t F, int32_t D, float* out, dr_stream_t stream) {
    if (B < 0 || F <= 0 || D <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!x || !out) return DR_EINVAL;
    hipLaunchKernelGGL(fm2_fwd_kernel, dim3(dr_grid_for(B, 4)), dim3(256), 0, dr_s(stream), x, B, F, D, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_fm2_bwd(const float* x, const float* d_out, int64_t B, int32_t F, int32_t D, float* dx,
                          dr_stream_t stream) {
    if (B < 0 || F <= 0 || D <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!x || !d_out || !dx) return DR_EINVAL;
    hipLaunchKernelGGL(fm2_bwd_kernel, dim3(dr_grid_for(B, 4)), dim3(256), 0, dr_s(stream), x, d_out, B, F, D, dx);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_bce_fwd_bwd(const float* logits, const float* logits_b, int64_t ld_b, const float* labels, int64_t n,
                              int32_t mode, float* prob, float* d_logit, float* loss_out, float* workspace,
                              dr_stream_t stream) {
    if (n <= 0 || mode < 0 || mode > 2) return DR_EINVAL;
    if (!logits || !labels || !loss_out || !workspace) return DR_EINVAL;
    const int grid = dr_grid_for(n, 256, BCE_BLOCKS);
    hipLaunchKernelGGL(bce_stage1, dim3(grid), dim3(256), 0, dr_s(stream), logits, logits_b, ld_b, labels, n, mode,
                       prob, d_logit, workspace);
    hipLaunchKernelGGL(bce_stage2, dim3(1), dim3(256), 0, dr_s(stream), workspace, grid, n, loss_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_sigmoid_fwd(const float* x, int64_t n, float* y, dr_stream_t stream) {
    if (n < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!x || !y) return DR_EINVAL;
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), x, n, y);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_sigmoid_bwd(const float* y, const float* dy, int64_t n, float* dx, dr_stream_t stream) {
    if (n < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!y || !dy || !dx) return DR_EINVAL;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), y, dy, n, dx);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_bce_prob_fwd_bwd(const float* prob, const float* labels, int64_t n, int32_t mode, float* d_prob,
                                   float* loss_out, float* workspace, dr_stream_t stream) {
    if (n <= 0 || (mode != 1 && mode != 2)) return DR_EINVAL;
    if (!prob || !labels || !loss_out || !workspace) return DR_EINVAL;
    const int grid = dr_grid_for(n, 256, BCE_BLOCKS);
    hipLaunchKernelGGL(bce_prob_stage1, dim3(grid), dim3(256), 0, dr_s(stream), prob, labels, n, mode, d_prob, workspace);
    hipLaunchKernelGGL(bce_stage2, dim3(1), dim3(256), 0, dr_s(stream), workspace, grid, n, loss_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
