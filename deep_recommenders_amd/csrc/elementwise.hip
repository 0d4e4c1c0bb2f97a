// Small element-wise / reduction kernels that close the reference's layer surface around the GEMM path:
//   * Dense activations other than ReLU (keras/models/ranking/deepfm.py:30-34 takes any `dnn_activation`;
//     estimator/models/feature_interaction/dnn.py:9-14 any callable): sigmoid / tanh applied to a linear GEMM's output, and their
//     backward through the saved output;
//   * dropout (estimator/models/feature_interaction/dnn.py:26-27: tf.nn.dropout after every hidden layer, with NO train / eval
//     switch -- always on): counter-based mask (one 32-bit hash of (seed, element index) per element; the
//     reference's TF Philox stream is not reproducible and is not a goal), kept elements scaled by 1 / (1 - rate),
//     mask saved as bytes for the backward;
//   * sum of squares (Keras `kernel_regularizer` / `bias_regularizer` L2 terms of dcn.Cross / xdeepfm.CIN / din.ActivationUnit:
//     l2 * sum(w^2)) and a plain sum (first-order bias gradient), both as fixed-order two-stage reductions (deterministic).
// All HBM-bound streaming kernels: 8 B per element (activations), 9 B (dropout), 4 B (reductions).
#include "dr_common.h"

namespace {

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) return tanhf(v);
    return v;
}
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y * (1.f - y);
    if (act == 3) return 1.f - y * y;
    return 1.f;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(float* __restrict__ x, int64_t M, int32_t N, int64_t ld, int32_t act) {
    const int64_t n = M * N, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / N;
        float* p = x + r * ld + (i - r * N);
        *p = act_apply(*p, act);
    }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, int64_t ld_y, float* __restrict__ dy,
                                                      int64_t ld_dy, int64_t M, int32_t N, int32_t act) {
    const int64_t n = M * N, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / N;
        const int64_t c = i - r * N;
        dy[r * ld_dy + c] *= act_grad_from_out(y[r * ld_y + c], act);
    }
}

// lowbias32-style integer hash of (seed, index): well mixed, one multiply chain per element
__device__ __forceinline__ uint32_t mix32(uint64_t seed, uint64_t idx) {
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, int64_t ld_x, int64_t M, int32_t N,
                                                          float rate, uint64_t seed, float* __restrict__ y, int64_t ld_y,
                                                          uint8_t* __restrict__ mask) {
    const int64_t n = M * N, stride = (int64_t)gridDim.x * blockDim.x;
    const uint32_t thresh = (uint32_t)fminf(4294967040.f, rate * 4294967296.f);      // keep iff hash >= rate * 2^32
    const float scale = 1.f / (1.f - rate);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / N;
        const int64_t c = i - r * N;
        const bool keep = mix32(seed, (uint64_t)i) >= thresh;
        mask[i] = keep ? 1 : 0;
        y[r * ld_y + c] = keep ? x[r * ld_x + c] * scale : 0.f;
    }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, int64_t ld_dy, const uint8_t* __restrict__ mask,
                                                          int64_t M, int32_t N, float rate, float* __restrict__ dx, int64_t ld_dx) {
    const int64_t n = M * N, stride = (int64_t)gridDim.x * blockDim.x;
    const float scale = 1.f / (1.f - rate);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / N;
        const int64_t c = i - r * N;
        dx[r * ld_dx + c] = mask[i] ? dy[r * ld_dy + c] * scale : 0.f;
    }
}

// stage 1: per-block partial of sum(x) (SQ = false) or sum(x^2) (SQ = true) in a fixed order; stage 2: one block sums the partials
template <bool SQ>
__global__ __launch_bounds__(256) void reduce_stage1_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
    __shared__ float ws[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        acc += SQ ? v * v : v;
    }
    acc = dr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(256) void reduce_stage2_kernel(const float* __restrict__ partial, int32_t nb, float alpha,
                                                            float* __restrict__ out, int32_t accumulate) {
    __shared__ float ws[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
    acc = dr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = alpha * ((ws[0] + ws[1]) + (ws[2] + ws[3]));
        out[0] = accumulate ? out[0] + s : s;
    }
}

}  // namespace

extern "C" int dr_act_fwd(float* x, int64_t M, int32_t N, int64_t ld, int32_t act, dr_stream_t stream) {
    if (M < 0 || N <= 0 || ld < N || act < 0 || act > 3) return DR_EINVAL;
    if (M == 0 || act == 0) return DR_OK;
    if (!x) return DR_EINVAL;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(dr_grid_for(M * N, 256)), dim3(256), 0, dr_s(stream), x, M, N, ld, act);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_act_bwd(const float* y, int64_t ld_y, float* dy, int64_t ld_dy, int64_t M, int32_t N, int32_t act,
                          dr_stream_t stream) {
    if (M < 0 || N <= 0 || ld_y < N || ld_dy < N || act < 0 || act > 3) return DR_EINVAL;
    if (M == 0 || act == 0) return DR_OK;
    if (!y || !dy) return DR_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(dr_grid_for(M * N, 256)), dim3(256), 0, dr_s(stream), y, ld_y, dy, ld_dy, M, N, act);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_dropout_fwd(const float* x, int64_t ld_x, int64_t M, int32_t N, float rate, uint64_t seed, float* y,
                              int64_t ld_y, uint8_t* mask, dr_stream_t stream) {
    if (M < 0 || N <= 0 || ld_x < N || ld_y < N || !(rate >= 0.f) || !(rate < 1.f)) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x || !y || !mask) return DR_EINVAL;
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(dr_grid_for(M * N, 256)), dim3(256), 0, dr_s(stream), x, ld_x, M, N, rate, seed, y,
                       ld_y, mask);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_dropout_bwd(const float* dy, int64_t ld_dy, const uint8_t* mask, int64_t M, int32_t N, float rate, float* dx,
                              int64_t ld_dx, dr_stream_t stream) {
    if (M < 0 || N <= 0 || ld_dy < N || ld_dx < N || !(rate >= 0.f) || !(rate < 1.f)) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!dy || !dx || !mask) return DR_EINVAL;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(dr_grid_for(M * N, 256)), dim3(256), 0, dr_s(stream), dy, ld_dy, mask, M, N, rate, dx,
                       ld_dx);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// out[0] (+)= alpha * sum_i x[i] (squared = 0) or alpha * sum_i x[i]^2 (squared = 1); workspace: 1024 floats
extern "C" int dr_reduce_sum(const float* x, int64_t n, int32_t squared, float alpha, int32_t accumulate, float* out,
                             float* workspace, dr_stream_t stream) {
    if (n < 0 || !out || !workspace) return DR_EINVAL;
    const int nb = n == 0 ? 1 : dr_grid_for(n, 256 * 16, 1024);
    if (n > 0 && !x) return DR_EINVAL;
    if (squared) hipLaunchKernelGGL(reduce_stage1_kernel<true>, dim3(nb), dim3(256), 0, dr_s(stream), x, n, workspace);
    else hipLaunchKernelGGL(reduce_stage1_kernel<false>, dim3(nb), dim3(256), 0, dr_s(stream), x, n, workspace);
    hipLaunchKernelGGL(reduce_stage2_kernel, dim3(1), dim3(256), 0, dr_s(stream), workspace, nb, alpha, out, accumulate);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// dst[0] = the device's constant-rate wall clock (100 MHz, s_memrealtime) at the moment this one-thread kernel runs on `stream`.
// Two of these around a cross-stream wait measure how long the stream sat idle there -- without HIP timing events, whose records
// around a wait turned out to serialise the sharded step (round 4: 1.93 ms -> 4.1 ms with four bracketed waits per step).
__global__ void clock_stamp_kernel(uint64_t* __restrict__ dst) { dst[0] = wall_clock64(); }

// Measurement plumbing (bench.py's `measured_copy_ceiling`): a streaming copy of 16-byte vectors with nontemporal loads and stores,
// 4 independent vectors in flight per thread, grid = 16 blocks per CU -- the guide's "float4 copy" (6.29 TB/s read + write on MI355X).
// What an HBM-bound kernel of this library can at best reach on THIS box; hipMemcpyDtoD / a torch copy is not a ceiling.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_nt_kernel(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 4) * 256 + threadIdx.x; i < n16; i += stride) {
        copy_f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * 256;
            v[u] = __builtin_nontemporal_load(src + (j < n16 ? j : i));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * 256;
            if (j < n16) __builtin_nontemporal_store(v[u], dst + j);
        }
    }
}

extern "C" int dr_copy_nt(const void* src, void* dst, int64_t bytes, dr_stream_t stream) {
    if (bytes < 0 || (bytes & 15)) return DR_EINVAL;
    if (bytes == 0) return DR_OK;
    if (!src || !dst || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) return DR_EINVAL;
    const int64_t n16 = bytes / 16;
    const int64_t want = (n16 + 1023) / 1024;
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(copy_nt_kernel, dim3(grid), dim3(256), 0, dr_s(stream), static_cast<const copy_f4*>(src), static_cast<copy_f4*>(dst), n16);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_clock_stamp(uint64_t* dst, dr_stream_t stream) {
    if (!dst) return DR_EINVAL;
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, dr_s(stream), dst);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
