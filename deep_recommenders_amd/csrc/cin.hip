// Compressed Interaction Network layer (xDeepFM) -- keras/models/ranking/xdeepfm.py:71-96 of the reference:
//
//   outer[b, d, i * Hk + j] = x0[b, i, d] * x[b, j, d]                      (:80-86: split per d, matmul, reshape, transpose)
//   pre[b, d, f]            = sum_{i,j} outer[b, d, i * Hk + j] * W[i * Hk + j, f]  (+ bias[f])   (:88-91: conv1d, width 1)
//   out[b, f, d]            = activation(pre[b, d, f])                     (:93-94)
//
// i.e. a GEMM  [B * D, H0 * Hk] x [H0 * Hk, Fm]  whose left operand is an outer product that is never written: the kernel
// keeps the x0 / x values of 64 (b, d) rows in LDS, forms each operand value with one multiply on the way into the MFMA
// (v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate -- the reference multiplies fp32 tensors), and runs the product
// TRANSPOSED (rows = feature maps f, columns = (b, d)) so that the accumulator layout stores [B, Fm, D] with d contiguous.
// Backward (autodiff of the above; g = d_out * activation'(out)):
//   d_x0[b, i, d] = sum_j x[b, j, d]  sum_f g[b, f, d] W[i, j, f]
//   d_x [b, j, d] = sum_i x0[b, i, d] sum_f g[b, f, d] W[i, j, f]
//   dW[i, j, f]   = sum_{b, d} g[b, f, d] x0[b, i, d] x[b, j, d] ;   dbias[f] = sum_{b, d} g[b, f, d]
// as straightforward fp32 kernels (one thread per output element, W broadcast across the lanes of a wave; dW as a block
// reduction per (i, j)): correct and coalesced, not tuned -- CIN is outside BASELINE.json's configurations.
#include "dr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CIN_ROWS = 64;        // (b, d) rows per block
constexpr int CIN_PITCH = 65;       // LDS pitch of a field's row values (bank-conflict-free for the per-field reads)

__device__ __forceinline__ float cin_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) return tanhf(v);
    return v;
}
// activation'(pre) expressed through out = activation(pre)
__device__ __forceinline__ float cin_act_grad(float out, int act) {
    if (act == 1) return out > 0.f ? 1.f : 0.f;
    if (act == 2) return out * (1.f - out);
    if (act == 3) return 1.f - out * out;
    return 1.f;
}

// grid.x: blocks of 64 (b, d) rows; grid.y: groups of 64 feature maps.  4 waves: wave = (row half, f half): 32 f x 32 rows each.
__global__ __launch_bounds__(256) void cin_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ x, int64_t B,
                                                      int32_t H0, int32_t Hk, int32_t D, const float* __restrict__ W,
                                                      int32_t Fm, const float* __restrict__ bias, int32_t act,
                                                      float* __restrict__ out) {
    extern __shared__ float cin_lds[];                   // [H0 + Hk][CIN_PITCH]
    float* xs0 = cin_lds;
    float* xs = cin_lds + (size_t)H0 * CIN_PITCH;
    const int64_t rows = B * D;
    const int64_t m0 = (int64_t)blockIdx.x * CIN_ROWS;
    for (int idx = threadIdx.x; idx < (H0 + Hk) * CIN_ROWS; idx += blockDim.x) {
        const int fld = idx / CIN_ROWS, r = idx % CIN_ROWS;
        const int64_t m = m0 + r;
        float v = 0.f;
        if (m < rows) {
            const int64_t b = m / D;
            const int d = (int)(m - b * D);
            v = fld < H0 ? x0[(b * H0 + fld) * D + d] : x[(b * Hk + (fld - H0)) * D + d];
        }
        cin_lds[(size_t)fld * CIN_PITCH + r] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int r = (wave & 1) * 32 + l31;                 // this lane's (b, d) row within the block (operand B column)
    const int f = blockIdx.y * 64 + (wave >> 1) * 32 + l31;   // this lane's feature map (operand A row)
    const bool fv = f < Fm;
    const int fc = fv ? f : Fm - 1;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int i = 0; i < H0; ++i) {
        const float a0 = xs0[(size_t)i * CIN_PITCH + r];
        const float* wrow = W + (int64_t)i * Hk * Fm + fc;
        for (int j = 0; j < Hk; j += 2) {                // reduction pair (j, j + 1): lane half `hi` supplies element j + hi
            const int jj = j + hi;
            const bool jv = jj < Hk;
            const int jc = jv ? jj : Hk - 1;
            const float wv = (jv && fv) ? wrow[(int64_t)jc * Fm] : 0.f;
            const float zv = jv ? a0 * xs[(size_t)jc * CIN_PITCH + r] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, zv, acc, 0, 0, 0);
        }
    }
    // C/D layout: column (row m) = lane & 31, row (feature map) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t m = m0 + r;
    if (m >= rows) return;
    const int64_t b = m / D;
    const int d = (int)(m - b * D);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int fo = blockIdx.y * 64 + (wave >> 1) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hi;
        if (fo < Fm) out[(b * Fm + fo) * D + d] = cin_act(acc[reg] + (bias != nullptr ? bias[fo] : 0.f), act);
    }
}

// one thread per (b, fld, d) of d_x0 (fld < H0) or d_x (fld >= H0); wave lanes run over d (W broadcast, x / g coalesced)
__global__ __launch_bounds__(256) void cin_bwd_dx_kernel(const float* __restrict__ x0, const float* __restrict__ x, int64_t B,
                                                         int32_t H0, int32_t Hk, int32_t D, const float* __restrict__ W,
                                                         int32_t Fm, int32_t act, const float* __restrict__ out,
                                                         const float* __restrict__ d_out, float* __restrict__ d_x0,
                                                         float* __restrict__ d_x) {
    const int64_t total = B * (H0 + Hk) * D;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int d = (int)(t % D);
        const int64_t q = t / D;
        const int fld = (int)(q % (H0 + Hk));
        const int64_t b = q / (H0 + Hk);
        const float* go = d_out + b * Fm * D + d;
        const float* oo = out + b * Fm * D + d;
        float acc = 0.f;
        if (fld < H0) {
            const int i = fld;
            for (int j = 0; j < Hk; ++j) {
                const float* w = W + ((int64_t)i * Hk + j) * Fm;
                float s = 0.f;
                for (int f = 0; f < Fm; ++f) s = fmaf(go[(int64_t)f * D] * cin_act_grad(oo[(int64_t)f * D], act), w[f], s);
                acc = fmaf(x[(b * Hk + j) * D + d], s, acc);
            }
            d_x0[(b * H0 + i) * D + d] = acc;
        } else {
            const int j = fld - H0;
            for (int i = 0; i < H0; ++i) {
                const float* w = W + ((int64_t)i * Hk + j) * Fm;
                float s = 0.f;
                for (int f = 0; f < Fm; ++f) s = fmaf(go[(int64_t)f * D] * cin_act_grad(oo[(int64_t)f * D], act), w[f], s);
                acc = fmaf(x0[(b * H0 + i) * D + d], s, acc);
            }
            d_x[(b * Hk + j) * D + d] = acc;
        }
    }
}

// block per (i, j) (blockIdx.x < H0 * Hk) or for the bias (blockIdx.x == H0 * Hk): fixed-order block reduction over (b, d)
__global__ __launch_bounds__(256) void cin_bwd_dw_kernel(const float* __restrict__ x0, const float* __restrict__ x, int64_t B,
                                                         int32_t H0, int32_t Hk, int32_t D, int32_t Fm, int32_t act,
                                                         const float* __restrict__ out, const float* __restrict__ d_out,
                                                         float* __restrict__ dW, float* __restrict__ dbias) {
    __shared__ float red[4];
    const bool is_bias = (int)blockIdx.x == H0 * Hk;
    if (is_bias && dbias == nullptr) return;
    const int i = is_bias ? 0 : blockIdx.x / Hk, j = is_bias ? 0 : blockIdx.x % Hk;
    const int64_t rows = B * D;
    for (int f = 0; f < Fm; ++f) {
        float acc = 0.f;
        for (int64_t m = threadIdx.x; m < rows; m += blockDim.x) {
            const int64_t b = m / D;
            const int d = (int)(m - b * D);
            const int64_t o = (b * Fm + f) * D + d;
            const float gv = d_out[o] * cin_act_grad(out[o], act);
            acc += is_bias ? gv : gv * x0[(b * H0 + i) * D + d] * x[(b * Hk + j) * D + d];
        }
        acc = dr_wave_sum(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float s = (red[0] + red[1]) + (red[2] + red[3]);
            if (is_bias) dbias[f] = s;
            else dW[((int64_t)i * Hk + j) * Fm + f] = s;
        }
    }
}

}  // namespace

extern "C" int dr_cin_fwd(const float* x0, const float* x, int64_t B, int32_t H0, int32_t Hk, int32_t D, const float* W,
                          int32_t Fm, const float* bias, int32_t act, float* out, dr_stream_t stream) {
    if (B < 0 || H0 <= 0 || Hk <= 0 || D <= 0 || Fm <= 0 || act < 0 || act > 3) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!x0 || !x || !W || !out) return DR_EINVAL;
    const size_t lds = (size_t)(H0 + Hk) * CIN_PITCH * sizeof(float);
    if (lds > 160 * 1024) return DR_ESHAPE;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(cin_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DR_ELAUNCH;
    const int64_t rows = B * D;
    const int64_t gx = (rows + CIN_ROWS - 1) / CIN_ROWS;
    if (gx > 0x7fffffff) return DR_EINVAL;
    hipLaunchKernelGGL(cin_fwd_kernel, dim3((unsigned)gx, (unsigned)((Fm + 63) / 64)), dim3(256), lds, dr_s(stream), x0, x, B, H0, Hk,
                       D, W, Fm, bias, act, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_cin_bwd(const float* x0, const float* x, int64_t B, int32_t H0, int32_t Hk, int32_t D, const float* W,
                          int32_t Fm, int32_t act, const float* out, const float* d_out, float* d_x0, float* d_x, float* dW,
                          float* dbias, dr_stream_t stream) {
    if (B < 0 || H0 <= 0 || Hk <= 0 || D <= 0 || Fm <= 0 || act < 0 || act > 3) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!x0 || !x || !W || !out || !d_out || !d_x0 || !d_x || !dW) return DR_EINVAL;
    hipLaunchKernelGGL(cin_bwd_dx_kernel, dim3(dr_grid_for(B * (H0 + Hk) * D, 256, 8192)), dim3(256), 0, dr_s(stream), x0, x, B, H0,
                       Hk, D, W, Fm, act, out, d_out, d_x0, d_x);
    hipLaunchKernelGGL(cin_bwd_dw_kernel, dim3((unsigned)(H0 * Hk + 1)), dim3(256), 0, dr_s(stream), x0, x, B, H0, Hk, D, Fm, act,
                       out, d_out, dW, dbias);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// ---- DIN ActivationUnit input (keras/models/ranking/din.py:59-67 of the reference): concat([x, y, interacter([x, y])], axis 1)
// in one pass; mode 0: no interacter ([x, y]), 1: Subtract (x - y), 2: Multiply (x * y).  Backward: the three column blocks of
// d_concat folded back into d_x, d_y.
namespace {
__global__ __launch_bounds__(256) void din_concat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t B,
                                                             int32_t D, int32_t mode, float* __restrict__ out, int64_t ld) {
    const int64_t n = B * D, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const int64_t b = t / D;
        const int d = (int)(t - b * D);
        const float xv = x[t], yv = y[t];
        float* o = out + b * ld;
        o[d] = xv;
        o[D + d] = yv;
        if (mode == 1) o[2 * D + d] = xv - yv;
        else if (mode == 2) o[2 * D + d] = xv * yv;
    }
}
__global__ __launch_bounds__(256) void din_concat_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t B,
                                                             int32_t D, int32_t mode, const float* __restrict__ d_out, int64_t ld,
                                                             float* __restrict__ d_x, float* __restrict__ d_y) {
    const int64_t n = B * D, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const int64_t b = t / D;
        const int d = (int)(t - b * D);
        const float* g = d_out + b * ld;
        float gx = g[d], gy = g[D + d];
        if (mode == 1) { gx += g[2 * D + d]; gy -= g[2 * D + d]; }
        else if (mode == 2) { gx += g[2 * D + d] * y[t]; gy += g[2 * D + d] * x[t]; }
        d_x[t] = gx;
        d_y[t] = gy;
    }
}
}  // namespace

extern "C" int dr_din_concat_fwd(const float* x, const float* y, int64_t B, int32_t D, int32_t mode, float* out, int64_t ld_out,
                                 dr_stream_t stream) {
    if (B < 0 || D <= 0 || mode < 0 || mode > 2 || ld_out < (mode ? 3 : 2) * (int64_t)D) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!x || !y || !out) return DR_EINVAL;
    hipLaunchKernelGGL(din_concat_fwd_kernel, dim3(dr_grid_for(B * D, 256)), dim3(256), 0, dr_s(stream), x, y, B, D, mode, out, ld_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_din_concat_bwd(const float* x, const float* y, int64_t B, int32_t D, int32_t mode, const float* d_out,
                                 int64_t ld_dout, float* d_x, float* d_y, dr_stream_t stream) {
    if (B < 0 || D <= 0 || mode < 0 || mode > 2 || ld_dout < (mode ? 3 : 2) * (int64_t)D) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!x || !y || !d_out || !d_x || !d_y) return DR_EINVAL;
    hipLaunchKernelGGL(din_concat_bwd_kernel, dim3(dr_grid_for(B * D, 256)), dim3(256), 0, dr_s(stream), x, y, B, D, mode, d_out,
                       ld_dout, d_x, d_y);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
