// EXPERIMENT: sustained bf16 MFMA rate (v_mfma_f32_32x32x16_bf16) of a register-only loop, alone and with the bf16x3 split
// VALU work of one k-tile (32 fp32 values -> 3 x bf16) issued by the same wave -- the ceiling for the emulated-fp32 GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int VALU>
__global__ __launch_bounds__(256, 2) void loop(float* out, int iters, float a0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x + j); b[j] = (__bf16)(a0 * j); }
    f32x2 v[16];
    for (int j = 0; j < 16; ++j) v[j] = f32x2{a0 * j + threadIdx.x, a0 - j};
    unsigned keep = 0;
    for (int it = 0; it < iters; ++it) {
        if (VALU) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bf16x2 h0 = __builtin_convertvector(v[j], bf16x2);
                const f32x2 r1 = v[j] - __builtin_convertvector(h0, f32x2);
                const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
                const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
                const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
                keep ^= __builtin_bit_cast(unsigned, h0) ^ __builtin_bit_cast(unsigned, h1) ^ __builtin_bit_cast(unsigned, h2);
                v[j] = v[j] * 1.0001f + r2;
            }
        }
#pragma unroll
        for (int r = 0; r < 12; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = (float)keep;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 12345.678f) out[0] = s;
}
template <int VALU>
void run(float* out, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = 256 * bpc;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((loop<VALU>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = (double)grid * 4 * iters * 48;
            printf("BF16PEAK %s blocks/CU=%d  %.3f ms  %.0f TFLOP/s(bf16)  = %.0f TF fp32-equivalent at 6 products  (%.1f cyc/MFMA/SIMD @2.4GHz)\n",
                   name, bpc, ms, mfmas * 32768.0 / ms / 1e9, mfmas * 32768.0 / 6 / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / ((double)iters * 48 * bpc));
        }
    }
}
int main() {
    float* out; hipMalloc(&out, 4);
    run<0>(out, "mfma-only ");
    run<1>(out, "mfma+split");
    return 0;
}
