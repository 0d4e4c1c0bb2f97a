// EXPERIMENT: sustained fp32 MFMA rate of a register-only loop (no memory traffic) -- the practical ceiling for the
// GEMM kernels, to be compared with the 157.3 TFLOP/s paper peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        const int grid = 256 * blocks_per_cu;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((mfma_loop<4>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 /*waves*/ * iters * 8 * 4 * 4096.0;
            printf("MFMAPEAK blocks/CU=%d rep=%d  %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, rep, ms, flops / ms / 1e9);
        }
    }
    // long run (~1 s) to expose clock throttling under sustained load
    hipEventRecord(e0);
    for (int k = 0; k < 40; ++k) hipLaunchKernelGGL((mfma_loop<4>), dim3(1024), dim3(256), 0, 0, out, iters * 4, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("MFMAPEAK sustained %.1f ms  %.1f TFLOP/s\n", ms, 40.0 * 1024 * 4 * iters * 4 * 8 * 4 * 4096.0 / ms / 1e9);
    return 0;
}
