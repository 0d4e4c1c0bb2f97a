// GPU experiment (not product): where does a k-tile of the planes NT GEMM go?  Ablations of bf3_gemm_nt_kernel at the first
// tower layer's forward shape (M = 65536, K = 1696, N = 256) on random operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/exp/bf3_ablate.hip -o /tmp/bf3_ablate
#include "../../deep_recommenders_amd/csrc/bf3_gemm.hip"
#include <cstdio>
#include <vector>

__global__ void fill_kernel(uint16_t* p, int64_t n, uint32_t seed) {       // random bf16 in +-[0.5, 2): realistic bit toggling
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (uint16_t)((h & 0x80ffu) | 0x3f00u);
    }
}

template <int NW, int DBG>
static float run_pipe(const NtArgs& g, int grid, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((bf3_gemm_nt_pipe_kernel<NW, DBG>), dim3(grid), dim3(64 * NW), 0, 0, g);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((bf3_gemm_nt_pipe_kernel<NW, DBG>), dim3(grid), dim3(64 * NW), 0, 0, g);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

template <int DBG>
static float run_rs(const RsArgs& g, int grid, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, DBG>), dim3(grid), dim3(512), 0, 0, g);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, DBG>), dim3(grid), dim3(512), 0, 0, g);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

template <int DBG>
static float run(const NtArgs& g, int grid, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((bf3_gemm_nt_kernel<2, 4, DBG>), dim3(grid), dim3(NTHREADS), 0, 0, g);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((bf3_gemm_nt_kernel<2, 4, DBG>), dim3(grid), dim3(NTHREADS), 0, 0, g);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int64_t M = 65536; const int K = 1696;
    for (int N : {256, 1024}) {
        __bf16 *A, *B; float* C;
        (void)hipMalloc(&A, 3 * M * K * 2); (void)hipMalloc(&B, 3 * (int64_t)N * K * 2); (void)hipMalloc(&C, M * N * 4);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint16_t*)A, 3 * M * K, 1u);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint16_t*)B, 3 * (int64_t)N * K, 7u);
        NtArgs g{A, M * K, K, B, (int64_t)N * K, K, M, N, K, C, N, nullptr, 0, nullptr, 0};
        const int64_t tiles = ((M + 127) / 128) * ((N + 255) / 256);
        const int grid = tiles < 256 ? (int)tiles : 256;
        const double fl = 2.0 * M * 1677 * N;
        float t;
        t = run<0>(g, grid, 10); printf("N=%4d base            %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run<1>(g, grid, 10); printf("N=%4d no DMA          %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run<2>(g, grid, 10); printf("N=%4d no MFMA         %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run<4>(g, grid, 10); printf("N=%4d DMA cache hits  %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run<3>(g, grid, 10); printf("N=%4d no DMA no MFMA  %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run<6>(g, grid, 10); printf("N=%4d hits, no MFMA   %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        const int64_t ptiles = ((M + 127) / 128) * ((N + 127) / 128);
        const int pgrid = ptiles < 256 ? (int)ptiles : 256;
        t = run_pipe<8, 0>(g, pgrid, 10); printf("N=%4d PIPE8 base            %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run_pipe<16, 0>(g, pgrid, 10); printf("N=%4d PIPE16 base           %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run_pipe<16, 1>(g, pgrid, 10); printf("N=%4d PIPE16 no DMA         %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run_pipe<16, 2>(g, pgrid, 10); printf("N=%4d PIPE16 no MFMA        %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run_pipe<16, 4>(g, pgrid, 10); printf("N=%4d PIPE16 DMA cache hits %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        t = run_pipe<16, 6>(g, pgrid, 10); printf("N=%4d PIPE16 hits, no MFMA  %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
        {   // register-split kernel: fp32 A [M, 1680] (K = 1677), B planes as above
            float* Af; (void)hipMalloc(&Af, M * 1680 * 4);
            hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint16_t*)Af, M * 1680 * 2, 3u);
            RsArgs r{Af, 1680, B, (int64_t)N * K, K, M, N, 1677, C, N, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 0.f, nullptr};
            const int64_t rt = ((M + 255) / 256) * ((N + 255) / 256);
            const int rg = rt < 256 ? (int)rt : 256;
            t = run_rs<0>(r, rg, 10); printf("N=%4d RS base             %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
            t = run_rs<1>(r, rg, 10); printf("N=%4d RS no B DMA         %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
            t = run_rs<2>(r, rg, 10); printf("N=%4d RS no MFMA          %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
            t = run_rs<8>(r, rg, 10); printf("N=%4d RS A not advancing  %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
            t = run_rs<9>(r, rg, 10); printf("N=%4d RS A fixed, no DMA  %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
            t = run_rs<4>(r, rg, 10); printf("N=%4d RS B cache hits     %8.1f us %6.1f TF/s\n", N, t, fl / t / 1e6);
            (void)hipFree(Af);
        }
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
    }
    return 0;
}
