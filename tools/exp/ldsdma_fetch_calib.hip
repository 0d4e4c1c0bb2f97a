// CALIBRATION: does rocprofv3's FETCH_SIZE tally the fused first layer's gather -- LDS-DMA, 8 lanes x 16 bytes
// per 128-byte line, structured buffer resource, random rows -- in full, or at 1/2 like a wide coalesced streaming read
// (MI355X_MICROARCH.md, HBM section)?  Three kernels over buffers far larger than every cache, each moving a KNOWN byte count
// exactly once:
//   stream_dwordx4     plain coalesced global_load_dwordx4 of the whole buffer                (the guide's reference pattern)
//   gather_ldsdma      struct buffer_load ... lds, 16 B / lane, lane -> (row = lane >> 3, chunk = lane & 7) of a random
//                      permutation of 256-byte rows, half row h of every row -- the fused kernel's instruction and mapping
//                      (csrc/bf3_gemm.hip issue_gather)
//   gather_ldsdma_both the same with both half rows fetched by consecutive instructions
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv`; tools/exp/ldsdma_fetch_calib.sh prints
// FETCH_SIZE * 1024 / known bytes per kernel (1.0 = tallied in full, 0.5 = needs the x2).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(256) void stream_dwordx4(const float4* __restrict__ src, int64_t n4, float* __restrict__ sink) {
    float4 a = make_float4(0, 0, 0, 0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = src[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = a.x;       // keep the loads
}

template <int BOTH>
__global__ __launch_bounds__(256) void gather_ldsdma(const float* __restrict__ table, const int* __restrict__ perm, int64_t nrows,
                                                     int half, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 256, 0x7fffffff, 0x00020000);
    const int64_t nwaves = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
    unsigned char* dst = smem + wave * 2048;
    for (int64_t g = w0; g * 8 < nrows; g += nwaves) {           // 8 rows per instruction
        int64_t r = g * 8 + (lane >> 3);
        r = r < nrows ? r : nrows - 1;
        const int idx = perm[r];
        __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, idx, (lane & 7) * 16 + half * 128, 0, 0, 0);
        if (BOTH) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + 1024), 16, idx, (lane & 7) * 16 + (half ^ 1) * 128, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (reinterpret_cast<float*>(smem)[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int64_t nrows = 1 << 22;                                // 4 M rows x 256 B = 1 GiB
    float* table; int* perm; float* sink;
    CK(hipMalloc(&table, nrows * 256));
    CK(hipMalloc(&perm, nrows * 4));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(table, 0, nrows * 256));
    std::vector<int> p(nrows);
    for (int64_t i = 0; i < nrows; ++i) p[i] = (int)i;
    uint64_t s = 88172645463325252ull;
    for (int64_t i = nrows - 1; i > 0; --i) {                     // Fisher-Yates with xorshift64
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int64_t j = (int64_t)(s % (uint64_t)(i + 1));
        std::swap(p[i], p[j]);
    }
    CK(hipMemcpy(perm, p.data(), nrows * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_dwordx4, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(table), nrows * 16, sink);
        hipLaunchKernelGGL((gather_ldsdma<0>), dim3(4096), dim3(256), 0, 0, table, perm, nrows, rep & 1, sink);
        hipLaunchKernelGGL((gather_ldsdma<1>), dim3(4096), dim3(256), 0, 0, table, perm, nrows, 0, sink);
        CK(hipDeviceSynchronize());
    }
    printf("known bytes per launch: stream_dwordx4 %lld  gather_ldsdma<0> %lld (+ %lld of indices)  gather_ldsdma<1> %lld (+ %lld)\n",
           (long long)(nrows * 256), (long long)(nrows * 128), (long long)(nrows * 4), (long long)(nrows * 256), (long long)(nrows * 4));
    return 0;
}
