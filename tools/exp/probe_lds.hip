// GPU probe (not product): lane maps of ds_read_b64_tr_b16 and of the 16-byte LDS-DMA load on gfx950, printed as tables.
// Build: hipcc --offload-arch=gfx950 -O2 tools/exp/probe_lds.hip -o tools/exp/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe_tr(const uint16_t* in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
// each lane fetches 16 bytes from ITS OWN global address (lane l reads element block perm(l)); LDS destination is the
// wave-uniform base + 16 * lane
__global__ void probe_glds(const uint32_t* in, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadu;
    __syncthreads();
    const int l = threadIdx.x;
    const uint32_t* src = in + 4 * ((l * 7) % 64);          // a permutation of the 64 16-byte blocks
    __builtin_amdgcn_global_load_lds(src, (uint32_t __attribute__((address_space(3)))*)lds, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(src + 256, (uint32_t __attribute__((address_space(3)))*)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    static uint16_t h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
    uint16_t *d, *o; (void)hipMalloc(&d, 8192); (void)hipMalloc(&o, 512);
    (void)hipMemcpy(d, h, 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d, o);
    uint16_t r[256]; (void)hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, lane address = base + 8*lane bytes, lds[i] = i (b16 units)\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    static uint32_t g[1024]; for (int i = 0; i < 1024; ++i) g[i] = i;
    uint32_t *gd, *go; (void)hipMalloc(&gd, 4096); (void)hipMalloc(&go, 2048);
    (void)hipMemcpy(gd, g, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, gd, go);
    static uint32_t gr[512]; (void)hipMemcpy(gr, go, 2048, hipMemcpyDeviceToHost);
    printf("global_load_lds_dwordx4: lane l sources block (7l mod 64); lds dword 4*slot holds:\n");
    for (int s = 0; s < 128; ++s) printf("%s%4u", (s % 16 == 0) ? "\n  " : " ", gr[4 * s]);
    printf("\n");
    return 0;
}
