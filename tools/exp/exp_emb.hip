// EXPERIMENT kernels (not product): variants of the embedding gather / scatter to locate the ceilings.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdint.h>

#define LPR 16
#define NS 4

// mode bits: 1 = skip concat store, 2 = nontemporal concat store, 4 = nontemporal table load
template <int U, int MODE>
__global__ __launch_bounds__(256) void fwd_var(const int64_t* __restrict__ ids, int64_t B, int F,
                                               const float* __restrict__ table, int64_t V, float* __restrict__ concat,
                                               int64_t ld, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 S = make_float4(0, 0, 0, 0);
    for (int64_t b = w0; b < B; b += nw) {
        int64_t my_row = -1;
        if (lane < F) my_row = ids[b * F + lane] + (int64_t)lane * V;
        float* out_row = concat + b * ld;
        for (int f0 = 0; f0 < F; f0 += NS * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * NS + slot;
                int64_t row = __shfl(my_row, f < F ? f : 0, 64);
                v[u] = make_float4(0, 0, 0, 0);
                if (f < F) {
                    const float4* p = reinterpret_cast<const float4*>(table + row * 64 + sub * 4);
                    if (MODE & 4) {
                        v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y);
                        v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w);
                    } else v[u] = *p;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * NS + slot;
                if (f < F) {
                    if (MODE & 1) { S.x += v[u].x; S.y += v[u].y; S.z += v[u].z; S.w += v[u].w; }
                    else if (MODE & 2) {
                        float* q = out_row + f * 64 + sub * 4;
                        __builtin_nontemporal_store(v[u].x, q); __builtin_nontemporal_store(v[u].y, q + 1);
                        __builtin_nontemporal_store(v[u].z, q + 2); __builtin_nontemporal_store(v[u].w, q + 3);
                    } else *reinterpret_cast<float4*>(out_row + f * 64 + sub * 4) = v[u];
                }
            }
        }
    }
    if ((MODE & 1) && S.x + S.y + S.z + S.w == 12345.678f) sink[0] = S.x;
}

// bwd variants. MODE bits: 1 = plain RMW (float4) instead of atomics, 2 = skip concat/sum_x (no FM term)
template <int MODE>
__global__ __launch_bounds__(256) void bwd_var(const int64_t* __restrict__ ids, int64_t B, int F, int64_t V,
                                               const float* __restrict__ d_concat, const float* __restrict__ concat,
                                               int64_t ld, const float* __restrict__ sum_x,
                                               const float* __restrict__ dl_, float scale, float* __restrict__ table) {
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t b = w0; b < B; b += nw) {
        const float dl = dl_[b];
        if (MODE & 1) {
            float4 s = make_float4(0, 0, 0, 0);
            if (!(MODE & 2)) s = *reinterpret_cast<const float4*>(sum_x + b * 64 + sub * 4);
            for (int f = slot; f < F; f += NS) {
                const int64_t row = ids[b * F + f] + (int64_t)f * V;
                float4 g = *reinterpret_cast<const float4*>(d_concat + b * ld + f * 64 + sub * 4);
                if (!(MODE & 2)) {
                    float4 x = *reinterpret_cast<const float4*>(concat + b * ld + f * 64 + sub * 4);
                    g.x += dl * (s.x - x.x); g.y += dl * (s.y - x.y); g.z += dl * (s.z - x.z); g.w += dl * (s.w - x.w);
                }
                float4* p = reinterpret_cast<float4*>(table + row * 64 + sub * 4);
                float4 t = *p;
                t.x += scale * g.x; t.y += scale * g.y; t.z += scale * g.z; t.w += scale * g.w;
                *p = t;
            }
        } else {
            float s[4] = {0, 0, 0, 0};
            if (!(MODE & 2))
                for (int j = 0; j < 4; ++j) s[j] = sum_x[b * 64 + j * LPR + sub];
            for (int f = slot; f < F; f += NS) {
                const int64_t row = ids[b * F + f] + (int64_t)f * V;
                for (int j = 0; j < 4; ++j) {
                    const int d = j * LPR + sub;
                    float g = d_concat[b * ld + f * 64 + d];
                    if (!(MODE & 2)) g += dl * (s[j] - concat[b * ld + f * 64 + d]);
                    unsafeAtomicAdd(table + row * 64 + d, scale * g);
                }
            }
        }
    }
}

extern "C" void exp_fwd(int variant, int grid, const int64_t* ids, int64_t B, int F, const float* table, int64_t V,
                        float* concat, int64_t ld, float* sink, hipStream_t s) {
    switch (variant) {
        case 0: hipLaunchKernelGGL((fwd_var<8, 0>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
        case 1: hipLaunchKernelGGL((fwd_var<8, 1>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
        case 2: hipLaunchKernelGGL((fwd_var<8, 2>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
        case 3: hipLaunchKernelGGL((fwd_var<8, 4>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
        case 4: hipLaunchKernelGGL((fwd_var<8, 6>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
        case 5: hipLaunchKernelGGL((fwd_var<4, 0>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
        case 6: hipLaunchKernelGGL((fwd_var<4, 1>), dim3(grid), dim3(256), 0, s, ids, B, F, table, V, concat, ld, sink); break;
    }
}
extern "C" void exp_bwd(int variant, int grid, const int64_t* ids, int64_t B, int F, int64_t V, const float* d_concat,
                        const float* concat, int64_t ld, const float* sum_x, const float* dl, float scale, float* table,
                        hipStream_t s) {
    switch (variant) {
        case 0: hipLaunchKernelGGL((bwd_var<0>), dim3(grid), dim3(256), 0, s, ids, B, F, V, d_concat, concat, ld, sum_x, dl, scale, table); break;
        case 1: hipLaunchKernelGGL((bwd_var<1>), dim3(grid), dim3(256), 0, s, ids, B, F, V, d_concat, concat, ld, sum_x, dl, scale, table); break;
        case 2: hipLaunchKernelGGL((bwd_var<2>), dim3(grid), dim3(256), 0, s, ids, B, F, V, d_concat, concat, ld, sum_x, dl, scale, table); break;
        case 3: hipLaunchKernelGGL((bwd_var<3>), dim3(grid), dim3(256), 0, s, ids, B, F, V, d_concat, concat, ld, sum_x, dl, scale, table); break;
    }
}

// ---- experiment: first-order weight stored separately (pitch 64 + lin[] gather) vs packed behind the row (pitch 68)
template <int PACKED>
__global__ __launch_bounds__(256) void fwd_lin(const int64_t* __restrict__ ids, int64_t B, int F,
                                               const float* __restrict__ table, const float* __restrict__ lin, int64_t V,
                                               float* __restrict__ concat, int64_t ld, float* __restrict__ out_lin) {
    constexpr int U = 8;
    constexpr int64_t PITCH = PACKED ? 68 : 64;
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t b = w0; b < B; b += nw) {
        int64_t my_row = -1;
        if (lane < F) my_row = ids[b * F + lane] + (int64_t)lane * V;
        float* out_row = concat + b * ld;
        float lacc = 0.f;
        for (int f0 = 0; f0 < F; f0 += NS * U) {
            float4 v[U];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * NS + slot;
                int64_t row = __shfl(my_row, f < F ? f : 0, 64);
                v[u] = make_float4(0, 0, 0, 0);
                w[u] = 0.f;
                if (f < F) {
                    v[u] = *reinterpret_cast<const float4*>(table + row * PITCH + sub * 4);
                    if (sub == 0) w[u] = PACKED ? table[row * PITCH + 64] : lin[row];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * NS + slot;
                if (f < F) {
                    float* q = out_row + f * 64 + sub * 4;
                    __builtin_nontemporal_store(v[u].x, q); __builtin_nontemporal_store(v[u].y, q + 1);
                    __builtin_nontemporal_store(v[u].z, q + 2); __builtin_nontemporal_store(v[u].w, q + 3);
                    lacc += w[u];
                }
            }
        }
        for (int m = 16; m < 64; m <<= 1) lacc += __shfl_xor(lacc, m, 64);
        if (lane == 0) out_lin[b] = lacc;
    }
}
extern "C" void exp_fwd_lin(int packed, int grid, const int64_t* ids, int64_t B, int F, const float* table, const float* lin,
                            int64_t V, float* concat, int64_t ld, float* out_lin, hipStream_t s) {
    if (packed) hipLaunchKernelGGL((fwd_lin<1>), dim3(grid), dim3(256), 0, s, ids, B, F, table, lin, V, concat, ld, out_lin);
    else hipLaunchKernelGGL((fwd_lin<0>), dim3(grid), dim3(256), 0, s, ids, B, F, table, lin, V, concat, ld, out_lin);
}

// ---- EXPERIMENT: cost of N random read-modify-writes of WIDTH bytes (is the first-order weight update expensive
// because it is a partial-line write?).  idx[i] = element index of a 128-byte-aligned line; WIDTH/4 floats touched.
template <int WF, bool READ>
__global__ __launch_bounds__(256) void rmw_var(const int64_t* __restrict__ idx, int64_t n, float* __restrict__ buf) {
    constexpr int LP = WF >= 4 ? WF / 4 : 1;                 // lanes per item (float4 each), or one lane scalar
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = t / LP;
    const int sub = (int)(t % LP);
    if (item >= n) return;
    float* p = buf + idx[item] * 32;                         // line start
    if (WF >= 4) {
        float4* q = reinterpret_cast<float4*>(p) + sub;
        float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
        if (READ) v = *q;
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        *q = v;
    } else {
        float v = READ ? p[0] : 1.f;
        p[0] = v + 1.f;
    }
}

extern "C" void exp_rmw(int width_bytes, int read, const int64_t* idx, int64_t n, float* buf, hipStream_t s) {
#define RM(WF)                                                                                                  \
    {                                                                                                           \
        constexpr int LP = WF >= 4 ? WF / 4 : 1;                                                                \
        const int64_t threads = n * LP;                                                                         \
        const int grid = (int)((threads + 255) / 256);                                                          \
        if (read) hipLaunchKernelGGL((rmw_var<WF, true>), dim3(grid), dim3(256), 0, s, idx, n, buf);            \
        else hipLaunchKernelGGL((rmw_var<WF, false>), dim3(grid), dim3(256), 0, s, idx, n, buf);                \
    }
    switch (width_bytes) {
        case 4: RM(1) break;
        case 16: RM(4) break;
        case 32: RM(8) break;
        case 64: RM(16) break;
        case 128: RM(32) break;
    }
#undef RM
}
