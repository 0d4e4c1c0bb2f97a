// EXPERIMENT (not product): the ablation ladder of K4.  Three families, all run on the same tensors in one
// process by k4_ladder.py:
//   M<FEAT>  the round-1 microbenchmark "plain RMW, no FM read" (exp_emb.hip bwd_var<3>, 255 us) with the shipped kernel's
//            ingredients added back one at a time;
//   S        the SHIPPED kernel (emb_sorted.hip is included below, its template is launched directly so that the duplicate
//            blocks at the head of the grid and the block-0 bias sum can be switched off);
//   N<FEAT>  a flat-slot candidate: a wave takes 16 CONSECUTIVE slots of the [B, F] id matrix per iteration (no padded
//            field positions -- the shipped kernel's 2 x 16 field positions per example hold 26 fields), ids prefetched one
//            iteration ahead, slot info handed to the lane groups by DPP row broadcasts.
#include "../../deep_recommenders_amd/csrc/emb_sorted.hip"

namespace {

enum : int { X_LIN = 1, X_FM = 2, X_FLAGS = 4, X_BIAS = 8, X_NTG = 16, X_NTS = 32, X_REV = 64, X_NTT = 128, X_LINW = 256, X_NTL = 512 };   // X_NTL: nontemporal first-order accesses too
// X_NTT: nontemporal table-row loads; X_LINW: the first-order weight's OLD value comes from a dense per-slot array the forward saved
// (lin_old[b * F + f], streamed) -- K4 only WRITES lin_w[row]: one line operation per slot instead of two

typedef float f4v __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ float4 ld4(const float* p) {
    if (NT) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float* p, float4 v) {
    if (NT) {
        f4v w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}

// ---- M: the microbenchmark's loop (one wave per example, a lane group walks fields slot, slot + 4, ...), D = 64 ------------------
template <int FEAT>
__global__ __launch_bounds__(256) void k4_m(const int64_t* __restrict__ ids, const uint8_t* __restrict__ flags, int64_t B, int F,
                                            const int64_t* __restrict__ row_base, const float* __restrict__ grad, int64_t ld,
                                            const float* __restrict__ sum_x, const float* __restrict__ dl_, float scale,
                                            float* __restrict__ table, float* __restrict__ lin_w, float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, slot = lane >> 4, sub = lane & 15;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((FEAT & X_BIAS) && blockIdx.x == 0) dr_block_sum_axpy(dl_, B, scale, bias);
    for (int64_t b0 = w0; b0 < B; b0 += nw) {
        const int64_t b = (FEAT & X_REV) ? B - 1 - b0 : b0;
        const float dl = dl_[b];
        float4 s = make_float4(0, 0, 0, 0);
        if (FEAT & X_FM) s = *reinterpret_cast<const float4*>(sum_x + b * 64 + sub * 4);
        for (int f = slot; f < F; f += 4) {
            const int64_t id = ids[b * F + f];
            bool ok = id >= 0;
            if (FEAT & X_FLAGS) ok = ok && flags[b * F + f] != 0;
            const int64_t row = row_base[f] + (id >= 0 ? id : 0);
            float4 g = ld4<(FEAT & X_NTG) != 0>(grad + b * ld + f * 64 + sub * 4);
            float* p = table + row * 64 + sub * 4;
            float4 t = *reinterpret_cast<const float4*>(p);
            float lw = 0.f;
            if (FEAT & X_LIN) lw = lin_w[row];
            if (FEAT & X_FM) {
                g.x += dl * (s.x - t.x); g.y += dl * (s.y - t.y); g.z += dl * (s.z - t.z); g.w += dl * (s.w - t.w);
            }
            t.x = fmaf(scale, g.x, t.x); t.y = fmaf(scale, g.y, t.y); t.z = fmaf(scale, g.z, t.z); t.w = fmaf(scale, g.w, t.w);
            if (ok) {
                st4<(FEAT & X_NTS) != 0>(p, t);
                if ((FEAT & X_LIN) && sub == 0) lin_w[row] = fmaf(scale, dl, lw);
            }
        }
    }
}

// ---- N: flat slots ---------------------------------------------------------------------------------------------------------------
// Slot s = b * F + f (single-valued fields).  A wave iteration = 16 consecutive slots; lane (grp, sub) prepares slot
// (sub & 3) * 4 + grp of the chunk, so that DPP row_newbcast:u hands lane group grp the info of slot u * 4 + grp: the four
// groups of load instruction u then cover four consecutive slots = 1 KB of contiguous gradient.
__device__ __forceinline__ uint32_t bcast16(uint32_t v, int u) {
    switch (u) {
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150, 0xf, 0xf, false);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x151, 0xf, 0xf, false);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x152, 0xf, 0xf, false);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x153, 0xf, 0xf, false);
    }
}

template <int FEAT>
__global__ __launch_bounds__(256) void k4_n(const int64_t* __restrict__ ids, const uint8_t* __restrict__ flags, int64_t B, int F,
                                            const int64_t* __restrict__ row_base, const float* __restrict__ grad, int64_t ld,
                                            const float* __restrict__ sum_x, const float* __restrict__ dl_, float scale,
                                            float* __restrict__ table, float* __restrict__ lin_w, float* __restrict__ bias,
                                            const float* __restrict__ lin_old) {
    constexpr int U = 4, SPW = 16, D = 64;
    constexpr bool NTG = (FEAT & X_NTG) != 0, NTS = (FEAT & X_NTS) != 0, NTT = (FEAT & X_NTT) != 0;
    const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
    const int64_t nwaves = (int64_t)gridDim.x * 4, wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((FEAT & X_BIAS) && blockIdx.x == 0) dr_block_sum_axpy(dl_, B, scale, bias);
    const int64_t n = B * F, nchunks = (n + SPW - 1) / SPW;
    const int mine = (sub & 3) * 4 + grp;                    // which slot of a chunk this lane prepares
    // (b, f) of this lane's slot advance by a constant per iteration: one division per kernel
    const int64_t stride = nwaves * SPW;
    const uint32_t dq = (uint32_t)(stride / F), dr = (uint32_t)(stride % F);
    int64_t c = wave0;
    if (c >= nchunks) return;
    uint32_t b, f;
    {
        const int64_t s0 = c * SPW + mine;
        b = (uint32_t)(s0 / F);
        f = (uint32_t)(s0 - (int64_t)b * F);
    }
    auto prep = [&](int64_t chunk, uint32_t bb, uint32_t ff, uint32_t& row32, uint32_t& st, uint32_t& goff, uint32_t& bidx, float& lo) {
        int64_t s = chunk * SPW + mine;
        if (FEAT & X_REV) s = (nchunks - 1 - chunk) * SPW + mine;
        const bool in = s < n;
        const int64_t sc = in ? s : n - 1;
        if (FEAT & X_REV) { bb = (uint32_t)(sc / F); ff = (uint32_t)(sc - (int64_t)bb * F); }
        else if (!in) { bb = (uint32_t)(B - 1); ff = (uint32_t)(F - 1); }
        const int64_t id = ids[sc];
        bool ok = in && id >= 0;
        if (FEAT & X_FLAGS) ok = ok && flags[sc] != 0;
        row32 = (uint32_t)(row_base[ff] + (id >= 0 ? id : 0));       // always a readable row; `st` says whether it is ours to write
        st = ok ? 1u : 0u;
        goff = bb * (uint32_t)ld + ff * D;
        bidx = bb;
        lo = 0.f;
        if (FEAT & X_LINW) lo = lin_old[sc];
    };
    uint32_t row32, st, goff, bidx;
    float lold;
    prep(c, b, f, row32, st, goff, bidx, lold);
    for (; c < nchunks; c += nwaves) {
        // next chunk's slot info: loads issued now, consumed at the loop bottom
        uint32_t nb = b + dq, nf = f + dr;
        if (nf >= (uint32_t)F) { nf -= F; ++nb; }
        const int64_t cn = c + nwaves < nchunks ? c + nwaves : c;
        uint32_t n_row32, n_st, n_goff, n_bidx;
        float n_lold;
        prep(cn, cn == c ? b : nb, cn == c ? f : nf, n_row32, n_st, n_goff, n_bidx, n_lold);
        float4 g[U], t[U], sx[U];
        float lw[U], dl[U];
        uint32_t r[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = bcast16(row32, u);
            w[u] = bcast16(st, u);
            const uint32_t go = bcast16(goff, u), bb = bcast16(bidx, u);
            g[u] = ld4<NTG>(grad + (int64_t)go + sub * 4);
            t[u] = ld4<NTT>(table + (int64_t)r[u] * D + sub * 4);
            if (FEAT & X_LINW) lw[u] = __uint_as_float(bcast16(__float_as_uint(lold), u));
            else if (FEAT & X_LIN) lw[u] = (FEAT & X_NTL) ? __builtin_nontemporal_load(lin_w + r[u]) : lin_w[r[u]];
            dl[u] = dl_[bb];
            if (FEAT & X_FM) sx[u] = *reinterpret_cast<const float4*>(sum_x + (int64_t)bb * D + sub * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 x = t[u];
            if (FEAT & X_FM) {
                g[u].x += dl[u] * (sx[u].x - x.x); g[u].y += dl[u] * (sx[u].y - x.y);
                g[u].z += dl[u] * (sx[u].z - x.z); g[u].w += dl[u] * (sx[u].w - x.w);
            }
            x.x = fmaf(scale, g[u].x, x.x); x.y = fmaf(scale, g[u].y, x.y);
            x.z = fmaf(scale, g[u].z, x.z); x.w = fmaf(scale, g[u].w, x.w);
            if (w[u]) {
                st4<NTS>(table + (int64_t)r[u] * D + sub * 4, x);
                if ((FEAT & X_LIN) && sub == 0) {
                    if (FEAT & X_NTL) __builtin_nontemporal_store(fmaf(scale, dl[u], lw[u]), lin_w + r[u]);
                    else lin_w[r[u]] = fmaf(scale, dl[u], lw[u]);
                }
            }
        }
        row32 = n_row32; st = n_st; goff = n_goff; bidx = n_bidx; lold = n_lold;
        b = nb; f = nf;
    }
}

}  // namespace

#define ARGS ids, flags, B, F, row_base, grad, ld, sum_x, dl, scale, table, lin_w, bias
#define L_M(FEAT) case FEAT: hipLaunchKernelGGL((k4_m<FEAT>), dim3(grid), dim3(256), 0, s, ARGS); break;
#define L_N(FEAT) case FEAT: hipLaunchKernelGGL((k4_n<FEAT>), dim3(grid), dim3(256), 0, s, ARGS, lin_old); break;

extern "C" int exp_k4_m(int feat, int grid, const int64_t* ids, const uint8_t* flags, int64_t B, int F, const int64_t* row_base,
                        const float* grad, int64_t ld, const float* sum_x, const float* dl, float scale, float* table,
                        float* lin_w, float* bias, hipStream_t s) {
    switch (feat) {
        L_M(0) L_M(1) L_M(2) L_M(3) L_M(4) L_M(7) L_M(15) L_M(16) L_M(32) L_M(48) L_M(31) L_M(63) L_M(64) L_M(127)
        default: return -1;
    }
    return 0;
}
extern "C" int exp_k4_n(int feat, int grid, const int64_t* ids, const uint8_t* flags, int64_t B, int F, const int64_t* row_base,
                        const float* grad, int64_t ld, const float* sum_x, const float* dl, float scale, float* table,
                        float* lin_w, float* bias, const float* lin_old, hipStream_t s) {
    switch (feat) {
        L_N(0) L_N(1) L_N(2) L_N(3) L_N(4) L_N(7) L_N(15) L_N(16) L_N(32) L_N(48) L_N(31) L_N(47) L_N(63) L_N(64) L_N(79) L_N(95) L_N(127)
        L_N(128) L_N(176) L_N(143) L_N(191) L_N(255) L_N(271) L_N(287) L_N(303) L_N(447) L_N(511) L_N(431) L_N(495) L_N(703) L_N(959)
        default: return -1;
    }
    return 0;
}

// the shipped kernel, launched directly: grid_d duplicate blocks at the head of the grid (0: none), bias optional
extern "C" int exp_k4_s(int grid_u, int grid_d, const int64_t* ids, const uint8_t* flags, int64_t B, int F, const int64_t* row_base,
                        const float* grad, int64_t ld, const float* sum_x, const float* dl, float scale, float* table,
                        float* lin_w, float* bias, const int64_t* rows, const int32_t* slots, const int32_t* dup_heads,
                        const int32_t* dup_count, int64_t num_rows, float* x_sorted, hipStream_t s) {
    const BwdSortedArgs ba{ids, flags, B, F, row_base, 64, grad, ld, nullptr, 0, sum_x, dl, nullptr,
                           scale, table, lin_w, bias, reinterpret_cast<const uint64_t*>(rows), slots, B * F, dup_heads, dup_count,
                           (uint64_t)num_rows, x_sorted, x_sorted != nullptr ? 1 : 0, 0};
    AdamArgs ad{};
    hipLaunchKernelGGL((emb_bwd_sorted_kernel<16, 4, false>), dim3(grid_d + grid_u), dim3(256), 0, s, ba, ad, grid_d);
    return 0;
}
