// A sorted top-k list (k <= 128) held across the 64 lanes of one wavefront: lane l owns slots l and l + 64.
// Shared by the dense / candidate-list selection kernel (retrieval.hip) and the IVF scan (ivf.hip).
// Order: score descending, ties by index ascending (== a left-to-right scan, == tf.math.top_k), NaN-free input.
//
// Cross-lane traffic (DR_TOPK_XLANE, round 5).  An insertion is a chain of dependent cross-lane steps; as first written every
// one of them was a ds_bpermute_b32 through the LDS crossbar (26 in the select kernel's loop: broadcast of the candidate, a
// 6-step butterfly for its rank, the shift by one lane, the k-th entry).  None of them needs the crossbar:
//   1 (default): the candidate, the k-th entry and lane 63's spill-over are WAVE-UNIFORM lane reads (v_readlane_b32 into
//      SGPRs); the rank is two ballots + s_bcnt1; the shift by one lane is a DPP wave_shr:1 move (gfx9 DPP control 0x138).
//   0: the ds_bpermute formulation (kept for the A/B: `profiles/r05_topk_xlane.log`).
// Both give the same list bit for bit (the same comparisons in the same order).  Every lane of the wave must be active in
// offer() / refresh() (the callers keep surplus waves alive on a mirrored row instead of masking lanes).
#pragma once
#include "dr_common.h"

#ifndef DR_TOPK_XLANE
#define DR_TOPK_XLANE 1
#endif

namespace drtk {

constexpr int KMAX = 128;

__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
    int lo = __shfl((int)(v & 0xffffffffll), src, 64);
    int hi = __shfl((int)(v >> 32), src, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int d) {
    int lo = __shfl_up((int)(v & 0xffffffffll), d, 64);
    int hi = __shfl_up((int)(v >> 32), d, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
// lane `src` (wave-uniform) of v, as a scalar
__device__ __forceinline__ float lane_f32(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int64_t lane_i64(int64_t v, int src) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int32_t lane_idx(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int64_t lane_idx(int64_t v, int src) { return lane_i64(v, src); }
__device__ __forceinline__ int32_t shfl_idx(int32_t v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int64_t shfl_idx(int64_t v, int src) { return shfl_i64(v, src); }
__device__ __forceinline__ int32_t shfl_up_idx(int32_t v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ int64_t shfl_up_idx(int64_t v, int d) { return shfl_up_i64(v, d); }
// the value of lane - 1 (lane 0: unspecified; its callers never use it)
__device__ __forceinline__ int up1_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float up1_f32(float v) { return __int_as_float(up1_i32(__float_as_int(v))); }
__device__ __forceinline__ int64_t up1_i64(int64_t v) {
    const int lo = up1_i32((int)(v & 0xffffffffll)), hi = up1_i32((int)(v >> 32));
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int32_t up1_idx(int32_t v) { return up1_i32(v); }
__device__ __forceinline__ int64_t up1_idx(int64_t v) { return up1_i64(v); }

// IT: the index type the list carries -- int64_t (arbitrary identifiers: the IVF scan) or int32_t (the exact scan whenever every
// candidate index fits: half the cross-lane moves and selects per insertion; widened when the list is stored)
template <typename IT>
struct ListT {
    float e0, e1;
    IT i0, i1;
    int k, lane;
    bool full;
    float tau;        // score of slot k-1 once the list is full
    IT tau_i;         // ... and its index (-1 while the list is not full)

    __device__ __forceinline__ void refresh() {
        const int s = (k - 1) & 63;
#if DR_TOPK_XLANE
        if ((k - 1) < 64) {                                  // (k is wave-uniform)
            tau = lane_f32(e0, s);
            tau_i = lane_idx(i0, s);
        } else {
            tau = lane_f32(e1, s);
            tau_i = lane_idx(i1, s);
        }
#else
        const float a = __shfl(e0, s, 64), b = __shfl(e1, s, 64);
        const IT ia = shfl_idx(i0, s), ib = shfl_idx(i1, s);
        tau = (k - 1) < 64 ? a : b;
        tau_i = (k - 1) < 64 ? ia : ib;
#endif
        full = tau_i >= 0;
    }
    // can (v, idx) still enter the list?  The list's order is TOTAL (score, then lower index), and so is this gate: a candidate that
    // ties the k-th entry's score enters iff its index is lower, whatever order the candidates arrive in
    __device__ __forceinline__ bool beats_kth(float v, IT idx) const {
        return !full | (v > tau) | ((v == tau) & (idx < tau_i));
    }
    __device__ __forceinline__ void init(int k_, int lane_) {
        k = k_; lane = lane_;
        e0 = e1 = -INFINITY;
        i0 = i1 = -1;
        refresh();
    }
    __device__ __forceinline__ void load(const float* s, const int64_t* i) {      // s, i: this row's k entries
        if (lane < k) { e0 = s[lane]; i0 = (IT)i[lane]; }
        if (lane + 64 < k) { e1 = s[lane + 64]; i1 = (IT)i[lane + 64]; }
        refresh();
    }
    __device__ __forceinline__ void store(float* s, int64_t* i) const {
        if (lane < k) { s[lane] = e0; i[lane] = (int64_t)i0; }
        if (lane + 64 < k) { s[lane + 64] = e1; i[lane + 64] = (int64_t)i1; }
    }
    // wave-collective: every lane offers one (score, index) candidate (valid == false: nothing)
    __device__ __forceinline__ void offer(float v, IT idx, bool valid) {
        const int p0 = lane, p1 = lane + 64;
        unsigned long long mask = __ballot(valid & beats_kth(v, idx));
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
#if DR_TOPK_XLANE
            const float cand = lane_f32(v, l);
            const IT cidx = lane_idx(idx, l);
            if (!beats_kth(cand, cidx)) continue;            // (the list may have moved since the ballot)
            // rank = number of list entries that stay ahead of cand: higher score, or equal score and lower index
            const bool a0 = (p0 < k) & (i0 >= 0) & ((e0 > cand) | ((e0 == cand) & (i0 < cidx)));
            const bool a1 = (p1 < k) & (i1 >= 0) & ((e1 > cand) | ((e1 == cand) & (i1 < cidx)));
            const int pos = __popcll(__ballot(a0)) + __popcll(__ballot(a1));
            if (pos >= k) continue;
            const float pe0 = up1_f32(e0), pe1 = up1_f32(e1), w0 = lane_f32(e0, 63);
            const IT pi0 = up1_idx(i0), pi1 = up1_idx(i1), wi0 = lane_idx(i0, 63);
#else
            const float cand = __shfl(v, l, 64);
            const IT cidx = shfl_idx(idx, l);
            if (!beats_kth(cand, cidx)) continue;
            int c = 0;
            if (p0 < k && i0 >= 0 && (e0 > cand || (e0 == cand && i0 < cidx))) ++c;
            if (p1 < k && i1 >= 0 && (e1 > cand || (e1 == cand && i1 < cidx))) ++c;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            const int pos = c;
            if (pos >= k) continue;
            const float pe0 = __shfl_up(e0, 1, 64), pe1 = __shfl_up(e1, 1, 64), w0 = __shfl(e0, 63, 64);
            const IT pi0 = shfl_up_idx(i0, 1), pi1 = shfl_up_idx(i1, 1), wi0 = shfl_idx(i0, 63);
#endif
            const float n0 = p0 < pos ? e0 : (p0 == pos ? cand : pe0);
            const IT ni0 = p0 < pos ? i0 : (p0 == pos ? cidx : pi0);
            const float n1 = p1 < pos ? e1 : (p1 == pos ? cand : (lane == 0 ? w0 : pe1));
            const IT ni1 = p1 < pos ? i1 : (p1 == pos ? cidx : (lane == 0 ? wi0 : pi1));
            e0 = n0; i0 = ni0; e1 = n1; i1 = ni1;
            refresh();
        }
    }
};

typedef ListT<int64_t> List;

}  // namespace drtk
