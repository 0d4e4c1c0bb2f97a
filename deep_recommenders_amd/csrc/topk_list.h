// A sorted top-k list (k <= 128) held across the 64 lanes of one wavefront: lane l owns slots l and l + 64.
// Shared by the dense / candidate-list selection kernel (retrieval.hip) and the IVF scan (ivf.hip).
// Order: score descending, ties by index ascending (== a left-to-right scan, == tf.math.top_k), NaN-free input.
#pragma once
#include "dr_common.h"

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

struct List {
    float e0, e1;
    int64_t i0, i1;
    int k, lane;
    bool full;
    float tau;        // score of slot k-1 once the list is full

    __device__ __forceinline__ void refresh() {
        const int s = (k - 1) & 63;
        const float a = __shfl(e0, s, 64), b = __shfl(e1, s, 64);
        const int64_t ia = shfl_i64(i0, s), ib = shfl_i64(i1, s);
        tau = (k - 1) < 64 ? a : b;
        full = ((k - 1) < 64 ? ia : ib) >= 0;
    }
    __device__ __forceinline__ void init(int k_, int lane_) {
        k = k_; lane = lane_;
        e0 = e1 = -INFINITY;
        i0 = i1 = -1;
        refresh();
    }
    __device__ __forceinline__ void load(const float* s, const int64_t* i) {      // s, i: this row's k entries
        if (lane < k) { e0 = s[lane]; i0 = i[lane]; }
        if (lane + 64 < k) { e1 = s[lane + 64]; i1 = i[lane + 64]; }
        refresh();
    }
    __device__ __forceinline__ void store(float* s, int64_t* i) const {
        if (lane < k) { s[lane] = e0; i[lane] = i0; }
        if (lane + 64 < k) { s[lane + 64] = e1; i[lane + 64] = i1; }
    }
    // wave-collective: every lane offers one (score, index) candidate (valid == false: nothing)
    __device__ __forceinline__ void offer(float v, int64_t idx, bool valid) {
        const int p0 = lane, p1 = lane + 64;
        unsigned long long mask = __ballot(valid && (!full || v > tau));
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float cand = __shfl(v, l, 64);
            if (full && !(cand > tau)) continue;
            const int64_t cidx = shfl_i64(idx, l);
            // rank = number of list entries that stay ahead of cand: higher score, or equal score and lower index
            int c = 0;
            if (p0 < k && i0 >= 0 && (e0 > cand || (e0 == cand && i0 < cidx))) ++c;
            if (p1 < k && i1 >= 0 && (e1 > cand || (e1 == cand && i1 < cidx))) ++c;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            const int pos = c;
            if (pos >= k) continue;
            const float pe0 = __shfl_up(e0, 1, 64), pe1 = __shfl_up(e1, 1, 64), w0 = __shfl(e0, 63, 64);
            const int64_t pi0 = shfl_up_i64(i0, 1), pi1 = shfl_up_i64(i1, 1), wi0 = shfl_i64(i0, 63);
            const float n0 = p0 < pos ? e0 : (p0 == pos ? cand : pe0);
            const int64_t ni0 = p0 < pos ? i0 : (p0 == pos ? cidx : pi0);
            const float n1 = p1 < pos ? e1 : (p1 == pos ? cand : (lane == 0 ? w0 : pe1));
            const int64_t ni1 = p1 < pos ? i1 : (p1 == pos ? cidx : (lane == 0 ? wi0 : pi1));
            e0 = n0; i0 = ni0; e1 = n1; i1 = ni1;
            refresh();
        }
    }
};

}  // namespace drtk
