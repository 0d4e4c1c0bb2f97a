// K7 / K8 — fp32 GEMMs on the gfx950 matrix cores with the layer's elementwise tail fused into the epilogue.  Inputs,
// outputs and accumulators are fp32 (the reference's parity bar is 1e-5 relative on the loss: no reduced-precision
// path).  Two ways of forming the products, same results to fp32 rounding (dr_set_gemm_mode, include/dr_hotpath.h):
//   native   v_mfma_f32_32x32x2_f32
//   bf16x3   every operand value split exactly into three bf16 terms on its way into LDS, six
//            v_mfma_f32_32x32x16_bf16 products per fp32 product (the fp32 MFMA runs at 1/16 of the bf16 rate, so this
//            has a 2.65x higher ceiling); default for the wide-tile tower / cross GEMMs
// Entry points:
//   fwd     y   = act(x @ W + b)                       (Dense: keras deepfm.py:30-34, estimator dnn.py:17-29)
//   cross   out = x0 * (x @ W + b + diag*x) + x        (Cross.call: keras dcn.py:81-88)
//   bwd_dx  dx  = (dy @ W^T) * (relu_src > 0) [+ dx]   (autodiff of the above)
//   bwd_dw  dst += scale * x^T @ dy, dstb += scale*colsum(dy)   (split over the batch, fp32 atomics)
//
// One kernel template.  C[i][j] = sum_r A(i,r) * B(r,j), block tile 128 x 128 x 32, 4 waves in a
// 2 x 2 arrangement, each wave owns 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).
// Operands are staged HBM -> registers -> LDS with the next tile's global loads issued before the
// current tile's MFMAs (register double buffering).  The LDS image is always [r][i] (reduction-major):
//   - an operand whose memory layout is reduction-contiguous (a[i*ld + r], "RC": x in fwd, dy and W
//     in bwd_dx) is transposed on the way in: float4 global loads along r, four ds_write_b32 with row
//     pitch 129 floats (129 % 32 == 1 makes the 4 x 8 (i, r4) lanes of a write group hit 32 banks);
//   - an operand already reduction-major (a[r*ld + i]: W in fwd, x and dy in bwd_dw) goes in with
//     ds_write_b128 at pitch 132 floats.
// Fragment reads are ds_read_b32 of 32 consecutive floats per half-wave: conflict-free in both cases.
// (bf16x3 mode keeps three bf16 planes [i][k] instead, see put4_bf3 below.)
// Consecutive workgroup ids are remapped so that the tiles sharing an A row-panel run on the same XCD
// (same L2): dispatch places block b on XCD b % 8.
#include "dr_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;   // wide configuration; the narrow one is 128 x 32 (template NARROW)
constexpr int LD_T = 129;   // pitch of a transposed-in operand tile
constexpr int LD_D = 132;   // pitch of a direct operand tile

enum Epi { EPI_BIAS_ACT = 0, EPI_CROSS = 1, EPI_MASK = 2, EPI_ATOMIC = 3, EPI_FMGRAD = 4, EPI_LSE = 5, EPI_SMGRAD = 6, EPI_HEAD = 7, EPI_FILTER = 8 };
constexpr int HEAD_PART = 34;   // per-block partials of the fused tower head: dw2[32], db2, loss

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    int64_t M;      // rows of C (i)
    int32_t N;      // cols of C (j)
    int64_t R;      // reduction length
    float* C; int64_t ldc;
    // epilogue operands
    const float* bias;        // [N]            (BIAS_ACT, CROSS)
    int32_t act;              // 0 / 1          (BIAS_ACT)
    const float* e0; int64_t lde0;   // CROSS: x0 ; MASK: relu_src
    const float* e1; int64_t lde1;   // CROSS: x
    float* aux; int64_t ldaux;       // CROSS: prod_out (may be null)
    float alpha;              // CROSS: diag_scale ; ATOMIC: scale
    int32_t accumulate;       // MASK: add to existing C
    float* colsum_dst;        // ATOMIC: dstb (may be null)
    int32_t split;            // ATOMIC: number of reduction splits (gridDim.y)
    int64_t per;              // ATOMIC: reduction rows per split (multiple of BK)
    float* partial;           // ATOMIC: if non-null, block (tile, y) stores its tile to partial[y][M][N] instead of atomics
    int32_t a_vec, b_vec;     // operand base 16-B aligned and ld % 4 == 0 -> float4 loads allowed
    // FMGRAD: C = acc + dl[i] * (S[i][j % fm_D] - x[i][j]) for j < fm_FD   (e0 = x, e1 = S [M, fm_D])
    const float* vec; int32_t fm_D, fm_FD;
    // LSE / SMGRAD (in-batch softmax, Retrieval.call): score s_ij = (acc - log p_j + dupmask_ij * MIN_FLOAT) * inv_t
    const float* cand_prob;       // [N] or null
    const int64_t* cand_ids;      // [N] or null (N == M)
    float inv_t;
    float* part_m; float* part_l; // LSE: partial row max / sum-exp, [2*tiles_n][M]
    float* pos;                   // LSE: s_ii
    const float* lse;             // SMGRAD: row log-sum-exp ; vec = sample_weight (or null) ; alpha = d_loss
    // HEAD (narrow tile only): y = act(acc + bias) is the last hidden layer [M, N<=32]; logit = y . head_w + head_b + extra;
    // loss / gradient per example (dr_bce_terms), d_h = d_logit * head_w * act'(y); C (h itself) optional
    const float* head_w; int64_t ld_head_w;
    const float* head_b;
    const float* head_extra;      // [M] or null (the FM logit)
    const float* labels;          // [M]
    int32_t loss_mode;
    float inv_n;
    float* prob; float* d_logit;  // [M] (either may be null)
    float* d_h; int64_t ld_dh;    // [M, N] or null
    float* head_partial;          // [gridDim.x][HEAD_PART]
    // FILTER (top-K scan): a score is kept only if it beats its row's current k-th best `tau[row]`; kept scores are
    // appended to the row's candidate list (one atomic per 32-column group that has any) instead of writing C
    const float* tau;             // [M]
    float* cand_s; int32_t* cand_c;   // [M][cand_cap] scores / column numbers
    int32_t* cand_cnt;            // [M] append cursors (may exceed cand_cap: the consumer clamps; cap == N never overflows)
    int64_t cand_cap;
};

// exp() of a non-positive softmax argument; masked logits sit at ~-5e36 (MIN_FLOAT / temperature), far outside the
// range the libm range reduction is exact for, so anything below -87 (exp < FLT_MIN) is taken as exactly 0
__device__ __forceinline__ float safe_exp(float d) { return d < -87.f ? 0.f : expf(d); }

// bijective XCD-aware remap (cdna guide T1): consecutive logical tile ids -> same XCD
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Branch-free edge-safe float4 load of an operand tile element (row, col..col+3) of a [nrows, ncols] matrix with
// pitch ld (ncols >= 4).  Out-of-range rows are clamped to the last row, a vector that would run past the last
// valid column is shifted left so that it ends exactly at ncols (a dword-aligned, possibly 16-byte-unaligned
// global_load_dwordx4), and the components are rotated back / zeroed with selects.  No control flow: hipcc keeps
// all eight loads of a k-tile in flight across the MFMA block (with exec-mask branches it drains them first).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
struct EdgeFix { int shift; bool ok; };
__device__ __forceinline__ EdgeFix edge_of(int64_t row, int64_t nrows, int64_t col, int64_t ncols) {
    const int64_t over = col + 4 - ncols;
    EdgeFix e;
    e.shift = over <= 0 ? 0 : (over >= 4 ? 4 : (int)over);
    e.ok = row < nrows;
    return e;
}
// raw load at the clamped address (no dependence on the loaded data -> stays in flight)
__device__ __forceinline__ f4u ld4_raw(const float* __restrict__ p, int64_t ld, int64_t row, int64_t nrows, int64_t col,
                                       int64_t ncols) {
    const int64_t rr = row < nrows ? row : nrows - 1;
    const int64_t over = col + 4 - ncols;
    const int shift = over <= 0 ? 0 : (over >= 4 ? 4 : (int)over);
    const int64_t cc = (col < ncols ? col : ncols) - shift;
    return *reinterpret_cast<const f4u*>(p + rr * ld + cc);
}
// applied when the tile is written to LDS, i.e. after the MFMA block the load was hidden under
__device__ __forceinline__ float4 fix4(f4u v, EdgeFix e) {
    float4 o;
    const int s = e.shift;
    o.x = !e.ok ? 0.f : (s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : s == 3 ? v.w : 0.f);
    o.y = !e.ok ? 0.f : (s == 0 ? v.y : s == 1 ? v.z : s == 2 ? v.w : 0.f);
    o.z = !e.ok ? 0.f : (s == 0 ? v.z : s == 1 ? v.w : 0.f);
    o.w = !e.ok ? 0.f : (s == 0 ? v.w : 0.f);
    return o;
}

// ---- fp32 product emulation on the bf16 matrix pipe ("bf16x3", 6 of the 9 cross products) --------------------------
// x = x0 + x1 + x2 with x0 = bf16_rn(x), x1 = bf16_rn(x - x0), x2 = bf16_rn(x - x0 - x1): |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|,
// the two subtractions are exact in fp32.  a * b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0); every bf16 x bf16 product
// is exact in the MFMA's fp32 accumulator and the dropped terms (a1b2, a2b1, a2b2) are below 2^-24 |ab|.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// LDS image of an operand tile in bf16x3 mode: three planes [rows][PK] of bf16, k-contiguous.  PK = 40 (80-byte rows): the
// ds_read_b128 fragment reads (lane -> row lane & 31, 16 bytes) are conflict-free for that instruction's 16-lane groups
// (20 * row mod 64 is a permutation of the 4-bank slots over each group, MI355X_MICROARCH.md LDS table).
constexpr int PK = 40;
// four k-consecutive fp32 values of one row -> 4 bf16 in each plane (one ds_write_b64 per plane)
__device__ __forceinline__ void put4_bf3(__bf16* __restrict__ plane0, int plane_stride, int row, int k, float v0, float v1,
                                         float v2, float v3) {
    const f32x2 a = {v0, v1}, b = {v2, v3};
    const bf16x2 a0 = __builtin_convertvector(a, bf16x2), b0 = __builtin_convertvector(b, bf16x2);
    const f32x2 ra = a - __builtin_convertvector(a0, f32x2), rb = b - __builtin_convertvector(b0, f32x2);
    const bf16x2 a1 = __builtin_convertvector(ra, bf16x2), b1 = __builtin_convertvector(rb, bf16x2);
    const f32x2 sa = ra - __builtin_convertvector(a1, f32x2), sb = rb - __builtin_convertvector(b1, f32x2);
    const bf16x2 a2 = __builtin_convertvector(sa, bf16x2), b2 = __builtin_convertvector(sb, bf16x2);
    __bf16* d = plane0 + row * PK + k;
    *reinterpret_cast<bf16x4*>(d) = bf16x4{a0[0], a0[1], b0[0], b0[1]};
    *reinterpret_cast<bf16x4*>(d + plane_stride) = bf16x4{a1[0], a1[1], b1[0], b1[1]};
    *reinterpret_cast<bf16x4*>(d + 2 * plane_stride) = bf16x4{a2[0], a2[1], b2[0], b2[1]};
}

// Occupancy: 3 blocks per CU for the wide tile (168 VGPRs).  At 4 (128 VGPRs) the next k-tile's 8 prefetch registers
// cannot stay live across the MFMA block without spilling, so the compiler sinks the global loads BELOW the 64 MFMAs
// and their latency is exposed in front of every barrier; pinned ahead of the MFMAs at 3 blocks/CU is 2-4 % faster.
template <bool A_RC, bool B_RC, int EPI, bool NARROW, bool OCC4 = false, bool BF3 = false>
__global__ __launch_bounds__(256, BF3 ? 2 : ((NARROW || OCC4) ? 4 : 3)) void gemm_f32_mfma_kernel(GemmArgs g) {
    // wide: 2 x 2 waves, each 2 x 2 MFMA tiles (128 x 128);  narrow: 4 x 1 waves, each 1 x 1 tile (128 x 32)
    constexpr int BN = NARROW ? 32 : 128;
    constexpr int TM = NARROW ? 1 : 2, TN = NARROW ? 1 : 2;
    constexpr int RCS = 32;                                  // rows between a thread's float4's of a reduction-contiguous operand
    constexpr int NQB = NARROW ? 1 : 4;                      // float4's of the B tile per thread
    constexpr int LDA = A_RC ? LD_T : LD_D;
    constexpr int LDB = B_RC ? (NARROW ? 33 : LD_T) : (NARROW ? 36 : LD_D);
    static_assert(!(BF3 && NARROW), "bf16x3 mode uses the wide tile");
    constexpr int PLANE_A = BM * PK, PLANE_B = BN * PK;       // bf16 elements per plane (bf16x3 mode)
    constexpr int BF3_BUF = 3 * (PLANE_A + PLANE_B);          // bf16 elements of the tile image
    constexpr int SMEM_FLOATS = BF3 ? BF3_BUF / 2 : BK * LDA + BK * LDB;       // bf16x3: 60 KB, two blocks per CU
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* As = smem;
    float* Bs = smem + (BF3 ? 0 : BK * LDA);
    __bf16* const Ap = reinterpret_cast<__bf16*>(smem);       // bf16x3: [3][BM][PK] then [3][BN][PK]
    __bf16* const Bp = Ap + 3 * PLANE_A;
    __bf16* const ApW = Ap; __bf16* const BpW = Bp;
    const __bf16* const ApR = Ap; const __bf16* const BpR = Bp;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = NARROW ? wave : (wave >> 1), wn = NARROW ? 0 : (wave & 1);

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int nwg = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int64_t m0 = (int64_t)(lid / tiles_n) * BM;
    const int n0 = (lid % tiles_n) * BN;

    // reduction range of this block (split-K only for EPI_ATOMIC)
    int64_t r_begin = 0, r_end = g.R;
    if (EPI == EPI_ATOMIC) {
        const int64_t per = g.per;                                // host-computed: every launched slice is non-empty
        r_begin = (int64_t)blockIdx.y * per;
        r_end = r_begin + per < g.R ? r_begin + per : g.R;
        if (r_begin >= r_end) return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[a][b][k] = 0.f;

    // per-thread coordinates of its float4's.  RC operand: (i = tid>>3 (+32q), r4 = tid&7); reduction-major operand, wide:
    // (r = tid>>5 (+8q), c4 = tid&31), narrow B (32 columns): (r = tid>>3, c4 = tid&7), one float4 per thread.
    const int rc_i = tid >> 3, rc_r4 = tid & 7;
    // bf16x3 mode, reduction-major operand: each thread owns a 4 (k) x 4 (i) block so that it can write k-contiguous bf16
    // quads; a 16-lane group spans 4 column-quads x 4 k-quads (64-byte global segments, 2-way LDS store conflicts at most)
    const int t_kq = (tid >> 7) * 4 + ((tid & 15) >> 2), t_c4 = ((tid >> 4) & 7) * 4 + (tid & 3);
    const int dr_r0 = BF3 ? 4 * t_kq : (tid >> 5), dr_rs = BF3 ? 1 : 8;     // row of float4 q: dr_r0 + dr_rs * q
    const int dr_c4 = BF3 ? t_c4 : (tid & 31);
    const int nb_r = tid >> 3, nb_c4 = tid & 7;
    f4u va[4], vb[NQB];
    auto load_tiles = [&](f4u (&va)[4], f4u (&vb)[NQB], int64_t r0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (A_RC) va[q] = ld4_raw(g.A, g.lda, m0 + rc_i + RCS * q, g.M, r0 + rc_r4 * 4, r_end);
            else      va[q] = ld4_raw(g.A, g.lda, r0 + dr_r0 + dr_rs * q, r_end, m0 + dr_c4 * 4, g.M);
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            if (B_RC) vb[q] = ld4_raw(g.B, g.ldb, (int64_t)n0 + rc_i + RCS * q, g.N, r0 + rc_r4 * 4, r_end);
            else if (NARROW) vb[q] = ld4_raw(g.B, g.ldb, r0 + nb_r, r_end, (int64_t)n0 + nb_c4 * 4, g.N);
            else      vb[q] = ld4_raw(g.B, g.ldb, r0 + dr_r0 + dr_rs * q, r_end, (int64_t)n0 + dr_c4 * 4, g.N);
        }
    };
    auto store_tiles = [&](const f4u (&va)[4], const f4u (&vb)[NQB], int64_t r0) {     // r0 = offset the set was loaded for
        if constexpr (BF3) {
            float4 ta[4], tb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (A_RC) ta[q] = fix4(va[q], edge_of(m0 + rc_i + RCS * q, g.M, r0 + rc_r4 * 4, r_end));
                else      ta[q] = fix4(va[q], edge_of(r0 + dr_r0 + dr_rs * q, r_end, m0 + dr_c4 * 4, g.M));
                if (B_RC) tb[q] = fix4(vb[q], edge_of((int64_t)n0 + rc_i + RCS * q, g.N, r0 + rc_r4 * 4, r_end));
                else      tb[q] = fix4(vb[q], edge_of(r0 + dr_r0 + dr_rs * q, r_end, (int64_t)n0 + dr_c4 * 4, g.N));
            }
            if (A_RC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) put4_bf3(ApW, PLANE_A, rc_i + RCS * q, rc_r4 * 4, ta[q].x, ta[q].y, ta[q].z, ta[q].w);
            } else {
                put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 0, dr_r0, ta[0].x, ta[1].x, ta[2].x, ta[3].x);
                put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 1, dr_r0, ta[0].y, ta[1].y, ta[2].y, ta[3].y);
                put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 2, dr_r0, ta[0].z, ta[1].z, ta[2].z, ta[3].z);
                put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 3, dr_r0, ta[0].w, ta[1].w, ta[2].w, ta[3].w);
            }
            if (B_RC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) put4_bf3(BpW, PLANE_B, rc_i + RCS * q, rc_r4 * 4, tb[q].x, tb[q].y, tb[q].z, tb[q].w);
            } else {
                put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 0, dr_r0, tb[0].x, tb[1].x, tb[2].x, tb[3].x);
                put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 1, dr_r0, tb[0].y, tb[1].y, tb[2].y, tb[3].y);
                put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 2, dr_r0, tb[0].z, tb[1].z, tb[2].z, tb[3].z);
                put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 3, dr_r0, tb[0].w, tb[1].w, tb[2].w, tb[3].w);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (A_RC) {
                const float4 t = fix4(va[q], edge_of(m0 + rc_i + RCS * q, g.M, r0 + rc_r4 * 4, r_end));
                float* d = As + (rc_r4 * 4) * LDA + rc_i + RCS * q;
                d[0] = t.x; d[LDA] = t.y; d[2 * LDA] = t.z; d[3 * LDA] = t.w;
            } else {
                const float4 t = fix4(va[q], edge_of(r0 + dr_r0 + dr_rs * q, r_end, m0 + dr_c4 * 4, g.M));
                *reinterpret_cast<float4*>(&As[(dr_r0 + dr_rs * q) * LDA + dr_c4 * 4]) = t;
            }
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            if (B_RC) {
                const float4 t = fix4(vb[q], edge_of((int64_t)n0 + rc_i + RCS * q, g.N, r0 + rc_r4 * 4, r_end));
                float* d = Bs + (rc_r4 * 4) * LDB + rc_i + RCS * q;
                d[0] = t.x; d[LDB] = t.y; d[2 * LDB] = t.z; d[3 * LDB] = t.w;
            } else if (NARROW) {
                const float4 t = fix4(vb[q], edge_of(r0 + nb_r, r_end, (int64_t)n0 + nb_c4 * 4, g.N));
                *reinterpret_cast<float4*>(&Bs[nb_r * LDB + nb_c4 * 4]) = t;
            } else {
                const float4 t = fix4(vb[q], edge_of(r0 + dr_r0 + dr_rs * q, r_end, (int64_t)n0 + dr_c4 * 4, g.N));
                *reinterpret_cast<float4*>(&Bs[(dr_r0 + dr_rs * q) * LDB + dr_c4 * 4]) = t;
            }
        }
    };

    float colsum = 0.f;   // EPI_ATOMIC: column sums of B (dy) accumulated by the m-tile-0 blocks
    const bool do_colsum = (EPI == EPI_ATOMIC) && g.colsum_dst != nullptr && m0 == 0 && tid < BN;

    const float* as = As + (lane >> 5) * LDA + wm * (TM * 32) + (lane & 31);
    const float* bs = Bs + (lane >> 5) * LDB + wn * (TN * 32) + (lane & 31);
    // bf16x3 mode: column sums of the B tile from its three planes (x0 + x1 + x2 == x up to 2^-24)
    auto bf3_colsum = [&]() {
        if (do_colsum) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int c = 0; c < BK / 8; ++c) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(BpR + pl * PLANE_B + tid * PK + 8 * c);
#pragma unroll
                    for (int j = 0; j < 8; ++j) colsum += (float)v[j];
                }
        }
    };
    auto mfma_block = [&]() {
        if constexpr (BF3) {
            bf3_colsum();
        } else {
            if (do_colsum) {
#pragma unroll 8
                for (int r = 0; r < BK; ++r) colsum += Bs[r * LDB + tid];
            }
        }
        if constexpr (BF3) {
            // fragment of a 32 x 16 sub-tile: lane (row = lane & 31, kg = lane >> 5) takes k = k0 + 8 * kg + 0..7, one ds_read_b128 per plane
            const __bf16* ap = ApR + (wm * (TM * 32) + (lane & 31)) * PK + 8 * (lane >> 5);
            const __bf16* bp = BpR + (wn * (TN * 32) + (lane & 31)) * PK + 8 * (lane >> 5);
#pragma unroll
            for (int k0 = 0; k0 < BK; k0 += 16) {
                bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) af[pl][t] = *reinterpret_cast<const bf16x8*>(ap + pl * PLANE_A + t * 32 * PK + k0);
#pragma unroll
                    for (int t = 0; t < TN; ++t) bf[pl][t] = *reinterpret_cast<const bf16x8*>(bp + pl * PLANE_B + t * 32 * PK + k0);
                }
                // smallest terms first; the four accumulators are interleaved so that back-to-back MFMAs are independent
#pragma unroll
                for (int term = 0; term < 6; ++term) {
                    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[term]][a], bf[PB[term]][b], acc[a][b], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) af[t] = as[kk * LDA + 32 * t];
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[t] = bs[kk * LDB + 32 * t];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    };
    // ---- lean path for interior blocks: raw pointer-bumped dwordx4 loads, direct LDS stores, no edge logic.
    // (PMC, tools/exp/gemm_only.py: the clamped loader + fix-up costs ~600 SALU/VALU instructions per wave per k-tile,
    // as long as the 64-MFMA block itself; the lean loop issues ~60.)
    // Row / column edges of the OUTPUT tile need no masking here: an operand row (RC) or column (reduction-major)
    // that lies outside the matrix only feeds accumulator rows / columns the epilogue never stores, so its address is
    // simply clamped into the matrix (computed once, outside the loop).  Only the reduction tail needs zero fill.
    const int64_t nfull = (r_end - r_begin) / BK;             // whole k-tiles
    const bool has_tail = r_begin + nfull * BK < r_end;
    // a float4 that straddles the last column of a reduction-major operand is loaded unshifted, which is only legal
    // when the row pitch covers it (padded buffers); a tight pitch sends that edge block down the clamped slow path
    const bool a_tight = !A_RC && (m0 + BM > g.M) && (g.M & 3) && g.lda < ((g.M + 3) & ~(int64_t)3);
    const bool b_tight = !B_RC && (n0 + BN > g.N) && (g.N & 3) && g.ldb < (((int64_t)g.N + 3) & ~(int64_t)3);
    if (nfull > 0 && !a_tight && !b_tight) {
        const float* pa[4];
        const float* pb[NQB];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (A_RC) {
                int64_t row = m0 + rc_i + RCS * q;
                row = row < g.M ? row : g.M - 1;
                pa[q] = g.A + row * g.lda + r_begin + rc_r4 * 4;
            } else {
                int64_t col = m0 + dr_c4 * 4;
                col = col < g.M ? col : g.M - 4;             // fully outside -> anywhere legal; straddling -> unshifted
                pa[q] = g.A + (r_begin + dr_r0 + dr_rs * q) * g.lda + col;
            }
        }
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
            if (B_RC) {
                int64_t row = (int64_t)n0 + rc_i + RCS * q;
                row = row < g.N ? row : g.N - 1;
                pb[q] = g.B + row * g.ldb + r_begin + rc_r4 * 4;
            } else {
                int64_t col = (int64_t)n0 + (NARROW ? nb_c4 : dr_c4) * 4;
                col = col < g.N ? col : g.N - 4;
                pb[q] = g.B + (r_begin + (NARROW ? nb_r : dr_r0 + dr_rs * q)) * g.ldb + col;
            }
        }
        const int64_t a_it = A_RC ? (int64_t)BK : BK * g.lda;
        const int64_t b_it = B_RC ? (int64_t)BK : BK * g.ldb;
        auto load_fast = [&](f4u (&va)[4], f4u (&vb)[NQB]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                va[q] = *reinterpret_cast<const f4u*>(pa[q]);
                pa[q] += a_it;
            }
#pragma unroll
            for (int q = 0; q < NQB; ++q) {
                vb[q] = *reinterpret_cast<const f4u*>(pb[q]);
                pb[q] += b_it;
            }
        };
        auto store_fast = [&](const f4u (&va)[4], const f4u (&vb)[NQB]) {
            if constexpr (BF3) {
                if (A_RC) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) put4_bf3(ApW, PLANE_A, rc_i + RCS * q, rc_r4 * 4, va[q].x, va[q].y, va[q].z, va[q].w);
                } else {
                    put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 0, dr_r0, va[0].x, va[1].x, va[2].x, va[3].x);
                    put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 1, dr_r0, va[0].y, va[1].y, va[2].y, va[3].y);
                    put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 2, dr_r0, va[0].z, va[1].z, va[2].z, va[3].z);
                    put4_bf3(ApW, PLANE_A, dr_c4 * 4 + 3, dr_r0, va[0].w, va[1].w, va[2].w, va[3].w);
                }
                if (B_RC) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) put4_bf3(BpW, PLANE_B, rc_i + RCS * q, rc_r4 * 4, vb[q].x, vb[q].y, vb[q].z, vb[q].w);
                } else {
                    put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 0, dr_r0, vb[0].x, vb[1].x, vb[2].x, vb[3].x);
                    put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 1, dr_r0, vb[0].y, vb[1].y, vb[2].y, vb[3].y);
                    put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 2, dr_r0, vb[0].z, vb[1].z, vb[2].z, vb[3].z);
                    put4_bf3(BpW, PLANE_B, dr_c4 * 4 + 3, dr_r0, vb[0].w, vb[1].w, vb[2].w, vb[3].w);
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (A_RC) {
                    float* d = As + (rc_r4 * 4) * LDA + rc_i + RCS * q;
                    d[0] = va[q].x; d[LDA] = va[q].y; d[2 * LDA] = va[q].z; d[3 * LDA] = va[q].w;
                } else {
                    *reinterpret_cast<f4u*>(&As[(dr_r0 + dr_rs * q) * LDA + dr_c4 * 4]) = va[q];
                }
            }
#pragma unroll
            for (int q = 0; q < NQB; ++q) {
                if (B_RC) {
                    float* d = Bs + (rc_r4 * 4) * LDB + rc_i + RCS * q;
                    d[0] = vb[q].x; d[LDB] = vb[q].y; d[2 * LDB] = vb[q].z; d[3 * LDB] = vb[q].w;
                } else if (NARROW) {
                    *reinterpret_cast<f4u*>(&Bs[nb_r * LDB + nb_c4 * 4]) = vb[q];
                } else {
                    *reinterpret_cast<f4u*>(&Bs[(dr_r0 + dr_rs * q) * LDB + dr_c4 * 4]) = vb[q];
                }
            }
        };
        if constexpr (BF3) {
            // Two register sets, so that a tile's global loads are issued two k-tiles before they are needed: with one set
            // they have only the 48-MFMA block (~2000 cycles) to land and the store phase waits 1000-2000 cycles for them
            // in most iterations (tools/exp/bf3_phases.py: clock64 timeline of one block).
            f4u wa[4], wb[NQB];
            load_fast(va, vb);                                 // tile 0
            if (nfull > 1) load_fast(wa, wb);                  // tile 1
            int64_t t = 0;
            for (; t + 3 < nfull; t += 2) {
                store_fast(va, vb);                            // tile t
                __syncthreads();
                load_fast(va, vb);                             // tile t + 2
                __builtin_amdgcn_sched_barrier(0);
                mfma_block();
                __syncthreads();
                store_fast(wa, wb);                            // tile t + 1
                __syncthreads();
                load_fast(wa, wb);                             // tile t + 3
                __builtin_amdgcn_sched_barrier(0);
                mfma_block();
                __syncthreads();
            }
            // up to three whole tiles left: t (in va / vb), t + 1 (in wa / wb), t + 2 (not loaded yet), then the tail
            const int64_t left = nfull - t;                    // 1, 2 or 3
            store_fast(va, vb);
            __syncthreads();
            if (left == 3) load_fast(va, vb);
            else if (left == 1 && has_tail) load_tiles(va, vb, r_begin + nfull * BK);
            mfma_block();
            __syncthreads();
            if (left >= 2) {
                store_fast(wa, wb);
                __syncthreads();
                if (left == 2 && has_tail) load_tiles(wa, wb, r_begin + nfull * BK);
                mfma_block();
                __syncthreads();
            }
            if (left == 3) {
                store_fast(va, vb);
                __syncthreads();
                if (has_tail) load_tiles(wa, wb, r_begin + nfull * BK);
                mfma_block();
                __syncthreads();
            }
            if (has_tail) {
                if (left == 1) store_tiles(va, vb, r_begin + nfull * BK); else store_tiles(wa, wb, r_begin + nfull * BK);
                __syncthreads();
                mfma_block();
                __syncthreads();
            }
        } else {
        load_fast(va, vb);
        for (int64_t t = 0; t + 1 < nfull; ++t) {
            store_fast(va, vb);
            __syncthreads();
            load_fast(va, vb);
            if (!OCC4) __builtin_amdgcn_sched_barrier(0);    // keep the global loads AHEAD of the MFMA block (needs > 128 VGPRs)
            mfma_block();
            __syncthreads();
        }
        // last whole tile (+ the clamped tail tile, if any), straight-line
        store_fast(va, vb);
        __syncthreads();
        if (has_tail) load_tiles(va, vb, r_begin + nfull * BK);
        mfma_block();
        __syncthreads();
        if (has_tail) {
            store_tiles(va, vb, r_begin + nfull * BK);
            __syncthreads();
            mfma_block();
            __syncthreads();
        }
        }
    } else {
        load_tiles(va, vb, r_begin);
        for (int64_t r0 = r_begin; r0 < r_end; r0 += BK) {
            store_tiles(va, vb, r0);
            __syncthreads();
            if (r0 + BK < r_end) load_tiles(va, vb, r0 + BK);
            mfma_block();
            __syncthreads();
        }
    }

    if constexpr ((EPI == EPI_LSE || EPI == EPI_SMGRAD) && !NARROW) {
        constexpr float MIN_FLOAT = -3.4028234663852886e36f;   // np.finfo(np.float32).min / 100 (sbcnm.py:10)
        static_assert(!((EPI == EPI_LSE || EPI == EPI_SMGRAD) && NARROW), "softmax epilogues use the wide tile");
        float colcorr[2];
        int64_t colid[2];
        int colj[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            colj[ni] = n0 + wn * 64 + ni * 32 + (lane & 31);
            const bool cv = colj[ni] < g.N;
            colcorr[ni] = (cv && g.cand_prob != nullptr) ? -logf(g.cand_prob[colj[ni]]) : 0.f;
            colid[ni] = (cv && g.cand_ids != nullptr) ? g.cand_ids[colj[ni]] : 0;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t row = m0 + wm * 64 + mi * 32 + 4 * (lane >> 5) + (reg & 3) + 8 * (reg >> 2);
                const bool rv = row < g.M;
                const int64_t rid = (rv && g.cand_ids != nullptr) ? g.cand_ids[row] : 0;
                float sv[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    float v = acc[mi][ni][reg] + colcorr[ni];
                    if (g.cand_ids != nullptr && rid == colid[ni] && row != colj[ni]) v += MIN_FLOAT;
                    sv[ni] = v * g.inv_t;
                }
                if (EPI == EPI_LSE) {
                    float m = -INFINITY;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (colj[ni] < g.N) m = fmaxf(m, sv[ni]);
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                    float l = 0.f;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (colj[ni] < g.N) l += safe_exp(sv[ni] - m);
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) l += __shfl_xor(l, o, 64);
                    if (rv && (lane & 31) == 0) {
                        const int64_t pc = (int64_t)((n0 / BN) * 2 + wn);
                        g.part_m[pc * g.M + row] = m;
                        g.part_l[pc * g.M + row] = l;
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (rv && row == colj[ni]) g.pos[row] = sv[ni];
                } else {
                    if (!rv) continue;
                    const float w = g.vec != nullptr ? g.vec[row] : 1.f;
                    const float lse = g.lse[row];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if (colj[ni] >= g.N) continue;
                        const float pr = safe_exp(sv[ni] - lse) - (row == colj[ni] ? 1.f : 0.f);
                        g.C[row * g.ldc + colj[ni]] = w * pr * g.inv_t * g.alpha;
                    }
                }
            }
        }
        return;
    }
    if constexpr (EPI == EPI_FILTER && !NARROW) {
        const int c31 = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int64_t row = m0 + wm * (TM * 32) + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                const bool rv = row < g.M;
                const float t = rv ? g.tau[row] : INFINITY;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int col = n0 + wn * (TN * 32) + ni * 32 + c31;
                    const float v = acc[mi][ni][reg];
                    const bool pass = rv && col < g.N && v > t;
                    const unsigned half = (unsigned)((__ballot(pass) >> (32 * hh)) & 0xffffffffull);
                    if (half != 0u) {                      // rare once tau has warmed up: one atomic per (row, 32 columns)
                        const int leader = 32 * hh + __ffs((int)half) - 1;
                        int base = 0;
                        if (lane == leader) base = atomicAdd(g.cand_cnt + row, __popc(half));
                        base = __shfl(base, leader, 64);
                        if (pass) {
                            const int64_t pos = base + __popc(half & ((1u << c31) - 1u));
                            if (pos < g.cand_cap) {
                                g.cand_s[row * g.cand_cap + pos] = v;
                                g.cand_c[row * g.cand_cap + pos] = col;
                            }
                        }
                    }
                }
            }
        }
        return;
    }
    if constexpr (EPI == EPI_HEAD && NARROW) {
        // Fused tower head (the Dense(1) that follows the last hidden layer, the loss, and their backward):
        // lane (col, half) holds 16 rows of column col of y; the Dense(1) dot product is a butterfly over the 32 columns.
        const int col = lane & 31, hh = lane >> 5;
        const bool cv = col < g.N;
        const int colc = cv ? col : g.N - 1;
        const float bj = g.bias != nullptr ? g.bias[colc] : 0.f;
        const float wj = cv ? g.head_w[(int64_t)colc * g.ld_head_w] : 0.f;
        const float b2 = g.head_b != nullptr ? g.head_b[0] : 0.f;
        const int64_t row_b = m0 + wm * 32 + 4 * hh;
        float dw_acc = 0.f, db_acc = 0.f, loss_acc = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ro = (reg & 3) + 8 * (reg >> 2);
            const int64_t row = row_b + ro;
            const bool rv = row < g.M;
            const int64_t rc = rv ? row : g.M - 1;
            float v = acc[0][0][reg] + bj;
            if (g.act == 1) v = fmaxf(v, 0.f);
            if (!cv) v = 0.f;
            float dot = v * wj;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) dot += __shfl_xor(dot, o, 64);
            const float x = (dot + b2) + (g.head_extra != nullptr ? g.head_extra[rc] : 0.f);
            float p, l, gr;
            dr_bce_terms(x, g.labels[rc], g.loss_mode, p, l, gr);
            float gs = gr * g.inv_n;
            if (!rv) { l = 0.f; gs = 0.f; }
            if (rv && col == 0) {
                if (g.prob != nullptr) g.prob[row] = p;
                if (g.d_logit != nullptr) g.d_logit[row] = gs;
            }
            if (rv && cv) {
                if (g.d_h != nullptr) g.d_h[row * g.ld_dh + col] = (g.act == 1 && !(v > 0.f)) ? 0.f : gs * wj;
                if (g.C != nullptr) g.C[row * g.ldc + col] = v;
            }
            dw_acc = fmaf(v, gs, dw_acc);
            if (col == 0) { db_acc += gs; loss_acc += l; }
        }
        __syncthreads();                                   // every wave is done with the operand tiles
        float* red = smem;                                 // [8 = wave * 2 + half][HEAD_PART]
        red[(wave * 2 + hh) * HEAD_PART + col] = dw_acc;
        if (col == 0) {
            red[(wave * 2 + hh) * HEAD_PART + 32] = db_acc;
            red[(wave * 2 + hh) * HEAD_PART + 33] = loss_acc;
        }
        __syncthreads();
        if (tid < HEAD_PART) {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sacc += red[i * HEAD_PART + tid];
            g.head_partial[(int64_t)blockIdx.x * HEAD_PART + tid] = sacc;
        }
        return;
    }
    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = n0 + wn * (TN * 32) + ni * 32 + (lane & 31);
            if (!full && col >= g.N) continue;
            float bj = 0.f;
            if ((EPI == EPI_BIAS_ACT || EPI == EPI_CROSS) && g.bias != nullptr) bj = g.bias[col];
            int cmod = 0;
            if (EPI == EPI_FMGRAD) cmod = col % g.fm_D;
            const int64_t row_b = m0 + wm * (TM * 32) + mi * 32 + 4 * (lane >> 5);
            float* cp = g.C + row_b * g.ldc + col;
            const float* e0p = (EPI == EPI_CROSS || EPI == EPI_MASK || EPI == EPI_FMGRAD) && g.e0 != nullptr
                                   ? g.e0 + row_b * g.lde0 + col : nullptr;
            const float* e1p = (EPI == EPI_CROSS) ? g.e1 + row_b * g.lde1 + col
                               : (EPI == EPI_FMGRAD ? g.e1 + row_b * g.lde1 + cmod : nullptr);
            float* auxp = (EPI == EPI_CROSS && g.aux != nullptr) ? g.aux + row_b * g.ldaux + col : nullptr;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ro = (reg & 3) + 8 * (reg >> 2);
                if (!full && row_b + ro >= g.M) continue;
                float v = acc[mi][ni][reg];
                if (EPI == EPI_BIAS_ACT) {
                    v += bj;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    cp[ro * g.ldc] = v;
                } else if (EPI == EPI_CROSS) {
                    const float xv = e1p[ro * g.lde1];
                    const float prod = v + bj + g.alpha * xv;
                    if (auxp != nullptr) auxp[ro * g.ldaux] = prod;
                    cp[ro * g.ldc] = e0p[ro * g.lde0] * prod + xv;
                } else if (EPI == EPI_FMGRAD) {
                    if (col < g.fm_FD) v += g.vec[row_b + ro] * (e1p[ro * g.lde1] - e0p[ro * g.lde0]);
                    cp[ro * g.ldc] = v;
                } else if (EPI == EPI_MASK) {
                    if (e0p != nullptr && !(e0p[ro * g.lde0] > 0.f)) v = 0.f;
                    if (g.accumulate) v += cp[ro * g.ldc];
                    cp[ro * g.ldc] = v;
                } else {
                    if (g.partial != nullptr)
                        g.partial[((int64_t)blockIdx.y * g.M + row_b + ro) * g.N + col] = v;
                    else
                        unsafeAtomicAdd(cp + ro * g.ldc, g.alpha * v);
                }
            }
        }
    }
    if (do_colsum && n0 + tid < g.N) unsafeAtomicAdd(g.colsum_dst + n0 + tid, g.alpha * colsum);
}


// process-wide GEMM mode (dr_set_gemm_mode); the default can be overridden with DR_GEMM_MODE=native|bf16x3
static int gemm_mode_default() {
    const char* e = getenv("DR_GEMM_MODE");
    if (e != nullptr && (e[0] == 'n' || e[0] == 'N' || e[0] == '1')) return DR_GEMM_NATIVE_F32;
    return DR_GEMM_BF16X3;
}
static std::atomic<int> g_gemm_mode{gemm_mode_default()};
// process-wide operand split of the register-split GEMMs (dr_set_gemm_split): THE one parser of DR_GEMM_SPLIT -- f16x2 iff the
// variable is unset or spells exactly "f16x2"; anything else is the six-product bf16x3 split
static int gemm_split_default() {
    const char* e = getenv("DR_GEMM_SPLIT");
    return (e == nullptr || strcmp(e, "f16x2") == 0) ? DR_GEMM_SPLIT_F16X2 : DR_GEMM_SPLIT_BF16X3;
}
static std::atomic<int> g_gemm_split{gemm_split_default()};

template <bool A_RC, bool B_RC, int EPI>
int launch(GemmArgs& g, hipStream_t s) {

    g.a_vec = ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (g.lda & 3) == 0) ? 1 : 0;
    g.b_vec = ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0 && (g.ldb & 3) == 0) ? 1 : 0;
    const bool narrow = g.N <= 32 && (EPI == EPI_BIAS_ACT || EPI == EPI_MASK || EPI == EPI_ATOMIC || EPI == EPI_HEAD);
    const int bn = narrow ? 32 : BN;
    const int bm = BM;
    const int tiles_n = (g.N + bn - 1) / bn;
    const int64_t tiles_m = (g.M + bm - 1) / bm;
    if (tiles_m * tiles_n > 0x7fffffff) return DR_EINVAL;
    dim3 grid((unsigned)(tiles_m * tiles_n), EPI == EPI_ATOMIC ? g.split : 1);
    if (narrow) {
        if constexpr (EPI == EPI_BIAS_ACT || EPI == EPI_MASK || EPI == EPI_ATOMIC || EPI == EPI_HEAD)
            hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, true>), grid, dim3(256), 0, s, g);
    } else {
        if constexpr (EPI == EPI_HEAD) return DR_ESHAPE;
        else if constexpr (EPI == EPI_FILTER || EPI == EPI_LSE || EPI == EPI_SMGRAD) {
            // short reductions with heavy epilogues (the two-tower rows: K = 128 = 4 k-tiles per output tile): the per-tile
            // prologue / epilogue weigh more than the steady-state loop, so a fourth resident block per CU beats the pinned
            // prefetch (measured: in-batch softmax forward 0.343 -> 0.320 ms, top-K scan 26.65 -> 25.65 ms; plain scores: no)
            // the top-K scan follows the GEMM mode (bf16x3 products: 25.7 -> 23.3 ms at 8192 x 1 M x 128) together with
            // dr_scores_nt, which scores its first chunk: equal candidates must tie bit-exactly across the two kernels.
            // The in-batch softmax pair (LSE forward / gradient) stays on the fp32 MFMA (bf16x3 measured 3 % slower there).
            if (EPI == EPI_FILTER && g_gemm_mode.load(std::memory_order_relaxed) == DR_GEMM_BF16X3) {
                if constexpr (EPI == EPI_FILTER)
                    hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, false, false, true>), grid, dim3(256), 0, s, g);
            } else if (g.R <= 256)
                hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, false, true>), grid, dim3(256), 0, s, g);
            else
                hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, false>), grid, dim3(256), 0, s, g);
        } else if constexpr (EPI == EPI_BIAS_ACT || EPI == EPI_CROSS || EPI == EPI_MASK || EPI == EPI_FMGRAD || EPI == EPI_ATOMIC) {
            // the tower / cross-layer GEMMs: fp32 products on the bf16 matrix pipe unless the caller asked for the native one
            if (g_gemm_mode.load(std::memory_order_relaxed) == DR_GEMM_BF16X3)
                hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, false, false, true>), grid, dim3(256), 0, s, g);
            else
                hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, false>), grid, dim3(256), 0, s, g);
        } else hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_RC, B_RC, EPI, false>), grid, dim3(256), 0, s, g);
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}

__global__ __launch_bounds__(256) void cross_combine_fwd_kernel(const float* __restrict__ x0,
                                                                const float* __restrict__ x,
                                                                float* __restrict__ prod, const float* __restrict__ b,
                                                                int64_t M, int32_t Dm, int64_t ld, float diag,
                                                                float* __restrict__ out) {
    const int64_t n = M * Dm;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / Dm;
        const int c = (int)(i - r * Dm);
        const int64_t o = r * ld + c;
        const float xv = x[o];
        const float p = prod[o] + (b != nullptr ? b[c] : 0.f) + diag * xv;
        prod[o] = p;
        out[o] = x0[o] * p + xv;
    }
}

__global__ __launch_bounds__(256) void cross_combine_bwd_kernel(const float* __restrict__ x0,
                                                                const float* __restrict__ prod,
                                                                const float* __restrict__ d_out, int64_t M, int32_t Dm,
                                                                int64_t ld, float diag, float* __restrict__ d_prod,
                                                                float* __restrict__ d_x0, float* __restrict__ d_x,
                                                                uint32_t* __restrict__ dp_amax) {
    const int64_t n = M * Dm;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float mx = 0.f;                                           // (dp_amax != NULL: the record of d_prod for the f16x2 GEMMs that read it)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / Dm;
        const int c = (int)(i - r * Dm);
        const int64_t o = r * ld + c;
        const float go = d_out[o];
        const float dp = go * x0[o];
        d_prod[o] = dp;
        mx = fmaxf(mx, fabsf(dp));
        if (d_x0 != nullptr) d_x0[o] += go * prod[o];
        if (d_x != nullptr) d_x[o] += go + diag * dp;
    }
    if (dp_amax != nullptr) {
        uint32_t m = __float_as_uint(mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(dp_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dp_amax, m);
    }
}

// dst[k][n] += scale * sum_s partial[s][k][n]   (deterministic split-K combine with the SGD step fused)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int32_t split, int64_t K,
                                                            int32_t N, float scale, float* __restrict__ dst,
                                                            int64_t ld) {
    const int64_t total = K * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float acc = 0.f;
        for (int s = 0; s < split; ++s) acc += partial[(int64_t)s * total + i];
        const int64_t k = i / N;
        const int n = (int)(i - k * N);
        dst[k * ld + n] = fmaf(scale, acc, dst[k * ld + n]);
    }
}

// ---- skinny shapes (min(K, N) < 4: the Dense(1) heads) : streaming kernels, no MFMA -------------------------------
// y[m][n] = act(sum_k x[m][k] W[k][n] + b[n]) ; one lane group of 16 per row, k strided over the lanes
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ b, int64_t M, int32_t K, int32_t N,
                                                         int32_t act, float* __restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 15;
    const int64_t groups = (int64_t)gridDim.x * (blockDim.x >> 4);
    for (int64_t m = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); m < M; m += groups) {
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int k = lane; k < K; k += 16) acc = fmaf(x[m * ldx + k], W[(int64_t)k * ldw + n], acc);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
            if (lane == 0) {
                float v = acc + (b != nullptr ? b[n] : 0.f);
                if (act == 1) v = fmaxf(v, 0.f);
                y[m * ldy + n] = v;
            }
        }
    }
}
// dx[m][k] = (sum_n dy[m][n] W[k][n]) * (relu_src[m][k] > 0) (+ dx)
__global__ __launch_bounds__(256) void skinny_dx_kernel(const float* __restrict__ dy, int64_t lddy,
                                                        const float* __restrict__ W, int64_t ldw, int64_t M, int32_t K,
                                                        int32_t N, const float* __restrict__ rs, int64_t ldrs,
                                                        int32_t accumulate, float* __restrict__ dx, int64_t lddx) {
    const int64_t total = M * K;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / K;
        const int k = (int)(i - m * K);
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(dy[m * lddy + n], W[(int64_t)k * ldw + n], acc);
        if (rs != nullptr && !(rs[m * ldrs + k] > 0.f)) acc = 0.f;
        if (accumulate) acc += dx[m * lddx + k];
        dx[m * lddx + k] = acc;
    }
}
// dst[k][n] += scale * sum_m x[m][k] dy[m][n] ; dstb[n] += scale * sum_m dy[m][n].  A block owns a slab of rows and
// all (k, n) pairs (strided over its threads), accumulates in registers, then one atomic per (k, n) per block.
// (128 rows per block: 512 blocks at M = 65536.  With 1024 the Dense(1) weight gradient of the DCN tower ran on 64 of 256 CUs,
// every thread a 1024-long dependent chain: 258 us for a 67 MB read)
constexpr int SK_ROWS = 128;
__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ dy, int64_t lddy, int64_t M, int32_t K,
                                                        int32_t N, float scale, float* __restrict__ dst, int64_t ldw,
                                                        float* __restrict__ dstb) {
    __shared__ float sm[256];
    const int64_t m0 = (int64_t)blockIdx.x * SK_ROWS;
    const int64_t m1 = m0 + SK_ROWS < M ? m0 + SK_ROWS : M;
    const int KN = K * N;
    const int lanes_e = KN < 256 ? KN : 256;     // threads along the (k, n) pairs; the rest split the rows
    const int R = 256 / lanes_e;
    const int e0 = threadIdx.x % lanes_e, rl = threadIdx.x / lanes_e;
    // the same-address atomics of different blocks serialise (~88 per us): combine the row-lanes in LDS first,
    // then ONE atomic per (k, n) per block
    for (int eb = 0; eb < KN; eb += lanes_e) {
        const int e = eb + e0;
        float acc = 0.f;
        if (e < KN && rl < R) {
            const int k = e / N, n = e - k * N;
            for (int64_t m = m0 + rl; m < m1; m += R) acc = fmaf(x[m * ldx + k], dy[m * lddy + n], acc);
        }
        sm[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0 && e < KN) {
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += sm[e0 + r * lanes_e];
            const int k = e / N, n = e - k * N;
            unsafeAtomicAdd(dst + (int64_t)k * ldw + n, scale * t);
        }
        __syncthreads();
    }
    if (dstb != nullptr) {
        const int ln = N < 256 ? N : 256;
        const int Rb = 256 / ln;
        const int n0 = threadIdx.x % ln, rb = threadIdx.x / ln;
        for (int nb = 0; nb < N; nb += ln) {
            const int n = nb + n0;
            float acc = 0.f;
            if (n < N && rb < Rb)
                for (int64_t m = m0 + rb; m < m1; m += Rb) acc += dy[m * lddy + n];
            sm[threadIdx.x] = acc;
            __syncthreads();
            if (rb == 0 && n < N) {
                float t = 0.f;
                for (int r = 0; r < Rb; ++r) t += sm[n0 + r * ln];
                unsafeAtomicAdd(dstb + n, scale * t);
            }
            __syncthreads();
        }
    }
}

bool bad_ld(int64_t ld, int64_t min) { return ld < min; }
bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) != 0; }

}  // namespace

// w2[n] += scale * sum_b partial[b][n] ; b2 += scale * sum_b partial[b][32] ; loss = inv_n * sum_b partial[b][33].
// Fixed summation order.  7 groups of 34 threads stride over the blocks with 8 loads in flight each: a plain
// "acc += partial[b]" loop is a chain of dependent L2 round trips (128 of them cost ~45 us for a 70 KB reduction).
__global__ __launch_bounds__(256) void head_finish_kernel(const float* __restrict__ partial, int32_t nblocks, int32_t N,
                                                          float scale, float inv_n, float* w2, int64_t ldw2,
                                                          float* b2, float* __restrict__ loss_out) {
    constexpr int NG = 7;
    __shared__ float red[NG][HEAD_PART];
    const int grp = threadIdx.x / HEAD_PART, c = threadIdx.x % HEAD_PART;
    if (grp < NG) {
        float acc = 0.f;
        for (int b0 = grp; b0 < nblocks; b0 += NG * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * NG;
                v[u] = partial[(int64_t)(b < nblocks ? b : b0) * HEAD_PART + c];
                if (b >= nblocks) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        red[grp][c] = acc;
    }
    __syncthreads();
    if (threadIdx.x < HEAD_PART) {
        float sacc = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) sacc += red[g2][c];
        if (c < 32) {
            if (c < N && w2 != nullptr && scale != 0.f) w2[(int64_t)c * ldw2] = fmaf(scale, sacc, w2[(int64_t)c * ldw2]);
        } else if (c == 32) {
            if (b2 != nullptr && scale != 0.f) b2[0] = fmaf(scale, sacc, b2[0]);
        } else if (loss_out != nullptr) {
            loss_out[0] = sacc * inv_n;
        }
    }
}

extern "C" int64_t dr_tower_head_workspace_bytes(int64_t M) {
    const int64_t tiles = (M + BM - 1) / BM;
    return (tiles > 0 ? tiles : 1) * HEAD_PART * (int64_t)sizeof(float);
}

static int tower_head_impl(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1,
                           int64_t M, int64_t n_total, int32_t K, int32_t H, int32_t act, const float* w2,
                           int64_t ld_w2, const float* b2, const float* extra_logit, const float* labels,
                           int32_t loss_mode, float scale, float* dst_w2, int64_t ld_dst_w2, float* dst_b2,
                           float* h_out, int64_t ld_h, float* prob, float* d_logit, float* d_h, int64_t ld_dh,
                           float* loss_out, void* workspace, int64_t workspace_bytes, int32_t parts, dr_stream_t stream) {
    if (M <= 0 || K <= 0 || H <= 0) return DR_EINVAL;
    if (H > 32) return DR_ESHAPE;
    if (!x || !W1 || !w2 || !labels || !workspace || loss_mode < 0 || loss_mode > 2) return DR_EINVAL;
    if (ld_x < K || ld_w1 < H || ld_w2 < 1 || (dst_w2 && ld_dst_w2 < 1) || (h_out && ld_h < H) || (d_h && ld_dh < H)) return DR_EINVAL;
    if (workspace_bytes < dr_tower_head_workspace_bytes(M)) return DR_EINVAL;
    GemmArgs g{};
    g.A = x; g.lda = ld_x; g.B = W1; g.ldb = ld_w1; g.M = M; g.N = H; g.R = K;
    g.C = h_out; g.ldc = ld_h;
    g.bias = b1; g.act = act;
    g.head_w = w2; g.ld_head_w = ld_w2; g.head_b = b2; g.head_extra = extra_logit; g.labels = labels;
    g.loss_mode = loss_mode; g.inv_n = 1.f / (float)(n_total > 0 ? n_total : M);
    g.prob = prob; g.d_logit = d_logit; g.d_h = d_h; g.ld_dh = ld_dh;
    g.head_partial = static_cast<float*>(workspace);
    g.split = 1;
    if (parts & 1) {
        int rc = launch<true, false, EPI_HEAD>(g, dr_s(stream));
        if (rc != DR_OK) return rc;
    }
    if (parts & 2) {
        const int nblocks = (int)((M + BM - 1) / BM);
        hipLaunchKernelGGL(head_finish_kernel, dim3(1), dim3(256), 0, dr_s(stream), g.head_partial, nblocks, H, scale, g.inv_n,
                           dst_w2, ld_dst_w2, dst_b2, loss_out);
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_tower_head_fwd_bwd(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1,
                                     int64_t M, int64_t n_total, int32_t K, int32_t H, int32_t act, const float* w2,
                                     int64_t ld_w2, const float* b2, const float* extra_logit, const float* labels,
                                     int32_t loss_mode, float scale, float* dst_w2, int64_t ld_dst_w2, float* dst_b2,
                                     float* h_out, int64_t ld_h, float* prob, float* d_logit, float* d_h, int64_t ld_dh,
                                     float* loss_out, void* workspace, int64_t workspace_bytes, dr_stream_t stream) {
    return tower_head_impl(x, ld_x, W1, ld_w1, b1, M, n_total, K, H, act, w2, ld_w2, b2, extra_logit, labels, loss_mode, scale, dst_w2,
                           ld_dst_w2, dst_b2, h_out, ld_h, prob, d_logit, d_h, ld_dh, loss_out, workspace, workspace_bytes, 3, stream);
}

// The same call in two halves (parts = 1: the GEMM + head kernel -- prob, d_logit, d_h and the per-block partials; parts = 2: the small
// finish kernel that sums the partials into dst_w2 / dst_b2 / loss_out; 3 = both = the call above).  Nothing the rest of the step
// reads comes out of part 2, so a caller may run it on another stream (it must finish before the NEXT call's part 1: w2 / b2 and the
// workspace).  Round 4: the three small reduce kernels of the step off the training stream.
extern "C" int dr_tower_head_fwd_bwd_parts(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1,
                                           int64_t M, int64_t n_total, int32_t K, int32_t H, int32_t act, const float* w2,
                                           int64_t ld_w2, const float* b2, const float* extra_logit, const float* labels,
                                           int32_t loss_mode, float scale, float* dst_w2, int64_t ld_dst_w2, float* dst_b2,
                                           float* h_out, int64_t ld_h, float* prob, float* d_logit, float* d_h, int64_t ld_dh,
                                           float* loss_out, void* workspace, int64_t workspace_bytes, int32_t parts,
                                           dr_stream_t stream) {
    if (parts < 1 || parts > 3) return DR_EINVAL;
    return tower_head_impl(x, ld_x, W1, ld_w1, b1, M, n_total, K, H, act, w2, ld_w2, b2, extra_logit, labels, loss_mode, scale, dst_w2,
                           ld_dst_w2, dst_b2, h_out, ld_h, prob, d_logit, d_h, ld_dh, loss_out, workspace, workspace_bytes, parts, stream);
}

extern "C" int dr_linear_fwd(const float* x, int64_t ld_x, const float* W, int64_t ld_w, const float* b, int64_t M,
                             int32_t K, int32_t N, int32_t act, float* y, int64_t ld_y, dr_stream_t stream) {
    if (M < 0 || K <= 0 || N <= 0 || act < 0 || act > 1) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x || !W || !y || bad_ld(ld_x, K) || bad_ld(ld_w, N) || ld_y < N || misaligned(x) || misaligned(W))
        return DR_EINVAL;
    if (K < 4 || N < 4) {
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(dr_grid_for(M, 16)), dim3(256), 0, dr_s(stream), x, ld_x, W, ld_w, b, M, K,
                           N, act, y, ld_y);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    GemmArgs g{};
    g.A = x; g.lda = ld_x; g.B = W; g.ldb = ld_w; g.M = M; g.N = N; g.R = K; g.C = y; g.ldc = ld_y;
    g.bias = b; g.act = act; g.split = 1;
    return launch<true, false, EPI_BIAS_ACT>(g, dr_s(stream));
}

extern "C" int dr_linear_bwd_dx(const float* dy, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M, int32_t K,
                                int32_t N, const float* relu_src, int64_t ld_relu_src, int32_t accumulate, float* dx,
                                int64_t ld_dx, dr_stream_t stream) {
    if (M < 0 || K <= 0 || N <= 0) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!dy || !W || !dx || bad_ld(ld_dy, N) || bad_ld(ld_w, N) || ld_dx < K || misaligned(dy) || misaligned(W))
        return DR_EINVAL;
    if (relu_src != nullptr && ld_relu_src < K) return DR_EINVAL;
    if (K < 4 || N < 4) {
        hipLaunchKernelGGL(skinny_dx_kernel, dim3(dr_grid_for(M * K, 256)), dim3(256), 0, dr_s(stream), dy, ld_dy, W, ld_w, M,
                           K, N, relu_src, ld_relu_src, accumulate, dx, ld_dx);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    // dx[i=m][j=k] = sum_{r=n} dy[m][n] * W[k][n]  -> A = dy (RC), B(r=n, j=k) = W[k*ld_w + n] (RC)
    GemmArgs g{};
    g.A = dy; g.lda = ld_dy; g.B = W; g.ldb = ld_w; g.M = M; g.N = K; g.R = N; g.C = dx; g.ldc = ld_dx;
    g.e0 = relu_src; g.lde0 = ld_relu_src; g.accumulate = accumulate; g.split = 1;
    return launch<true, true, EPI_MASK>(g, dr_s(stream));
}

// dgrad of the FIRST tower layer with the FM second-order gradient folded in:
//   d_concat[m, j] = (dy @ W^T)[m, j] + d_fm_logit[m] * (sum_x[m, j % D] - concat[m, j])   for j < F*D
// so the embedding backward needs ONE gradient stream and never re-reads concat (HBM-bound there, free here).
extern "C" int dr_linear_bwd_dx_fm(const float* dy, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M, int32_t K,
                                   int32_t N, const float* d_fm_logit, const float* sum_x, const float* concat,
                                   int64_t ld_concat, int32_t D, int32_t FD, float* dx, int64_t ld_dx,
                                   dr_stream_t stream) {
    if (M < 0 || K < 4 || N < 4 || D <= 0 || FD < 0 || FD > K) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!dy || !W || !dx || !d_fm_logit || !sum_x || !concat || bad_ld(ld_dy, N) || bad_ld(ld_w, N) || ld_dx < K ||
        ld_concat < FD || misaligned(dy) || misaligned(W))
        return DR_EINVAL;
    GemmArgs g{};
    g.A = dy; g.lda = ld_dy; g.B = W; g.ldb = ld_w; g.M = M; g.N = K; g.R = N; g.C = dx; g.ldc = ld_dx;
    g.e0 = concat; g.lde0 = ld_concat; g.e1 = sum_x; g.lde1 = D; g.vec = d_fm_logit; g.fm_D = D; g.fm_FD = FD;
    g.split = 1;
    return launch<true, true, EPI_FMGRAD>(g, dr_s(stream));
}

static int dw_split_for(int64_t M, int32_t K, int32_t N, int mode) {
    const int bn = N <= 32 ? 32 : BN;
    const int64_t tiles = ((int64_t)(K + BM - 1) / BM) * ((N + bn - 1) / bn);
    // resident blocks of one wave of the grid: 256 CUs x 3 (native wide tile) or x 2 (bf16x3: 60 KB of LDS per block)
    const int64_t slots = (mode == DR_GEMM_BF16X3 && N > 32) ? 512 : 768;
    int64_t max_split = (M + 8 * BK - 1) / (8 * BK);           // at least 8 k-tiles per block
    if (max_split > 64) max_split = 64;
    if (max_split < 1) max_split = 1;
    // fill whole waves of resident blocks: the smallest split whose last wave is (nearly) as full as the best one
    // (28 tiles: 18 or 27 splits = 98 %; 196 tiles (1677 x 1677 cross wgrad): 5 splits = 96 % instead of 2 = 77 %)
    double best = 0.0;
    for (int64_t sp = 1; sp <= max_split; ++sp) {
        const int64_t blocks = tiles * sp, waves = (blocks + slots - 1) / slots;
        const double eff = (double)blocks / (double)(waves * slots);
        if (eff > best) best = eff;
    }
    for (int64_t sp = 1; sp <= max_split; ++sp) {
        const int64_t blocks = tiles * sp, waves = (blocks + slots - 1) / slots;
        if ((double)blocks / (double)(waves * slots) >= 0.95 * best) return (int)sp;
    }
    return 1;
}
static int dw_split(int64_t M, int32_t K, int32_t N) {
    return dw_split_for(M, K, N, g_gemm_mode.load(std::memory_order_relaxed));
}
// the workspace must be large enough for either mode (the mode may change between the allocation and the call)
static int dw_split_max(int64_t M, int32_t K, int32_t N) {
    const int a = dw_split_for(M, K, N, DR_GEMM_BF16X3), b = dw_split_for(M, K, N, DR_GEMM_NATIVE_F32);
    return a > b ? a : b;
}

extern "C" int64_t dr_linear_bwd_dw_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return (int64_t)dw_split_max(M, K, N) * K * N * (int64_t)sizeof(float);
}

extern "C" int dr_linear_bwd_dw(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, int64_t M, int32_t K,
                                int32_t N, float scale, float* dstW, int64_t ld_w, float* dstb, float* workspace,
                                int64_t workspace_bytes, dr_stream_t stream) {
    if (M < 0 || K <= 0 || N <= 0) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x || !dy || !dstW || bad_ld(ld_x, K) || bad_ld(ld_dy, N) || ld_w < N || misaligned(x) || misaligned(dy))
        return DR_EINVAL;
    if (K < 4 || N < 4 || M < 4) {
        hipLaunchKernelGGL(skinny_dw_kernel, dim3((unsigned)((M + SK_ROWS - 1) / SK_ROWS)), dim3(256), 0, dr_s(stream), x, ld_x,
                           dy, ld_dy, M, K, N, scale, dstW, ld_w, dstb);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    // dW[i=k][j=n] = sum_{r=m} x[m][k] * dy[m][n] -> A(i=k, r=m) = x[m*ld_x + k] (not RC), B = dy (not RC)
    GemmArgs g{};
    g.A = x; g.lda = ld_x; g.B = dy; g.ldb = ld_dy; g.M = K; g.N = N; g.R = M; g.C = dstW; g.ldc = ld_w;
    g.alpha = scale; g.colsum_dst = dstb;
    g.split = dw_split(M, K, N);
    {
        // a grid.y slice covers `per` = roundup(ceil(M / split), BK) reduction rows, so trailing slices can be EMPTY
        // (M = 8192, split = 31: per = 288, slices 29 and 30 start past M).  An empty block would return before storing
        // its partial tile while the reduce still summed that (uninitialised) workspace slice: launch and reduce the
        // effective number of slices only.
        g.per = ((M + g.split - 1) / g.split + BK - 1) / BK * BK;
        g.split = (int32_t)((M + g.per - 1) / g.per);
    }
    const bool use_ws = workspace != nullptr && workspace_bytes >= dr_linear_bwd_dw_workspace_bytes(M, K, N) && g.split > 1;
    g.partial = use_ws ? workspace : nullptr;
    int rc = launch<false, false, EPI_ATOMIC>(g, dr_s(stream));
    if (rc != DR_OK) return rc;
    if (use_ws) {
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(dr_grid_for((int64_t)K * N, 256)), dim3(256), 0, dr_s(stream), workspace,
                           g.split, (int64_t)K, N, scale, dstW, ld_w);
        DR_CHECK_LAUNCH();
    }
    return DR_OK;
}

// y[m][n] += sum_k x[m][k] W[k][n] for SHORT-AND-WIDE problems (few output tiles, long reduction: the two-tower dq = G c with
// G [B, B], c [B, 128] has 64 tiles of 128 x 128 for 256 CUs): the reduction is split over grid.y into a workspace and the
// slices are summed in a fixed order (deterministic), exactly as the weight gradients do.
extern "C" int64_t dr_linear_fwd_splitk_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    if (M <= 0 || K <= 0 || N <= 0 || M > 0x7fffffff) return 0;
    return (int64_t)dw_split_max(K, (int32_t)M, N) * M * N * (int64_t)sizeof(float);
}

extern "C" int dr_linear_fwd_splitk(const float* x, int64_t ld_x, const float* W, int64_t ld_w, int64_t M, int32_t K, int32_t N,
                                    float* y, int64_t ld_y, float* workspace, int64_t workspace_bytes, dr_stream_t stream) {
    if (M < 0 || M > 0x7fffffff || K < 4 || N < 4) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x || !W || !y || !workspace || bad_ld(ld_x, K) || bad_ld(ld_w, N) || ld_y < N || misaligned(x) || misaligned(W))
        return DR_EINVAL;
    if (workspace_bytes < dr_linear_fwd_splitk_workspace_bytes(M, K, N)) return DR_EINVAL;
    // A(i = m, r = k) = x[m ld_x + k] (row-contiguous), B(r = k, j = n) = W[k ld_w + n]
    GemmArgs g{};
    g.A = x; g.lda = ld_x; g.B = W; g.ldb = ld_w; g.M = M; g.N = N; g.R = K; g.C = y; g.ldc = ld_y;
    g.alpha = 1.f;
    g.split = dw_split(K, (int32_t)M, N);
    g.per = ((K + g.split - 1) / g.split + BK - 1) / BK * BK;      // launch and reduce the non-empty slices only (see dr_linear_bwd_dw)
    g.split = (int32_t)((K + g.per - 1) / g.per);
    g.partial = workspace;
    int rc = launch<true, false, EPI_ATOMIC>(g, dr_s(stream));
    if (rc != DR_OK) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(dr_grid_for(M * N, 256)), dim3(256), 0, dr_s(stream), workspace, g.split, M, N,
                       1.f, y, ld_y);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_cross_fwd(const float* x0, const float* x, int64_t ld, const float* W, int64_t ld_w, const float* b,
                            float diag_scale, int64_t M, int32_t Dm, float* out, float* prod_out,
                            dr_stream_t stream) {
    if (M < 0 || Dm <= 0 || diag_scale < 0.f) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x0 || !x || !out || bad_ld(ld, Dm)) return DR_EINVAL;
    if (W == nullptr) {
        if (prod_out == nullptr) return DR_EINVAL;
        hipLaunchKernelGGL(cross_combine_fwd_kernel, dim3(dr_grid_for(M * Dm, 256)), dim3(256), 0, dr_s(stream), x0, x,
                           prod_out, b, M, Dm, ld, diag_scale, out);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    if (bad_ld(ld_w, Dm) || misaligned(x) || misaligned(W)) return DR_EINVAL;
    if (Dm < 4) {   // tiny feature width: streaming product into prod (or out as scratch), then the combine pass
        float* pbuf = prod_out != nullptr ? prod_out : out;
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(dr_grid_for(M, 16)), dim3(256), 0, dr_s(stream), x, ld, W, ld_w,
                           (const float*)nullptr, M, Dm, Dm, 0, pbuf, ld);
        hipLaunchKernelGGL(cross_combine_fwd_kernel, dim3(dr_grid_for(M * Dm, 256)), dim3(256), 0, dr_s(stream), x0, x, pbuf,
                           b, M, Dm, ld, diag_scale, out);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    GemmArgs g{};
    g.A = x; g.lda = ld; g.B = W; g.ldb = ld_w; g.M = M; g.N = Dm; g.R = Dm; g.C = out; g.ldc = ld;
    g.bias = b; g.e0 = x0; g.lde0 = ld; g.e1 = x; g.lde1 = ld; g.aux = prod_out; g.ldaux = ld;
    g.alpha = diag_scale; g.split = 1;
    return launch<true, false, EPI_CROSS>(g, dr_s(stream));
}

extern "C" int dr_cross_combine_bwd(const float* x0, const float* prod, const float* d_out, int64_t M, int32_t Dm,
                                    int64_t ld, float diag_scale, float* d_prod, float* d_x0_accum, float* d_x_accum,
                                    dr_stream_t stream) {
    if (M < 0 || Dm <= 0) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x0 || !prod || !d_out || !d_prod || ld < Dm) return DR_EINVAL;
    hipLaunchKernelGGL(cross_combine_bwd_kernel, dim3(dr_grid_for(M * Dm, 256)), dim3(256), 0, dr_s(stream), x0, prod,
                       d_out, M, Dm, ld, diag_scale, d_prod, d_x0_accum, d_x_accum, static_cast<uint32_t*>(nullptr));
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// ... that also leaves max |d_prod| (float bits) in d_prod_amax[0] (reset first): the amax record of d_prod for the f16x2 GEMMs that
// take it as an operand (dr_h2_linear_nt, dr_h2_wgrad)
extern "C" int dr_cross_combine_bwd_amax(const float* x0, const float* prod, const float* d_out, int64_t M, int32_t Dm,
                                         int64_t ld, float diag_scale, float* d_prod, float* d_x0_accum, float* d_x_accum,
                                         uint32_t* d_prod_amax, dr_stream_t stream) {
    if (M < 0 || Dm <= 0 || !d_prod_amax) return DR_EINVAL;
    if (hipMemsetAsync(d_prod_amax, 0, sizeof(uint32_t), dr_s(stream)) != hipSuccess) return DR_ELAUNCH;
    if (M == 0) return DR_OK;
    if (!x0 || !prod || !d_out || !d_prod || ld < Dm) return DR_EINVAL;
    hipLaunchKernelGGL(cross_combine_bwd_kernel, dim3(dr_grid_for(M * Dm, 256)), dim3(256), 0, dr_s(stream), x0, prod,
                       d_out, M, Dm, ld, diag_scale, d_prod, d_x0_accum, d_x_accum, d_prod_amax);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// ---- K9: in-batch sampled softmax (Retrieval.call, keras/models/retrieval/sbcnm.py:120-151 of the reference) ----------
extern "C" int dr_inbatch_softmax_grad_scores(const float* q, const float* c, int64_t B, int32_t D, const float* cand_prob,
                                              const int64_t* cand_ids, const float* sample_weight, float inv_temperature,
                                              const float* row_lse, float d_loss, float* G, int64_t ld_g, dr_stream_t stream);
__global__ __launch_bounds__(256) void lse_finalize_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                           int32_t nparts, int64_t B, const float* __restrict__ pos,
                                                           const float* __restrict__ w, float* __restrict__ row_lse,
                                                           float* __restrict__ block_sums) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
        float m = -INFINITY;
        for (int p = 0; p < nparts; ++p) m = fmaxf(m, part_m[(int64_t)p * B + i]);
        float l = 0.f;
        for (int p = 0; p < nparts; ++p) l += part_l[(int64_t)p * B + i] * safe_exp(part_m[(int64_t)p * B + i] - m);
        const float lse = m + logf(l);
        row_lse[i] = lse;
        acc += (w != nullptr ? w[i] : 1.f) * (lse - pos[i]);
    }
    acc = dr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_blocks_kernel(const float* __restrict__ block_sums, int n, float* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (double)block_sums[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// f16x2 path of the two score passes (round 5): bf3_gemm.hip's register-split kernel with the LSE / softmax-gradient epilogues; the
// candidates' two fp16 planes and both amax records live behind the partials in the workspace.
int dr_h2_inbatch_lse(const float* q, int64_t ldq, const uint32_t* q_amax, const void* c_planes, int64_t c_ps, int64_t c_ld,
                      const uint32_t* c_amax, int64_t B, int32_t D, const float* cand_prob, const int64_t* cand_ids, float inv_t,
                      float* part_m, float* part_l, float* pos, dr_stream_t stream);
int dr_h2_inbatch_smgrad(const float* q, int64_t ldq, const uint32_t* q_amax, const void* c_planes, int64_t c_ps, int64_t c_ld,
                         const uint32_t* c_amax, int64_t B, int32_t D, const float* cand_prob, const int64_t* cand_ids, float inv_t,
                         const float* row_lse, const float* sample_weight, float d_loss, float* G, int64_t ld_g, dr_stream_t stream);
extern "C" int dr_h2_amax(const float* src, int64_t ld, int64_t R, int32_t C, uint32_t* amax, int32_t reset, dr_stream_t stream);
extern "C" int dr_h2_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride, int64_t ld_planes,
                           int64_t row_offset, int64_t col_offset, int32_t transpose, const uint32_t* amax, dr_stream_t stream);

constexpr int IB_H2_MAX_D = 512;                        // the workspace is sized without knowing D: planes budgeted for D <= 512
static int64_t ib_parts_floats(int64_t B) {
    const int64_t tiles_n = (B + BN - 1) / BN;
    return 2 * tiles_n * 2 * B + 1024;
}
static int64_t ib_h2_offset_bytes(int64_t B) { return (ib_parts_floats(B) * 4 + 255) / 256 * 256; }
static int64_t ib_h2_bytes(int64_t B) { return 256 + 2 * ((B + 31) / 32 * 32) * IB_H2_MAX_D * 2; }

extern "C" int64_t dr_inbatch_softmax_workspace_bytes(int64_t B) {
    return ib_h2_offset_bytes(B) + ib_h2_bytes(B);
}

// records + candidate planes into the workspace; returns false when the f16x2 path does not apply (the caller runs the fp32 kernel)
static bool ib_h2_prepare(const float* q, const float* c, int64_t B, int32_t D, float* workspace, int64_t workspace_bytes, hipStream_t stream,
                          uint32_t** rec, void** planes, int64_t* ps, int64_t* ld, int* rc) {
    *rc = DR_OK;
    if (dr_get_gemm_split() != DR_GEMM_SPLIT_F16X2 || workspace == nullptr || (D % 4) != 0 || D > IB_H2_MAX_D || B < 256) return false;
    if ((reinterpret_cast<uintptr_t>(q) & 15) != 0 || (reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return false;
    if (workspace_bytes < ib_h2_offset_bytes(B) + ib_h2_bytes(B)) return false;
    char* base = reinterpret_cast<char*>(workspace) + ib_h2_offset_bytes(B);
    *rec = reinterpret_cast<uint32_t*>(base);
    *planes = base + 256;
    *ld = ((int64_t)D + 31) / 32 * 32;
    *ps = ((B + 31) / 32 * 32) * *ld;
    if ((D % 32) != 0 && hipMemsetAsync(*planes, 0, (size_t)(2 * *ps * 2), stream) != hipSuccess) { *rc = DR_ELAUNCH; return true; }
    *rc = dr_h2_amax(q, D, B, D, *rec, 1, stream);
    if (*rc == DR_OK) *rc = dr_h2_amax(c, D, B, D, *rec + 1, 1, stream);
    if (*rc == DR_OK) *rc = dr_h2_split(c, D, B, D, *planes, *ps, *ld, 0, 0, 0, *rec + 1, stream);
    return true;
}

extern "C" int dr_inbatch_softmax_fwd(const float* q, const float* c, int64_t B, int32_t D, const float* cand_prob,
                                      const int64_t* cand_ids, const float* sample_weight, float inv_temperature,
                                      float* row_lse, float* pos_score, float* loss_out, float* workspace,
                                      int64_t workspace_bytes, dr_stream_t stream) {
    if (B <= 0 || D < 4 || B > 0x7fffffff) return DR_EINVAL;
    if (!q || !c || !row_lse || !pos_score || !loss_out || !workspace) return DR_EINVAL;
    if (workspace_bytes < dr_inbatch_softmax_workspace_bytes(B)) return DR_EINVAL;
    const int tiles_n = (int)((B + BN - 1) / BN);
    GemmArgs g{};
    g.A = q; g.lda = D; g.B = c; g.ldb = D; g.M = B; g.N = (int32_t)B; g.R = D; g.C = nullptr; g.ldc = 0;
    g.cand_prob = cand_prob; g.cand_ids = cand_ids; g.inv_t = inv_temperature;
    g.part_m = workspace; g.part_l = workspace + (int64_t)2 * tiles_n * B; g.pos = pos_score; g.split = 1;
    int nparts = 2 * tiles_n;
    uint32_t* rec = nullptr;
    void* planes = nullptr;
    int64_t ps = 0, ld = 0;
    int rc = DR_OK;
    if (ib_h2_prepare(q, c, B, D, workspace, workspace_bytes, dr_s(stream), &rec, &planes, &ps, &ld, &rc)) {
        // the scores on the f16x2 register-split kernel (three fp16 products per fp32 product; 256-column tiles: one partial per tile)
        if (rc == DR_OK)
            rc = dr_h2_inbatch_lse(q, D, rec, planes, ps, ld, rec + 1, B, D, cand_prob, cand_ids, inv_temperature, g.part_m, g.part_l,
                                   pos_score, stream);
        nparts = (int)((B + 255) / 256);
    } else {
        rc = launch<true, true, EPI_LSE>(g, dr_s(stream));
    }
    if (rc != DR_OK) return rc;
    float* block_sums = workspace + (int64_t)4 * tiles_n * B;
    const int grid = dr_grid_for(B, 256, 512);
    hipLaunchKernelGGL(lse_finalize_kernel, dim3(grid), dim3(256), 0, dr_s(stream), g.part_m, g.part_l, nparts, B,
                       pos_score, sample_weight, row_lse, block_sums);
    hipLaunchKernelGGL(sum_blocks_kernel, dim3(1), dim3(256), 0, dr_s(stream), block_sums, grid, loss_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// G[i][j] = d_loss * w_i * (softmax_ij - delta_ij) * inv_t  (the gradient of the loss wrt the raw q.c^T scores);
// the caller finishes with two plain GEMMs: dq = G @ c (dr_linear_fwd), dc = G^T @ q (dr_linear_bwd_dw).
// ... with a workspace (dr_inbatch_softmax_workspace_bytes(B); the forward's may be reused, its contents are not needed): lets the
// pass run on the f16x2 register-split kernel, which wants the candidates as fp16 planes.  workspace NULL / too small, or a split /
// shape the f16x2 path does not take: the fp32 kernel, as dr_inbatch_softmax_grad_scores.
extern "C" int dr_inbatch_softmax_grad_scores_ws(const float* q, const float* c, int64_t B, int32_t D, const float* cand_prob,
                                                 const int64_t* cand_ids, const float* sample_weight, float inv_temperature,
                                                 const float* row_lse, float d_loss, float* G, int64_t ld_g, float* workspace,
                                                 int64_t workspace_bytes, dr_stream_t stream) {
    if (B <= 0 || D < 4 || B > 0x7fffffff || ld_g < B) return DR_EINVAL;
    if (!q || !c || !row_lse || !G) return DR_EINVAL;
    uint32_t* rec = nullptr;
    void* planes = nullptr;
    int64_t ps = 0, ld = 0;
    int rc = DR_OK;
    if (ib_h2_prepare(q, c, B, D, workspace, workspace_bytes, dr_s(stream), &rec, &planes, &ps, &ld, &rc)) {
        if (rc != DR_OK) return rc;
        return dr_h2_inbatch_smgrad(q, D, rec, planes, ps, ld, rec + 1, B, D, cand_prob, cand_ids, inv_temperature, row_lse, sample_weight,
                                    d_loss, G, ld_g, stream);
    }
    return dr_inbatch_softmax_grad_scores(q, c, B, D, cand_prob, cand_ids, sample_weight, inv_temperature, row_lse, d_loss, G, ld_g, stream);
}

extern "C" int dr_inbatch_softmax_grad_scores(const float* q, const float* c, int64_t B, int32_t D, const float* cand_prob,
                                              const int64_t* cand_ids, const float* sample_weight, float inv_temperature,
                                              const float* row_lse, float d_loss, float* G, int64_t ld_g,
                                              dr_stream_t stream) {
    if (B <= 0 || D < 4 || B > 0x7fffffff || ld_g < B) return DR_EINVAL;
    if (!q || !c || !row_lse || !G) return DR_EINVAL;
    GemmArgs g{};
    g.A = q; g.lda = D; g.B = c; g.ldb = D; g.M = B; g.N = (int32_t)B; g.R = D; g.C = G; g.ldc = ld_g;
    g.cand_prob = cand_prob; g.cand_ids = cand_ids; g.inv_t = inv_temperature; g.lse = row_lse; g.vec = sample_weight;
    g.alpha = d_loss; g.split = 1;
    return launch<true, true, EPI_SMGRAD>(g, dr_s(stream));
}

// plain scores = a @ b^T for two reduction-contiguous operands (queries x candidates), used by the top-K search
// internal (C++ linkage, used by retrieval.hip): scores = a @ b^T, filtered against tau into per-row candidate lists
int dr_scores_nt_filter(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M, int32_t N, int32_t D,
                        const float* tau, float* cand_s, int32_t* cand_c, int32_t* cand_cnt, int64_t cand_cap,
                        dr_stream_t stream) {
    if (M < 0 || N <= 0 || D < 4 || lda < D || ldb < D || cand_cap <= 0) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!a || !b || !tau || !cand_s || !cand_c || !cand_cnt) return DR_EINVAL;
    GemmArgs g{};
    g.A = a; g.lda = lda; g.B = b; g.ldb = ldb; g.M = M; g.N = N; g.R = D; g.split = 1;
    g.tau = tau; g.cand_s = cand_s; g.cand_c = cand_c; g.cand_cnt = cand_cnt; g.cand_cap = cand_cap;
    return launch<true, true, EPI_FILTER>(g, dr_s(stream));
}

extern "C" int dr_scores_nt(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M, int32_t N, int32_t D,
                            float* out, int64_t ld_out, dr_stream_t stream) {
    if (M < 0 || N <= 0 || D < 4 || ld_out < N || lda < D || ldb < D) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!a || !b || !out) return DR_EINVAL;
    GemmArgs g{};
    g.A = a; g.lda = lda; g.B = b; g.ldb = ldb; g.M = M; g.N = N; g.R = D; g.C = out; g.ldc = ld_out; g.split = 1;
    return launch<true, true, EPI_BIAS_ACT>(g, dr_s(stream));
}


extern "C" int32_t dr_set_gemm_mode(int32_t mode) {
    if (mode != DR_GEMM_BF16X3 && mode != DR_GEMM_NATIVE_F32) return DR_EINVAL;
    return g_gemm_mode.exchange(mode);
}

extern "C" int32_t dr_get_gemm_mode(void) { return g_gemm_mode.load(); }

extern "C" int32_t dr_set_gemm_split(int32_t split) {
    if (split != DR_GEMM_SPLIT_BF16X3 && split != DR_GEMM_SPLIT_F16X2) return DR_EINVAL;
    return g_gemm_split.exchange(split);
}

extern "C" int32_t dr_get_gemm_split(void) { return g_gemm_split.load(); }

extern "C" const char* dr_version(void) { return "deep_recommenders_amd hot path / gfx950 / f32"; }
