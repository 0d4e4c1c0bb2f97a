// fp32 GEMMs of the first tower layer on PRE-SPLIT operands ("planes" form of the bf16x3 product mode, dense.hip).
//
// dense.hip's bf16x3 kernel splits every fp32 operand value into three bf16 terms on its way into LDS: per k-tile 24
// v_cvt_pk + 48 exact subtractions + 24 ds_write_b64 per thread sit between the global loads and the 48 MFMAs, and the
// kernel reaches 0.33-0.38 of the bf16 pipe's fp32-equivalent ceiling (2.5 PFLOP/s / 6).  Here the PRODUCERS of the
// operands write the three planes once (K3 writes the pooled embeddings as planes, the tower-tail backward writes its
// dx as planes, a small kernel splits W after each update), so the GEMM's staging is pure LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no VALU, no ds_write) and its loop is
//     barrier -> issue next k-tile's DMA -> 24 ds_read_b128 (or 48 ds_read_b64_tr_b16) + 48 MFMAs
// with one barrier per k-tile and two LDS stages.  Three planes of x reproduce x exactly ((x2 + x1) + x0 == x), so the
// results are those of the bf16x3 mode: fp32 operands, fp32 accumulation, dropped terms below 2^-24 |ab|.
//
// Replaces, for the first (wide) Dense layer of keras/models/ranking/deepfm.py:30-34 / estimator dnn.py:17-29 of the
// reference and its autodiff:  y = act(x W + b),  dx = dy W^T,  dW = x^T dy,  db = colsum(dy).
//
// Operand format: planes[p][row][col], p = 0..2, bf16, `ld` elements per row (multiple of 8), plane stride `ps`.
//   NT kernel  C[m][n] = sum_k A[m][k] B[n][k]   both operands reduction-contiguous; K % 32 == 0 with ZERO padding in
//              both operands' planes (forward: A = x planes, B = W^T planes; dgrad: A = dy planes, B = W planes)
//   TN kernel  C[f][n] = sum_r X[r][f] Y[r][n]   both operands reduction-major (wgrad: X = x planes, Y = dy planes);
//              rows r >= R must exist up to the next multiple of 32 and be ZERO; split over r, fp32 partials + reduce
// Tile 64*WM x 64*WN x 32, 8 waves (2 per SIMD), each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 x 16 (bf16), 6 MFMAs per
// (A-fragment, B-fragment) pair.  LDS: 2 stages x 72 KB.
//
// LDS images (written by LDS-DMA: lane i of a wave-instruction lands at base + 16 i, so the image is lane-linear and
// any swizzle is applied to the per-lane SOURCE address and, identically, to the fragment read address):
//   NT: per plane [rows][32 k] = 64-byte rows; 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3)  -> the 16 rows
//       of a ds_read_b128 lane group fall on 16 distinct 16-byte slots of the 256-byte bank row (conflict-free)
//   TN: per plane [32 r][cols] = 256/512-byte rows; chunk c of row r sits at c ^ ((r & 3) << 2) -> the 4 rows x 64 bytes a
//       32-lane half of ds_read_b64_tr_b16 touches fall on 4 distinct 64-byte quarters of the bank row
// Cache-policy experiments (round 4): streams that are read / written ONCE marked nontemporal so that they do not wash the weight
// planes (and, in K4, the first-order lines) out of the L2 / Infinity Cache.  0 = default policy; aux 2 = nt.
#ifndef DR_NT_FWD_GATHER
#define DR_NT_FWD_GATHER 1
#endif
#ifndef DR_NT_WGRAD_GATHER
#define DR_NT_WGRAD_GATHER 0
#endif
#ifndef DR_NT_DGRAD_STORE
#define DR_NT_DGRAD_STORE 0
#endif
#include "dr_common.h"
#include "bf3_split.h"
#include "rs_args.h"
#include <cstdlib>

namespace {

using namespace drrs;
using bf3::bf16x4;
using bf3::bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BK = 32;
constexpr int NWAVES = 8;
constexpr int NTHREADS = 64 * NWAVES;


// 16-byte LDS-DMA: lane i's 16 bytes at `src` land at `dst` (wave-uniform) + 16 i.  A plain (non-template) device function:
// hipcc's host pass cannot substitute the builtin inside a kernel template and silently drops the instantiation.
__device__ __forceinline__ void lds_dma16(const void* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
}

// the six products of one (A-fragment, B-fragment) pair, smallest terms first; the four accumulators of a wave are
// interleaved so that back-to-back MFMAs are independent
template <int TM, int TN>
__device__ __forceinline__ void mma6(const bf16x8 (&af)[3][TM], const bf16x8 (&bf)[3][TN], f32x16 (&acc)[TM][TN]) {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[term]][a], bf[PB[term]][b], acc[a][b], 0, 0, 0);
}

// =====================================================================================================================
// NT:  C[m][n] = epilogue( sum_k A[m][k] B[n][k] )
// =====================================================================================================================
struct NtArgs {
    const __bf16* A; int64_t a_ps, a_ld;
    const __bf16* B; int64_t b_ps, b_ld;
    int64_t M; int32_t N; int32_t K;
    float* C; int64_t ldc;
    const float* bias; int32_t act;              // C = act(acc + bias[n])
    const float* mask; int64_t ld_mask;          // optional: C = 0 where mask[m][n] <= 0   (ReLU' of the layer below)
};

// DBG (tools/exp/bf3_ablate.hip only; 0 in the library): 1 = no LDS-DMA after the first k-tile, 2 = no MFMAs,
// 4 = LDS-DMA re-reads the first k-tile (cache hits)
template <int WM, int WN, int DBG = 0>
__global__ __launch_bounds__(NTHREADS, 2) void bf3_gemm_nt_kernel(NtArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;                 // bytes: rows x 64-byte rows
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    constexpr int NKB = STAGE / 1024, KB_A = 3 * A_PLANE / 1024, PER_WAVE = NKB / NWAVES;
    static_assert(WM * WN == NWAVES && NKB % NWAVES == 0, "tile / wave layout");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int nk = g.K / BK;

    // fragment read offsets (bytes within a stage): row = 64 wm + 32 t + l31, logical chunk = 2 ks + hi
    const int sw = (l31 >> 2) & 3;
    const int a_off = (wm * 64 + l31) * 64 + ((hi ^ sw) << 4);
    const int b_off = 3 * A_PLANE + (wn * 64 + l31) * 64 + ((hi ^ sw) << 4);

    const __bf16* src[PER_WAVE];
    int64_t m0 = 0;
    int n0 = 0;
    auto setup = [&](int tile) {
        const int lid = xcd_remap(tile, ntiles);
        m0 = (int64_t)(lid / tiles_n) * BM;
        n0 = (lid % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int j = wave + NWAVES * i;                            // 1-KB piece of the stage (wave-uniform)
            const bool is_a = j < KB_A;
            const int jj = is_a ? j : j - KB_A;
            const int rows16 = is_a ? BM / 16 : BN / 16;
            const int plane = jj / rows16, rb = jj % rows16;
            const int row = rb * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            int64_t grow = (is_a ? m0 : (int64_t)n0) + row;
            const int64_t lim = is_a ? g.M : (int64_t)g.N;
            grow = grow < lim ? grow : lim - 1;                         // rows past the edge only feed unstored outputs
            src[i] = is_a ? g.A + plane * g.a_ps + grow * g.a_ld + c * 8 : g.B + plane * g.b_ps + grow * g.b_ld + c * 8;
        }
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            unsigned char* dst = smem + stage * STAGE + (wave + NWAVES * i) * 1024;
            lds_dma16(src[i], dst);
            if constexpr (!(DBG & 4)) src[i] += BK;
        }
    };

    f32x16 acc[2][2];
    // One k-tile: all 24 fragment reads are issued up front (the first k-step's MFMAs start as soon as its 12 have landed, the
    // second k-step's land under them), and the next k-tile's LDS-DMA pieces are issued one at a time BETWEEN MFMAs of the
    // first k-step, where the ~60-cycle issue cost of a piece sits in the shadow of the matrix pipe instead of in front of it
    // (issued as a block after the barrier, the 9 pieces kept both waves of a SIMD off the pipe for ~1300 cycles per k-tile).
    auto compute = [&](int stage, bool prefetch) {
        // offsets into the __shared__ array itself (not generic pointers: an XOR on a generic pointer makes hipcc emit
        // flat_load, which also drags the LDS-DMA's vmcnt into every fragment wait)
        const int sa = stage * STAGE + a_off, sb = stage * STAGE + b_off;
        bf16x8 af[2][3][2], bf[2][3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[ks][p][t] = *reinterpret_cast<const bf16x8*>(&smem[(sa ^ (ks * 32)) + p * A_PLANE + t * 32 * 64]);
                    bf[ks][p][t] = *reinterpret_cast<const bf16x8*>(&smem[(sb ^ (ks * 32)) + p * B_PLANE + t * 32 * 64]);
                }
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
        unsigned char* const dma_dst = smem + (stage ^ 1) * STAGE + wave * 1024;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                if constexpr (DBG & 2) {
                    acc[0][0][term] += (float)af[ks][PA[term]][0][0] + (float)af[ks][PA[term]][1][0];
                    acc[0][1][term] += (float)bf[ks][PB[term]][0][0] + (float)bf[ks][PB[term]][1][0];
                } else {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][PA[term]][a], bf[ks][PB[term]][b], acc[a][b], 0, 0, 0);
                }
                if constexpr (!(DBG & 1)) {
                    // pieces 0..8 of the next k-tile after MFMA groups 0..8 (of 12): every piece has at least a quarter of a
                    // k-tile of matrix-pipe time to land before the barrier that publishes it
                    const int piece = ks * 6 + term;
                    if (piece < PER_WAVE) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (prefetch) {
                            lds_dma16(src[piece], dma_dst + NWAVES * piece * 1024);
                            if constexpr (!(DBG & 4)) src[piece] += BK;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int buf = 0;
    setup(tile);
    issue(buf);
    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[a][b][k] = 0.f;
        const int64_t tm0 = m0;
        const int tn0 = n0;
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();              // this stage's DMA has landed (vmcnt(0) per wave, then the barrier); all waves are
                                          // done reading the other stage
            compute(buf, kt + 1 < nk);
            buf ^= 1;
        }
        // every wave is past the last barrier, i.e. done with stage `buf` (read in iteration nk - 2): the next tile's first
        // k-tile streams into it under this tile's epilogue
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            setup(next);
            issue(buf);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = tn0 + wn * 64 + ni * 32 + l31;
                const bool cv = col < g.N;
                const int colc = cv ? col : g.N - 1;
                const float bj = g.bias != nullptr ? g.bias[colc] : 0.f;
                const int64_t row_b = tm0 + wm * 64 + mi * 32 + 4 * hi;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int64_t row = row_b + (reg & 3) + 8 * (reg >> 2);
                    if (!cv || row >= g.M) continue;
                    float v = acc[mi][ni][reg] + bj;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    if (g.mask != nullptr && !(g.mask[row * g.ld_mask + col] > 0.f)) v = 0.f;
                    g.C[row * g.ldc + col] = v;
                }
            }
        if (next >= ntiles) break;
        tile = next;
    }
}

// =====================================================================================================================
// NT, three-stage pipeline ("pipe"): the same product on 128 x 128 x 32 tiles with THREE LDS stages, so that the HBM stream
// of the big operand runs one to two k-tiles ahead of the matrix pipe instead of in lock step with it.
//
// What the two-stage kernel above measured (tools/exp/bf3_ablate.hip, M = 65536, K = 1696, N = 256): 369 us, of which the
// pieces -- A stream from HBM 132 us (667 MB at 5.05 TB/s), B tiles from L2 75 us, barrier + fragment reads 106 us, MFMAs
// ~165 us at the clock the chip holds -- run essentially back to back: with two 72 KB stages a k-tile's LDS-DMA is issued one
// k-tile before the barrier that needs it, every wave waits vmcnt(0) there, and the two waves of a SIMD read fragments and
// issue MFMAs in phase with each other.
//
// Here a block's k-tiles form ONE stream of steps g = 0, 1, 2, ... over all its output tiles (stage = g mod 3):
//   H0(g):  fragment reads (g, k-step 1) -> set 1 | 12 MFMAs on set 0, pieces 3..5 of step g+2 between them
//           lgkmcnt(0) (this wave is done with stage g)  ;  vmcnt(6)  (step g+1 has landed; step g+2 may be in flight)
//           s_barrier                                     -> stage of step g+1 published, stage of step g free
//   H1(g):  fragment reads (g+1, k-step 0) -> set 0 | 12 MFMAs on set 1, pieces 0..2 of step g+3 (into step g's stage) between them
// One barrier per k-tile, fragment reads always one phase ahead of their MFMAs, every LDS-DMA piece one and a half to two
// k-tiles ahead of its barrier, the next output tile's first k-tiles in flight under this tile's epilogue.
// The fragment reads and the waits are inline asm: for a ds_read the compiler's wait-count pass conservatively waits for EVERY
// outstanding LDS-DMA (it cannot tell which stage a read touches), which would serialise the pipeline again.
// Ordering rules used (MI355X_MICROARCH.md, LDS-DMA): a staged buffer is read only after the issuing wave's counted vmcnt AND a
// barrier the reader has passed; a stage is re-filled only after a barrier that every wave reaches with its reads of that
// stage retired (the lgkmcnt(0) in front of it).  VMEM operations retire in issue order on gfx9 (one counter for loads and
// stores), so the epilogue's stores only make the counted waits conservative.
// =====================================================================================================================
#define BF3_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// NW = 8 (2 x 4 waves of 64 x 32 outputs) or 16 (4 x 4 waves of 32 x 32, four per SIMD): an LDS-DMA piece costs its wave
// 100 - 350 cycles of issue time (address coalescer queue), during which only OTHER waves of the SIMD can feed the matrix
// pipe -- with two waves per SIMD the pipe idles about half of the time, with four the stalls overlap.
template <int NW, int DBG = 0>
__global__ __launch_bounds__(64 * NW, NW / 4) void bf3_gemm_nt_pipe_kernel(NtArgs g) {
    constexpr int BM = 128, BN = 128, NS = 3;
    constexpr int WN = 4, WM = NW / WN, TM = BM / (32 * WM);            // wave grid, MFMA tiles per wave along m (n: one)
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;                 // bytes: rows x 64-byte rows (32 bf16)
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);                      // 48 KB
    constexpr int KB_A = 3 * A_PLANE / 1024, PW = STAGE / 1024 / NW;    // 24 of the 48 1-KB pieces are A; PW pieces per wave
    constexpr int P1 = (PW + 1) / 2, P0 = PW - P1;                      // issued in H1 (pieces 0..P1-1) / in H0 (the rest)
    static_assert((NW == 8 || NW == 16) && PW * NW * 1024 == STAGE, "wave layout");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int nk = g.K / BK;
    if ((int)blockIdx.x >= ntiles) return;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nk;                                    // steps of this block

    // fragment read addresses (LDS byte address within stage 0), one per k-step: row = 32 TM wm + 32 t + l31 (A) / 32 wn + l31 (B),
    // logical 16-byte chunk 2 ks + hi at physical chunk (2 ks + hi) ^ ((row >> 2) & 3)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int sw = (l31 >> 2) & 3;
    unsigned a_addr[2], b_addr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_addr[ks] = lds0 + (wm * 32 * TM + l31) * 64 + (((2 * ks + hi) ^ sw) << 4);
        b_addr[ks] = lds0 + 3 * A_PLANE + (wn * 32 + l31) * 64 + (((2 * ks + hi) ^ sw) << 4);
    }

    // ---- producer: the block's LDS-DMA stream -----------------------------------------------------------------------
    const __bf16* src[PW];
    int p_tile = blockIdx.x, p_kt = 0, p_stage = 0;                     // next step to issue, its stage
    auto setup_src = [&](int tile) {
        const int lid = xcd_remap(tile, ntiles);
        const int64_t m0 = (int64_t)(lid / tiles_n) * BM;
        const int n0 = (lid % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int j = wave + NW * i;                                // 1-KB piece of the stage (wave-uniform)
            const bool is_a = j < KB_A;
            const int jj = is_a ? j : j - KB_A;
            const int plane = jj >> 3, rb = jj & 7;                     // 8 pieces of 16 rows per plane
            const int row = rb * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            int64_t grow = (is_a ? m0 : (int64_t)n0) + row;
            const int64_t lim = is_a ? g.M : (int64_t)g.N;
            grow = grow < lim ? grow : lim - 1;                         // rows past the edge only feed unstored outputs
            src[i] = is_a ? g.A + plane * g.a_ps + grow * g.a_ld + c * 8 : g.B + plane * g.b_ps + grow * g.b_ld + c * 8;
        }
    };
    auto issue_piece = [&](int i) {
        lds_dma16(src[i], smem + p_stage * STAGE + (wave + NW * i) * 1024);
        if constexpr (!(DBG & 4)) src[i] += BK;
    };
    auto advance = [&]() {                                              // after the last piece of a step
        p_stage = p_stage == NS - 1 ? 0 : p_stage + 1;
        if (++p_kt == nk) {
            p_kt = 0;
            p_tile += gridDim.x;
            if (p_tile < ntiles) setup_src(p_tile);
        }
    };

    bf16x8 fa[2][3][TM], fb[2][3];                                      // [set][plane][t]: set s holds k-step s of a k-tile
    auto read_set = [&](int set, int stage) {
        const unsigned aa = a_addr[set] + stage * STAGE, bb = b_addr[set] + stage * STAGE;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            BF3_DS_READ_B128(fa[set][p][0], aa, p * A_PLANE);
            if constexpr (TM == 2) BF3_DS_READ_B128(fa[set][p][TM - 1], aa, p * A_PLANE + 32 * 64);
            BF3_DS_READ_B128(fb[set][p], bb, p * B_PLANE);
        }
    };
    auto wait_set = [&](int set) {      // the reads into `set` have landed; ties the MFMAs that follow to the wait
        if constexpr (TM == 2)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[set][0][0]), "+v"(fa[set][0][TM - 1]), "+v"(fa[set][1][0]), "+v"(fa[set][1][TM - 1]),
                           "+v"(fa[set][2][0]), "+v"(fa[set][2][TM - 1]), "+v"(fb[set][0]), "+v"(fb[set][1]), "+v"(fb[set][2]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[set][0][0]), "+v"(fa[set][1][0]), "+v"(fa[set][2][0]), "+v"(fb[set][0]), "+v"(fb[set][1]),
                           "+v"(fb[set][2]));
    };
    f32x16 acc[TM];
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
    // 6 TM MFMAs on `set`; the LDS-DMA pieces [i0, i0 + n) of the producer's current step after terms 0, 2 and 4
    auto mma_phase = [&](int set, int i0, int n, bool dma) {
#pragma unroll
        for (int term = 0; term < 6; ++term) {
            if constexpr (!(DBG & 2)) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][PA[term]][t], fb[set][PB[term]], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < TM; ++t) acc[t][term] += (float)fa[set][PA[term]][t][0] + (float)fb[set][PB[term]][0];
            }
            if ((term & 1) == 0 && (term >> 1) < n) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(DBG & 1)) {
                    if (dma) issue_piece(i0 + (term >> 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: steps 0 and 1 entirely, the first P1 pieces of step 2 ---------------------------------------------------
    setup_src(p_tile);
#pragma unroll
    for (int i = 0; i < PW; ++i) issue_piece(i);
    advance();
    if (total > 1) {
#pragma unroll
        for (int i = 0; i < PW; ++i) issue_piece(i);
        advance();
    }
    if (total > 2) {
#pragma unroll
        for (int i = 0; i < P1; ++i) issue_piece(i);
    }
    if (total > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((PW + P1) & 15) | ((((PW + P1) >> 4) & 3) << 14));   // vmcnt(PW + P1)
    else if (total > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | PW);                                             // vmcnt(PW)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                 // vmcnt(0)
    asm volatile("s_barrier" ::: "memory");
    read_set(0, 0);

    int tile = blockIdx.x, kt = 0, stage = 0;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    for (int step = 0; step < total; ++step) {
        // ---- H0 ----
        wait_set(0);
        read_set(1, stage);
        __builtin_amdgcn_sched_barrier(0);
        mma_phase(0, P1, P0, step + 2 < total);                         // the last P0 pieces of step + 2
        if (step + 2 < total) advance();
        wait_set(1);
        if (step + 2 < total) __builtin_amdgcn_s_waitcnt(0x0F70 | PW);  // vmcnt(PW): step + 1 landed, step + 2 may be in flight
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("s_barrier" ::: "memory");
        // ---- H1 ----
        const int nstage = stage == NS - 1 ? 0 : stage + 1;
        if (step + 1 < total) read_set(0, nstage);
        __builtin_amdgcn_sched_barrier(0);
        mma_phase(1, 0, P1, step + 3 < total);                          // the first P1 pieces of step + 3 (into this step's stage)
        stage = nstage;
        if (++kt < nk) continue;
        // ---- epilogue of an output tile: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        kt = 0;
        {
            const int lid = xcd_remap(tile, ntiles);
            const int64_t tm0 = (int64_t)(lid / tiles_n) * BM;
            const int tn0 = (lid % tiles_n) * BN;
            const int col = tn0 + wn * 32 + l31;
            const bool cv = col < g.N;
            const int colc = cv ? col : g.N - 1;
            float bj = g.bias != nullptr ? g.bias[colc] : 0.f;
            // consume the load HERE on every path: left pending into a branch, its register keeps hipcc's wait-count pass
            // inserting a vmcnt(0) at the top of the main loop (which would drain the LDS-DMA pipeline every k-tile)
            asm volatile("" : "+v"(bj));
            const bool relu = g.act == 1;
            // interior tiles (all but the last row / column of tiles): every store unconditional -- a store under a divergent
            // branch makes hipcc wait vmcnt(0) in front of each one
            if (tm0 + BM <= g.M && tn0 + BN <= g.N) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int64_t r0 = tm0 + wm * 32 * TM + mi * 32 + 4 * hi;
                    float* crow = g.C + r0 * g.ldc + col;
                    float mk[16];
                    if (g.mask != nullptr) {
                        const float* mrow = g.mask + r0 * g.ld_mask + col;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) mk[reg] = mrow[(int64_t)((reg & 3) + 8 * (reg >> 2)) * g.ld_mask];
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = acc[mi][reg] + bj;
                        acc[mi][reg] = 0.f;
                        v = relu ? fmaxf(v, 0.f) : v;
                        if (g.mask != nullptr) v = mk[reg] > 0.f ? v : 0.f;
                        crow[(int64_t)((reg & 3) + 8 * (reg >> 2)) * g.ldc] = v;
                    }
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int64_t row_b = tm0 + wm * 32 * TM + mi * 32 + 4 * hi;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t row = row_b + (reg & 3) + 8 * (reg >> 2);
                        float v = acc[mi][reg] + bj;
                        acc[mi][reg] = 0.f;
                        if (!cv || row >= g.M) continue;
                        v = relu ? fmaxf(v, 0.f) : v;
                        if (g.mask != nullptr && !(g.mask[row * g.ld_mask + col] > 0.f)) v = 0.f;
                        g.C[row * g.ldc + col] = v;
                    }
                }
            }
        }
        tile += gridDim.x;
    }
}

// =====================================================================================================================
// NT, "register split" (RS):  C[m][n] = epilogue( sum_k A[m][k] B[n][k] ),  A fp32 row-major (an ACTIVATION: x, dy),
// B pre-split planes (a WEIGHT: W^T for the forward, W for the dgrad).
//
// Why: every LDS-staged variant above lands at 150 - 165 TFLOP/s whatever its schedule, because a CU cannot INGEST operand
// tiles faster than ~14 - 21 bytes per clock through LDS-DMA (tools/exp/bf3_ablate.hip: the DMA stream alone takes 300 us of
// the 340, the MFMAs alone 250) -- and 128 x 128 / 128 x 256 tiles of six-byte elements need 31 / 23 bytes per clock at the
// matrix pipe's rate.  A 256 x 256 tile needs 16, but its two operands do not fit the LDS three stages deep.  So only B (shared
// by the block's 8 waves) goes through the LDS; each wave owns 32 rows x all 256 columns of the tile and loads ITS rows of A
// straight from HBM into registers as fp32 -- 16 contiguous floats per lane and k-tile, lanes l and l + 32 together one
// 128-byte line per row -- and splits them into the three bf16 terms in registers (VALU work in the shadow of the other
// wave's MFMAs; no LDS write, no fragment read for A).  Activations therefore stay fp32 in HBM (4 bytes per element instead
// of 6, nothing for their producers to do); only the weights are kept as planes (dr_bf3_split after each update).
// The k index inside a k-tile is permuted consistently on both operands: lane half `hi` holds k = 16 hi .. 16 hi + 15, k-step s
// uses 16 hi + 8 s .. + 7, i.e. B's 16-byte chunk 2 hi + s.
//
// Per step (k-tile of 32), B ring of three 48 KB stages, all counts per wave:
//   start:     split A(g) in registers (6 fragments), issue the 4 global loads of A(g + 1)
//   q = 0..15  (k-step q >> 3, column tile q & 7): fragment reads for group q + 1, 6 MFMAs on acc[q & 7];
//              LDS-DMA pieces of step g + 2 after q = 1, 3, .., 11 (into the stage step g - 1 used)
//   q = 15:    before reading group 0 of step g + 1: lgkmcnt(0) (done with stage g), vmcnt(10) (step g + 1 landed; A(g + 1) and
//              step g + 2 may be in flight), s_barrier
// =====================================================================================================================

__device__ __forceinline__ void h2_split8(const float4& lo, const float4& hi4, float s, bf16x8& p0, bf16x8& p1) {
    const f32x8 v = f32x8{lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w} * s;
    const f16x8 h = __builtin_convertvector(v, f16x8);
    const f32x8 r = v - __builtin_convertvector(h, f32x8);
    const f16x8 l = __builtin_convertvector(r, f16x8);
    p0 = __builtin_bit_cast(bf16x8, h);
    p1 = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ f32x16 h2_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void rs_split8(const float4& lo, const float4& hi4, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    bf16x4 a0, a1, a2, b0, b1, b2;
    bf3::split4(lo.x, lo.y, lo.z, lo.w, a0, a1, a2);
    bf3::split4(hi4.x, hi4.y, hi4.z, hi4.w, b0, b1, b2);
    p0 = __builtin_shufflevector(a0, b0, 0, 1, 2, 3, 4, 5, 6, 7);
    p1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    p2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
}

// EPI: 0 = bias / ReLU, 1 = + ReLU' mask, 3 = accumulate (C +=), 2 = DCN cross combine, 4 = top-K filter, 5 = scatter into the
// embedding-gradient send layout (one epilogue per
// instantiation: all of them unrolled over the 8 column tiles in one kernel cost 50 spilled registers)
// MS: 32-row sets per wave.  1 = 8 waves x 32 rows (two waves per SIMD, 128 accumulator registers each); 2 = 4 waves x 64 rows (ONE
// wave per SIMD with all 512 registers, 256 of them accumulators): every weight fragment read from the LDS feeds 12 MFMAs instead
// of 6 and the ring's DMA pieces per MFMA stay the same, at the price of a single in-order instruction stream per SIMD.
// CS: column split of the tile between waves.  CS = 2 (with MS = 2): 8 waves again, wave w owns rows 64 (w & 3) .. + 63 and column tiles
// 4 (w >> 2) .. + 3 -- the same 128 accumulator registers as the default and two waves per SIMD, but half the fragment reads per
// MFMA; the price is that each activation row is loaded and split by two waves (w and w + 4: one SIMD, one L1).
// H2: the f16x2 operand mode (see h2_split8 above): B holds TWO fp16 planes (32 KB stages, 4 pieces per wave and k-tile), three MFMAs
// per fragment pair.
// AR (round 5; H2 only): RESIDENT activation operand for short reductions, K == 32 AR (AR = 4: the top-K scan's K = 128).  A block
// keeps ONE row tile for all of its output tiles (block b: row tile b % tiles_m, column tiles b / tiles_m + j * (grid / tiles_m); the
// launcher makes grid a multiple of tiles_m), loads and splits its 32 K floats per lane ONCE into 16 AR registers and the step
// loop carries no activation loads, no split and no wait for them: only the weight ring, the fragment reads and the MFMAs.  The
// products and their order are those of the streaming kernel -- bit-identical results (the scan's dense first chunk, which runs
// the streaming kernel, must tie exactly with the filtered chunks).
template <int EPI, int DBG = 0, int MS = 1, int CS = 1, int H2 = 0, int AR = 0>
__global__ __launch_bounds__(512 * CS / MS, MS == CS ? 2 : 1) void bf3_gemm_rs_kernel(RsArgs g) {
    constexpr int NW = 8 * CS / MS, RGW = NW / CS, BM = 32 * MS * RGW, BN = 256, NT = BN / 32, NTW = NT / CS, NS = 3;
    static_assert(CS == 1 || (CS == 2 && MS == 2), "column split only with 64-row waves");
    static_assert(!H2 || CS == 1, "f16x2: no column split");
    static_assert(AR == 0 || (H2 && MS == 1 && CS == 1 && DBG == 0), "resident activations: f16x2, 8 waves x 32 rows");
    constexpr int NPL = H2 ? 2 : 3;                                     // operand planes
    constexpr int B_PLANE = BN * 64;                                    // bytes: 256 rows x 64-byte rows (32 bf16 / fp16)
    constexpr int STAGE = NPL * B_PLANE;                                // 48 KB (32 KB)
    constexpr int PW = STAGE / 1024 / NW;                               // 6 MS (4) LDS-DMA pieces per wave and k-tile
    static_assert(PW * NW == 16 * NPL && (MS == 1 || MS == 2), "piece schedule below assumes 16 NPL / NW pieces per wave and k-tile");
    constexpr int KT_UNROLL = AR ? AR : 1;                              // (AR: the k-tile loop is unrolled, see there)
    constexpr int VM_STEP = (AR ? 0 : 4 * MS) + PW;                     // VMEM operations of one step (A loads + pieces)
    constexpr int VM_WAIT_STEP = 0x0F70 | (VM_STEP & 15) | ((VM_STEP >> 4) << 14);   // s_waitcnt vmcnt(VM_STEP) (6-bit field, split)
    // EPI 5 in the f16x2 mode stages each wave's 32 x 32 accumulator block through the LDS (row-major, pitch 36 floats) behind the ring
    // (+ the wave's 32 x 4 destinations and 32 d_fm_logit values of the tile: registers are what this epilogue is short of)
    constexpr int PK_PITCH = 36, PK_STG = 32 * PK_PITCH * 4, PK_WAVE = PK_STG + 32 * 4 * 4 + 32 * 4;    // 4608 + 512 + 128 bytes per wave
    constexpr int PK_BYTES = (EPI == 5 && H2) ? NW * PK_WAVE : 0;       // 41 KB: 96 + 41 of the 160 KB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE + PK_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // STG (DBG bit 16): STAGGERED wave groups.  Waves w and w + 4 share a SIMD; with one barrier per k-tile they run in phase, so the
    // non-MFMA part of a step (waiting for and splitting the activation registers, issuing the next loads: ~1.5 us of a 5.1 us
    // step) is paid by both at the same time with the matrix pipe idle.  Here waves 4..7 lag waves 0..3 by 6 of the 16 MFMA groups
    // (one extra barrier before their first step, one extra for waves 0..3 after their last; a second barrier per step at group
    // 6): while one group splits, the other multiplies.  The three-stage ring still suffices: stage s % 3 is refilled for step
    // s + 3 only after the barrier at which the lagging group ends step s, which is the leading group's mid barrier of step s + 1
    // -- so every wave issues its pieces in groups 6..11, behind its own mid barrier -- and a wave has its pieces of step t + 1
    // landed before its mid barrier of step t (vmcnt(4): only this step's 4 activation loads are younger), which is the barrier
    // in front of the leading group's first read of that stage.  Bit-identical results; no gain measured (see rs_launch).
    constexpr bool STG = (DBG & 16) != 0 && MS == 1;
    const int rgi = CS == 1 ? wave : (wave & (RGW - 1));               // row group / column half of this wave
    const int chi = CS == 1 ? 0 : (wave / RGW);
    const int grp = STG ? (wave >> 2) : 0;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int nk = (g.K + BK - 1) / BK;
    const bool ktail = (g.K % BK) != 0;
    // the block's tiles: `tile` = blockIdx.x + j gridDim.x, j < my_tiles; lid_of(tile) = its logical id (row tile * tiles_n + column tile)
    const int ar_npg = AR ? (int)gridDim.x / tiles_m : 1;               // AR: column tiles are dealt round-robin to the grid / tiles_m blocks of a row tile
    const int ar_n0 = AR ? (int)blockIdx.x / tiles_m : 0;
    const int my_tiles = AR ? (ar_n0 < tiles_n ? (tiles_n - ar_n0 + ar_npg - 1) / ar_npg : 0)
                            : ((int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0);
    if (my_tiles <= 0) return;
    const int tile_end = (int)blockIdx.x + my_tiles * (int)gridDim.x;
    auto lid_of = [&](int tile) {
        if constexpr (AR != 0) return ((int)blockIdx.x % tiles_m) * tiles_n + ar_n0 + ((tile - (int)blockIdx.x) / (int)gridDim.x) * ar_npg;
        else return xcd_remap(tile, ntiles);
    };
    const int total = my_tiles * nk;                                    // steps of this block

    // B fragment read addresses (stage 0, plane 0, column tile 0), one per k-step: row = 32 nt + l31, chunk 2 hi + s swizzled
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int sw = (l31 >> 2) & 3;
    unsigned b_addr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) b_addr[s] = lds0 + l31 * 64 + (((2 * hi + s) ^ sw) << 4) + chi * (NTW * 2048);

    // ---- producer state: the block's stream of steps (tile, k-tile), shared by the B pieces and the A loads -----------------
    // B: piece j = wave + 8 i of a stage: plane j / 16, rows 16 (j % 16) .. + 15, lane -> (row, 16-byte chunk).
    // Issued as MUBUF `buffer_load_dwordx4 ... offen lds` (raw buffer builtin), not as global_load_lds: hipcc's wait-count pass
    // treats the FLAT-encoded form as "may touch LDS through flat" and from then on waits vmcnt(0) for every VMEM result,
    // the buffer form is counted exactly (a register load followed by 6 pieces gets vmcnt(6)).
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(g.B), 0, (int)min((int64_t)0x7fffffff, NPL * g.b_ps * 2), 0x00020000);
    float h2_sa = 1.f, h2_out = 1.f;                                    // H2: A's scale, 1 / (s_a s_b)
    float cmax = 0.f;                                                   // H2: largest |value| this lane stored into C
    if constexpr (H2) {
        float ia, sb, ib;
        h2_scale_of(g.a_amax[0], h2_sa, ia);
        h2_scale_of(g.b_amax[0], sb, ib);
        h2_out = ia * ib;
        h2_mode_on();
    }
    int bvoff[12];                                                      // per-lane byte offset of piece i < PW at k = 0 (arrays the
                                                                        // lambdas capture have FIXED sizes: with a template-dependent size
                                                                        // hipcc silently drops the kernel's host stub)
    int pb_tile = blockIdx.x, pb_kt = 0, pb_stage = 0, pb_koff = 0;     // pb_koff: byte offset along k of the next real step
    auto setup_b = [&](int tile) {
        const int n0 = (lid_of(tile) % tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int j = wave + NW * i;
            const int plane = j >> 4, rb = j & 15;
            const int row = rb * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            int64_t grow = (int64_t)n0 + row;
            grow = grow < g.N ? grow : g.N - 1;                         // columns past the edge only feed unstored outputs
            bvoff[i] = (int)((plane * g.b_ps + grow * g.b_ld + c * 8) * 2);
        }
    };
    // EVERY step issues exactly 4 A loads and 6 pieces, also the last ones of the block (whose successors do not exist): with a
    // fixed number of VMEM operations per step the counted waits are constants, and hipcc's own wait for the A registers comes
    // out as vmcnt(6).  A step that does not exist re-fetches the last k-tile into a stage nobody will read again.
    auto issue_b = [&](int i) {
        unsigned char* dst = smem + pb_stage * STAGE + (wave + NW * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)dst, 16, bvoff[i], (DBG & 4) ? 0 : pb_koff, 0, 0);
    };
    auto advance_b = [&]() {                                            // after the 6 pieces of a REAL step
        pb_stage = pb_stage == NS - 1 ? 0 : pb_stage + 1;
        pb_koff += BK * 2;
        if (++pb_kt == nk) {
            pb_kt = 0;
            pb_tile += gridDim.x;
            if (pb_tile < tile_end) {
                setup_b(pb_tile);
                pb_koff = 0;
            } else {
                pb_koff -= BK * 2;                                      // end of the stream: dummies re-fetch the last k-tile
            }
        }
    };
    // A: this lane's 16 floats of a k-tile
    const float* asrc[2];
    int pa_tile = blockIdx.x, pa_kt = 0;
    auto setup_a = [&](int tile) {
        const int64_t m0 = (int64_t)(lid_of(tile) / tiles_n) * BM;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            int64_t row = m0 + rgi * (32 * MS) + 32 * ms + l31;
            row = row < g.M ? row : g.M - 1;                            // rows past the edge only feed unstored outputs
            asrc[ms] = g.A + row * g.lda + 16 * hi;
        }
    };
    float4 an[2][4];                                                    // A of the NEXT step, in flight
    const int kv4 = (g.K + 3) / 4 * 4;                                  // rows are readable up to here (lda % 4 == 0, lda >= K)
    int an_k0 = 0;                                                      // k index of an[0].x
    auto mask_a = [&]() {
        if (!ktail) return;                                             // kernel-uniform: K % 32 == 0 has nothing to zero
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = an_k0 + 4 * q;
                an[ms][q].x = k < g.K ? an[ms][q].x : 0.f;
                an[ms][q].y = k + 1 < g.K ? an[ms][q].y : 0.f;
                an[ms][q].z = k + 2 < g.K ? an[ms][q].z : 0.f;
                an[ms][q].w = k + 3 < g.K ? an[ms][q].w : 0.f;
            }
    };
    auto load_a = [&](bool real) {
        // Every load unconditional (a load under a divergent branch makes hipcc's wait-count pass fall back to vmcnt(0), which
        // would also drain the LDS-DMA pieces in flight): in the last k-tile of a row a 16-byte load that would start past the
        // row's readable end is pulled back inside it, and everything at k >= K is zeroed in registers -- B's planes are zero
        // there, but 0 * NaN is not.
        // The zeroing happens when the values are CONSUMED (mask_a, at the start of the step that uses them): touching them
        // here would wait for the loads on the spot.
        an_k0 = pa_kt * BK + 16 * hi;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int over = max(0, an_k0 + 4 * q - (kv4 - 4));
                an[ms][q] = *reinterpret_cast<const float4*>(asrc[ms] + 4 * q - over);
            }
        if (!real) return;                                              // (uniform; no VMEM below)
        if constexpr (!(DBG & 8)) {
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) asrc[ms] += BK;
        }
        if (++pa_kt == nk) {
            pa_kt = 0;
            pa_tile += gridDim.x;
            if (pa_tile < tile_end) setup_a(pa_tile);
            else if constexpr (!(DBG & 8)) {                            // end of the stream: the dummy loads re-read k-tile 0 of the
#pragma unroll                                                          // last rows (pa_kt == 0, so `over` keeps them inside the row)
                for (int ms = 0; ms < MS; ++ms) asrc[ms] -= (int64_t)nk * BK;
            }
        }
    };

    bf16x8 fa[MS][2][3];                                                // [row set][k-step][plane] of the CURRENT step
    bf16x8 fb[4][3];                                                    // [buffer][plane]: group q uses buffer q & 3; the
                                                                        // reads run TWO groups (12 MFMAs) ahead of their use
                                                                        // (CS == 2: buffers 0 / 1, ONE group = 12 MFMAs ahead)
    auto read_b = [&](int buf, int stage, int q) {                      // group q = (k-step q / NTW, column tile q % NTW)
        if constexpr (DBG & 32) return;                                 // ablation: no fragment reads (stale registers)
        const unsigned bb = b_addr[q / NTW] + stage * STAGE;
        const int nt = q % NTW;
        // immediates must be literal: dispatch on the column tile
#define RS_READ3(NTI)                                                        \
        BF3_DS_READ_B128(fb[buf][0], bb, 0 * B_PLANE + NTI * 2048);          \
        BF3_DS_READ_B128(fb[buf][1], bb, 1 * B_PLANE + NTI * 2048);          \
        if constexpr (!(DBG & 256) && !H2) BF3_DS_READ_B128(fb[buf][2], bb, 2 * B_PLANE + NTI * 2048);
        switch (nt) {
            case 0: RS_READ3(0) break; case 1: RS_READ3(1) break; case 2: RS_READ3(2) break; case 3: RS_READ3(3) break;
            case 4: RS_READ3(4) break; case 5: RS_READ3(5) break; case 6: RS_READ3(6) break; default: RS_READ3(7) break;
        }
#undef RS_READ3
    };
    auto wait_b = [&](int buf, bool all) {   // group's fragments landed; `all`: every LDS read of this wave retired
        if constexpr (DBG & 32) return;
        if constexpr (H2) {
            if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]));
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]));   // the next group's 2 may fly
            return;
        }
        if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]));
        else if constexpr (DBG & 256) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]));
        else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]));   // the next group's 3 may fly
    };
    f32x16 acc[MS][NTW];
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};

    // ---- prologue: B steps 0 and 1, A step 0 ---------------------------------------------------------------------------------
    setup_b(pb_tile);
    setup_a(pa_tile);
    bf16x8 fa_res[AR ? AR : 1][2][2];                                   // AR: [k-tile][k-step][plane] of the block's row tile
    if constexpr (AR != 0) {                                            // ... loaded and split here, before any weight piece is in flight
#pragma unroll
        for (int t = 0; t < (AR ? AR : 1); ++t) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(asrc[0] + t * BK + 4 * q);
            h2_split8(v[0], v[1], h2_sa, fa_res[t][0][0], fa_res[t][0][1]);
            h2_split8(v[2], v[3], h2_sa, fa_res[t][1][0], fa_res[t][1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) issue_b(i);
    advance_b();
    if constexpr (AR == 0) load_a(true);                                // BEFORE step 1's pieces: 6 VMEM operations follow the A
#pragma unroll                                                          // loads on every path into the loop, as inside it
    for (int i = 0; i < PW; ++i) issue_b(i);
    if (total > 1) advance_b();
    __builtin_amdgcn_s_waitcnt(VM_WAIT_STEP);                           // vmcnt(4 MS + PW): step 0's pieces landed
    asm volatile("s_barrier" ::: "memory");
    if (STG && grp) asm volatile("s_barrier" ::: "memory");             // the lagging group starts at the leading group's first mid barrier
    read_b(0, 0, 0);
    if constexpr (CS == 1) read_b(1, 0, 1);

    int tile = blockIdx.x, kt = 0, stage = 0;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[ms][t][k] = 0.f;
    int step = 0;
    for (; tile < tile_end; tile += gridDim.x) {
    // (AR: the k-tiles unrolled, so that each step names its resident registers directly -- fa is an alias then, not a copy)
#pragma unroll KT_UNROLL
    for (kt = 0; kt < (AR ? AR : nk); ++kt, ++step) {
        // ---- step start: A of this step -> bf16 terms (the compiler waits for the loads here), next step's A into flight ----
        if constexpr (AR != 0) {                                        // resident: this k-tile's terms out of the block's registers
#define RS_AR_PICK(KT)                                                                                     \
            if constexpr (KT < (AR ? AR : 1)) {                                                            \
                if (kt == KT) {                                                                            \
                    fa[0][0][0] = fa_res[KT][0][0]; fa[0][0][1] = fa_res[KT][0][1];                        \
                    fa[0][1][0] = fa_res[KT][1][0]; fa[0][1][1] = fa_res[KT][1][1];                        \
                }                                                                                          \
            }
            RS_AR_PICK(0) RS_AR_PICK(1) RS_AR_PICK(2) RS_AR_PICK(3)
#undef RS_AR_PICK
        } else {
        mask_a();
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            if constexpr (DBG & 64) {                                   // ablation: no split (the loaded bits as operands)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fa[ms][s2][pl] = __builtin_bit_cast(bf16x8, an[ms][2 * s2 + (pl & 1)]);
            } else if constexpr (H2 != 0) {
                h2_split8(an[ms][0], an[ms][1], h2_sa, fa[ms][0][0], fa[ms][0][1]);
                h2_split8(an[ms][2], an[ms][3], h2_sa, fa[ms][1][0], fa[ms][1][1]);
            } else {
                rs_split8(an[ms][0], an[ms][1], fa[ms][0][0], fa[ms][0][1], fa[ms][0][2]);
                rs_split8(an[ms][2], an[ms][3], fa[ms][1][0], fa[ms][1][1], fa[ms][1][2]);
            }
        }
        }   // !AR
        __builtin_amdgcn_sched_barrier(0);
        const bool has_a = step + 1 < total, has_b = step + 2 < total;
        if constexpr (AR == 0) load_a(has_a);
        __builtin_amdgcn_sched_barrier(0);
        const int nstage = stage == NS - 1 ? 0 : stage + 1;
        if constexpr (CS == 2) {
#pragma unroll
            for (int q = 0; q < 2 * NTW; ++q) {
                // this group's three fragments were issued a whole group (12 MFMAs) ago
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[q & 1][0]), "+v"(fb[q & 1][1]), "+v"(fb[q & 1][2]));
                if (q + 1 < 2 * NTW) {
                    read_b((q + 1) & 1, stage, q + 1);
                } else {
                    // this wave is done reading stage `stage`; publish step + 1
                    __builtin_amdgcn_s_waitcnt(VM_WAIT_STEP);
                    asm volatile("s_barrier" ::: "memory");
                    if (has_a) read_b(0, nstage, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        acc[ms][q % NTW] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ms][q / NTW][PA[term]], fb[q & 1][PB[term]],
                                                                                   acc[ms][q % NTW], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < PW) {
                    if constexpr (!(DBG & 1)) issue_b(q);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (STG && q == 6) {
                __builtin_amdgcn_s_waitcnt(0x0F70 | 4);                // vmcnt(4): the pieces of step + 1 (issued a step ago) landed
                asm volatile("s_barrier" ::: "memory");
            }
            if (q < 14) {
                wait_b(q & 3, false);
                read_b((q + 2) & 3, stage, q + 2);
            } else if (q == 14) {
                // groups 14 and 15 are in registers and this wave is done reading stage `stage`; publish step + 1
                wait_b(2, true);
                __builtin_amdgcn_s_waitcnt(VM_WAIT_STEP);              // all but this step's 4 MS A loads + PW pieces
                if constexpr (!(DBG & 512)) asm volatile("s_barrier" ::: "memory");   // (DBG 512, ablation: what the step's rendezvous costs -- racy garbage)
                if (has_a) read_b(0, nstage, 0);
            } else {
                if (has_a) read_b(1, nstage, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (H2 && !(DBG & 2)) {
                // smallest terms first: h_a l_b, l_a h_b, h_a h_b
                constexpr int HA[3] = {0, 1, 0}, HB[3] = {1, 0, 0};
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        acc[ms][q & 7] = h2_mfma(fa[ms][q >> 3][HA[term]], fb[q & 3][HB[term]], acc[ms][q & 7]);
            } else if constexpr (!(DBG & 2)) {
                // (MS == 2: the two row sets' chains interleaved or one after the other -- fenced, or the machine scheduler re-interleaves
                // them -- measure the same)
                // (DBG & 128, ablation only: THREE products per fragment pair -- what a two-plane fp16 split would issue)
#pragma unroll
                for (int term = (DBG & 128) ? 3 : 0; term < 6; ++term)
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        acc[ms][q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ms][q >> 3][PA[term]], fb[q & 3][PB[term]],
                                                                                 acc[ms][q & 7], 0, 0, 0);
            } else {
                acc[0][q & 7][0] += (float)fa[0][q >> 3][0][0] + (float)fb[q & 3][0][0] + (float)fb[q & 3][1][0] + (H2 ? 0.f : (float)fb[q & 3][2][0]);
            }
            __builtin_amdgcn_sched_barrier(0);    // keeps the next group's lgkmcnt wait from being hoisted between these MFMAs
            if (STG ? (q >= 6 && q < 6 + PW) : (MS == 2 ? q < PW : ((q & 1) == 1 && q < 2 * PW))) {
                if constexpr (!(DBG & 1)) issue_b(STG ? q - 6 : (MS == 2 ? q : q >> 1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }   // CS == 1
        if (has_b) advance_b();
        stage = nstage;
    }
        // ---- epilogue of an output tile: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        {
            const int lid = lid_of(tile);
            const int64_t tm0 = (int64_t)(lid / tiles_n) * BM;
            const int tn0 = (lid % tiles_n) * BN;
            const bool relu = g.act == 1;
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) {
            const int64_t r0 = tm0 + rgi * (32 * MS) + 32 * ms + 4 * hi;
            const int cb0 = tn0 + chi * (NTW * 32);                     // first column of this wave's column tiles
            // interior tiles: every load / store of the epilogue unconditional (a memory operation under a divergent branch makes
            // hipcc wait vmcnt(0) in front of each one); edge tiles take the guarded loop
            const bool interior = tm0 + BM <= g.M && tn0 + BN <= g.N;
            constexpr bool cross = EPI == 2;
            if constexpr (EPI == 4) {
                // top-K filter: nothing is stored unless a score beats its row's threshold -- rare once tau has warmed up (expected
                // k / items_seen of a chunk), plentiful in the first filtered chunks (~400 per row right behind the dense chunk).
                // Written in round 2 as "per (register, column tile): ballot, one atomic by the first survivor, wait for its result,
                // store" -- up to 128 dependent atomic round trips per wave and tile, each behind the previous one's stores (the first
                // filtered chunk took 858 us against 316 us in steady state).  Now: (0) the 16 thresholds of the lane's rows loaded
                // together; (1) a counting pass -- the survivors of each row of this half-wave, summed over the 8 column tiles with
                // scalar popcounts; (2) ONE atomic instruction for the tile: lane r < 16 of each half reserves row r's slots;
                // (3) a storing pass that recomputes the masks (compares are cheap, keeping 128 masks is not).  Stores are
                // unconditional inside a wave-uniform branch: lanes without a survivor repeat the first survivor's store (same address,
                // same data) -- a memory operation under a divergent branch would make hipcc drain vmcnt in front of each one.
                // Which slot of its row's list a candidate lands in is arbitrary, as before (the list's order is total: topk_list.h).
                // lane r < 16 of a half stands for row r of that half: its threshold here, its survivor count and its list slots below
                const int64_t lrow = r0 + (l31 & 3) + 8 * ((l31 >> 2) & 3);
                float tlane = g.tau[lrow < g.M ? lrow : g.M - 1];
                // compared with the RAW accumulators (f16x2: the threshold in accumulator units -- a power-of-two factor, exact): the
                // scaled scores are formed only where one is stored
                if constexpr (H2) tlane = tlane / h2_out;
                if (!(lrow < g.M)) tlane = INFINITY;                    // rows past the edge: nothing passes
                int mycnt = 0;                                          // lane r < 16 of a half: survivors of row r of that half
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int c_lo = 0, c_hi = 0;                             // (scalar)
                    const float tq = __shfl(tlane, 32 * hi + reg, 64);
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const unsigned long long b = __ballot(cb0 + nt * 32 + l31 < g.N && acc[ms][nt][reg] > tq);
                        c_lo += __popc((unsigned)b);
                        c_hi += __popc((unsigned)(b >> 32));
                    }
                    if (l31 == reg) mycnt = hi ? c_hi : c_lo;
                    __builtin_amdgcn_sched_barrier(0);                  // one register at a time (hoisted, the 128 scaled scores spill)
                }
                if (__ballot(mycnt > 0) != 0ull) {                      // (wave-uniform)
                    int mybase = 0;
                    if (mycnt > 0) mybase = atomicAdd(g.cand_cnt + lrow, mycnt);     // (only lanes l31 < 16 count)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t row = r0 + (reg & 3) + 8 * (reg >> 2);
                        int run = __shfl(mybase, 32 * hi + reg, 64);    // next free slot of this half's row
                        const float tq = __shfl(tlane, 32 * hi + reg, 64);
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) {
                            const int col = cb0 + nt * 32 + l31;
                            float raw = acc[ms][nt][reg];
                            asm volatile("" : "+v"(raw));               // opaque: else the counting pass's 128 compare results are kept (spilled) for reuse
                            acc[ms][nt][reg] = 0.f;
                            const bool pass = col < g.N && raw > tq;
                            const unsigned long long b = __ballot(pass);
                            if (b == 0ull) continue;                    // (wave-uniform)
                            const float v = H2 ? raw * h2_out : raw;
                            const unsigned half = (unsigned)(b >> (32 * hi));
                            const int pos = run + __popc(half & ((1u << l31) - 1u));
                            run += __popc(half);
                            const bool ok = pass && pos < g.cand_cap;
                            const unsigned long long okb = __ballot(ok);
                            if (okb == 0ull) continue;
                            const int first = __ffsll((long long)okb) - 1;   // (wave-uniform) the survivor the idle lanes repeat
                            int64_t dst = row * g.cand_cap + pos;
                            const int d_lo = __builtin_amdgcn_readlane((int)(dst & 0xffffffffll), first);
                            const int d_hi = __builtin_amdgcn_readlane((int)(dst >> 32), first);
                            const float v1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), first));
                            const int c1 = __builtin_amdgcn_readlane(col, first);
                            if (!ok) dst = ((int64_t)d_hi << 32) | (uint32_t)d_lo;
                            g.cand_s[dst] = ok ? v : v1;
                            g.cand_c[dst] = ok ? col : c1;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) acc[ms][nt][reg] = 0.f;
                }
            } else if constexpr (EPI == 6 || EPI == 7) {
                // in-batch softmax on the scores of this tile (see RsArgs).  Lane r < 16 of a half stands for row r of that half (its
                // candidate id, lse, weight: one load each per tile, handed out by shuffles); a lane's 8 columns keep their
                // correction and id in registers.  Exponentials by v_exp_f32 (__expf: 2 ulp): the arguments carry the products' 1e-6
                // already, and libm's expf was a third of this epilogue (128 per lane and tile).
                constexpr float MIN_FLOAT = -3.4028234663852886e36f;    // np.finfo(np.float32).min / 100 (sbcnm.py:10)
                const bool ids_on = g.sm_cand_ids != nullptr;
                const int64_t lrow = r0 + (l31 & 3) + 8 * ((l31 >> 2) & 3);
                const int64_t lrc = lrow < g.M ? lrow : g.M - 1;
                const int64_t rid_l = ids_on ? g.sm_cand_ids[lrc] : 0;
                float lse_l = 0.f, w_l = 1.f;
                if constexpr (EPI == 7) {
                    lse_l = g.sm_lse[lrc];
                    w_l = g.sm_w != nullptr ? g.sm_w[lrc] : 1.f;
                }
                float colcorr[NTW];
                int64_t colid[NTW];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int col = cb0 + nt * 32 + l31;
                    const int cc = col < g.N ? col : g.N - 1;
                    colcorr[nt] = g.sm_cand_prob != nullptr ? -logf(g.sm_cand_prob[cc]) : 0.f;
                    colid[nt] = ids_on ? g.sm_cand_ids[cc] : 0;
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int64_t row = r0 + (reg & 3) + 8 * (reg >> 2);
                    const bool rv = row < g.M;
                    const int src = 32 * hi + reg;
                    const int rlo = __shfl((int)(rid_l & 0xffffffffll), src, 64), rhi = __shfl((int)(rid_l >> 32), src, 64);
                    const int64_t rid = ((int64_t)rhi << 32) | (uint32_t)rlo;
                    float sv[NTW];
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        float v = fmaf(acc[ms][nt][reg], h2_out, colcorr[nt]);
                        acc[ms][nt][reg] = 0.f;
                        if (ids_on && rid == colid[nt] && row != cb0 + nt * 32 + l31) v += MIN_FLOAT;
                        sv[nt] = v * g.sm_inv_t;
                    }
                    if constexpr (EPI == 6) {
                        float m = -INFINITY;
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            if (cb0 + nt * 32 + l31 < g.N) m = fmaxf(m, sv[nt]);
#pragma unroll
                        for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                        float l = 0.f;
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            if (cb0 + nt * 32 + l31 < g.N) l += (sv[nt] - m < -87.f ? 0.f : __expf(sv[nt] - m));
#pragma unroll
                        for (int o = 1; o < 32; o <<= 1) l += __shfl_xor(l, o, 64);
                        if (rv && l31 == 0) {
                            const int64_t pc = tn0 / BN;
                            g.sm_part_m[pc * g.M + row] = m;
                            g.sm_part_l[pc * g.M + row] = l;
                        }
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            if (rv && row == cb0 + nt * 32 + l31) g.sm_pos[row] = sv[nt];
                    } else {
                        const float lse = __shfl(lse_l, src, 64), w = __shfl(w_l, src, 64);
                        const float scale = w * g.sm_inv_t * g.sm_alpha;
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt) {
                            const int col = cb0 + nt * 32 + l31;
                            const float d = sv[nt] - lse;
                            const float pr = (d < -87.f ? 0.f : __expf(d)) - (row == col ? 1.f : 0.f);
                            if (interior) g.C[row * g.ldc + col] = pr * scale;
                            else if (rv && col < g.N) g.C[row * g.ldc + col] = pr * scale;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);                  // one row at a time (hoisted, the exponentials' temporaries spill)
                }
            } else if constexpr (EPI == 5 && H2 != 0) {
                // The pack epilogue, second construction (round 5; the bf16x3 one below moves 4 bytes per lane and instruction and lost to
                // dgrad + pack by 27 %).  Each 32 x 32 accumulator block goes through a per-wave LDS buffer, row-major, and comes back
                // as float4s with 8 lanes on one row: a row's 32 columns of a field are ONE 128-byte line of its destination row and of
                // the x row it reads -- 16-byte loads and stores, 4 passes of 8 rows per block.  Loads unconditional (rows clamped);
                // stores unconditional on interior row tiles (a memory operation under a per-lane condition makes hipcc drain vmcnt
                // in front of each one); the first-order copy is written by all 8 lanes of a row (one address, one value).
                // (the launcher admits this epilogue for interior tiles only -- M a multiple of 256 -- with the FM term present and
                // destinations that fit 32 bits: no per-lane guards, no optional loads)
                const int kemb = min(64 * g.pack_F, g.N);
                float* stg = reinterpret_cast<float*>(smem + NS * STAGE + wave * PK_WAVE);
                const int64_t rw0 = r0 - 4 * hi;                                        // the wave's first row
                const int prow = lane >> 3, pd4 = (lane & 7) * 4;                       // storing pass: row within the 8-row group, column
                const int f0 = cb0 >> 6;                                                // the tile's first field (4 fields per 256 columns)
                // per tile: the wave's 32 x 4 destinations (x 64: float offsets) and its rows' d_fm_logit, through the LDS (lane -> row lane / 2,
                // fields 2 (lane & 1) + {0, 1}; lanes < 32 -> d_fm_logit of row lane)
                int* pk_pos = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(stg) + PK_STG);
                float* pk_dl = reinterpret_cast<float*>(pk_pos + 128);
                {
                    const int64_t prw = rw0 + (lane >> 1);
                    const int fa0 = min(f0 + 2 * (lane & 1), g.pack_F - 1), fa1 = min(f0 + 2 * (lane & 1) + 1, g.pack_F - 1);
                    const int64_t q0 = g.pack_pos[prw * g.pack_F + fa0], q1 = g.pack_pos[prw * g.pack_F + fa1];
                    const float dlr = g.pack_dl[rw0 + l31];
                    pk_pos[(lane >> 1) * 4 + 2 * (lane & 1)] = (int)q0 * 64;
                    pk_pos[(lane >> 1) * 4 + 2 * (lane & 1) + 1] = (int)q1 * 64;
                    if (hi == 0) pk_dl[l31] = dlr;
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int cg = cb0 + nt * 32;                                       // (uniform) first column of the block
                    if (cg >= kemb) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) acc[ms][nt][reg] = 0.f;
                        continue;
                    }
                    const int fi = nt >> 1, hb = nt & 1, dbase = 32 * hb;
                    float4 xv[4], sxv[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int64_t row = rw0 + 8 * it + prow;
                        xv[it] = *reinterpret_cast<const float4*>(g.xin + row * g.ldx + cg + pd4);
                        sxv[it] = *reinterpret_cast<const float4*>(g.pack_sumx + row * 64 + dbase + pd4);
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        stg[((reg & 3) + 8 * (reg >> 2) + 4 * hi) * PK_PITCH + l31] = acc[ms][nt][reg] * h2_out;
                        acc[ms][nt][reg] = 0.f;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (one wave, in-order LDS: written before it is read)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        float4 v = *reinterpret_cast<const float4*>(stg + (8 * it + prow) * PK_PITCH + pd4);
                        const int pp = pk_pos[(8 * it + prow) * 4 + fi];
                        const float dl = pk_dl[8 * it + prow];
                        v.x += dl * (sxv[it].x - xv[it].x); v.y += dl * (sxv[it].y - xv[it].y);
                        v.z += dl * (sxv[it].z - xv[it].z); v.w += dl * (sxv[it].w - xv[it].w);
                        *reinterpret_cast<float4*>(g.C + pp + dbase + pd4) = v;
                        if (hb == 0 && g.pack_lin != nullptr) g.pack_lin[pp >> 6] = dl;            // (uniform condition; 8 lanes, one address, one value)
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the block's reads retired before the next block's writes
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (EPI == 5) {
                // Every load unconditional (clamped).  The stores are under WAVE-UNIFORM conditions only: a 32-column group lies inside
                // one field or entirely behind the embeddings (64 F and the group width are multiples of 32), and all rows of an
                // interior row tile exist -- a store under a per-lane condition makes hipcc drain vmcnt in front of each one (the
                // first version, with per-lane guards: 627 us for the half batch instead of 175 + 150).  A group's 32 lanes write
                // one contiguous 128-byte piece of the destination row.
                const bool fmterm = g.pack_sumx != nullptr;
                const float* sxp = fmterm ? g.pack_sumx : g.pack_dl;
                const float* xp = fmterm ? g.xin : g.pack_dl;
                const int64_t xld = fmterm ? g.ldx : 0, sxld = fmterm ? 64 : 0;
                const int kemb = min(64 * g.pack_F, g.N);
                const bool rows_in = tm0 + BM <= g.M;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int cg = cb0 + nt * 32;                                       // (uniform) first column of the group
                    if (cg >= kemb) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) acc[ms][nt][reg] = 0.f;
                        continue;
                    }
                    const int col = cg + l31;
                    const int f = cg >> 6, d = col & 63;
                    const bool lin_here = (cg & 63) == 0 && g.pack_lin != nullptr;      // (uniform) the group that holds d == 0
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        int64_t pp[4];
                        float dlv[4], sxv[4], xv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int64_t row = r0 + e + 8 * c4;
                            const int64_t rc = row < g.M ? row : g.M - 1;
                            pp[e] = g.pack_pos[rc * g.pack_F + f];
                            dlv[e] = g.pack_dl[rc];
                            sxv[e] = sxp[rc * sxld + (fmterm ? d : 0)];
                            xv[e] = xp[rc * xld + (fmterm ? col : 0)];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int reg = 4 * c4 + e;
                            float v = acc[ms][nt][reg];
                            acc[ms][nt][reg] = 0.f;
                            if (fmterm) v += dlv[e] * (sxv[e] - xv[e]);
                            if (rows_in) {
                                g.C[pp[e] * 64 + d] = v;
                                if (lin_here) g.pack_lin[pp[e]] = dlv[e];               // 32 lanes, one address, one value
                            } else if (r0 + e + 8 * c4 < g.M) {
                                g.C[pp[e] * 64 + d] = v;
                                if (lin_here) g.pack_lin[pp[e]] = dlv[e];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);     // (four rows at a time: hoisted across groups, the loads spill 93 registers)
                    }
                }
            } else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int col = cb0 + nt * 32 + l31;
                const bool cv = col < g.N;
                float bj = g.bias != nullptr ? g.bias[cv ? col : g.N - 1] : 0.f;
                asm volatile("" : "+v"(bj));      // consume the load on every path (see bf3_gemm_nt_pipe_kernel)
                if (interior) {
                    float* crow = g.C + r0 * g.ldc + col;
                    if constexpr (cross) {
                        const int64_t xo = r0 * g.ldx + col;
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {             // four rows at a time: short-lived temporaries (the 128
                            float x0v[4], xv[4];                      // accumulators are all live here)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                x0v[e] = g.x0[xo + (int64_t)(e + 8 * c4) * g.ldx];
                                xv[e] = g.xin[xo + (int64_t)(e + 8 * c4) * g.ldx];
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int reg = 4 * c4 + e;
                                const float pr = (H2 ? fmaf(acc[ms][nt][reg], h2_out, bj) : acc[ms][nt][reg] + bj) + g.diag * xv[e];
                                acc[ms][nt][reg] = 0.f;
                                if (g.prod_out != nullptr) g.prod_out[xo + (int64_t)(e + 8 * c4) * g.ldx] = pr;
                                const float o = fmaf(x0v[e], pr, xv[e]);
                                if constexpr (H2) cmax = fmaxf(cmax, fabsf(o));
                                crow[(int64_t)(e + 8 * c4) * g.ldc] = o;
                            }
                        }
                    } else if constexpr (EPI == 1 || EPI == 3) {
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            float aux[4];                              // EPI 1: the mask values, EPI 3: the old outputs
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                aux[e] = EPI == 1 ? g.mask[(r0 + e + 8 * c4) * g.ld_mask + col] : crow[(int64_t)(e + 8 * c4) * g.ldc];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int reg = 4 * c4 + e;
                                float v = H2 ? fmaf(acc[ms][nt][reg], h2_out, bj) : acc[ms][nt][reg] + bj;
                                acc[ms][nt][reg] = 0.f;
                                v = relu ? fmaxf(v, 0.f) : v;
                                const float o = EPI == 1 ? (aux[e] > 0.f ? v : 0.f) : aux[e] + v;
                                if constexpr (H2) cmax = fmaxf(cmax, fabsf(o));
                                crow[(int64_t)(e + 8 * c4) * g.ldc] = o;
                            }
                        }
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            float v = H2 ? fmaf(acc[ms][nt][reg], h2_out, bj) : acc[ms][nt][reg] + bj;
                            acc[ms][nt][reg] = 0.f;
                            if constexpr (H2) cmax = fmaxf(cmax, relu ? fmaxf(v, 0.f) : fabsf(v));
                            if (DR_NT_DGRAD_STORE && EPI == 0) __builtin_nontemporal_store(relu ? fmaxf(v, 0.f) : v, &crow[(int64_t)((reg & 3) + 8 * (reg >> 2)) * g.ldc]);
                            else crow[(int64_t)((reg & 3) + 8 * (reg >> 2)) * g.ldc] = relu ? fmaxf(v, 0.f) : v;
                        }
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t row = r0 + (reg & 3) + 8 * (reg >> 2);
                        float v = H2 ? fmaf(acc[ms][nt][reg], h2_out, bj) : acc[ms][nt][reg] + bj;
                        acc[ms][nt][reg] = 0.f;
                        if (!cv || row >= g.M) continue;
                        float* dst = g.C + row * g.ldc + col;
                        if constexpr (cross) {
                            const float xv = g.xin[row * g.ldx + col];
                            const float pr = v + g.diag * xv;
                            if (g.prod_out != nullptr) g.prod_out[row * g.ldx + col] = pr;
                            const float o = fmaf(g.x0[row * g.ldx + col], pr, xv);
                            if constexpr (H2) cmax = fmaxf(cmax, fabsf(o));
                            *dst = o;
                            continue;
                        }
                        v = relu ? fmaxf(v, 0.f) : v;
                        if constexpr (EPI == 1) {
                            if (!(g.mask[row * g.ld_mask + col] > 0.f)) v = 0.f;
                        } else if constexpr (EPI == 3) {
                            v = *dst + v;
                        }
                        if constexpr (H2) cmax = fmaxf(cmax, fabsf(v));
                        *dst = v;
                    }
                }
                // MS == 2: all 256 accumulators live in AGPRs; without a fence per column tile the scheduler copies them out
                // ahead of the stores and spills 265 registers
                if constexpr (MS == 2) __builtin_amdgcn_sched_barrier(0);
            }
            }   // EPI != 4
          }     // ms
            // Stores and loads share vmcnt on gfx9 and hipcc treats a mix of the two as unordered: left pending into the next
            // k-tile, the stores turn its wait for the A registers into vmcnt(0) on EVERY k-tile.  Draining here costs the
            // pipeline one refill per output tile instead.
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    }
    if (STG && !grp) asm volatile("s_barrier" ::: "memory");            // the leading group's extra barrier (see STG above)
    if constexpr (H2 && EPI != 4) {
        if (g.c_amax != nullptr) {                                      // (kernel-uniform) one load per wave, an atomic only if it raises the record
            uint32_t m = __float_as_uint(cmax);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
            if (lane == 0 && m > __hip_atomic_load(g.c_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g.c_amax, m);
        }
    }
}

// =====================================================================================================================
// Fused first layer of the DeepFM / DCN tower:  hash ids -> [gather + stack/concat + first-order + FM second-order] -> Dense.
// The register-split GEMM (bf3_gemm_rs_kernel's tile: 256 x 256 x 32, 8 waves x 32 rows, weights pre-split in an LDS ring) whose
// activation operand is GATHERED: k-tile kt of example b is dims 32 (kt & 1) .. + 31 of table row row_base[f] + ids[b][f],
// f = kt >> 1 (D == 64) -- the same 128-byte lines K3 (dr_emb_pool_fwd) reads -- so the kernel that multiplies the concatenated
// embeddings by the first Dense kernel is also the one that fetches them.  On their way through the CU the values are (1) stored
// to `concat` (the backward kernels read it) and (2) summed into the FM terms K3 produced (sum_x, sum of squares, first-order
// weights): keras/models/ranking/fm.py:23-37 and deepfm.py:36-47 of the reference in ONE launch with deepfm.py:30-34's first
// Dense.  K3's 0.9 GB of HBM traffic moves in the shadow of an MFMA-bound kernel instead of in a 220 us kernel of its own.
//
// How the rows travel (what the first two versions of this kernel taught):
//   * A load in which every lane addresses its own row costs the CU's address path one request per LANE: 256 per wave and
//     k-tile for the row loads, as many again for the concat stores.  With ids and first-order weights on top the REQUEST rate
//     -- not HBM, not the matrix pipe -- set the pace: 437 us (273 for the plain GEMM + 218 for K3 = 491), and 488 us with an
//     L2 prefetch added (+128 requests per wave and k-tile => +0.9 us per k-tile).
//   * So the rows are fetched by LDS-DMA, 8 lanes x 16 bytes per 128-byte line (8 requests per instruction, 32 per wave and
//     k-tile), two k-tiles ahead, into a wave-private 2 x 4 KB LDS image -- no destination VGPRs, which is what allows the
//     two-step lead.  Each field is addressed as a STRUCTURED buffer (base = its first row, index = bucket id, stride 256 B;
//     buffer offsets are 32 bits wide, so a field may have 2^24 rows).  The image is XOR-swizzled by the choice of which (row, chunk) each DMA lane fetches, so that
//     both readers are conflict-free: the MFMA-operand read (lane = row, 4 x ds_read_b128) and the position-wise read that
//     feeds the COALESCED concat stores (8 lanes per 128-byte line again).
//   * LDS: 2 x 48 KB weight stages + 2 x 32 KB activation stages = all 160 KB.  The weight pieces of k-tile s + 1 are issued first
//     in step s, the gather of s + 2 and the id / weight loads after them, so the one counted wait per step (vmcnt(10) in front
//     of the barrier) covers exactly the pieces and leaves the younger gather in flight.
// Single-valued fields, D == 64, at most 32 dense features (one more k-tile, read from `dense_pad` [M, 32], zero-padded; the
// caller also places them in concat[:, 64 F : K) for the backward).  Any N works: only the first column tile of a row panel
// stores the side outputs.
// =====================================================================================================================
struct EmbArgs {
    const int64_t* ids; int32_t F;               // [M, F] bucket ids (-1 = missing -> zero embedding, no first-order term)
    const int64_t* row_base;                     // [F] first row of each field in the slab
    const float* table;                          // [R, 64]
    const float* lin_w; const float* lin_bias;   // first-order weights [R] / bias [1] (may be null)
    const float* dense_pad;                      // [M, 32] dense features, zero-padded (null when K == 64 F)
    float* concat; int64_t ld_concat;            // out: [M, ld], columns [0, 64 F)
    float* sum_x; float* fm_logit;               // out: [M, 64], [M]
    float* lin_vals;                             // out (may be null): [F, M] field-major, the first-order weight every slot read -- K4
                                                 // then only WRITES lin_w[row] (one line operation per slot instead of two; round 4)
    const uint32_t* dense_amax;                  // f16x2 mode: amax record of dense_pad (null: none); the table's is RsArgs::a_amax
};

__device__ __forceinline__ int sload_i32(const void* base, int byte_off) {      // scalar load of a wave-uniform word, on the spot
    int v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(byte_off) : "memory");
    return v;
}

// H2: the f16x2 operand mode (h2_split8): two fp16 weight planes (32 KB stages, 4 pieces per wave and k-tile), three MFMAs per
// fragment pair; the activation scale comes from the larger of the table's and the dense features' amax records.
template <int H2>
__global__ __launch_bounds__(512, 2) void bf3_emb_linear_kernel(RsArgs g, EmbArgs e) {
    constexpr int NW = 8, BM = 32 * NW, BN = 256, NT = BN / 32, NS = 2;
    constexpr int NPL = H2 ? 2 : 3;
    constexpr int B_PLANE = BN * 64;                                    // bytes: 256 rows x 64-byte rows (32 bf16 / fp16)
    constexpr int STAGE = NPL * B_PLANE;                                // 48 KB (32 KB)
    constexpr int PW = STAGE / 1024 / NW;                               // 6 (4) LDS-DMA pieces per wave and k-tile
    constexpr int A_WAVE = 32 * 128, A_STAGE = NW * A_WAVE;             // 4 KB per wave, 32 KB per stage
    constexpr int A_BASE = NS * STAGE;
    static_assert(PW == 2 * NPL, "piece schedule below assumes 2 pieces per plane, wave and k-tile");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE + 2 * A_STAGE];
    typedef float f32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int grow = lane >> 3;                                         // gather layout: DMA i of this lane fetches row 8 i + grow,
    const int gchunk = (lane & 7) ^ grow;                               // 16-byte chunk gchunk (image slot lane & 7: XOR swizzle)

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int nk = (g.K + BK - 1) / BK;
    const int nke = 2 * e.F;                                            // gathered k-tiles (nk == nke or nke + 1)
    if ((int)blockIdx.x >= ntiles) return;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nk;                                    // steps of this block

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int sw = (l31 >> 2) & 3;
    unsigned b_addr[2];                                                 // B fragment reads: row 32 nt + l31, chunk 2 hi + s swizzled
#pragma unroll
    for (int s = 0; s < 2; ++s) b_addr[s] = lds0 + l31 * 64 + (((2 * hi + s) ^ sw) << 4);
    // A image of this wave: position p = 8 row + (chunk ^ (row & 7)), 16 bytes each
    const unsigned a_rd = lds0 + A_BASE + wave * A_WAVE + l31 * 128 + (((4 * hi) ^ (l31 & 7)) << 4);    // own row, chunk 4 hi (^ c << 4)
    const unsigned a_st = lds0 + A_BASE + wave * A_WAVE + lane * 16;                                    // position 64 i + lane

    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(g.B), 0, (int)min((int64_t)0x7fffffff, NPL * g.b_ps * 2), 0x00020000);
    float h2_sa = 1.f, h2_out = 1.f;                                    // H2: the activations' scale, 1 / (s_a s_b)
    if constexpr (H2) {
        float ia, sb, ib;
        h2_scale_of(max(g.a_amax[0], e.dense_amax != nullptr ? e.dense_amax[0] : 0u), h2_sa, ia);
        h2_scale_of(g.b_amax[0], sb, ib);
        h2_out = ia * ib;
        h2_mode_on();
    }
    // (the table resource is built per FIELD, base = its first row: a buffer offset -- index x stride included -- is 32 bits
    // wide, so one resource reaches 4 GB = 2^24 rows; a resource over the whole 66 GB slab wraps, measured)
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(e.dense_pad != nullptr ? e.dense_pad : e.table), 128, 0x7fffffff, 0x00020000);
    const int b_lane = (int)((((int64_t)(wave * 16 + (lane >> 2))) * g.b_ld + ((lane & 3) ^ ((((wave * 16 + (lane >> 2))) >> 2) & 3)) * 8) * 2);

    // ---- the block's stream of steps (tile, k-tile): iterators for steps s + 1, s + 2, s + 3 ---------------------------------
    auto m0_of = [&](int tile) -> int { return (xcd_remap(tile, ntiles) / tiles_n) * BM; };
    auto n0_of = [&](int tile) -> int { return (xcd_remap(tile, ntiles) % tiles_n) * BN; };
    int kt1, tile1, m01, kt2, tile2, m02, kt3, tile3, m03;
    auto advance = [&](int& kt, int& tile, int& m0) {                   // past the end of the stream: stay on the last step
        if (kt + 1 < nk) { ++kt; return; }
        if (tile + (int)gridDim.x < ntiles) { tile += gridDim.x; kt = 0; m0 = m0_of(tile); }
    };
    auto grow_row = [&](int m0, int i) -> int { return min(m0 + wave * 32 + 8 * i + grow, (int)g.M - 1); };
    auto own_row = [&](int m0) -> int { return min(m0 + wave * 32 + l31, (int)g.M - 1); };
    const int* ids32 = reinterpret_cast<const int*>(e.ids);             // low words: bucket ids fit 31 bits, -1 stays negative
    int idg[4], ido = 0;                                                // ids in flight: gather layout (step s + 3), own row (step s + 2)
    auto load_idg = [&](int kt, int m0) {
        const int f = min(kt >> 1, e.F - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) idg[i] = ids32[2 * ((int64_t)grow_row(m0, i) * e.F + f)];
    };
    auto load_ido = [&](int kt, int m0) { ido = ids32[2 * ((int64_t)own_row(m0) * e.F + min(kt >> 1, e.F - 1))]; };
    int rb_next = 0;                                                    // row_base of step s + 1's field (step s + 2's when loaded)
    int m4q = 0;                                                        // missing bits of the gathers in flight: step s low nibble, s + 1 next
    bool mo_cur = false, mo_nxt = false;                                // own row missing: step s / s + 1
    float lwn = 0.f;                                                    // own row's first-order weight of the next step
    // gather of step (kt, m0) into A stage `ast` from the ids in idg; returns the 4 missing bits
    auto issue_gather = [&](int kt, int m0, int rb, int ast) -> int {
        const bool dense = kt >= nke;                                   // (wave-uniform)
        const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(e.table + (int64_t)rb * 64), 256, 0x7fffffff, 0x00020000);
        int m4 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool miss = !dense && idg[i] < 0;
            m4 |= (miss ? 1 : 0) << i;
            const int idx = dense ? grow_row(m0, i) : max(idg[i], 0);
            unsigned char* dst = smem + A_BASE + ast * A_STAGE + wave * A_WAVE + i * 1024;
            // (one DMA with a selected resource, not one under each arm of a branch: with the branch hipcc's wait for anything
            // older than these DMAs comes out as vmcnt(0))
            __builtin_amdgcn_struct_ptr_buffer_load_lds(dense ? drsrc : trsrc, (lds_ptr_t)dst, 16, idx,
                                                        gchunk * 16 + (dense ? 0 : (kt & 1) * 128), 0, 0, DR_NT_FWD_GATHER ? 2 : 0);
        }
        return m4;
    };
    // (without first-order weights the load still happens, from the table: every step issues the same number of VMEM operations,
    // which is what makes the counted wait in front of the barrier a constant)
    const bool has_lw = e.lin_w != nullptr;
    const float* const lwp = has_lw ? e.lin_w : e.table;
    auto issue_lw = [&](int kt, int rb) {                               // own row of step (kt, .): first-order weight, missing flag
        const bool dense = kt >= nke;
        mo_nxt = !dense && ido < 0;
        lwn = lwp[dense ? 0 : rb + max(ido, 0)];
    };
    auto issue_b = [&](int i, int kt, int n0, int stage) {              // piece wave + 8 i of step (kt, tile with column base n0)
        unsigned char* dst = smem + stage * STAGE + (wave + NW * i) * 1024;
        const int uni = (int)(((int64_t)(i >> 1) * g.b_ps + ((int64_t)(i & 1) * 128 + n0) * g.b_ld) * 2) + kt * (BK * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)dst, 16, b_lane, uni, 0, 0);
    };

    float S0[16], S1[16], ssq = 0.f, lin = 0.f;                         // FM terms of the lane's row (its 16 dims of each half row)
#pragma unroll
    for (int j = 0; j < 16; ++j) { S0[j] = 0.f; S1[j] = 0.f; }
    bf16x8 fa[2][3];                                                    // [k-step][plane] of the CURRENT step
    bf16x8 fb[2][3];                                                    // [buffer][plane]: group q uses buffer q & 1 (one group ahead)
    auto read_b = [&](int buf, int stage, int q) {                      // group q = (k-step q >> 3, column tile q & 7)
        const unsigned bb = b_addr[q >> 3] + stage * STAGE;
        const int nt = q & 7;
#define RS_READ3(NTI)                                                        \
        BF3_DS_READ_B128(fb[buf][0], bb, 0 * B_PLANE + NTI * 2048);          \
        BF3_DS_READ_B128(fb[buf][1], bb, 1 * B_PLANE + NTI * 2048);          \
        if constexpr (!H2) BF3_DS_READ_B128(fb[buf][2], bb, 2 * B_PLANE + NTI * 2048);
        switch (nt) {
            case 0: RS_READ3(0) break; case 1: RS_READ3(1) break; case 2: RS_READ3(2) break; case 3: RS_READ3(3) break;
            case 4: RS_READ3(4) break; case 5: RS_READ3(5) break; case 6: RS_READ3(6) break; default: RS_READ3(7) break;
        }
#undef RS_READ3
    };
    auto wait_b = [&](int buf) {
        if constexpr (H2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]));
    };
    f32x16 acc[NT];
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};

    // ---- prologue: pieces of step 0, gathers of steps 0 and 1, ids of step 2 in flight --------------------------------------
    int kt = 0, tile = blockIdx.x, m0c = m0_of(tile);                   // consumer: step s
    kt1 = 0; tile1 = tile; m01 = m0c;
#pragma unroll
    for (int i = 0; i < PW; ++i) issue_b(i, 0, n0_of(tile), 0);
    {
        int rb = sload_i32(e.row_base, 0);
        load_idg(0, m0c);
        load_ido(0, m0c);
        m4q = issue_gather(0, m0c, rb, 0);
        issue_lw(0, rb);
        mo_cur = mo_nxt;
        advance(kt1, tile1, m01);                                       // step 1
        rb = sload_i32(e.row_base, 8 * min(kt1 >> 1, e.F - 1));
        load_idg(kt1, m01);
        m4q |= issue_gather(kt1, m01, rb, 1) << 4;
        load_ido(kt1, m01);                                             // consumed by step 0's clump (own row of step 1)
        rb_next = rb;
        kt2 = kt1; tile2 = tile1; m02 = m01;
        advance(kt2, tile2, m02);                                       // step 2
        load_idg(kt2, m02);                                             // consumed by step 0's clump (gather of step 2)
        kt3 = kt2; tile3 = tile2; m03 = m02;
        advance(kt3, tile3, m03);                                       // step 3
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                 // everything landed (once per block)
    asm volatile("s_barrier" ::: "memory");
    read_b(0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    int stage = 0;
    for (int step = 0; step < total; ++step) {
        const int astage = step & 1;
        const bool gathered = kt < nke;                                 // (wave-uniform)
        const bool c_fm = (xcd_remap(tile, ntiles) % tiles_n) == 0;     // the first column tile of a row panel owns concat / FM
        // ---- step start: this step's rows out of the LDS image ----------------------------------------------------------------
        f32x4 an[4];
        {
            const unsigned ra = a_rd + astage * A_STAGE;
            const unsigned r1 = ra ^ 16u, r2 = ra ^ 32u, r3 = ra ^ 48u;
            BF3_DS_READ_B128(an[0], ra, 0); BF3_DS_READ_B128(an[1], r1, 0);
            BF3_DS_READ_B128(an[2], r2, 0); BF3_DS_READ_B128(an[3], r3, 0);
        }
        if (gathered && c_fm && e.concat != nullptr) {                     // (kernel-uniform: concat == NULL skips the stores)
            // the image position-wise (8 lanes per 128-byte line) -> concat, for the backward kernels; missing ids store zeros
            const unsigned sa = a_st + astage * A_STAGE;
            f32x4 st[4];
            BF3_DS_READ_B128(st[0], sa, 0); BF3_DS_READ_B128(st[1], sa, 1024);
            BF3_DS_READ_B128(st[2], sa, 2048); BF3_DS_READ_B128(st[3], sa, 3072);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0c + wave * 32 + 8 * i + grow;
                if (row < g.M) {
                    float* dst = e.concat + (int64_t)row * e.ld_concat + kt * BK + 4 * gchunk;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 v = ((m4q >> i) & 1) ? z : st[i];
                    // inline asm on purpose: stores the compiler can see make it treat vmcnt as unordered (loads + stores
                    // pending) and wait vmcnt(0) for everything in flight
                    // (s_nop: a VALU write to the data registers of a > 64-bit store needs a wait state after the store; the
                    // hazard recogniser does not look inside inline asm, and the next instruction did reuse v.x)
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
                }
            }
        }
        // ("memory": the gather DMA that refills this A stage further down must not be moved above these reads)
        if constexpr (H2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an[0]), "+v"(an[1]), "+v"(an[2]), "+v"(an[3]), "+v"(fb[0][0]), "+v"(fb[0][1]) :: "memory");
        else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an[0]), "+v"(an[1]), "+v"(an[2]), "+v"(an[3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2])
                     :: "memory");
        if (gathered && mo_cur) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            an[0] = z; an[1] = z; an[2] = z; an[3] = z;
        }
        if (gathered && c_fm) {
            if ((kt & 1) == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { S0[4 * q] += an[q][0]; S0[4 * q + 1] += an[q][1]; S0[4 * q + 2] += an[q][2]; S0[4 * q + 3] += an[q][3]; }
                lin += (hi == 0 && !mo_cur && has_lw) ? lwn : 0.f;
                if (e.lin_vals != nullptr) {                                // (kernel-uniform)
                    // lanes l and l + 32 hold the same row's weight: both store it (same address, same value) -- no divergent
                    // branch around a memory operation; asm for the reason given at the concat stores above
                    float* lv = e.lin_vals + (int64_t)(kt >> 1) * g.M + min(m0c + wave * 32 + l31, (int)g.M - 1);
                    asm volatile("global_store_dword %0, %1, off" :: "v"(lv), "v"(lwn) : "memory");
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { S1[4 * q] += an[q][0]; S1[4 * q + 1] += an[q][1]; S1[4 * q + 2] += an[q][2]; S1[4 * q + 3] += an[q][3]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) ssq += (an[q][0] * an[q][0] + an[q][1] * an[q][1]) + (an[q][2] * an[q][2] + an[q][3] * an[q][3]);
        }
        asm volatile("" : "+v"(lwn));                                   // the weight load is consumed on every path
        {
            const float4 a0 = make_float4(an[0][0], an[0][1], an[0][2], an[0][3]), a1 = make_float4(an[1][0], an[1][1], an[1][2], an[1][3]);
            const float4 a2 = make_float4(an[2][0], an[2][1], an[2][2], an[2][3]), a3 = make_float4(an[3][0], an[3][1], an[3][2], an[3][3]);
            if constexpr (H2) {
                h2_split8(a0, a1, h2_sa, fa[0][0], fa[0][1]);
                h2_split8(a2, a3, h2_sa, fa[1][0], fa[1][1]);
            } else {
                rs_split8(a0, a1, fa[0][0], fa[0][1], fa[0][2]);
                rs_split8(a2, a3, fa[1][0], fa[1][1], fa[1][2]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int nstage = stage ^ 1;
        const int n01 = n0_of(tile1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            wait_b(q & 1);
            if (q < 15) {
                read_b((q + 1) & 1, stage, q + 1);
            } else {
                // this wave is done reading the stages of step `step`; publish step + 1.  vmcnt(10): the 6 weight pieces of step + 1
                // (and everything older: the gather of step + 1) have landed, the clump issued after them (5 id loads, the
                // first-order weight, the 4 gather DMAs of step + 2) stays in flight
                __builtin_amdgcn_s_waitcnt(0x0F70 | 10);
                asm volatile("s_barrier" ::: "memory");
                if (step + 1 < total) read_b(0, nstage, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (H2) {
                acc[q & 7] = h2_mfma(fa[q >> 3][0], fb[q & 1][1], acc[q & 7]);
                acc[q & 7] = h2_mfma(fa[q >> 3][1], fb[q & 1][0], acc[q & 7]);
                acc[q & 7] = h2_mfma(fa[q >> 3][0], fb[q & 1][0], acc[q & 7]);
            } else {
#pragma unroll
            for (int term = 0; term < 6; ++term)
                acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q >> 3][PA[term]], fb[q & 1][PB[term]], acc[q & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);    // keeps the next group's lgkmcnt wait from being hoisted between these MFMAs
            if (q < PW) {
                issue_b(q, kt1, n01, nstage);                           // weight pieces of step + 1 (a dummy re-fetch at the end of the stream)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == PW) {
                // the clump: ids first (they are needed one step from now), then the weight, then the DMAs -- a wait for an
                // older operation never forces a younger one
                const int rb1 = rb_next;                                // field of step + 1
                const int rb2 = sload_i32(e.row_base, 8 * min(kt2 >> 1, e.F - 1));
                const int ido_use = ido;
                int idg_use[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) idg_use[i] = idg[i];
                load_idg(kt3, m03);                                     // gather layout, step + 3
                load_ido(kt2, m02);                                     // own row, step + 2
                {   // own row of step + 1
                    const bool dense = kt1 >= nke;
                    mo_nxt = !dense && ido_use < 0;
                    lwn = lwp[dense ? 0 : rb1 + max(ido_use, 0)];
                }
                {   // gather of step + 2 into the A stage this step has just consumed
                    const bool dense = kt2 >= nke;
                    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(e.table + (int64_t)rb2 * 64), 256, 0x7fffffff, 0x00020000);
                    int m4 = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool miss = !dense && idg_use[i] < 0;
                        m4 |= (miss ? 1 : 0) << i;
                        const int idx = dense ? grow_row(m02, i) : max(idg_use[i], 0);
                        unsigned char* dst = smem + A_BASE + astage * A_STAGE + wave * A_WAVE + i * 1024;
                        __builtin_amdgcn_struct_ptr_buffer_load_lds(dense ? drsrc : trsrc, (lds_ptr_t)dst, 16, idx,
                                                                    gchunk * 16 + (dense ? 0 : (kt2 & 1) * 128), 0, 0, DR_NT_FWD_GATHER ? 2 : 0);
                    }
                    m4q = (m4q >> 4) | (m4 << 4);
                }
                rb_next = rb2;
                kt1 = kt2; tile1 = tile2; m01 = m02;
                kt2 = kt3; tile2 = tile3; m02 = m03;
                advance(kt3, tile3, m03);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mo_cur = mo_nxt;
        stage = nstage;
        if (++kt < nk) continue;
        // ---- epilogue of an output tile: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        kt = 0;
        {
            const int lid = xcd_remap(tile, ntiles);
            const int64_t tm0 = (int64_t)(lid / tiles_n) * BM;
            const int tn0 = (lid % tiles_n) * BN;
            const bool relu = g.act == 1;
            const int64_t r0 = tm0 + wave * 32 + 4 * hi;
            // interior tiles: every load / store of the epilogue unconditional (a memory operation under a divergent branch makes
            // hipcc wait vmcnt(0) in front of each one); edge tiles take the guarded loop
            const bool interior = tm0 + BM <= g.M && tn0 + BN <= g.N;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = tn0 + nt * 32 + l31;
                const bool cv = col < g.N;
                float bj = g.bias != nullptr ? g.bias[cv ? col : g.N - 1] : 0.f;
                asm volatile("" : "+v"(bj));      // consume the load on every path (see bf3_gemm_nt_pipe_kernel)
                if (interior) {
                    float* crow = g.C + r0 * g.ldc + col;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = H2 ? fmaf(acc[nt][reg], h2_out, bj) : acc[nt][reg] + bj;
                        acc[nt][reg] = 0.f;
                        crow[(int64_t)((reg & 3) + 8 * (reg >> 2)) * g.ldc] = relu ? fmaxf(v, 0.f) : v;
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int64_t row = r0 + (reg & 3) + 8 * (reg >> 2);
                        float v = H2 ? fmaf(acc[nt][reg], h2_out, bj) : acc[nt][reg] + bj;
                        acc[nt][reg] = 0.f;
                        if (!cv || row >= g.M) continue;
                        g.C[row * g.ldc + col] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
            }
            if (c_fm) {
                // this row panel's FM outputs (keras/models/ranking/fm.py:28-37): sum_x for the backward, the logit part
                float t2 = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) t2 += S0[j] * S0[j] + S1[j] * S1[j];
                t2 += __shfl_xor(t2, 32, 64);
                const float ss_all = ssq + __shfl_xor(ssq, 32, 64);
                const int64_t row = tm0 + wave * 32 + l31;
                if (row < g.M) {
                    float* sx = e.sum_x + row * 64 + 16 * hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        *reinterpret_cast<float4*>(sx + 4 * q) = make_float4(S0[4 * q], S0[4 * q + 1], S0[4 * q + 2], S0[4 * q + 3]);
                        *reinterpret_cast<float4*>(sx + 32 + 4 * q) = make_float4(S1[4 * q], S1[4 * q + 1], S1[4 * q + 2], S1[4 * q + 3]);
                    }
                    if (hi == 0) e.fm_logit[row] = (e.lin_bias != nullptr ? e.lin_bias[0] : 0.f) + lin + 0.5f * (t2 - ss_all);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) { S0[j] = 0.f; S1[j] = 0.f; }
                ssq = 0.f;
                lin = 0.f;
            }
            // Stores and loads share vmcnt on gfx9 and hipcc treats a mix of the two as unordered: left pending into the next
            // k-tile, the stores turn every wait of the loop into vmcnt(0).  Draining here costs one refill per output tile.
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        tile += gridDim.x;
        if (tile < ntiles) m0c = m0_of(tile);
    }
}

// =====================================================================================================================
// TN, "register split" wgrad:  dst[f][n] += scale * sum_r X[r][f] Y[r][n],  dstb[n] += scale * sum_r Y[r][n]
// Both operands are fp32 ACTIVATIONS, reduction-major (x [R, F] and dy [R, N] as their producers write them); nothing is
// pre-split.  A block owns a 256 (f) x 256 (n) tile of one reduction slice; each of its 8 waves owns 32 f x all 256 n.
//   A (x):  every lane loads the 16 reduction elements of ITS column f for a k-tile with 16 dword loads (the 32 lanes of a half
//           wave cover 128 contiguous bytes of one row) and splits them in registers -- the transposition MFMA's A operand
//           needs ("8 consecutive k per lane") is free because the lane index runs along f.
//   B (dy): the same loads with the lane index along n give every lane 8 consecutive r of its column, i.e. exactly one
//           16-byte fragment chunk per plane: wave w splits the 32 columns 32 w .. 32 w + 31 of the tile once and writes the
//           three planes as ds_write_b128 into the [n][32 r] image the fragment reads of the NT kernels use; the 8 waves'
//           pieces make the tile that all of them read.  Two LDS stages: step g's MFMAs read stage g while stage g + 1 is
//           written from registers loaded during step g - 1.
// The in-kernel-split wgrad of dense.hip re-stages BOTH operands through the LDS per 128 x 128 tile; here x never touches
// it and each dy element is split once per 256 rows of x.
// Partials go to a padded workspace [split][tiles_f * 256][tiles_n * 256] (+ [split][tiles_n * 256] column sums): every store of
// the epilogue is unconditional; a fixed-order reduce applies them (deterministic).
// =====================================================================================================================
struct TnRsArgs {
    const float* X; int64_t ldx;
    const float* Y; int64_t ldy;
    int64_t R; int32_t F; int32_t N;
    int64_t per; int32_t split;                  // reduction rows per slice (multiple of 32), number of slices
    float* partial; float* colsum;               // [split][Fp][Np], [split][Np] (colsum may be null)
    // GATHER form (the first layer of the DeepFM / DCN tower): X is never materialised -- column c < 64 nf of reduction row r is
    // element (c & 63) of table row row_base[c >> 6] + ids_t[c >> 6][r] (zero for a missing id), columns [64 nf, F) come from
    // dense_pad[r][c - 64 nf]
    const int32_t* ids_t; const int64_t* row_base; const float* table; int32_t nf; const float* dense_pad;
    // f16x2 mode: amax records of x (GATHER: the table's; x2 = the dense features', may be null) and of dy
    const uint32_t* x_amax; const uint32_t* x2_amax; const uint32_t* y_amax;
};

__device__ __forceinline__ void rs_split8v(const float (&v)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    rs_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p0, p1, p2);
}

__device__ __forceinline__ void h2_split8v(const float (&v)[8], float s, bf16x8& p0, bf16x8& p1) {
    h2_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), s, p0, p1);
}

// H2: the f16x2 operand mode (h2_split8): both operands split into two fp16 terms with their tensors' scales; the partials carry
// s_x s_y and the reduce kernel divides it out.
template <int GATHER, int H2 = 0>
__global__ __launch_bounds__(512, 2) void bf3_gemm_tn_rs_kernel(TnRsArgs g) {
    constexpr int NW = 8, BMF = 32 * NW, BN = 256, NT = BN / 32;
    constexpr int NPL = H2 ? 2 : 3;
    constexpr int B_PLANE = BN * 64;                                    // bytes: 256 n-rows x 64 bytes (32 r)
    constexpr int STAGE = NPL * B_PLANE;                                // 48 KB (32 KB)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_f = (g.F + BMF - 1) / BMF;
    const int per_slice = tiles_f * tiles_n;
    const int Fp = tiles_f * BMF, Np = tiles_n * BN;
    const int lid = xcd_remap(blockIdx.x, per_slice * g.split);         // consecutive logical ids share a reduction slice
    const int slice = lid / per_slice, t = lid % per_slice;
    const int f0 = (t / tiles_n) * BMF, n0 = (t % tiles_n) * BN;
    const int64_t r_begin = (int64_t)slice * g.per;
    int64_t r_end = r_begin + g.per;
    if (r_end > g.R) r_end = g.R;
    const int nk = (int)((r_end - r_begin + BK - 1) / BK);              // >= 1 by construction of split
    const bool want_cs = g.colsum != nullptr && f0 == 0;
    float h2_sx = 1.f, h2_sy = 1.f;
    if constexpr (H2) {
        float inv;
        h2_scale_of(max(g.x_amax[0], g.x2_amax != nullptr ? g.x2_amax[0] : 0u), h2_sx, inv);
        h2_scale_of(g.y_amax[0], h2_sy, inv);
        h2_mode_on();
    }

    // this lane's columns (clamped: columns past the edge only feed outputs nobody reads)
    const int fcol = min(f0 + wave * 32 + l31, g.F - 1);
    const int ncol = min(n0 + wave * 32 + l31, g.N - 1);
    // fragment read addresses as in bf3_gemm_rs_kernel: B row = 32 nt + l31, chunk (2 hi + s) ^ ((row >> 2) & 3)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    const int sw = (l31 >> 2) & 3;
    unsigned b_addr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) b_addr[s] = lds0 + l31 * 64 + (((2 * hi + s) ^ sw) << 4);
    // where this lane writes its own column's chunks: row 32 wave + l31 of the image, same swizzle
    unsigned char* const wrow = smem + (wave * 32 + l31) * 64;

    // raw operand values of one k-tile: element e = 8 s + j  <->  reduction row r0 + 16 hi + 8 s + j   (the k permutation of
    // the RS kernels: lane half hi holds k = 16 hi .. 16 hi + 15, k-step s uses 16 hi + 8 s .. + 7 = chunk 2 hi + s)
    float xa[16], yb[16];
    // per-lane byte offset inside a k-tile's rows (32-bit) + a wave-uniform row pointer per load: one address register, the
    // row stepping stays on the scalar unit
    const unsigned xoff = (unsigned)((16 * hi * g.ldx + fcol) * 4), yoff = (unsigned)((16 * hi * g.ldy + ncol) * 4);
    auto load_raw = [&](float (&dst)[16], const float* base, int64_t ld, unsigned voff, int64_t r0) {
        if (r0 + BK <= r_end) {                                         // (wave-uniform) a full k-tile: no checks
            const char* rowp = reinterpret_cast<const char*>(base + r0 * ld);
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[e] = *reinterpret_cast<const float*>(rowp + (int64_t)e * ld * 4 + voff);
        } else {                                                        // the slice's last, partial k-tile (or past its end)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t r = r0 + 16 * hi + e;
                const bool ok = r < r_end;
                const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (ok ? r : r_end - 1) * ld) +
                                                                (voff - (unsigned)(16 * hi * ld * 4)));
                dst[e] = ok ? v : 0.f;
            }
        }
    };
    // GATHER: where this wave's 32 columns live.  A wave covers half a field's row (32 of its 64 dims: one 128-byte line per
    // reduction row and lane half, exactly the lines the forward's gather fetched), or the dense features, or nothing (columns
    // past F: a valid dummy source, the outputs are never read).  Every load stays unconditional.
    const int c0w = f0 + wave * 32;
    const bool w_field = GATHER && c0w < 64 * g.nf;
    const bool w_dense = GATHER && !w_field && g.dense_pad != nullptr && c0w < 64 * g.nf + 32;
    // One raw buffer resource per wave (its field's rows / the dense features / a dummy): the address of a load is then ONE 32-bit
    // VALU operation, id * 256 + column (a field is below 2^24 rows = 4 GB, as in the fused forward), instead of 64-bit pointer
    // arithmetic per lane and load.  The resource's range check does the masking: a missing id (-1) becomes offset 0xFFFFFF00 +
    // column >= num_records and the hardware returns 0 -- no clamp, no select.  The 32 ids of a k-tile sit in the lanes so that
    // DPP row_share:e hands every lane the id of ITS row e (lanes 0-31: rows 0-15 twice, lanes 32-63: rows 16-31 twice) -- one
    // VALU move per load, no LDS shuffle.  (First cut: 64-bit pointers + ds_bpermute + clamp + select: +53 us on the kernel.)
    const float* gptr = g.table;
    unsigned gpitch = 0;                                                // bytes per source row
    const int32_t* idrow = nullptr;
    if (GATHER) {
        if (w_field) {
            const int fld = c0w >> 6;
            gptr = g.table + g.row_base[fld] * 64 + (c0w & 32);
            gpitch = 256;
            idrow = g.ids_t + (int64_t)fld * g.R;
        } else if (w_dense) {
            gptr = g.dense_pad;
            gpitch = 128;
        }
    }
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gptr), 0, (int)0xFFFFFF00u, 0x00020000);
    const unsigned gcol = (unsigned)l31 * 4u;
    const int id_lane = (lane & 15) + 16 * hi;                          // the row of the k-tile whose id this lane keeps
    int idv = 0;                                                        // ids of the k-tile whose rows are fetched next
    int xmask = -1;                                                     // validity of xa[e] (partial k-tiles only)
    auto load_ids = [&](int64_t r0) -> int {
        if (!w_field) return 0;                                         // (wave-uniform)
        const int64_t r = r0 + id_lane < g.R ? r0 + id_lane : g.R - 1;
        return idrow[r];
    };
    auto load_gather = [&](float (&dst)[16], int ids_of_tile, int64_t r0) {
        const bool full = r0 + BK <= r_end;                             // (wave-uniform) all but a slice's last k-tile
        int vm = -1;
#define GATHER_ONE(E)                                                                                                   \
        {                                                                                                                \
            const int64_t r = r0 + 16 * hi + E;                                                                          \
            unsigned idx;                                                                                                \
            if (w_field) idx = (unsigned)__builtin_amdgcn_update_dpp(0, ids_of_tile, 0x150 + E, 0xf, 0xf, false);        \
            else idx = (unsigned)(r < r_end ? r : r_end - 1);                                                            \
            if (!full && !(r < r_end)) { idx = w_field ? 0x00FFFFFFu : idx; vm &= ~(1 << E); }                           \
            dst[E] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grsrc, (int)(idx * gpitch + gcol), 0, DR_NT_WGRAD_GATHER ? 2 : 0)); \
        }
        GATHER_ONE(0) GATHER_ONE(1) GATHER_ONE(2) GATHER_ONE(3) GATHER_ONE(4) GATHER_ONE(5) GATHER_ONE(6) GATHER_ONE(7)
        GATHER_ONE(8) GATHER_ONE(9) GATHER_ONE(10) GATHER_ONE(11) GATHER_ONE(12) GATHER_ONE(13) GATHER_ONE(14) GATHER_ONE(15)
#undef GATHER_ONE
        if (!(w_field || w_dense)) vm = 0;                              // columns past F: zeros (their outputs are never read)
        xmask = vm;
    };
    bf16x8 fa[2][3];
    bf16x8 fb[4][3];                                                    // group q uses buffer q & 3, read two groups ahead
    float cs = 0.f;
    auto stage_b = [&](int stage) {                                     // yb -> three planes of this lane's column
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 p0, p1, p2;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = yb[8 * s + j]; cs += v[j]; }
            if constexpr (H2) h2_split8v(v, h2_sy, p0, p1);
            else rs_split8v(v, p0, p1, p2);
            unsigned char* w = wrow + stage * STAGE + (((2 * hi + s) ^ sw) << 4);
            *reinterpret_cast<bf16x8*>(w) = p0;
            *reinterpret_cast<bf16x8*>(w + B_PLANE) = p1;
            if constexpr (!H2) *reinterpret_cast<bf16x8*>(w + 2 * B_PLANE) = p2;
        }
    };
    auto read_b = [&](int buf, int stage, int q) {
        const unsigned bb = b_addr[q >> 3] + stage * STAGE;
        const int nt = q & 7;
#define RS_READ3(NTI)                                                        \
        BF3_DS_READ_B128(fb[buf][0], bb, 0 * B_PLANE + NTI * 2048);          \
        BF3_DS_READ_B128(fb[buf][1], bb, 1 * B_PLANE + NTI * 2048);          \
        if constexpr (!H2) BF3_DS_READ_B128(fb[buf][2], bb, 2 * B_PLANE + NTI * 2048);
        switch (nt) {
            case 0: RS_READ3(0) break; case 1: RS_READ3(1) break; case 2: RS_READ3(2) break; case 3: RS_READ3(3) break;
            case 4: RS_READ3(4) break; case 5: RS_READ3(5) break; case 6: RS_READ3(6) break; default: RS_READ3(7) break;
        }
#undef RS_READ3
    };
    auto wait_b = [&](int buf, bool all) {
        if constexpr (H2) {
            if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]));
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]));
            return;
        }
        if (all) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]));
        else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]));
    };
    f32x16 acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[tt][k] = 0.f;
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};

    // ---- prologue: B(0) into stage 0, A(0) and B(1) into registers -------------------------------------------------------------
    load_raw(yb, g.Y, g.ldy, yoff, r_begin);
    stage_b(0);
    if (GATHER) {
        idv = load_ids(r_begin);
        load_gather(xa, idv, r_begin);
        idv = load_ids(r_begin + BK);
    } else {
        load_raw(xa, g.X, g.ldx, xoff, r_begin);
    }
    load_raw(yb, g.Y, g.ldy, yoff, r_begin + BK);                       // (all zeros when nk == 1)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read_b(0, 0, 0);
    read_b(1, 0, 1);

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        // A of this k-tile -> bf16 terms; B of the next k-tile -> the other stage (its readers passed the barrier at the end of
        // the previous step); the loads of the k-tile after that go into flight
        {
            float v[8];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (!GATHER || xmask == -1 || ((xmask >> (8 * s + j)) & 1)) ? xa[8 * s + j] : 0.f;
                if constexpr (H2) h2_split8v(v, h2_sx, fa[s][0], fa[s][1]);
                else rs_split8v(v, fa[s][0], fa[s][1], fa[s][2]);
            }
        }
        if (kt + 1 < nk) stage_b(stage ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (GATHER) {
            // (the 16 id shuffles are LDS operations: they and the loads they address stay between these two scheduling
            // barriers, i.e. in front of every fragment read the counted lgkmcnt waits of the MFMA loop reckon with)
            load_gather(xa, idv, r_begin + (int64_t)(kt + 1) * BK);
            idv = load_ids(r_begin + (int64_t)(kt + 2) * BK);
        } else {
            load_raw(xa, g.X, g.ldx, xoff, r_begin + (int64_t)(kt + 1) * BK);
        }
        load_raw(yb, g.Y, g.ldy, yoff, r_begin + (int64_t)(kt + 2) * BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < 14) {
                wait_b(q & 3, false);
                read_b((q + 2) & 3, stage, q + 2);
            } else if (q == 14) {
                // groups 14 and 15 are in registers, this wave is done reading this stage and its ds_writes of the next one
                // have retired (lgkmcnt(0) covers both): publish
                wait_b(2, true);
                asm volatile("s_barrier" ::: "memory");
                if (kt + 1 < nk) read_b(0, stage ^ 1, 0);
            } else {
                if (kt + 1 < nk) read_b(1, stage ^ 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (H2) {
                acc[q & 7] = h2_mfma(fa[q >> 3][0], fb[q & 3][1], acc[q & 7]);
                acc[q & 7] = h2_mfma(fa[q >> 3][1], fb[q & 3][0], acc[q & 7]);
                acc[q & 7] = h2_mfma(fa[q >> 3][0], fb[q & 3][0], acc[q & 7]);
            } else {
#pragma unroll
            for (int term = 0; term < 6; ++term)
                acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q >> 3][PA[term]], fb[q & 3][PB[term]], acc[q & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: partial tile (padded workspace: unconditional stores); C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 hi
    float* out = g.partial + ((int64_t)slice * Fp + f0 + wave * 32 + 4 * hi) * Np + n0 + l31;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) out[(int64_t)((reg & 3) + 8 * (reg >> 2)) * Np + nt * 32] = acc[nt][reg];
    if (want_cs) {
        cs += __shfl_xor(cs, 32, 64);
        if (hi == 0) g.colsum[(int64_t)slice * Np + n0 + wave * 32 + l31] = cs;
    }
}

// dst[f][n] += scale * sum_s partial[s][f][n]  (fixed order);  dstb[n] += scale * sum_s colsum[s][n]
__global__ __launch_bounds__(256) void bf3_tn_rs_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ colsum,
                                                               int32_t split, int32_t F, int32_t N, int32_t Fp, int32_t Np,
                                                               float scale, float* __restrict__ dst, int64_t ld,
                                                               float* __restrict__ dstb, const uint32_t* __restrict__ x_amax = nullptr,
                                                               const uint32_t* __restrict__ x2_amax = nullptr,
                                                               const uint32_t* __restrict__ y_amax = nullptr) {
    const int64_t total = (int64_t)F * N, stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t ps = (int64_t)Fp * Np;
    float wscale = scale;                                               // f16x2 partials carry s_x s_y (powers of two: exact)
    if (x_amax != nullptr) {
        float sx, ix, sy, iy;
        h2_scale_of(max(x_amax[0], x2_amax != nullptr ? x2_amax[0] : 0u), sx, ix);
        h2_scale_of(y_amax[0], sy, iy);
        wscale = scale * (ix * iy);
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t f = i / N;
        const int n = (int)(i - f * N);
        const float* p = partial + f * Np + n;
        float acc = 0.f;
        int s = 0;
        for (; s + 4 <= split; s += 4) {             // four loads in flight, summed in slice order
            const float v0 = p[s * ps], v1 = p[(s + 1) * ps], v2 = p[(s + 2) * ps], v3 = p[(s + 3) * ps];
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; s < split; ++s) acc += p[s * ps];
        dst[f * ld + n] = fmaf(wscale, acc, dst[f * ld + n]);
    }
    if (blockIdx.x == 0 && colsum != nullptr && dstb != nullptr)
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            float acc = 0.f;
            for (int s = 0; s < split; ++s) acc += colsum[(int64_t)s * Np + n];
            dstb[n] = fmaf(scale, acc, dstb[n]);
        }
}

// =====================================================================================================================
// TN split-K:  partial[s][f][n] = sum_{r in slice s} X[r][f] Y[r][n]
// =====================================================================================================================
struct TnArgs {
    const __bf16* X; int64_t x_ps, x_ld;
    const __bf16* Y; int64_t y_ps, y_ld;
    int64_t R; int32_t F; int32_t N;
    int64_t per; int32_t split;                  // reduction rows per slice (multiple of 32), number of slices
    float* partial;                              // [split][F][N]
};

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_ptr_t)const_cast<unsigned char*>(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_ptr_t)const_cast<unsigned char*>(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int WM, int WN>
__global__ __launch_bounds__(NTHREADS, 2) void bf3_gemm_tn_kernel(TnArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;                           // BM: columns of X (output rows f), BN: columns of Y
    constexpr int ROW_A = BM * 2, ROW_B = BN * 2;                        // bytes per reduction row of a plane image
    constexpr int A_PLANE = BK * ROW_A, B_PLANE = BK * ROW_B;
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    constexpr int NKB = STAGE / 1024, KB_A = 3 * A_PLANE / 1024, PER_WAVE = NKB / NWAVES;
    static_assert(WM * WN == NWAVES && NKB % NWAVES == 0, "tile / wave layout");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5, g16 = (lane >> 4) & 1, s = lane & 15;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_f = (g.F + BM - 1) / BM;
    const int per_slice = tiles_f * tiles_n;
    const int ntiles = per_slice * g.split;

    // transposing fragment reads: this lane supplies the address of row (8 hi + 4 q + (s >> 2)) of the k-step, 4 elements from
    // column 16 g16 + 4 (s & 3) of the 32-column MFMA tile; it receives column 16 g16 + s, rows 8 hi + 4 q + 0..3
    const int swz = (s >> 2) << 2;
    const int a_off = (8 * hi + (s >> 2)) * ROW_A + ((((wm * 8) ^ swz) + 2 * g16 + ((s & 3) >> 1)) << 4) + ((s & 1) << 3);
    const int b_off = 3 * A_PLANE + (8 * hi + (s >> 2)) * ROW_B + ((((wn * 8) ^ swz) + 2 * g16 + ((s & 3) >> 1)) << 4) + ((s & 1) << 3);

    const __bf16* src[PER_WAVE];
    int64_t src_step[2];
    src_step[0] = (int64_t)BK * g.x_ld;
    src_step[1] = (int64_t)BK * g.y_ld;
    int f0 = 0, n0 = 0, slice = 0, nk = 0;
    auto setup = [&](int tile) {
        const int lid = xcd_remap(tile, ntiles);                        // consecutive logical ids share a reduction slice
        slice = lid / per_slice;
        const int t = lid % per_slice;
        f0 = (t / tiles_n) * BM;
        n0 = (t % tiles_n) * BN;
        const int64_t r0 = (int64_t)slice * g.per;
        int64_t r1 = r0 + g.per;
        const int64_t rpad = (g.R + BK - 1) / BK * BK;
        if (r1 > rpad) r1 = rpad;
        nk = (int)((r1 - r0) / BK);
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int j = wave + NWAVES * i;
            const bool is_a = j < KB_A;
            const int jj = is_a ? j : j - KB_A;
            const int kbs = is_a ? A_PLANE / 1024 : B_PLANE / 1024;
            const int rowb = is_a ? ROW_A : ROW_B;
            const int plane = jj / kbs, kb = jj % kbs;
            const int o = kb * 1024 + lane * 16;
            const int row = o / rowb;
            const int pc = (o % rowb) >> 4;
            const int c = pc ^ ((row & 3) << 2);
            int64_t col = (is_a ? f0 : n0) + c * 8;
            const int64_t ld = is_a ? g.x_ld : g.y_ld;
            col = col <= ld - 8 ? col : ld - 8;                          // columns past the pitch only feed unstored outputs
            src[i] = is_a ? g.X + plane * g.x_ps + (r0 + row) * g.x_ld + col : g.Y + plane * g.y_ps + (r0 + row) * g.y_ld + col;
        }
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int j = wave + NWAVES * i;
            unsigned char* dst = smem + stage * STAGE + j * 1024;
            lds_dma16(src[i], dst);
            src[i] += (j < KB_A) ? src_step[0] : src_step[1];
        }
    };

    f32x16 acc[2][2];
    // same schedule as the NT kernel: fragment reads up front, the next k-tile's LDS-DMA pieces between the first MFMA groups
    auto compute = [&](int stage, bool prefetch) {
        const int sa = stage * STAGE + a_off, sb = stage * STAGE + b_off;
        bf16x8 af[2][3][2], bf[2][3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned char* pa = &smem[(sa ^ (t * 64)) + p * A_PLANE + ks * 16 * ROW_A];
                    const unsigned char* pb = &smem[(sb ^ (t * 64)) + p * B_PLANE + ks * 16 * ROW_B];
                    af[ks][p][t] = tr_pair(pa, pa + 4 * ROW_A);
                    bf[ks][p][t] = tr_pair(pb, pb + 4 * ROW_B);
                }
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
        unsigned char* const dma_dst = smem + (stage ^ 1) * STAGE + wave * 1024;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int term = 0; term < 6; ++term) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][PA[term]][a], bf[ks][PB[term]][b], acc[a][b], 0, 0, 0);
                const int piece = ks * 6 + term;
                if (piece < PER_WAVE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (prefetch) {
                        lds_dma16(src[piece], dma_dst + NWAVES * piece * 1024);
                        src[piece] += (wave + NWAVES * piece < KB_A) ? src_step[0] : src_step[1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int buf = 0;
    setup(tile);
    if (nk > 0) issue(buf);
    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[a][b][k] = 0.f;
        const int tf0 = f0, tn0 = n0, tslice = slice;
        const int tnk = nk;
        for (int kt = 0; kt < tnk; ++kt) {
            __syncthreads();
            compute(buf, kt + 1 < tnk);
            buf ^= 1;
        }
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            setup(next);
            if (tnk == 1) __syncthreads();        // a one-k-tile tile: stage `buf` may still be read by a slower wave
            if (nk > 0) issue(buf);
        }
        __builtin_amdgcn_sched_barrier(0);
        float* out = g.partial + (int64_t)tslice * g.F * g.N;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = tn0 + wn * 64 + ni * 32 + l31;
                const int row_b = tf0 + wm * 64 + mi * 32 + 4 * hi;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = row_b + (reg & 3) + 8 * (reg >> 2);
                    if (col < g.N && row < g.F) out[(int64_t)row * g.N + col] = acc[mi][ni][reg];
                }
            }
        if (next >= ntiles) break;
        tile = next;
    }
}

// dst[f][n] += scale * sum_s partial[s][f][n]  (fixed order);  dstb[n] += scale * colsum[n]
__global__ __launch_bounds__(256) void bf3_splitk_reduce_kernel(const float* __restrict__ partial, int32_t split, int64_t F,
                                                                int32_t N, float scale, float* __restrict__ dst, int64_t ld,
                                                                const float* __restrict__ colsum, float* __restrict__ dstb) {
    const int64_t total = F * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float acc = 0.f;
        int sidx = 0;
        for (; sidx + 4 <= split; sidx += 4) {       // four loads in flight, summed in slice order
            const float v0 = partial[(int64_t)sidx * total + i], v1 = partial[(int64_t)(sidx + 1) * total + i];
            const float v2 = partial[(int64_t)(sidx + 2) * total + i], v3 = partial[(int64_t)(sidx + 3) * total + i];
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; sidx < split; ++sidx) acc += partial[(int64_t)sidx * total + i];
        const int64_t f = i / N;
        const int n = (int)(i - f * N);
        dst[f * ld + n] = fmaf(scale, acc, dst[f * ld + n]);
    }
    if (blockIdx.x == 0 && colsum != nullptr && dstb != nullptr)
        for (int n = threadIdx.x; n < N; n += blockDim.x) dstb[n] = fmaf(scale, colsum[n], dstb[n]);
}

// fp32 [R][C] -> planes[p][r0 + r][c0 + c]   (transpose = 0)   or   planes[p][r0 + c][c0 + r]   (transpose = 1)
__global__ __launch_bounds__(256) void bf3_split_kernel(const float* __restrict__ src, int64_t ld_src, int64_t R, int32_t C,
                                                        __bf16* __restrict__ planes, int64_t ps, int64_t ldp, int64_t r0,
                                                        int64_t c0, int32_t transpose) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (!transpose) {
        const int cq = (C + 3) / 4;
        const int64_t total = R * cq;
        const bool vec = (C & 3) == 0 && (c0 & 3) == 0 && (ldp & 3) == 0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int64_t r = i / cq;
            const int c = (int)(i - r * cq) * 4;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c + j < C ? src[r * ld_src + c + j] : 0.f;
            const int64_t off = (r0 + r) * ldp + c0 + c;
            if (vec) {
                bf3::store4(planes, ps, off, v[0], v[1], v[2], v[3]);
            } else {
                bf3::bf16x4 p0, p1, p2;
                bf3::split4(v[0], v[1], v[2], v[3], p0, p1, p2);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < C) {
                        planes[off + j] = p0[j];
                        planes[ps + off + j] = p1[j];
                        planes[2 * ps + off + j] = p2[j];
                    }
            }
        }
    } else {
        // a thread takes 4 consecutive rows r of one column c: coalesced reads across c, one 8-byte store per plane
        const int64_t rq = (R + 3) / 4;
        const int64_t total = rq * C;
        const bool vec = (c0 & 3) == 0 && (ldp & 3) == 0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int64_t q = i / C;
            const int c = (int)(i - q * C);
            const int64_t r = q * 4;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = r + j < R ? src[(r + j) * ld_src + c] : 0.f;
            const int64_t off = (r0 + c) * ldp + c0 + r;
            bf3::bf16x4 p0, p1, p2;
            bf3::split4(v[0], v[1], v[2], v[3], p0, p1, p2);
            if (vec && r + 4 <= R) {
                *reinterpret_cast<bf3::bf16x4*>(planes + off) = p0;
                *reinterpret_cast<bf3::bf16x4*>(planes + ps + off) = p1;
                *reinterpret_cast<bf3::bf16x4*>(planes + 2 * ps + off) = p2;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r + j < R) {
                        planes[off + j] = p0[j];
                        planes[ps + off + j] = p1[j];
                        planes[2 * ps + off + j] = p2[j];
                    }
            }
        }
    }
}

// ---- f16x2 mode: amax records and the weight split ---------------------------------------------------------------------------
// amax[0] = max(amax[0], max |src|) as float bits (non-negative floats order like their bit patterns; a NaN lands above inf)
__global__ __launch_bounds__(256) void h2_amax_kernel(const float* __restrict__ src, int64_t ld, int64_t R, int32_t C,
                                                      uint32_t* __restrict__ amax) {
    uint32_t m = 0u;
    const bool al16 = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    auto take4 = [&](const float4& v) {
        m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
    };
    if (ld == C || R == 1) {                                            // contiguous: one flat stream, four 16-byte loads in flight
        const int64_t n = R * (int64_t)C, nv = al16 ? (n >> 2) : 0;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < nv; i += 4 * stride) {
            const float4 v0 = s4[i], v1 = s4[i + stride], v2 = s4[i + 2 * stride], v3 = s4[i + 3 * stride];
            take4(v0); take4(v1); take4(v2); take4(v3);
        }
        for (; i < nv; i += stride) take4(s4[i]);
        for (int64_t j = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) m = max(m, __float_as_uint(fabsf(src[j])));
    } else {                                                            // padded rows: a block per row (and stride)
        const bool vec = al16 && (ld & 3) == 0;
        const int cv = vec ? (C >> 2) : 0;
        for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
            const float* row = src + r * ld;
            for (int c = threadIdx.x; c < cv; c += blockDim.x) take4(reinterpret_cast<const float4*>(row)[c]);
            for (int c = (cv << 2) + threadIdx.x; c < C; c += blockDim.x) m = max(m, __float_as_uint(fabsf(row[c])));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, m);
    }
}

// fp32 [R][C] * s -> two fp16 planes; planes[p][r0 + r][c0 + c] (transpose = 0) or planes[p][r0 + c][c0 + r] (transpose = 1); s from the record
__global__ __launch_bounds__(256) void h2_split_kernel(const float* __restrict__ src, int64_t ld_src, int64_t R, int32_t C,
                                                       _Float16* __restrict__ planes, int64_t ps, int64_t ldp, int64_t r0,
                                                       int64_t c0, int32_t transpose, const uint32_t* __restrict__ amax) {
    float sc, inv;
    h2_scale_of(amax[0], sc, inv);
    h2_mode_on();
    const int64_t total = R * C, stride = (int64_t)gridDim.x * blockDim.x;
    if (!transpose && (C & 3) == 0 && (ldp & 3) == 0 && (c0 & 3) == 0 && (ld_src & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(planes) & 7) == 0 && (ps & 3) == 0 && total < (int64_t)0x7fffffff) {
        // four columns per thread: one 16-byte load, one 8-byte store per plane (the top-K scan splits the whole corpus through here)
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const unsigned cq = (unsigned)C >> 2, tq = (unsigned)(total >> 2);
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < tq; i += (unsigned)stride) {
            const unsigned r = i / cq, c = (i - r * cq) << 2;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld_src + c) * sc;
            const f16x4 h = __builtin_convertvector(v, f16x4);
            const f16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), f16x4);
            const int64_t off = (r0 + r) * ldp + c0 + c;
            *reinterpret_cast<f16x4*>(planes + off) = h;
            *reinterpret_cast<f16x4*>(planes + ps + off) = l;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        // consecutive threads: consecutive DESTINATION elements (2-byte stores coalesce; the source is small and cached)
        int64_t r, c, off;
        if (!transpose) { r = i / C; c = i - r * C; off = (r0 + r) * ldp + c0 + c; }
        else { c = i / R; r = i - c * R; off = (r0 + c) * ldp + c0 + r; }
        const float v = src[r * ld_src + c] * sc;
        const _Float16 h = (_Float16)v;
        planes[off] = h;
        planes[ps + off] = (_Float16)(v - (float)h);
    }
}

// ---- a weight's record and BOTH of its plane images in two launches (dr_h2_refresh_weight, round 5) -----------------------------------
// dr_h2_amax + two dr_h2_split are four launches (a 4-byte memset, the atomicMax pass, two splits); in the sharded engine they sit on
// the serial chain between the wgrad and the next forward, where each small launch waits its turn among the exchange's HBM-bound
// kernels (rocprofv3: memset 47 us, amax 30, splits 16 + 6).  Here: per-block maxima with plain stores, then ONE kernel that reduces
// them in every block (256 values), stores the record from block 0 and writes both orientations.  Same record, same planes.
constexpr int H2_REFRESH_PARTS = 256;
__global__ __launch_bounds__(256) void h2_amax_parts_kernel(const float* __restrict__ src, int64_t ld, int64_t R, int32_t C,
                                                            uint32_t* __restrict__ parts) {
    uint32_t m = 0u;
    const int64_t total = R * (int64_t)C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / C;
        m = max(m, __float_as_uint(fabsf(src[r * ld + (i - r * C)])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}
__global__ __launch_bounds__(256) void h2_split_both_kernel(const float* __restrict__ src, int64_t ld, int64_t R, int32_t C,
                                                            _Float16* __restrict__ w, int64_t w_ps, int64_t w_ld,
                                                            _Float16* __restrict__ wt, int64_t wt_ps, int64_t wt_ld,
                                                            const uint32_t* __restrict__ parts, int32_t nparts, uint32_t* __restrict__ amax) {
    uint32_t m = (int)threadIdx.x < nparts ? parts[threadIdx.x] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) amax[0] = m;               // the record the GEMMs read (they run behind this kernel)
    float sc, inv;
    h2_scale_of(m, sc, inv);
    h2_mode_on();
    const int64_t total = R * (int64_t)C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += stride) {
        // consecutive threads: consecutive DESTINATION elements, first of the plain image, then of the transposed one
        int64_t r, c, off;
        _Float16* dst;
        int64_t ps;
        if (i < total) { r = i / C; c = i - r * C; off = r * w_ld + c; dst = w; ps = w_ps; }
        else { const int64_t j = i - total; c = j / R; r = j - c * R; off = c * wt_ld + r; dst = wt; ps = wt_ps; }
        const float v = src[r * ld + c] * sc;
        const _Float16 h = (_Float16)v;
        dst[off] = h;
        dst[ps + off] = (_Float16)(v - (float)h);
    }
}

// planes -> fp32 (tests, debugging): dst[r][c] = (p2 + p1) + p0
__global__ __launch_bounds__(256) void bf3_join_kernel(const __bf16* __restrict__ planes, int64_t ps, int64_t ldp, int64_t R,
                                                       int32_t C, float* __restrict__ dst, int64_t ld_dst) {
    const int64_t total = R * C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const int64_t off = r * ldp + c;
        dst[r * ld_dst + c] = bf3::join(planes[off], planes[ps + off], planes[2 * ps + off]);
    }
}

int rs_launch(const RsArgs& g, hipStream_t stream) {
    const int64_t tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    if (tiles > 0x7fffffff) return DR_EINVAL;
    const int grid = (int)(tiles < 256 ? tiles : 256);                  // persistent: one block per CU
    // staggered wave groups (see STG in bf3_gemm_rs_kernel): measured neutral to slightly negative (DeepFM step 1.628 vs 1.605 ms, DCN
    // 26.93 vs 26.67 ms, same box, alternating) -- the kernel sits at the clock the power budget allows, keeping the pipe busier
    // buys nothing -- so it is opt-in (DR_BF3_STAGGER=1)
    static const bool stg = [] { const char* e = getenv("DR_BF3_STAGGER"); return e != nullptr && e[0] == '1'; }();
    if (g.mask != nullptr && g.accumulate) return DR_EINVAL;            // (no caller needs both)
    // the kernels address B's planes through ONE buffer resource with 32-bit offsets: planes further apart than that would be clamped
    // away and read as zeros without any error -- refuse instead
    if ((int64_t)(g.a_amax != nullptr ? 2 : 3) * g.b_ps * 2 > (int64_t)0x7fffffff) return DR_EINVAL;
    if (g.a_amax != nullptr) {                                          // f16x2 operand mode
        if (g.b_amax == nullptr) return DR_EINVAL;
        if (g.c_amax != nullptr && hipMemsetAsync(g.c_amax, 0, sizeof(uint32_t), stream) != hipSuccess) return DR_ELAUNCH;
        if (g.pack_pos != nullptr) {
            hipLaunchKernelGGL((bf3_gemm_rs_kernel<5, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
            DR_CHECK_LAUNCH();
            return DR_OK;
        }
        if (g.sm_part_m != nullptr || g.sm_lse != nullptr) {
            // in-batch softmax.  The streaming kernel: with the resident activations of AR = 4 the epilogue's per-column corrections and
            // ids (24 registers) do not fit beside 64 resident operand registers (30 - 54 spilled, reloaded inside the step loop)
            if (g.sm_part_m != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<6, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
            else hipLaunchKernelGGL((bf3_gemm_rs_kernel<7, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
            DR_CHECK_LAUNCH();
            return DR_OK;
        }
        if (g.tau != nullptr) {
            // the scan's reduction is short (K = the embedding width): resident activations (AR, see the kernel) when K is 128 or 64
            // and the row tiles divide into the grid; DR_TOPK_RESIDENT=0: the streaming kernel (A/B, tools/exp/topk_xlane.sh)
            static const bool resident = [] { const char* e = getenv("DR_TOPK_RESIDENT"); return e == nullptr || e[0] != '0'; }();
            const int tiles_m = (int)((g.M + 255) / 256);
            if (resident && (g.K == 128 || g.K == 64) && tiles_m <= 256) {
                const int ar_grid = 256 / tiles_m * tiles_m;
                if (g.K == 128) hipLaunchKernelGGL((bf3_gemm_rs_kernel<4, 0, 1, 1, 1, 4>), dim3(ar_grid), dim3(512), 0, stream, g);
                else hipLaunchKernelGGL((bf3_gemm_rs_kernel<4, 0, 1, 1, 1, 2>), dim3(ar_grid), dim3(512), 0, stream, g);
                DR_CHECK_LAUNCH();
                return DR_OK;
            }
            hipLaunchKernelGGL((bf3_gemm_rs_kernel<4, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
            DR_CHECK_LAUNCH();
            return DR_OK;
        }
        if (g.x0 != nullptr) {
            hipLaunchKernelGGL((bf3_gemm_rs_kernel<2, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
            DR_CHECK_LAUNCH();
            return DR_OK;
        }
        {   // round 6: the 16-wave form (h2_occ.hip) for the plain epilogues; DR_H2_OCC=0 keeps the 8-wave kernel (A/B)
            static const bool occ = [] { const char* e = getenv("DR_H2_OCC"); return e == nullptr || e[0] != '0'; }();
            if (occ) {
                const int rc = occ_nt_launch(g, stream);
                if (rc != 1) return rc;
            }
        }
#ifdef DR_BF3_ABLATE
        {   // tools/exp/h2_ablate.py: DR_BF3_RS_DBG = 32 no fragment reads, 64 no split, 96 both, 2 no MFMA; DR_BF3_RS64=1: 4 waves x 64 rows
            static const int dbg = [] { const char* e = getenv("DR_BF3_RS_DBG"); return e ? atoi(e) : 0; }();
            static const bool ms2 = [] { const char* e = getenv("DR_BF3_RS64"); return e != nullptr && e[0] == '1'; }();
            if (g.mask == nullptr && !g.accumulate) {
#define RS_ABL2(D)                                                                                                    \
                if (dbg == D) {                                                                                       \
                    if (ms2) hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, D, 2, 1, 1>), dim3(grid), dim3(256), 0, stream, g); \
                    else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, D, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);     \
                    DR_CHECK_LAUNCH();                                                                                \
                    return DR_OK;                                                                                     \
                }
                RS_ABL2(0) RS_ABL2(32) RS_ABL2(64) RS_ABL2(96) RS_ABL2(2) RS_ABL2(1)
                // what the no-MFMA time (131 us of the plain forward's 223) is made of -- combinations for the next GPU call:
                // 34 = no MFMA + no fragment reads, 3 = no MFMA + no weight DMA, 10 = no MFMA + A from cache, 98 = no MFMA + no reads + no split,
                // 35 = no MFMA + no reads + no DMA, 43 = that + A from cache (what is left: the loop, the barrier, the epilogue stores)
                RS_ABL2(34) RS_ABL2(3) RS_ABL2(10) RS_ABL2(98) RS_ABL2(35) RS_ABL2(43)
                // round 5: WITH the MFMAs -- 8 = A from cache (its loads re-read k-tile 0), 4 = B always k-tile 0, 12 = both, 9 = no weight DMA + A from cache, 16 = staggered wave groups
                RS_ABL2(8) RS_ABL2(4) RS_ABL2(12) RS_ABL2(9) RS_ABL2(16)
                // 512 = no per-step barrier; 544 = that + no fragment reads; 521 = that + A from cache + no weight DMA
                RS_ABL2(512) RS_ABL2(544) RS_ABL2(521)
#undef RS_ABL2
            }
        }
#endif
        if (g.mask != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<1, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
        else if (g.accumulate) hipLaunchKernelGGL((bf3_gemm_rs_kernel<3, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, 0, 1, 1, 1>), dim3(grid), dim3(512), 0, stream, g);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    if (g.tau != nullptr) {
        hipLaunchKernelGGL((bf3_gemm_rs_kernel<4>), dim3(grid), dim3(512), 0, stream, g);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    if (g.pack_pos != nullptr) {
        hipLaunchKernelGGL((bf3_gemm_rs_kernel<5>), dim3(grid), dim3(512), 0, stream, g);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
#ifdef DR_BF3_ABLATE
    {   // ablations of the plain epilogue (tools/exp/rs64_bench.py): DR_BF3_RS_DBG = 32 no fragment reads, 64 no split, 96 both, 2 no MFMA
        static const int dbg = [] { const char* e = getenv("DR_BF3_RS_DBG"); return e ? atoi(e) : 0; }();
        static const bool ms2 = [] { const char* e = getenv("DR_BF3_RS64"); return e != nullptr && e[0] == '1'; }();
        if (dbg != 0 && g.x0 == nullptr && g.mask == nullptr && !g.accumulate && g.tau == nullptr) {
#define RS_ABL(D)                                                                                               \
            if (dbg == D) {                                                                                     \
                if (ms2) hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, D, 2>), dim3(grid), dim3(256), 0, stream, g); \
                else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, D, 1>), dim3(grid), dim3(512), 0, stream, g);     \
                DR_CHECK_LAUNCH();                                                                              \
                return DR_OK;                                                                                   \
            }
            RS_ABL(32) RS_ABL(64) RS_ABL(96) RS_ABL(2) RS_ABL(1) RS_ABL(128) RS_ABL(384) RS_ABL(448)
#undef RS_ABL
        }
    }
#endif
    // one 64-row wave per SIMD (MS = 2, see the kernel): DR_BF3_RS64=1
    static const bool rs64 = [] { const char* e = getenv("DR_BF3_RS64"); return e != nullptr && e[0] == '1'; }();
    static const bool rs64x2 = [] { const char* e = getenv("DR_BF3_RS64"); return e != nullptr && e[0] == '2'; }();
    if (rs64x2 && g.x0 == nullptr) {                                    // 8 waves x (64 rows x 128 columns): DR_BF3_RS64=2
        if (g.mask != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<1, 0, 2, 2>), dim3(grid), dim3(512), 0, stream, g);
        else if (g.accumulate) hipLaunchKernelGGL((bf3_gemm_rs_kernel<3, 0, 2, 2>), dim3(grid), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, 0, 2, 2>), dim3(grid), dim3(512), 0, stream, g);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    if (rs64 && g.x0 == nullptr) {
        if (g.mask != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<1, 0, 2>), dim3(grid), dim3(256), 0, stream, g);
        else if (g.accumulate) hipLaunchKernelGGL((bf3_gemm_rs_kernel<3, 0, 2>), dim3(grid), dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, 0, 2>), dim3(grid), dim3(256), 0, stream, g);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    if (stg) {
        if (g.x0 != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<2, 16>), dim3(grid), dim3(512), 0, stream, g);
        else if (g.mask != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<1, 16>), dim3(grid), dim3(512), 0, stream, g);
        else if (g.accumulate) hipLaunchKernelGGL((bf3_gemm_rs_kernel<3, 16>), dim3(grid), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0, 16>), dim3(grid), dim3(512), 0, stream, g);
    } else {
        if (g.x0 != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<2>), dim3(grid), dim3(512), 0, stream, g);
        else if (g.mask != nullptr) hipLaunchKernelGGL((bf3_gemm_rs_kernel<1>), dim3(grid), dim3(512), 0, stream, g);
        else if (g.accumulate) hipLaunchKernelGGL((bf3_gemm_rs_kernel<3>), dim3(grid), dim3(512), 0, stream, g);
        else hipLaunchKernelGGL((bf3_gemm_rs_kernel<0>), dim3(grid), dim3(512), 0, stream, g);
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}

bool planes_ok(const void* p, int64_t ps, int64_t ld) {
    return p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld > 0 && (ld & 7) == 0 && (ps & 7) == 0;
}

int tn_split_for(int64_t R, int32_t F, int32_t N, int bm, int bn) {
    const int64_t tiles = (int64_t)((F + bm - 1) / bm) * ((N + bn - 1) / bn);
    int64_t max_split = (R + 16 * BK - 1) / (16 * BK);                  // at least 16 k-tiles per slice
    if (max_split < 1) max_split = 1;
    if (max_split > 128) max_split = 128;
    int64_t sp = 256 / tiles;                                           // one block per CU: fill the chip once
    if (sp < 1) sp = 1;
    if (sp > max_split) sp = max_split;
    return (int)sp;
}

}  // namespace

extern "C" int dr_bf3_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride,
                            int64_t ld_planes, int64_t row_offset, int64_t col_offset, int32_t transpose,
                            dr_stream_t stream) {
    if (R < 0 || C < 0 || ld_src < C || row_offset < 0 || col_offset < 0 || ld_planes <= 0 || plane_stride <= 0) return DR_EINVAL;
    if (R == 0 || C == 0) return DR_OK;
    if (!src || !planes) return DR_EINVAL;
    if (!transpose && col_offset + C > ld_planes) return DR_EINVAL;
    if (transpose && col_offset + R > ld_planes) return DR_EINVAL;
    hipLaunchKernelGGL(bf3_split_kernel, dim3(dr_grid_for(R * ((C + 3) / 4), 256)), dim3(256), 0, dr_s(stream), src, ld_src, R, C,
                       static_cast<__bf16*>(planes), plane_stride, ld_planes, row_offset, col_offset, transpose);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_bf3_join(const void* planes, int64_t plane_stride, int64_t ld_planes, int64_t R, int32_t C, float* dst,
                           int64_t ld_dst, dr_stream_t stream) {
    if (R < 0 || C < 0 || ld_planes < C || ld_dst < C || plane_stride <= 0) return DR_EINVAL;
    if (R == 0 || C == 0) return DR_OK;
    if (!planes || !dst) return DR_EINVAL;
    hipLaunchKernelGGL(bf3_join_kernel, dim3(dr_grid_for(R * C, 256)), dim3(256), 0, dr_s(stream),
                       static_cast<const __bf16*>(planes), plane_stride, ld_planes, R, C, dst, ld_dst);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// C[m][n] = act(sum_k A[m][k] B[n][k] + bias[n]) (* (mask[m][n] > 0)).  K is the PADDED reduction length (multiple of 32);
// columns [true K, K) of both operands' planes must be zero.
extern "C" int dr_bf3_gemm_nt(const void* a_planes, int64_t a_plane_stride, int64_t a_ld, const void* b_planes,
                              int64_t b_plane_stride, int64_t b_ld, int64_t M, int32_t N, int32_t K, const float* bias,
                              int32_t act, const float* mask, int64_t ld_mask, float* C, int64_t ldc, dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || (K % BK) != 0 || act < 0 || act > 1) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!planes_ok(a_planes, a_plane_stride, a_ld) || !planes_ok(b_planes, b_plane_stride, b_ld) || !C) return DR_EINVAL;
    if (a_ld < K || b_ld < K || ldc < N || (mask != nullptr && ld_mask < N)) return DR_EINVAL;
    NtArgs g{static_cast<const __bf16*>(a_planes), a_plane_stride, a_ld, static_cast<const __bf16*>(b_planes), b_plane_stride, b_ld,
             M, N, K, C, ldc, bias, act, mask, ld_mask};
    static const bool pipe = [] { const char* e = getenv("DR_BF3_PIPE"); return e == nullptr || e[0] != '0'; }();
    if (pipe) {
        // three-stage 128 x 128 x 32 pipeline; consecutive tiles share an A row-panel (same XCD, L2)
        const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
        if (tiles > 0x7fffffff) return DR_EINVAL;
        const int grid = (int)(tiles < 256 ? tiles : 256);              // persistent: one block per CU
        static const bool w16 = [] { const char* e = getenv("DR_BF3_WAVES"); return e == nullptr || atoi(e) != 8; }();
        if (w16) hipLaunchKernelGGL((bf3_gemm_nt_pipe_kernel<16>), dim3(grid), dim3(1024), 0, dr_s(stream), g);
        else hipLaunchKernelGGL((bf3_gemm_nt_pipe_kernel<8>), dim3(grid), dim3(512), 0, dr_s(stream), g);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    // DR_BF3_PIPE=0: the two-stage 128 x 256 kernel (A/B reference)
    const int64_t tiles = ((M + 127) / 128) * ((N + 255) / 256);
    if (tiles > 0x7fffffff) return DR_EINVAL;
    const int grid = (int)(tiles < 256 ? tiles : 256);                  // persistent: one block per CU
    hipLaunchKernelGGL((bf3_gemm_nt_kernel<2, 4>), dim3(grid), dim3(NTHREADS), 0, dr_s(stream), g);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int64_t dr_bf3_gemm_tn_workspace_bytes(int64_t R, int32_t F, int32_t N) {
    if (R <= 0 || F <= 0 || N <= 0) return 0;
    return (int64_t)tn_split_for(R, F, N, 128, 256) * F * N * (int64_t)sizeof(float);
}

// dst[f][n] += scale * sum_r X[r][f] Y[r][n];  dstb[n] += scale * y_colsum[n] (both optional).  Planes must hold rows up to the
// next multiple of 32 past R, zero-filled.
extern "C" int dr_bf3_gemm_tn(const void* x_planes, int64_t x_plane_stride, int64_t x_ld, const void* y_planes,
                              int64_t y_plane_stride, int64_t y_ld, int64_t R, int32_t F, int32_t N, float scale, float* dst,
                              int64_t ld_dst, const float* y_colsum, float* dstb, void* workspace, int64_t workspace_bytes,
                              dr_stream_t stream) {
    if (R <= 0 || F <= 0 || N <= 0) return DR_EINVAL;
    if (!planes_ok(x_planes, x_plane_stride, x_ld) || !planes_ok(y_planes, y_plane_stride, y_ld) || !dst || !workspace) return DR_EINVAL;
    if (x_ld < F || y_ld < N || ld_dst < N) return DR_EINVAL;
    if (workspace_bytes < dr_bf3_gemm_tn_workspace_bytes(R, F, N)) return DR_EINVAL;
    TnArgs g{static_cast<const __bf16*>(x_planes), x_plane_stride, x_ld, static_cast<const __bf16*>(y_planes), y_plane_stride, y_ld,
             R, F, N, 0, 0, static_cast<float*>(workspace)};
    int split = tn_split_for(R, F, N, 128, 256);
    const int64_t rpad = (R + BK - 1) / BK * BK;
    g.per = ((rpad + split - 1) / split + BK - 1) / BK * BK;
    g.split = (int32_t)((rpad + g.per - 1) / g.per);                    // every launched slice is non-empty
    const int64_t tiles = (int64_t)((F + 127) / 128) * ((N + 255) / 256) * g.split;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL((bf3_gemm_tn_kernel<2, 4>), dim3(grid), dim3(NTHREADS), 0, dr_s(stream), g);
    hipLaunchKernelGGL(bf3_splitk_reduce_kernel, dim3(dr_grid_for((int64_t)F * N, 256)), dim3(256), 0, dr_s(stream), g.partial,
                       g.split, (int64_t)F, N, scale, dst, ld_dst, y_colsum, dstb);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// C[m][n] (+)= act(sum_k A[m][k] B[n][k] + bias[n]) (zeroed where mask[m][n] <= 0).  A: fp32 [M, K] row-major (ld multiple of 4,
// 16-byte aligned base); B: planes [3][N][b_ld], b_ld >= roundup(K, 32), columns [K, roundup(K, 32)) zero.
extern "C" int dr_bf3_linear_nt(const float* A, int64_t lda, const void* b_planes, int64_t b_plane_stride, int64_t b_ld, int64_t M,
                                int32_t N, int32_t K, const float* bias, int32_t act, const float* mask, int64_t ld_mask,
                                int32_t accumulate, float* C, int64_t ldc, dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > 1) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!A || !C || !planes_ok(b_planes, b_plane_stride, b_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(A) & 15) != 0 || (lda & 3) != 0 || lda < K) return DR_EINVAL;
    if (b_ld < (K + BK - 1) / BK * BK || ldc < N || (mask != nullptr && ld_mask < N)) return DR_EINVAL;
    RsArgs g{A, lda, static_cast<const __bf16*>(b_planes), b_plane_stride, b_ld, M, N, K, C, ldc, bias, act, mask, ld_mask, accumulate,
             nullptr, nullptr, 0, 0.f, nullptr};
    return rs_launch(g, dr_s(stream));
}

// ---- f16x2 operand mode (see h2_split8): amax records, the weight split, forward / dgrad on fp32 activations -----------------------
// amax[0] = max(reset ? 0 : amax[0], max |src[r][c]|) as float bits.  The record of a GEMM operand must be >= its true largest
// magnitude when the GEMM runs (a producer may keep a running maximum instead of an exact one).
extern "C" int dr_h2_amax(const float* src, int64_t ld, int64_t R, int32_t C, uint32_t* amax, int32_t reset, dr_stream_t stream) {
    if (R < 0 || C < 0 || ld < C || !amax) return DR_EINVAL;
    if (reset && hipMemsetAsync(amax, 0, sizeof(uint32_t), dr_s(stream)) != hipSuccess) return DR_ELAUNCH;
    if (R == 0 || C == 0) return DR_OK;
    if (!src) return DR_EINVAL;
    const int grid = (ld == C || R == 1) ? dr_grid_for(R * ((C + 3) / 4), 256 * 4) : (int)(R < 4096 ? R : 4096);
    hipLaunchKernelGGL(h2_amax_kernel, dim3(grid), dim3(256), 0, dr_s(stream), src, ld, R, C, amax);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// Two fp16 planes of src * s(amax) -- the layout of dr_bf3_split with two planes instead of three.  The GEMMs that read the planes
// are handed the same record (unchanged since the split).
extern "C" int dr_h2_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride,
                           int64_t ld_planes, int64_t row_offset, int64_t col_offset, int32_t transpose, const uint32_t* amax,
                           dr_stream_t stream) {
    if (R < 0 || C < 0 || ld_src < C || row_offset < 0 || col_offset < 0 || ld_planes <= 0 || plane_stride <= 0 || !amax) return DR_EINVAL;
    if (R == 0 || C == 0) return DR_OK;
    if (!src || !planes) return DR_EINVAL;
    if (!transpose && col_offset + C > ld_planes) return DR_EINVAL;
    if (transpose && col_offset + R > ld_planes) return DR_EINVAL;
    hipLaunchKernelGGL(h2_split_kernel, dim3(dr_grid_for(R * (int64_t)C, 256)), dim3(256), 0, dr_s(stream), src, ld_src, R, C,
                       static_cast<_Float16*>(planes), plane_stride, ld_planes, row_offset, col_offset, transpose, amax);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// A weight W [K, N] (row stride ldw): its amax record, its planes [2][K][w_ld] and the planes of its transpose [2][N][wt_ld] -- what
// dr_h2_amax(reset) + dr_h2_split + dr_h2_split(transpose) produce, in two launches without a memset or an atomic.  parts: 256 uint32 of scratch.
extern "C" int dr_h2_refresh_weight(const float* W, int64_t ldw, int64_t K, int32_t N, void* w_planes, int64_t w_ps, int64_t w_ld,
                                    void* wt_planes, int64_t wt_ps, int64_t wt_ld, uint32_t* amax, uint32_t* parts, dr_stream_t stream) {
    if (K <= 0 || N <= 0 || ldw < N || !W || !w_planes || !wt_planes || !amax || !parts) return DR_EINVAL;
    if (w_ld < N || wt_ld < K || w_ps <= 0 || wt_ps <= 0) return DR_EINVAL;
    const int64_t total = K * (int64_t)N;
    const int np = (int)(total < (int64_t)H2_REFRESH_PARTS * 256 ? (total + 255) / 256 : H2_REFRESH_PARTS);
    hipLaunchKernelGGL(h2_amax_parts_kernel, dim3(np), dim3(256), 0, dr_s(stream), W, ldw, K, N, parts);
    hipLaunchKernelGGL(h2_split_both_kernel, dim3(dr_grid_for(2 * total, 256, 2048)), dim3(256), 0, dr_s(stream), W, ldw, K, N,
                       static_cast<_Float16*>(w_planes), w_ps, w_ld, static_cast<_Float16*>(wt_planes), wt_ps, wt_ld, parts, np, amax);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// dr_bf3_linear_nt in the f16x2 mode: b_planes = two fp16 planes (dr_h2_split with b_amax), A fp32 with the record a_amax.
// c_amax (may be NULL): reset, then raised to max |value stored into C| -- the record the next GEMM wants for C as ITS operand.
extern "C" int dr_h2_linear_nt(const float* A, int64_t lda, const uint32_t* a_amax, const void* b_planes, int64_t b_plane_stride,
                               int64_t b_ld, const uint32_t* b_amax, int64_t M, int32_t N, int32_t K, const float* bias, int32_t act,
                               const float* mask, int64_t ld_mask, int32_t accumulate, float* C, int64_t ldc, uint32_t* c_amax,
                               dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || act < 0 || act > 1 || !a_amax || !b_amax) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!A || !C || !planes_ok(b_planes, b_plane_stride, b_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(A) & 15) != 0 || (lda & 3) != 0 || lda < K) return DR_EINVAL;
    if (b_ld < (K + BK - 1) / BK * BK || ldc < N || (mask != nullptr && ld_mask < N)) return DR_EINVAL;
    RsArgs g{A, lda, static_cast<const __bf16*>(b_planes), b_plane_stride, b_ld, M, N, K, C, ldc, bias, act, mask, ld_mask, accumulate,
             nullptr, nullptr, 0, 0.f, nullptr};
    g.a_amax = a_amax;
    g.b_amax = b_amax;
    g.c_amax = c_amax;
    return rs_launch(g, dr_s(stream));
}

// dr_bf3_cross_fwd in the f16x2 mode (DCN cross layer, keras/models/ranking/dcn.py:81-88): x_amax the record of x (the GEMM's
// activation operand), wt_planes two fp16 planes of W^T with w_amax; out_amax (may be NULL) receives the record of `out`, which is
// the next cross layer's x.
extern "C" int dr_h2_cross_fwd(const float* x0, const float* x, int64_t ld, const uint32_t* x_amax, const void* wt_planes,
                               int64_t plane_stride, int64_t ld_planes, const uint32_t* w_amax, const float* b, float diag_scale,
                               int64_t M, int32_t Dm, float* out, float* prod_out, uint32_t* out_amax, dr_stream_t stream) {
    if (M < 0 || Dm <= 0 || diag_scale < 0.f || !x_amax || !w_amax) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x0 || !x || !out || !planes_ok(wt_planes, plane_stride, ld_planes)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (ld & 3) != 0 || ld < Dm || ld_planes < (Dm + BK - 1) / BK * BK) return DR_EINVAL;
    RsArgs g{x, ld, static_cast<const __bf16*>(wt_planes), plane_stride, ld_planes, M, Dm, Dm, out, ld, b, 0, nullptr, 0, 0,
             x0, x, ld, diag_scale, prod_out};
    g.a_amax = x_amax;
    g.b_amax = w_amax;
    g.c_amax = out_amax;
    return rs_launch(g, dr_s(stream));
}

// dr_bf3_scores_filter in the f16x2 mode (retrieval.hip's top-K scan)
int dr_h2_scores_filter(const float* a, int64_t lda, const uint32_t* a_amax, const void* b_planes, int64_t b_plane_stride, int64_t b_ld,
                        const uint32_t* b_amax, int64_t M, int32_t N, int32_t K, const float* tau, float* cand_s, int32_t* cand_c,
                        int32_t* cand_cnt, int64_t cand_cap, dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || cand_cap <= 0 || !a_amax || !b_amax) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!a || !tau || !cand_s || !cand_c || !cand_cnt || !planes_ok(b_planes, b_plane_stride, b_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a) & 15) != 0 || (lda & 3) != 0 || lda < K || b_ld < (K + BK - 1) / BK * BK) return DR_EINVAL;
    RsArgs g{a, lda, static_cast<const __bf16*>(b_planes), b_plane_stride, b_ld, M, N, K, nullptr, 0, nullptr, 0, nullptr, 0, 0,
             nullptr, nullptr, 0, 0.f, nullptr, tau, cand_s, cand_c, cand_cnt, cand_cap};
    g.a_amax = a_amax;
    g.b_amax = b_amax;
    return rs_launch(g, dr_s(stream));
}

namespace {
__global__ __launch_bounds__(256) void bf3_pack_bias_kernel(const float* dl, int64_t n, float* bias_sum) {
    dr_block_sum_axpy(dl, n, 1.f, bias_sum);
}
}  // namespace

// First-layer dgrad of a tower whose input is the concatenation of F 64-wide embeddings (+ dense features) fused with
// dr_emb_pack_grads: the gradient of slot (m, f) -- sum_k dy[m][k] W[64 f + d][k] + d_fm_logit[m] (sum_x[m][d] - x[m][64 f + d]) --
// goes straight to out_rows[pos[m, f], d], the all-to-all send layout; d_concat is never written (0.44 GB less written and 0.66 GB
// less read per 65 536 examples than dgrad + pack).  w_planes: W as planes [3][K_in rows][b_ld] (rows = input features, as for
// dr_bf3_linear_nt's dgrad use); N = number of input features (>= 64 F; the columns behind the embeddings are dropped).
// out_lin (may be NULL): out_lin[pos[m, f]] = d_fm_logit[m].  bias_sum (may be NULL): += sum_m d_fm_logit[m], fixed order.
// sum_x == NULL: no FM term (x / ld_x unused).  pos must be a permutation of the slots, as for dr_emb_pack_grads.
extern "C" int dr_bf3_linear_nt_pack(const float* dy, int64_t ld_dy, const void* w_planes, int64_t plane_stride, int64_t b_ld, int64_t M,
                                     int32_t N, int32_t K, const int64_t* pos, int32_t F, const float* d_fm_logit, const float* sum_x,
                                     const float* x, int64_t ld_x, float* out_rows, float* out_lin, float* bias_sum,
                                     dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || F <= 0 || (int64_t)F * 64 > N) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!dy || !pos || !d_fm_logit || !out_rows || !planes_ok(w_planes, plane_stride, b_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) != 0 || (ld_dy & 3) != 0 || ld_dy < K || b_ld < (K + BK - 1) / BK * BK) return DR_EINVAL;
    if (sum_x != nullptr && (x == nullptr || ld_x < (int64_t)F * 64)) return DR_EINVAL;
    RsArgs g{dy, ld_dy, static_cast<const __bf16*>(w_planes), plane_stride, b_ld, M, N, K, out_rows, 0, nullptr, 0, nullptr, 0, 0,
             nullptr, x, ld_x, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, pos, F, d_fm_logit, sum_x, out_lin};
    const int rc = rs_launch(g, dr_s(stream));
    if (rc != DR_OK) return rc;
    if (bias_sum != nullptr) {
        hipLaunchKernelGGL(bf3_pack_bias_kernel, dim3(1), dim3(256), 0, dr_s(stream), d_fm_logit, M, bias_sum);
        DR_CHECK_LAUNCH();
    }
    return DR_OK;
}

// dr_bf3_linear_nt_pack in the f16x2 operand mode: dy with its amax record, the weights as two fp16 planes with theirs (dr_h2_split).
// x / ld_x: the concatenated embeddings of the forward, rows 16-byte aligned (ld_x % 4 == 0); sum_x rows are 64 floats.
extern "C" int dr_h2_linear_nt_pack(const float* dy, int64_t ld_dy, const uint32_t* dy_amax, const void* w_planes, int64_t plane_stride,
                                    int64_t b_ld, const uint32_t* w_amax, int64_t M, int32_t N, int32_t K, const int64_t* pos, int32_t F,
                                    const float* d_fm_logit, const float* sum_x, const float* x, int64_t ld_x, float* out_rows,
                                    float* out_lin, float* bias_sum, dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || F <= 0 || (int64_t)F * 64 > N) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!dy || !dy_amax || !w_amax || !pos || !d_fm_logit || !out_rows || !planes_ok(w_planes, plane_stride, b_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) != 0 || (ld_dy & 3) != 0 || ld_dy < K || b_ld < (K + BK - 1) / BK * BK) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(out_rows) & 15) != 0) return DR_EINVAL;
    if (!sum_x || !x || ld_x < (int64_t)F * 64 || (ld_x & 3) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(sum_x) & 15) != 0)
        return DR_EINVAL;
    // the epilogue's domain (see there): interior row tiles only, destinations in 32 bits.  Outside it the caller runs
    // dr_h2_linear_nt + dr_emb_pack_grads.
    if ((M % 256) != 0 || M * (int64_t)F >= ((int64_t)1 << 25)) return DR_ESHAPE;
    RsArgs g{dy, ld_dy, static_cast<const __bf16*>(w_planes), plane_stride, b_ld, M, N, K, out_rows, 0, nullptr, 0, nullptr, 0, 0,
             nullptr, x, ld_x, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, pos, F, d_fm_logit, sum_x, out_lin, dy_amax, w_amax,
             nullptr};
    const int rc = rs_launch(g, dr_s(stream));
    if (rc != DR_OK) return rc;
    if (bias_sum != nullptr) {
        hipLaunchKernelGGL(bf3_pack_bias_kernel, dim3(1), dim3(256), 0, dr_s(stream), d_fm_logit, M, bias_sum);
        DR_CHECK_LAUNCH();
    }
    return DR_OK;
}

// internal (C++ linkage, used by retrieval.hip): the top-K scan's scores = a @ B^T on the register-split kernel -- a [M, K] fp32
// queries, B = a corpus chunk as planes [3][N rows][b_ld] -- filtered against tau into per-row candidate lists (the epilogue of
// dense.hip's dr_scores_nt_filter on this file's tile: 256 x 256, weights through the LDS ring, activations split in registers)
int dr_bf3_scores_filter(const float* a, int64_t lda, const void* b_planes, int64_t b_plane_stride, int64_t b_ld, int64_t M, int32_t N,
                         int32_t K, const float* tau, float* cand_s, int32_t* cand_c, int32_t* cand_cnt, int64_t cand_cap,
                         dr_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || cand_cap <= 0) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!a || !tau || !cand_s || !cand_c || !cand_cnt || !planes_ok(b_planes, b_plane_stride, b_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a) & 15) != 0 || (lda & 3) != 0 || lda < K || b_ld < (K + BK - 1) / BK * BK) return DR_EINVAL;
    RsArgs g{a, lda, static_cast<const __bf16*>(b_planes), b_plane_stride, b_ld, M, N, K, nullptr, 0, nullptr, 0, nullptr, 0, 0,
             nullptr, nullptr, 0, 0.f, nullptr, tau, cand_s, cand_c, cand_cnt, cand_cap};
    return rs_launch(g, dr_s(stream));
}

// internal (C++ linkage, used by dense.hip's dr_inbatch_softmax_*): the in-batch softmax's two score passes on the register-split
// f16x2 kernel.  q [B, D] fp32 with its record, c as two fp16 planes [2][B][c_ld] with its record.
int dr_h2_inbatch_lse(const float* q, int64_t ldq, const uint32_t* q_amax, const void* c_planes, int64_t c_ps, int64_t c_ld,
                      const uint32_t* c_amax, int64_t B, int32_t D, const float* cand_prob, const int64_t* cand_ids, float inv_t,
                      float* part_m, float* part_l, float* pos, dr_stream_t stream) {
    if (B <= 0 || B > 0x7fffffff || !q || !q_amax || !c_amax || !part_m || !part_l || !pos || !planes_ok(c_planes, c_ps, c_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(q) & 15) != 0 || (ldq & 3) != 0 || ldq < D) return DR_EINVAL;
    RsArgs g{};
    g.A = q; g.lda = ldq; g.B = static_cast<const __bf16*>(c_planes); g.b_ps = c_ps; g.b_ld = c_ld; g.M = B; g.N = (int32_t)B; g.K = D;
    g.a_amax = q_amax; g.b_amax = c_amax;
    g.sm_cand_prob = cand_prob; g.sm_cand_ids = cand_ids; g.sm_inv_t = inv_t; g.sm_part_m = part_m; g.sm_part_l = part_l; g.sm_pos = pos;
    return rs_launch(g, dr_s(stream));
}
int dr_h2_inbatch_smgrad(const float* q, int64_t ldq, const uint32_t* q_amax, const void* c_planes, int64_t c_ps, int64_t c_ld,
                         const uint32_t* c_amax, int64_t B, int32_t D, const float* cand_prob, const int64_t* cand_ids, float inv_t,
                         const float* row_lse, const float* sample_weight, float d_loss, float* G, int64_t ld_g, dr_stream_t stream) {
    if (B <= 0 || B > 0x7fffffff || !q || !q_amax || !c_amax || !row_lse || !G || ld_g < B || !planes_ok(c_planes, c_ps, c_ld)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(q) & 15) != 0 || (ldq & 3) != 0 || ldq < D) return DR_EINVAL;
    RsArgs g{};
    g.A = q; g.lda = ldq; g.B = static_cast<const __bf16*>(c_planes); g.b_ps = c_ps; g.b_ld = c_ld; g.M = B; g.N = (int32_t)B; g.K = D;
    g.C = G; g.ldc = ld_g; g.a_amax = q_amax; g.b_amax = c_amax;
    g.sm_cand_prob = cand_prob; g.sm_cand_ids = cand_ids; g.sm_inv_t = inv_t; g.sm_lse = row_lse; g.sm_w = sample_weight; g.sm_alpha = d_loss;
    return rs_launch(g, dr_s(stream));
}

// DCN cross layer forward on pre-split weights (same math as dr_cross_fwd, keras/models/ranking/dcn.py:81-88):
//   prod = x @ W + b + diag_scale * x ;  out = x0 * prod + x ;  prod_out (may be NULL) saves prod for the backward.
// wt_planes: W^T as planes [3][Dm][ld_planes] (dr_bf3_split(..., transpose = 1)).  x0, x, out, prod_out share `ld`.
extern "C" int dr_bf3_cross_fwd(const float* x0, const float* x, int64_t ld, const void* wt_planes, int64_t plane_stride,
                                int64_t ld_planes, const float* b, float diag_scale, int64_t M, int32_t Dm, float* out,
                                float* prod_out, dr_stream_t stream) {
    if (M < 0 || Dm <= 0 || diag_scale < 0.f) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x0 || !x || !out || !planes_ok(wt_planes, plane_stride, ld_planes)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (ld & 3) != 0 || ld < Dm || ld_planes < (Dm + BK - 1) / BK * BK) return DR_EINVAL;
    RsArgs g{x, ld, static_cast<const __bf16*>(wt_planes), plane_stride, ld_planes, M, Dm, Dm, out, ld, b, 0, nullptr, 0, 0,
             x0, x, ld, diag_scale, prod_out};
    return rs_launch(g, dr_s(stream));
}

namespace {
void tn_rs_plan(int64_t R, int32_t F, int32_t N, int& split, int64_t& per, int& Fp, int& Np) {
    const int tf = (F + 255) / 256, tn = (N + 255) / 256;
    Fp = tf * 256;
    Np = tn * 256;
    int64_t sp = 256 / ((int64_t)tf * tn);                              // about one block per CU
    const int64_t max_split = (R + 16 * BK - 1) / (16 * BK);            // at least 16 k-tiles per slice
    if (sp > max_split) sp = max_split;
    if (sp < 1) sp = 1;
    per = ((R + sp - 1) / sp + BK - 1) / BK * BK;
    split = (int)((R + per - 1) / per);                                 // every slice non-empty
}
}  // namespace

extern "C" int64_t dr_bf3_wgrad_workspace_bytes(int64_t R, int32_t F, int32_t N) {
    if (R <= 0 || F <= 0 || N <= 0) return 0;
    int split, Fp, Np;
    int64_t per;
    tn_rs_plan(R, F, N, split, per, Fp, Np);
    return ((int64_t)split * Fp * Np + (int64_t)split * Np) * (int64_t)sizeof(float);
}

// dstW[f][n] += scale * sum_r x[r][f] dy[r][n];  dstb[n] += scale * sum_r dy[r][n] (dstb may be NULL).  x [R, F], dy [R, N] fp32
// row-major; the bf16x3 product mode, deterministic (fixed-order reduce over the reduction slices).
static int wgrad_impl(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, int64_t R, int32_t F, int32_t N,
                      float scale, float* dstW, int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes,
                      dr_stream_t stream, const uint32_t* x_amax = nullptr, const uint32_t* dy_amax = nullptr) {
    if (R <= 0 || F <= 0 || N <= 0) return DR_EINVAL;
    if (!x || !dy || !dstW || !workspace || ld_x < F || ld_dy < N || ld_w < N) return DR_EINVAL;
    if (workspace_bytes < dr_bf3_wgrad_workspace_bytes(R, F, N)) return DR_EINVAL;
    int split, Fp, Np;
    int64_t per;
    tn_rs_plan(R, F, N, split, per, Fp, Np);
    float* partial = static_cast<float*>(workspace);
    float* colsum = partial + (int64_t)split * Fp * Np;
    TnRsArgs g{x, ld_x, dy, ld_dy, R, F, N, per, split, partial, dstb != nullptr ? colsum : nullptr, nullptr, nullptr, nullptr, 0, nullptr,
               x_amax, nullptr, dy_amax};
    const int grid = (Fp / 256) * (Np / 256) * split;
    if (x_amax != nullptr) hipLaunchKernelGGL((bf3_gemm_tn_rs_kernel<0, 1>), dim3(grid), dim3(512), 0, dr_s(stream), g);
    else hipLaunchKernelGGL((bf3_gemm_tn_rs_kernel<0, 0>), dim3(grid), dim3(512), 0, dr_s(stream), g);
    hipLaunchKernelGGL(bf3_tn_rs_reduce_kernel, dim3(dr_grid_for((int64_t)F * N, 256)), dim3(256), 0, dr_s(stream), partial,
                       dstb != nullptr ? colsum : nullptr, split, F, N, Fp, Np, scale, dstW, ld_w, dstb, x_amax,
                       static_cast<const uint32_t*>(nullptr), dy_amax);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_bf3_wgrad(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, int64_t R, int32_t F, int32_t N,
                            float scale, float* dstW, int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes,
                            dr_stream_t stream) {
    return wgrad_impl(x, ld_x, dy, ld_dy, R, F, N, scale, dstW, ld_w, dstb, workspace, workspace_bytes, stream);
}

// dr_bf3_wgrad in the f16x2 operand mode (h2_split8): x_amax / dy_amax are the operands' amax records (dr_h2_amax, or a producer's).
// Same workspace, same fixed-order reduce.
extern "C" int dr_h2_wgrad(const float* x, int64_t ld_x, const uint32_t* x_amax, const float* dy, int64_t ld_dy, const uint32_t* dy_amax,
                           int64_t R, int32_t F, int32_t N, float scale, float* dstW, int64_t ld_w, float* dstb, void* workspace,
                           int64_t workspace_bytes, dr_stream_t stream) {
    if (!x_amax || !dy_amax) return DR_EINVAL;
    return wgrad_impl(x, ld_x, dy, ld_dy, R, F, N, scale, dstW, ld_w, dstb, workspace, workspace_bytes, stream, x_amax, dy_amax);
}

// The same wgrad for the FIRST tower layer, whose x = concat(field embeddings, dense features) is never read from a buffer: the
// kernel gathers it from the tables (GATHER form of TnRsArgs; D = 64).  ids_t [nf][R] int32: the batch's bucket ids, field-major
// (dr_ids_transpose_i32), -1 = missing; dense_pad [R, 32] zero-padded dense features (NULL iff F == 64 nf).  With this the forward
// need not store `concat` at all (keras/models/ranking/deepfm.py:44-45: stack / concat become pure fiction).
static int wgrad_emb_impl(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                          const float* dense_pad, const float* dy, int64_t ld_dy, int32_t F, int32_t N, float scale, float* dstW,
                          int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes, int32_t parts, dr_stream_t stream,
                          const uint32_t* table_amax = nullptr, const uint32_t* dense_amax = nullptr, const uint32_t* dy_amax = nullptr) {
    if (R <= 0 || F <= 0 || N <= 0 || nf <= 0) return DR_EINVAL;
    if (table_amax != nullptr && (!dy_amax || (F > 64 * nf && !dense_amax))) return DR_EINVAL;
    if (D != 64 || F < 64 * nf || F > 64 * nf + 32) return DR_ESHAPE;
    if (!ids_t || !row_base || !table || !dy || !dstW || !workspace || ld_dy < N || ld_w < N) return DR_EINVAL;
    if (F > 64 * nf && !dense_pad) return DR_EINVAL;
    if (workspace_bytes < dr_bf3_wgrad_workspace_bytes(R, F, N)) return DR_EINVAL;
    int split, Fp, Np;
    int64_t per;
    tn_rs_plan(R, F, N, split, per, Fp, Np);
    float* partial = static_cast<float*>(workspace);
    float* colsum = partial + (int64_t)split * Fp * Np;
    TnRsArgs g{nullptr, 0, dy, ld_dy, R, F, N, per, split, partial, dstb != nullptr ? colsum : nullptr, ids_t, row_base, table, nf,
               F > 64 * nf ? dense_pad : nullptr, table_amax, F > 64 * nf ? dense_amax : nullptr, dy_amax};
    const int grid = (Fp / 256) * (Np / 256) * split;
    if (parts & 1) {
        if (table_amax != nullptr) hipLaunchKernelGGL((bf3_gemm_tn_rs_kernel<1, 1>), dim3(grid), dim3(512), 0, dr_s(stream), g);
        else hipLaunchKernelGGL((bf3_gemm_tn_rs_kernel<1, 0>), dim3(grid), dim3(512), 0, dr_s(stream), g);
    }
    if (parts & 2)
        hipLaunchKernelGGL(bf3_tn_rs_reduce_kernel, dim3(dr_grid_for((int64_t)F * N, 256)), dim3(256), 0, dr_s(stream), partial,
                           dstb != nullptr ? colsum : nullptr, split, F, N, Fp, Np, scale, dstW, ld_w, dstb, table_amax,
                           F > 64 * nf ? dense_amax : static_cast<const uint32_t*>(nullptr), dy_amax);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_bf3_wgrad_emb(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                                const float* dense_pad, const float* dy, int64_t ld_dy, int32_t F, int32_t N, float scale, float* dstW,
                                int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes, dr_stream_t stream) {
    return wgrad_emb_impl(ids_t, R, nf, row_base, table, D, dense_pad, dy, ld_dy, F, N, scale, dstW, ld_w, dstb, workspace, workspace_bytes,
                          3, stream);
}

// In two halves: parts = 1 the split-K GEMM into the workspace, parts = 2 the fixed-order reduce that applies it (dstW += scale * sum,
// dstb likewise), 3 = both.  Part 2 may run on another stream (the engine puts it in front of the weight-plane refresh, which lives
// there already); it must finish before anything reads dstW / dstb and before the next part 1 over the same workspace.
extern "C" int dr_bf3_wgrad_emb_parts(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                                      const float* dense_pad, const float* dy, int64_t ld_dy, int32_t F, int32_t N, float scale,
                                      float* dstW, int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes, int32_t parts,
                                      dr_stream_t stream) {
    if (parts < 1 || parts > 3) return DR_EINVAL;
    return wgrad_emb_impl(ids_t, R, nf, row_base, table, D, dense_pad, dy, ld_dy, F, N, scale, dstW, ld_w, dstb, workspace, workspace_bytes,
                          parts, stream);
}

// dr_bf3_wgrad_emb_parts in the f16x2 operand mode: table_amax as in dr_h2_emb_linear_fwd, dense_amax the record of dense_pad (required
// iff F > 64 nf), dy_amax the record of dy.  The records must be the same for part 1 and part 2 of one product.
extern "C" int dr_h2_wgrad_emb(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                               const uint32_t* table_amax, const float* dense_pad, const uint32_t* dense_amax, const float* dy, int64_t ld_dy,
                               const uint32_t* dy_amax, int32_t F, int32_t N, float scale, float* dstW, int64_t ld_w, float* dstb,
                               void* workspace, int64_t workspace_bytes, int32_t parts, dr_stream_t stream) {
    if (parts < 1 || parts > 3 || !table_amax || !dy_amax) return DR_EINVAL;
    return wgrad_emb_impl(ids_t, R, nf, row_base, table, D, dense_pad, dy, ld_dy, F, N, scale, dstW, ld_w, dstb, workspace, workspace_bytes,
                          parts, stream, table_amax, dense_amax, dy_amax);
}

// Fused K3 + first Dense layer (see bf3_emb_linear_kernel): h[m][n] = act(sum_k x[m][k] W[k][n] + bias[n]) with
// x = concat(field embeddings of ids[m], dense features = dense_pad[m, : K - 64 F]); also writes concat[:, : 64 F],
// sum_x [M, 64] and fm_logit [M] = lin_bias + sum_f lin_w[row] + 0.5 sum_d ((sum_f x_fd)^2 - sum_f x_fd^2).
static int emb_linear_fwd_impl(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max,
                               const float* table, int32_t D, const float* lin_w, const float* lin_bias, const float* dense_pad, float* concat,
                               int64_t ld_concat, int32_t K, const void* wt_planes, int64_t plane_stride, int64_t ld_planes, int32_t N,
                               const float* bias, int32_t act, float* sum_x, float* fm_logit, float* out, int64_t ld_out,
                               float* lin_vals_t, dr_stream_t stream, const uint32_t* table_amax = nullptr,
                               const uint32_t* dense_amax = nullptr, const uint32_t* w_amax = nullptr) {
    if (M < 0 || M > 0x7fffff00 || F <= 0 || N <= 0 || K < 64 * F || act < 0 || act > 1) return DR_EINVAL;
    // the k-tile <-> (field, half row) map is built for 64-wide rows; the dense features are one k-tile; a field is one 4 GB buffer
    if (D != 64 || K > 64 * F + 32 || field_rows_max <= 0 || field_rows_max > (1 << 24)) return DR_ESHAPE;
    if (M == 0) return DR_OK;
    // concat may be NULL: nothing then stores the gathered embeddings (the wgrad gathers them itself, dr_bf3_wgrad_emb)
    if (!ids || !row_base || !table || !sum_x || !fm_logit || !out || !planes_ok(wt_planes, plane_stride, ld_planes))
        return DR_EINVAL;
    if (K > 64 * F && (!dense_pad || (reinterpret_cast<uintptr_t>(dense_pad) & 15) != 0)) return DR_EINVAL;
    if ((concat != nullptr && ((reinterpret_cast<uintptr_t>(concat) & 15) != 0 || (ld_concat & 3) != 0 || ld_concat < K)) ||
        (reinterpret_cast<uintptr_t>(table) & 15) != 0 || (reinterpret_cast<uintptr_t>(sum_x) & 15) != 0)
        return DR_EINVAL;
    if (ld_planes < (K + BK - 1) / BK * BK || ld_out < N) return DR_EINVAL;
    RsArgs g{nullptr, 0, static_cast<const __bf16*>(wt_planes), plane_stride, ld_planes, M, N, K, out, ld_out, bias, act, nullptr, 0, 0,
             nullptr, nullptr, 0, 0.f, nullptr};
    EmbArgs e{ids, F, row_base, table, lin_w, lin_bias, K > 64 * F ? dense_pad : nullptr, concat, ld_concat, sum_x, fm_logit,
              lin_w != nullptr ? lin_vals_t : nullptr, K > 64 * F ? dense_amax : nullptr};
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    if (tiles > 0x7fffffff) return DR_EINVAL;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    if (table_amax != nullptr) {                                        // f16x2 operand mode
        if (!w_amax || (K > 64 * F && !dense_amax)) return DR_EINVAL;
        g.a_amax = table_amax;
        g.b_amax = w_amax;
        hipLaunchKernelGGL(bf3_emb_linear_kernel<1>, dim3(grid), dim3(512), 0, dr_s(stream), g, e);
    } else {
        hipLaunchKernelGGL(bf3_emb_linear_kernel<0>, dim3(grid), dim3(512), 0, dr_s(stream), g, e);
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_bf3_emb_linear_fwd(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max,
                                     const float* table, int32_t D, const float* lin_w, const float* lin_bias, const float* dense_pad, float* concat,
                                     int64_t ld_concat, int32_t K, const void* wt_planes, int64_t plane_stride, int64_t ld_planes, int32_t N,
                                     const float* bias, int32_t act, float* sum_x, float* fm_logit, float* out, int64_t ld_out,
                                     dr_stream_t stream) {
    return emb_linear_fwd_impl(ids, M, F, row_base, field_rows_max, table, D, lin_w, lin_bias, dense_pad, concat, ld_concat, K, wt_planes,
                               plane_stride, ld_planes, N, bias, act, sum_x, fm_logit, out, ld_out, nullptr, stream);
}

// The same, also saving the first-order weight of every slot as it was read: lin_vals_t [F, M] field-major (lin_vals_t[f * M + m] =
// lin_w[row_base[f] + ids[m, f]]; undefined for a missing id).  dr_emb_pool_bwd_sorted_ex takes it as `lin_old_t`: the backward then
// updates a unique row's first-order weight with ONE write instead of a read-modify-write of a line it would have to fetch again.
extern "C" int dr_bf3_emb_linear_fwd_lv(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max,
                                        const float* table, int32_t D, const float* lin_w, const float* lin_bias, const float* dense_pad,
                                        float* concat, int64_t ld_concat, int32_t K, const void* wt_planes, int64_t plane_stride,
                                        int64_t ld_planes, int32_t N, const float* bias, int32_t act, float* sum_x, float* fm_logit,
                                        float* out, int64_t ld_out, float* lin_vals_t, dr_stream_t stream) {
    return emb_linear_fwd_impl(ids, M, F, row_base, field_rows_max, table, D, lin_w, lin_bias, dense_pad, concat, ld_concat, K, wt_planes,
                               plane_stride, ld_planes, N, bias, act, sum_x, fm_logit, out, ld_out, lin_vals_t, stream);
}

// dr_bf3_emb_linear_fwd_lv in the f16x2 operand mode: wt_planes = two fp16 planes (dr_h2_split with w_amax); table_amax >= the largest
// magnitude in `table` (the engine keeps it as a running maximum: dr_h2_amax over the table once, K4 afterwards); dense_amax = the
// record of dense_pad (required iff K > 64 F).  lin_vals_t may be NULL.
extern "C" int dr_h2_emb_linear_fwd(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max,
                                    const float* table, int32_t D, const uint32_t* table_amax, const float* lin_w, const float* lin_bias,
                                    const float* dense_pad, const uint32_t* dense_amax, float* concat, int64_t ld_concat, int32_t K,
                                    const void* wt_planes, int64_t plane_stride, int64_t ld_planes, const uint32_t* w_amax, int32_t N,
                                    const float* bias, int32_t act, float* sum_x, float* fm_logit, float* out, int64_t ld_out,
                                    float* lin_vals_t, dr_stream_t stream) {
    if (!table_amax || !w_amax) return DR_EINVAL;
    return emb_linear_fwd_impl(ids, M, F, row_base, field_rows_max, table, D, lin_w, lin_bias, dense_pad, concat, ld_concat, K, wt_planes,
                               plane_stride, ld_planes, N, bias, act, sum_x, fm_logit, out, ld_out, lin_vals_t, stream, table_amax,
                               dense_amax, w_amax);
}
