// The wide fp32 GEMMs in the f16x2 operand mode, 16 waves per block ("occupancy" form; round 6).
//
// What it replaces: bf3_gemm.hip's register-split kernels (8 waves x (32 rows x 256 columns), two in-order waves per SIMD) for the
// Dense / Cross matmuls of the reference (keras/models/ranking/deepfm.py:30-34, dcn.py:81-88 and their autodiff).  Those sit at
// 0.22-0.31 of the f16x2 ceiling because every wait of a wave (lgkmcnt in front of a fragment, vmcnt in front of the split, the
// barrier) is a hole in one of only two instruction streams per SIMD; four rounds of software pipelining did not close it
// (docs/DESIGN_HISTORY.md).  Here latency is hidden by OCCUPANCY instead:
//
//   block = 256 x 256 x 32 tile, 1024 threads = 16 waves = FOUR per SIMD, wave (wr, wc) owns 64 rows x 64 columns
//           (2 x 2 accumulator tiles of v_mfma_f32_32x32x16_f16 = 64 registers; <= 128 registers per lane in all)
//   A (activations, fp32 in HBM): every thread loads 2 x 16 bytes of a 128-byte line (8 lanes per line, rows 16 w + r and + 8),
//           splits its 8 values ONCE into the two fp16 terms (x s = h + l) and writes them as ds_write_b64 into the swizzled
//           [plane][256 rows][32 k] image that the 4 waves of its row group read -- the split is done once per element, not once
//           per consuming wave, and the loads sit one whole k-tile ahead in 8 registers
//   B (weights, pre-split fp16 planes [2][N][ld]): LDS-DMA, 2 pieces of 1 KB per wave and k-tile, source-swizzled (the image of
//           bf3_gemm_rs_kernel: 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments)
//   two 64 KB stages, ONE barrier per k-tile: tile t is multiplied out of stage t & 1 while A(t + 1) is split into / B(t + 1)
//           lands in the other stage and A(t + 2) is in flight to registers
//   per wave and k-tile: 16 ds_read_b128 + 24 MFMAs + 2 global loads + 2 DMA pieces + ~40 VALU + 4 ds_write_b64
//   LDS fragment traffic per MFMA equals the 8-wave kernel's (8 reads per 12 MFMAs); the epilogue turns every 32 x 64 accumulator
//   block through the (then idle) ring and stores / loads 16 bytes per lane, 16 lanes on one 256-byte row piece.
//
// Products and their order per output element are those of bf3_gemm_rs_kernel<.., H2 = 1> (h_a l_b, l_a h_b, h_a h_b per 16-deep step,
// steps in k order): results are bit-identical to that kernel's, so every tolerance and tie argument made for it carries over.
#include "dr_common.h"
#include "rs_args.h"
#include <cstdlib>

namespace drrs {
namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define OCC_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define OCC_DS_WRITE_B64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define OCC_DS_WRITE_B32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int PLANE = 256 * 64;                                         // bytes: 256 rows x 64-byte rows (32 fp16)
constexpr int A_OFF = 0, B_OFF = 2 * PLANE, STAGE = 4 * PLANE;          // a stage: A h, A l, B h, B l = 64 KB

// EPI 8: the first-layer dgrad of the DeepFM tower with K4's unique-row pass as its epilogue (dr_h2_dgrad_emb_sgd below).  Column
// group f (64 columns) of row m IS the gradient row of slot (m, f): instead of storing it to d_concat for K4 to read back, the
// epilogue applies K4's update itself to every slot whose table row no other slot of the batch shares (the slot plan's flag):
//     g = dx + d_fm_logit[m] (sum_x[m] - x)        x = the table row as it stands (a unique row's value IS the forward's x)
//     table[row] = x + scale g ;  lin_w[row] = lin_old[f, m] + scale d_fm_logit[m]
// -- the arithmetic of emb_bwd_unique_body (emb_sorted.hip), operation for operation, so the tables end up bit-identical to
// dgrad + K4.  Slots that share a row (or are missing) get their dx stored to d_concat as before: K4's duplicate pass
// (dr_emb_pool_bwd_sorted_ex, parts | 8) reads it there.  0.88 GB of d_concat traffic per step disappears at config 3.
struct K4Args {
    const int32_t* ids_t;                        // [F, M] field-major bucket ids (-1 = missing)
    const uint8_t* flags;                        // [M, F] 1 = the slot's row is unique in the batch (slot plan)
    const int64_t* row_base; int32_t F;          // [F] first row of each field
    float* table; float* lin_w;                  // [R, 64], [R] (lin_w may be null)
    const float* lin_old_t;                      // [F, M] first-order weights as the forward read them (null iff lin_w is)
    const float* sum_x; const float* d_fm_logit; // [M, 64], [M]
    float scale;                                 // -lr
    float* d_concat; int64_t ld_dc;              // [M, ld] gradient rows of the NON-unique slots
    uint32_t* table_amax;                        // running amax record of the table (may be null)
};

// EPI: 0 = bias / ReLU, 1 = + ReLU' mask, 3 = accumulate (C +=), 8 = K4's unique-row update (above)
// DBG (tools/exp only; 0 in the product path): 2 = no MFMAs, 32 = no fragment reads, 1 = no B DMA after the prologue, 8 = A from cache
// (its loads re-read k-tile 0), 64 = no split / no A image writes
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(1024) void h2_occ_nt_kernel(RsArgs g, K4Args e) {
    // (EPI 8: + 8 KB behind the ring for the epilogue's per-row scalars -- ONE __shared__ object on purpose)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + (EPI == 8 ? 10240 : 0)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (int)((g.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int nk = (g.K + BK - 1) / BK;
    const bool ktail = (g.K % BK) != 0;
    const int kv4 = (g.K + 3) / 4 * 4;                                  // rows of A are readable up to here (lda % 4 == 0, lda >= K)

    float s_a, inv_a, s_b, inv_b;
    h2_scale_of(g.a_amax[0], s_a, inv_a);
    h2_scale_of(g.b_amax[0], s_b, inv_b);
    const float h2_out = inv_a * inv_b;
    h2_mode_on();
    float cmax = 0.f;

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    // fragment reads: row 32 t + l31 of the wave's 64 rows / columns, 16-byte chunk 2 hi + s of the k-tile, swizzled: the address of
    // k-step 1 is that of k-step 0 ^ 16
    const int sw = (l31 >> 2) & 3;
    const unsigned a_rd = lds0 + A_OFF + (64 * wr + l31) * 64 + (((2 * hi) ^ sw) << 4);
    const unsigned b_rd = lds0 + B_OFF + (64 * wc + l31) * 64 + (((2 * hi) ^ sw) << 4);
    // A staging: this thread's float4 c4 of rows 16 wave + r8 and + 8 (the second row's image address = (first ^ 32) + 512: its
    // swizzle term (row >> 2) & 3 differs in bit 1)
    const int r8 = lane >> 3, c4 = lane & 7;
    const int arow0 = 16 * wave + r8;
    const unsigned a_wr = lds0 + A_OFF + arow0 * 64 + ((((c4 >> 1) ^ ((arow0 >> 2) & 3)) << 4) | ((c4 & 1) << 3));
    const int avoff0 = (int)((arow0 * g.lda + 4 * c4) * 4), avoff1 = avoff0 + (int)(8 * g.lda * 4);
    // B pieces: piece p = plane p, rows 16 wave .. + 15; lane -> row 16 wave + (lane >> 2), image slot lane & 3 = source chunk ^ swizzle
    const int brow = 16 * wave + (lane >> 2);
    const int bchunk = (lane & 3) ^ ((brow >> 2) & 3);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(g.B), 0, (int)min((int64_t)0x7fffffff, 2 * g.b_ps * 2), 0x00020000);
    const int b_plane1 = (int)(g.b_ps * 2);                             // byte offset of plane 1 (scalar)

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[a][b][k] = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int lid = xcd_remap(tile, ntiles);
        const int64_t tm0 = (int64_t)(lid / tiles_n) * BM;
        const int tn0 = (lid % tiles_n) * BN;
        // ---- per-tile addresses -----------------------------------------------------------------------------------------------
        // A through a buffer resource over the tile's rows: base = row tm0, range = up to the readable end of row M - 1, so rows
        // past the edge read zeros (they only feed unstored outputs) and no load leaves the matrix; offsets are 32 bits
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g.A + tm0 * g.lda), 0, (int)min((int64_t)0x7fffffff, ((g.M - tm0 - 1) * g.lda + kv4) * 4), 0x00020000);
        const int bvoff = (int)(((int64_t)min(tn0 + brow, g.N - 1) * g.b_ld + bchunk * 8) * 2);
        auto issue_b = [&](int kt, int stage) {
            if constexpr (DBG & 1) { if (kt > 0) return; }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                unsigned char* dst = smem + stage * STAGE + B_OFF + p * PLANE + wave * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)dst, 16, bvoff, kt * (BK * 2) + p * b_plane1, 0, 0);
            }
        };
        f32x4 an0, an1;                                                 // A of the next k-tile, in flight
        int an_kt = 0;                                                  // ... its k-tile
        // (compiler-visible loads on purpose.  Issued from inline asm with hand-counted waits they measured the same -- hipcc's own waits
        // in front of the split come out as vmcnt(1) / vmcnt(0), i.e. they also wait for the weight pieces issued a moment earlier,
        // which have landed by then anyway -- and they are UNSAFE: in an instantiation short of registers (EPI 8) the compiler
        // spilled the destination registers of the in-flight loads, whose contents it believed valid, and reloaded stale bits)
        auto load_a = [&](int kt) {
            if constexpr (DBG & 8) kt = 0;
            an_kt = kt;
            an0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff0, kt * (BK * 4), 0));
            an1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, avoff1, kt * (BK * 4), 0));
        };
        auto wait_a = [&](bool) {};
        // one of the thread's two float4s of the k-tile in flight -> the two fp16 terms -> stage's A image
        auto stage_a = [&](int which, int stage) {
            if constexpr (DBG & 64) return;
            f32x4 v = which == 0 ? an0 : an1;
            if (ktail) {                                                // (kernel-uniform) B's planes are zero at k >= K, but 0 * NaN is not
                const int k = an_kt * BK + 4 * c4;
                v[0] = k < g.K ? v[0] : 0.f; v[1] = k + 1 < g.K ? v[1] : 0.f; v[2] = k + 2 < g.K ? v[2] : 0.f; v[3] = k + 3 < g.K ? v[3] : 0.f;
            }
            f16x4 h, l;
            if constexpr (DBG & 256) {                                  // ablation: no split arithmetic (raw bits)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                h = __builtin_bit_cast(f16x4, f32x2{v[0], v[1]});
                l = __builtin_bit_cast(f16x4, f32x2{v[2], v[3]});
            } else {
                const f32x4 x = v * s_a;
                h = __builtin_convertvector(x, f16x4);
                l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), f16x4);
            }
            const unsigned w = (which == 0 ? a_wr : (a_wr ^ 32u) + 512u) + stage * STAGE;
            if constexpr (DBG & 128) {                                  // ablation: no image writes
                asm volatile("" :: "v"(h), "v"(l), "v"(w));
            } else {
                OCC_DS_WRITE_B64(w, h, 0);
                OCC_DS_WRITE_B64(w, l, PLANE);
            }
        };
        f16x8 ah[2], al[2], bh[2], bl[2];                               // fragments [tile]: the two terms of A's rows / B's columns
#define OCC_MMA(A, B)                                                                                       \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                       \
            _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                   \
                if constexpr (!(DBG & 2)) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a], B[b], acc[a][b], 0, 0, 0);
#define OCC_MMA2(A, B, AI)                                                                                  \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                       \
            if constexpr (!(DBG & 2)) acc[AI][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[AI], B[b], acc[AI][b], 0, 0, 0);
#define OCC_RD(DST, ADDR, OFF) if constexpr (!(DBG & 32)) OCC_DS_READ_B128(DST, ADDR, OFF); else asm volatile("" : "=v"(DST));
#define OCC_FENCE() __builtin_amdgcn_sched_barrier(0)

        // ---- prologue: B(0) by DMA, A(0) split into stage 0, A(1) into flight; then B(1) and k-step 0's fragments ---------------------
        issue_b(0, 0);
        load_a(0);
        OCC_FENCE();
        wait_a(false);
        stage_a(0, 0);
        stage_a(1, 0);
        OCC_FENCE();
        load_a(nk > 1 ? 1 : 0);
        OCC_FENCE();
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (nk > 1) issue_b(1, 1);
        {
            const unsigned aa = a_rd, bb = b_rd;
            OCC_RD(bl[0], bb, PLANE) OCC_RD(bl[1], bb, PLANE + 2048) OCC_RD(al[0], aa, PLANE) OCC_RD(al[1], aa, PLANE + 2048)
            OCC_RD(ah[0], aa, 0) OCC_RD(ah[1], aa, 2048) OCC_RD(bh[0], bb, 0) OCC_RD(bh[1], bb, 2048)
        }
        OCC_FENCE();

        // ---- main loop: a ROLLING pipeline over the 16-deep steps.  Per step and wave 12 MFMAs in the order h_a l_b (x 4), l_a h_b (x 4),
        // h_a h_b (x 4); a fragment register is re-read for the NEXT step as soon as the last MFMA that names it has been issued (b_l
        // after the first four, a_l after the second, a_h / b_h after the third), so the matrix pipe always has this wave's next
        // instructions queued while its LDS reads are in flight.  The one barrier per k-tile sits INSIDE step 1's chain, behind its first
        // four MFMAs: it publishes stage ^ 1 (A(kt + 1) split into it during step 0's chain, B(kt + 1) landed) and frees `stage` (every
        // read of it has returned) -- and the eight MFMAs after it need no LDS data, so nothing waits for a read right behind a barrier.
        // LDS queue of a wave per k-tile, in order: bl' x2 | W x2 | al' x2 | W x2 | ah' bh' x4 || barrier || bl" x2 | al" x2 | ah" bh" x4
        // (' = step 1 of this k-tile, " = step 0 of the next; W = image writes) -- the counted waits below name positions in it.
        for (int kt = 0; kt < nk; ++kt) {
            const int stage = kt & 1;
            const bool more = kt + 1 < nk;                              // (uniform) a next k-tile exists
            const unsigned aa1 = (a_rd + stage * STAGE) ^ 16u, bb1 = (b_rd + stage * STAGE) ^ 16u;      // k-step 1 of this k-tile
            // ---- step 0 ----
            if constexpr (!(DBG & 32)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[0]), "+v"(bl[0]), "+v"(ah[1]), "+v"(bl[1]), "+v"(al[0]), "+v"(al[1]));
            OCC_MMA(ah, bl)
            OCC_FENCE();
            OCC_RD(bl[0], bb1, PLANE) OCC_RD(bl[1], bb1, PLANE + 2048)
            if constexpr (!(DBG & 32)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[0]), "+v"(bh[1]));
            OCC_MMA2(al, bh, 0)
            OCC_FENCE();
            if (more) {
                wait_a(true);                                           // A(kt + 1); the pieces of B(kt + 1) were issued after its loads
                stage_a(0, stage ^ 1);
            }
            OCC_FENCE();
            OCC_MMA2(al, bh, 1)
            OCC_FENCE();
            OCC_RD(al[0], aa1, PLANE) OCC_RD(al[1], aa1, PLANE + 2048)
            OCC_MMA2(ah, bh, 0)
            OCC_FENCE();
            if (more) stage_a(1, stage ^ 1);
            OCC_FENCE();
            OCC_MMA2(ah, bh, 1)
            OCC_FENCE();
            OCC_RD(ah[0], aa1, 0) OCC_RD(ah[1], aa1, 2048) OCC_RD(bh[0], bb1, 0) OCC_RD(bh[1], bb1, 2048)
            load_a(kt + 2 < nk ? kt + 2 : nk - 1);                      // (always issued: the counted waits assume [pieces x2][loads x2])
            OCC_FENCE();
            // ---- step 1 ----
            if constexpr (!(DBG & 32)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[0]), "+v"(ah[1]), "+v"(bl[0]), "+v"(bl[1]), "+v"(al[0]), "+v"(al[1]));
            OCC_MMA(ah, bl)
            OCC_FENCE();
            // every read of `stage` has returned, this wave's share of stage ^ 1 is complete (image writes retired, pieces landed: the
            // two A loads issued after them may stay in flight)
            if constexpr (!(DBG & 512)) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" : "+v"(bh[0]), "+v"(bh[1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" : "+v"(bh[0]), "+v"(bh[1]) :: "memory");
            if (more) issue_b(kt + 2 < nk ? kt + 2 : nk - 1, stage);    // (past the end: a dummy re-fetch, so that the count above holds)
            const unsigned aa2 = a_rd + (stage ^ 1) * STAGE, bb2 = b_rd + (stage ^ 1) * STAGE;           // k-step 0 of the next k-tile
            if (more) { OCC_RD(bl[0], bb2, PLANE) OCC_RD(bl[1], bb2, PLANE + 2048) }
            OCC_FENCE();
            OCC_MMA(al, bh)
            OCC_FENCE();
            if (more) { OCC_RD(al[0], aa2, PLANE) OCC_RD(al[1], aa2, PLANE + 2048) }
            OCC_MMA(ah, bh)
            OCC_FENCE();
            if (more) { OCC_RD(ah[0], aa2, 0) OCC_RD(ah[1], aa2, 2048) OCC_RD(bh[0], bb2, 0) OCC_RD(bh[1], bb2, 2048) }
            OCC_FENCE();
            if constexpr (DBG & 2) acc[0][0][0] += (float)ah[0][0] + (float)ah[1][0] + (float)al[0][0] + (float)al[1][0] + (float)bh[0][0] + (float)bh[1][0] + (float)bl[0][0] + (float)bl[1][0];
        }
        // (the last k-tile's barrier was the last LDS access of the main loop: the ring is free for the epilogue)
#undef OCC_MMA
#undef OCC_MMA2
#undef OCC_RD
#undef OCC_FENCE

        // ---- epilogue: C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ------------
        // Each 32 x 64 block of the wave (row tile a) goes through the wave's 8 KB of the idle ring, row-major (256-byte rows), and
        // comes back as float4s, 16 lanes on one row: 16-byte loads / stores, 4 rows per instruction.
        {
            const bool relu = g.act == 1;
            const bool interior = tm0 + BM <= g.M && tn0 + BN <= g.N;
            // (the lane index laundered: everything the epilogue derives from it is recomputed here, per tile -- hoisted out of the tile
            // loop by the compiler these values live across the main loop, whose 128 registers are spoken for, and are spilled INTO it)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const unsigned stg = lds0 + wave * 8192;
            const unsigned st_wr = stg + (4 * (lane_e >> 5)) * 256 + (lane_e & 31) * 4;      // + ((reg & 3) + 8 (reg >> 2)) * 256 + b * 128
            const int prow = lane_e >> 4, pc4 = (lane_e & 15) * 4;      // storing pass: row within a 4-row group, first of 4 columns
            const unsigned st_rd = stg + prow * 256 + pc4 * 4;          // + 4 it * 256
            const int col0 = tn0 + 64 * wc + pc4;
            float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias != nullptr) {
                if (interior) bj = *reinterpret_cast<const float4*>(g.bias + col0);
                else {
                    bj.x = g.bias[min(col0, g.N - 1)]; bj.y = g.bias[min(col0 + 1, g.N - 1)];
                    bj.z = g.bias[min(col0 + 2, g.N - 1)]; bj.w = g.bias[min(col0 + 3, g.N - 1)];
                }
            }
            // EPI 8: the tile's buffer resources (bases made wave-uniform by hand: a resource the compiler cannot prove uniform is
            // applied lane by lane in a waterfall loop)
            const int k4_f = (tn0 >> 6) + wc;
            const bool k4_on = EPI == 8 && k4_f < e.F;
            const int k4_rows = (int)min((int64_t)BM, g.M - tm0);
            const int k4_ldc4 = (int)(e.ld_dc * 4);
            __amdgpu_buffer_rsrc_t k4_t, k4_dl, k4_lo, k4_sx, k4_dc, k4_lw, k4_id, k4_fl;
            if constexpr (EPI == 8) {
                auto mk = [&](const void* p, int64_t bytes) {
                    const uint64_t b = reinterpret_cast<uint64_t>(p);
                    void* q = reinterpret_cast<void*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                                      (unsigned)__builtin_amdgcn_readfirstlane((int)b));
                    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)(unsigned)(bytes < 0xffffff00ll ? bytes : 0xffffff00ll), 0x00020000);
                };
                const int fc = k4_on ? k4_f : 0;
                const int64_t rb = e.row_base[fc];
                const bool has_lin = e.lin_w != nullptr;
                k4_t = mk(e.table + rb * 64, 0xffffff00ll);             // (the field's rows: id * 256 < 4 GB - 256, i.e. < 2^24 - 1 rows per field)
                k4_dl = mk(e.d_fm_logit + tm0, (int64_t)k4_rows * 4);
                k4_lo = has_lin ? mk(e.lin_old_t + (int64_t)fc * g.M + tm0, (int64_t)k4_rows * 4) : k4_dl;
                k4_sx = mk(e.sum_x + tm0 * 64, (int64_t)k4_rows * 256);
                k4_dc = mk(e.d_concat + tm0 * e.ld_dc + 64 * fc, ((int64_t)k4_rows - 1) * e.ld_dc * 4 + 256);
                k4_lw = has_lin ? mk(e.lin_w + rb, 0x7fffffffll) : mk(e.table, 0);      // (no first-order table: every store out of range)
                k4_id = mk(e.ids_t + (int64_t)fc * g.M + tm0, (int64_t)k4_rows * 4);
                k4_fl = mk(e.flags + tm0 * e.F, (int64_t)k4_rows * e.F);
            }
            if constexpr (EPI == 8) {
                // ---- K4's unique-row pass on the tile's 256 x 4 slots.  What bounds it is HBM LATENCY (a table row can only be asked for once
                // its id is known, and written once it has arrived): the fewer dependent round trips per tile and the more rows in flight per
                // round trip, the better.  So everything a slot needs BESIDES its table row is made LDS-resident first -- the tile's 256 rows
                // of sum_x (64 KB, by LDS-DMA), d_fm_logit and the saved first-order weights -- the accumulators are staged 16 rows at a time
                // (4 KB per wave), and a pass keeps its EIGHT row groups' table rows in flight together (32 registers, nothing else waits in
                // registers): two HBM round trips per tile.  Everything is addressed through buffer resources with 32-bit offsets relative
                // to the tile, and every store is UNCONDITIONAL: a lane that must not write is given an offset past the end of its resource
                // (and far from 2^32, where offset + size would wrap), which the memory unit drops -- the updated table row for a unique
                // slot, the gradient row into d_concat for the others.
                constexpr unsigned SXP = 65536, DLP = 2 * STAGE, LOP = 2 * STAGE + 1024, UIP = 2 * STAGE + 1024 + 4096;   // sum_x panel / d_fm_logit [256] / lin_old [16][64] / row ids [16][64]
                const unsigned stq = lds0 + wave * 4096;                                     // this wave's 16-row staging area
                const unsigned stq_wr = stq + (4 * (lane_e >> 5)) * 256 + (lane_e & 31) * 4; // + ((reg & 3) + 8 ((reg >> 2) & 1)) * 256 + b * 128
                const unsigned stq_rd = stq + prow * 256 + pc4 * 4;                          // + 4 i * 256
                {
                    // sum_x rows 16 wave .. + 15 of the tile (4 pieces of 4 rows: lane -> row lane / 16, 16-byte chunk lane % 16: lane-linear)
#pragma unroll
                    for (int p4 = 0; p4 < 4; ++p4)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(k4_sx, (lds_ptr_t)(smem + SXP + (16 * wave + 4 * p4) * 256), 16,
                                                                 (16 * wave + 4 * p4 + (lane_e >> 4)) * 256 + (lane_e & 15) * 16, 0, 0, 0);
                    // per-row scalars of slot (row 64 wr + lane, this wave's field): d_fm_logit, the saved first-order weight, and the slot's
                    // row id -- or -1 when the slot is not for this pass to update (row shared with another slot, id missing, row past M)
                    unsigned dlv = __builtin_amdgcn_raw_buffer_load_b32(k4_dl, (64 * wr + lane_e) * 4, 0, 0);
                    unsigned lov = __builtin_amdgcn_raw_buffer_load_b32(k4_lo, (64 * wr + lane_e) * 4, 0, 0);
                    int idl = (int)__builtin_amdgcn_raw_buffer_load_b32(k4_id, (64 * wr + lane_e) * 4, 0, 0);
                    int fll = (int)__builtin_amdgcn_raw_buffer_load_b8(k4_fl, (64 * wr + lane_e) * e.F + k4_f, 0, 0);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dlv), "+v"(lov), "+v"(idl), "+v"(fll) :: "memory");
                    const int uid = (k4_on && 64 * wr + lane_e < k4_rows && idl >= 0 && fll != 0) ? idl : -1;
                    OCC_DS_WRITE_B32(lds0 + DLP + (64 * wr + lane_e) * 4, dlv, 0);
                    OCC_DS_WRITE_B32(lds0 + LOP + wave * 256 + lane_e * 4, lov, 0);
                    OCC_DS_WRITE_B32(lds0 + UIP + wave * 256 + lane_e * 4, uid, 0);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // panel, scalars: visible to every wave
                }
                if (k4_on) {                                            // (uniform) this wave's field exists
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int ml0 = 64 * wr + 32 * a + prow;        // row within the tile of this lane's group 0 (+ 4 it)
                        unsigned voff[8];
                        unsigned uqm = 0;
                        u32x4 t[8];
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int uidv = *reinterpret_cast<const int*>(smem + UIP + wave * 256 + (32 * a + 4 * it + prow) * 4);
                            if (uidv >= 0) uqm |= 1u << it;
                            voff[it] = (unsigned)max(uidv, 0) * 256u + (unsigned)pc4 * 4u;
                            t[it] = __builtin_amdgcn_raw_buffer_load_b128(k4_t, (int)voff[it], 0, 2);       // (aux 2 = nt)
                        }
                        // (an EXPLICIT wait for the rows.  hipcc counts vmcnt as if loads and stores retired in issue order; they do not -- a
                        // register spill to scratch or a store the memory unit drops is acknowledged at once -- and with one of those between
                        // a load and its use the counted wait the compiler inserts is satisfied early: measured, an FM term computed from a
                        // sum_x that had not arrived.  This wait also covers the previous pass's stores.)
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
                        for (int q = 0; q < 2; ++q) {                   // 16 rows at a time through the wave's 4 KB
#pragma unroll
                            for (int b = 0; b < 2; ++b)
#pragma unroll
                                for (int r8 = 0; r8 < 8; ++r8) {
                                    const int reg = 8 * q + r8;
                                    const float v = acc[a][b][reg] * h2_out;
                                    acc[a][b][reg] = 0.f;
                                    switch (b * 8 + r8) {                // (immediates must be literal)
#define OCC_W(I) case I: OCC_DS_WRITE_B32(stq_wr, v, ((I & 3) + 8 * ((I & 7) >> 2)) * 256 + (I >> 3) * 128); break;
                                        OCC_W(0) OCC_W(1) OCC_W(2) OCC_W(3) OCC_W(4) OCC_W(5) OCC_W(6) OCC_W(7) OCC_W(8) OCC_W(9) OCC_W(10) OCC_W(11)
                                        OCC_W(12) OCC_W(13) OCC_W(14) OCC_W(15)
#undef OCC_W
                                    }
                                }
                            f32x4 v[4];
                            // the quarter's four staged rows: writes retired, reads AND their wait in one asm statement (with the wait apart
                            // the compiler, short of registers here, copied / spilled the destinations before the data had arrived)
                            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:1024\n\t"
                                         "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(stq_rd) : "memory");
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int it = 4 * q + i;
                                const int ml = ml0 + 4 * it;
                                const bool uq = (uqm >> it) & 1u;
                                const f32x4 tv = __builtin_bit_cast(f32x4, t[it]);
                                const f32x4 sv = *reinterpret_cast<const f32x4*>(smem + SXP + ml * 256 + pc4 * 4);
                                const float dl = *reinterpret_cast<const float*>(smem + DLP + ml * 4);
                                const float lo = *reinterpret_cast<const float*>(smem + LOP + wave * 256 + (32 * a + 4 * it + prow) * 4);
                                // emb_bwd_unique_body's arithmetic (g += dl (sx - x) ; x = fma(scale, g, x))
                                float4 gq = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
                                gq.x += dl * (sv[0] - tv[0]); gq.y += dl * (sv[1] - tv[1]);
                                gq.z += dl * (sv[2] - tv[2]); gq.w += dl * (sv[3] - tv[3]);
                                f32x4 r;
                                r[0] = fmaf(e.scale, gq.x, tv[0]); r[1] = fmaf(e.scale, gq.y, tv[1]);
                                r[2] = fmaf(e.scale, gq.z, tv[2]); r[3] = fmaf(e.scale, gq.w, tv[3]);
                                if (uq) cmax = fmaxf(cmax, fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3]))));
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), k4_t, uq ? (int)voff[it] : (int)0xffffff00u, 0, 2);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), k4_dc,
                                                                       (uq || ml >= k4_rows) ? (int)0x80000000u : ml * k4_ldc4 + pc4 * 4, 0, 2);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaf(e.scale, dl, lo)), k4_lw,
                                                                      (uq && pc4 == 0) ? (int)(voff[it] >> 6) : (int)0x80000000u, 0, 0);
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the quarter's reads retired before the next one's writes
                        }
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int reg = 0; reg < 16; ++reg) acc[a][b][reg] = 0.f;
                }
            } else
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float v = acc[a][b][reg] * h2_out;
                        acc[a][b][reg] = 0.f;
                        switch (b * 16 + reg) {                          // (immediates must be literal)
#define OCC_W(I) case I: OCC_DS_WRITE_B32(st_wr, v, ((I & 3) + 8 * ((I & 15) >> 2)) * 256 + (I >> 4) * 128); break;
                            OCC_W(0) OCC_W(1) OCC_W(2) OCC_W(3) OCC_W(4) OCC_W(5) OCC_W(6) OCC_W(7) OCC_W(8) OCC_W(9) OCC_W(10) OCC_W(11)
                            OCC_W(12) OCC_W(13) OCC_W(14) OCC_W(15) OCC_W(16) OCC_W(17) OCC_W(18) OCC_W(19) OCC_W(20) OCC_W(21) OCC_W(22)
                            OCC_W(23) OCC_W(24) OCC_W(25) OCC_W(26) OCC_W(27) OCC_W(28) OCC_W(29) OCC_W(30) OCC_W(31)
#undef OCC_W
                        }
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave, in-order LDS: written before it is read)
                const int64_t rbase = tm0 + 64 * wr + 32 * a + prow;
#pragma unroll
                for (int half = 0; half < 2; ++half) {                  // 4 row groups at a time: short-lived temporaries
                    f32x4 v[4];
                    float4 aux[4] = {};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int it = 4 * half + i;
                        switch (it) {
#define OCC_R(I) case I: OCC_DS_READ_B128(v[I & 3], st_rd, I * 1024); break;
                            OCC_R(0) OCC_R(1) OCC_R(2) OCC_R(3) OCC_R(4) OCC_R(5) OCC_R(6) OCC_R(7)
#undef OCC_R
                        }
                        if constexpr (EPI == 1 || EPI == 3) {
                            const int64_t row = rbase + 4 * it;
                            if (interior) {
                                aux[i] = EPI == 1 ? *reinterpret_cast<const float4*>(g.mask + row * g.ld_mask + col0)
                                                  : *reinterpret_cast<const float4*>(g.C + row * g.ldc + col0);
                            } else {
                                const float* src = EPI == 1 ? g.mask + min(row, g.M - 1) * g.ld_mask : g.C + min(row, g.M - 1) * g.ldc;
                                aux[i].x = src[min(col0, g.N - 1)]; aux[i].y = src[min(col0 + 1, g.N - 1)];
                                aux[i].z = src[min(col0 + 2, g.N - 1)]; aux[i].w = src[min(col0 + 3, g.N - 1)];
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t row = rbase + 4 * (4 * half + i);
                        float o[4] = {v[i][0] + bj.x, v[i][1] + bj.y, v[i][2] + bj.z, v[i][3] + bj.w};
                        const float ax[4] = {aux[i].x, aux[i].y, aux[i].z, aux[i].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = relu ? fmaxf(o[e], 0.f) : o[e];
                            if constexpr (EPI == 1) o[e] = ax[e] > 0.f ? o[e] : 0.f;
                            if constexpr (EPI == 3) o[e] = ax[e] + o[e];
                        }
                        if (interior) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) cmax = fmaxf(cmax, fabsf(o[e]));
                            *reinterpret_cast<float4*>(g.C + row * g.ldc + col0) = make_float4(o[0], o[1], o[2], o[3]);
                        } else if (row < g.M) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col0 + e < g.N) {
                                    cmax = fmaxf(cmax, fabsf(o[e]));
                                    g.C[row * g.ldc + col0 + e] = o[e];
                                }
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the block's reads retired before the next block's writes
            }
            // the ring is the next tile's again: every wave done with its staging area, and no store of this tile pending into the
            // next tile's counted waits
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    uint32_t* const amax_out = EPI == 8 ? e.table_amax : g.c_amax;
    if (amax_out != nullptr) {                                          // (kernel-uniform) one load per wave, an atomic only if it raises the record
        uint32_t m = __float_as_uint(cmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0 && m > __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_out, m);
    }
}

}  // namespace

int occ_nt_launch(const RsArgs& g, hipStream_t stream) {
    // domain: the f16x2 mode's plain epilogues; 16-byte rows of C (and of the mask) so that the epilogue's float4 moves are aligned
    if (g.a_amax == nullptr || g.b_amax == nullptr) return 1;
    if (g.x0 != nullptr || g.tau != nullptr || g.pack_pos != nullptr || g.sm_part_m != nullptr || g.sm_lse != nullptr) return 1;
    // short reductions with the plain store epilogue (the first-layer dgrad shape, K = 256: eight k-tiles per output tile) stay on the
    // 8-wave kernel, whose pipeline runs on across tile boundaries: 266 - 279 us against 291 here; everything else measured equal or
    // faster on this kernel (forward shape 206 vs 231, dgrad + mask 406 vs 428, 1677^2 accumulate 1264 vs 1329; tools/exp/occ_bench.py)
    if (g.mask == nullptr && !g.accumulate && g.K < 512) return 1;
    if ((g.ldc & 3) != 0 || (reinterpret_cast<uintptr_t>(g.C) & 15) != 0) return 1;
    if (g.mask != nullptr && ((g.ld_mask & 3) != 0 || (reinterpret_cast<uintptr_t>(g.mask) & 15) != 0)) return 1;
    if (g.bias != nullptr && (reinterpret_cast<uintptr_t>(g.bias) & 15) != 0) return 1;
    if (2 * g.b_ps * 2 + (int64_t)BK * 2 * ((g.K + BK - 1) / BK) > 0x7fffffffll) return 1;     // 32-bit buffer offsets (B planes)
    if ((int64_t)(BM + 8) * g.lda * 4 + (int64_t)g.K * 4 > 0x7fffffffll) return 1;               // ... and within a row tile of A
    const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (tiles > 0x7fffffff) return DR_EINVAL;
    const int grid = (int)(tiles < 256 ? tiles : 256);                  // persistent: one block per CU
#ifdef DR_OCC_ABLATE
    {
        static const int dbg = [] { const char* e = getenv("DR_OCC_DBG"); return e ? atoi(e) : 0; }();
        if (dbg != 0 && g.mask == nullptr && !g.accumulate) {
#define OCC_ABL(D)                                                                                      \
            if (dbg == D) {                                                                             \
                hipLaunchKernelGGL((h2_occ_nt_kernel<0, D>), dim3(grid), dim3(1024), 0, stream, g, K4Args{});     \
                DR_CHECK_LAUNCH();                                                                      \
                return DR_OK;                                                                           \
            }
            OCC_ABL(2) OCC_ABL(32) OCC_ABL(34) OCC_ABL(1) OCC_ABL(8) OCC_ABL(9) OCC_ABL(64) OCC_ABL(73) OCC_ABL(137) OCC_ABL(265) OCC_ABL(393) OCC_ABL(521) OCC_ABL(585)
#undef OCC_ABL
        }
    }
#endif
    if (g.mask != nullptr) hipLaunchKernelGGL((h2_occ_nt_kernel<1>), dim3(grid), dim3(1024), 0, stream, g, K4Args{});
    else if (g.accumulate) hipLaunchKernelGGL((h2_occ_nt_kernel<3>), dim3(grid), dim3(1024), 0, stream, g, K4Args{});
    else hipLaunchKernelGGL((h2_occ_nt_kernel<0>), dim3(grid), dim3(1024), 0, stream, g, K4Args{});
    DR_CHECK_LAUNCH();
    return DR_OK;
}

}  // namespace drrs

// First-layer dgrad of the DeepFM tower + K4's unique-row pass in one launch (EPI 8 above):  dx = dy W^T  (W as fp16 planes [2][>= 64 F
// rows][ld >= roundup(K, 32)], the layer's kernel rows = input columns), never stored for slots whose table row is unique in the batch --
// those rows and their first-order weights receive K4's SGD update on the spot -- and stored to d_concat [M, ld_dc] for the others.
// Follow with dr_emb_pool_bwd_sorted_ex(parts | 8) (the duplicate pass + the first-order bias) on the same stream.
// Autodiff of keras/models/ranking/deepfm.py:30-34 w.r.t. the concatenated embeddings + of fm.py:23-37 / safe_embedding_lookup_sparse
// w.r.t. the tables (reference root), with the SGD step of examples/train_fm_on_movielens_estimator.py:51-52 fused (SGD instead of Adam).
extern "C" int dr_h2_dgrad_emb_sgd(const float* dy, int64_t ld_dy, const uint32_t* dy_amax, const void* w_planes, int64_t w_ps,
                                   int64_t w_ld, const uint32_t* w_amax, int64_t M, int32_t F, int32_t K, const int32_t* ids_t,
                                   const uint8_t* unique_flags, const int64_t* row_base, float* table, float* lin_w,
                                   const float* lin_old_t, const float* sum_x, const float* d_fm_logit, float scale, float* d_concat,
                                   int64_t ld_dconcat, uint32_t* table_amax, dr_stream_t stream) {
    using namespace drrs;
    if (M < 0 || F <= 0 || F > 64 || K <= 0 || !dy_amax || !w_amax) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!dy || !w_planes || !ids_t || !unique_flags || !row_base || !table || !sum_x || !d_fm_logit || !d_concat) return DR_EINVAL;
    if ((lin_w != nullptr) != (lin_old_t != nullptr)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) != 0 || (ld_dy & 3) != 0 || ld_dy < K) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(w_planes) & 15) != 0 || w_ld < (K + BK - 1) / BK * BK || (w_ld & 7) != 0 || (w_ps & 7) != 0) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_concat) & 15) != 0 || (ld_dconcat & 3) != 0 || ld_dconcat < 64 * (int64_t)F) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(table) & 15) != 0 || (reinterpret_cast<uintptr_t>(sum_x) & 15) != 0) return DR_EINVAL;
    if (2 * w_ps * 2 + (int64_t)BK * 2 * ((K + BK - 1) / BK) > 0x7fffffffll) return DR_ESHAPE;
    if ((int64_t)(BM + 8) * ld_dy * 4 + (int64_t)K * 4 > 0x7fffffffll) return DR_ESHAPE;
    RsArgs g{};
    g.A = dy; g.lda = ld_dy; g.B = static_cast<const __bf16*>(w_planes); g.b_ps = w_ps; g.b_ld = w_ld;
    g.M = M; g.N = 64 * F; g.K = K;
    g.a_amax = dy_amax; g.b_amax = w_amax;
    K4Args e{ids_t, unique_flags, row_base, F, table, lin_w, lin_old_t, sum_x, d_fm_logit, scale, d_concat, ld_dconcat, table_amax};
    const int64_t tiles = ((M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (tiles > 0x7fffffff) return DR_EINVAL;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL((h2_occ_nt_kernel<8>), dim3(grid), dim3(1024), 0, dr_s(stream), g, e);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
