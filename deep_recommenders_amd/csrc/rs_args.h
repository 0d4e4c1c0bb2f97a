// Argument block of the register-split / occupancy GEMM kernels and the f16x2 operand format's helpers, shared by bf3_gemm.hip (the
// 8-wave register-split kernels) and h2_occ.hip (the 16-wave kernels of round 6).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace drrs {

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {          // bijective, consecutive logical ids -> same XCD
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct RsArgs {
    const float* A; int64_t lda;
    const __bf16* B; int64_t b_ps, b_ld;
    int64_t M; int32_t N; int32_t K;             // K: true reduction length; B's planes are zero in [K, roundup(K, 32))
    float* C; int64_t ldc;
    const float* bias; int32_t act;              // C = act(acc + bias[n])
    const float* mask; int64_t ld_mask;          // optional: C = 0 where mask[m][n] <= 0   (ReLU' of the layer below)
    int32_t accumulate;                          // C += instead of C =
    // DCN cross layer (keras/models/ranking/dcn.py:81-88): prod = acc + bias + diag * x ; C = x0 * prod + x ; prod saved
    const float* x0; const float* xin; int64_t ldx; float diag; float* prod_out;   // x0 == nullptr: plain epilogue
    // top-K scan (factorized_top_k.py:201-233 of the reference: scores of a corpus chunk that can still enter a row's top k): C is not
    // written; a score is kept only if it beats its row's current k-th best tau[row], appended to the row's candidate list
    // cand_s / cand_c [M][cand_cap] through the cursor cand_cnt[row] (one atomic per (row, 32 columns) that has any)
    const float* tau; float* cand_s; int32_t* cand_c; int32_t* cand_cnt; int64_t cand_cap;
    // per-slot gradient rows into an all-to-all send layout (the sharded engine's first-layer dgrad + dr_emb_pack_grads in one): column
    // c < 64 pack_F of example m is dimension c & 63 of slot (m, c >> 6), whose destination row is pack_pos[m, c >> 6]:
    //   C[pack_pos * 64 + (c & 63)] = acc + pack_dl[m] * (pack_sumx[m, c & 63] - xin[m, c])      (FM term iff pack_sumx != nullptr)
    //   pack_lin[pack_pos] = pack_dl[m]                                                           (iff pack_lin != nullptr)
    // columns >= 64 pack_F (dense features) are computed and dropped.  xin / ldx: the concatenated embeddings of the forward.
    const int64_t* pack_pos; int32_t pack_F; const float* pack_dl; const float* pack_sumx; float* pack_lin;
    // f16x2 mode (H2 kernels only): the amax records of A and of the tensor B's two planes were split from
    const uint32_t* a_amax; const uint32_t* b_amax;
    uint32_t* c_amax;                            // (H2, may be null) out: raised to max |value stored into C| -- the record of the NEXT GEMM's operand
    // in-batch sampled softmax (keras/models/retrieval/sbcnm.py:120-151; H2 kernels only, round 5): the scores A B^T never leave the
    // tile.  s_ij = (acc + (-log cand_prob[j]) [+ MIN_FLOAT where cand_ids[i] == cand_ids[j], i != j]) * inv_t
    //   EPI 6: per (row, 256-column tile) the running max and sum-exp -> sm_part_m / sm_part_l [tile_n][M]; s_ii -> sm_pos[i]
    //   EPI 7: C[i][j] = sm_w[i] * (exp(s_ij - sm_lse[i]) - delta_ij) * inv_t * sm_alpha      (the gradient wrt the raw scores)
    const float* sm_cand_prob; const int64_t* sm_cand_ids; float sm_inv_t;
    float* sm_part_m; float* sm_part_l; float* sm_pos;
    const float* sm_lse; const float* sm_w; float sm_alpha;
};

// ---------------------------------------------------------------------------------------------------------------------
// "f16x2" operand mode (H2 = 1 in the register-split kernels; round 4): every fp32 value, multiplied by a power-of-two scale s of
// its TENSOR, is carried as two fp16 terms  x s = h + l,  h = f16_rn(x s),  l = f16_rn(x s - h)  (22 significant bits; the
// subtraction is exact), and a product is formed as  h_a l_b + l_a h_b + h_a h_b  -- THREE MFMAs (v_mfma_f32_32x32x16_f16, same
// shape and rate as the bf16 one) where the bf16x3 mode needs six; the dropped l_a l_b is below 2^-22 |ab|.  fp16 has 5 exponent
// bits, hence the scale: s = 2^(140 - e) with e the biased exponent of the tensor's largest magnitude (an `amax record`: one
// uint32 holding max |x| as float bits, maintained by the tensor's PRODUCER with atomicMax -- dr_h2_amax, K4, the tower tail --
// or an upper bound of it), so that |x s| < 2^14: a factor 4 below fp16's largest finite value (a bound that is stale by less
// than that cannot overflow; beyond it FP16_OVFL, set at kernel entry, clamps instead of producing inf).  Elements more than
// 2^-17 below the tensor's largest carry fewer than 22 bits in l (subnormal); their absolute error is 2^-39 of the largest.
// The epilogue multiplies the accumulator by 1 / (s_a s_b) -- exact.  Accuracy against fp64: tests/test_gpu_h2_gemm.py.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void h2_scale_of(uint32_t amax_bits, float& s, float& inv) {
    int e = (int)((amax_bits >> 23) & 0xffu);                           // max |x| < 2^(e - 126)
    e = e < 20 ? 20 : (e > 250 ? 250 : e);                              // (all-zero / denormal tensors: any scale works)
    s = __uint_as_float((uint32_t)(267 - e) << 23);                     // 2^(140 - e)
    inv = __uint_as_float((uint32_t)(e - 13) << 23);                    // 2^(e - 140)
}
__device__ __forceinline__ void h2_mode_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }   // MODE.FP16_OVFL: clamp, no inf

// h2_occ.hip: the 16-wave ("occupancy") form of the f16x2 NT GEMM for the plain epilogues (bias / ReLU, ReLU' mask, accumulate).
// Returns DR_OK after launching, or 1 when the shape / epilogue is not its domain (the caller then runs the 8-wave kernel).
int occ_nt_launch(const RsArgs& g, hipStream_t stream);

}  // namespace drrs
