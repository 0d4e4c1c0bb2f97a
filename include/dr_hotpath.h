/*
 * dr_hotpath.h — C-ABI of the MI355X-native embedding-lookup + feature-interaction hot path.
 *
 * The reference (LongmaoTeamTf/deep_recommenders) is pure Python over TensorFlow and has no
 * FFI of its own; the drop-in boundary is therefore the set of TF-op clusters
 * its Python classes invoke.  Each entry point below replaces one such cluster and cites the
 * reference call site it stands behind (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless marked [host]; the caller allocates all
 *     inputs and outputs; the library holds no state between calls;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and stream-ordered,
 *     no hidden synchronisation, re-entrant across streams;
 *   - return value: DR_OK (0) or a negative DR_E* code; nothing throws across the boundary;
 *   - ids are int64, -1 = "missing" (TF drops -1 / "" before lookup);
 *   - all floating point is IEEE fp32 ("f32" in bench.py's dtype field).
 *
 * Layout of categorical inputs ("columns"): a batch is an int64 matrix ids[B, C]; field f owns the
 * contiguous columns [col_start[f], col_start[f+1]) (its bag, padded with -1), C = col_start[F].
 * All F embedding tables live in ONE fp32 slab table[R, D]; field f's rows start at row_base[f].
 * The first-order ("linear"/"indicator") weights use the same row numbering: lin_w[R].
 */
#ifndef DR_HOTPATH_H
#define DR_HOTPATH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_OK 0
#define DR_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported D)     */
#define DR_ELAUNCH (-2)  /* hipLaunch / runtime error (hipGetLastError() != hipSuccess)  */
#define DR_ESHAPE (-3)   /* shape contract violated (e.g. x0 / x dim mismatch, k > n)    */

typedef void* dr_stream_t; /* hipStream_t */

/* Library self-description: returns the gfx arch string the kernels were compiled for. */
const char* dr_version(void);

/* How the fp32 GEMMs of the tower and cross layers (dr_linear_fwd / _bwd_dx / _bwd_dx_fm / _bwd_dw, dr_cross_fwd) and the
 * top-K score kernels (dr_scores_nt, the scan inside dr_topk_mips -- always the same mode, so equal candidates tie) form
 * their products.  Inputs, outputs and accumulators are fp32 in both modes; the reference's own ops are tf.matmul on
 * fp32 tensors (keras/models/ranking/deepfm.py:30-34, dcn.py:81-88), compared at 1e-5 on the loss.
 *   DR_GEMM_BF16X3      (default) every operand value x is split exactly into three bf16 terms x0 + x1 + x2 and a * b is
 *                       accumulated as a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 on v_mfma_f32_32x32x16_bf16 (each bf16
 *                       product is exact in the fp32 accumulator; the dropped terms are < 2^-24 |ab|).  Measured error
 *                       against an fp64 reference is equal to or below the native path's (tests/test_gpu_kernels.py).
 *   DR_GEMM_NATIVE_F32  v_mfma_f32_32x32x2_f32 products.
 * dr_set_gemm_mode returns the previous mode (DR_EINVAL for an unknown one); the setting is process-wide and read at
 * launch time.  Environment override of the initial value: DR_GEMM_MODE=native | bf16x3. */
#define DR_GEMM_BF16X3 0
#define DR_GEMM_NATIVE_F32 1
int32_t dr_set_gemm_mode(int32_t mode);
int32_t dr_get_gemm_mode(void);

/* The operand split of the register-split GEMMs under DR_GEMM_BF16X3 (the dr_bf3_* / dr_h2_* families below): how many low-precision
 * terms carry one fp32 operand value.
 *   DR_GEMM_SPLIT_F16X2   (default) two fp16 terms of x * 2^k, three matrix instructions per fragment pair (dr_h2_*; 22 significant
 *                         bits per operand, whole-GEMM error against fp64 at or below the six-product split's: DESIGN.md §3)
 *   DR_GEMM_SPLIT_BF16X3  three bf16 terms, six matrix instructions per fragment pair (dr_bf3_*)
 * The dr_bf3_* and dr_h2_* entry points are explicit and ignore this setting; it is the ONE switch the callers that choose between
 * them consult -- the engines of the host package at construction, and dr_topk_mips (the exact scan picks dr_h2_* or dr_bf3_*) at
 * every call.  dr_set_gemm_split returns the previous value (DR_EINVAL for an unknown one); process-wide.  Environment override of
 * the initial value, parsed here and nowhere else: DR_GEMM_SPLIT=f16x2 (or unset) | anything else = bf16x3. */
#define DR_GEMM_SPLIT_BF16X3 0
#define DR_GEMM_SPLIT_F16X2 1
int32_t dr_set_gemm_split(int32_t split);
int32_t dr_get_gemm_split(void);

/* ------------------------------------------------------------------------------------------
 * K1  hash-bucket column:  id = int64( FarmHash::Fingerprint64(as_string(key)) mod N )
 * replaces [TF] categorical_column_with_hash_bucket ->
 *   string_to_hash_bucket_fast(as_string(x), N)
 * reference call sites: examples/train_fm_on_movielens_estimator.py:12-13,20-21;
 *   examples/train_deepfm_on_movielens_keras.py:13-14,21-22; tests/keras/test_fm.py:70-73
 * keys[B, C] int64; col_buckets[C] uint64: N for a hashed column, 0 = pass the value through
 * unchanged (column already holds ids, e.g. from dr_vocab_lookup_*).  key == -1 -> id -1.
 * Bit-exact integer path.
 * ---------------------------------------------------------------------------------------- */
int dr_hash_bucket_i64(const int64_t* keys, int64_t B, int32_t C, const uint64_t* col_buckets,
                       int64_t* ids_out, dr_stream_t stream);

/* Same for byte strings in CSR form (bytes + offsets[n+1]); "" -> id -1. One bucket count. */
/* ids [B, F] int64 -> out [F][B] int32, field-major (what dr_bf3_wgrad_emb reads: one 128-byte line per 32 examples of a field) */
int dr_ids_transpose_i32(const int64_t* ids, int64_t B, int32_t F, int32_t* out, dr_stream_t stream);
int dr_hash_bucket_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n,
                         uint64_t num_buckets, int64_t* ids_out, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  vocabulary-list column: id = position of key in vocab, OOV -> -1 (default_value=-1,
 * num_oov_buckets=0).  replaces [TF] categorical_column_with_vocabulary_list
 * reference call sites: examples/train_fm_on_movielens_estimator.py:14-19,22-23
 * ---------------------------------------------------------------------------------------- */
int dr_vocab_lookup_i64(const int64_t* keys, int64_t n, const int64_t* vocab, int32_t vocab_len,
                        int64_t* ids_out, dr_stream_t stream);
int dr_vocab_lookup_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n,
                          const uint8_t* vocab_bytes, const int64_t* vocab_offsets,
                          int32_t vocab_len, int64_t* ids_out, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K3 (+K5, K6 fused)  embedding gather + mean-pool forward.
 * replaces, in ONE pass: [TF] safe_embedding_lookup_sparse(combiner='mean') x F
 *   (keras/models/ranking/fm.py:48-51,57-61; deepfm.py:25-28,39-43; estimator/.../fm.py:48-52),
 *   tf.stack / tf.concat (fm.py:62; deepfm.py:44-45), the first-order term
 *   DenseFeatures(indicator)->Dense(1) / linear_model (fm.py:16-20,47,55; estimator fm.py:43-44)
 *   and the FM second-order term (fm.py:28-35; estimator fm.py:22-26).
 *
 *   concat[b, f*D + d] = mean over valid ids of field f of table[row_base[f]+id, d]  (0 if empty)
 *   sum_x[b, d]        = sum_f concat[b, f*D+d]                                   (optional)
 *   fm_logit[b]        = lin_bias[0] + sum over valid ids lin_w[row] + 0.5*sum_d(sum_x^2 - sum_f x^2)
 *
 * col_start[F+1] (device, int32) may be NULL when every field is single-valued (then C == F).
 * concat has leading dimension ld_concat (>= F*D) so the caller can append dense features.
 * lin_w may be NULL (no first-order term); lin_bias is a 1-element device array (the trained bias) or
 * NULL; sum_x / fm_logit may be NULL (pure lookup).
 * D must be a multiple of 4, 4 <= D <= 256.
 * ---------------------------------------------------------------------------------------- */
int dr_emb_pool_fwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                    const int64_t* row_base, const float* table, int32_t D,
                    const float* lin_w, const float* lin_bias,
                    float* concat, int64_t ld_concat, float* sum_x, float* fm_logit,
                    dr_stream_t stream);

/* Same, with flags.  DR_POOL_FIRST_ORDER_ONLY: fm_logit = lin_bias + sum w (no second-order term) -- the "wide" logit of
 * WDL (`tf.feature_column.linear_model`, estimator/models/ranking/wide_and_deep.py:30-32 of the reference). */
#define DR_POOL_FIRST_ORDER_ONLY 1
int dr_emb_pool_fwd_ex(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                       const int64_t* row_base, const float* table, int32_t D, const float* lin_w,
                       const float* lin_bias, float* concat, int64_t ld_concat, float* sum_x,
                       float* fm_logit, int32_t flags, dr_stream_t stream);
/* Per-field first-order outputs (FNN, estimator/models/ranking/fnn.py:53-64: a bias-free Dense(1) over every indicator
 * column's multi-hot input, concatenated):  out[b][f] = sum over the bag of field f of lin_w[row_base[f] + id]  (ids < 0
 * skipped).  bwd: dst_lin[row] += scale * d_out[b][f] for every id of the bag (fp32 atomics). */
int dr_lin_fields_fwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                      const int64_t* row_base, const float* lin_w, float* out, int64_t ld_out,
                      dr_stream_t stream);
int dr_lin_fields_bwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                      const int64_t* row_base, const float* d_out, int64_t ld_dout, float scale,
                      float* dst_lin, dr_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * K4  transposed scatter-add backward of K3 (autodiff of the lookup: IndexedSlices ->
 * unsorted_segment_sum into the variable; implicit in optimizer.minimize / model.fit:
 * examples/train_fm_on_movielens_estimator.py:51-52, examples/train_deepfm_on_movielens_keras.py:49).
 *
 *   g[b,f,:]   = d_concat[b, f*D:(f+1)*D] + d_fm_logit[b] * (sum_x[b,:] - concat[b, f*D:(f+1)*D])
 *   for every valid id of (b,f):  dst_table[row,:] += scale * g[b,f,:] / bag_count(b,f)
 *                                  dst_lin[row]    += scale * d_fm_logit[b]
 *   dst_bias[0] += scale * sum_b d_fm_logit[b]                       (dst_bias may be NULL)
 *
 * dst_table / dst_lin are either gradient buffers (scale = 1) or the parameters themselves
 * (scale = -learning_rate: fused SGD step, no gradient materialised).  d_concat may be NULL
 * (FM only), d_fm_logit may be NULL (pure lookup backward), dst_lin may be NULL.  concat == sum_x == NULL with d_fm_logit
 * given: the logit was first-order only (DR_POOL_FIRST_ORDER_ONLY), its gradient reaches dst_lin / dst_bias only.
 * Accumulation uses hardware fp32 atomics; rows touched once are bit-exact.
 * ---------------------------------------------------------------------------------------- */
int dr_emb_pool_bwd(const int64_t* ids, int64_t B, int32_t F, int32_t C, const int32_t* col_start,
                    const int64_t* row_base, int32_t D,
                    const float* d_concat, int64_t ld_dconcat,
                    const float* concat, int64_t ld_concat, const float* sum_x,
                    const float* d_fm_logit, float scale,
                    float* dst_table, float* dst_lin, float* dst_bias, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K4, deterministic form (single-valued fields, C == F <= 64): build the slot plan once per batch
 * (dr_emb_sort_slots -- depends only on ids, so it can run on a side stream): flag the slots whose row is unique in
 * the batch and group the slots of shared rows by row; dr_emb_pool_bwd_sorted then applies
 *     dst_table[row,:] += scale * sum_slots grad[b, f*D:(f+1)*D]      dst_lin[row] += scale * sum_slots d_fm_logit[b]
 *     dst_bias[0]      += scale * sum_b d_fm_logit[b]
 * with (a) a streaming pass in example order doing ONE plain read-modify-write per unique row, and (b) a
 * segmented pass in sorted order for rows hit by several slots.  Per-slot gradient = grad[b, f*D:(f+1)*D]
 * (+ d_fm_logit[b] * (sum_x[b,:] - concat[b, f*D:(f+1)*D]) when concat / sum_x are given, as in
 * dr_emb_pool_bwd; pass NULL when `grad` already holds it).  slot_lin_grad[n] (may be NULL) replaces d_fm_logit[b]
 * as the per-slot first-order gradient -- the owner side of the sharded exchange receives gradients per slot.  Rows hit by <= 32 slots are bit-reproducible; hotter rows are cut into
 * pieces that combine with fp32 atomics.
 * The plan (csrc/emb_plan.hip; hand-written, no library sort): unique_flags[n] uint8 (indexed by slot = b*F + f);
 * dup_count[2] int32 = {number of work-list entries, length L of the sorted arrays}; sorted_rows[0..L) int64 ascending with
 * sorted_slots[0..L) int32 ordered by slot inside a row; dup_heads[n] int32 = the duplicate pass's work list (sorted positions
 * heading a piece of a multiply-hit row).  L is either the number of slots on shared rows (claim path: an open-addressed
 * table claims every row, the few shared-row slots are sorted by one block in LDS -- uniform ids) or n with missing ids
 * carrying num_rows at the end (radix path: shared-row slots beyond the LDS list, n > 2^24 or num_rows >= 2^31 - 1); the
 * choice is made on the device, the host never waits.  Arrays must hold n entries either way.  workspace >=
 * dr_emb_sort_workspace_bytes(n) bytes.  dr_emb_plan_set_small_limit(limit): largest shared-row list the LDS sort takes
 * (default / maximum 16384; 0 forces the radix path whenever a row is shared) -- returns the previous value, process-wide.
 * ---------------------------------------------------------------------------------------- */
int64_t dr_emb_sort_workspace_bytes(int64_t n);
int32_t dr_emb_plan_set_small_limit(int32_t limit);
int dr_emb_sort_slots(const int64_t* ids, int64_t B, int32_t F, const int64_t* row_base, int64_t num_rows,
                      int64_t* sorted_rows, int32_t* sorted_slots, uint8_t* unique_flags,
                      int32_t* dup_heads, int32_t* dup_count, void* workspace, int64_t workspace_bytes,
                      dr_stream_t stream);
/* Round 6 -- K1, the field-major ids and the slot plan of one batch in ONE chain (the engines' next-batch prefetch):
 * dr_hash_bucket_i64(keys [B, F], col_buckets -> ids_out [B, F]), dr_ids_transpose_i32(ids_out -> ids_t_out [F, B] int32; ids_t_out may be
 * NULL) and dr_emb_sort_slots(ids_out, ...), bit for bit, with hash, transpose, composite keys and partition histogram as one kernel when
 * F <= 64 and the geometry fits the composite key (otherwise the three entry points are called one after the other).  Replaces the same
 * reference lines as K1 ([TF] categorical_column_with_hash_bucket, examples/train_fm_on_movielens_estimator.py:12-13,20-21) and the
 * plan (the grouping half of unsorted_segment_sum in the autodiff of keras/models/ranking/fm.py:57-61). */
int dr_hash_sort_slots(const int64_t* keys, int64_t B, int32_t F, const uint64_t* col_buckets, int64_t* ids_out,
                       int32_t* ids_t_out, const int64_t* row_base, int64_t num_rows, int64_t* sorted_rows,
                       int32_t* sorted_slots, uint8_t* unique_flags, int32_t* dup_heads, int32_t* dup_count, void* workspace,
                       int64_t workspace_bytes, dr_stream_t stream);
int dr_emb_pool_bwd_sorted(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                           const int32_t* sorted_slots, const uint8_t* unique_flags,
                           const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                           int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                           const float* concat, int64_t ld_concat, const float* sum_x,
                           const float* d_fm_logit, const float* slot_lin_grad, float scale,
                           float* dst_table, float* dst_lin, float* dst_bias, float* x_sorted, dr_stream_t stream);
/* The FM term (sum_x, d_fm_logit given) needs x[b, f, :] of every slot: for a slot that owns its row it is the row's own value
 * (read by the update anyway); for slots that SHARE a row it comes from `concat` if the forward stored it, else from
 * x_sorted [B * F, D], where dr_emb_snapshot_sorted_rows has placed -- before the update starts -- the row of every work-list
 * head at the head's sorted position (one row per piece of a shared row: 5.6 K rows for uniform ids at config 3)
 * (round 3: the fused first layer no longer stores concat, the wgrad gathers its operand from the tables -- dr_bf3_wgrad_emb).
 * (With the deterministic SGD update below the snapshot is not needed: x is then read from the table inside the update -- no row
 * is written before its reader has it.  The Adam variant and DR_K4_DETERMINISTIC=0 still read the snapshot.)
 * x_sorted also makes the SGD update of HOT rows (more than 32 slots of the batch, cut into pieces of 32 that run in parallel)
 * deterministic: with x_sorted == NULL the pieces combine with fp32 atomics (order-dependent rounding); with x_sorted given each
 * piece parks its sum in its own row of x_sorted (overwriting the snapshot it has consumed) and a second small launch inside
 * the call adds a row's pieces in sorted order and updates the row once -- the same batch then gives the same bits, whatever
 * the id distribution.  (The Adam variant always sums a row's slots in one lane group and never used atomics.) */
/* dr_emb_pool_bwd_sorted in two halves (parts: 1 = the update kernel, 2 = the ordered combination of hot rows' parked pieces, 3 =
 * both = dr_emb_pool_bwd_sorted): for callers that time or overlap the halves separately.  Part 2 must follow part 1 on the same
 * stream before the tables are read; it is a no-op without x_sorted. */
int dr_emb_pool_bwd_sorted_parts(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                 const int32_t* sorted_slots, const uint8_t* unique_flags,
                                 const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                 int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                 const float* concat, int64_t ld_concat, const float* sum_x,
                                 const float* d_fm_logit, const float* slot_lin_grad, float scale,
                                 float* dst_table, float* dst_lin, float* dst_bias, float* x_sorted, int32_t parts,
                                 dr_stream_t stream);
/* parts | 4: first-order weights of rows unique in the batch are NOT updated by the call; dr_emb_lin_update_unique applies exactly that
 * part (dst_lin[row] += scale * gradient for every slot whose unique flag is set), on any stream between the head's backward and
 * the next forward.  Why: a random 4-byte read-modify-write fetches a 128-byte line -- 0.27 GB of K4's 1.64 GB at config 3 -- and
 * beside a matrix-bound GEMM that traffic is free. */
int dr_emb_lin_update_unique(const int64_t* ids, const uint8_t* unique_flags, int64_t B, int32_t F, const int64_t* row_base,
                             const float* d_fm_logit, const float* slot_lin_grad, float scale, float* dst_lin, dr_stream_t stream);
/* dr_emb_pool_bwd_sorted_parts with lin_old_t [F, B] (field-major, may be NULL): the first-order weights as the forward of THIS step
 * read them (dr_bf3_emb_linear_fwd_lv).  A row unique in the batch then gets dst_lin[row] = lin_old + scale * g as ONE write instead
 * of a read-modify-write (K4 is bound by 128-byte line operations; this removes one of eight per slot).  Valid only if nothing
 * wrote dst_lin since that forward; ignored with slot_lin_grad. */
int dr_emb_pool_bwd_sorted_ex(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                              const int32_t* sorted_slots, const uint8_t* unique_flags,
                              const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                              int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                              const float* concat, int64_t ld_concat, const float* sum_x,
                              const float* d_fm_logit, const float* slot_lin_grad, float scale,
                              float* dst_table, float* dst_lin, float* dst_bias, float* x_sorted,
                              const float* lin_old_t, int32_t parts, uint32_t* table_amax, dr_stream_t stream);
int dr_emb_snapshot_sorted_rows(const int64_t* sorted_rows, const int32_t* dup_heads, const int32_t* dup_count,
                                const float* table, int32_t D, int64_t num_rows, float* x_sorted, dr_stream_t stream);

/* SURVEY.md section 8f rank 1 -- the optimizer of the reference's own examples (tf.train.AdamOptimizer(0.01),
 * examples/train_fm_on_movielens_estimator.py:51-52; tf.keras.optimizers.Adam(), examples/train_deepfm_on_movielens_keras.py:44)
 * fused into the sorted K4: every row the batch touches gets ONE Adam update from the SUM of its slots' gradients
 *   m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  w -= lr_t m / (sqrt(v) + eps),   lr_t = lr sqrt(1-b2^t) / (1-b1^t)
 * (lr_t computed by the caller).  `grad` is the gradient of the mean loss (no scale).  m_table / v_table [R, D] and
 * m_lin / v_lin [R] are the moment slabs.  Rows the batch does not touch keep their moments (row-wise "lazy" Adam):
 * equal to TF's Adam on the first step and for rows touched on every step; TF decays m/v of the whole variable each step
 * (SURVEY App. B15), which is a full pass over the 66 GB table at config 3.  The first-order bias is a dense parameter:
 * dr_adam_step.  Deterministic (a row's slots are summed by one lane group in sorted order). */
/* (round 4) m_lin / v_lin may be the two columns of ONE [R, 2] array -- pass v_lin == m_lin + 1 -- and are then addressed with a row
 * stride of 2: a row's first-order moments share a cache line (one line operation to read, one to write, instead of two each).  The
 * same convention holds for dr_adam_catchup_rows. */
int dr_emb_pool_bwd_sorted_adam(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                const int32_t* sorted_slots, const uint8_t* unique_flags,
                                const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                const float* concat, int64_t ld_concat, const float* sum_x,
                                const float* d_fm_logit, const float* slot_lin_grad, float lr_t, float beta1,
                                float beta2, float eps, float* table, float* m_table, float* v_table,
                                float* lin_w, float* m_lin, float* v_lin, float* x_sorted, dr_stream_t stream);
/* ... with lin_old_t as in dr_emb_pool_bwd_sorted_ex */
int dr_emb_pool_bwd_sorted_adam_ex(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                const int32_t* sorted_slots, const uint8_t* unique_flags,
                                const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                const float* concat, int64_t ld_concat, const float* sum_x,
                                const float* d_fm_logit, const float* slot_lin_grad, float lr_t, float beta1,
                                float beta2, float eps, float* table, float* m_table, float* v_table,
                                float* lin_w, float* m_lin, float* v_lin, float* x_sorted, const float* lin_old_t, uint32_t* table_amax, dr_stream_t stream);
/* Dense Adam step (same formula) over a flat parameter buffer; grad is multiplied by grad_scale first. */
/* TF's NON-lazy sparse Adam, evaluated lazily (examples/train_fm_on_movielens_estimator.py:51-52: tf.train.AdamOptimizer decays
 * m / v of the WHOLE variable and moves every row on every step).  row_step[R] int32 (zero-initialised) counts
 * the optimizer steps a row has received; dr_adam_catchup_rows brings the rows named by ids[n] (field of slot p = p % F, -1 =
 * missing) to `upto` steps by replaying their missed decay-only steps  m *= b1 ; v *= b2 ; w -= lr_s m / (sqrt(v) + eps)
 * (lr_s = lr sqrt(1 - b2^s) / (1 - b1^s)) in order, and stamps them `stamp`.  Before the forward of step t: upto = t - 1,
 * stamp = t, then dr_emb_pool_bwd_sorted_adam applies step t -- every value the model reads equals TF's dense update.  To
 * export the tables: upto = stamp = steps taken, over all rows.  One replayer per row and call (atomic exchange of the stamp). */
int dr_adam_catchup_rows(const int64_t* ids, int64_t n, int32_t F, const int64_t* row_base, int32_t D, float* table,
                         float* m_table, float* v_table, float* lin_w, float* m_lin, float* v_lin, int32_t* row_step,
                         int32_t upto, int32_t stamp, float lr, float beta1, float beta2, float eps, dr_stream_t stream);
int dr_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                 float beta2, float eps, float grad_scale, dr_stream_t stream);

/* Dense FTRL-Proximal step in TensorFlow's formulation -- tf.train.FtrlOptimizer(0.01, l1_regularization_strength=0.5), the
 * optimizer of WDL's wide part in examples/train_wdl_on_movielens_estimator.py:66-70 (TF defaults: learning_rate_power -0.5,
 * accumulators initialised to 0.1, l2 0):
 *   accum' = accum + g^2 ;  linear += g - (accum'^-p - accum^-p) / lr * w ;  w = |linear| > l1 ? (sign(linear) l1 - linear) /
 *   (accum'^-p / lr + 2 l2) : 0.   grad is multiplied by grad_scale first. */
int dr_ftrl_step(float* param, const float* grad, float* accum, float* linear, int64_t n, float lr,
                 float lr_power, float l1, float l2, float grad_scale, dr_stream_t stream);



/* ------------------------------------------------------------------------------------------
 * K6  stand-alone FM second-order term on a caller-provided [B, F, D] tensor
 * replaces keras FM.call (keras/models/ranking/fm.py:28-35) and estimator fm(x)
 * (estimator/models/feature_interaction/fm.py:10-26).
 *   out[b] = 0.5 * sum_d( (sum_f x)^2 - sum_f x^2 )      dx = d_out[b] * (sum_f x - x)
 * ---------------------------------------------------------------------------------------- */
int dr_fm2_fwd(const float* x, int64_t B, int32_t F, int32_t D, float* out, dr_stream_t stream);
int dr_fm2_bwd(const float* x, const float* d_out, int64_t B, int32_t F, int32_t D, float* dx,
               dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7  Dense layer on fp32 MFMA (v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered accumulation).
 * replaces tf.keras.layers.Dense / tf.layers.dense (keras/models/ranking/deepfm.py:30-34;
 * estimator/models/feature_interaction/dnn.py:17-29) and, with N = 1, the FM layer's first-order
 * Dense(1) on an explicit multi-hot matrix (keras/models/ranking/fm.py:16-20,26,37)  [K5].
 *
 *   fwd     y[M,N]  = act(x[M,K] @ W[K,N] + b[N])           act: 0 linear, 1 relu; b may be NULL
 *   bwd_dx  dx[M,K] = (dy[M,N] @ W^T) * (relu_src > 0)  (+ dx if accumulate)
 *           dy is the PRE-activation gradient of this layer; relu_src[M,K] (may be NULL) is this
 *           layer's input when that input is itself a relu output — so the product is directly the
 *           pre-activation gradient of the layer below (relu' folded into the producer's epilogue).
 *   bwd_dw  dstW[K,N] += scale * x^T @ dy ;  dstb[N] += scale * colsum(dy)
 *           split over the batch dimension: dst is a zeroed gradient buffer (scale = 1) or the
 *           parameter itself (scale = -lr, fused SGD).  With a workspace of
 *           dr_linear_bwd_dw_workspace_bytes() the split partials are combined by a deterministic
 *           reduce kernel; with workspace == NULL they combine with fp32 atomics.  dstb may be NULL.
 * All matrices row-major with explicit leading dimensions; 16-byte aligned bases with pitches that are
 * multiples of 4 floats take the float4 load path, anything else a scalar-load path.
 * ---------------------------------------------------------------------------------------- */
int dr_linear_fwd(const float* x, int64_t ld_x, const float* W, int64_t ld_w, const float* b,
                  int64_t M, int32_t K, int32_t N, int32_t act, float* y, int64_t ld_y,
                  dr_stream_t stream);
int dr_linear_bwd_dx(const float* dy, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M,
                     int32_t K, int32_t N, const float* relu_src, int64_t ld_relu_src,
                     int32_t accumulate, float* dx, int64_t ld_dx, dr_stream_t stream);
/* dr_linear_fwd_splitk: y += x W for few output tiles and a long reduction (the in-batch softmax's dq = G c of
 * keras/models/retrieval/sbcnm.py:120-163's backward: G [B, B], c [B, 128]): the reduction is split over the grid into `workspace`
 * (dr_linear_fwd_splitk_workspace_bytes) and the slices are summed in a fixed order.  Accumulates into y (zero it for a plain
 * product). */
int64_t dr_linear_fwd_splitk_workspace_bytes(int64_t M, int32_t K, int32_t N);
int dr_linear_fwd_splitk(const float* x, int64_t ld_x, const float* W, int64_t ld_w, int64_t M, int32_t K, int32_t N,
                         float* y, int64_t ld_y, float* workspace, int64_t workspace_bytes, dr_stream_t stream);

/* dgrad of the first tower layer with the FM second-order gradient folded into the epilogue:
 *   dx[m, j] = (dy @ W^T)[m, j] + d_fm_logit[m] * (sum_x[m, j % D] - concat[m, j])   for j < FD (= F*D)
 * (autodiff of keras/models/ranking/fm.py:28-35 + deepfm.py:46 in one stream), so K4 reads one gradient. */
int dr_linear_bwd_dx_fm(const float* dy, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M,
                        int32_t K, int32_t N, const float* d_fm_logit, const float* sum_x,
                        const float* concat, int64_t ld_concat, int32_t D, int32_t FD, float* dx,
                        int64_t ld_dx, dr_stream_t stream);
int64_t dr_linear_bwd_dw_workspace_bytes(int64_t M, int32_t K, int32_t N);
int dr_linear_bwd_dw(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, int64_t M,
                     int32_t K, int32_t N, float scale, float* dstW, int64_t ld_w, float* dstb,
                     float* workspace, int64_t workspace_bytes, dr_stream_t stream);
/* Fused backward of a NARROW layer (N <= 32, e.g. the Dense(32) that precedes Dense(1) in the reference's
 * towers, keras/models/ranking/deepfm.py:30-34, estimator/models/feature_interaction/dnn.py:17-29): one pass over x
 * produces  dx = (dy @ W^T) * (x > 0 if relu_mask)  and the fused SGD step  dstW += scale * x^T dy,
 * dstb += scale * colsum(dy)  (dx uses the pre-update W even when dstW == W).  Deterministic.
 * Domain: N <= 32, K in {128, 256, 512}, M a multiple of 32, x/dx rows 4*(K/128)-byte aligned; anything else returns
 * DR_ESHAPE and the caller uses dr_linear_bwd_dx + dr_linear_bwd_dw. */
int64_t dr_linear_bwd_narrow_workspace_bytes(int64_t M, int32_t K, int32_t N);
int dr_linear_bwd_narrow(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W,
                         int64_t ld_w, int64_t M, int32_t K, int32_t N, int32_t relu_mask, float scale,
                         float* dstW, int64_t ld_dstw, float* dstb, float* dx, int64_t ld_dx,
                         void* workspace, int64_t workspace_bytes, dr_stream_t stream);

/* Fused tower head: the last hidden layer (H <= 32 units), the Dense(1) output that follows it, the FM logit, the loss,
 * and the backward of the Dense(1) -- i.e. keras/models/ranking/deepfm.py:30-34,41-47 `Dense(32, relu)`, `Dense(1)`,
 * `sigmoid(fm + dnn)` plus the loss of examples/train_deepfm_on_movielens_keras.py:43 /
 * examples/train_deepfm_on_movielens_estimator.py:47, in one GEMM launch:
 *   h      = act(x @ W1 + b1)                       [M, H]   (written to h_out if not NULL)
 *   logit  = h @ w2 + b2 + extra_logit              [M]      (w2[n] at w2[n * ld_w2]; extra_logit may be NULL)
 *   prob, loss, d_logit = BCE(logit, labels, loss_mode) as dr_bce_fwd_bwd  (d_logit = d mean-loss / d logit)
 *   d_h    = d_logit (x) w2 * act'(h)               [M, H]   (gradient for dr_linear_bwd_narrow / _bwd_dx)
 *   dst_w2 += scale * h^T d_logit ;  dst_b2 += scale * sum(d_logit)
 *            dst_* = the parameters themselves with scale = -lr (fused SGD; d_h always uses the pre-update w2), or
 *            gradient buffers with scale = 1 (data-parallel: all-reduce, then apply); NULL / scale = 0: skipped
 * n_total (0 = M): the number of examples the loss is a mean over, when these M rows are one slice of a larger batch
 * (micro-batches): loss_out = sum(l) / n_total, d_logit = dl/dlogit / n_total.
 * Deterministic.  H > 32 returns DR_ESHAPE (use dr_linear_fwd x2 + dr_bce_fwd_bwd + dr_linear_bwd_*). */
int64_t dr_tower_head_workspace_bytes(int64_t M);
int dr_tower_head_fwd_bwd(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1,
                          int64_t M, int64_t n_total, int32_t K, int32_t H, int32_t act, const float* w2,
                          int64_t ld_w2, const float* b2, const float* extra_logit, const float* labels,
                          int32_t loss_mode, float scale, float* dst_w2, int64_t ld_dst_w2, float* dst_b2, float* h_out, int64_t ld_h, float* prob, float* d_logit, float* d_h, int64_t ld_dh,
                          float* loss_out, void* workspace, int64_t workspace_bytes, dr_stream_t stream);
/* The whole tower tail in ONE pass over x (round 5): dr_tower_head_fwd_bwd (act = relu) and dr_linear_bwd_narrow (relu_mask = 1) of the
 * SAME layer W1 [K, H] -- in the reference's towers the last hidden layer, Dense(H <= 32, relu), is at once the head's first factor and
 * the layer whose backward follows (keras/models/ranking/deepfm.py:30-34,41-47; estimator/models/feature_interaction/dnn.py:17-29):
 *   h = relu(x W1 + b1); logit = h w2 + b2 + extra_logit; prob, loss, d_logit = BCE(logit, labels, loss_mode) (mean over n_total, 0 = M);
 *   d_h = d_logit (x) w2 * (h > 0)  (written if d_h != NULL);   dx = (d_h W1^T) * (x > 0);
 *   dst_w1 += scale * x^T d_h;  dst_b1 += scale * colsum(d_h);  dst_w2 += scale * h^T d_logit;  dst_b2 += scale * sum(d_logit).
 * Both products use the pre-update W1 / w2 even when dst_* are the parameters themselves (scale = -lr: fused SGD); with gradient
 * buffers scale = 1.  parts = 1: the one-pass kernel (prob, d_logit, d_h, dx, per-block partials), 2: the two fixed-order reduces that
 * apply the partials (may run on another stream; must finish before the next part 1 over the same workspace), 3: both.  dx_amax (may
 * be NULL): receives max |dx| as float bits -- the f16x2 GEMMs' record of dx (parts = 3: stored by the reduce launch, no reset needed;
 * part 1 alone: reset and raised by the kernel, complete when part 1 is).  Deterministic.
 * Domain: H <= 32, K in {128, 256}, M a multiple of 32; anything else returns DR_ESHAPE (use the two calls it replaces).  Results equal those two calls' up to the summation order of the K-long dot products. */
int64_t dr_tower_tail_workspace_bytes(int64_t M, int32_t K);
int dr_tower_tail_fused(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1, int64_t M, int64_t n_total,
                        int32_t K, int32_t H, const float* w2, int64_t ld_w2, const float* b2, const float* extra_logit,
                        const float* labels, int32_t loss_mode, float scale, float* dst_w1, int64_t ld_dst_w1, float* dst_b1,
                        float* dst_w2, int64_t ld_dst_w2, float* dst_b2, float* prob, float* d_logit, float* d_h, int64_t ld_dh,
                        float* dx, int64_t ld_dx, float* loss_out, void* workspace, int64_t workspace_bytes, int32_t parts,
                        uint32_t* dx_amax, dr_stream_t stream);

/* dr_linear_bwd_narrow / dr_tower_head_fwd_bwd in two halves (round 4): parts = 1 the main kernel (everything the rest of the step
 * reads: dx; prob, d_logit, d_h), parts = 2 the small reduce that applies the per-block partials to the weights (and writes the
 * loss), 3 = both.  Part 2 may run on ANOTHER stream -- the engine keeps the three reduce kernels of a step off its training stream;
 * it must finish before anything reads the updated weights and before the next part 1 over the same workspace. */
int dr_linear_bwd_narrow_parts(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W,
                               int64_t ld_w, int64_t M, int32_t K, int32_t N, int32_t relu_mask, float scale,
                               float* dstW, int64_t ld_dstw, float* dstb, float* dx, int64_t ld_dx,
                               void* workspace, int64_t workspace_bytes, int32_t parts, dr_stream_t stream);
/* ... that also leaves max |dx| as float bits in dx_amax[0] (reset and rebuilt by part 1): the amax record of dx for the f16x2 GEMMs
 * (dr_h2_linear_nt / dr_h2_wgrad_emb below) */
int dr_linear_bwd_narrow_amax(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M,
                              int32_t K, int32_t N, int32_t relu_mask, float scale, float* dstW, int64_t ld_dstw, float* dstb, float* dx,
                              int64_t ld_dx, void* workspace, int64_t workspace_bytes, int32_t parts, uint32_t* dx_amax,
                              dr_stream_t stream);
int dr_tower_head_fwd_bwd_parts(const float* x, int64_t ld_x, const float* W1, int64_t ld_w1, const float* b1,
                                int64_t M, int64_t n_total, int32_t K, int32_t H, int32_t act, const float* w2,
                                int64_t ld_w2, const float* b2, const float* extra_logit, const float* labels,
                                int32_t loss_mode, float scale, float* dst_w2, int64_t ld_dst_w2, float* dst_b2, float* h_out, int64_t ld_h,
                                float* prob, float* d_logit, float* d_h, int64_t ld_dh, float* loss_out, void* workspace,
                                int64_t workspace_bytes, int32_t parts, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K8  DCN cross layer (keras/models/ranking/dcn.py:70-88):
 *   prod = x @ W + b + diag_scale * x (:81,:85-86) ;  out = x0 * prod + x (:88)
 * One GEMM with the whole combine fused in the epilogue; `prod_out` (may be NULL) saves prod for
 * the backward.  W == NULL: "prod given" in prod_out (low-rank form :83 = two dr_linear_fwd calls,
 * then this with W NULL adds bias-free combine).  x0, x, out, prod share leading dimension ld.
 * bwd (elementwise part): d_prod = d_out * x0 ; d_x0 += d_out * prod ; d_x += d_out + diag_scale*d_prod;
 * the GEMM parts are dr_linear_bwd_dx(d_prod, W, accumulate=1 -> d_x) and dr_linear_bwd_dw(x, d_prod).
 * ---------------------------------------------------------------------------------------- */
int dr_cross_fwd(const float* x0, const float* x, int64_t ld, const float* W, int64_t ld_w,
                 const float* b, float diag_scale, int64_t M, int32_t Dm, float* out,
                 float* prod_out, dr_stream_t stream);
int dr_cross_combine_bwd(const float* x0, const float* prod, const float* d_out, int64_t M,
                         int32_t Dm, int64_t ld, float diag_scale, float* d_prod, float* d_x0_accum,
                         float* d_x_accum, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K11  fused sigmoid + binary cross-entropy (+ gradient wrt the logit).
 * mode 0: tf.losses.sigmoid_cross_entropy(labels, logits)  (examples/train_fm_on_movielens_estimator.py:46)
 * mode 1: tf.losses.log_loss(labels, sigmoid(logits))      (examples/train_deepfm_on_movielens_estimator.py:47)
 * mode 2: tf.keras.losses.binary_crossentropy(labels, sigmoid(logits)) (examples/train_deepfm_on_movielens_keras.py:43)
 * logit[i] = logits[i] + logits_b[i*ld_b] (logits_b may be NULL): DeepFM's fm_outputs + dnn_outputs
 * (keras/models/ranking/deepfm.py:46) is summed here instead of in a separate pass.
 * prob[n] = sigmoid(logit) (may be NULL); d_logit[n] = d(mean loss)/d logit (may be NULL);
 * loss_out[1] = mean loss (deterministic two-stage reduction; workspace >= 1024 floats).
 * ---------------------------------------------------------------------------------------- */
int dr_bce_fwd_bwd(const float* logits, const float* logits_b, int64_t ld_b, const float* labels,
                   int64_t n, int32_t mode, float* prob, float* d_logit, float* loss_out,
                   float* workspace, dr_stream_t stream);
/* unfused forms for callers that hold probabilities (the reference models return sigmoid outputs:
 * keras/models/ranking/fm.py:64, deepfm.py:47; estimator/models/ranking/deepfm.py:43) */
int dr_sigmoid_fwd(const float* x, int64_t n, float* y, dr_stream_t stream);
int dr_sigmoid_bwd(const float* y, const float* dy, int64_t n, float* dx, dr_stream_t stream);
/* mode 1 = log_loss, mode 2 = keras binary_crossentropy, on probabilities; d_prob = d(mean loss)/dp */
int dr_bce_prob_fwd_bwd(const float* prob, const float* labels, int64_t n, int32_t mode, float* d_prob,
                        float* loss_out, float* workspace, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K9  in-batch sampled-softmax retrieval loss (Retrieval.call, keras/models/retrieval/sbcnm.py:120-151)
 *   scores = q @ c^T (:129); labels = eye (:134); optional -log(p_j) (:86), duplicate-id mask
 *   (+MIN_FLOAT = finfo(f32).min/100, :66-75), * 1/temperature (:148-149);
 *   loss = CCE(from_logits, reduction=SUM) = sum_i w_i * (logsumexp_j s_ij - s_ii)   (:100-102,151)
 * The B x B score matrix is never materialised in the forward: the MFMA GEMM's epilogue reduces each
 * 32-column group to (max, sum-exp) partials, a finalize pass combines them.
 *   row_lse[B], pos_score[B] (= s_ii) are outputs the backward reuses; loss_out[1].
 * Backward: dr_inbatch_softmax_grad_scores writes G = d_loss * w_i * (softmax_ij - delta_ij) / T into a
 *   caller-provided [B, ld_g] buffer; then dq = G @ c is dr_linear_fwd(G, c) and dc = G^T @ q is
 *   dr_linear_bwd_dw(G, q) — two plain GEMMs.  cand_prob / cand_ids / sample_weight may be NULL.
 * ---------------------------------------------------------------------------------------- */
int64_t dr_inbatch_softmax_workspace_bytes(int64_t B);
int dr_inbatch_softmax_fwd(const float* q, const float* c, int64_t B, int32_t D, const float* cand_prob,
                           const int64_t* cand_ids, const float* sample_weight, float inv_temperature,
                           float* row_lse, float* pos_score, float* loss_out, float* workspace,
                           int64_t workspace_bytes, dr_stream_t stream);
int dr_inbatch_softmax_grad_scores(const float* q, const float* c, int64_t B, int32_t D,
                                   const float* cand_prob, const int64_t* cand_ids,
                                   const float* sample_weight, float inv_temperature, const float* row_lse,
                                   float d_loss, float* G, int64_t ld_g, dr_stream_t stream);
/* Round 5: in the f16x2 operand split (dr_get_gemm_split) both score passes run on the register-split kernel with the LSE /
 * softmax-gradient epilogues (three fp16 products per fp32 product, scores never leave the tile) whenever D % 4 == 0, D <= 512,
 * B >= 256 and the workspace (256-byte aligned, dr_inbatch_softmax_workspace_bytes) is there: it also holds the candidates' two fp16
 * planes and both amax records.  dr_inbatch_softmax_grad_scores has no workspace and stays on the fp32 kernel;
 * dr_inbatch_softmax_grad_scores_ws takes one (the forward's may be reused; its contents are not needed). */
int dr_inbatch_softmax_grad_scores_ws(const float* q, const float* c, int64_t B, int32_t D,
                                      const float* cand_prob, const int64_t* cand_ids,
                                      const float* sample_weight, float inv_temperature, const float* row_lse,
                                      float d_loss, float* G, int64_t ld_g, float* workspace,
                                      int64_t workspace_bytes, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K10  exact top-K maximum-inner-product search (BruteForce.call factorized_top_k.py:316-334,
 * Streaming.call :178-260): scores = q @ cand^T, k largest per query, descending, ties -> lower candidate
 * index first ([TF] tf.math.top_k).  The [Bq, N] score matrix is never materialised: candidates are scored
 * in chunks (dr_scores_nt into `workspace`) and folded into a running sorted list per query
 * (dr_topk_select).  k <= 128.  out_scores[Bq,k] fp32, out_index[Bq,k] int64 (candidate position +
 * index_base, -1 = empty slot).  init = 1 starts a new search, init = 0 continues one (Streaming's
 * per-batch map + reduce, :201-233).  Returns DR_ESHAPE if init && k > N (the reference's
 * "input must have at least k columns", :13-23).
 * A CONTINUED search (init = 0) must be handed its batches with non-decreasing index_base, as Streaming does (its counter-based row
 * ids, :244-254): the chunk GEMM's filter epilogue keeps a candidate only if its score is strictly above the row's current k-th
 * best, so a candidate that TIES that score wins its lower-index tie only against entries of LATER batches -- with index_base going
 * backwards an equal-score, lower-index candidate could be dropped or kept depending on where a chunk boundary falls.
 * dr_topk_merge: merge two sorted lists per row, list a wins ties (Streaming's reduce; cross-rank merge).
 * ---------------------------------------------------------------------------------------- */
int dr_scores_nt(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M, int32_t N, int32_t D,
                 float* out, int64_t ld_out, dr_stream_t stream);
int dr_topk_select(const float* scores, int64_t ld, int64_t Bq, int64_t n, int32_t k, int64_t index_base,
                   int32_t init, float* out_scores, int64_t* out_index, dr_stream_t stream);
int64_t dr_topk_workspace_bytes(int64_t Bq, int64_t N, int32_t k);
int dr_topk_mips(const float* q, int64_t Bq, const float* cand, int64_t N, int32_t D, int32_t k,
                 int64_t index_base, int32_t init, float* out_scores, int64_t* out_index,
                 float* workspace, int64_t workspace_bytes, dr_stream_t stream);
/* The corpus side of an index (BruteForce.index factorized_top_k.py:275-297 / Streaming's candidates: candidates are handed over ONCE,
 * queries arrive many times).  dr_topk_index_build derives, once, what the f16x2 scan of dr_topk_mips derives from the corpus on
 * every call -- its amax record and the two fp16 planes of cand * 2^k -- into `index` (256-byte aligned, dr_topk_index_bytes(N, D)
 * bytes; D a multiple of 4, <= 512).  dr_topk_mips_indexed(q, ..., cand, index, ...) is dr_topk_mips with that work skipped: same
 * arguments, same result bit for bit (the planes and the record are the ones the call would have made); `cand` must still be the
 * corpus the index was built from -- the scan reads it instead of the index whenever it does not run in the f16x2 split
 * (dr_get_gemm_split) or falls back to the generic kernel.  The index must be rebuilt when the corpus changes. */
int64_t dr_topk_index_bytes(int64_t N, int32_t D);
int dr_topk_index_build(const float* cand, int64_t N, int32_t D, void* index, int64_t index_bytes, dr_stream_t stream);
int dr_topk_mips_indexed(const float* q, int64_t Bq, const float* cand, const void* index, int64_t N, int32_t D, int32_t k,
                         int64_t index_base, int32_t init, float* out_scores, int64_t* out_index,
                         float* workspace, int64_t workspace_bytes, dr_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * SURVEY.md section 8f rank 4 -- IVF-Flat approximate top-K, the GPU counterpart of the reference's `Faiss` index
 * (keras/models/retrieval/factorized_top_k.py:337-461: faiss.IndexIVFFlat(faiss.IndexFlatIP(d), d, nlist,
 * faiss.METRIC_INNER_PRODUCT), `searcher.nprobe = nprobe`, `searcher.search(queries, k)`).
 *   dr_ivf_pack: builds the index storage.  `order` = candidate numbers grouped by list (list l = order[list_start[l] ..
 *                list_start[l+1])), `blk_off[l]` = first 64-vector block of list l (lists padded to whole blocks),
 *                packed[(blk*D + d)*64 + lane] = component d of the vector in slot (blk, lane), packed_ids its identifier
 *                (ids[src] or src; -1 in padding slots).
 *   dr_ivf_scan: probes[q, 0..nprobe) = lists to scan for query q (any negative entry is skipped); exact inner products
 *                over those lists, top k (score descending, ties by identifier ascending); slots that find fewer than k
 *                vectors keep score -inf / index -1 (faiss returns -1 labels there too).
 * ---------------------------------------------------------------------------------------- */
/* dr_ivf_build_lists (round 4): the grouping half of `index.add` (factorized_top_k.py:374-391) -- a STABLE counting sort of the N
 * vectors by their coarse assignment: order[list_start[l] .. list_start[l+1]) = the vectors of list l in input order (so equal scores
 * tie on the lower candidate number, as in the exact search); an assignment outside [0, nlist) drops its vector.  nlist <= 8192;
 * workspace >= dr_ivf_build_workspace_bytes(N, nlist).  Integer, bit-exact. */
int64_t dr_ivf_build_workspace_bytes(int64_t N, int32_t nlist);
int dr_ivf_build_lists(const int64_t* assign, int64_t N, int32_t nlist, int64_t* order, int64_t* list_start, void* workspace,
                       int64_t workspace_bytes, dr_stream_t stream);
int dr_ivf_pack(const float* cand, int64_t N, int32_t D, const int64_t* order, const int64_t* list_start,
                const int64_t* blk_off, int32_t nlist, int64_t total_blocks, const int64_t* ids, float* packed,
                int64_t* packed_ids, dr_stream_t stream);
int dr_ivf_scan(const float* q, int64_t Bq, int32_t D, const int64_t* probes, int32_t nprobe,
                const int64_t* blk_off, const float* packed, const int64_t* packed_ids, int32_t k,
                float* out_scores, int64_t* out_index, dr_stream_t stream);
int dr_topk_merge(const float* sa, const int64_t* ia, int32_t ka, const float* sb, const int64_t* ib,
                  int32_t kb, int64_t Bq, int32_t k, float* out_scores, int64_t* out_index,
                  dr_stream_t stream);
/* small retrieval ops: positive scores sum(q*c, axis=1) (factorized_top_k.py:494-495); identifier gather
 * (:334); _take_long_axis (:26-41) / _gather_elements_along_row (sbcnm.py:15-30); _exclude's penalty
 * (:57-62); FactorizedTopK's in_top_k counting (:499-512, [TF] B14); and the sbcnm helper layers on explicit
 * logits: out = logits - log(p_j) (:86) + (dup_ij - labels_ij) * MIN_FLOAT (:66-75) + labels_ij * scale (:44) */
int dr_rowdot(const float* a, const float* b, int64_t B, int32_t D, float* out, dr_stream_t stream);
/* out[r, :] = x[r, :] * f(s[r]) for contiguous [M, D] matrices -- the row scalings of the `Faiss` index (factorized_top_k.py:370-371,
 * 450-451: faiss.normalize_L2; :372: index.train's centroid update).  mode 0: f = s.  mode 1: f = 1 / sqrt(s), rows with s == 0 unchanged
 * (s = dr_rowdot(x, x): L2 normalisation that leaves zero rows alone).  mode 2: f = 1 / s, rows with s == 0 taken from `fallback`
 * [M, D] (centroid = member sum / member count; an empty cluster keeps its previous centroid).  out may alias x. */
int dr_rows_scale(const float* x, const float* s, int32_t mode, const float* fallback, int64_t M, int32_t D, float* out,
                  dr_stream_t stream);
int dr_gather_i64(const int64_t* src, int64_t nsrc, const int64_t* idx, int64_t n, int64_t* out,
                  dr_stream_t stream);
int dr_take_along_rows_f32(const float* arr, int64_t ld, int64_t B, int32_t C, const int64_t* idx, int32_t K,
                           float* out, dr_stream_t stream);
int dr_take_along_rows_i64(const int64_t* arr, int64_t ld, int64_t B, int32_t C, const int64_t* idx,
                           int32_t K, int64_t* out, dr_stream_t stream);
int dr_topk_hits(const float* pos, const float* topk, int64_t B, int32_t K, const int32_t* ks, int32_t nk,
                 uint64_t* hits, dr_stream_t stream);
int dr_exclude_adjust(const float* scores, const int64_t* ids, int64_t B, int32_t K, const int64_t* exclude,
                      int32_t E, float* adjusted, dr_stream_t stream);
int dr_logits_adjust(const float* logits, const float* labels, int64_t B, int32_t C, const float* cand_prob,
                     const int64_t* cand_ids, float add_label_scale, float* out, dr_stream_t stream);
/* CCE(from_logits=True, reduction=SUM) on an explicit [B, C] logits / labels pair (the hard-negative branch,
 * sbcnm.py:145-151): row_loss[B] scratch, loss_out[1] = sum_i w_i * (lse_i * sum_j y_ij - sum_j y_ij s_ij / T) */
int dr_softmax_ce_rows(const float* logits, const float* labels, int64_t B, int32_t C, float inv_temperature,
                       const float* sample_weight, float* row_loss, float* loss_out, dr_stream_t stream);

/* Backward of dr_softmax_ce_rows: g[r][j] = w_r * inv_temperature * d_loss * (sum_j(labels) * softmax_j - labels[r][j]).
 * cols == NULL: out is [B, >= C] dense.  cols [B, C] (the columns HardNegativeMining kept, sbcnm.py:41-49 of the reference):
 * the gradient is scattered to out[r][cols[r][j]] of a pre-zeroed [B, ld_out] score-gradient matrix, from which
 * dq = G c and dc = G^T q follow as in the default path. */
int dr_softmax_ce_rows_bwd(const float* logits, const float* labels, int64_t B, int32_t C,
                           float inv_temperature, const float* sample_weight, float d_loss,
                           const int64_t* cols, float* out, int64_t ld_out, dr_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Row-sharded tables over N ranks (new design — the reference is single-process; the
 * exchange mirrors no reference code).  owner(id) = id % world; on the owner, field f's shard holds
 * rows_per_shard rows and id lives at local row f*rows_per_shard + id / world.
 *
 * dr_shard_bucket_ids: stable counting sort of the n = B*C (example, column) slots by owner:
 *   counts[world]   slots routed to each rank (device int64; the all-to-all split sizes)
 *   send_rows[n]    owner-local row of every slot, grouped by owner (missing ids travel as -1)
 *   pos[n]          position of slot p = b*C + c inside send_rows (the inverse map: rows come back from
 *                   the all-to-all in send order, so K3 over `pos` as ids re-assembles example order)
 *   workspace       dr_shard_bucket_workspace_bytes(n, world) bytes.  world <= 16.  Integer, bit-exact.
 * dr_rows_gather / dr_rows_scatter_add: owner side of the forward / backward exchange (packed [n, D]
 *   rows + first-order weights); scatter accumulates with fp32 atomics (dst += scale * grad).
 * dr_axpy: y += alpha * x, applies an all-reduced dense gradient.
 * ---------------------------------------------------------------------------------------- */
int64_t dr_shard_bucket_workspace_bytes(int64_t n, int32_t world);
int dr_shard_bucket_ids(const int64_t* ids, int64_t n, int32_t C, int64_t rows_per_shard, int32_t world,
                        int64_t* counts, int64_t* send_rows, int64_t* pos, int64_t* workspace,
                        dr_stream_t stream);
/* Requester-side de-duplication of the exchange (round 4): what [TF] safe_embedding_lookup_sparse's `unique` does inside the lookup
 * reached from keras/models/ranking/fm.py:57-61 -- a row that several slots of a micro-batch look up travels once each way.
 * dr_shard_dedup_slots: rep[p] = the lowest slot looking up the same row as slot p (p itself for an unshared row or a missing id), from
 *   the slot plan of the micro-batch's ids (dr_emb_sort_slots over the global rows row_base[f] + id, with the same num_rows).
 * dr_shard_bucket_ids_dedup: dr_shard_bucket_ids in which only the representatives get a send slot (counts sum to the number of
 *   distinct rows) and pos[p] = pos[rep[p]] for every other slot.  Integer, bit-exact. */
int dr_shard_dedup_slots(const int64_t* sorted_rows, const int32_t* sorted_slots, const int32_t* dup_count, int64_t n,
                         int64_t num_rows, int64_t* rep, dr_stream_t stream);
int dr_shard_bucket_ids_dedup(const int64_t* ids, const int64_t* rep, int64_t n, int32_t C, int64_t rows_per_shard,
                              int32_t world, int64_t* counts, int64_t* send_rows, int64_t* pos, int64_t* workspace,
                              dr_stream_t stream);
int dr_rows_gather(const int64_t* rows, int64_t n, const float* table, int32_t D, const float* lin_w,
                   float* out_rows, float* out_lin, dr_stream_t stream);
int dr_rows_scatter_add(const int64_t* rows, int64_t n, const float* grads, int32_t D,
                        const float* lin_grads, float scale, float* table, float* lin_w,
                        dr_stream_t stream);
int dr_axpy(int64_t n, float alpha, const float* x, float* y, dr_stream_t stream);
/* requesting-rank half of the backward exchange: per-slot gradient rows written straight into the all-to-all send
 * layout (pos is a permutation, every destination is written exactly once):
 *   out_rows[pos[b,f],:] = d_concat[b, f*D:(f+1)*D] + d_fm_logit[b] * (sum_x[b,:] - concat[b, f*D:(f+1)*D]) ;
 *   out_lin[pos[b,f]] = d_fm_logit[b] ;  bias_sum[0] += sum_b d_fm_logit[b].
 *   concat / sum_x / d_fm_logit / out_lin / bias_sum may be NULL. */
int dr_emb_pack_grads(const int64_t* pos, int64_t B, int32_t F, int32_t D, const float* d_concat,
                      int64_t ld_dconcat, const float* concat, int64_t ld_concat, const float* sum_x,
                      const float* d_fm_logit, float* out_rows, float* out_lin, float* bias_sum,
                      dr_stream_t stream);
/* the same for a de-duplicated exchange: pos maps the slots of a shared row to ONE destination; unique_flags [B * F] (the slot plan's
 * flags of the micro-batch) says which slots own their row -- those store, the others ADD with fp32 atomics (their order is not
 * fixed).  out_rows / out_lin must be zero-filled by the caller. */
int dr_emb_pack_grads_dedup(const int64_t* pos, const uint8_t* unique_flags, int64_t B, int32_t F, int32_t D, const float* d_concat,
                            int64_t ld_dconcat, const float* concat, int64_t ld_concat, const float* sum_x,
                            const float* d_fm_logit, float* out_rows, float* out_lin, float* bias_sum, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7p  first (wide) tower layer on PRE-SPLIT operands -- the "planes" form of the bf16x3 product mode.
 * Replaces the same reference ops as dr_linear_fwd / _bwd_dx / _bwd_dw (keras/models/ranking/deepfm.py:30-34,
 * estimator/models/feature_interaction/dnn.py:17-29 and their autodiff) for the one layer whose operands are large:
 * the producers of x (K3, dr_emb_pool_fwd_planes), of dy (dr_linear_bwd_narrow / dr_linear_bwd_dx with planes_out) and of W
 * (dr_bf3_split after each update) write every fp32 value v as three bf16 planes v0 + v1 + v2 (v0 = bf16(v),
 * v1 = bf16(v - v0), v2 = bf16(v - v0 - v1): exact, (v2 + v1) + v0 == v), so the GEMM stages tiles with LDS-DMA and does no
 * conversion work.  Results equal the bf16x3 mode of dr_linear_*: fp32 operands, six bf16 products per fp32 product,
 * fp32 accumulation.
 * Operand format: planes[p][row][col], p = 0..2, bf16; `ld` elements per row (multiple of 8), `plane_stride` elements
 * between planes (multiple of 8), base 16-byte aligned.
 *   dr_bf3_split    fp32 [R, C] -> planes (transpose = 1 writes planes[p][row_offset + c][col_offset + r])
 *   dr_bf3_join     planes -> fp32 (exact inverse; tests)
 *   dr_bf3_gemm_nt  C[m, n] = act(sum_k A[m, k] B[n, k] + bias[n]) (then zeroed where mask[m, n] <= 0);  K = padded
 *                   reduction length (multiple of 32), columns [true K, K) of BOTH operands' planes must be zero.
 *                   forward: A = x planes, B = W^T planes;  dgrad: A = dy planes, B = W planes.
 *   dr_bf3_gemm_tn  dst[f, n] += scale * sum_r X[r, f] Y[r, n];  dstb[n] += scale * y_colsum[n]  (split over r into
 *                   `workspace`, fixed-order reduce: deterministic).  Rows [R, roundup(R, 32)) of both operands' planes must
 *                   exist and be zero.  wgrad: X = x planes, Y = dy planes.
 * ---------------------------------------------------------------------------------------- */
int dr_bf3_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride,
                 int64_t ld_planes, int64_t row_offset, int64_t col_offset, int32_t transpose,
                 dr_stream_t stream);
int dr_bf3_join(const void* planes, int64_t plane_stride, int64_t ld_planes, int64_t R, int32_t C, float* dst,
                int64_t ld_dst, dr_stream_t stream);
int dr_bf3_gemm_nt(const void* a_planes, int64_t a_plane_stride, int64_t a_ld, const void* b_planes,
                   int64_t b_plane_stride, int64_t b_ld, int64_t M, int32_t N, int32_t K, const float* bias,
                   int32_t act, const float* mask, int64_t ld_mask, float* C, int64_t ldc, dr_stream_t stream);
/* dr_bf3_linear_nt: the same product with the ACTIVATION operand left in fp32 -- C[m, n] (+)= act(sum_k A[m, k] B[n, k] + bias[n])
 * (zeroed where mask[m, n] <= 0), A fp32 [M, K] row-major (lda % 4 == 0, 16-byte aligned), B planes [3][N][b_ld] with
 * b_ld >= roundup(K, 32) and columns [K, roundup(K, 32)) zero.  Each wave loads its 32 rows of A from HBM straight into
 * registers and splits them there; only B (the weights: W^T planes for the forward, W planes for the dgrad) is staged through
 * the LDS.  This is the form the engines use: activations and their producers are untouched, the weights' planes are refreshed
 * by dr_bf3_split after each update. */
int dr_bf3_linear_nt(const float* A, int64_t lda, const void* b_planes, int64_t b_plane_stride, int64_t b_ld, int64_t M,
                     int32_t N, int32_t K, const float* bias, int32_t act, const float* mask, int64_t ld_mask,
                     int32_t accumulate, float* C, int64_t ldc, dr_stream_t stream);
/* The sharded engine's first-layer dgrad and dr_emb_pack_grads in ONE launch (round 3): the gradient of slot (m, f) of F 64-wide
 * embeddings -- sum_k dy[m][k] W[64 f + d][k] + d_fm_logit[m] (sum_x[m][d] - x[m][64 f + d]) -- is written straight to
 * out_rows[pos[m, f], d], the all-to-all send layout (autodiff of keras/models/ranking/deepfm.py:44-45 + fm.py:23-37 of the
 * reference into the gradient exchange); d_concat never exists.  w_planes: W as planes [3][N rows][b_ld] (N = input features
 * >= 64 F; columns behind the embeddings are dropped).  out_lin (may be NULL)[pos[m, f]] = d_fm_logit[m]; bias_sum (may be NULL)
 * += sum_m d_fm_logit[m] in a fixed order; sum_x == NULL: no FM term.  pos: a permutation of the M F slots. */
int dr_bf3_linear_nt_pack(const float* dy, int64_t ld_dy, const void* w_planes, int64_t plane_stride, int64_t b_ld, int64_t M,
                          int32_t N, int32_t K, const int64_t* pos, int32_t F, const float* d_fm_logit, const float* sum_x,
                          const float* x, int64_t ld_x, float* out_rows, float* out_lin, float* bias_sum, dr_stream_t stream);
/* The same in the f16x2 operand mode (round 5): dy with its amax record, W as two fp16 planes (dr_h2_split) with theirs.  The epilogue
 * stages each 32 x 32 accumulator block through the LDS and moves float4s with 8 lanes per row: a row's 32 columns of a field are one
 * 128-byte line of its destination row and of the x row it reads.  x, sum_x (both required), out_rows 16-byte aligned, ld_x % 4 == 0.
 * Domain: M a multiple of 256 and M * F < 2^25; outside it DR_ESHAPE (the caller runs dr_h2_linear_nt + dr_emb_pack_grads). */
int dr_h2_linear_nt_pack(const float* dy, int64_t ld_dy, const uint32_t* dy_amax, const void* w_planes, int64_t plane_stride,
                         int64_t b_ld, const uint32_t* w_amax, int64_t M, int32_t N, int32_t K, const int64_t* pos, int32_t F,
                         const float* d_fm_logit, const float* sum_x, const float* x, int64_t ld_x, float* out_rows,
                         float* out_lin, float* bias_sum, dr_stream_t stream);
/* DCN cross layer forward (dr_cross_fwd's math, keras/models/ranking/dcn.py:81-88) on pre-split weights: wt_planes = W^T as
 * planes [3][Dm][ld_planes] (dr_bf3_split with transpose = 1).  x0, x, out, prod_out share leading dimension ld. */
int dr_bf3_cross_fwd(const float* x0, const float* x, int64_t ld, const void* wt_planes, int64_t plane_stride,
                     int64_t ld_planes, const float* b, float diag_scale, int64_t M, int32_t Dm, float* out,
                     float* prod_out, dr_stream_t stream);
/* dr_bf3_wgrad: dstW[f, n] += scale * sum_r x[r, f] dy[r, n];  dstb[n] += scale * sum_r dy[r, n] (dstb may be NULL) -- the wgrad of
 * a wide layer on its fp32 activations as they are (x [R, F], dy [R, N] row-major): x goes from HBM straight into registers
 * (lane index along f), dy is split once per 256 x-columns into a two-stage LDS image; bf16x3 product mode; split over r into
 * `workspace` (dr_bf3_wgrad_workspace_bytes) with a fixed-order reduce: deterministic. */
/* dr_bf3_emb_linear_fwd: K3 and the first Dense layer in ONE launch -- the register-split GEMM gathers its activation operand
 * straight from the embedding tables: k-tile kt of example m is dims 32 (kt & 1) .. + 31 of row row_base[f] + ids[m, f], f = kt / 2
 * (single-valued fields, D == 64, at most 32 dense features, no field longer than field_rows_max <= 2^24 rows -- the kernel
 * addresses a field as one buffer with 32-bit offsets; anything else returns DR_ESHAPE and the caller uses dr_emb_pool_fwd +
 * dr_bf3_linear_nt).
 * Replaces, together: the F DenseFeatures lookups + tf.stack / tf.concat + the first-order Dense(1) + FM.call
 * (keras/models/ranking/fm.py:23-37,47-64, deepfm.py:36-45) AND deepfm.py:30-34's first Dense(u, relu).
 *   in : ids [M, F] (-1 = missing), table [R, 64] (R < 2^31), lin_w [R] / lin_bias [1] (may be NULL), dense_pad [M, 32] = the
 *        K - 64 F dense features of every example, zero-padded to 32 (NULL iff K == 64 F), W^T planes [3][N][ld_planes]
 *        (columns [K, roundup(K, 32)) zero), bias [N]
 *   out: concat[:, 0 : 64 F) (for the backward kernels; the caller places the dense features in concat[:, 64 F : K) itself),
 *        sum_x [M, 64], fm_logit [M] (as dr_emb_pool_fwd), out [M, N] = act(x W + bias), x = [embeddings, dense features]. */
int dr_bf3_emb_linear_fwd(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max,
                          const float* table, int32_t D, const float* lin_w, const float* lin_bias, const float* dense_pad, float* concat,
                          int64_t ld_concat, int32_t K, const void* wt_planes, int64_t plane_stride, int64_t ld_planes, int32_t N,
                          const float* bias, int32_t act, float* sum_x, float* fm_logit, float* out, int64_t ld_out,
                          dr_stream_t stream);
/* dr_bf3_emb_linear_fwd that also saves the first-order weight of every slot as it was read: lin_vals_t [F, M] field-major,
 * lin_vals_t[f * M + m] = lin_w[row_base[f] + ids[m, f]] (undefined for a missing id; ignored when lin_w == NULL).  Handed to
 * dr_emb_pool_bwd_sorted_ex as `lin_old_t`, it turns the first-order update of a row that is unique in the batch into ONE write. */
int dr_bf3_emb_linear_fwd_lv(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max,
                             const float* table, int32_t D, const float* lin_w, const float* lin_bias, const float* dense_pad, float* concat,
                             int64_t ld_concat, int32_t K, const void* wt_planes, int64_t plane_stride, int64_t ld_planes, int32_t N,
                             const float* bias, int32_t act, float* sum_x, float* fm_logit, float* out, int64_t ld_out,
                             float* lin_vals_t, dr_stream_t stream);
int64_t dr_bf3_wgrad_workspace_bytes(int64_t R, int32_t F, int32_t N);
int dr_bf3_wgrad(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, int64_t R, int32_t F, int32_t N,
                 float scale, float* dstW, int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes,
                 dr_stream_t stream);
/* dr_bf3_wgrad for the FIRST tower layer, x = concat(field embeddings, dense features) gathered from the tables instead of read from
 * a buffer (D = 64; F = in_dim in [64 nf, 64 nf + 32]): ids_t [nf][R] int32 field-major ids (dr_ids_transpose_i32, -1 = missing),
 * dense_pad [R, 32] zero-padded dense features (NULL iff F == 64 nf).  Same workspace as dr_bf3_wgrad(R, F, N).  Every field
 * must have fewer than 2^24 - 1 rows (ids index a 4 GB raw buffer per field; -1 falls outside it and reads as zero).
 * Autodiff of the first Dense of keras/models/ranking/deepfm.py:30-34 w.r.t. its kernel, with deepfm.py:44-45's concat never built. */
int dr_bf3_wgrad_emb(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                     const float* dense_pad, const float* dy, int64_t ld_dy, int32_t F, int32_t N, float scale, float* dstW,
                     int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes, dr_stream_t stream);
/* ... in two halves like dr_linear_bwd_narrow_parts: parts = 1 the split-K GEMM into the workspace, 2 the fixed-order reduce that
 * applies it to dstW / dstb, 3 = both */
int dr_bf3_wgrad_emb_parts(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                           const float* dense_pad, const float* dy, int64_t ld_dy, int32_t F, int32_t N, float scale, float* dstW,
                           int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes, int32_t parts, dr_stream_t stream);
int64_t dr_bf3_gemm_tn_workspace_bytes(int64_t R, int32_t F, int32_t N);
int dr_bf3_gemm_tn(const void* x_planes, int64_t x_plane_stride, int64_t x_ld, const void* y_planes,
                   int64_t y_plane_stride, int64_t y_ld, int64_t R, int32_t F, int32_t N, float scale, float* dst,
                   int64_t ld_dst, const float* y_colsum, float* dstb, void* workspace, int64_t workspace_bytes,
                   dr_stream_t stream);

/* ---- "f16x2" operand mode of the first tower layer's three GEMMs (round 4) ---------------------------------------------------------
 * The same products as dr_bf3_emb_linear_fwd_lv / dr_bf3_linear_nt / dr_bf3_wgrad(_emb) (keras/models/ranking/deepfm.py:30-34 of the
 * reference and its autodiff; fp32 in, fp32 accumulate, fp32 out) with every operand value carried as TWO fp16 terms of x * s (22
 * significant bits) and a product formed from THREE matrix instructions instead of the bf16x3 mode's six.  s is a power of two per
 * TENSOR, derived inside the kernels from the tensor's `amax record`: one uint32 in device memory holding max |x| as float bits (or an
 * upper bound of it).  Records are written by dr_h2_amax or maintained by the tensor's producer (dr_emb_pool_bwd_sorted_ex and
 * dr_emb_pool_bwd_sorted_adam_ex keep the table's as a running maximum); a record that is too SMALL by a factor of 4 or more
 * saturates values instead of producing inf (still wrong: keep it an upper bound).  Accuracy: tests/test_gpu_h2_gemm.py.
 * Errors as the bf3 calls, plus DR_EINVAL for a missing record. */
int dr_h2_amax(const float* src, int64_t ld, int64_t R, int32_t C, uint32_t* amax, int32_t reset, dr_stream_t stream);
/* two fp16 planes of src * s(amax), laid out like dr_bf3_split's three (same arguments) */
int dr_h2_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride, int64_t ld_planes,
                int64_t row_offset, int64_t col_offset, int32_t transpose, const uint32_t* amax, dr_stream_t stream);
/* A weight's record and BOTH of its plane images in two launches (round 5): what dr_h2_amax(reset = 1) + dr_h2_split +
 * dr_h2_split(transpose = 1) produce -- same record, same planes -- without the 4-byte memset and the atomic pass (four small launches on
 * the serial chain between a wgrad and the next forward).  W [K, N] with row stride ldw; w_planes [2][K][w_ld], wt_planes [2][N][wt_ld];
 * parts: 256 uint32 of scratch. */
int dr_h2_refresh_weight(const float* W, int64_t ldw, int64_t K, int32_t N, void* w_planes, int64_t w_plane_stride, int64_t w_ld,
                         void* wt_planes, int64_t wt_plane_stride, int64_t wt_ld, uint32_t* amax, uint32_t* parts, dr_stream_t stream);
int dr_h2_linear_nt(const float* A, int64_t lda, const uint32_t* a_amax, const void* b_planes, int64_t b_plane_stride, int64_t b_ld,
                    const uint32_t* b_amax, int64_t M, int32_t N, int32_t K, const float* bias, int32_t act, const float* mask,
                    int64_t ld_mask, int32_t accumulate, float* C, int64_t ldc, uint32_t* c_amax, dr_stream_t stream);
/* c_amax / out_amax (may be NULL): reset, then raised to max |value stored into C / out| -- the record of the NEXT GEMM's operand */
/* Round 6 -- the first-layer dgrad of the DeepFM tower with K4's unique-row pass as its epilogue (csrc/h2_occ.hip; D = 64, single-valued
 * fields, SGD):  dx[m, 64 f + d] = sum_k dy[m][k] W[64 f + d][k] is the gradient row of slot (m, f).  For every slot whose table row no other
 * slot of the batch shares (unique_flags [M, F] of dr_emb_sort_slots; ids_t [F, M] int32 from dr_ids_transpose_i32, -1 = missing) the
 * kernel applies K4's update on the spot --  g = dx + d_fm_logit[m] (sum_x[m] - x), table[row] = x + scale g with x the row as it stands,
 * lin_w[row] = lin_old_t[f, m] + scale d_fm_logit[m]  -- operation for operation what dr_emb_pool_bwd_sorted_ex does with the stored
 * gradient (bit-identical tables); the other slots' dx goes to d_concat [M, ld_dconcat] for the duplicate pass.  Follow with
 * dr_emb_pool_bwd_sorted_ex(parts | 8) on the same stream (grad = d_concat): rows shared by several slots, hot rows, first-order
 * bias.  The tables must not be read by anything that needs their pre-step value after this call (run the first-layer wgrad that
 * gathers x BEFORE it).  w_planes: two fp16 planes [2][>= 64 F rows][w_ld] of the layer's kernel (dr_h2_split / dr_h2_refresh_weight's
 * `w`), w_ld >= roundup(K, 32); lin_w / lin_old_t both NULL or both given; table_amax (may be NULL): the table's running record, raised
 * to every value written.  Autodiff of keras/models/ranking/deepfm.py:30-34,44-45 and fm.py:23-37 w.r.t. the embedding tables. */
int dr_h2_dgrad_emb_sgd(const float* dy, int64_t ld_dy, const uint32_t* dy_amax, const void* w_planes, int64_t w_plane_stride,
                        int64_t w_ld, const uint32_t* w_amax, int64_t M, int32_t F, int32_t K, const int32_t* ids_t,
                        const uint8_t* unique_flags, const int64_t* row_base, float* table, float* lin_w, const float* lin_old_t,
                        const float* sum_x, const float* d_fm_logit, float scale, float* d_concat, int64_t ld_dconcat,
                        uint32_t* table_amax, dr_stream_t stream);
int dr_h2_cross_fwd(const float* x0, const float* x, int64_t ld, const uint32_t* x_amax, const void* wt_planes, int64_t plane_stride,
                    int64_t ld_planes, const uint32_t* w_amax, const float* b, float diag_scale, int64_t M, int32_t Dm, float* out,
                    float* prod_out, uint32_t* out_amax, dr_stream_t stream);
/* dr_cross_combine_bwd that also leaves the record of d_prod (the operand of the cross layer's dgrad and wgrad) */
int dr_cross_combine_bwd_amax(const float* x0, const float* prod, const float* d_out, int64_t M, int32_t Dm, int64_t ld, float diag_scale,
                              float* d_prod, float* d_x0_accum, float* d_x_accum, uint32_t* d_prod_amax, dr_stream_t stream);
/* dense_amax: the record of dense_pad, required iff K > 64 F; lin_vals_t may be NULL */
int dr_h2_emb_linear_fwd(const int64_t* ids, int64_t M, int32_t F, const int64_t* row_base, int64_t field_rows_max, const float* table,
                         int32_t D, const uint32_t* table_amax, const float* lin_w, const float* lin_bias, const float* dense_pad,
                         const uint32_t* dense_amax, float* concat, int64_t ld_concat, int32_t K, const void* wt_planes,
                         int64_t plane_stride, int64_t ld_planes, const uint32_t* w_amax, int32_t N, const float* bias, int32_t act,
                         float* sum_x, float* fm_logit, float* out, int64_t ld_out, float* lin_vals_t, dr_stream_t stream);
int dr_h2_wgrad(const float* x, int64_t ld_x, const uint32_t* x_amax, const float* dy, int64_t ld_dy, const uint32_t* dy_amax, int64_t R,
                int32_t F, int32_t N, float scale, float* dstW, int64_t ld_w, float* dstb, void* workspace, int64_t workspace_bytes,
                dr_stream_t stream);
int dr_h2_wgrad_emb(const int32_t* ids_t, int64_t R, int32_t nf, const int64_t* row_base, const float* table, int32_t D,
                    const uint32_t* table_amax, const float* dense_pad, const uint32_t* dense_amax, const float* dy, int64_t ld_dy,
                    const uint32_t* dy_amax, int32_t F, int32_t N, float scale, float* dstW, int64_t ld_w, float* dstb, void* workspace,
                    int64_t workspace_bytes, int32_t parts, dr_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * CIN  Compressed Interaction Network layer of xDeepFM (keras/models/ranking/xdeepfm.py:71-96):
 *   out[b, f, d] = act( sum_{i,j} W[i * Hk + j, f] * x0[b, i, d] * x[b, j, d] + bias[f] )
 * x0 [B, H0, D], x [B, Hk, D], W [H0 * Hk, Fm] (the conv1d kernel [1, H0 * Hk, Fm] of :55-61), out [B, Fm, D], all fp32
 * contiguous.  act: 0 none, 1 relu, 2 sigmoid (the layer's default), 3 tanh.  One MFMA GEMM whose left operand (the
 * outer product :80-86) is formed on the fly from LDS.  DR_ESHAPE when (H0 + Hk) * 65 floats exceed the 160 KB LDS.
 * dr_cin_bwd: autodiff of the above -- d_x0, d_x, dW [H0 * Hk, Fm], dbias [Fm] (may be NULL) from d_out and the saved out.
 * ---------------------------------------------------------------------------------------- */
int dr_cin_fwd(const float* x0, const float* x, int64_t B, int32_t H0, int32_t Hk, int32_t D, const float* W,
               int32_t Fm, const float* bias, int32_t act, float* out, dr_stream_t stream);
int dr_cin_bwd(const float* x0, const float* x, int64_t B, int32_t H0, int32_t Hk, int32_t D, const float* W,
               int32_t Fm, int32_t act, const float* out, const float* d_out, float* d_x0, float* d_x, float* dW,
               float* dbias, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DIN ActivationUnit input (keras/models/ranking/din.py:59-67): out[b, :] = concat(x[b], y[b], interacter(x, y)[b])
 * mode 0: no interacter (2 D columns), 1: x - y (keras Subtract, the reference test's interacter), 2: x * y (Multiply).
 * The two Dense layers that follow (:69-70) are dr_linear_fwd / dr_linear_bwd_*.
 * ---------------------------------------------------------------------------------------- */
int dr_din_concat_fwd(const float* x, const float* y, int64_t B, int32_t D, int32_t mode, float* out, int64_t ld_out,
                      dr_stream_t stream);
int dr_din_concat_bwd(const float* x, const float* y, int64_t B, int32_t D, int32_t mode, const float* d_out,
                      int64_t ld_dout, float* d_x, float* d_y, dr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise pieces around the GEMM path (deep_recommenders_amd/csrc/elementwise.hip):
 * dr_act_fwd / _bwd   a Dense activation other than ReLU (keras/models/ranking/deepfm.py:30-34 `dnn_activation`,
 *                     estimator/models/feature_interaction/dnn.py:9-14): act 1 relu, 2 sigmoid, 3 tanh, in place on the linear
 *                     layer's output [M, N] (leading dimension ld); backward dy *= act'(y) through the saved output.
 * dr_dropout_fwd/_bwd tf.nn.dropout(x, rate) of estimator/.../dnn.py:26-27 (always on there): keep iff hash(seed, index) >=
 *                     rate * 2^32, kept elements * 1 / (1 - rate); mask [M * N] bytes saved for the backward.
 * dr_reduce_sum       out[0] (+)= alpha * sum x (squared = 0) / alpha * sum x^2 (squared = 1: the L2 term of a Keras
 *                     kernel_regularizer / bias_regularizer, dcn.py:27-30,39-45); fixed-order two-stage reduction;
 *                     workspace: 1024 floats.
 * ---------------------------------------------------------------------------------------- */
int dr_act_fwd(float* x, int64_t M, int32_t N, int64_t ld, int32_t act, dr_stream_t stream);
int dr_act_bwd(const float* y, int64_t ld_y, float* dy, int64_t ld_dy, int64_t M, int32_t N, int32_t act,
               dr_stream_t stream);
int dr_dropout_fwd(const float* x, int64_t ld_x, int64_t M, int32_t N, float rate, uint64_t seed, float* y,
                   int64_t ld_y, uint8_t* mask, dr_stream_t stream);
int dr_dropout_bwd(const float* dy, int64_t ld_dy, const uint8_t* mask, int64_t M, int32_t N, float rate, float* dx,
                   int64_t ld_dx, dr_stream_t stream);
int dr_reduce_sum(const float* x, int64_t n, int32_t squared, float alpha, int32_t accumulate, float* out,
                  float* workspace, dr_stream_t stream);

/* dr_clock_stamp: dst[0] = the device's constant-rate wall clock (100 MHz ticks) when a one-thread kernel reaches the head of
 * `stream`.  Measurement plumbing with no reference counterpart: bench.py brackets the sharded step's cross-stream waits with two
 * stamps to report the EXPOSED part of the exchange (HIP timing events around a wait serialise the step). */
int dr_clock_stamp(uint64_t* dst, dr_stream_t stream);

/* Measurement plumbing: dst[0 .. bytes) = src[0 .. bytes) as a streaming copy of 16-byte vectors with nontemporal loads / stores
 * (bytes a multiple of 16, both pointers 16-byte aligned; DR_EINVAL otherwise).  bench.py times it over 1 GiB as the
 * `measured_copy_ceiling` beside the 8 TB/s spec peak: read + write bytes / time is what an HBM-bound kernel can reach on that box. */
int dr_copy_nt(const void* src, void* dst, int64_t bytes, dr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DR_HOTPATH_H */
