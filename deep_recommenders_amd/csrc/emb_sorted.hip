// K4, deterministic form -- de-duplicated scatter of the embedding gradients with the SGD step fused
// (dst_row += scale * sum of the gradients of every (example, field) slot that looked the row up).
//
// Why (measured on MI355X, profiles/r01_exp_emb_variants.log): fp32 atomics on 66 GB of random rows run at
// 2.7 TB/s algorithmic; a plain 16-byte load / add / store of each touched row runs at 5.2 TB/s -- but is only
// legal when exactly one lane group owns the row.  The slot plan (csrc/emb_plan.hip, dr_emb_sort_slots: hand-written claim
// table + LDS sort of the shared-row slots; a radix sort of all slots for skewed batches) makes ownership explicit,
// turns hot (Zipf) rows into a segmented sum instead of an atomic pile-up, and makes the update bit-reproducible
// run to run.  The plan depends only on the ids, so the engine builds it on a side stream.
//
// Replaces the autodiff of [TF] safe_embedding_lookup_sparse (IndexedSlices -> unsorted_segment_sum into the
// variable) reached from optimizer.minimize (examples/train_fm_on_movielens_estimator.py:51-52, reference root).
#include <cstdlib>
#include "dr_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstring>

namespace {

constexpr int CH = 32;   // a lane group sums at most CH consecutive sorted slots (bounds hot-row serialisation)

// Nontemporal 16-byte accesses for the three streams of the unique-row pass: the gradient (read once), the table rows (read once,
// written once, both lines of a row fully overwritten by the lane group) -- nothing of it is reused inside the kernel, and with the
// default policy those 1.3 GB of lines wash through the L2 / Infinity Cache next to the 4-byte first-order weights, whose
// read-modify-write is the one access that wants to find its line still cached.  Measured at config 3 (tools/exp/k4_ladder.py, round
// 4, same tensors in one process): 331 us default policy -> 316 nt row loads only -> 281 us nt on all three.  DR_K4_NT=0 at
// compile time restores the default policy (A/B builds).
#ifndef DR_K4_NT
#define DR_K4_NT 1
#endif
typedef float dr_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#if DR_K4_NT
    const dr_f4v v = __builtin_nontemporal_load(reinterpret_cast<const dr_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st4_stream(float* p, const float4 v) {
#if DR_K4_NT
    const dr_f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<dr_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// Fast path: streams the gradient in example order (coalesced), one wave per example; every slot whose row is
// unique in the batch gets ONE plain 16-byte-per-lane load / fma / store of its table row.  No atomics.
// Branch-free load phase (see emb_pool_fwd_sv_kernel in emb_pool.hip for why): every load of the
// loop body is unconditional with a clamped address -- a slot that is not unique / missing / past the last field
// reads table row 0 and its own (or field F-1's) gradient and simply stores nothing -- so all U row loads, U
// gradient loads and U first-order loads of an iteration are in flight together instead of one round trip each,
// and the next example's ids/flags are fetched ahead of them.  Only the stores stay predicated.
// Row-wise (lazy) Adam state for the fused optimizer form of K4: first/second moments with the table's shape, the
// [TF] B15 update  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr_t m / (sqrt(v) + eps),  lr_t = lr sqrt(1-b2^t)/(1-b1^t)
// (tf.train.AdamOptimizer, examples/train_fm_on_movielens_estimator.py:51 of the reference) applied to the rows a batch
// touches; duplicate slots of a row are summed first (one update per row and step).
struct AdamArgs {
    float* m; float* v;            // [R, D]
    float* m_lin; float* v_lin;    // [R] (may be null when there is no first-order table)
    int64_t lin_stride;            // 1: two arrays; 2: v_lin == m_lin + 1 -- the two moments of a row interleaved in ONE [R, 2] array, so
                                   // that a row's first-order state is one line operation to read and one to write instead of two each
                                   // (K4 is bound by line operations, DESIGN.md section 3 "Round 4")
    float lr_t, b1, b2, eps;
};

__device__ __forceinline__ float adam_elem(float g, float& m, float& v, const AdamArgs& a) {
    m = fmaf(a.b1, m, (1.f - a.b1) * g);
    v = fmaf(a.b2, v, (1.f - a.b2) * g * g);
    return -a.lr_t * m / (sqrtf(v) + a.eps);
}

// Running amax record of the table (round 4; include/dr_hotpath.h "f16x2 operand mode"): amax[0] = max(amax[0], |every value this
// kernel writes into dst_table|) as float bits -- the GEMMs that take table rows as an fp16-split operand derive their scale from it.
// One load per wave and an atomic only where it would raise the record (after the first few waves of the first step: never).
__device__ __forceinline__ float amax4(float mx, const float4& r) {
    return fmaxf(fmaxf(mx, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
}
__device__ __forceinline__ void amax_commit(uint32_t* __restrict__ rec, float mx) {
    uint32_t m = __float_as_uint(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(rec, m);
}

template <int LPR, int U, bool ADAM>
__device__ __forceinline__ void emb_bwd_unique_body(const int bid, const int nblk, const int64_t* __restrict__ ids,
                                                              const uint8_t* __restrict__ flags, int64_t B, int32_t F,
                                                              const int64_t* __restrict__ row_base, int32_t D,
                                                              const float* __restrict__ grad, int64_t ld,
                                                              const float* __restrict__ concat, int64_t ldc,
                                                              const float* __restrict__ sum_x,
                                                              const float* __restrict__ d_fm_logit,
                                                              const float* __restrict__ slot_lin, float scale,
                                                              float* __restrict__ table, float* __restrict__ lin_w,
                                                              float* __restrict__ lin_bias, AdamArgs ad,
                                                              const float* __restrict__ lin_old_t, uint32_t* __restrict__ amax) {
    constexpr int NS = DR_WAVE / LPR;
    float wmax = 0.f;
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int nq = D >> 2;
    const bool dvalid = sub < nq;
    // lin_old_t [F, B] (may be NULL; SGD only): the first-order weight every slot READ in the forward (dr_bf3_emb_linear_fwd_lv).  A
    // unique row's weight has not changed since, so its update is one WRITE of old + scale * g -- not a read-modify-write whose line
    // has to come back from HBM first.  K4 is bound by 128-byte line operations (8 per slot with the RMW: gradient 2, row 2 + 2,
    // first-order 1 + 1): this removes one of them (measured: 281 -> 264 us, tools/exp/k4_ladder.py LINW).
    const bool lold = lin_old_t != nullptr && lin_w != nullptr;
    const int subc = dvalid ? sub : nq - 1;
    const bool fm = sum_x != nullptr && d_fm_logit != nullptr;      // (a unique row's own value IS x: concat is never read here)
    const bool do_lin = lin_w != nullptr && (d_fm_logit != nullptr || slot_lin != nullptr) && sub == 0;
    const float* lsrc = lin_w != nullptr ? lin_w : table;                 // loaded value unused when lin_w == NULL
    const float* dlsrc = d_fm_logit != nullptr ? d_fm_logit : grad;       // idem
    const float* sxsrc = fm ? sum_x : grad;                               // idem
    const int64_t sx_pitch = fm ? D : 0;
    const int64_t nwaves = (int64_t)nblk * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // Bias gradient = scale * sum_b d_fm_logit[b].  One same-address atomic per wave (32 K of them at ~88/us on one L2
    // channel) used to cost more than the whole scatter; block 0 now sums the B values itself, in a fixed order
    // (deterministic), overlapped with every other block's work, and is the only writer of lin_bias.
    if (!ADAM && bid == 0 && lin_bias != nullptr && d_fm_logit != nullptr) dr_block_sum_axpy(d_fm_logit, B, scale, lin_bias);
    const float* mlsrc = (ADAM && ad.m_lin != nullptr) ? ad.m_lin : table;        // values unused when there is no first-order state
    const float* vlsrc = (ADAM && ad.v_lin != nullptr) ? ad.v_lin : table;
    if (wave0 >= B) return;
    const int lanec = lane < F ? lane : F - 1;
    const int64_t my_base = row_base[lanec];
    int64_t my_row;
    float dl;
    const float* losrc = lold ? lin_old_t + (int64_t)lanec * B : dlsrc;     // (dummy source when there is no saved weight)
    const int64_t lo_mul = lold ? 1 : 0;
    float my_lo;
    {
        const int64_t id = ids[wave0 * F + lanec];
        const uint8_t fl = flags[wave0 * F + lanec];
        my_row = (lane < F && id >= 0 && fl) ? my_base + id : -1;
        dl = dlsrc[wave0];
        my_lo = losrc[wave0 * lo_mul];
    }
    for (int64_t b = wave0; b < B; b += nwaves) {
        const int64_t bn = b + nwaves, bnc = bn < B ? bn : b;
        const int64_t next_id = ids[bnc * F + lanec];                    // prefetch, consumed at the loop bottom
        const uint8_t next_fl = flags[bnc * F + lanec];
        const float next_dl = dlsrc[bnc];
        const float next_lo = losrc[bnc * lo_mul];
        if (d_fm_logit == nullptr) dl = 0.f;
        const float* grow = grad + b * ld;
        const float4 sx = *reinterpret_cast<const float4*>(sxsrc + b * sx_pitch + subc * 4);
        for (int f0 = 0; f0 < F; f0 += NS * U) {
            int64_t row[U];
            float4 g[U], t[U];
            float4 mt[ADAM ? U : 1], vt[ADAM ? U : 1];
            float ml[ADAM ? U : 1], vl[ADAM ? U : 1];
            float lw[U], gl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = f0 + u * NS + slot;
                const int fc = f < F ? f : F - 1;
                row[u] = __shfl(my_row, fc, 64);
                if (f >= F) row[u] = -1;
                const int64_t rc = row[u] >= 0 ? row[u] : 0;
                // (without a first-order table the dummy loads below read ONE address: indexed by the row they were a random
                // 4-byte gather over the first gigabyte of `table` -- a line fetch per slot for nothing, 273 vs 236 us; round 4)
                const int64_t lc = (lin_w != nullptr && !lold) ? rc : 0;
                g[u] = ld4_stream(grow + fc * D + subc * 4);
                t[u] = ld4_stream(table + rc * D + subc * 4);
                lw[u] = lsrc[lc];
                const float lo_u = __shfl(my_lo, fc, 64);
                if (lold) lw[u] = lo_u;
                gl[u] = slot_lin != nullptr ? slot_lin[b * F + fc] : dl;
                if (ADAM) {
                    mt[u] = ld4_stream(ad.m + rc * D + subc * 4);
                    vt[u] = ld4_stream(ad.v + rc * D + subc * 4);
                    ml[u] = mlsrc[(ad.m_lin != nullptr) ? rc * ad.lin_stride : 0];
                    vl[u] = vlsrc[(ad.v_lin != nullptr) ? rc * ad.lin_stride : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 r = t[u];
                if (fm) {   // unique + single-valued: the row's current value IS the forward activation x[b, f, :]
                    g[u].x += dl * (sx.x - r.x); g[u].y += dl * (sx.y - r.y);
                    g[u].z += dl * (sx.z - r.z); g[u].w += dl * (sx.w - r.w);
                }
                float nl;
                if (ADAM) {
                    r.x += adam_elem(g[u].x, mt[u].x, vt[u].x, ad); r.y += adam_elem(g[u].y, mt[u].y, vt[u].y, ad);
                    r.z += adam_elem(g[u].z, mt[u].z, vt[u].z, ad); r.w += adam_elem(g[u].w, mt[u].w, vt[u].w, ad);
                    nl = lw[u] + adam_elem(gl[u], ml[u], vl[u], ad);
                } else {
                    r.x = fmaf(scale, g[u].x, r.x); r.y = fmaf(scale, g[u].y, r.y);
                    r.z = fmaf(scale, g[u].z, r.z); r.w = fmaf(scale, g[u].w, r.w);
                    nl = fmaf(scale, gl[u], lw[u]);
                }
                if (row[u] >= 0) {
                    if (dvalid) {
                        wmax = amax4(wmax, r);
                        st4_stream(table + row[u] * D + sub * 4, r);
                        if (ADAM) {
                            st4_stream(ad.m + row[u] * D + sub * 4, mt[u]);
                            st4_stream(ad.v + row[u] * D + sub * 4, vt[u]);
                        }
                    }
                    if (do_lin) {
                        lin_w[row[u]] = nl;
                        if (ADAM && ad.m_lin != nullptr) { ad.m_lin[row[u] * ad.lin_stride] = ml[u]; ad.v_lin[row[u] * ad.lin_stride] = vl[u]; }
                    }
                }
            }
        }
        my_row = (bn < B && lane < F && next_id >= 0 && next_fl) ? my_base + next_id : -1;
        dl = next_dl;
        my_lo = next_lo;
    }
    if (amax != nullptr) amax_commit(amax, wmax);
}

// Duplicate path: rows touched by >= 2 slots.  One lane group per segment head sums the slot gradients in sorted
// order.  A segment of <= CH slots is always owned by ONE group (plain RMW, deterministic); a hotter row is cut at
// CH-aligned positions that lie >= CH past its start, those pieces combine with fp32 atomics.
template <int LPR, bool ADAM>
__device__ __forceinline__ void emb_bwd_dups_body(const int bid, const int nblk, const uint64_t* __restrict__ rows,
                                                           const int32_t* __restrict__ slots, int64_t n,
                                                           const int32_t* __restrict__ dup_heads,
                                                           const int32_t* __restrict__ dup_count, int32_t F,
                                                           int32_t D, uint64_t num_rows,
                                                           const float* __restrict__ grad, int64_t ld,
                                                           const float* __restrict__ concat, int64_t ldc,
                                                           const float* __restrict__ sum_x,
                                                           const float* __restrict__ d_fm_logit,
                                                           const float* __restrict__ slot_lin, float scale,
                                                           float* __restrict__ table, float* __restrict__ lin_w, AdamArgs ad,
                                                           float* x_sorted, const bool det, uint32_t* __restrict__ amax) {
    float wmax = 0.f;
    // ADAM: the update is not linear in the gradient, so a row's slots must be summed completely before the one update:
    // the segment-start head walks the WHOLE segment (however long) and the aligned heads of hot rows do nothing.
    // One lane group (LPR lanes = one table row) per head.  The piece is walked in chunks of LPR sorted entries: the
    // chunk's row ids and slot numbers are fetched with ONE coalesced load each, the length of the matching prefix comes
    // from a ballot, and the gradient rows of the prefix are fetched UN at a time with clamped (unconditional) addresses.
    // (The first version chased slots[j] -> gradient row one entry at a time: a chain of dependent round trips that made
    // this kernel 3x the cost of everything else in K4 on Zipf-distributed ids.)
    constexpr int NS = DR_WAVE / LPR;
    constexpr int UN = LPR < 4 ? LPR : 4;
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int nq = D >> 2;
    const bool dvalid = sub < nq;
    const int subc = dvalid ? sub : nq - 1;
    const bool fm = sum_x != nullptr && d_fm_logit != nullptr;
    const bool any_lin = d_fm_logit != nullptr || slot_lin != nullptr;
    // x of a slot = its row's value before this step (single-valued fields): read from `concat` when the forward stored it, else
    // from x_sorted[i, :], the row of work-list head i snapshotted BEFORE this launch (dr_emb_snapshot_sorted_rows; every slot of
    // a piece shares the head's row) -- reading the table row itself here would race with the atomic pieces of a hot row
    const bool x_from_snap = fm && concat == nullptr;
    const float* xsrc = (fm && !x_from_snap) ? concat : grad;  // values unused when !fm or x_from_snap
    const int64_t xld = (fm && !x_from_snap) ? ldc : ld;
    const float* snap = x_from_snap ? x_sorted : grad;
    const float* sxsrc = fm ? sum_x : grad;
    const int64_t sxp = fm ? (int64_t)D : 0;
    const float* dlsrc = d_fm_logit != nullptr ? d_fm_logit : grad;
    const int64_t nwaves = (int64_t)nblk * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nheads = dup_count[0];
    n = dup_count[1];          // length of the sorted arrays: all B*F slots (radix path) or only the shared-row slots (claim path)
    const uint64_t gmask = LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull);
    for (int64_t h0 = wave0 * NS; h0 < nheads; h0 += nwaves * NS) {
        const int64_t h = h0 + slot;
        const bool live = h < nheads;
        const int64_t i = dup_heads[live ? h : nheads - 1];
        const uint64_t k = rows[i];
        const bool seg_start = (i == 0) || (rows[i > 0 ? i - 1 : 0] != k);
        // piece owned by this head: the segment start runs to the first aligned position >= i + CH, aligned heads run CH
        int64_t stop = seg_start ? (((i % CH) == 0) ? i + CH : ((i + CH - 1) / CH + 1) * CH) : i + CH;
        if (ADAM) stop = n;
        if (stop > n) stop = n;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        float dls = 0.f;
        int64_t j = i;
        // det: no piece of a hot row writes the row inside this kernel (the pieces are parked) and a row with ONE piece is read here,
        // by its only writer, before that writer's store -- so x comes straight from the table and no snapshot is needed
        const float4 xhead = (x_from_snap && det) ? *reinterpret_cast<const float4*>(table + k * (uint64_t)D + subc * 4)
                                                  : *reinterpret_cast<const float4*>(snap + (x_from_snap ? i : 0) * (int64_t)D + subc * 4);
        bool done = !live || (ADAM && !seg_start);
        while (__any(!done)) {
            const int64_t jj = j + sub;
            const bool inb = jj < stop;
            const int64_t jc = inb ? jj : stop - 1;                       // stop - 1 >= i: always a valid index
            const uint64_t rk = rows[jc];
            const int32_t sp = slots[jc];
            const bool match = !done && inb && rk == k;
            const uint64_t gb = (__ballot(match) >> (slot * LPR)) & gmask;
            const int cnt = gb == gmask ? LPR : (__ffsll((unsigned long long)~gb) - 1);     // matching prefix of the chunk
            for (int t0 = 0; t0 < LPR; t0 += UN) {
                if (!__any(t0 < cnt)) break;
                float4 v[UN], x[UN], sxv[UN];
                float dl[UN], gl[UN];
                bool act[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int t = t0 + u;
                    act[u] = t < cnt;
                    const int32_t p = __shfl(sp, slot * LPR + (act[u] ? t : 0), 64);
                    const int32_t b = p / F, f = p - b * F;
                    v[u] = *reinterpret_cast<const float4*>(grad + (int64_t)b * ld + f * D + subc * 4);
                    x[u] = *reinterpret_cast<const float4*>(xsrc + (int64_t)b * xld + f * D + subc * 4);
                    if (x_from_snap) x[u] = xhead;
                    sxv[u] = *reinterpret_cast<const float4*>(sxsrc + (int64_t)b * sxp + subc * 4);
                    dl[u] = dlsrc[b];
                    gl[u] = slot_lin != nullptr ? slot_lin[p] : dl[u];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    float4 w = v[u];
                    if (fm) {
                        w.x += dl[u] * (sxv[u].x - x[u].x); w.y += dl[u] * (sxv[u].y - x[u].y);
                        w.z += dl[u] * (sxv[u].z - x[u].z); w.w += dl[u] * (sxv[u].w - x[u].w);
                    }
                    if (act[u]) {
                        g.x += w.x; g.y += w.y; g.z += w.z; g.w += w.w;
                        dls += gl[u];
                    }
                }
            }
            j += cnt;
            if (cnt < LPR || j >= stop) done = true;
        }
        const bool exclusive = seg_start && (j >= n || rows[j < n ? j : n - 1] != k);
        float* dst = table + k * (uint64_t)D + sub * 4;
        if (ADAM) {
            if (live && seg_start) {
                if (dvalid) {
                    float4 t = *reinterpret_cast<const float4*>(dst);
                    float4 mt = *reinterpret_cast<const float4*>(ad.m + k * (uint64_t)D + sub * 4);
                    float4 vt = *reinterpret_cast<const float4*>(ad.v + k * (uint64_t)D + sub * 4);
                    t.x += adam_elem(g.x, mt.x, vt.x, ad); t.y += adam_elem(g.y, mt.y, vt.y, ad);
                    t.z += adam_elem(g.z, mt.z, vt.z, ad); t.w += adam_elem(g.w, mt.w, vt.w, ad);
                    wmax = amax4(wmax, t);
                    *reinterpret_cast<float4*>(dst) = t;
                    *reinterpret_cast<float4*>(ad.m + k * (uint64_t)D + sub * 4) = mt;
                    *reinterpret_cast<float4*>(ad.v + k * (uint64_t)D + sub * 4) = vt;
                }
                if (lin_w != nullptr && any_lin && sub == 0) {
                    float ml = ad.m_lin != nullptr ? ad.m_lin[k * ad.lin_stride] : 0.f, vl = ad.v_lin != nullptr ? ad.v_lin[k * ad.lin_stride] : 0.f;
                    lin_w[k] += adam_elem(dls, ml, vl, ad);
                    if (ad.m_lin != nullptr) { ad.m_lin[k * ad.lin_stride] = ml; ad.v_lin[k * ad.lin_stride] = vl; }
                }
            }
        } else if (live) {
            if (exclusive) {
                if (dvalid) {
                    float4 t = *reinterpret_cast<const float4*>(dst);
                    t.x = fmaf(scale, g.x, t.x); t.y = fmaf(scale, g.y, t.y);
                    t.z = fmaf(scale, g.z, t.z); t.w = fmaf(scale, g.w, t.w);
                    wmax = amax4(wmax, t);
                    *reinterpret_cast<float4*>(dst) = t;
                }
                if (lin_w != nullptr && any_lin && sub == 0) lin_w[k] = fmaf(scale, dls, lin_w[k]);
            } else if (det) {
                // a piece of a row hotter than CH slots, deterministic mode: park the piece's sum in ITS row of x_sorted (nobody
                // else reads or writes x_sorted[i]; its snapshot was consumed above) -- emb_bwd_hot_apply_kernel, launched behind
                // this kernel, adds a row's pieces in sorted order and updates the row (and its first-order weight) once
                if (dvalid) *reinterpret_cast<float4*>(x_sorted + i * (int64_t)D + sub * 4) = g;
                // the piece's first-order sum goes to the NEXT position's row, which no piece owns (positions inside a piece are
                // not heads) -- unless the piece is a single slot, whose neighbour belongs to the next row: the apply kernel
                // re-reads that one slot instead
                if (any_lin && sub == 0 && j - i >= 2) x_sorted[(i + 1) * (int64_t)D] = dls;
            } else {
                if (dvalid && amax != nullptr) {
                    // (every piece reports old + its own sum; the piece the memory system applies last reports the row's final value)
                    float4 o;
                    o.x = unsafeAtomicAdd(dst + 0, scale * g.x) + scale * g.x; o.y = unsafeAtomicAdd(dst + 1, scale * g.y) + scale * g.y;
                    o.z = unsafeAtomicAdd(dst + 2, scale * g.z) + scale * g.z; o.w = unsafeAtomicAdd(dst + 3, scale * g.w) + scale * g.w;
                    wmax = amax4(wmax, o);
                } else if (dvalid) {
                    unsafeAtomicAdd(dst + 0, scale * g.x); unsafeAtomicAdd(dst + 1, scale * g.y);
                    unsafeAtomicAdd(dst + 2, scale * g.z); unsafeAtomicAdd(dst + 3, scale * g.w);
                }
                if (lin_w != nullptr && any_lin && sub == 0) unsafeAtomicAdd(lin_w + k, scale * dls);
            }
        }
    }
    if (amax != nullptr) amax_commit(amax, wmax);
}

// K4 in ONE launch: the first `grid_d` blocks walk the duplicate segments, the rest stream the examples and update the rows
// that are unique in the batch.  The two touch disjoint rows (a row is unique or it is not), so nothing orders them; as two
// launches the duplicate pass (11 - 18 us on uniform ids, the longest part of K4 on Zipf ids) ran after the unique pass and a
// 12 us launch gap.  The duplicate blocks come first in the grid so that hot rows start early.
struct BwdSortedArgs {
    const int64_t* ids; const uint8_t* flags; int64_t B; int32_t F; const int64_t* row_base; int32_t D;
    const float* grad; int64_t ld; const float* concat; int64_t ldc; const float* sum_x; const float* d_fm_logit;
    const float* slot_lin; float scale; float* table; float* lin_w; float* lin_bias;
    const uint64_t* rows; const int32_t* slots; int64_t n; const int32_t* dup_heads; const int32_t* dup_count; uint64_t num_rows;
    float* x_sorted;
    int det;                                     // SGD: hot rows' pieces park their sums in x_sorted (emb_bwd_hot_apply_kernel adds them)
    int skip_unique_lin;                         // first-order weights of rows unique in the batch are updated by dr_emb_lin_update_unique
    const float* lin_old_t;                      // [F, B] first-order weights as the forward read them (may be NULL)
    uint32_t* amax;                              // running amax record of dst_table (may be NULL)
    int dups_only;                               // parts | 8: only the rows several slots share (+ the first-order bias)
};
#ifndef DR_K4_MINWAVES     // (experiment hooks: tools/exp/k4_occupancy.sh builds variants with -DDR_K4_MINWAVES=n / -DDR_K4_U=n)
#define DR_K4_LB __launch_bounds__(256)
#else
#define DR_K4_LB __launch_bounds__(256, DR_K4_MINWAVES)
#endif
template <int LPR, int U, bool ADAM>
__global__ DR_K4_LB void emb_bwd_sorted_kernel(BwdSortedArgs a, AdamArgs ad, int grid_d) {
    if ((int)blockIdx.x < grid_d)
        emb_bwd_dups_body<LPR, ADAM>(blockIdx.x, grid_d, a.rows, a.slots, a.n, a.dup_heads, a.dup_count, a.F, a.D, a.num_rows, a.grad,
                                     a.ld, a.concat, a.ldc, a.sum_x, a.d_fm_logit, a.slot_lin, a.scale, a.table, a.lin_w, ad, a.x_sorted,
                                     a.det != 0, a.amax);
    else if (a.dups_only) {
        // (parts | 8: the unique rows were updated by dr_h2_dgrad_emb_sgd; ONE extra block is launched for the first-order bias)
        if (!ADAM && a.lin_bias != nullptr && a.d_fm_logit != nullptr) dr_block_sum_axpy(a.d_fm_logit, a.B, a.scale, a.lin_bias);
    } else
        emb_bwd_unique_body<LPR, U, ADAM>(blockIdx.x - grid_d, gridDim.x - grid_d, a.ids, a.flags, a.B, a.F, a.row_base, a.D, a.grad,
                                          a.ld, a.concat, a.ldc, a.sum_x, a.d_fm_logit, a.slot_lin, a.scale, a.table,
                                          a.skip_unique_lin ? nullptr : a.lin_w, a.lin_bias, ad, a.lin_old_t, a.amax);
}

// Second half of the deterministic hot-row update (SGD with x_sorted given): for every row whose slots span more than one piece,
// the pieces' parked sums -- x_sorted[i] of the segment start, then the CH-aligned positions behind it -- are added in a fixed
// order, the row's first-order gradient is re-summed over its slots in a fixed order, and the row and its first-order weight
// get ONE plain read-modify-write.  Same head list, same piece geometry as emb_bwd_dups_body.
// Each lane screens one head (three dependent loads; uniform ids end here: no row has a second piece); a wave then takes the
// hot segments its lanes found one at a time with all 64 lanes: the pieces are counted 64 at a time with a ballot, the NS lane
// groups add contiguous runs of pieces (independent loads, four in flight) and are combined group 0 + 1 + 2 + ..., the
// first-order sum strides the segment with 64 lanes and is reduced by a fixed butterfly.  (A first version walked each hot row
// with one lane group and a loop-carried `rows[j] == k` test: 275 dependent round trips for the hottest row of a Zipf batch,
// +300 us per step.)
template <int LPR>
__global__ __launch_bounds__(256) void emb_bwd_hot_apply_kernel(BwdSortedArgs a) {
    constexpr int NS = DR_WAVE / LPR;
    const int lane = threadIdx.x & 63, grp = lane / LPR, sub = lane % LPR;
    const int D = a.D, F = a.F;
    const bool dvalid = sub < (D >> 2);
    const int subc = dvalid ? sub : (D >> 2) - 1;
    const bool any_lin = a.lin_w != nullptr && (a.d_fm_logit != nullptr || a.slot_lin != nullptr);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nheads = a.dup_count[0], n = a.dup_count[1];
    float wmax = 0.f;
    // 16 heads per wave and round, not 64: a wave works its hot segments off one after the other (8 - 12 dependent round trips
    // each), and a Zipf batch has ~1 400 of them among 88 K heads -- spread thin, they run in parallel (75 -> 30 us)
    constexpr int SCREEN = 16;
    for (int64_t h0 = wave0 * SCREEN; h0 < nheads; h0 += nwaves * SCREEN) {      // (wave-uniform loop)
        const int64_t h = h0 + lane;
        int64_t my_i = 0, my_stop = 0;
        bool hot = false;
        if (lane < SCREEN && h < nheads) {
            my_i = a.dup_heads[h];
            const uint64_t k = a.rows[my_i];
            const bool seg_start = (my_i == 0) || (a.rows[my_i - 1] != k);
            my_stop = ((my_i % CH) == 0) ? my_i + CH : ((my_i + CH - 1) / CH + 1) * CH;
            hot = seg_start && my_stop < n && a.rows[my_stop] == k;       // owner of a row with more than one piece
        }
        uint64_t todo = __ballot(hot);
        while (todo != 0ull) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1ull;
            const int64_t i = __shfl(my_i, src, 64), stop = __shfl(my_stop, src, 64);
            const uint64_t k = a.rows[i];
            // pieces behind the first: aligned positions stop, stop + CH, ... while they still hold k
            int64_t np = 0;
            for (;;) {
                const int64_t pos = stop + (np + lane) * CH;
                const bool ok = pos < n && a.rows[pos < n ? pos : n - 1] == k;
                const uint64_t m = __ballot(ok);
                const int c = m == ~0ull ? 64 : (__ffsll((unsigned long long)~m) - 1);
                np += c;
                if (c < 64) break;
            }
            // end of the segment: inside the last piece
            const int64_t lp = stop + (np - 1) * CH;
            int64_t e;
            {
                const int64_t pos = lp + lane;
                const bool ok = lane < CH && pos < n && a.rows[pos < n ? pos : n - 1] == k;
                const uint64_t m = __ballot(ok);
                e = lp + (__ffsll((unsigned long long)~m) - 1);           // (CH <= 32 < 64: ~m is never 0)
            }
            // parked sums: piece 0 at i, piece q >= 1 at stop + (q - 1) CH; lane group grp adds pieces [grp * per, (grp + 1) * per)
            const int64_t P = np + 1, per = (P + NS - 1) / NS;
            const int64_t q0 = grp * per, q1 = (q0 + per < P) ? q0 + per : P;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t q = q0; q < q1; q += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t qq = (q + u < q1) ? q + u : q1 - 1;
                    const int64_t pos = qq == 0 ? i : stop + (qq - 1) * CH;
                    v[u] = *reinterpret_cast<const float4*>(a.x_sorted + pos * (int64_t)D + subc * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (q + u < q1) { g.x += v[u].x; g.y += v[u].y; g.z += v[u].z; g.w += v[u].w; }
            }
#pragma unroll
            for (int o = 1; o < NS; ++o) {                                 // group 0 collects: ((g0 + g1) + g2) + ...
                const float ox = __shfl(g.x, sub + o * LPR, 64), oy = __shfl(g.y, sub + o * LPR, 64);
                const float oz = __shfl(g.z, sub + o * LPR, 64), ow = __shfl(g.w, sub + o * LPR, 64);
                if (grp == 0) { g.x += ox; g.y += oy; g.z += oz; g.w += ow; }
            }
            if (grp == 0 && dvalid) {
                float* dst = a.table + k * (uint64_t)D + sub * 4;
                float4 t = *reinterpret_cast<const float4*>(dst);
                t.x = fmaf(a.scale, g.x, t.x); t.y = fmaf(a.scale, g.y, t.y);
                t.z = fmaf(a.scale, g.z, t.z); t.w = fmaf(a.scale, g.w, t.w);
                wmax = amax4(wmax, t);
                *reinterpret_cast<float4*>(dst) = t;
            }
            if (any_lin) {
                // first-order sums of the pieces: parked at the row behind each piece's own (single-slot last piece: recomputed)
                float dls = 0.f;
                for (int64_t q = lane; q < P; q += 64) {
                    const int64_t pos = q == 0 ? i : stop + (q - 1) * CH;
                    const bool single = q == P - 1 && e - pos < 2;
                    float v = a.x_sorted[(single ? pos : pos + 1) * (int64_t)D];
                    if (single) {
                        const int32_t sp = a.slots[pos];
                        v = a.slot_lin != nullptr ? a.slot_lin[sp] : a.d_fm_logit[sp / F];
                    }
                    dls += v;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) dls += __shfl_xor(dls, off, 64);
                if (lane == 0) a.lin_w[k] = fmaf(a.scale, dls, a.lin_w[k]);
            }
        }
    }
    if (a.amax != nullptr) amax_commit(a.amax, wmax);
}

}  // namespace

static int bwd_sorted_impl(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                           const int32_t* sorted_slots, const uint8_t* unique_flags, const int32_t* dup_heads,
                           const int32_t* dup_count, int64_t B, int32_t F, int32_t D, int64_t num_rows, const float* grad,
                           int64_t ld_grad, const float* concat, int64_t ld_concat, const float* sum_x,
                           const float* d_fm_logit, const float* slot_lin_grad, float scale, float* dst_table,
                           float* dst_lin, float* dst_bias, const AdamArgs* adam, float* x_sorted, dr_stream_t stream,
                           int parts = 3, const float* lin_old_t = nullptr, uint32_t* table_amax = nullptr) {
    if (B < 0 || F <= 0 || F > 64 || D < 4 || D > 256 || (D & 3) || num_rows <= 0) return DR_EINVAL;
    const int64_t n = B * F;
    if (n == 0) return DR_OK;
    if (!ids || !row_base || !sorted_rows || !sorted_slots || !unique_flags || !dup_heads || !dup_count || !grad ||
        !dst_table ||
        ld_grad < (int64_t)F * D || (ld_grad & 3))
        return DR_EINVAL;
    if (concat != nullptr && (ld_concat < (int64_t)F * D || (ld_concat & 3) || sum_x == nullptr)) return DR_EINVAL;
    if (sum_x != nullptr && d_fm_logit == nullptr) return DR_EINVAL;
    if (sum_x != nullptr && concat == nullptr && x_sorted == nullptr) return DR_EINVAL;     // the FM term needs x from somewhere
    int lpr = 1;
    while (lpr * 4 < D) lpr <<= 1;
    const uint64_t* rows = reinterpret_cast<const uint64_t*>(sorted_rows);
    const int grid_u = dr_grid_for(B, 4, 8192);        // 1024 ... 16384 blocks measured: 338-353 us, no trend (bandwidth-bound)
    const int grid_d = 2048;    // the duplicate list's length lives on the device: uniform ids leave it nearly empty (surplus
                                // blocks exit at once), skewed ids fill it
    AdamArgs ad{};
    if (adam != nullptr) ad = *adam;
    // hot rows deterministic whenever the scratch rows exist (DR_K4_DETERMINISTIC=0: fp32 atomics as without x_sorted)
    static const bool det_on = [] { const char* e = getenv("DR_K4_DETERMINISTIC"); return !(e != nullptr && e[0] == '0'); }();
    const int det = (adam == nullptr && x_sorted != nullptr && det_on) ? 1 : 0;
    const BwdSortedArgs ba{ids, unique_flags, B, F, row_base, D, grad, ld_grad, concat, ld_concat, sum_x, d_fm_logit, slot_lin_grad,
                           scale, dst_table, dst_lin, dst_bias, rows, sorted_slots, n, dup_heads, dup_count, (uint64_t)num_rows,
                           x_sorted, det, (adam == nullptr && (parts & 4)) ? 1 : 0,
                           slot_lin_grad == nullptr ? lin_old_t : nullptr, table_amax, (parts & 8) ? 1 : 0};
#ifdef DR_K4_U
    constexpr int K4_U_OVERRIDE = DR_K4_U;
#else
    constexpr int K4_U_OVERRIDE = 0;
#endif
#define LAUNCH(L, ADAM_)                                                                                              \
    {                                                                                                                 \
        constexpr int NS_ = 64 / L;                                                                                   \
        constexpr int U_ = K4_U_OVERRIDE > 0 ? K4_U_OVERRIDE : (NS_ >= 16 ? 2 : 4);                                   \
        hipLaunchKernelGGL((emb_bwd_sorted_kernel<L, U_, ADAM_>), dim3(grid_d + ((parts & 8) ? 1 : grid_u)), dim3(256), 0, dr_s(stream), ba, ad,    \
                           grid_d);                                                                                   \
    }
#define CALL(L)                                                                                                       \
    if (adam != nullptr) LAUNCH(L, true) else LAUNCH(L, false)
    if (parts & 1)
    switch (lpr) {
        case 1: CALL(1); break;
        case 2: CALL(2); break;
        case 4: CALL(4); break;
        case 8: CALL(8); break;
        case 16: CALL(16); break;
        case 32: CALL(32); break;
        case 64: CALL(64); break;
        default: return DR_EINVAL;
    }
#undef CALL
#undef LAUNCH
    DR_CHECK_LAUNCH();
    if (det && (parts & 2)) {
        // deterministic hot rows: the pieces parked their sums in x_sorted, their owners add them up now (nothing to do when no row
        // has more than CH slots: the kernel scans the head list and exits)
#define HOT(L) hipLaunchKernelGGL((emb_bwd_hot_apply_kernel<L>), dim3(1024), dim3(256), 0, dr_s(stream), ba)
        switch (lpr) {
            case 1: HOT(1); break;
            case 2: HOT(2); break;
            case 4: HOT(4); break;
            case 8: HOT(8); break;
            case 16: HOT(16); break;
            case 32: HOT(32); break;
            default: HOT(64); break;
        }
#undef HOT
        DR_CHECK_LAUNCH();
    }
    return DR_OK;
}

extern "C" int dr_emb_pool_bwd_sorted(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                      const int32_t* sorted_slots, const uint8_t* unique_flags,
                                      const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                      int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                      const float* concat, int64_t ld_concat, const float* sum_x,
                                      const float* d_fm_logit, const float* slot_lin_grad, float scale,
                                      float* dst_table, float* dst_lin, float* dst_bias, float* x_sorted,
                                      dr_stream_t stream) {
    return bwd_sorted_impl(ids, row_base, sorted_rows, sorted_slots, unique_flags, dup_heads, dup_count, B, F, D, num_rows,
                           grad, ld_grad, concat, ld_concat, sum_x, d_fm_logit, slot_lin_grad, scale, dst_table, dst_lin,
                           dst_bias, nullptr, x_sorted, stream);
}

// The same call in two halves, for callers that time (or overlap) them separately: parts = 1 the update kernel, parts = 2 the ordered
// combination of hot rows' parked pieces (a no-op without x_sorted or with DR_K4_DETERMINISTIC=0), parts = 3 both = the call above.
// A caller that runs part 1 MUST run part 2 on the same stream before anything reads the tables.
// parts | 4: the first-order weights of rows that are UNIQUE in the batch are left alone -- the caller updates them with
// dr_emb_lin_update_unique (same values, any stream, any time between the head's backward and the next forward): a random 4-byte
// read-modify-write costs a 128-byte fetch, 0.27 GB of K4's 1.64 GB at config 3, and next to a GEMM it costs nothing.
extern "C" int dr_emb_pool_bwd_sorted_parts(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                            const int32_t* sorted_slots, const uint8_t* unique_flags,
                                            const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                            int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                            const float* concat, int64_t ld_concat, const float* sum_x,
                                            const float* d_fm_logit, const float* slot_lin_grad, float scale,
                                            float* dst_table, float* dst_lin, float* dst_bias, float* x_sorted,
                                            int32_t parts, dr_stream_t stream) {
    if (parts < 1 || parts > 7 || (parts & 3) == 0) return DR_EINVAL;
    return bwd_sorted_impl(ids, row_base, sorted_rows, sorted_slots, unique_flags, dup_heads, dup_count, B, F, D, num_rows,
                           grad, ld_grad, concat, ld_concat, sum_x, d_fm_logit, slot_lin_grad, scale, dst_table, dst_lin,
                           dst_bias, nullptr, x_sorted, stream, parts);
}

// dr_emb_pool_bwd_sorted_parts with `lin_old_t` [F, B] (field-major; may be NULL = as above): the first-order weight every slot read in
// the forward of THIS step (dr_bf3_emb_linear_fwd_lv).  Rows unique in the batch then get dst_lin[row] = lin_old + scale * g as ONE
// write -- valid only if nothing has written dst_lin since that forward.  Shared rows are summed and updated as before.
// table_amax (may be NULL): running amax record of dst_table -- raised to the largest |value| this call writes into the table (the
// f16x2 GEMMs, dr_h2_emb_linear_fwd / dr_h2_wgrad_emb, scale the table rows by it).
extern "C" int dr_emb_pool_bwd_sorted_ex(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                         const int32_t* sorted_slots, const uint8_t* unique_flags,
                                         const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                         int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                         const float* concat, int64_t ld_concat, const float* sum_x,
                                         const float* d_fm_logit, const float* slot_lin_grad, float scale,
                                         float* dst_table, float* dst_lin, float* dst_bias, float* x_sorted,
                                         const float* lin_old_t, int32_t parts, uint32_t* table_amax, dr_stream_t stream) {
    // parts | 8 (SGD form only): the slots whose row is unique in the batch were updated elsewhere (dr_h2_dgrad_emb_sgd) -- only the
    // duplicate pass, the hot rows and the first-order bias run here; `grad` need hold the rows of the non-unique slots only
    if (parts < 1 || parts > 15 || (parts & 3) == 0) return DR_EINVAL;
    return bwd_sorted_impl(ids, row_base, sorted_rows, sorted_slots, unique_flags, dup_heads, dup_count, B, F, D, num_rows,
                           grad, ld_grad, concat, ld_concat, sum_x, d_fm_logit, slot_lin_grad, scale, dst_table, dst_lin,
                           dst_bias, nullptr, x_sorted, stream, parts, lin_old_t, table_amax);
}

// dst_lin[row_base[f] + ids[b, f]] += scale * (slot_lin_grad ? slot_lin_grad[b, f] : d_fm_logit[b])  for every slot whose row no other
// slot of the batch shares (unique_flags of the slot plan) -- the part of K4's first-order update that dr_emb_pool_bwd_sorted_parts
// (parts | 4) leaves out.  One thread per slot, plain read-modify-write (a unique row has one writer): deterministic.
namespace {
__global__ __launch_bounds__(256) void lin_update_unique_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ flags,
                                                                int64_t n, int32_t F, const int64_t* __restrict__ row_base,
                                                                const float* __restrict__ d_fm_logit,
                                                                const float* __restrict__ slot_lin, float scale,
                                                                float* __restrict__ lin_w) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
        const int64_t id = ids[s];
        if (id < 0 || !flags[s]) continue;
        const int64_t b = s / F;
        const int f = (int)(s - b * F);
        const int64_t row = row_base[f] + id;
        const float g = slot_lin != nullptr ? slot_lin[s] : d_fm_logit[b];
        lin_w[row] = fmaf(scale, g, lin_w[row]);
    }
}
}  // namespace

extern "C" int dr_emb_lin_update_unique(const int64_t* ids, const uint8_t* unique_flags, int64_t B, int32_t F,
                                        const int64_t* row_base, const float* d_fm_logit, const float* slot_lin_grad, float scale,
                                        float* dst_lin, dr_stream_t stream) {
    if (B < 0 || F <= 0 || F > 64) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!ids || !unique_flags || !row_base || !dst_lin || (!d_fm_logit && !slot_lin_grad)) return DR_EINVAL;
    const int64_t n = B * F;
    hipLaunchKernelGGL(lin_update_unique_kernel, dim3(dr_grid_for(n, 256, 4096)), dim3(256), 0, dr_s(stream), ids, unique_flags, n, F,
                       row_base, d_fm_logit, slot_lin_grad, scale, dst_lin);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// x_sorted[i, :] = table[sorted_rows[i], :] for every position i on the duplicate pass's work list (dup_heads[0 .. dup_count[0])):
// the value, before this step's update, of each row that several slots share -- one row per PIECE the pass processes, at the
// piece's own sorted position.  Run BEFORE dr_emb_pool_bwd_sorted[_adam] when the forward did not store `concat`: the FM term of
// such slots needs their row's old value, and the table is being updated by the time the duplicate pass reads.
// (Uniform ids at config 3: 5.6 K rows = 1.4 MB; Zipf(1.05): 88 K rows.)  x_sorted must hold B * F rows (positions index it).
namespace {
__global__ __launch_bounds__(256) void snapshot_sorted_rows_kernel(const uint64_t* __restrict__ rows, const int32_t* __restrict__ dup_heads,
                                                                   const int32_t* __restrict__ dup_count, const float* __restrict__ table,
                                                                   int32_t D, uint64_t num_rows, float* __restrict__ x_sorted) {
    const int64_t nh = dup_count[0];
    const int q = D >> 2;                                                // float4 per row
    const int64_t total = nh * q, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t h = t / q;
        const int c = (int)(t - h * q);
        const int64_t i = dup_heads[h];
        const uint64_t k = rows[i];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < num_rows) v = *reinterpret_cast<const float4*>(table + k * (uint64_t)D + c * 4);
        *reinterpret_cast<float4*>(x_sorted + i * (int64_t)D + c * 4) = v;
    }
}
}  // namespace

extern "C" int dr_emb_snapshot_sorted_rows(const int64_t* sorted_rows, const int32_t* dup_heads, const int32_t* dup_count,
                                           const float* table, int32_t D, int64_t num_rows, float* x_sorted, dr_stream_t stream) {
    if (D < 4 || (D & 3) || num_rows <= 0) return DR_EINVAL;
    if (!sorted_rows || !dup_heads || !dup_count || !table || !x_sorted) return DR_EINVAL;
    hipLaunchKernelGGL(snapshot_sorted_rows_kernel, dim3(1024), dim3(256), 0, dr_s(stream), reinterpret_cast<const uint64_t*>(sorted_rows),
                       dup_heads, dup_count, table, D, (uint64_t)num_rows, x_sorted);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// Fused row-wise Adam form of the sorted K4: same inputs, `grad` is the gradient of the mean
// loss (no scale); rows touched by the batch get ONE [TF] B15 Adam update from the sum of their slots' gradients, with
// first/second moments m/v stored like the table.  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) is computed by the
// caller.  Untouched rows keep their moments ("lazy" Adam): identical to tf.train.AdamOptimizer on the first step,
// and on every step for rows that are touched every step; TF itself decays m/v of the whole variable each step.
// The bias of the first-order term is a dense parameter: update it with dr_adam_step on its own gradient.
extern "C" int dr_emb_pool_bwd_sorted_adam(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                           const int32_t* sorted_slots, const uint8_t* unique_flags,
                                           const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                           int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                           const float* concat, int64_t ld_concat, const float* sum_x,
                                           const float* d_fm_logit, const float* slot_lin_grad, float lr_t, float beta1,
                                           float beta2, float eps, float* table, float* m_table, float* v_table,
                                           float* lin_w, float* m_lin, float* v_lin, float* x_sorted, dr_stream_t stream) {
    if (!m_table || !v_table) return DR_EINVAL;
    if (lin_w != nullptr && (!m_lin || !v_lin)) return DR_EINVAL;
    // m_lin / v_lin as the two columns of one [R, 2] array (v_lin == m_lin + 1): row stride 2
    AdamArgs ad{m_table, v_table, m_lin, v_lin, (m_lin != nullptr && v_lin == m_lin + 1) ? 2 : 1, lr_t, beta1, beta2, eps};
    return bwd_sorted_impl(ids, row_base, sorted_rows, sorted_slots, unique_flags, dup_heads, dup_count, B, F, D, num_rows,
                           grad, ld_grad, concat, ld_concat, sum_x, d_fm_logit, slot_lin_grad, 0.f, table, lin_w, nullptr,
                           &ad, x_sorted, stream);
}

// ... with lin_old_t [F, B] (may be NULL) as in dr_emb_pool_bwd_sorted_ex: the first-order weight of a row unique in the batch is
// written from the value the forward of this step saved instead of being read again.  table_amax (may be NULL): as above.
extern "C" int dr_emb_pool_bwd_sorted_adam_ex(const int64_t* ids, const int64_t* row_base, const int64_t* sorted_rows,
                                              const int32_t* sorted_slots, const uint8_t* unique_flags,
                                              const int32_t* dup_heads, const int32_t* dup_count, int64_t B, int32_t F,
                                              int32_t D, int64_t num_rows, const float* grad, int64_t ld_grad,
                                              const float* concat, int64_t ld_concat, const float* sum_x,
                                              const float* d_fm_logit, const float* slot_lin_grad, float lr_t, float beta1,
                                              float beta2, float eps, float* table, float* m_table, float* v_table,
                                              float* lin_w, float* m_lin, float* v_lin, float* x_sorted, const float* lin_old_t,
                                              uint32_t* table_amax, dr_stream_t stream) {
    if (!m_table || !v_table) return DR_EINVAL;
    if (lin_w != nullptr && (!m_lin || !v_lin)) return DR_EINVAL;
    AdamArgs ad{m_table, v_table, m_lin, v_lin, (m_lin != nullptr && v_lin == m_lin + 1) ? 2 : 1, lr_t, beta1, beta2, eps};
    return bwd_sorted_impl(ids, row_base, sorted_rows, sorted_slots, unique_flags, dup_heads, dup_count, B, F, D, num_rows,
                           grad, ld_grad, concat, ld_concat, sum_x, d_fm_logit, slot_lin_grad, 0.f, table, lin_w, nullptr,
                           &ad, x_sorted, stream, 3, lin_old_t, table_amax);
}

// ---- TF's NON-lazy sparse Adam, evaluated lazily -----------------------------------------------------------------------------------
// tf.train.AdamOptimizer applies  m <- b1 m ;  v <- b2 v ;  w <- w - lr_s m / (sqrt(v) + eps)  to EVERY row on every step s, also to
// rows without a gradient (m * beta1 is assigned over the whole variable, the scaled gradients are scattered on
// top) -- a full pass over table + m + v, 200 GB at config 3.  Those decay-only steps commute with nothing else that touches the
// row, so they can be caught up the next time the row is LOOKED UP: `row_step[row]` = number of optimizer steps the row has
// received; before the forward of step t every slot's row is brought to t - 1 steps by replaying its missed decay-only steps
// (one claimer per row: atomicExch of the stamp), then K4's fused update applies step t itself.  Every value the model ever
// reads is then the value TF's dense update would have produced (same fp32 operations in the same order); rows never looked up
// again stay behind in memory until dr_adam_catchup_rows is called over them (checkpoint export).
// The replay stops early once a step can no longer change w (|lr_s m / (sqrt(v) + eps)| below half an ulp of w for every element of the
// row -- m shrinks by b1 per step while sqrt(v) shrinks by sqrt(b2), so the later steps are smaller still): the remaining decay
// of m, v is applied in closed form (b^k), which differs from k sequential multiplications in the last bits only.
__global__ __launch_bounds__(256) void adam_catchup_rows_kernel(const int64_t* __restrict__ ids, int64_t n, int32_t F,
                                                                const int64_t* __restrict__ row_base, int32_t D,
                                                                float* __restrict__ table, float* __restrict__ m,
                                                                float* __restrict__ v, float* __restrict__ lin_w,
                                                                float* __restrict__ m_lin, float* __restrict__ v_lin,
                                                                int32_t* __restrict__ row_step, int32_t upto, int32_t stamp,
                                                                float lr, float b1, float b2, float eps, int64_t ls) {
    // one 16-lane group per slot (D <= 64: 4 floats per lane; D > 64: the group walks the row in 64-float pieces)
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t groups = (int64_t)gridDim.x * (blockDim.x >> 4);
    const int64_t g0 = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const double lb1 = (double)b1, lb2 = (double)b2;
    for (int64_t p0 = g0 - (g0 & 3); p0 < n; p0 += groups) {                  // the 4 groups of a wave advance together
        const int64_t p = p0 + (lane >> 4);
        int64_t row = -1;
        int32_t old = upto;
        if (p < n) {
            const int64_t id = ids[p];
            if (id >= 0) {
                row = row_base[p % F] + id;
                if (sub == 0) old = atomicExch(&row_step[row], stamp);        // the first slot of a row in this call replays it
                old = __shfl(old, lane & ~15, 64);
            }
        }
        // (stamp 0 = the row never received an update: its moments are zero and every decay-only step is the identity)
        const int32_t k_total = row >= 0 && old > 0 && old < upto ? upto - old : 0;
        if (__ballot(k_total > 0) == 0) continue;
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + 4 * sub;
            const bool live = k_total > 0 && d < D;
            float4 w4 = make_float4(0, 0, 0, 0), m4 = w4, v4 = w4;
            if (live) {
                w4 = *reinterpret_cast<const float4*>(table + row * D + d);
                m4 = *reinterpret_cast<const float4*>(m + row * D + d);
                v4 = *reinterpret_cast<const float4*>(v + row * D + d);
            }
            // first-order weight of the row: lane 0 of the group, first piece only (rides in the .x slots of a second set)
            const bool lin_live = live && d0 == 0 && sub == 0 && lin_w != nullptr;
            float wl = 0.f, ml = 0.f, vl = 0.f;
            if (lin_live) { wl = lin_w[row]; ml = m_lin[row * ls]; vl = v_lin[row * ls]; }
            double p1 = pow(lb1, (double)(old + 1)), p2 = pow(lb2, (double)(old + 1));      // b^s of the first missed step
            int32_t done = 0;
            bool active = live;
            const uint64_t glive = __ballot(live) & (0xFFFFull << (lane & ~15));             // this row's lanes that hold elements
            while (__any(active)) {
                if (active) {
                    const float lr_s = (float)((double)lr * sqrt(1.0 - p2) / (1.0 - p1));
                    m4.x *= b1; m4.y *= b1; m4.z *= b1; m4.w *= b1;
                    v4.x *= b2; v4.y *= b2; v4.z *= b2; v4.w *= b2;
                    const float ux = lr_s * m4.x / (sqrtf(v4.x) + eps), uy = lr_s * m4.y / (sqrtf(v4.y) + eps);
                    const float uz = lr_s * m4.z / (sqrtf(v4.z) + eps), uw = lr_s * m4.w / (sqrtf(v4.w) + eps);
                    w4.x -= ux; w4.y -= uy; w4.z -= uz; w4.w -= uw;
                    float ul = 0.f;
                    if (lin_live) {
                        ml *= b1; vl *= b2;
                        ul = lr_s * ml / (sqrtf(vl) + eps);
                        wl -= ul;
                    }
                    p1 *= lb1; p2 *= lb2;
                    ++done;
                    // negligible from here on?  (|u| < 2^-26 |w| cannot change w; every later step is smaller by ~b1 / sqrt(b2))
                    const float tiny = 1.4901161e-8f;
                    const bool small = fabsf(ux) <= tiny * fabsf(w4.x) && fabsf(uy) <= tiny * fabsf(w4.y) &&
                                       fabsf(uz) <= tiny * fabsf(w4.z) && fabsf(uw) <= tiny * fabsf(w4.w) &&
                                       fabsf(ul) <= tiny * fabsf(wl);
                    // the verdict must hold for the whole row: vote among the group's live lanes (they are active together)
                    const bool row_small = ((__ballot(small) & glive) == glive);
                    if (done >= k_total || row_small) active = false;
                }
            }
            if (live) {
                const int32_t rem = k_total - done;                           // decay-only steps skipped as no-ops on w
                if (rem > 0) {
                    const float f1 = (float)pow(lb1, (double)rem), f2 = (float)pow(lb2, (double)rem);
                    m4.x *= f1; m4.y *= f1; m4.z *= f1; m4.w *= f1;
                    v4.x *= f2; v4.y *= f2; v4.z *= f2; v4.w *= f2;
                    ml *= f1; vl *= f2;
                }
                *reinterpret_cast<float4*>(table + row * D + d) = w4;
                *reinterpret_cast<float4*>(m + row * D + d) = m4;
                *reinterpret_cast<float4*>(v + row * D + d) = v4;
                if (lin_live) { lin_w[row] = wl; m_lin[row * ls] = ml; v_lin[row * ls] = vl; }
            }
        }
    }
}

// Brings the rows named by ids [n] (field of slot p = p % F, -1 = missing) to `upto` optimizer steps by replaying their missed
// decay-only Adam steps (see above), and stamps them `stamp`: before the forward of step t call with upto = t - 1, stamp = t (K4's
// fused update then applies step t); to export / inspect the tables call with upto = stamp = the number of steps taken, over all rows.
extern "C" int dr_adam_catchup_rows(const int64_t* ids, int64_t n, int32_t F, const int64_t* row_base, int32_t D, float* table,
                                    float* m_table, float* v_table, float* lin_w, float* m_lin, float* v_lin, int32_t* row_step,
                                    int32_t upto, int32_t stamp, float lr, float beta1, float beta2, float eps, dr_stream_t stream) {
    if (n < 0 || F <= 0 || D < 4 || (D & 3) || upto < 0 || stamp < upto) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!ids || !row_base || !table || !m_table || !v_table || !row_step) return DR_EINVAL;
    if (lin_w != nullptr && (!m_lin || !v_lin)) return DR_EINVAL;
    hipLaunchKernelGGL(adam_catchup_rows_kernel, dim3(dr_grid_for(n, 16, 8192)), dim3(256), 0, dr_s(stream), ids, n, F, row_base, D,
                       table, m_table, v_table, lin_w, m_lin, v_lin, row_step, upto, stamp, lr, beta1, beta2, eps,
                       (int64_t)((m_lin != nullptr && v_lin == m_lin + 1) ? 2 : 1));
    DR_CHECK_LAUNCH();
    return DR_OK;
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        float lr_t, float b1, float b2, float eps, float gscale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// Dense [TF] B15 Adam step over a flat parameter buffer: g is scaled by grad_scale first (e.g. 1 / world_size).
extern "C" int dr_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                            float beta2, float eps, float grad_scale, dr_stream_t stream) {
    if (n < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!param || !grad || !m || !v) return DR_EINVAL;
    hipLaunchKernelGGL(adam_step_kernel, dim3(dr_grid_for(n, 256 * 4)), dim3(256), 0, dr_s(stream), param, grad, m, v, n, lr_t,
                       beta1, beta2, eps, grad_scale);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

__global__ __launch_bounds__(256) void ftrl_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ accum, float* __restrict__ linear, int64_t n,
                                                        float lr, float lr_power, float l1, float l2, float gscale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gscale, a0 = accum[i], w = p[i];
        const float a1 = a0 + gi * gi;
        // accum^(-lr_power); lr_power = -0.5 (the default) is a square root
        const float pa0 = lr_power == -0.5f ? sqrtf(a0) : powf(a0, -lr_power);
        const float pa1 = lr_power == -0.5f ? sqrtf(a1) : powf(a1, -lr_power);
        const float lin = linear[i] + gi - (pa1 - pa0) / lr * w;
        const float quad = pa1 / lr + 2.f * l2;
        const float sgn = lin > 0.f ? 1.f : (lin < 0.f ? -1.f : 0.f);
        p[i] = fabsf(lin) > l1 ? (sgn * l1 - lin) / quad : 0.f;
        accum[i] = a1;
        linear[i] = lin;
    }
}

// Dense FTRL-Proximal step, TensorFlow's formulation (tf.train.FtrlOptimizer, the "wide" optimizer of the reference's
// examples/train_wdl_on_movielens_estimator.py:66-70: learning_rate 0.01, l1_regularization_strength 0.5; TF defaults
// learning_rate_power -0.5, initial_accumulator_value 0.1, l2 0, no shrinkage):
//   accum' = accum + g^2 ;  linear += g - (accum'^-p - accum^-p) / lr * w ;  quad = accum'^-p / lr + 2 l2
//   w = |linear| > l1 ? (sign(linear) l1 - linear) / quad : 0            (p = learning_rate_power)
extern "C" int dr_ftrl_step(float* param, const float* grad, float* accum, float* linear, int64_t n, float lr,
                            float lr_power, float l1, float l2, float grad_scale, dr_stream_t stream) {
    if (n < 0 || !(lr > 0.f)) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!param || !grad || !accum || !linear) return DR_EINVAL;
    hipLaunchKernelGGL(ftrl_step_kernel, dim3(dr_grid_for(n, 256 * 4)), dim3(256), 0, dr_s(stream), param, grad, accum, linear,
                       n, lr, lr_power, l1, l2, grad_scale);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
