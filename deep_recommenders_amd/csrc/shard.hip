// Row-sharded tables over N ranks (new design; the reference is single-process).
//   owner(id) = id % world ; local row on the owner = field * rows_per_shard + id / world
// dr_shard_bucket_ids : stable counting sort of the B*F (example, field) slots by owner rank, producing the
//                       all-to-all send layout + the inverse map.  Pure integer work, deterministic.
// dr_rows_gather      : owner side of the forward exchange: requested local rows -> packed [n, D] (+ first-order w)
// dr_rows_scatter_add : owner side of the backward exchange: packed row gradients -> table (fp32 atomics;
//                       the same row can arrive from several ranks)
// dr_axpy             : y += alpha * x (applies the all-reduced dense-tower gradient)
#include "dr_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// The owner-side gather and the requester-side pack move every byte exactly once (random rows in, packed rows out; gradient rows in,
// permuted rows out): all their 16-byte accesses are nontemporal, like K4's (emb_sorted.hip; round 4).  DR_SHARD_NT=0: default policy.
#ifndef DR_SHARD_NT
#define DR_SHARD_NT 1
#endif
typedef float sh_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_once(const float* p) {
#if DR_SHARD_NT
    const sh_f4v v = __builtin_nontemporal_load(reinterpret_cast<const sh_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st4_once(float* p, const float4 v) {
#if DR_SHARD_NT
    const sh_f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<sh_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

constexpr int MAXW = 16;           // max world size for the bucketing kernels
constexpr int SLOTS_PER_THREAD = 8;
constexpr int CHUNK = 256 * SLOTS_PER_THREAD;

__device__ __forceinline__ void slot_route(int64_t id, int64_t p, int32_t C, int64_t rows_per_shard, int32_t world,
                                           int& owner, int64_t& local_row) {
    if (id >= 0) {
        owner = (int)(id % world);
        local_row = (int64_t)(p % C) * rows_per_shard + id / world;
    } else {                        // missing id: still occupies a slot (returns a zero row); spread evenly
        owner = (int)(p % world);
        local_row = -1;
    }
}

// rep (may be NULL): the slot plan's representative of every slot's row (dr_shard_dedup_slots); a slot that is not its own
// representative gets NO send slot -- its row travels once, with the representative -- and its pos is copied from it afterwards.
__global__ __launch_bounds__(256) void bucket_hist_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ rep, int64_t n, int32_t C,
                                                          int64_t rps, int32_t world, int64_t* __restrict__ block_hist) {
    __shared__ int hist[MAXW];
    if (threadIdx.x < MAXW) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * CHUNK + (int64_t)threadIdx.x * SLOTS_PER_THREAD;
    for (int j = 0; j < SLOTS_PER_THREAD; ++j) {
        const int64_t p = base + j;
        if (p < n && (rep == nullptr || rep[p] == p)) {
            int o; int64_t lr;
            slot_route(ids[p], p, C, rps, world, o, lr);
            atomicAdd(&hist[o], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < world) block_hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan over the (owner-major, block-minor) histogram; single block
__global__ __launch_bounds__(256) void bucket_scan_kernel(int64_t* __restrict__ block_hist, int64_t nblk, int32_t world,
                                                          int64_t* __restrict__ counts) {
    __shared__ int64_t part[256];
    __shared__ int64_t carry;
    const int64_t total = nblk * world;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < total; base += 256) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < total ? block_hist[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int64_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < total) block_hist[i] = carry + part[threadIdx.x] - v;   // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    // counts[w] = start[w+1] - start[w]
    if (threadIdx.x < world) {
        const int64_t s0 = block_hist[(int64_t)threadIdx.x * nblk];
        const int64_t s1 = threadIdx.x + 1 < world ? block_hist[(int64_t)(threadIdx.x + 1) * nblk] : carry;
        counts[threadIdx.x] = s1 - s0;
    }
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ rep, int64_t n, int32_t C,
                                                             int64_t rps, int32_t world,
                                                             const int64_t* __restrict__ block_off,
                                                             int64_t* __restrict__ send_rows,
                                                             int64_t* __restrict__ pos) {
    __shared__ int cnt[MAXW][257];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * CHUNK + (int64_t)t * SLOTS_PER_THREAD;
    int owner[SLOTS_PER_THREAD];
    int64_t lrow[SLOTS_PER_THREAD];
    int local[MAXW];
#pragma unroll
    for (int w = 0; w < MAXW; ++w) local[w] = 0;
#pragma unroll
    for (int j = 0; j < SLOTS_PER_THREAD; ++j) {
        const int64_t p = base + j;
        owner[j] = -1;
        if (p < n && (rep == nullptr || rep[p] == p)) {
            slot_route(ids[p], p, C, rps, world, owner[j], lrow[j]);
#pragma unroll
            for (int w = 0; w < MAXW; ++w) local[w] += (owner[j] == w) ? 1 : 0;
        }
    }
#pragma unroll
    for (int w = 0; w < MAXW; ++w) cnt[w][t] = local[w];
    __syncthreads();
    // exclusive scan over threads for each owner (one wave-group per owner would be faster; this is tiny work)
    for (int w = t >> 6; w < world; w += 4) {
        // wave (t>>6) scans cnt[w][0..255] in 4 segments of 64 with shuffles
        const int lane = t & 63;
        int run = 0;
        for (int seg = 0; seg < 4; ++seg) {
            const int v = cnt[w][seg * 64 + lane];
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(incl, off, 64);
                if (lane >= off) incl += u;
            }
            cnt[w][seg * 64 + lane] = run + incl - v;
            run += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();
    int seen[MAXW];
#pragma unroll
    for (int w = 0; w < MAXW; ++w) seen[w] = 0;
#pragma unroll
    for (int j = 0; j < SLOTS_PER_THREAD; ++j) {
        const int64_t p = base + j;
        if (p < n && owner[j] >= 0) {
            const int o = owner[j];
            int before = 0;
#pragma unroll
            for (int w = 0; w < MAXW; ++w)
                if (w == o) { before = seen[w]; seen[w]++; }
            const int64_t dst = block_off[(int64_t)o * gridDim.x + blockIdx.x] + cnt[o][t] + before;
            send_rows[dst] = lrow[j];
            pos[p] = dst;
        }
    }
}

// ---- requester-side de-duplication (round 4) ----------------------------------------------------------------------------------
// [TF] safe_embedding_lookup_sparse looks every DISTINCT id up once (`unique` inside embedding_lookup_sparse, reached from
// keras/models/ranking/fm.py:57-61 of the reference); the exchange can do the same: a row several slots of a micro-batch share
// travels once each way.  The representative of a row = its lowest slot.  Built from the slot plan of the micro-batch's ids
// (dr_emb_sort_slots over global rows f * V + id): the plan's sorted arrays hold the slots of shared rows grouped by row, slots
// ascending inside a row -- the segment start is found by binary search (lower bound of the row), no scan needed.
__global__ __launch_bounds__(256) void dedup_fill_kernel(int64_t* __restrict__ rep, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) rep[p] = p;
}
__global__ __launch_bounds__(256) void dedup_rep_kernel(const uint64_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                                        const int32_t* __restrict__ dup_count, uint64_t num_rows,
                                                        int64_t* __restrict__ rep) {
    const int64_t L = dup_count[1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < L; j += stride) {
        const uint64_t k = rows[j];
        if (k >= num_rows) continue;                       // missing ids (radix path: sentinel at the end) keep their own slot
        int64_t lo = 0, hi = j;                            // first index whose row is k
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rows[mid] < k) lo = mid + 1; else hi = mid;
        }
        rep[slots[j]] = slots[lo];
    }
}
__global__ __launch_bounds__(256) void dedup_pos_kernel(const int64_t* __restrict__ rep, int64_t n, int64_t* __restrict__ pos) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const int64_t r = rep[p];
        if (r != p) pos[p] = pos[r];                       // (representatives are written by bucket_scatter_kernel of the same call, earlier in stream order)
    }
}

// ---- owner-side gather / scatter of packed rows ------------------------------------------------------
// Branch-free (see emb_pool.hip): every load and store is unconditional.  An index past n is clamped to n - 1 (same
// source row, same destination, same value: a benign duplicate); a missing row (-1) reads row 0 and stores zeros.
template <int LPR>
__global__ __launch_bounds__(256) void rows_gather_kernel(const int64_t* __restrict__ rows, int64_t n,
                                                          const float* __restrict__ table, int32_t D,
                                                          const float* __restrict__ lin_w, float* __restrict__ out,
                                                          float* __restrict__ out_lin) {
    constexpr int RPW = DR_WAVE / LPR;    // rows per wave-instruction
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int nq = D >> 2;
    const int subc = sub < nq ? sub : nq - 1;
    const float* lsrc = lin_w != nullptr ? lin_w : table;           // value unused when lin_w == NULL
    float* ldst = out_lin != nullptr ? out_lin : out;               // idem: the row store that follows overwrites it
    const int64_t lpitch = out_lin != nullptr ? 1 : D;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int64_t g = w0 * (RPW * U); g < n; g += nw * (RPW * U)) {
        int64_t ic[U];
        bool present[U];
        float4 v[U];
        float lw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = g + u * RPW + slot;
            ic[u] = i < n ? i : n - 1;
            const int64_t r = rows[ic[u]];
            present[u] = r >= 0;
            const int64_t rc = present[u] ? r : 0;
            v[u] = ld4_once(table + rc * D + subc * 4);
            lw[u] = lsrc[lin_w != nullptr ? rc : 0];           // (no first-order table: ONE dummy address, not a random gather)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 x = present[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (out_lin == nullptr) ldst[ic[u] * lpitch + subc * 4] = x.x;      // dummy (same value the row store writes)
            else ldst[ic[u]] = (present[u] && lin_w != nullptr) ? lw[u] : 0.f;
            st4_once(out + ic[u] * D + subc * 4, x);
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const int64_t* __restrict__ rows, int64_t n,
                                                               const float* __restrict__ grads, int32_t D,
                                                               const float* __restrict__ lin_grads, float scale,
                                                               float* __restrict__ table, float* __restrict__ lin_w) {
    constexpr int RPW = DR_WAVE / LPR;
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int64_t g = w0 * RPW; g < n; g += nw * RPW) {
        const int64_t i = g + slot;
        if (i >= n) continue;
        const int64_t r = rows[i];
        if (r < 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = j * LPR + sub;          // strided lane->d map: each atomic instruction covers a contiguous span
            if (d < D) unsafeAtomicAdd(table + r * D + d, scale * grads[i * D + d]);
        }
        if (lin_w != nullptr && lin_grads != nullptr && sub == 0) unsafeAtomicAdd(lin_w + r, scale * lin_grads[i]);
    }
}

// Backward half of the exchange on the requesting rank: per-slot gradient rows written straight into the all-to-all
// send layout (pos is a permutation of 0..n-1: every destination written exactly once — no zero fill, no atomics).
//   out_rows[pos[b,f], :] = d_concat[b, f*D:(f+1)*D] + d_fm_logit[b] * (sum_x[b,:] - concat[b, f*D:(f+1)*D])
//   out_lin[pos[b,f]]     = d_fm_logit[b]
template <int LPR, int U, bool DEDUP>
__global__ __launch_bounds__(256) void pack_grads_kernel(const int64_t* __restrict__ pos, int64_t B, int32_t F, int32_t D,
                                                         const float* __restrict__ d_concat, int64_t ld,
                                                         const float* __restrict__ concat, int64_t ldc,
                                                         const float* __restrict__ sum_x,
                                                         const float* __restrict__ d_fm_logit,
                                                         float* __restrict__ out_rows, float* __restrict__ out_lin,
                                                         float* __restrict__ bias_sum, const uint8_t* __restrict__ uniq) {
    // One wave per example; branch-free loop body (see emb_pool.hip): slots past the last field repeat field F-1 (same
    // destination, same value), lanes past D/4 repeat the last float4; the example's F positions are read once,
    // coalesced, and broadcast with shuffles; the next example's positions are prefetched.
    constexpr int NS = DR_WAVE / LPR;
    const int lane = threadIdx.x & 63, slot = lane / LPR, sub = lane % LPR;
    const int nq = D >> 2;
    const int subc = sub < nq ? sub : nq - 1;
    const bool fm = concat != nullptr && sum_x != nullptr && d_fm_logit != nullptr;
    const float* xsrc = fm ? concat : d_concat;                      // values unused when !fm
    const int64_t xld = fm ? ldc : ld;
    const float* sxsrc = fm ? sum_x : d_concat;
    const int64_t sxp = fm ? (int64_t)D : 0;
    const float* dlsrc = d_fm_logit != nullptr ? d_fm_logit : d_concat;
    float* ldst = out_lin != nullptr ? out_lin : out_rows;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && bias_sum != nullptr && d_fm_logit != nullptr) dr_block_sum_axpy(d_fm_logit, B, 1.f, bias_sum);
    if (w0 >= B) return;
    for (int fg = 0; fg < F; fg += 64) {                               // one pass per group of 64 fields (F <= 64: one)
        const int Fg = F - fg < 64 ? F - fg : 64;
        const int lanec = fg + (lane < Fg ? lane : Fg - 1);
        int64_t my_pos = pos[w0 * F + lanec];
        // de-duplicated exchange (uniq != NULL): slots that share a row share a destination (zero-filled by the caller); they add
        // their gradients with fp32 atomics, slots whose row is unique in the micro-batch store as before.  The flag rides in
        // bit 62 of the position.
        if (DEDUP && !uniq[w0 * F + lanec]) my_pos |= (int64_t)1 << 62;
        float dl = dlsrc[w0];
        for (int64_t b = w0; b < B; b += nw) {
            const int64_t bn = b + nw < B ? b + nw : b;
            int64_t next_pos = pos[bn * F + lanec];
            if (DEDUP && !uniq[bn * F + lanec]) next_pos |= (int64_t)1 << 62;
            const float next_dl = dlsrc[bn];
            if (d_fm_logit == nullptr) dl = 0.f;
            const float4 sx = *reinterpret_cast<const float4*>(sxsrc + b * sxp + subc * 4);
            for (int f0 = 0; f0 < Fg; f0 += NS * U) {
                float4 g[U], x[U];
                int64_t p[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int f = f0 + u * NS + slot;
                    const int fl = f < Fg ? f : Fg - 1;                   // field within the group (lane that holds its pos)
                    const int fc = fg + fl;
                    p[u] = __shfl(my_pos, fl, 64);
                    g[u] = ld4_once(d_concat + b * ld + fc * D + subc * 4);
                    x[u] = ld4_once(xsrc + b * xld + fc * D + subc * 4);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float4 v = g[u];
                    if (fm) {
                        v.x += dl * (sx.x - x[u].x); v.y += dl * (sx.y - x[u].y);
                        v.z += dl * (sx.z - x[u].z); v.w += dl * (sx.w - x[u].w);
                    }
                    const bool shared = DEDUP && ((p[u] >> 62) & 1);
                    const int64_t pp = DEDUP ? (p[u] & ~((int64_t)1 << 62)) : p[u];
                    if (!shared) {
                        if (out_lin != nullptr) ldst[pp] = dl;
                        st4_once(out_rows + pp * D + subc * 4, v);
                    } else if (f0 + u * NS + slot < Fg) {                  // (a padded slot repeats field Fg - 1: add it once)
                        if (out_lin != nullptr && sub == 0) unsafeAtomicAdd(out_lin + pp, dl);
                        if (sub < nq) {
                            float* q = out_rows + pp * D + sub * 4;
                            unsafeAtomicAdd(q + 0, v.x); unsafeAtomicAdd(q + 1, v.y); unsafeAtomicAdd(q + 2, v.z); unsafeAtomicAdd(q + 3, v.w);
                        }
                    }
                }
            }
            my_pos = next_pos;
            dl = next_dl;
        }
    }
}

// (no __restrict__: x == y is a legal call, y *= 1 + alpha)
__global__ __launch_bounds__(256) void axpy_kernel(int64_t n, float alpha, const float* x, float* y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = fmaf(alpha, x[i], y[i]);
}

int lpr_for_d(int D) {
    int l = 1;
    while (l * 4 < D) l <<= 1;
    return l;
}

}  // namespace

extern "C" int64_t dr_shard_bucket_workspace_bytes(int64_t n, int32_t world) {
    const int64_t nblk = (n + CHUNK - 1) / CHUNK;
    return (nblk * world + 16) * (int64_t)sizeof(int64_t);
}

static int bucket_ids_impl(const int64_t* ids, const int64_t* rep, int64_t n, int32_t C, int64_t rows_per_shard, int32_t world,
                           int64_t* counts, int64_t* send_rows, int64_t* pos, int64_t* workspace, dr_stream_t stream) {
    if (n < 0 || C <= 0 || world <= 0 || world > MAXW || rows_per_shard <= 0) return DR_EINVAL;
    if (!counts) return DR_EINVAL;
    if (n == 0) {
        return hipMemsetAsync(counts, 0, sizeof(int64_t) * world, dr_s(stream)) == hipSuccess ? DR_OK : DR_ELAUNCH;
    }
    if (!ids || !send_rows || !pos || !workspace) return DR_EINVAL;
    const int nblk = (int)((n + CHUNK - 1) / CHUNK);
    hipLaunchKernelGGL(bucket_hist_kernel, dim3(nblk), dim3(256), 0, dr_s(stream), ids, rep, n, C, rows_per_shard, world,
                       workspace);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(256), 0, dr_s(stream), workspace, (int64_t)nblk, world, counts);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(nblk), dim3(256), 0, dr_s(stream), ids, rep, n, C, rows_per_shard, world,
                       workspace, send_rows, pos);
    if (rep != nullptr)
        hipLaunchKernelGGL(dedup_pos_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), rep, n, pos);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_shard_bucket_ids(const int64_t* ids, int64_t n, int32_t C, int64_t rows_per_shard, int32_t world,
                                   int64_t* counts, int64_t* send_rows, int64_t* pos, int64_t* workspace,
                                   dr_stream_t stream) {
    return bucket_ids_impl(ids, nullptr, n, C, rows_per_shard, world, counts, send_rows, pos, workspace, stream);
}

// The same with the representative map of dr_shard_dedup_slots: only slots that are their own representative get a send slot
// (counts then sum to the number of DISTINCT rows of the micro-batch, missing ids included one by one); pos[p] of every other slot =
// pos of its representative.
extern "C" int dr_shard_bucket_ids_dedup(const int64_t* ids, const int64_t* rep, int64_t n, int32_t C, int64_t rows_per_shard,
                                         int32_t world, int64_t* counts, int64_t* send_rows, int64_t* pos, int64_t* workspace,
                                         dr_stream_t stream) {
    if (!rep) return DR_EINVAL;
    return bucket_ids_impl(ids, rep, n, C, rows_per_shard, world, counts, send_rows, pos, workspace, stream);
}

// rep[p] = lowest slot of the micro-batch that looks up the same row as slot p (p itself for a row no other slot shares, and for a
// missing id), from the slot plan of the micro-batch's ids (dr_emb_sort_slots over the global rows row_base[f] + id; num_rows as
// given there).
extern "C" int dr_shard_dedup_slots(const int64_t* sorted_rows, const int32_t* sorted_slots, const int32_t* dup_count, int64_t n,
                                    int64_t num_rows, int64_t* rep, dr_stream_t stream) {
    if (n < 0 || num_rows <= 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!sorted_rows || !sorted_slots || !dup_count || !rep) return DR_EINVAL;
    hipLaunchKernelGGL(dedup_fill_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), rep, n);
    hipLaunchKernelGGL(dedup_rep_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream),
                       reinterpret_cast<const uint64_t*>(sorted_rows), sorted_slots, dup_count, (uint64_t)num_rows, rep);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

#define DR_LPR_SWITCH(lpr, CALL) \
    switch (lpr) {               \
        case 1: CALL(1); break;  \
        case 2: CALL(2); break;  \
        case 4: CALL(4); break;  \
        case 8: CALL(8); break;  \
        case 16: CALL(16); break;\
        case 32: CALL(32); break;\
        case 64: CALL(64); break;\
        default: return DR_EINVAL;\
    }

extern "C" int dr_rows_gather(const int64_t* rows, int64_t n, const float* table, int32_t D, const float* lin_w,
                              float* out_rows, float* out_lin, dr_stream_t stream) {
    if (n < 0 || D < 4 || D > 256 || (D & 3)) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!rows || !table || !out_rows) return DR_EINVAL;
    const int lpr = lpr_for_d(D);
    const int grid = dr_grid_for(n, 4 * (64 / lpr) * 4);
#define CALL(L) hipLaunchKernelGGL((rows_gather_kernel<L>), dim3(grid), dim3(256), 0, dr_s(stream), rows, n, table, D, lin_w, out_rows, out_lin)
    DR_LPR_SWITCH(lpr, CALL)
#undef CALL
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_rows_scatter_add(const int64_t* rows, int64_t n, const float* grads, int32_t D,
                                   const float* lin_grads, float scale, float* table, float* lin_w,
                                   dr_stream_t stream) {
    if (n < 0 || D < 4 || D > 256 || (D & 3)) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!rows || !grads || !table) return DR_EINVAL;
    const int lpr = lpr_for_d(D);
    const int grid = dr_grid_for(n, 4 * (64 / lpr));
#define CALL(L) hipLaunchKernelGGL((rows_scatter_add_kernel<L>), dim3(grid), dim3(256), 0, dr_s(stream), rows, n, grads, D, lin_grads, scale, table, lin_w)
    DR_LPR_SWITCH(lpr, CALL)
#undef CALL
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_axpy(int64_t n, float alpha, const float* x, float* y, dr_stream_t stream) {
    if (n < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!x || !y) return DR_EINVAL;
    hipLaunchKernelGGL(axpy_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), n, alpha, x, y);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

static int pack_grads_impl(const int64_t* pos, const uint8_t* uniq, int64_t B, int32_t F, int32_t D, const float* d_concat, int64_t ld_dconcat,
                           const float* concat, int64_t ld_concat, const float* sum_x, const float* d_fm_logit,
                           float* out_rows, float* out_lin, float* bias_sum, dr_stream_t stream) {
    if (B < 0 || F <= 0 || D < 4 || D > 256 || (D & 3)) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!pos || !d_concat || !out_rows || ld_dconcat < (int64_t)F * D || (ld_dconcat & 3)) return DR_EINVAL;
    if (concat != nullptr && (ld_concat < (int64_t)F * D || (ld_concat & 3))) return DR_EINVAL;
    const int lpr = lpr_for_d(D);
    const int grid = dr_grid_for(B, 4, 8192);
#define CALL(L)                                                                                                                       \
    if (uniq != nullptr)                                                                                                              \
        hipLaunchKernelGGL((pack_grads_kernel<L, (64 / L >= 16 ? 2 : 4), true>), dim3(grid), dim3(256), 0, dr_s(stream), pos, B, F, D, \
                           d_concat, ld_dconcat, concat, ld_concat, sum_x, d_fm_logit, out_rows, out_lin, bias_sum, uniq);             \
    else                                                                                                                              \
        hipLaunchKernelGGL((pack_grads_kernel<L, (64 / L >= 16 ? 2 : 4), false>), dim3(grid), dim3(256), 0, dr_s(stream), pos, B, F, D, \
                           d_concat, ld_dconcat, concat, ld_concat, sum_x, d_fm_logit, out_rows, out_lin, bias_sum, uniq)
    DR_LPR_SWITCH(lpr, CALL)
#undef CALL
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_emb_pack_grads(const int64_t* pos, int64_t B, int32_t F, int32_t D, const float* d_concat, int64_t ld_dconcat,
                                 const float* concat, int64_t ld_concat, const float* sum_x, const float* d_fm_logit,
                                 float* out_rows, float* out_lin, float* bias_sum, dr_stream_t stream) {
    return pack_grads_impl(pos, nullptr, B, F, D, d_concat, ld_dconcat, concat, ld_concat, sum_x, d_fm_logit, out_rows, out_lin, bias_sum,
                           stream);
}

// De-duplicated form: pos maps several slots to one destination (dr_shard_bucket_ids_dedup); unique_flags [B * F] = the slot plan's
// flags of the same micro-batch.  Slots whose row is unique store, the others ADD (fp32 atomics: the order in which a shared row's
// slots arrive is not fixed) -- out_rows / out_lin must be zero where shared rows land (the caller zero-fills the buffers).
extern "C" int dr_emb_pack_grads_dedup(const int64_t* pos, const uint8_t* unique_flags, int64_t B, int32_t F, int32_t D,
                                       const float* d_concat, int64_t ld_dconcat, const float* concat, int64_t ld_concat,
                                       const float* sum_x, const float* d_fm_logit, float* out_rows, float* out_lin, float* bias_sum,
                                       dr_stream_t stream) {
    if (!unique_flags) return DR_EINVAL;
    return pack_grads_impl(pos, unique_flags, B, F, D, d_concat, ld_dconcat, concat, ld_concat, sum_x, d_fm_logit, out_rows, out_lin,
                           bias_sum, stream);
}
