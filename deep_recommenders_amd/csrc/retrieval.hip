// K10 + helpers — exact top-K maximum-inner-product search and the small retrieval-task ops.
//
// Replaces BruteForce.call / Streaming.call (keras/models/retrieval/factorized_top_k.py:178-260,316-334 of the
// reference: tf.matmul + tf.math.top_k, plus Streaming's concat/re-top_k reduce), _exclude / _take_long_axis
// (:26-67), FactorizedTopK.update_state (:489-512) and the helper layers of sbcnm.py:15-86.
//
// Search = chunked scores (dr_scores_nt, MFMA) into a workspace small enough to stay in the 256 MB Infinity
// Cache, then one wavefront per query row folds the chunk into a running, sorted top-k list held in registers
// (2 entries per lane, k <= 128).  Only elements strictly greater than the current k-th score enter the list
// (expected k*ln(N/k) insertions per row over the whole corpus), insertion = one rank count (wave sum) + a
// lane shift (__shfl_up).  Ties resolve to the lower candidate index, as tf.math.top_k does.
#include <cstdlib>
#include "dr_common.h"
#include "topk_list.h"
#include <math.h>

extern "C" int dr_scores_nt(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M, int32_t N, int32_t D,
                            float* out, int64_t ld_out, dr_stream_t stream);

namespace {

using drtk::KMAX;
using drtk::shfl_i64;
using drtk::shfl_up_i64;

// fold scores[row, 0..n) (candidate index = index_base + j) into the running sorted list of `row`.
// Candidate-list mode (cols != NULL): row `row` holds min(cnt[row], ld) entries (score, column) in ARBITRARY order (the
// filtering GEMM epilogue appends them with atomics); the list's index tie-break makes the result independent of arrival
// order.  On exit the kernel optionally publishes tau[row] (k-th best, -inf while the list is not full) and clears cnt[row].
template <bool CAND, typename IT = int64_t>                // CAND: candidate-list mode (cols != NULL); IT: index type inside the list (topk_list.h)
__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ scores, int64_t ld, int64_t Bq,
                                                          int64_t n, int32_t k, int64_t index_base, int32_t init,
                                                          float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                          const int32_t* __restrict__ cols, int32_t* __restrict__ cnt,
                                                          float* __restrict__ tau_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= Bq) return;
    drtk::ListT<IT> L;
    L.init(k, lane);
    if (!init) L.load(out_s + row * k, out_i + row * k);
    const float* srow = scores + row * ld;
    const int32_t* crow = CAND ? cols + row * ld : nullptr;
    if (cnt != nullptr) {
        const int64_t have = cnt[row];
        n = have < ld ? have : ld;
    }
    // SEL_U x 64 entries per round, the NEXT round's loads issued before this round's insertions (an insertion is a chain of
    // dependent cross-lane steps: a wave that waited for one 256-byte load per 64 entries, as first written, spent the dense first
    // chunk -- 8192 entries per row -- waiting for 128 memory round trips).  Entries are offered in index order, as before.
    constexpr int SEL_U = 4;
    float nv[SEL_U];
    int32_t nc[SEL_U];
    auto fetch = [&](int64_t j0) {                            // (n > 0 here; loads past the end re-read the last entry)
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            const int64_t j = j0 + 64 * u + lane;
            const int64_t jj = j < n ? j : n - 1;
            nv[u] = srow[jj];
            if constexpr (CAND) nc[u] = crow[jj];
            else nc[u] = 0;
        }
    };
    if (n > 0) fetch(0);
    for (int64_t j0 = 0; j0 < n; j0 += 64 * SEL_U) {
        float v[SEL_U];
        int32_t c[SEL_U];
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) { v[u] = nv[u]; c[u] = nc[u]; }
        if (j0 + 64 * SEL_U < n) fetch(j0 + 64 * SEL_U);
#pragma unroll
        for (int u = 0; u < SEL_U; ++u) {
            const int64_t j = j0 + 64 * u + lane;
            if (j0 + 64 * u >= n) break;                      // (wave-uniform)
            L.offer(v[u], (IT)(index_base + (CAND ? (int64_t)c[u] : j)), j < n);
        }
    }
    L.store(out_s + row * k, out_i + row * k);
    if (lane == 0) {
        if (tau_out != nullptr) tau_out[row] = L.full ? L.tau : -INFINITY;
        if (cnt != nullptr) cnt[row] = 0;
    }
}

// two-pointer merge of two sorted lists per row, list a wins ties (Streaming's reduce; cross-rank merge)
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ sa, const int64_t* __restrict__ ia, int32_t ka,
                                                         const float* __restrict__ sb, const int64_t* __restrict__ ib, int32_t kb,
                                                         int64_t Bq, int32_t k, float* __restrict__ os, int64_t* __restrict__ oi) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= Bq) return;
    int a = 0, b = 0;
    for (int t = 0; t < k; ++t) {
        const bool av = a < ka && ia[row * ka + a] >= 0, bv = b < kb && ib[row * kb + b] >= 0;
        float s = -INFINITY;
        int64_t id = -1;
        if (av && (!bv || sa[row * ka + a] >= sb[row * kb + b])) { s = sa[row * ka + a]; id = ia[row * ka + a]; ++a; }
        else if (bv) { s = sb[row * kb + b]; id = ib[row * kb + b]; ++b; }
        os[row * k + t] = s;
        oi[row * k + t] = id;
    }
}

__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t B, int32_t D,
                                                     float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < B; r += nw) {
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc = fmaf(a[r * D + d], b[r * D + d], acc);
        acc = dr_wave_sum(acc);
        if (lane == 0) out[r] = acc;
    }
}

__global__ __launch_bounds__(256) void gather_i64_kernel(const int64_t* __restrict__ src, int64_t nsrc, const int64_t* __restrict__ idx,
                                                         int64_t n, int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t j = idx[i];
        out[i] = (j >= 0 && j < nsrc) ? src[j] : -1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void take_along_rows_kernel(const T* __restrict__ arr, int64_t ld, int64_t B, int32_t C,
                                                              const int64_t* __restrict__ idx, int32_t K, T* __restrict__ out) {
    const int64_t n = B * K;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / K;
        const int64_t j = idx[i];
        out[i] = (j >= 0 && j < C) ? arr[r * ld + j] : (T)0;
    }
}

// hits[t] += #rows whose positive score has fewer than ks[t] top-k scores strictly above it ([TF] in_top_k, B14)
__global__ __launch_bounds__(256) void topk_hits_kernel(const float* __restrict__ pos, const float* __restrict__ topk, int64_t B,
                                                        int32_t K, const int32_t* __restrict__ ks, int32_t nk,
                                                        unsigned long long* __restrict__ hits) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < B; r += stride) {
        const float p = pos[r];
        int cnt = 0;
        for (int j = 0; j < K; ++j) cnt += topk[r * K + j] > p ? 1 : 0;
        for (int t = 0; t < nk; ++t)
            if (cnt < ks[t]) atomicAdd(&hits[t], 1ull);
    }
}

// adjusted = scores - isin(identifiers, exclude) * 1e5   (factorized_top_k.py:57-62)
__global__ __launch_bounds__(256) void exclude_adjust_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                                             int64_t B, int32_t K, const int64_t* __restrict__ excl, int32_t E,
                                                             float* __restrict__ out) {
    const int64_t n = B * K;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / K;
        const int64_t id = ids[i];
        bool isin = false;
        for (int e = 0; e < E; ++e) isin |= excl[r * E + e] == id;
        out[i] = scores[i] - (isin ? 1.0e5f : 0.f);
    }
}

// sbcnm helper layers on explicit [B, C] logits
__global__ __launch_bounds__(256) void logits_adjust_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                            int64_t B, int32_t C, const float* __restrict__ cand_prob,
                                                            const int64_t* __restrict__ cand_ids, float add_label_scale,
                                                            float* __restrict__ out) {
    constexpr float MIN_FLOAT = -3.4028234663852886e36f;
    // one wave per row: argmax(labels) (first max) then the per-element adjustment
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < B; r += nw) {
        int best = C;
        float bv = -INFINITY;
        if (cand_ids != nullptr) {
            for (int j = lane; j < C; j += 64) {
                const float v = labels[r * C + j];
                if (v > bv) { bv = v; best = j; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int ob = __shfl_xor(best, o, 64);
                if (ov > bv || (ov == bv && ob < best)) { bv = ov; best = ob; }
            }
        }
        const int64_t pid = (cand_ids != nullptr && best < C) ? cand_ids[best] : 0;
        for (int j = lane; j < C; j += 64) {
            float v = logits[r * C + j];
            if (cand_prob != nullptr) v -= logf(cand_prob[j]);                                  // sbcnm.py:86
            if (cand_ids != nullptr) v += ((cand_ids[j] == pid ? 1.f : 0.f) - labels[r * C + j]) * MIN_FLOAT;   // :66-75
            if (add_label_scale != 0.f) v += labels[r * C + j] * add_label_scale;                // :44 (logits + labels*MAX_FLOAT)
            out[r * C + j] = v;
        }
    }
}

// CCE(from_logits) per row on an explicit [B, C] matrix: row_loss = w * (lse * sum_j y_j - sum_j y_j s_j)
__global__ __launch_bounds__(256) void softmax_ce_rows_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                              int64_t B, int32_t C, float inv_t, const float* __restrict__ w,
                                                              float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < B; r += nw) {
        float m = -INFINITY;
        for (int j = lane; j < C; j += 64) m = fmaxf(m, logits[r * C + j] * inv_t);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float l = 0.f, ys = 0.f, y1 = 0.f;
        for (int j = lane; j < C; j += 64) {
            const float sv = logits[r * C + j] * inv_t, y = labels[r * C + j];
            l += expf(sv - m);
            ys = fmaf(y, sv, ys);
            y1 += y;
        }
        l = dr_wave_sum(l); ys = dr_wave_sum(ys); y1 = dr_wave_sum(y1);
        if (lane == 0) row_loss[r] = (w != nullptr ? w[r] : 1.f) * ((m + logf(l)) * y1 - ys);
    }
}
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += (double)v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void fill_topk_kernel(float* __restrict__ s, int64_t* __restrict__ i, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) { s[t] = -INFINITY; i[t] = -1; }
}

}  // namespace

extern "C" int dr_topk_select(const float* scores, int64_t ld, int64_t Bq, int64_t n, int32_t k, int64_t index_base,
                              int32_t init, float* out_scores, int64_t* out_index, dr_stream_t stream) {
    if (Bq < 0 || n < 0 || k <= 0 || k > KMAX || ld < n) return DR_EINVAL;
    if (Bq == 0) return DR_OK;
    if (!out_scores || !out_index || (n > 0 && !scores)) return DR_EINVAL;
    hipLaunchKernelGGL((topk_select_kernel<false, int64_t>), dim3((unsigned)((Bq + 3) / 4)), dim3(256), 0, dr_s(stream), scores, ld, Bq, n, k,
                       index_base, init, out_scores, out_index, (const int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// workspace = [append cursors Bq x i32][tau Bq x f32] (256-B aligned) + Bq x chunk x 8 B, used as a dense score matrix
// [Bq, chunk] for the first chunk and as candidate lists (scores + columns, capacity = chunk: cannot overflow) afterwards
// (+ two amax records behind tau: the queries' and the corpus', for the f16x2 scan)
static int64_t topk_hdr_bytes(int64_t Bq) { return (Bq * 8 + 8 + 255) / 256 * 256; }

// corpus chunk of the scan: ~256 MB of [Bq, chunk] scores (1024 output tiles of the register-split GEMM at Bq = 8192: 4 per CU)
constexpr int64_t TOPK_MAX_COLS = 32768;      // longest dense first chunk (the filtered chunks: TOPK_MAX_SCAN)
static int64_t topk_env(const char* name, int64_t dflt) {
    const char* e = getenv(name);
    return e != nullptr && e[0] != 0 ? atoll(e) : dflt;
}
static int64_t topk_chunk_for(int64_t Bq, int64_t N) {
    static const int64_t first_bytes = topk_env("DR_TOPK_FIRST_MB", 256) << 20;     // (experiment knob)
    int64_t chunk = first_bytes / (Bq > 0 ? Bq * 4 : 4);
    chunk = chunk / 256 * 256;
    if (chunk < 256) chunk = 256;
    // bounded in COLUMNS too, whatever Bq: a small query batch over a large corpus (serving) used to get a chunk of 256 MB / (4 Bq)
    // columns and planes / candidate lists sized for it -- 8.5 GB of workspace at Bq = 128, N = 10 M
    if (chunk > TOPK_MAX_COLS) chunk = TOPK_MAX_COLS;
    if (chunk > N) chunk = (N + 3) / 4 * 4;
    if (chunk < 4) chunk = 4;
    return chunk;
}
// bf16 planes of one corpus chunk for the register-split scan: [3][roundup(chunk, 32)][roundup(D, 32)]; D is not known to
// dr_topk_workspace_bytes, so the planes are budgeted for D <= TOPK_PLANES_MAX_D (larger D scans on the generic kernel)
constexpr int TOPK_PLANES_MAX_D = 512;
static int64_t topk_planes_bytes(int64_t chunk) { return ((3 * ((chunk + 31) / 32 * 32) * TOPK_PLANES_MAX_D * 2) + 255) / 256 * 256; }

// chunk of the FILTERED part of the scan (everything behind the dense first chunk): four times as long -- a launch of the
// register-split kernel then walks 16 tiles per block instead of 4 (one pipeline fill and drain per launch, one selection pass and
// one split per chunk: 122 -> 32 rounds over a 1 M corpus).  The candidate lists' capacity stays "one slot per scanned column"
// (cannot overflow, whatever the corpus order), so the workspace grows with it: 2.1 GB at Bq = 8192.
// Round 5: eight times, up to 65 536 columns (4.3 GB of lists at Bq = 8192): with the insertions and the filter's appends cheaper, a
// chunk's fixed costs -- launch, pipeline fill, one selection pass -- outweigh the candidates a staler threshold lets through
// (8192 x 1 M x 128: 11.2 ms at 32 768, 10.9 at 65 536, 10.8 at 131 072; `profiles/r05_topk.log`).  The dense first chunk keeps its
// own bound (TOPK_MAX_COLS): its selection runs one wavefront per query over the whole chunk.
constexpr int TOPK_SCAN_MULT = 8;
constexpr int64_t TOPK_MAX_SCAN = 65536;
static int64_t topk_scan_for(int64_t chunk, int64_t N) {
    if (N <= chunk) return chunk;
    int64_t rest = (N - chunk + 255) / 256 * 256;
    static const int64_t scan_cols = topk_env("DR_TOPK_SCAN_COLS", 0);              // (experiment knob) 0: TOPK_SCAN_MULT x chunk
    int64_t scan = scan_cols > 0 ? (scan_cols + 255) / 256 * 256 : chunk * TOPK_SCAN_MULT;
    if (scan > TOPK_MAX_SCAN && scan_cols <= 0) scan = TOPK_MAX_SCAN;
    if (scan > rest) scan = rest;
    return scan < chunk ? chunk : scan;
}

extern "C" int64_t dr_topk_workspace_bytes(int64_t Bq, int64_t N, int32_t k) {
    (void)k;
    const int64_t chunk = topk_chunk_for(Bq, N);
    const int64_t scan = topk_scan_for(chunk, N);
    return topk_hdr_bytes(Bq) + Bq * scan * 8 + topk_planes_bytes(scan);
}

int dr_scores_nt_filter(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M, int32_t N, int32_t D,
                        const float* tau, float* cand_s, int32_t* cand_c, int32_t* cand_cnt, int64_t cand_cap,
                        dr_stream_t stream);        // dense.hip
int dr_bf3_scores_filter(const float* a, int64_t lda, const void* b_planes, int64_t b_plane_stride, int64_t b_ld, int64_t M, int32_t N,
                         int32_t K, const float* tau, float* cand_s, int32_t* cand_c, int32_t* cand_cnt, int64_t cand_cap,
                         dr_stream_t stream);       // bf3_gemm.hip
extern "C" int dr_bf3_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride, int64_t ld_planes,
                            int64_t row_offset, int64_t col_offset, int32_t transpose, dr_stream_t stream);
extern "C" int dr_bf3_linear_nt(const float* A, int64_t lda, const void* b_planes, int64_t b_plane_stride, int64_t b_ld, int64_t M,
                                int32_t N, int32_t K, const float* bias, int32_t act, const float* mask, int64_t ld_mask,
                                int32_t accumulate, float* C, int64_t ldc, dr_stream_t stream);
extern "C" int32_t dr_get_gemm_mode(void);
extern "C" int32_t dr_get_gemm_split(void);
// the same three in the f16x2 operand mode (bf3_gemm.hip; include/dr_hotpath.h "f16x2 operand mode")
int dr_h2_scores_filter(const float* a, int64_t lda, const uint32_t* a_amax, const void* b_planes, int64_t b_plane_stride, int64_t b_ld,
                        const uint32_t* b_amax, int64_t M, int32_t N, int32_t K, const float* tau, float* cand_s, int32_t* cand_c,
                        int32_t* cand_cnt, int64_t cand_cap, dr_stream_t stream);
extern "C" int dr_h2_amax(const float* src, int64_t ld, int64_t R, int32_t C, uint32_t* amax, int32_t reset, dr_stream_t stream);
extern "C" int dr_h2_split(const float* src, int64_t ld_src, int64_t R, int32_t C, void* planes, int64_t plane_stride, int64_t ld_planes,
                           int64_t row_offset, int64_t col_offset, int32_t transpose, const uint32_t* amax, dr_stream_t stream);
extern "C" int dr_h2_linear_nt(const float* A, int64_t lda, const uint32_t* a_amax, const void* b_planes, int64_t b_plane_stride,
                               int64_t b_ld, const uint32_t* b_amax, int64_t M, int32_t N, int32_t K, const float* bias, int32_t act,
                               const float* mask, int64_t ld_mask, int32_t accumulate, float* C, int64_t ldc, uint32_t* c_amax,
                               dr_stream_t stream);

__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// Exact top-k of q @ cand^T over N candidates, never materialising [Bq, N].  Chunk 0 is scored densely and selected
// (nothing is known about the rows yet); every later chunk runs the GEMM with the FILTER epilogue: a score is written
// only if it beats its row's current k-th best, into a per-row candidate list that the selection kernel then folds in.
// After a few chunks almost nothing passes (expected k / items_seen of a chunk), so neither the [Bq, chunk] score
// matrix nor its re-read exist any more.
// `index` (may be null): a corpus pre-split by dr_topk_index_build -- [256-byte header: the corpus' amax record][2 fp16 planes of
// roundup(N, 32) x roundup(D, 32)].  Used when the scan runs in the f16x2 split: the per-call pass over the corpus for its record and
// the per-chunk split disappear (0.36 ms of an 11.1 ms pass at 8192 x 1 M x 128); the planes and the record are the ones the call would
// have made, so the result is the same bit for bit.  Any other mode ignores it and reads `cand`.
constexpr int64_t TOPK_INDEX_HDR = 256;
static int64_t topk_index_rows(int64_t N) { return (N + 31) / 32 * 32; }
static int64_t topk_index_ld(int32_t D) { return ((int64_t)D + 31) / 32 * 32; }

static int topk_mips_impl(const float* q, int64_t Bq, const float* cand, const void* index, int64_t N, int32_t D, int32_t k,
                          int64_t index_base, int32_t init, float* out_scores, int64_t* out_index, float* workspace,
                          int64_t workspace_bytes, dr_stream_t stream) {
    if (Bq < 0 || N < 0 || D < 4 || k <= 0 || k > KMAX) return DR_EINVAL;
    if (Bq == 0) return DR_OK;
    if (!q || !out_scores || !out_index || !workspace || (N > 0 && !cand)) return DR_EINVAL;
    if (init && k > N) return DR_ESHAPE;                 // "input must have at least k columns"
    const int64_t hdr = topk_hdr_bytes(Bq);
    // Register-split scan (bf16x3 product mode, D a multiple of 4 up to TOPK_PLANES_MAX_D, a workspace of dr_topk_workspace_bytes):
    // every chunk of the corpus is split into bf16 planes once (6 bytes per element, ~5 us per 8192 x 128 chunk) and BOTH the
    // dense first chunk and the filtered later chunks run on bf3_gemm_rs_kernel -- one kernel, so equal candidates tie
    // bit-exactly whichever chunk they sit in (the lower-index rule of tf.math.top_k depends on it).  Round 2 scanned with the
    // generic 128 x 128 kernel: 95 us per 4096-item chunk, 0.21 of the bf16x3 ceiling (24 ms per 8192 queries x 1 M items).
    int64_t chunk = topk_chunk_for(Bq, N);
    int64_t scan = topk_scan_for(chunk, N);                                   // length of the filtered chunks (>= chunk)
    const bool rs_ok = dr_get_gemm_mode() == DR_GEMM_BF16X3 && (D % 4) == 0 && D <= TOPK_PLANES_MAX_D && N > 0 &&
                       (reinterpret_cast<uintptr_t>(q) & 15) == 0;
    if (rs_ok && workspace_bytes < hdr + Bq * scan * 8 + topk_planes_bytes(scan)) scan = chunk;   // a round-2 sized workspace
    const bool rs_scan = rs_ok && workspace_bytes >= hdr + Bq * scan * 8 + topk_planes_bytes(scan);
    if (!rs_scan) {
        chunk = (workspace_bytes - hdr) / (Bq * 8);
        chunk = chunk >= 128 ? chunk / 128 * 128 : chunk / 4 * 4;
        scan = chunk;
    }
    if (chunk < 4 && N > 0) return DR_EINVAL;
    char* wsb = reinterpret_cast<char*>(workspace);
    int32_t* cnt = reinterpret_cast<int32_t*>(wsb);
    float* tau = reinterpret_cast<float*>(wsb + Bq * 4);
    float* dense = reinterpret_cast<float*>(wsb + hdr);                       // [Bq, chunk] scores (chunk 0)
    float* cand_s = dense;                                                    // [Bq, scan] candidate scores (later chunks)
    int32_t* cand_c = reinterpret_cast<int32_t*>(wsb + hdr + Bq * scan * 4);  // [Bq, scan] candidate columns
    void* planes = wsb + hdr + Bq * scan * 8;
    // f16x2 operand mode of the scan (round 4; DR_GEMM_SPLIT=bf16x3 restores the six-product scan): queries and corpus as two fp16 terms
    // of x * 2^k, three matrix instructions per fragment pair.  ONE scale for the whole corpus (a pass over it: 0.1 ms per million
    // 128-wide items) and one for the queries, so that equal candidates still tie bit-exactly whichever chunk they sit in.
    const bool h2 = rs_scan && dr_get_gemm_split() == DR_GEMM_SPLIT_F16X2;      // read at every call (dr_set_gemm_split)
    uint32_t* rec_q = reinterpret_cast<uint32_t*>(wsb + Bq * 8);
    const uint32_t* rec_c = rec_q + 1;
    // the corpus' planes and record come with the index -- unless its plane stride no longer fits the GEMM kernel's 32-bit buffer
    // offsets (2 planes x rows x ld x 2 bytes > 2^31: ~8.3 M items at D = 128): plane 1 would fall behind the resource's clamped
    // range and read as zeros, i.e. the scan would silently run on the h terms alone.  Such a corpus is split per chunk
    // into the workspace like an un-indexed one; only the index' record is used.
    const int64_t idx_ps = topk_index_rows(N) * (((int64_t)D + 31) / 32 * 32);
    const bool pre = h2 && index != nullptr && 2 * idx_ps * 2 <= (int64_t)0x7fffffff;
    if (h2) {
        int rc = dr_h2_amax(q, D, Bq, D, rec_q, 1, stream);
        if (rc == DR_OK && index == nullptr) rc = dr_h2_amax(cand, D, N, D, rec_q + 1, 1, stream);
        if (rc != DR_OK) return rc;
        if (index != nullptr) rec_c = const_cast<uint32_t*>(static_cast<const uint32_t*>(index));
    }
    const int64_t p_rows = pre ? topk_index_rows(N) : (scan + 31) / 32 * 32, p_ld = ((int64_t)D + 31) / 32 * 32, p_ps = p_rows * p_ld;
    if (rs_scan && !pre && (D % 32) != 0)      // the reduction padding of the planes must be zero (the kernel multiplies it)
        if (hipMemsetAsync(planes, 0, (size_t)(3 * p_ps * 2), dr_s(stream)) != hipSuccess) return DR_ELAUNCH;
    const unsigned sel_grid = (unsigned)((Bq + 3) / 4);
    // a search that starts here (init) only ever holds indices of [index_base, index_base + N): when they fit, the lists carry 32-bit
    // indices (half the cross-lane traffic of an insertion); a continued search (Streaming's later batches) may hold anything
    const bool idx32 = init != 0 && index_base >= 0 && index_base + N <= (int64_t)0x7fffffff;
    if (N == 0 && init) {
        hipLaunchKernelGGL(fill_topk_kernel, dim3(dr_grid_for(Bq * k, 256)), dim3(256), 0, dr_s(stream), out_scores, out_index,
                           Bq * k);
    }
    int first = init;
    for (int64_t c0 = 0; c0 < N; c0 += (c0 == 0 ? chunk : scan)) {
        const int64_t len = c0 == 0 ? chunk : scan;
        const int64_t nc = N - c0 < len ? N - c0 : len;
        if (pre) {                                                              // this chunk's rows of the index' planes
            planes = const_cast<char*>(static_cast<const char*>(index)) + TOPK_INDEX_HDR + c0 * p_ld * 2;
        } else if (rs_scan) {
            int rc = h2 ? dr_h2_split(cand + c0 * D, D, nc, D, planes, p_ps, p_ld, 0, 0, 0, rec_c, stream)
                        : dr_bf3_split(cand + c0 * D, D, nc, D, planes, p_ps, p_ld, 0, 0, 0, stream);
            if (rc != DR_OK) return rc;
        }
        if (c0 == 0) {
            int rc = h2 ? dr_h2_linear_nt(q, D, rec_q, planes, p_ps, p_ld, rec_c, Bq, (int32_t)nc, D, nullptr, 0, nullptr, 0, 0, dense, chunk,
                                          nullptr, stream)
                     : rs_scan ? dr_bf3_linear_nt(q, D, planes, p_ps, p_ld, Bq, (int32_t)nc, D, nullptr, 0, nullptr, 0, 0, dense, chunk, stream)
                             : dr_scores_nt(q, D, cand, D, Bq, (int32_t)nc, D, dense, chunk, stream);
            if (rc != DR_OK) return rc;
            if (N > chunk) hipLaunchKernelGGL(zero_i32_kernel, dim3(dr_grid_for(Bq, 256)), dim3(256), 0, dr_s(stream), cnt, Bq);
            if (idx32) hipLaunchKernelGGL((topk_select_kernel<false, int32_t>), dim3(sel_grid), dim3(256), 0, dr_s(stream), dense, chunk, Bq, nc, k,
                                          index_base, first, out_scores, out_index, (const int32_t*)nullptr, (int32_t*)nullptr, tau);
            else hipLaunchKernelGGL((topk_select_kernel<false, int64_t>), dim3(sel_grid), dim3(256), 0, dr_s(stream), dense, chunk, Bq, nc, k,
                                    index_base, first, out_scores, out_index, (const int32_t*)nullptr, (int32_t*)nullptr, tau);
        } else {
            int rc = h2 ? dr_h2_scores_filter(q, D, rec_q, planes, p_ps, p_ld, rec_c, Bq, (int32_t)nc, D, tau, cand_s, cand_c, cnt, scan, stream)
                     : rs_scan ? dr_bf3_scores_filter(q, D, planes, p_ps, p_ld, Bq, (int32_t)nc, D, tau, cand_s, cand_c, cnt, scan, stream)
                             : dr_scores_nt_filter(q, D, cand + c0 * D, D, Bq, (int32_t)nc, D, tau, cand_s, cand_c, cnt, scan, stream);
            if (rc != DR_OK) return rc;
            if (idx32) hipLaunchKernelGGL((topk_select_kernel<true, int32_t>), dim3(sel_grid), dim3(256), 0, dr_s(stream), cand_s, scan, Bq, nc, k,
                                          index_base + c0, 0, out_scores, out_index, cand_c, cnt, tau);
            else hipLaunchKernelGGL((topk_select_kernel<true, int64_t>), dim3(sel_grid), dim3(256), 0, dr_s(stream), cand_s, scan, Bq, nc, k,
                                    index_base + c0, 0, out_scores, out_index, cand_c, cnt, tau);
        }
        first = 0;
    }
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_topk_mips(const float* q, int64_t Bq, const float* cand, int64_t N, int32_t D, int32_t k,
                            int64_t index_base, int32_t init, float* out_scores, int64_t* out_index, float* workspace,
                            int64_t workspace_bytes, dr_stream_t stream) {
    return topk_mips_impl(q, Bq, cand, nullptr, N, D, k, index_base, init, out_scores, out_index, workspace, workspace_bytes, stream);
}

// The corpus side of BruteForce.index / Streaming's candidates (factorized_top_k.py:275-297 of the reference: candidates are handed over
// ONCE, queries arrive many times): what the f16x2 scan derives from the corpus on every call -- its amax record and the two fp16
// planes of corpus * 2^k -- built once.
extern "C" int64_t dr_topk_index_bytes(int64_t N, int32_t D) {
    if (N < 0 || D <= 0) return 0;
    return TOPK_INDEX_HDR + 2 * topk_index_rows(N) * topk_index_ld(D) * 2;
}

extern "C" int dr_topk_index_build(const float* cand, int64_t N, int32_t D, void* index, int64_t index_bytes, dr_stream_t stream) {
    if (N < 0 || D < 4 || (D % 4) != 0 || D > TOPK_PLANES_MAX_D) return DR_EINVAL;
    if (!index || (N > 0 && !cand) || index_bytes < dr_topk_index_bytes(N, D)) return DR_EINVAL;
    if ((reinterpret_cast<uintptr_t>(index) & 255) != 0) return DR_EINVAL;
    // header and padding (rows past N, columns past D) zero: the kernel multiplies the reduction padding
    if (hipMemsetAsync(index, 0, (size_t)dr_topk_index_bytes(N, D), dr_s(stream)) != hipSuccess) return DR_ELAUNCH;
    if (N == 0) return DR_OK;
    uint32_t* rec = static_cast<uint32_t*>(index);
    int rc = dr_h2_amax(cand, D, N, D, rec, 1, stream);
    const int64_t ld = topk_index_ld(D), ps = topk_index_rows(N) * ld;
    // in pieces: the split kernel's fast path indexes with 32 bits
    const int64_t piece = (int64_t)1 << 22;
    for (int64_t r0 = 0; r0 < N && rc == DR_OK; r0 += piece) {
        const int64_t nr = N - r0 < piece ? N - r0 : piece;
        rc = dr_h2_split(cand + r0 * D, D, nr, D, static_cast<char*>(index) + TOPK_INDEX_HDR, ps, ld, r0, 0, 0, rec, stream);
    }
    return rc;
}

extern "C" int dr_topk_mips_indexed(const float* q, int64_t Bq, const float* cand, const void* index, int64_t N, int32_t D, int32_t k,
                                    int64_t index_base, int32_t init, float* out_scores, int64_t* out_index, float* workspace,
                                    int64_t workspace_bytes, dr_stream_t stream) {
    if (index == nullptr || (reinterpret_cast<uintptr_t>(index) & 255) != 0) return DR_EINVAL;
    return topk_mips_impl(q, Bq, cand, index, N, D, k, index_base, init, out_scores, out_index, workspace, workspace_bytes, stream);
}

extern "C" int dr_topk_merge(const float* sa, const int64_t* ia, int32_t ka, const float* sb, const int64_t* ib, int32_t kb,
                             int64_t Bq, int32_t k, float* out_scores, int64_t* out_index, dr_stream_t stream) {
    if (Bq < 0 || ka < 0 || kb < 0 || k <= 0) return DR_EINVAL;
    if (Bq == 0) return DR_OK;
    if (!out_scores || !out_index || (ka > 0 && (!sa || !ia)) || (kb > 0 && (!sb || !ib))) return DR_EINVAL;
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((Bq + 255) / 256)), dim3(256), 0, dr_s(stream), sa, ia, ka, sb, ib, kb,
                       Bq, k, out_scores, out_index);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_rowdot(const float* a, const float* b, int64_t B, int32_t D, float* out, dr_stream_t stream) {
    if (B < 0 || D <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!a || !b || !out) return DR_EINVAL;
    hipLaunchKernelGGL(rowdot_kernel, dim3(dr_grid_for(B, 4)), dim3(256), 0, dr_s(stream), a, b, B, D, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

namespace {
// out[r, :] = x[r, :] * f(s[r]).  mode 0: f = s;  mode 1: f = 1 / sqrt(s) for s > 0, 1 for s == 0 (faiss.normalize_L2 on squared
// norms: zero rows stay);  mode 2: f = 1 / s for s > 0 and the row is taken from `fallback` for s == 0 (a k-means centroid = sum of its
// members / their count; an empty cluster keeps its centroid).
__global__ __launch_bounds__(256) void rows_scale_kernel(const float* __restrict__ x, const float* __restrict__ s, int mode,
                                                         const float* __restrict__ fallback, int64_t M, int32_t D, float* __restrict__ out) {
    const int64_t total = M * D, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / D;
        const float sv = s[r], xv = x[i];
        float o;
        if (mode == 0) o = xv * sv;
        else if (mode == 1) o = sv > 0.f ? xv / sqrtf(sv) : xv;
        else o = sv > 0.f ? xv / sv : fallback[i];
        out[i] = o;
    }
}
}  // namespace

extern "C" int dr_rows_scale(const float* x, const float* s, int32_t mode, const float* fallback, int64_t M, int32_t D, float* out,
                             dr_stream_t stream) {
    if (M < 0 || D <= 0 || mode < 0 || mode > 2) return DR_EINVAL;
    if (M == 0) return DR_OK;
    if (!x || !s || !out || (mode == 2 && !fallback)) return DR_EINVAL;
    hipLaunchKernelGGL(rows_scale_kernel, dim3(dr_grid_for(M * D, 256)), dim3(256), 0, dr_s(stream), x, s, mode, fallback, M, D, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_gather_i64(const int64_t* src, int64_t nsrc, const int64_t* idx, int64_t n, int64_t* out, dr_stream_t stream) {
    if (n < 0 || nsrc < 0) return DR_EINVAL;
    if (n == 0) return DR_OK;
    if (!src || !idx || !out) return DR_EINVAL;
    hipLaunchKernelGGL(gather_i64_kernel, dim3(dr_grid_for(n, 256)), dim3(256), 0, dr_s(stream), src, nsrc, idx, n, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_take_along_rows_f32(const float* arr, int64_t ld, int64_t B, int32_t C, const int64_t* idx, int32_t K, float* out,
                                      dr_stream_t stream) {
    if (B < 0 || C <= 0 || K < 0 || ld < C) return DR_EINVAL;
    if (B == 0 || K == 0) return DR_OK;
    if (!arr || !idx || !out) return DR_EINVAL;
    hipLaunchKernelGGL((take_along_rows_kernel<float>), dim3(dr_grid_for(B * K, 256)), dim3(256), 0, dr_s(stream), arr, ld, B, C,
                       idx, K, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_take_along_rows_i64(const int64_t* arr, int64_t ld, int64_t B, int32_t C, const int64_t* idx, int32_t K,
                                      int64_t* out, dr_stream_t stream) {
    if (B < 0 || C <= 0 || K < 0 || ld < C) return DR_EINVAL;
    if (B == 0 || K == 0) return DR_OK;
    if (!arr || !idx || !out) return DR_EINVAL;
    hipLaunchKernelGGL((take_along_rows_kernel<int64_t>), dim3(dr_grid_for(B * K, 256)), dim3(256), 0, dr_s(stream), arr, ld, B, C,
                       idx, K, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_topk_hits(const float* pos, const float* topk, int64_t B, int32_t K, const int32_t* ks, int32_t nk,
                            uint64_t* hits, dr_stream_t stream) {
    if (B < 0 || K < 0 || nk <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!pos || !ks || !hits || (K > 0 && !topk)) return DR_EINVAL;
    hipLaunchKernelGGL(topk_hits_kernel, dim3(dr_grid_for(B, 256)), dim3(256), 0, dr_s(stream), pos, topk, B, K, ks, nk,
                       reinterpret_cast<unsigned long long*>(hits));
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_exclude_adjust(const float* scores, const int64_t* ids, int64_t B, int32_t K, const int64_t* exclude, int32_t E,
                                 float* adjusted, dr_stream_t stream) {
    if (B < 0 || K <= 0 || E < 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!scores || !ids || !adjusted || (E > 0 && !exclude)) return DR_EINVAL;
    hipLaunchKernelGGL(exclude_adjust_kernel, dim3(dr_grid_for(B * K, 256)), dim3(256), 0, dr_s(stream), scores, ids, B, K, exclude,
                       E, adjusted);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_logits_adjust(const float* logits, const float* labels, int64_t B, int32_t C, const float* cand_prob,
                                const int64_t* cand_ids, float add_label_scale, float* out, dr_stream_t stream) {
    if (B < 0 || C <= 0) return DR_EINVAL;
    if (B == 0) return DR_OK;
    if (!logits || !out || ((cand_ids != nullptr || add_label_scale != 0.f) && !labels)) return DR_EINVAL;
    hipLaunchKernelGGL(logits_adjust_kernel, dim3(dr_grid_for(B, 4)), dim3(256), 0, dr_s(stream), logits, labels, B, C, cand_prob,
                       cand_ids, add_label_scale, out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

// d(row loss)/d logits of softmax_ce_rows_kernel, scaled by d_loss:  g[r][j] = w_r * inv_t * d_loss * (y1_r * softmax_j - y_j).
// cols == NULL: written densely to out[r * ld_out + j]; cols given ([B, C] column numbers, the hard-negative selection):
// scattered to out[r * ld_out + cols[r][j]] (out pre-zeroed by the caller).
__global__ __launch_bounds__(256) void softmax_ce_rows_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                                  int64_t B, int32_t C, float inv_t, const float* __restrict__ w,
                                                                  float d_loss, const int64_t* __restrict__ cols,
                                                                  float* __restrict__ out, int64_t ld_out) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < B; r += nw) {
        float m = -INFINITY;
        for (int j = lane; j < C; j += 64) m = fmaxf(m, logits[r * C + j] * inv_t);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float l = 0.f, y1 = 0.f;
        for (int j = lane; j < C; j += 64) {
            l += expf(logits[r * C + j] * inv_t - m);
            y1 += labels[r * C + j];
        }
        l = dr_wave_sum(l); y1 = dr_wave_sum(y1);
        const float k = (w != nullptr ? w[r] : 1.f) * inv_t * d_loss, inv_l = 1.f / l;
        for (int j = lane; j < C; j += 64) {
            const float g = k * (y1 * expf(logits[r * C + j] * inv_t - m) * inv_l - labels[r * C + j]);
            const int64_t col = cols != nullptr ? cols[r * C + j] : j;
            out[r * ld_out + col] = g;
        }
    }
}

extern "C" int dr_softmax_ce_rows_bwd(const float* logits, const float* labels, int64_t B, int32_t C, float inv_temperature,
                                      const float* sample_weight, float d_loss, const int64_t* cols, float* out, int64_t ld_out,
                                      dr_stream_t stream) {
    if (B <= 0 || C <= 0 || ld_out <= 0) return DR_EINVAL;
    if (!logits || !labels || !out) return DR_EINVAL;
    if (cols == nullptr && ld_out < C) return DR_EINVAL;
    hipLaunchKernelGGL(softmax_ce_rows_bwd_kernel, dim3(dr_grid_for(B, 4)), dim3(256), 0, dr_s(stream), logits, labels, B, C,
                       inv_temperature, sample_weight, d_loss, cols, out, ld_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_softmax_ce_rows(const float* logits, const float* labels, int64_t B, int32_t C, float inv_temperature,
                                  const float* sample_weight, float* row_loss, float* loss_out, dr_stream_t stream) {
    if (B <= 0 || C <= 0) return DR_EINVAL;
    if (!logits || !labels || !row_loss || !loss_out) return DR_EINVAL;
    hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3(dr_grid_for(B, 4)), dim3(256), 0, dr_s(stream), logits, labels, B, C,
                       inv_temperature, sample_weight, row_loss);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, dr_s(stream), row_loss, B, loss_out);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
