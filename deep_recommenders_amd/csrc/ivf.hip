// IVF-Flat (inverted file, inner product) approximate top-K -- the GPU counterpart of the reference's `Faiss` index
// (keras/models/retrieval/factorized_top_k.py:337-461: faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT),
// searcher.nprobe = nprobe).  faiss-cpu 1.6.3 is a third-party dependency that is absent from the reference tree and from
// this image; the algorithm restated here is the published one: vectors are assigned to the coarse centroid with the
// largest inner product, a query scans the `nprobe` lists whose centroids score highest, exactly, and keeps the top k.
// (Centroid TRAINING is host-orchestrated in deep_recommenders_amd/keras/models/retrieval/factorized_top_k.py.)
//
// Layout: list l owns blocks [blk_off[l], blk_off[l+1]) of 64 vectors; a block stores its vectors dimension-major,
// packed[(blk * D + d) * 64 + lane], so the scan -- one wavefront per query, one lane per vector -- reads 256 contiguous
// bytes per instruction; the query sits in LDS and is broadcast.  The running top-k is the wave-wide register list of
// topk_list.h.  HBM/L2-bound: every probed vector is read once per query that probes its list.
#include "dr_common.h"
#include "topk_list.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void ivf_pack_kernel(const float* __restrict__ cand, int64_t N, int32_t D,
                                                       const int64_t* __restrict__ order,
                                                       const int64_t* __restrict__ list_start,
                                                       const int64_t* __restrict__ blk_off, int32_t nlist,
                                                       const int64_t* __restrict__ ids, float* __restrict__ packed,
                                                       int64_t* __restrict__ packed_ids) {
    const int64_t total = blk_off[nlist] * 64;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
        const int64_t blk = s >> 6;
        const int lane = (int)(s & 63);
        int lo = 0, hi = nlist;                  // list of this block: largest l with blk_off[l] <= blk
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (blk_off[mid] <= blk) lo = mid; else hi = mid;
        }
        const int64_t pos = (blk - blk_off[lo]) * 64 + lane;
        const int64_t cnt = list_start[lo + 1] - list_start[lo];
        int64_t src = -1;
        if (pos < cnt) src = order[list_start[lo] + pos];
        packed_ids[s] = src < 0 ? -1 : (ids != nullptr ? ids[src] : src);
        float* dst = packed + blk * (int64_t)D * 64 + lane;
        if (src >= 0) {
            const float* row = cand + src * D;
            for (int d = 0; d < D; ++d) dst[(int64_t)d * 64] = row[d];
        } else {
            for (int d = 0; d < D; ++d) dst[(int64_t)d * 64] = 0.f;
        }
    }
}


// ---- list build: stable counting sort of the N vectors by their coarse assignment (round 4: torch.argsort / bincount / cumsum until
// then -- the `index.add(candidates)` half of faiss' IVF build, factorized_top_k.py:374-391).  Stable = inside a list the vectors keep
// their input order, so equal scores tie on the lower candidate number exactly as the exact search does.
//   pass 1  per block (IVB_CHUNK consecutive vectors): LDS histogram of its list ids -> block_hist[list][block]
//   pass 2  ONE block: exclusive scan over (list-major, block-minor); list_start[l] = start of (l, block 0)
//   pass 3  per block again, 256 vectors a round: rank inside the round by comparing against the earlier keys of the round (LDS),
//           plus what earlier rounds of the block placed in the same list (LDS counters), plus the block's scanned offset
constexpr int IVB_CHUNK = 2048;          // vectors per block
constexpr int IVB_MAXLIST = 8192;        // list ids an LDS histogram holds (32 KB)

__global__ __launch_bounds__(256) void ivf_build_hist_kernel(const int64_t* __restrict__ assign, int64_t N, int32_t nlist,
                                                             int32_t nblk, int64_t* __restrict__ block_hist) {
    __shared__ int hist[IVB_MAXLIST];
    for (int l = threadIdx.x; l < nlist; l += 256) hist[l] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * IVB_CHUNK;
    for (int j = threadIdx.x; j < IVB_CHUNK; j += 256) {
        const int64_t i = base + j;
        if (i < N) {
            const int64_t a = assign[i];
            if (a >= 0 && a < nlist) atomicAdd(&hist[(int)a], 1);
        }
    }
    __syncthreads();
    for (int l = threadIdx.x; l < nlist; l += 256) block_hist[(int64_t)l * nblk + blockIdx.x] = hist[l];
}

__global__ __launch_bounds__(256) void ivf_build_scan_kernel(int64_t* __restrict__ block_hist, int64_t total, int32_t nlist, int32_t nblk,
                                                             int64_t* __restrict__ list_start) {
    __shared__ int64_t part[256];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    // 8 consecutive entries per thread and round: 2048 entries per round
    for (int64_t base = 0; base < total; base += 2048) {
        int64_t v[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t i = base + (int64_t)threadIdx.x * 8 + j;
            v[j] = i < total ? block_hist[i] : 0;
            sum += v[j];
        }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int64_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        int64_t run = carry + part[threadIdx.x] - sum;                  // exclusive prefix of this thread's 8 entries
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t i = base + (int64_t)threadIdx.x * 8 + j;
            if (i < total) {
                block_hist[i] = run;
                if (i % nblk == 0) list_start[i / nblk] = run;
            }
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) list_start[nlist] = carry;
}

__global__ __launch_bounds__(256) void ivf_build_scatter_kernel(const int64_t* __restrict__ assign, int64_t N, int32_t nlist, int32_t nblk,
                                                                const int64_t* __restrict__ block_off, int64_t* __restrict__ order) {
    __shared__ int placed[IVB_MAXLIST];          // vectors of this block already placed in each list (earlier rounds)
    __shared__ int keys[256];
    for (int l = threadIdx.x; l < nlist; l += 256) placed[l] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * IVB_CHUNK;
    for (int r = 0; r < IVB_CHUNK / 256; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        int key = -1;
        if (i < N) {
            const int64_t a = assign[i];
            if (a >= 0 && a < nlist) key = (int)a;
        }
        keys[threadIdx.x] = key;
        __syncthreads();
        int rank = 0;
        if (key >= 0) {
            for (int t = 0; t < (int)threadIdx.x; ++t) rank += keys[t] == key ? 1 : 0;
            order[block_off[(int64_t)key * nblk + blockIdx.x] + placed[key] + rank] = i;
        }
        __syncthreads();
        if (key >= 0) atomicAdd(&placed[key], 1);
        __syncthreads();
    }
}

template <int DU>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const float* __restrict__ q, int64_t Bq, int32_t D,
                                                       const int64_t* __restrict__ probes, int32_t nprobe,
                                                       const int64_t* __restrict__ blk_off,
                                                       const float* __restrict__ packed,
                                                       const int64_t* __restrict__ packed_ids, int32_t k,
                                                       float* __restrict__ out_s, int64_t* __restrict__ out_i) {
    extern __shared__ float qs[];                       // [4 waves][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    const int64_t rowc = row < Bq ? row : Bq - 1;      // surplus waves mirror the last query (no stores)
    float* myq = qs + wave * D;
    for (int d = lane; d < D; d += 64) myq[d] = q[rowc * D + d];
    __syncthreads();
    drtk::List L;
    L.init(k, lane);
    for (int p = 0; p < nprobe; ++p) {
        const int64_t l = probes[rowc * nprobe + p];
        if (l < 0) continue;
        const int64_t b0 = blk_off[l], b1 = blk_off[l + 1];
        for (int64_t blk = b0; blk < b1; ++blk) {
            const float* vp = packed + blk * (int64_t)D * 64 + lane;
            float acc = 0.f;
            int d = 0;
            for (; d + DU <= D; d += DU) {
                float x[DU];
#pragma unroll
                for (int u = 0; u < DU; ++u) x[u] = vp[(int64_t)(d + u) * 64];
#pragma unroll
                for (int u = 0; u < DU; ++u) acc = fmaf(myq[d + u], x[u], acc);
            }
            for (; d < D; ++d) acc = fmaf(myq[d], vp[(int64_t)d * 64], acc);
            const int64_t id = packed_ids[blk * 64 + lane];
            L.offer(acc, id, id >= 0);
        }
    }
    if (row < Bq) L.store(out_s + row * k, out_i + row * k);
}

}  // namespace

extern "C" int dr_ivf_pack(const float* cand, int64_t N, int32_t D, const int64_t* order, const int64_t* list_start,
                           const int64_t* blk_off, int32_t nlist, int64_t total_blocks, const int64_t* ids, float* packed,
                           int64_t* packed_ids, dr_stream_t stream) {
    if (N < 0 || D <= 0 || nlist <= 0 || total_blocks < 0) return DR_EINVAL;
    if (total_blocks == 0) return DR_OK;
    if (!cand || !order || !list_start || !blk_off || !packed || !packed_ids) return DR_EINVAL;
    hipLaunchKernelGGL(ivf_pack_kernel, dim3(dr_grid_for(total_blocks * 64, 256, 4096)), dim3(256), 0, dr_s(stream), cand, N, D,
                       order, list_start, blk_off, nlist, ids, packed, packed_ids);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int dr_ivf_scan(const float* q, int64_t Bq, int32_t D, const int64_t* probes, int32_t nprobe,
                           const int64_t* blk_off, const float* packed, const int64_t* packed_ids, int32_t k,
                           float* out_scores, int64_t* out_index, dr_stream_t stream) {
    if (Bq < 0 || D <= 0 || D > 4096 || nprobe <= 0 || k <= 0 || k > drtk::KMAX) return DR_EINVAL;
    if (Bq == 0) return DR_OK;
    if (!q || !probes || !blk_off || !packed || !packed_ids || !out_scores || !out_index) return DR_EINVAL;
    const unsigned grid = (unsigned)((Bq + 3) / 4);
    hipLaunchKernelGGL((ivf_scan_kernel<8>), dim3(grid), dim3(256), 4 * D * sizeof(float), dr_s(stream), q, Bq, D, probes, nprobe,
                       blk_off, packed, packed_ids, k, out_scores, out_index);
    DR_CHECK_LAUNCH();
    return DR_OK;
}

extern "C" int64_t dr_ivf_build_workspace_bytes(int64_t N, int32_t nlist) {
    const int64_t nblk = (N + IVB_CHUNK - 1) / IVB_CHUNK;
    return (nblk > 0 ? nblk : 1) * (int64_t)(nlist > 0 ? nlist : 1) * (int64_t)sizeof(int64_t);
}

// Groups the N vectors by coarse list: order[list_start[l] .. list_start[l + 1]) = the vectors assigned to list l, ascending (stable);
// an assignment outside [0, nlist) drops its vector.  workspace: dr_ivf_build_workspace_bytes(N, nlist).  nlist <= 8192.
extern "C" int dr_ivf_build_lists(const int64_t* assign, int64_t N, int32_t nlist, int64_t* order, int64_t* list_start, void* workspace,
                                  int64_t workspace_bytes, dr_stream_t stream) {
    if (N < 0 || nlist <= 0 || nlist > IVB_MAXLIST) return DR_EINVAL;
    if (!list_start) return DR_EINVAL;
    if (N == 0) return hipMemsetAsync(list_start, 0, sizeof(int64_t) * (nlist + 1), dr_s(stream)) == hipSuccess ? DR_OK : DR_ELAUNCH;
    if (!assign || !order || !workspace || workspace_bytes < dr_ivf_build_workspace_bytes(N, nlist)) return DR_EINVAL;
    const int64_t nblk64 = (N + IVB_CHUNK - 1) / IVB_CHUNK;
    if (nblk64 > 0x7fffffff) return DR_EINVAL;
    const int nblk = (int)nblk64;
    int64_t* bh = static_cast<int64_t*>(workspace);
    hipLaunchKernelGGL(ivf_build_hist_kernel, dim3(nblk), dim3(256), 0, dr_s(stream), assign, N, nlist, nblk, bh);
    hipLaunchKernelGGL(ivf_build_scan_kernel, dim3(1), dim3(256), 0, dr_s(stream), bh, (int64_t)nblk * nlist, nlist, nblk, list_start);
    hipLaunchKernelGGL(ivf_build_scatter_kernel, dim3(nblk), dim3(256), 0, dr_s(stream), assign, N, nlist, nblk, bh, order);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
