// The slot plan of the deterministic K4 (dr_emb_sort_slots): which (example, field) slots own their table row alone this batch,
// and -- for the rows several slots share -- those slots grouped by row in a fixed order.  Hand-written for gfx950; replaces the
// general-purpose 64-bit rocPRIM radix sort of rounds 1-2 ("a library call on the hot path").
//
// Criteo-shaped batches are almost duplicate-free (uniform hashed ids: 99.7 % of the B*F slots are the only slot of their row),
// so sorting all 1.7 M (row, slot) pairs to find 5 K duplicates is the wrong algorithm.  Instead:
//   1. CLAIM   every slot inserts its row into an open-addressed table (2n .. 4n 32-bit entries, multiplicative hash, linear
//              probing): atomicCAS(EMPTY -> row) claims, a slot that finds its own row already there sets the entry's DUP bit.
//   2. FLAGS   slot p is unique  <=>  its entry's DUP bit is clear.  Slots of shared rows are appended (one atomic per wave) to a
//              short list of composite keys  row << 24 | slot.
//   3. SMALL   m = |list| <= 16384 (uniform ids at config 3: ~5.3 K): ONE block sorts the list in LDS (bitonic, the composite key
//              orders by row, then slot: the arrival order of step 2's atomics does not matter -> bit-reproducible plan) and emits
//              the duplicate pass's work list (segment heads).  The "sorted" arrays then hold ONLY the m shared-row slots;
//              dup_count[1] = m tells K4 how long they are.
//   4. LARGE   m beyond the LDS list (skewed / Zipf keys, where most slots share rows) or geometry the composite key cannot hold
//              (n > 2^24 slots, >= 2^31 - 1 rows): a plain LSD radix sort of ALL n slots by row (8-bit digits; per pass
//              histogram -> single-block scan -> stable scatter), then the same head / flag marking over the full list.  Every
//              kernel of this path is always launched and returns at once unless the device-side switch says LARGE (m lives on
//              the device; the host never waits for it).
//
// Replaces the ordering half of the autodiff of [TF] safe_embedding_lookup_sparse (IndexedSlices -> unsorted_segment_sum into the
// variable) reached from optimizer.minimize (examples/train_fm_on_movielens_estimator.py:51-52, reference root); K4 itself is
// csrc/emb_sorted.hip.
#include "dr_common.h"
#include <atomic>

namespace {

constexpr int CH = 32;                       // must equal emb_sorted.hip's CH (piece length of hot rows)
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t DUPBIT = 0x80000000u;
constexpr int SMALL_CAP = 16384;             // composite keys one block sorts in LDS (128 KB)
constexpr int RADIX_BITS = 8, RADIX = 1 << RADIX_BITS;

std::atomic<int> g_small_limit{SMALL_CAP};

struct Ctrl {            // device-side state of one plan build
    int32_t m;           // shared-row slots found by the claim pass
    int32_t large;       // != 0: the LARGE path runs
    int32_t pad[2];
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

unsigned bits_for(uint64_t v) {
    unsigned b = 1;
    while (b < 64 && (v >> b) != 0) ++b;
    return b;
}

int64_t table_entries(int64_t n) {
    int64_t t = 1024;
    while (t < 2 * n) t <<= 1;
    return t;
}

int radix_blocks(int64_t n) {
    int64_t nb = (n + 2047) / 2048;
    return (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
}

struct Layout {
    size_t tab, entry, keys_x, vals_x, ghist, ctrl, total;
};
Layout layout_for(int64_t n) {
    Layout L{};
    size_t off = 0;
    L.tab = off;    off += align_up((size_t)table_entries(n) * 4, 256);
    L.entry = off;  off += align_up((size_t)n * 4, 256);
    L.keys_x = off; off += align_up((size_t)n * 8, 256);       // SMALL: the composite-key list; LARGE: radix ping-pong keys
    L.vals_x = off; off += align_up((size_t)n * 4, 256);
    L.ghist = off;  off += align_up((size_t)RADIX * 256 * 4, 256);
    L.ctrl = off;   off += 256;
    L.total = off;
    return L;
}

// ---- 1. claim -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plan_claim_kernel(const int64_t* __restrict__ ids, int32_t n, int32_t F,
                                                         const int64_t* __restrict__ row_base, uint32_t* __restrict__ tab,
                                                         int32_t log_t, uint32_t* __restrict__ entry) {
    const uint32_t mask = (1u << log_t) - 1u;
    const int32_t stride = gridDim.x * blockDim.x;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const int64_t id = ids[p];
        if (id < 0) {
            entry[p] = EMPTY;
            continue;
        }
        const uint32_t r = (uint32_t)(row_base[p % F] + id);
        uint32_t h = (r * 2654435761u) >> (32 - log_t);
        for (;;) {
            const uint32_t old = atomicCAS(&tab[h], EMPTY, r);
            if (old == EMPTY) break;                                  // claimed: first slot of this row (so far the only one)
            if ((old & ~DUPBIT) == r) {                               // the row is already there: shared
                if (!(old & DUPBIT)) atomicOr(&tab[h], DUPBIT);
                break;
            }
            h = (h + 1) & mask;
        }
        entry[p] = h;
    }
}

// ---- 2. flags + the list of shared-row slots --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plan_flags_kernel(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ entry,
                                                         int32_t n, uint8_t* __restrict__ flags, uint64_t* __restrict__ list,
                                                         Ctrl* __restrict__ ctrl) {
    const int32_t stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    const int32_t n_up = (n + 63) & ~63;                              // whole waves stay in the loop (ballot below)
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_up; p += stride) {
        const bool in = p < n;
        const uint32_t e = in ? entry[p] : EMPTY;
        const uint32_t v = e != EMPTY ? tab[e] : 0u;
        const bool dup = e != EMPTY && (v & DUPBIT);
        if (in) flags[p] = (e != EMPTY && !dup) ? 1 : 0;
        const uint64_t bal = __ballot(dup);
        if (bal != 0) {
            const int leader = __ffsll((unsigned long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&ctrl->m, __popcll(bal));
            base = __shfl(base, leader, 64);
            if (dup) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((uint64_t)(v & ~DUPBIT) << 24) | (uint32_t)p;
        }
    }
}

// ---- 3. SMALL: one block sorts the list in LDS and emits the duplicate pass's work list ------------------------------------------
__global__ __launch_bounds__(1024) void plan_small_sort_kernel(const uint64_t* __restrict__ list, Ctrl* __restrict__ ctrl,
                                                               int32_t small_limit, uint64_t* __restrict__ rows,
                                                               int32_t* __restrict__ slots, int32_t* __restrict__ dup_heads,
                                                               int32_t* __restrict__ dup_count) {
    __shared__ uint64_t s[SMALL_CAP];
    __shared__ int32_t nheads;
    const int t = threadIdx.x;
    if (ctrl->large != 0) return;                                     // geometry forced the LARGE path
    const int32_t m = ctrl->m;
    if (m > small_limit) {
        if (t == 0) ctrl->large = 1;
        return;
    }
    int n2 = 1024;
    while (n2 < m) n2 <<= 1;
    for (int i = t; i < n2; i += 1024) s[i] = i < m ? list[i] : ~0ull;
    if (t == 0) nheads = 0;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = t; q < (n2 >> 1); q += 1024) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));   // the pair (i, i | j), i has bit j clear
                const int ixj = i | j;
                const uint64_t a = s[i], b = s[ixj];
                const bool asc = (i & k) == 0;
                if ((a > b) == asc) {
                    s[i] = b;
                    s[ixj] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i0 = 0; i0 < n2; i0 += 1024) {                           // uniform trip count: the ballots below need whole waves
        const int i = i0 + t;
        const bool in = i < m;
        const uint64_t key = in ? s[i] : 0ull;
        const uint64_t k = key >> 24;
        bool head = false;
        if (in) {
            rows[i] = k;
            slots[i] = (int32_t)(key & 0xFFFFFFu);
            const bool seg_start = (i == 0) || ((s[i - 1] >> 24) != k);
            // every listed slot shares its row with at least one other: a segment start always has a successor
            head = seg_start || ((i % CH) == 0 && i >= CH && (s[i - CH] >> 24) == k);
        }
        const uint64_t bal = __ballot(head);
        if (bal != 0) {
            const int lane = t & 63;
            const int leader = __ffsll((unsigned long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&nheads, __popcll(bal));
            base = __shfl(base, leader, 64);
            if (head) dup_heads[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        }
    }
    __syncthreads();
    if (t == 0) {
        dup_count[0] = nheads;
        dup_count[1] = m;                                             // length of the sorted arrays K4's duplicate pass walks
    }
}

// ---- 4. LARGE: LSD radix sort of all n (row, slot) pairs + marking --------------------------------------------------------------
__global__ __launch_bounds__(256) void radix_make_keys_kernel(const int64_t* __restrict__ ids, int32_t n, int32_t F,
                                                              const int64_t* __restrict__ row_base, uint64_t sentinel,
                                                              uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                              const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    const int32_t stride = gridDim.x * blockDim.x;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const int64_t id = ids[p];
        keys[p] = id >= 0 ? (uint64_t)(row_base[p % F] + id) : sentinel;   // missing ids sort to the end
        vals[p] = (uint32_t)p;
    }
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* __restrict__ keys, int32_t n, int shift,
                                                         uint32_t* __restrict__ ghist, const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    __shared__ uint32_t h[RADIX];
    const int nb = gridDim.x, t = threadIdx.x;
    const int32_t chunk = (n + nb - 1) / nb;
    const int32_t beg = (int32_t)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    h[t] = 0;
    __syncthreads();
    for (int32_t i = beg + t; i < end; i += 256) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (RADIX - 1)], 1u);
    __syncthreads();
    ghist[t * nb + blockIdx.x] = h[t];
}

// exclusive scan of ghist[RADIX * nb] in place (digit-major: all keys of smaller digits, then the same digit in earlier blocks)
__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t* __restrict__ ghist, int32_t total,
                                                          const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int32_t per = (total + 1023) / 1024;
    const int32_t beg = t * per, end = beg + per < total ? beg + per : total;
    uint32_t sum = 0;
    for (int32_t i = beg; i < end; ++i) sum += ghist[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                              // Hillis-Steele inclusive scan of the 1024 partials
        const uint32_t v = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;                                     // exclusive prefix of this thread's range
    for (int32_t i = beg; i < end; ++i) {
        const uint32_t c = ghist[i];
        ghist[i] = run;
        run += c;
    }
}

// stable scatter: block b re-reads its chunk in order, 256 keys at a time; a key's destination = (scanned histogram entry of its
// digit for this block, advanced by the tiles already written) + the number of EARLIER keys of the tile with the same digit
// (wave-level match by eight ballots, earlier waves' counts through LDS)
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                            uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, int32_t n,
                                                            int shift, const uint32_t* __restrict__ ghist,
                                                            const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    __shared__ uint32_t running[RADIX];
    __shared__ uint32_t wcnt[4][RADIX];
    const int nb = gridDim.x, t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int32_t chunk = (n + nb - 1) / nb;
    const int32_t beg = (int32_t)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    running[t] = ghist[t * nb + blockIdx.x];
    __syncthreads();
    for (int32_t base = beg; base < end; base += 256) {
        const int32_t i = base + t;
        const bool live = i < end;
        const uint64_t k = live ? kin[i] : 0ull;
        const uint32_t v = live ? vin[i] : 0u;
        const uint32_t d = (uint32_t)(k >> shift) & (RADIX - 1);
        wcnt[0][t] = 0; wcnt[1][t] = 0; wcnt[2][t] = 0; wcnt[3][t] = 0;
        __syncthreads();
        uint64_t mask = __ballot(live);                               // lanes with my digit (dead lanes never match)
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t vote = __ballot(bit);
            mask &= bit ? vote : ~vote;
        }
        const uint32_t rank_in_wave = __popcll(mask & ((1ull << lane) - 1ull));
        if (live && rank_in_wave == 0) wcnt[w][d] = __popcll(mask);
        __syncthreads();
        uint32_t pos = 0;
        if (live) {
            pos = running[d] + rank_in_wave;
            for (int ww = 0; ww < w; ++ww) pos += wcnt[ww][d];
        }
        __syncthreads();                                              // every read of running[] precedes its update
        running[t] += wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
        if (live) {
            kout[pos] = k;
            vout[pos] = v;
        }
        __syncthreads();
    }
}

// LARGE path's marking over the fully sorted list: flag[p] = 1 iff slot p is the only slot of the batch that touches its row;
// every sorted position that heads a piece of a row touched by >= 2 slots (the segment start, plus each CH-aligned position
// >= CH past it) is appended to dup_heads -- ONE atomic per wave (a skewed batch has ~1e5 heads and same-address atomics retire
// at ~88 per microsecond).
__global__ __launch_bounds__(256) void radix_mark_kernel(const uint64_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                                         int32_t n, uint64_t num_rows, uint8_t* __restrict__ flags,
                                                         int32_t* __restrict__ dup_heads, int32_t* __restrict__ dup_count,
                                                         const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    const int32_t stride = gridDim.x * blockDim.x;
    const int32_t n_up = (n + 63) & ~63;
    if (blockIdx.x == 0 && threadIdx.x == 0) dup_count[1] = n;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += stride) {
        const bool in = i < n;
        const uint64_t k = in ? rows[i] : ~0ull;
        const bool valid = in && k < num_rows;
        const bool seg_start = in && ((i == 0) || (rows[i - 1] != k));
        const bool has_next = in && (i + 1 < n) && (rows[i + 1] == k);
        if (in) flags[slots[i]] = (valid && seg_start && !has_next) ? 1 : 0;
        const bool head = valid && ((seg_start && has_next) || (!seg_start && (i % CH) == 0 && i >= CH && rows[i - CH] == k));
        const uint64_t bal = __ballot(head);
        if (bal != 0) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((unsigned long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(dup_count, __popcll(bal));
            base = __shfl(base, leader, 64);
            if (head) dup_heads[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        }
    }
}

}  // namespace

// Test / tuning hook: the largest shared-row list the one-block LDS sort takes (default and maximum 16384); 0 sends every batch
// that has any shared row down the LARGE (radix sort) path.  Process-wide; returns the previous value.
extern "C" int32_t dr_emb_plan_set_small_limit(int32_t limit) {
    if (limit < 0) limit = 0;
    if (limit > SMALL_CAP) limit = SMALL_CAP;
    return g_small_limit.exchange(limit);
}

extern "C" int64_t dr_emb_sort_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return (int64_t)layout_for(n).total;
}

extern "C" int dr_emb_sort_slots(const int64_t* ids, int64_t B, int32_t F, const int64_t* row_base, int64_t num_rows,
                                 int64_t* sorted_rows, int32_t* sorted_slots, uint8_t* unique_flags, int32_t* dup_heads,
                                 int32_t* dup_count, void* workspace, int64_t workspace_bytes, dr_stream_t stream) {
    if (B < 0 || F <= 0 || num_rows <= 0) return DR_EINVAL;
    const int64_t n64 = B * F;
    if (n64 == 0) return DR_OK;
    if (n64 > 0x7fffff00) return DR_EINVAL;
    if (!ids || !row_base || !sorted_rows || !sorted_slots || !unique_flags || !dup_heads || !dup_count || !workspace)
        return DR_EINVAL;
    if (workspace_bytes < dr_emb_sort_workspace_bytes(n64)) return DR_EINVAL;
    const int32_t n = (int32_t)n64;
    const Layout L = layout_for(n64);
    char* w = static_cast<char*>(workspace);
    uint32_t* tab = reinterpret_cast<uint32_t*>(w + L.tab);
    uint32_t* entry = reinterpret_cast<uint32_t*>(w + L.entry);
    uint64_t* keys_x = reinterpret_cast<uint64_t*>(w + L.keys_x);
    uint32_t* vals_x = reinterpret_cast<uint32_t*>(w + L.vals_x);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(w + L.ghist);
    Ctrl* ctrl = reinterpret_cast<Ctrl*>(w + L.ctrl);
    uint64_t* rows_y = reinterpret_cast<uint64_t*>(sorted_rows);
    uint32_t* slots_y = reinterpret_cast<uint32_t*>(sorted_slots);
    hipStream_t s = dr_s(stream);
    const int grid = dr_grid_for(n, 256);

    if (hipMemsetAsync(ctrl, 0, sizeof(Ctrl), s) != hipSuccess) return DR_ELAUNCH;
    if (hipMemsetAsync(dup_count, 0, 2 * sizeof(int32_t), s) != hipSuccess) return DR_ELAUNCH;
    // the composite key holds 24 bits of slot and the table 31 bits of row; anything larger sorts all slots
    const bool claimable = n64 <= (1 << 24) && num_rows < 0x7fffffffLL;
    if (claimable) {
        const int64_t T = table_entries(n64);
        int log_t = 0;
        while ((1ll << log_t) < T) ++log_t;
        if (hipMemsetAsync(tab, 0xFF, (size_t)T * 4, s) != hipSuccess) return DR_ELAUNCH;
        hipLaunchKernelGGL(plan_claim_kernel, dim3(grid), dim3(256), 0, s, ids, n, F, row_base, tab, log_t, entry);
        hipLaunchKernelGGL(plan_flags_kernel, dim3(grid), dim3(256), 0, s, tab, entry, n, unique_flags, keys_x, ctrl);
        hipLaunchKernelGGL(plan_small_sort_kernel, dim3(1), dim3(1024), 0, s, keys_x, ctrl, (int32_t)g_small_limit.load(), rows_y,
                           sorted_slots, dup_heads, dup_count);
    } else {
        if (hipMemsetAsync(&ctrl->large, 1, sizeof(int32_t), s) != hipSuccess) return DR_ELAUNCH;
    }
    // LARGE path (each kernel returns at once unless ctrl->large): the final pass must land in the output arrays
    const unsigned bits = bits_for((uint64_t)num_rows);            // the sentinel == num_rows needs these bits too
    const int passes = (int)((bits + RADIX_BITS - 1) / RADIX_BITS);
    const int nb = radix_blocks(n);
    uint64_t* kbuf[2] = {keys_x, rows_y};
    uint32_t* vbuf[2] = {vals_x, slots_y};
    int cur = (passes & 1) ? 0 : 1;                                // passes odd: start in the workspace, end in the outputs
    hipLaunchKernelGGL(radix_make_keys_kernel, dim3(grid), dim3(256), 0, s, ids, n, F, row_base, (uint64_t)num_rows, kbuf[cur],
                       vbuf[cur], ctrl);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * RADIX_BITS;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(256), 0, s, kbuf[cur], n, shift, ghist, ctrl);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, s, ghist, (int32_t)(RADIX * nb), ctrl);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(256), 0, s, kbuf[cur], vbuf[cur], kbuf[cur ^ 1], vbuf[cur ^ 1], n,
                           shift, ghist, ctrl);
        cur ^= 1;
    }
    hipLaunchKernelGGL(radix_mark_kernel, dim3(grid), dim3(256), 0, s, rows_y, sorted_slots, n, (uint64_t)num_rows, unique_flags,
                       dup_heads, dup_count, ctrl);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
