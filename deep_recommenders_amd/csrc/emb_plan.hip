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
//   3. SMALL   the shared-row slots are dropped into 128 row-range buckets (bucket = row * 128 / num_rows: monotone in the row, so
//              bucket order is row order; 128 counters instead of one -- 5 K same-address atomics cost 60 us on this chip).  If
//              no bucket overflows its 256 entries and m = their total <= the limit (uniform ids at config 3: m ~ 11 K), ONE
//              launch of 128 blocks sorts each bucket in LDS (bitonic; the composite key orders by row, then slot, so the arrival
//              order of the atomics does not matter -> bit-reproducible plan), writes it at its offset and emits the duplicate
//              pass's work list (segment heads).  The "sorted" arrays then hold ONLY the m shared-row slots; dup_count[1] = m
//              tells K4 how long they are.
//   4. LARGE   m beyond that (skewed / Zipf keys, where most slots share rows) or geometry the composite key cannot hold
//              (n > 2^24 slots, >= 2^31 - 1 rows): a plain LSD radix sort of ALL n slots by row (8-bit digits; per pass
//              per-block histograms -> one scan block per digit -> stable scatter), then the same head / flag marking over the
//              full list.  Every kernel of this path is always launched and returns at once unless the device-side switch says
//              LARGE (m lives on the device; the host never waits for it).
//
// Measured (tools/exp/plan_bench.py + rocprofv3, config 3's 1.7 M slots): the first version of this file -- one counter for the
// list, one 1024-thread block sorting it, a single-block histogram scan -- took 412 us (uniform) / 1300 us (Zipf) against
// rocPRIM's 167 / ~200: 111 us of it were 5.6 K same-address atomics, 80 us the one-block sort, 100 us PER PASS the single-block
// scan of 65 K histogram entries.  The claim kernel itself is 96 us (1.7 M returning device-scope atomics on random lines).
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
constexpr int SMALL_CAP = 16384;             // largest shared-row list the SMALL path takes
constexpr int NBK = 128, CAPB = 256;         // row-range buckets of the SMALL path and their capacity (NBK * CAPB >= SMALL_CAP)
constexpr int RADIX_BITS = 8, RADIX = 1 << RADIX_BITS;
constexpr int MAX_PASSES = 8;
constexpr int RADIX_CHUNK = 1664, RADIX_MAX_BLOCKS = 1024;
constexpr int MARK_BLOCKS = 512;

std::atomic<int> g_small_limit{SMALL_CAP};

struct Ctrl {            // device-side state of one plan build (zeroed at its start)
    int32_t large;       // != 0: the LARGE path runs
    int32_t overflow;    // hint: some bucket of the SMALL path is full (later slots stop counting)
    int32_t pad[2];
    int32_t bcnt[NBK];   // shared-row slots per row-range bucket
    uint32_t gtot[MAX_PASSES][RADIX];   // LARGE: digit totals of each radix pass
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
    int64_t nb = (n + RADIX_CHUNK - 1) / RADIX_CHUNK;
    return (int)(nb < 1 ? 1 : (nb > RADIX_MAX_BLOCKS ? RADIX_MAX_BLOCKS : nb));
}

struct Layout {
    size_t tab, entry, keys_x, vals_x, ghist, ctrl, total;
};
Layout layout_for(int64_t n) {
    Layout L{};
    size_t off = 0;
    L.tab = off;    off += align_up((size_t)table_entries(n) * 4, 256);
    L.entry = off;  off += align_up((size_t)n * 4, 256);
    L.keys_x = off; off += align_up((size_t)(n > NBK * CAPB ? n : NBK * CAPB) * 8, 256);   // SMALL: the bucket lists; LARGE: radix ping-pong keys
    L.vals_x = off; off += align_up((size_t)n * 4, 256);
    L.ghist = off;  off += align_up((size_t)RADIX * RADIX_MAX_BLOCKS * 4, 256);
    L.ctrl = off;   off += align_up(sizeof(Ctrl), 256);
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

// ---- 2. flags + the shared-row slots into their row-range buckets -----------------------------------------------------------
__global__ __launch_bounds__(256) void plan_flags_kernel(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ entry,
                                                         int32_t n, uint32_t bucket_mul, uint8_t* __restrict__ flags,
                                                         uint64_t* __restrict__ blist, Ctrl* __restrict__ ctrl) {
    const int32_t stride = gridDim.x * blockDim.x;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const uint32_t e = entry[p];
        const uint32_t v = e != EMPTY ? tab[e] : 0u;
        const bool dup = e != EMPTY && (v & DUPBIT);
        flags[p] = (e != EMPTY && !dup) ? 1 : 0;
        // (a full bucket means the LARGE path will redo everything: stop counting -- a Zipf batch would otherwise queue 1.5 M
        // atomics on 128 words)
        if (dup && *reinterpret_cast<volatile const int32_t*>(&ctrl->overflow) == 0) {
            const uint32_t r = v & ~DUPBIT;
            const int b = (int)(((uint64_t)r * bucket_mul) >> 32);                  // r * NBK / num_rows, monotone in r
            const int pos = atomicAdd(&ctrl->bcnt[b], 1);
            if (pos < CAPB) blist[b * CAPB + pos] = ((uint64_t)r << 24) | (uint32_t)p;
            else ctrl->overflow = 1;
        }
    }
}

// ---- 3. SMALL: one block per bucket sorts it in LDS, writes it at its offset, emits the duplicate pass's work list -----------------
__global__ __launch_bounds__(256) void plan_bucket_sort_kernel(const uint64_t* __restrict__ blist, Ctrl* __restrict__ ctrl,
                                                               int32_t small_limit, uint64_t* __restrict__ rows,
                                                               int32_t* __restrict__ slots, int32_t* __restrict__ dup_heads,
                                                               int32_t* __restrict__ dup_count) {
    __shared__ uint64_t s[CAPB];
    __shared__ int32_t cnts[NBK];
    __shared__ int32_t nheads, hbase;
    const int t = threadIdx.x, b = blockIdx.x;
    if (ctrl->large != 0) return;                                     // geometry forced the LARGE path
    if (t < NBK) cnts[t] = ctrl->bcnt[t];
    if (t == 0) nheads = 0;
    __syncthreads();
    int32_t m = 0, off = 0;
    bool over = false;
    for (int i = 0; i < NBK; ++i) {                                   // 128 LDS broadcasts: every thread gets the same totals
        const int32_t c = cnts[i];
        over |= c > CAPB;
        if (i < b) off += c;
        m += c;
    }
    if (over || m > small_limit) {                                    // every block reaches the same verdict
        if (b == 0 && t == 0) ctrl->large = 1;
        return;
    }
    const int cnt = cnts[b];
    int n2 = 64;
    while (n2 < cnt) n2 <<= 1;
    if (t < n2) s[t] = t < cnt ? blist[b * CAPB + t] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (t < (n2 >> 1)) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // the pair (i, i | j), i has bit j clear
                const int ixj = i | j;
                const uint64_t x = s[i], y = s[ixj];
                const bool asc = (i & k) == 0;
                if ((x > y) == asc) {
                    s[i] = y;
                    s[ixj] = x;
                }
            }
            __syncthreads();
        }
    }
    // every listed slot shares its row with at least one other, and a row lives in exactly one bucket: segments never straddle
    // blocks.  Positions are global (off + t): the 32-aligned cut points of hot rows are those of the concatenated list.
    const bool in = t < cnt;
    const int gi = off + t;
    bool head = false;
    if (in) {
        const uint64_t k = s[t] >> 24;
        rows[gi] = k;
        slots[gi] = (int32_t)(s[t] & 0xFFFFFFu);
        const bool seg_start = (t == 0) || ((s[t - 1] >> 24) != k);
        head = seg_start || ((gi % CH) == 0 && t >= CH && (s[t - CH] >> 24) == k);
    }
    const uint64_t bal = __ballot(head);
    const int lane = t & 63;
    int wbase = 0;
    if (bal != 0 && lane == __ffsll((unsigned long long)bal) - 1) wbase = atomicAdd(&nheads, __popcll(bal));
    __syncthreads();
    if (t == 0) hbase = nheads > 0 ? atomicAdd(&dup_count[0], nheads) : 0;        // ONE global atomic per bucket
    __syncthreads();
    if (bal != 0) {
        wbase = __shfl(wbase, __ffsll((unsigned long long)bal) - 1, 64);
        if (head) dup_heads[hbase + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = gi;
    }
    if (b == 0 && t == 0) dup_count[1] = m;                           // length of the sorted arrays K4's duplicate pass walks
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

__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* __restrict__ keys, int32_t n, int shift, int pass,
                                                         uint32_t* __restrict__ ghist, Ctrl* __restrict__ ctrl) {
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
    if (h[t] != 0) atomicAdd(&ctrl->gtot[pass][t], h[t]);            // digit totals: nb atomics per word, 256 words
}

// One block per digit d: ghist[d][0 .. nb) -> exclusive scan over the blocks + the number of keys with a smaller digit
// (digit-major order of the destination: all keys of smaller digits, then the same digit in earlier blocks).
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ ghist, int32_t nb, int pass,
                                                         const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    __shared__ uint32_t part[256];
    __shared__ uint32_t wsum[4];
    const int t = threadIdx.x, d = blockIdx.x, lane = t & 63, w = t >> 6;
    // base = sum of the totals of the digits below d
    uint32_t v = t < d ? ctrl->gtot[pass][t] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) wsum[w] = v;
    __syncthreads();
    const uint32_t base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    // this thread's run of the column
    const int32_t per = (nb + 255) / 256;
    const int32_t beg = t * per, end = beg + per < nb ? beg + per : nb;
    uint32_t* col = ghist + (int64_t)d * nb;
    uint32_t sum = 0;
    for (int32_t i = beg; i < end; ++i) sum += col[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                               // Hillis-Steele inclusive scan of the 256 partials
        const uint32_t x = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = base + part[t] - sum;
    for (int32_t i = beg; i < end; ++i) {
        const uint32_t c = col[i];
        col[i] = run;
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
// >= CH past it) goes to dup_heads.  A block owns a contiguous chunk, collects its heads in LDS and reserves their place with ONE
// global atomic (a Zipf batch has ~1e5 heads; one same-address atomic per wave cost 300 us, they retire at ~88 per microsecond).
constexpr int MARK_MAX_CHUNK = 8192;
__global__ __launch_bounds__(256) void radix_mark_kernel(const uint64_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                                         int32_t n, uint64_t num_rows, uint8_t* __restrict__ flags,
                                                         int32_t* __restrict__ dup_heads, int32_t* __restrict__ dup_count,
                                                         const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    __shared__ int32_t hl[MARK_MAX_CHUNK];
    __shared__ int32_t nh, gbase;
    const int t = threadIdx.x, lane = t & 63;
    const int32_t chunk = (n + (int32_t)gridDim.x - 1) / (int32_t)gridDim.x;        // <= MARK_MAX_CHUNK by the launch
    const int32_t beg = (int32_t)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    if (blockIdx.x == 0 && t == 0) dup_count[1] = n;
    if (t == 0) nh = 0;
    __syncthreads();
    for (int32_t i0 = beg; i0 < end; i0 += 256) {                     // block-uniform trip count (ballots below)
        const int32_t i = i0 + t;
        const bool in = i < end;
        const uint64_t k = in ? rows[i] : ~0ull;
        const bool valid = in && k < num_rows;
        const bool seg_start = in && ((i == 0) || (rows[i - 1] != k));
        const bool has_next = in && (i + 1 < n) && (rows[i + 1] == k);
        if (in) flags[slots[i]] = (valid && seg_start && !has_next) ? 1 : 0;
        const bool head = valid && ((seg_start && has_next) || (!seg_start && (i % CH) == 0 && i >= CH && rows[i - CH] == k));
        const uint64_t bal = __ballot(head);
        if (bal != 0) {
            const int leader = __ffsll((unsigned long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&nh, __popcll(bal));
            base = __shfl(base, leader, 64);
            if (head) hl[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        }
    }
    __syncthreads();
    if (t == 0) gbase = nh > 0 ? atomicAdd(dup_count, nh) : 0;
    __syncthreads();
    for (int j = t; j < nh; j += 256) dup_heads[gbase + j] = hl[j];
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
        const uint32_t bucket_mul = (uint32_t)((((uint64_t)NBK) << 32) / (uint64_t)num_rows);     // floor: row * mul >> 32 < NBK
        if (hipMemsetAsync(tab, 0xFF, (size_t)T * 4, s) != hipSuccess) return DR_ELAUNCH;
        hipLaunchKernelGGL(plan_claim_kernel, dim3(grid), dim3(256), 0, s, ids, n, F, row_base, tab, log_t, entry);
        hipLaunchKernelGGL(plan_flags_kernel, dim3(grid), dim3(256), 0, s, tab, entry, n, bucket_mul, unique_flags, keys_x, ctrl);
        hipLaunchKernelGGL(plan_bucket_sort_kernel, dim3(NBK), dim3(256), 0, s, keys_x, ctrl, (int32_t)g_small_limit.load(), rows_y,
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
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(256), 0, s, kbuf[cur], n, shift, p, ghist, ctrl);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(RADIX), dim3(256), 0, s, ghist, (int32_t)nb, p, ctrl);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(256), 0, s, kbuf[cur], vbuf[cur], kbuf[cur ^ 1], vbuf[cur ^ 1], n,
                           shift, ghist, ctrl);
        cur ^= 1;
    }
    int mark_blocks = MARK_BLOCKS;
    while ((n + mark_blocks - 1) / mark_blocks > MARK_MAX_CHUNK) mark_blocks *= 2;
    hipLaunchKernelGGL(radix_mark_kernel, dim3(mark_blocks), dim3(256), 0, s, rows_y, sorted_slots, n, (uint64_t)num_rows,
                       unique_flags, dup_heads, dup_count, ctrl);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
