// The slot plan of the deterministic K4 (dr_emb_sort_slots): which (example, field) slots own their table row alone this batch,
// and -- for the rows several slots share -- those slots grouped by row in a fixed order.  Hand-written for gfx950; replaces the
// general-purpose 64-bit rocPRIM radix sort of rounds 1-2 ("a library call on the hot path").
//
// Criteo-shaped batches are almost duplicate-free (uniform hashed ids: 99.3 % of the B*F slots are the only slot of their row),
// so sorting all 1.7 M (row, slot) pairs four digits deep to find 11 K duplicates is the wrong algorithm.  Instead:
//   SMALL  1. PARTITION  one counting-sort pass with the digit "row * 256 / num_rows" splits the slots into 256 row-range partitions
//                        of ~6.6 K slots: plan_front_kernel (composite keys + per-block digit counts -- and, through
//                        dr_hash_sort_slots, K1's hash and the field-major int32 ids of the same batch in the same kernel) ->
//                        radix_scan_kernel -> plan_scatter_kernel (stable, 1024-thread blocks).
//          2. CLAIM      one block per partition inserts its rows into an open-addressed table IN LDS (32 K entries, ds_cmpst):
//                        a slot that finds its row already there sets the entry's DUP bit.  A second sweep over the partition
//                        writes flag[slot] = "DUP bit clear" and collects the shared-row slots (composite key row << 24 | slot)
//                        in the partition's bucket.  No device-scope atomics on data: the first version of this file claimed
//                        rows in a GLOBAL table -- 1.7 M returning atomics on random lines cost 96 us, 11 K appends to
//                        128 counters (4 cache lines) another 100+ (tools/exp/plan_bench.py under rocprofv3).
//          3. SORT       if no bucket overflows and the m shared-row slots are <= the limit (uniform ids at config 3: m ~ 11 K),
//                        ONE launch of 256 blocks sorts each bucket in LDS (bitonic; the composite key orders by row, then slot,
//                        so nothing depends on arrival order -> bit-reproducible plan), writes it at its offset (partitions are
//                        row ranges: bucket order is row order) and emits the duplicate pass's work list (segment heads).  The
//                        "sorted" arrays then hold ONLY the m shared-row slots; dup_count[1] = m tells K4 how long they are.
//   LARGE  m beyond that or a partition beyond its LDS table (skewed / Zipf keys, where most slots share rows): ONE more launch
//          (plan_large_kernel, round 6), always made, that returns at once unless the device-side switch says LARGE (the verdict
//          lives on the device; the host never waits for it): block b sorts partition b -- the partitions are row ranges, so the
//          concatenation of the sorted partitions is the list sorted by row -- with a block-local LSD radix sort over the bits in
//          which the partition's rows differ, then marks flags and segment heads over the full list.  Rounds 3-5 ran a chip-wide
//          LSD radix sort of all n slots here: 14 more launches that were always made and gated on the device; on uniform ids each
//          of them was an empty kernel -- and an empty kernel on the side stream still costs the training stream ~2.7 us (the
//          dispatcher has to find it a CU between the GEMMs' blocks): 38 us of a 1.06 ms step (tools/exp/exp_plan.sh).
//          Geometry the composite key cannot hold (n > 2^24 slots, >= 2^31 - 1 rows; known on the host): the chip-wide radix sort
//          of ALL n slots by row (8-bit digits; per pass per-block histograms -> one scan block per digit -> stable scatter), then
//          the same head / flag marking over the full list.
//
// Replaces the ordering half of the autodiff of [TF] safe_embedding_lookup_sparse (IndexedSlices -> unsorted_segment_sum into the
// variable) reached from optimizer.minimize (examples/train_fm_on_movielens_estimator.py:51-52, reference root); K4 itself is
// csrc/emb_sorted.hip.
#include "dr_common.h"
#include "hash_i64.h"
#include <atomic>

namespace {

constexpr int CH = 32;                       // must equal emb_sorted.hip's CH (piece length of hot rows)
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t DUPBIT = 0x80000000u;
constexpr int SMALL_CAP = 16384;             // largest shared-row list the SMALL path takes
constexpr int RADIX_BITS = 8, RADIX = 1 << RADIX_BITS;
constexpr int NBK = RADIX, CAPB = 128;       // row-range partitions (= buckets of the shared-row list) and a bucket's capacity
constexpr int LDS_TAB_MAX_LOG = 15;          // a partition's claim table: 2^12 .. 2^15 entries (16 .. 128 KB of LDS), sized so that
                                             // the MEAN partition loads it to <= 0.5; a partition beyond load 0.7 sends the batch to LARGE
constexpr int RADIX_CHUNK = 1664, RADIX_MAX_BLOCKS = 1024;
constexpr int MARK_BLOCKS = 512, MARK_MAX_CHUNK = 8192;
static_assert(NBK * CAPB >= SMALL_CAP, "buckets must hold the largest SMALL list");

std::atomic<int> g_small_limit{SMALL_CAP};

struct Ctrl {            // device-side state of one plan build (zeroed at its start)
    int32_t large;       // != 0: the LARGE path runs
    int32_t pad[3];
    int32_t bcnt[NBK];   // shared-row slots per partition
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

unsigned bits_for(uint64_t v) {
    unsigned b = 1;
    while (b < 64 && (v >> b) != 0) ++b;
    return b;
}

__device__ __forceinline__ unsigned bits_for_dev(uint32_t v) { return v == 0 ? 1u : 32u - (unsigned)__clz((int)v); }

int radix_blocks(int64_t n) {
    int64_t nb = (n + RADIX_CHUNK - 1) / RADIX_CHUNK;
    return (int)(nb < 1 ? 1 : (nb > RADIX_MAX_BLOCKS ? RADIX_MAX_BLOCKS : nb));
}

struct Layout {
    size_t keys_x, vals_x, blist, ghist, tot, ctrl, total;
};
Layout layout_for(int64_t n) {
    Layout L{};
    size_t off = 0;
    L.keys_x = off; off += align_up((size_t)n * 8, 256);          // partition pass output / radix ping-pong keys
    L.vals_x = off; off += align_up((size_t)n * 4, 256);
    L.blist = off;  off += align_up((size_t)NBK * CAPB * 8, 256);  // the shared-row slots, one bucket per partition
    L.ghist = off;  off += align_up((size_t)RADIX * RADIX_MAX_BLOCKS * 4, 256);
    L.tot = off;    off += align_up((size_t)RADIX * 4, 256);       // digit totals of the current pass
    L.ctrl = off;   off += align_up(sizeof(Ctrl), 256);
    L.total = off;
    return L;
}

// digit of a key: MODE 0 = bits [shift, shift + 8) (LSD radix pass); MODE 1 = row * 256 / num_rows through a multiply-shift
// (the argument carries the multiplier), the missing-id sentinel (== num_rows) lands in the last partition
template <int MODE>
__device__ __forceinline__ uint32_t digit_of(uint64_t k, uint32_t shift_or_mul) {
    if (MODE == 0) return (uint32_t)(k >> shift_or_mul) & (RADIX - 1);
    const uint32_t d = (uint32_t)(((k >> 24) * shift_or_mul) >> 32);           // MODE 1: k = row << 24 | slot
    return d < RADIX - 1 ? d : RADIX - 1;
}

// ---- the chip-wide LSD radix sort (geometry beyond the composite key): keys, per-block histograms, per-digit scan, stable scatter ----
// (rounds 3 - 5 also ran these kernels, gated on a device-side switch, behind every SMALL plan: round 6 replaced that use by
// plan_large_kernel; radix_scan_kernel is shared with the SMALL path's partition pass)
// keys of the chip-wide sort (geometry beyond the composite key): row per slot, slot number as the payload
__global__ __launch_bounds__(256) void radix_make_keys_kernel(const int64_t* __restrict__ ids, int32_t n, int32_t F,
                                                              const int64_t* __restrict__ row_base, uint64_t sentinel,
                                                              uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int32_t stride = gridDim.x * blockDim.x;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const int64_t id = ids[p];
        // missing ids sort to the end -- and so does an id beyond the slab (sentinel == num_rows): it would alias a valid row, which
        // K4 would then update
        const uint64_t r0 = id >= 0 ? (uint64_t)(row_base[p % F] + id) : sentinel;
        keys[p] = r0 < sentinel ? r0 : sentinel;
        vals[p] = (uint32_t)p;
    }
}

// SMALL 1a (round 6: ONE launch for what were up to four -- K1, the field-major ids, make_keys, hist).  Block b owns `sub` consecutive
// examples (64 at config 3: 1024 blocks), walks them 64 at a time and leaves, per slot: the bucket id (HASH: Fingerprint64(text(key)) mod
// buckets, exactly dr_hash_bucket_i64; else the ids are an input), the field-major int32 id (TRANS: through a 64 x (F + 1) LDS tile so that
// both sides are coalesced, exactly dr_ids_transpose_i32), the composite key row << 24 | slot, and the block's partition-digit counts.
// Four consecutive blocks are one chunk of plan_scatter_kernel (their counts are adjacent in the scan).
constexpr int FRONT_GROUP = 4;
template <bool HASH, bool TRANS>
__global__ __launch_bounds__(256) void plan_front_kernel(const int64_t* __restrict__ src, int64_t B, int32_t F,
                                                         const uint64_t* __restrict__ col_buckets, int64_t* __restrict__ ids_out,
                                                         int32_t* __restrict__ ids_t, const int64_t* __restrict__ row_base,
                                                         uint64_t sentinel, uint32_t mul, int32_t sub, uint64_t* __restrict__ keys,
                                                         uint32_t* __restrict__ ghist) {
    extern __shared__ uint64_t fsm[];                      // [F] row_base, (HASH) [F] buckets + [F] floor((2^64 - 1) / buckets), (TRANS) tile
    __shared__ uint32_t h[RADIX];
    int64_t* rb = reinterpret_cast<int64_t*>(fsm);
    uint64_t* cb = fsm + F;
    int32_t* tile = reinterpret_cast<int32_t*>(fsm + (HASH ? 3 : 1) * F);
    const int nb = gridDim.x, t = threadIdx.x;
    for (int c = t; c < F; c += 256) {
        rb[c] = row_base[c];
        if (HASH) {
            const uint64_t nbk = col_buckets[c];
            cb[c] = nbk;
            cb[F + c] = nbk != 0 ? ~0ull / nbk : 0;
        }
    }
    h[t] = 0;
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * sub;
    const int64_t e1 = e0 + sub < B ? e0 + sub : B;
    for (int64_t g0 = e0; g0 < e1; g0 += 64) {             // block-uniform trip count
        const int ge = (int)(e1 - g0 < 64 ? e1 - g0 : 64);
        const int ns = ge * F;
        const int64_t p0 = g0 * F;
        for (int q = t; q < ns; q += 256) {
            const int r = q / F, f = q - r * F;
            const int64_t p = p0 + q;
            int64_t id = src[p];
            if (HASH) {
                id = drhash::bucket_of_key(id, cb[f], cb[F + f]);
                ids_out[p] = id;
            }
            if (TRANS) tile[r * (F + 1) + f] = (int32_t)id;
            const uint64_t r0 = id >= 0 ? (uint64_t)(rb[f] + id) : sentinel;           // (see radix_make_keys_kernel)
            const uint64_t row = r0 < sentinel ? r0 : sentinel;
            const uint64_t ck = (row << 24) | (uint32_t)p;
            keys[p] = ck;
            atomicAdd(&h[digit_of<1>(ck, mul)], 1u);
        }
        if (TRANS) {
            __syncthreads();
            for (int q = t; q < F * 64; q += 256) {
                const int f = q >> 6, r = q & 63;
                if (r < ge) ids_t[(int64_t)f * B + g0 + r] = tile[r * (F + 1) + f];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    ghist[t * nb + blockIdx.x] = h[t];
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* __restrict__ keys, int32_t n, uint32_t shift,
                                                         uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[RADIX];
    const int nb = gridDim.x, t = threadIdx.x;
    const int32_t chunk = (n + nb - 1) / nb;
    const int32_t beg = (int32_t)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    h[t] = 0;
    __syncthreads();
    for (int32_t i = beg + t; i < end; i += 256) atomicAdd(&h[digit_of<0>(keys[i], shift)], 1u);
    __syncthreads();
    ghist[t * nb + blockIdx.x] = h[t];
}

// One block per digit d: ghist[d][0 .. nb) -> exclusive scan over the blocks (position inside the digit's range); the digit's
// total goes to tot[d] -- the scatter blocks turn the 256 totals into range starts themselves (an LDS scan), so no block here
// needs another digit's column and nothing is combined with atomics.
// ZERO (the SMALL path's partition pass): block 0 also zeroes the plan's device-side state, which nothing before the claim kernel
// touches (round 6: two hipMemsetAsync launches less on the side stream).
template <bool ZERO>
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ ghist, int32_t nb, uint32_t* __restrict__ tot,
                                                         Ctrl* __restrict__ ctrl, int32_t* __restrict__ dup_count) {
    if (ZERO && blockIdx.x == 0) {
        int32_t* c = reinterpret_cast<int32_t*>(ctrl);
        for (int i = threadIdx.x; i < (int)(sizeof(Ctrl) / 4); i += 256) c[i] = 0;
        if (threadIdx.x < 2) dup_count[threadIdx.x] = 0;
    }
    __shared__ uint32_t part[256];
    const int t = threadIdx.x, d = blockIdx.x;
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
    uint32_t run = part[t] - sum;
    for (int32_t i = beg; i < end; ++i) {
        const uint32_t c = col[i];
        col[i] = run;
        run += c;
    }
    if (t == 255) tot[d] = part[255];
}

// stable scatter: block b re-reads its chunk in order, 256 keys at a time; a key's destination = start of its digit's range
// + the scanned histogram entry of its digit for this block, advanced by the tiles already written + the number of EARLIER keys
// of the tile with the same digit (wave-level match by eight ballots, earlier waves' counts through LDS).
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                            uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, int32_t n,
                                                            uint32_t shift, const uint32_t* __restrict__ ghist,
                                                            const uint32_t* __restrict__ tot) {
    __shared__ uint32_t running[RADIX];
    __shared__ uint32_t wcnt[4][RADIX];
    const int nb = gridDim.x, t = threadIdx.x, w = t >> 6, lane = t & 63;
    const int32_t chunk = (n + nb - 1) / nb;
    const int32_t beg = (int32_t)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    {   // range starts = exclusive scan of the 256 digit totals
        const uint32_t mine = tot[t];
        running[t] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const uint32_t x = t >= o ? running[t - o] : 0u;
            __syncthreads();
            running[t] += x;
            __syncthreads();
        }
        const uint32_t start = running[t] - mine;
        __syncthreads();
        running[t] = start + ghist[t * nb + blockIdx.x];
    }
    __syncthreads();
    for (int32_t base = beg; base < end; base += 256) {
        const int32_t i = base + t;
        const bool live = i < end;
        const uint64_t k = live ? kin[i] : 0ull;
        const uint32_t v = live ? vin[i] : 0u;
        const uint32_t d = digit_of<0>(k, shift);
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

// ---- the stable scatter step of a 1024-thread block: key k with digit d goes to running[d] + (earlier keys of this step with the same
// digit): wave-level match by eight ballots, earlier waves' counts through LDS; running[] advances by the step's counts.  Every thread of
// the block calls it (barriers inside).
constexpr int LG_T = 1024, LG_W = LG_T / 64;
__device__ __forceinline__ void scatter_step_1024(bool live, uint64_t k, uint32_t d, uint32_t* __restrict__ running,
                                                  uint32_t (*__restrict__ wcnt)[RADIX], uint64_t* __restrict__ dst) {
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
#pragma unroll
    for (int q = 0; q < LG_W * RADIX / LG_T; ++q) (&wcnt[0][0])[q * LG_T + t] = 0;
    __syncthreads();
    uint64_t mask = __ballot(live);                                   // lanes with my digit (dead lanes never match)
#pragma unroll
    for (int bb = 0; bb < RADIX_BITS; ++bb) {
        const bool bit = (d >> bb) & 1u;
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
    __syncthreads();                                                  // every read of running[] precedes its update
    if (t < RADIX) {
        uint32_t add = 0;
#pragma unroll
        for (int ww = 0; ww < LG_W; ++ww) add += wcnt[ww][t];
        running[t] += add;
    }
    if (live) dst[pos] = k;
    __syncthreads();
}

// SMALL 1b: the partition pass's stable scatter (round 6: 1024 threads per block -- 7 steps per chunk at config 3 instead of 26; one
// block per CU keeps the 256 x 256 write frontiers of the partitions resident in the L2s).  Block b re-reads chunk b = FRONT_GROUP front
// blocks' slots in order; its range starts = scanned counts of its first front block.
__global__ __launch_bounds__(LG_T) void plan_scatter_kernel(const uint64_t* __restrict__ kin, uint64_t* __restrict__ kout, int32_t n,
                                                            uint32_t mul, int32_t chunk, const uint32_t* __restrict__ ghist,
                                                            int32_t gstride, const uint32_t* __restrict__ tot) {
    __shared__ uint32_t running[RADIX];
    __shared__ uint32_t wcnt[LG_W][RADIX];
    const int t = threadIdx.x;
    const int64_t beg64 = (int64_t)blockIdx.x * chunk;
    const int32_t beg = (int32_t)(beg64 < n ? beg64 : n), end = (int32_t)(beg64 + chunk < n ? beg64 + chunk : n);
    {   // range starts = exclusive scan of the 256 digit totals
        const uint32_t mine = t < RADIX ? tot[t] : 0u;
        if (t < RADIX) running[t] = mine;
        __syncthreads();
        for (int o = 1; o < RADIX; o <<= 1) {
            const uint32_t x = (t < RADIX && t >= o) ? running[t - o] : 0u;
            __syncthreads();
            if (t < RADIX) running[t] += x;
            __syncthreads();
        }
        if (t < RADIX) running[t] = running[t] - mine + ghist[t * gstride + blockIdx.x * FRONT_GROUP];
    }
    __syncthreads();
    for (int32_t base = beg; base < end; base += LG_T) {              // block-uniform trip count
        const int32_t i = base + t;
        const bool live = i < end;
        const uint64_t k = live ? kin[i] : 0ull;
        scatter_step_1024(live, k, digit_of<1>(k, mul), running, wcnt, kout);
    }
}

// ---- SMALL 2: one block per partition claims its rows in an LDS table, flags the slots, collects the shared-row slots -------------
constexpr int CLAIM_T = 1024;          // (round 6: 512 -> 1024 threads, 7 keys per thread and sweep at config 3)
__global__ __launch_bounds__(CLAIM_T) void plan_lds_claim_kernel(const uint64_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ tot, uint64_t num_rows, int32_t log_tab,
                                                             uint8_t* __restrict__ flags, uint64_t* __restrict__ blist,
                                                             Ctrl* __restrict__ ctrl) {
    extern __shared__ uint32_t tab[];                                 // 2^log_tab entries (dynamic: sized for the batch)
    __shared__ uint32_t starts[RADIX];
    __shared__ int32_t ndup;
    const int t = threadIdx.x, b = blockIdx.x;
    const int LDS_TAB = 1 << log_tab;
    const int hsh = 32 - log_tab;
    // this partition's range = [sum of the totals below b, + tot[b])
    if (t < RADIX) starts[t] = tot[t];
    if (t == 0) ndup = 0;
    for (int i = t; i < LDS_TAB; i += CLAIM_T) tab[i] = EMPTY;
    __syncthreads();
    uint32_t beg = 0;
    for (int i = 0; i < b; ++i) beg += starts[i];                     // (LDS broadcasts; b <= 255)
    const uint32_t cnt = starts[b];
    if (cnt > (uint32_t)(LDS_TAB / 10 * 7)) {                         // a partition the table cannot take (skewed keys): LARGE
        if (t == 0) ctrl->large = 1;
        return;
    }
    // sweep 1: insert
    for (uint32_t i = t; i < cnt; i += CLAIM_T) {
        const uint64_t k = keys[beg + i] >> 24;
        if (k >= num_rows) continue;                                  // missing id
        const uint32_t r = (uint32_t)k;
        uint32_t h = (r * 2654435761u) >> hsh;
        for (;;) {
            const uint32_t old = atomicCAS(&tab[h], EMPTY, r);
            if (old == EMPTY) break;                                  // claimed: first slot of this row (so far the only one)
            if ((old & ~DUPBIT) == r) {                               // the row is already there: shared
                if (!(old & DUPBIT)) atomicOr(&tab[h], DUPBIT);
                break;
            }
            h = (h + 1) & (LDS_TAB - 1);
        }
    }
    __syncthreads();
    // sweep 2: flags + the shared-row slots (composite key row << 24 | slot) into this partition's bucket
    for (uint32_t i = t; i < cnt; i += CLAIM_T) {
        const uint64_t ck = keys[beg + i];
        const uint64_t k = ck >> 24;
        const uint32_t slot = (uint32_t)(ck & 0xFFFFFFu);
        bool dup = false;
        if (k < num_rows) {
            const uint32_t r = (uint32_t)k;
            uint32_t h = (r * 2654435761u) >> hsh;
            uint32_t v = tab[h];
            while ((v & ~DUPBIT) != r) {
                h = (h + 1) & (LDS_TAB - 1);
                v = tab[h];
            }
            dup = (v & DUPBIT) != 0;
            if (dup) {
                const int pos = atomicAdd(&ndup, 1);
                if (pos < CAPB) blist[b * CAPB + pos] = ((uint64_t)r << 24) | slot;
            }
        }
        flags[slot] = (k < num_rows && !dup) ? 1 : 0;
    }
    __syncthreads();
    if (t == 0) ctrl->bcnt[b] = ndup;                                 // (> CAPB: the sort kernel sends the batch to LARGE)
}

// ---- SMALL 3: one block per bucket sorts it in LDS, writes it at its offset, emits the duplicate pass's work list ------------------
__global__ __launch_bounds__(256) void plan_bucket_sort_kernel(const uint64_t* __restrict__ blist, Ctrl* __restrict__ ctrl,
                                                               int32_t small_limit, uint64_t* __restrict__ rows,
                                                               int32_t* __restrict__ slots, int32_t* __restrict__ dup_heads,
                                                               int32_t* __restrict__ dup_count) {
    __shared__ uint64_t s[CAPB];
    __shared__ int32_t cnts[NBK];
    __shared__ int32_t nheads, hbase;
    const int t = threadIdx.x, b = blockIdx.x;
    if (ctrl->large != 0) return;                                     // geometry or a partition forced the LARGE path
    if (t < NBK) cnts[t] = ctrl->bcnt[t];
    if (t == 0) nheads = 0;
    __syncthreads();
    int32_t m = 0, off = 0;
    bool over = false;
    for (int i = 0; i < NBK; ++i) {                                   // LDS broadcasts: every thread gets the same totals
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

// ---- LARGE: marking over the fully sorted list ---------------------------------------------------------------------------------
// flag[p] = 1 iff slot p is the only slot of the batch that touches its row; every sorted position that heads a piece of a row
// touched by >= 2 slots (the segment start, plus each CH-aligned position >= CH past it) goes to dup_heads.  A block owns a
// contiguous chunk, collects its heads in LDS and reserves their place with ONE global atomic (a Zipf batch has ~1e5 heads; one
// same-address atomic per wave cost 300 us, they retire at ~88 per microsecond).
__global__ __launch_bounds__(256) void radix_mark_kernel(const uint64_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                                         int32_t n, uint64_t num_rows, uint8_t* __restrict__ flags,
                                                         int32_t* __restrict__ dup_heads, int32_t* __restrict__ dup_count) {
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

// ---- LARGE in one launch (claimable geometry): block b sorts partition b by (row, slot), then marks it -------------------------
// The partition pass left partition b -- the slots whose row falls into the b-th of 256 row ranges -- as composite keys
// (row << 24 | slot) at [beg, beg + cnt) of `ka`, in slot order.  A block-local LSD radix sort (8-bit digits of row - min row, as
// many passes as that difference has digits; stable, so equal rows stay in slot order) ping-pongs between `ka` and the same range
// of `rows` (both n x 8 bytes) and ends in `ka`; then rows / slots / flags / heads are written exactly as radix_mark_kernel
// writes them over the chip-wide sort's output: the list K4's duplicate pass walks is the same list (the order of dup_heads is
// arrival order in both).  Any partition size is taken (one block walks it 1024 keys at a time: a field whose every example
// carries the same id is 65 536 keys = 64 steps per pass); Zipf batches at config 3 have 4 - 15 K keys per partition.
__global__ __launch_bounds__(LG_T) void plan_large_kernel(uint64_t* __restrict__ ka, const uint32_t* __restrict__ tot,
                                                          uint64_t num_rows, int32_t n, uint64_t* __restrict__ rows,
                                                          int32_t* __restrict__ slots, uint8_t* __restrict__ flags,
                                                          int32_t* __restrict__ dup_heads, int32_t* __restrict__ dup_count,
                                                          const Ctrl* __restrict__ ctrl) {
    if (ctrl->large == 0) return;
    __shared__ uint32_t running[RADIX];
    __shared__ uint32_t wcnt[LG_W][RADIX];
    __shared__ uint32_t starts[RADIX];
    __shared__ uint32_t rmin, rmax;
    __shared__ int32_t nh, gbase;
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63;
    if (t < RADIX) starts[t] = tot[t];
    if (t == 0) { rmin = 0xFFFFFFFFu; rmax = 0u; }
    if (b == 0 && t == 0) dup_count[1] = n;
    __syncthreads();
    uint32_t beg = 0;
    for (int i = 0; i < b; ++i) beg += starts[i];
    const int32_t cnt = (int32_t)starts[b];
    uint64_t* src = ka + beg;
    uint64_t* dst = rows + beg;
    // the bits in which this partition's rows differ (rows < 2^31 here: 32-bit LDS atomics)
    {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int32_t i = t; i < cnt; i += LG_T) {
            const uint32_t r = (uint32_t)(src[i] >> 24);
            lo = r < lo ? r : lo;
            hi = r > hi ? r : hi;
        }
        if (lo <= hi) { atomicMin(&rmin, lo); atomicMax(&rmax, hi); }
    }
    __syncthreads();
    const uint32_t base_row = rmin;
    int passes = 0;
    if (cnt > 1 && rmax > rmin) passes = (int)((bits_for_dev(rmax - rmin) + RADIX_BITS - 1) / RADIX_BITS);
    if (passes & 1) {                                                 // an odd number of passes must START in `rows` to end in `ka`
        for (int32_t i = t; i < cnt; i += LG_T) dst[i] = src[i];
        uint64_t* x = src; src = dst; dst = x;
        __syncthreads();
    }
    for (int p = 0; p < passes; ++p) {
        const uint32_t shift = (uint32_t)(p * RADIX_BITS);
        if (t < RADIX) running[t] = 0;
        __syncthreads();
        for (int32_t i = t; i < cnt; i += LG_T)
            atomicAdd(&running[(((uint32_t)(src[i] >> 24) - base_row) >> shift) & (RADIX - 1)], 1u);
        __syncthreads();
        {   // exclusive scan of the 256 digit counts (every thread passes every barrier)
            const uint32_t mine = t < RADIX ? running[t] : 0u;
            for (int o = 1; o < RADIX; o <<= 1) {
                const uint32_t x = (t < RADIX && t >= o) ? running[t - o] : 0u;
                __syncthreads();
                if (t < RADIX) running[t] += x;
                __syncthreads();
            }
            if (t < RADIX) running[t] -= mine;
        }
        __syncthreads();
        for (int32_t base = 0; base < cnt; base += LG_T) {            // block-uniform trip count
            const int32_t i = base + t;
            const bool live = i < cnt;
            const uint64_t k = live ? src[i] : 0ull;
            scatter_step_1024(live, k, (((uint32_t)(k >> 24) - base_row) >> shift) & (RADIX - 1), running, wcnt, dst);
        }
        uint64_t* x = src; src = dst; dst = x;
    }
    // `src` == ka + beg holds the partition sorted by (row, slot): decode + mark (cf. radix_mark_kernel; positions are global)
    for (int32_t base = 0; base < cnt; base += LG_T) {
        const int32_t i = base + t;
        const bool in = i < cnt;
        const uint64_t ck = in ? src[i] : ~0ull;
        const uint64_t k = ck >> 24;
        const int32_t gi = (int32_t)beg + i;
        const bool valid = in && k < num_rows;
        const bool seg_start = in && (i == 0 || (src[i - 1] >> 24) != k);          // a row never straddles partitions
        const bool has_next = in && (i + 1 < cnt) && (src[i + 1] >> 24) == k;
        if (in) {
            const int32_t slot = (int32_t)(ck & 0xFFFFFFu);
            rows[gi] = k;
            slots[gi] = slot;
            flags[slot] = (valid && seg_start && !has_next) ? 1 : 0;
        }
        const bool head = valid && ((seg_start && has_next) || (!seg_start && (gi % CH) == 0 && i >= CH && (src[i - CH] >> 24) == k));
        if (t == 0) nh = 0;
        __syncthreads();
        const uint64_t bal = __ballot(head);
        int wbase = 0;
        const int leader = bal != 0 ? __ffsll((unsigned long long)bal) - 1 : 0;
        if (bal != 0 && lane == leader) wbase = atomicAdd(&nh, __popcll(bal));
        __syncthreads();
        if (t == 0) gbase = nh > 0 ? atomicAdd(dup_count, nh) : 0;                 // one global atomic per 1024 keys
        __syncthreads();
        if (bal != 0) {
            wbase = __shfl(wbase, leader, 64);
            if (head) dup_heads[gbase + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = gi;
        }
    }
}

}  // namespace

// Test / tuning hook: the largest shared-row list the SMALL path takes (default and maximum 16384); 0 sends every batch that has
// any shared row down the LARGE (radix sort) path.  Process-wide; returns the previous value.
extern "C" int32_t dr_emb_plan_set_small_limit(int32_t limit) {
    if (limit < 0) limit = 0;
    if (limit > SMALL_CAP) limit = SMALL_CAP;
    return g_small_limit.exchange(limit);
}

extern "C" int64_t dr_emb_sort_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return (int64_t)layout_for(n).total;
}

namespace {
// ids != nullptr: the plan of these bucket ids.  raw != nullptr: K1 in front -- ids_out (and ids_t, may be null) are OUTPUTS of the front
// kernel.  Returns DR_OK + *done = false when the geometry is not the fused front's (the caller then runs the plain entry points).
int plan_small_front(const int64_t* raw, const uint64_t* col_buckets, int64_t* ids_out, int32_t* ids_t, const int64_t* ids, int64_t B,
                     int32_t F, const int64_t* row_base, int64_t num_rows, uint64_t* rows_y, uint64_t* keys_x, uint32_t* ghist,
                     uint32_t* tot, Ctrl* ctrl, int32_t* dup_count, hipStream_t s) {
    const int64_t n64 = B * F;
    const int32_t n = (int32_t)n64;
    // row * 256 / num_rows as a multiply-shift; floor, so that row * mul >> 32 <= 255 for every row < num_rows
    const uint32_t mul = (uint32_t)((((uint64_t)RADIX) << 32) / (uint64_t)num_rows);
    // front blocks of `sub` examples (a multiple of 64), at most RADIX_MAX_BLOCKS of them; FRONT_GROUP of them = one scatter chunk
    int64_t sub = 64;
    while ((B + sub - 1) / sub > RADIX_MAX_BLOCKS) sub += 64;
    const int nbf = (int)((B + sub - 1) / sub);
    const int nbs = (nbf + FRONT_GROUP - 1) / FRONT_GROUP;
    const int64_t chunk = sub * FRONT_GROUP * F;
    if (chunk > 0x7fffffff) return DR_EINVAL;
    const bool trans = ids_t != nullptr;
    const size_t lds = (size_t)8 * (raw ? 3 : 1) * F + (trans ? (size_t)4 * 64 * (F + 1) : 0);
    if (raw) {
        if (trans)
            hipLaunchKernelGGL((plan_front_kernel<true, true>), dim3(nbf), dim3(256), lds, s, raw, B, F, col_buckets, ids_out, ids_t, row_base,
                               (uint64_t)num_rows, mul, (int32_t)sub, rows_y, ghist);
        else
            hipLaunchKernelGGL((plan_front_kernel<true, false>), dim3(nbf), dim3(256), lds, s, raw, B, F, col_buckets, ids_out, ids_t,
                               row_base, (uint64_t)num_rows, mul, (int32_t)sub, rows_y, ghist);
    } else {
        hipLaunchKernelGGL((plan_front_kernel<false, false>), dim3(nbf), dim3(256), lds, s, ids, B, F, col_buckets, ids_out, ids_t, row_base,
                           (uint64_t)num_rows, mul, (int32_t)sub, rows_y, ghist);
    }
    hipLaunchKernelGGL((radix_scan_kernel<true>), dim3(RADIX), dim3(256), 0, s, ghist, (int32_t)nbf, tot, ctrl, dup_count);
    hipLaunchKernelGGL(plan_scatter_kernel, dim3(nbs), dim3(LG_T), 0, s, rows_y, keys_x, n, mul, (int32_t)chunk, ghist, (int32_t)nbf, tot);
    return DR_OK;
}

int sort_slots_impl(const int64_t* raw, const uint64_t* col_buckets, int32_t* ids_t, const int64_t* ids, int64_t B, int32_t F,
                    const int64_t* row_base, int64_t num_rows, int64_t* sorted_rows, int32_t* sorted_slots, uint8_t* unique_flags,
                    int32_t* dup_heads, int32_t* dup_count, void* workspace, int64_t workspace_bytes, dr_stream_t stream) {
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
    uint64_t* keys_x = reinterpret_cast<uint64_t*>(w + L.keys_x);
    uint32_t* vals_x = reinterpret_cast<uint32_t*>(w + L.vals_x);
    uint64_t* blist = reinterpret_cast<uint64_t*>(w + L.blist);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(w + L.ghist);
    uint32_t* tot = reinterpret_cast<uint32_t*>(w + L.tot);
    Ctrl* ctrl = reinterpret_cast<Ctrl*>(w + L.ctrl);
    uint64_t* rows_y = reinterpret_cast<uint64_t*>(sorted_rows);
    uint32_t* slots_y = reinterpret_cast<uint32_t*>(sorted_slots);
    hipStream_t s = dr_s(stream);
    const int grid = dr_grid_for(n, 256);
    const int nb = radix_blocks(n);

    // the composite key holds 24 bits of slot and the LDS table 31 bits of row; anything larger sorts all slots
    const bool claimable = n64 <= (1 << 24) && num_rows < 0x7fffffffLL;
    // K1 inside the front kernel: its LDS holds the per-field constants and (for the field-major ids) a 64 x (F + 1) tile
    const bool fused_front = raw != nullptr && claimable && F <= 64;
    if (raw && !fused_front) {
        int rc = dr_hash_bucket_i64(raw, B, F, col_buckets, const_cast<int64_t*>(ids), stream);
        if (rc == DR_OK && ids_t) rc = dr_ids_transpose_i32(ids, B, F, ids_t, stream);
        if (rc != DR_OK) return rc;
    }
    if (claimable) {
        int rc = plan_small_front(fused_front ? raw : nullptr, col_buckets, const_cast<int64_t*>(ids), fused_front ? ids_t : nullptr, ids, B,
                                  F, row_base, num_rows, rows_y, keys_x, ghist, tot, ctrl, dup_count, s);
        if (rc != DR_OK) return rc;
        // (dynamic LDS beyond 64 KB needs the opt-in)
        static const hipError_t lds_optin = hipFuncSetAttribute(reinterpret_cast<const void*>(plan_lds_claim_kernel),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 << LDS_TAB_MAX_LOG);
        if (lds_optin != hipSuccess) return DR_ELAUNCH;
        int log_tab = 12;
        while (log_tab < LDS_TAB_MAX_LOG && (1 << log_tab) < 2 * ((n + NBK - 1) / NBK)) ++log_tab;
        hipLaunchKernelGGL(plan_lds_claim_kernel, dim3(NBK), dim3(CLAIM_T), (size_t)4 << log_tab, s, keys_x, tot, (uint64_t)num_rows,
                           (int32_t)log_tab, unique_flags, blist, ctrl);
        hipLaunchKernelGGL(plan_bucket_sort_kernel, dim3(NBK), dim3(256), 0, s, blist, ctrl, (int32_t)g_small_limit.load(), rows_y,
                           sorted_slots, dup_heads, dup_count);
        // LARGE, gated on the device: the partitions (still in keys_x) sorted one per block
        hipLaunchKernelGGL(plan_large_kernel, dim3(NBK), dim3(LG_T), 0, s, keys_x, tot, (uint64_t)num_rows, n, rows_y, sorted_slots,
                           unique_flags, dup_heads, dup_count, ctrl);
        DR_CHECK_LAUNCH();
        return DR_OK;
    }
    // geometry beyond the composite key: the chip-wide LSD radix sort of all n slots; the final pass must land in the output arrays
    if (hipMemsetAsync(dup_count, 0, 2 * sizeof(int32_t), s) != hipSuccess) return DR_ELAUNCH;
    const unsigned bits = bits_for((uint64_t)num_rows);            // the sentinel == num_rows needs these bits too
    const int passes = (int)((bits + RADIX_BITS - 1) / RADIX_BITS);
    uint64_t* kbuf[2] = {keys_x, rows_y};
    uint32_t* vbuf[2] = {vals_x, slots_y};
    int cur = (passes & 1) ? 0 : 1;                                // passes odd: start in the workspace, end in the outputs
    hipLaunchKernelGGL(radix_make_keys_kernel, dim3(grid), dim3(256), 0, s, ids, n, F, row_base, (uint64_t)num_rows, kbuf[cur], vbuf[cur]);
    for (int p = 0; p < passes; ++p) {
        const uint32_t shift = (uint32_t)(p * RADIX_BITS);
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(256), 0, s, kbuf[cur], n, shift, ghist);
        hipLaunchKernelGGL((radix_scan_kernel<false>), dim3(RADIX), dim3(256), 0, s, ghist, (int32_t)nb, tot, ctrl, dup_count);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(256), 0, s, kbuf[cur], vbuf[cur], kbuf[cur ^ 1], vbuf[cur ^ 1], n, shift, ghist,
                           tot);
        cur ^= 1;
    }
    int mark_blocks = MARK_BLOCKS;
    while ((n + mark_blocks - 1) / mark_blocks > MARK_MAX_CHUNK) mark_blocks *= 2;
    hipLaunchKernelGGL(radix_mark_kernel, dim3(mark_blocks), dim3(256), 0, s, rows_y, sorted_slots, n, (uint64_t)num_rows, unique_flags, dup_heads,
                       dup_count);
    DR_CHECK_LAUNCH();
    return DR_OK;
}
}  // namespace

extern "C" int dr_emb_sort_slots(const int64_t* ids, int64_t B, int32_t F, const int64_t* row_base, int64_t num_rows,
                                 int64_t* sorted_rows, int32_t* sorted_slots, uint8_t* unique_flags, int32_t* dup_heads,
                                 int32_t* dup_count, void* workspace, int64_t workspace_bytes, dr_stream_t stream) {
    return sort_slots_impl(nullptr, nullptr, nullptr, ids, B, F, row_base, num_rows, sorted_rows, sorted_slots, unique_flags, dup_heads,
                           dup_count, workspace, workspace_bytes, stream);
}

// K1 + the field-major ids + the slot plan of the same batch: dr_hash_bucket_i64(keys -> ids_out), dr_ids_transpose_i32(ids_out -> ids_t_out;
// ids_t_out may be NULL) and dr_emb_sort_slots(ids_out) with the first three kernels of the chain as ONE (the engines' next-batch prefetch:
// every launch on the side stream costs the training stream 3 - 5 us beside the GEMMs).  Same outputs, bit for bit.
extern "C" int dr_hash_sort_slots(const int64_t* keys, int64_t B, int32_t F, const uint64_t* col_buckets, int64_t* ids_out,
                                  int32_t* ids_t_out, const int64_t* row_base, int64_t num_rows, int64_t* sorted_rows,
                                  int32_t* sorted_slots, uint8_t* unique_flags, int32_t* dup_heads, int32_t* dup_count, void* workspace,
                                  int64_t workspace_bytes, dr_stream_t stream) {
    if (!keys || !col_buckets || !ids_out) return DR_EINVAL;
    return sort_slots_impl(keys, col_buckets, ids_t_out, ids_out, B, F, row_base, num_rows, sorted_rows, sorted_slots, unique_flags,
                           dup_heads, dup_count, workspace, workspace_bytes, stream);
}
