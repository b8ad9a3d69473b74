// sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs and an exclusive u32 scan,
// written for wave64 (gfx950).  Replaces the CUB DeviceRadixSort::SortPairs / DeviceScan::
// InclusiveSum calls of the reference's un-vendored CUDA module.
//
// The rasterizer uses a TWO-LEVEL sort instead of the reference's single 64-bit (tile<<32|depth)
// sort: (1) the P Gaussians are sorted once by their 32-bit depth key (4 passes over P pairs),
// (2) instances are emitted in that order and stably sorted by tile id only (2 passes over D pairs
// for up to 65 536 tiles).  A stable sort by tile of a depth-ordered sequence yields exactly the
// reference order (tile, depth bits, Gaussian index) at ~20 B x 2 passes per instance instead of
// ~24 B x 6 passes.
//
// One pass = two kernels (three from 16.8 M pairs up): per-block digit histograms + per-group digit sums (integer atomics),
// [a one-block scan of the group sums,] and a stable scatter whose blocks derive their own output bases from the two tables and
// reorder their chunk by digit in LDS before writing it out.  Stability inside a block comes from wave-level match-any
// (8 ballots) ranking with each wave owning a contiguous key segment, waves ordered by id.
// Every kernel is HBM-streaming: 4 B (hist) + 8 B read + 8 B write per pair per pass (6 + 6 with 16-bit tile keys).
// No inter-workgroup waiting anywhere.  (The look-back single-pass variants, the scanned-table and unstaged scatters of rounds
// 1-5 measured the same or slower and are gone from the product: profiles/r1_notes.md, git history.)
#include "msgs_internal.h"

#include <algorithm>

namespace msgs {

namespace {

// ---------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = (uint32_t)__shfl_up((int)v, off);
        if (lane >= off) v += n;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix and the
// block total through *total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = s_wave[k];
        if (k < w) base += t;
    }
    *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + inc - v;
}

// SCAN_ITEMS consecutive values of one thread: 16-byte accesses when the run is complete and aligned
__device__ __forceinline__ void scan_load_items(const uint32_t* in, int64_t base, int64_t n, uint32_t v[SCAN_ITEMS]) {
    static_assert(SCAN_ITEMS % 4 == 0, "uint4 runs");
    if (base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(in + base) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4*>(in + base + 4 * q);
            v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = base + k < n ? in[base + k] : 0u;
    }
}
__device__ __forceinline__ void scan_store_items(uint32_t* out, int64_t base, int64_t n, const uint32_t v[SCAN_ITEMS]) {
    if (base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(out + base) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q)
            *reinterpret_cast<uint4*>(out + base + 4 * q) = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) out[base + k] = v[k];
    }
}

// with `gather`, the gathered values are also STORED to staged[] (the caller passes the output array): the second pass
// then scans that array in place and the random gather happens once instead of twice
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                                   const uint32_t* __restrict__ gather,
                                                                   int64_t n, uint64_t* __restrict__ partials,
                                                                   uint32_t* __restrict__ staged,
                                                                   const uint32_t* __restrict__ n_ptr, uint32_t in_mask,
                                                                   uint32_t* __restrict__ side_out, uint32_t* __restrict__ side_flag) {
    __shared__ uint32_t s_wave[4];
    if (n_ptr) n = (int64_t)*n_ptr;
    if (side_flag && blockIdx.x == 0 && threadIdx.x == 0) *side_flag = 1u;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    if (gather) {
        uint32_t g[SCAN_ITEMS];
        scan_load_items(gather, base, n, g);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = base + k < n ? in[g[k]] : 0u;
        if (side_out) {                     // what travels above the scanned bits (GeomLayout::tiles: the cell ranges)
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; ++k) g[k] = v[k] >> TILE_COUNT_BITS;
            scan_store_items(side_out, base, n, g);
        }
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] &= in_mask;
        if (staged) scan_store_items(staged, base, n, v);
    } else {
        scan_load_items(in, base, n, v);
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) sum += v[k];
    uint32_t total;
    block_exclusive_scan(sum, s_wave, &total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// Two-kernel scan: scan_reduce_kernel leaves one total per block; every block of THIS kernel sums the
// totals in front of it itself (nb <= a few thousand values, L2-resident: 1 - 5 loads per thread) instead of waiting for a
// one-block kernel in between, and block 0 — which sums ALL of them — publishes the grand total first thing: device word,
// clamped copy for a speculative stage 2, status words and the polled host words.  One launch less per forward (~5 us).
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_fused_kernel(const uint32_t* in, uint32_t* out, int64_t n,
                                                                        const uint64_t* __restrict__ partials, int64_t nb,
                                                                        const uint32_t* __restrict__ n_ptr,
                                                                        uint64_t* __restrict__ total_out,
                                                                        uint64_t* __restrict__ status,
                                                                        volatile uint64_t* host, uint64_t ticket,
                                                                        uint32_t* __restrict__ clamped_total, uint64_t clamp,
                                                                        const uint32_t* __restrict__ extra,
                                                                        uint32_t* __restrict__ zero_word,
                                                                        uint32_t* __restrict__ overflow_flag) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint64_t s_sum[4];
    if (n_ptr) n = (int64_t)*n_ptr;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // blocks behind the data (n_ptr < grid capacity) still need nothing; block 0 always runs (it owns the publication)
    const int64_t chunk0 = (int64_t)blockIdx.x * SCAN_CHUNK;
    if (blockIdx.x != 0 && chunk0 >= n) return;
    const int64_t upto = blockIdx.x == 0 ? nb : (int64_t)blockIdx.x;       // block 0: the grand total
    uint64_t acc = 0;
    for (int64_t q = threadIdx.x; q < upto; q += SCAN_THREADS) acc += partials[q];
    for (int off = 32; off > 0; off >>= 1)
        acc += ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(acc >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)acc, off);
    if (lane == 0) s_sum[w] = acc;
    __syncthreads();
    const uint64_t sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    uint64_t base = sum;
    if (blockIdx.x == 0) {
        base = 0;
        if (threadIdx.x == 0) {
            if (total_out) *total_out = sum;
            if (zero_word) *zero_word = 0u;
            if (clamped_total) *clamped_total = (uint32_t)(sum < clamp ? sum : clamp);
            if (overflow_flag && sum > clamp) *overflow_flag = 1u;
            const uint64_t info = extra ? ((uint64_t)extra[0] | ((uint64_t)extra[1] << 32)) : 0ull;
            if (status) { status[0] = sum; status[1] = 0; status[2] = info; }
            if (host) {
                host[0] = sum;
                host[1] = 0;
                host[3] = info;
                __threadfence_system();
                host[2] = ticket;
            }
        }
        if (chunk0 >= n) return;
    }
    const int64_t tb = chunk0 + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    scan_load_items(in, tb, n, v);
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) tsum += v[k];
    uint32_t total;
    uint32_t run = block_exclusive_scan(tsum, s_wave, &total) + (uint32_t)base;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t x = v[k];
        v[k] = run;
        run += x;
    }
    scan_store_items(out, tb, n, v);
}

// ---------------------------------------------------------------------------------------------
// radix pass
// ---------------------------------------------------------------------------------------------
// Element layout inside a block's chunk: wave w owns [w*64*ITEMS, (w+1)*64*ITEMS); in round r lane l
// handles element w*64*ITEMS + r*64 + l.  Waves, rounds and lanes are therefore all in key order.
template <int ITEMS = SORT_ITEMS>
__device__ __forceinline__ int64_t elem_index(int64_t chunk_base, int w, int r, int lane) {
    return chunk_base + (int64_t)w * (64 * ITEMS) + r * 64 + lane;
}

// With `gsum` (grouped path) the block histograms are stored block-major (hist[block][digit]) and the
// per-(group of `gsize` blocks, digit) sums gsum[group][digit] are accumulated with atomics: with those every
// scatter block derives its own output bases (no scan kernels at all).
// Compaction (depth sort of the rasterizer): with DROP the pass ignores keys equal to 0xFFFFFFFF (Gaussians that are not
// rendered) — they are neither counted here nor written by the scatter, so the pass's output holds only the survivors, in
// stable order — and the scatter publishes their number; the later passes take their element count from that device word
// (n_ptr) and run over the survivors only: their grids are still sized for the upper bound, surplus blocks leave at once.
// KeyT: uint32_t, or uint16_t for the tile sort whenever the tile ids (and the sentinel) fit 16 bits — 6 instead of 8 bytes per
// pair and pass; the keys sit in uint16 arrays in HBM and are widened in registers / LDS.
template <int ITEMS, bool DROP = false, typename KeyT = uint32_t>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const KeyT* __restrict__ keys, int64_t n,
                                                                  int shift, uint32_t mask, int64_t nblocks,
                                                                  uint32_t* __restrict__ hist,
                                                                  uint32_t* __restrict__ gsum, int gsize, int ngroups,
                                                                  const uint32_t* __restrict__ n_ptr = nullptr) {
    __shared__ uint32_t s_hist[256];
    const int64_t chunk_base = (int64_t)blockIdx.x * (SORT_THREADS * ITEMS);
    if (n_ptr) {
        n = (int64_t)*n_ptr;
        if (chunk_base >= n) return;                   // (its histogram row is never read: rows of EARLIER blocks only)
    }
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int KPV = 16 / (int)sizeof(KeyT);            // keys per 16-byte load
    if (ITEMS % KPV == 0 && chunk_base + (int64_t)SORT_THREADS * ITEMS <= n) {
        // a full chunk: the block only needs the chunk's digit counts, whatever thread sees which key — 16-byte loads
        const uint4* kv = reinterpret_cast<const uint4*>(keys + chunk_base);
#pragma unroll
        for (int r = 0; r < ITEMS / KPV; ++r) {
            const uint4 v = kv[r * SORT_THREADS + threadIdx.x];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < KPV; ++j) {
                const uint32_t k = sizeof(KeyT) == 4 ? w4[j] : ((w4[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
                if (!DROP || k != 0xFFFFFFFFu) atomicAdd(&s_hist[(k >> shift) & mask], 1u);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int64_t i = elem_index<ITEMS>(chunk_base, w, r, lane);
            if (i < n) {
                const uint32_t k = (uint32_t)keys[i];
                if (!DROP || k != 0xFFFFFFFFu) atomicAdd(&s_hist[(k >> shift) & mask], 1u);
            }
        }
    }
    __syncthreads();
    const uint32_t c = s_hist[threadIdx.x];
    // both tables are [block or group][digit] so that thread = digit accesses coalesce
    hist[(int64_t)blockIdx.x * 256 + threadIdx.x] = c;
    if (c) atomicAdd(&gsum[(int64_t)(blockIdx.x / gsize) * 256 + threadIdx.x], c);
}

// Large inputs (SCANNED_MIN_BLOCKS): one small block turns the group sums into the scatter's bases in place —
// gsum[g][d] := (keys with a smaller digit) + (digit d in groups before g) — so that a scatter block reads ONE row of this
// table plus the histograms of the earlier blocks of its own group, instead of every group's sums (with 13 400 blocks that
// was 112 + 60 rows = 176 KB per block, more bytes than the keys the block moves).  A kernel of its own: folding it into the
// histogram kernel's last-arriving block needs a device-scope release fence in EVERY block, which on this part writes the
// L2 back each time (measured: the tile sort of 55 M pairs 1.15 -> 3.8 ms).
// Blocks 0 .. ngroups-1 of the same launch turn the block histograms of their group into exclusive in-group prefixes (per digit,
// in place): a scatter block then reads ONE histogram row and ONE group row instead of walking the rows of the earlier blocks of
// its group (on average 15.5 KB of table per 32 KB of keys and values, and a chain of dependent loads).
__global__ __launch_bounds__(256) void group_scan_kernel(uint32_t* __restrict__ gsum, int ngroups,
                                                         uint32_t* __restrict__ hist, int gsize, int64_t nblocks,
                                                         const uint32_t* __restrict__ n_ptr, int chunk) {
    if (n_ptr) {        // device-side element count (capacity-sized launch): only the blocks that hold keys wrote their tables
        const int64_t n = (int64_t)*n_ptr;
        nblocks = (n + chunk - 1) / chunk;
        ngroups = (int)((nblocks + gsize - 1) / gsize);
        if ((int)blockIdx.x > ngroups) return;              // (ngroups == 0: every workgroup leaves)
        if (nblocks == 0) return;
    }
    const int d = threadIdx.x;
    if ((int)blockIdx.x < ngroups) {
        const int64_t first = (int64_t)blockIdx.x * gsize;
        const int rows = (int)min((int64_t)gsize, nblocks - first);
        uint32_t* q = hist + first * 256 + d;
        uint32_t run = 0;
#pragma unroll 8
        for (int k = 0; k < rows; ++k) {
            const uint32_t v = q[(int64_t)k * 256];
            q[(int64_t)k * 256] = run;
            run += v;
        }
        return;
    }
    __shared__ uint32_t s_wv[4];
    uint32_t tot = 0;
#pragma unroll 8
    for (int k = 0; k < ngroups; ++k) tot += gsum[(int64_t)k * 256 + d];
    uint32_t dummy;
    uint32_t run = block_exclusive_scan(tot, s_wv, &dummy);
#pragma unroll 8
    for (int k = 0; k < ngroups; ++k) {
        uint32_t* q = &gsum[(int64_t)k * 256 + d];
        const uint32_t v = *q;
        *q = run;
        run += v;
    }
}

template <int ITEMS, bool DROP = false, typename KeyT = uint32_t>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const KeyT* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     KeyT* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, int64_t n,
                                                                     int shift, uint32_t mask, int64_t nblocks,
                                                                     const uint32_t* __restrict__ hist_scanned,
                                                                     const uint32_t* __restrict__ gsum, int gsize,
                                                                     int ngroups, bool gsum_is_base = false,
                                                                     const uint32_t* __restrict__ n_ptr = nullptr,
                                                                     uint32_t* __restrict__ n_out = nullptr) {
    __shared__ uint32_t s_cnt[4][256];   // per-wave digit counters, later per-wave global bases
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t chunk_base = (int64_t)blockIdx.x * (SORT_THREADS * ITEMS);
    if (n_ptr) {
        n = (int64_t)*n_ptr;
        if (chunk_base >= n) return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_cnt[k][threadIdx.x] = 0;
    __syncthreads();

    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = elem_index<ITEMS>(chunk_base, w, r, lane);
        key[r] = i < n ? (uint32_t)keys_in[i] : 0xFFFFFFFFu;
        const bool valid = i < n && (!DROP || key[r] != 0xFFFFFFFFu);      // DROP: not-rendered Gaussians leave the sort here
        val[r] = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
        const uint32_t d = (key[r] >> shift) & mask;
        // match-any over the 8 digit bits
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t prev = s_cnt[w][d];                        // same address for all peers: broadcast
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        rank[r] = prev + before;
        // LDS operations of one wave are issued in order, so the read above precedes this update
        if (valid && before == 0) s_cnt[w][d] = prev + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    uint32_t gbase;
    {   // digit d = threadIdx.x: global base of this block's run of digit d
        const uint32_t d = threadIdx.x;
        if (gsum_is_base) {
            // group_scan_kernel turned the group sums into bases and the block histograms into in-group prefixes (large inputs)
            const int g = blockIdx.x / gsize;
            gbase = gsum[(int64_t)g * 256 + d] + hist_scanned[(int64_t)blockIdx.x * 256 + d];
        } else {
            // base = (keys with a smaller digit) + (same digit in earlier groups) + (same digit in earlier blocks of
            // this group); hist_scanned holds the RAW block histograms here
            const int g = blockIdx.x / gsize;
            uint32_t tot = 0, before = 0;
#pragma unroll 8
            for (int k = 0; k < ngroups; ++k) {                 // independent coalesced loads, 8 in flight
                const uint32_t v = gsum[(int64_t)k * 256 + d];
                tot += v;
                before += k < g ? v : 0u;
            }
            const int nin = (int)(blockIdx.x - (int64_t)g * gsize);
            const uint32_t* hrow = hist_scanned + ((int64_t)g * gsize) * 256 + d;
#pragma unroll 8
            for (int k = 0; k < nin; ++k) before += hrow[(int64_t)k * 256];
            uint32_t total;
            gbase = block_exclusive_scan(tot, s_wave, &total) + before;
            if (n_out && blockIdx.x == 0 && threadIdx.x == 0) *n_out = total;     // survivors of a DROP pass
        }
    }
    // reorder the chunk by digit in LDS first, then write it out in local order — consecutive lanes write
    // consecutive addresses inside each digit run (a wave store touches a few runs instead of 64 scattered words)
    __shared__ uint32_t s_key[SORT_THREADS * ITEMS], s_val[SORT_THREADS * ITEMS], s_delta[256];
    uint32_t nkept;                                                        // keys of this chunk that stay in the sort
    {
        const uint32_t d = threadIdx.x;
        uint32_t c[4], ltot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = s_cnt[k][d]; ltot += c[k]; }
        const uint32_t lbase = block_exclusive_scan(ltot, s_wave, &nkept);   // position of digit d's run in the chunk
        uint32_t run = lbase;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_cnt[k][d] = run; run += c[k]; }
        s_delta[d] = gbase - lbase;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = elem_index<ITEMS>(chunk_base, w, r, lane);
        if (i < n && (!DROP || key[r] != 0xFFFFFFFFu)) {
            const uint32_t lp = s_cnt[w][(key[r] >> shift) & mask] + rank[r];
            s_key[lp] = key[r];
            s_val[lp] = val[r];
        }
    }
    __syncthreads();
    const int nvalid = DROP ? (int)nkept : (int)min((int64_t)(SORT_THREADS * ITEMS), n - chunk_base);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int j = r * SORT_THREADS + threadIdx.x;
        if (j < nvalid) {
            const uint32_t k = s_key[j];
            const uint32_t pos = (uint32_t)j + s_delta[(k >> shift) & mask];
            keys_out[pos] = (KeyT)k;
            vals_out[pos] = s_val[j];
        }
    }
}

}  // namespace

hipError_t exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n,
                              uint64_t* partials, uint64_t* total, hipStream_t s, uint64_t* status,
                              uint64_t* host_mapped, uint64_t ticket, const uint32_t* n_ptr, uint32_t* clamped_total,
                              uint64_t clamp, const uint32_t* extra, uint32_t* zero_word, uint32_t* overflow_flag,
                              uint32_t in_mask, uint32_t* side_out, uint32_t* side_flag) {
    if ((in_mask != 0xFFFFFFFFu || side_out) && gather == nullptr) return hipErrorInvalidValue;
    if (n <= 0) {       // nothing to scan: one block publishes a total of zero (no partials are read)
        hipLaunchKernelGGL(scan_apply_fused_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, (int64_t)0,
                           (const uint64_t*)partials, (int64_t)0, (const uint32_t*)nullptr, total, status,
                           (volatile uint64_t*)host_mapped, ticket, clamped_total, clamp, extra, zero_word, overflow_flag);
        return hipGetLastError();
    }
    if (gather != nullptr && out == in) return hipErrorInvalidValue;     // a gathered scan is staged through `out`
    const int64_t nb = scan_blocks(n);
    // with a gather the first pass leaves the gathered values in `out` and the second pass scans `out` in place
    const bool stage = gather != nullptr;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, gather, n, partials,
                       stage ? out : (uint32_t*)nullptr, n_ptr, in_mask, side_out, side_out ? side_flag : (uint32_t*)nullptr);
    hipLaunchKernelGGL(scan_apply_fused_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s,
                       stage ? (const uint32_t*)out : in, out, n, (const uint64_t*)partials, nb, n_ptr, total, status,
                       (volatile uint64_t*)host_mapped, ticket, clamped_total, clamp, extra, zero_word, overflow_flag);
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, size_t n_words) {
    const size_t n4 = n_words >> 2;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < (n_words & 3)) p[(n4 << 2) + threadIdx.x] = 0u;
}
}  // namespace

hipError_t launch_zero(void* ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(ptr) & 15) || (bytes & 3)) return hipMemsetAsync(ptr, 0, bytes, s);
    const size_t words = bytes >> 2;
    size_t blocks = (words / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)ptr, words);
    return hipGetLastError();
}

static int passes_for(int begin_bit, int end_bit) {
    const int passes = (end_bit - begin_bit + 7) / 8;
    return passes < 1 ? 1 : passes;
}
typedef SortGeom GroupGeom;

// does radix_sort_pairs(n, bits) honour a device-side element count?  (every sort of up to 32 key bits does)
bool radix_sort_supports_device_count(int64_t n, int begin_bit, int end_bit) {
    return n > 0 && passes_for(begin_bit, end_bit) <= 4;
}

// may radix_sort_pairs(..., keys16 = true) be used for this sort?  (16-bit key arrays: 6 instead of 8 bytes per pair and pass)
bool radix_sort_keys16_ok(int64_t n, int begin_bit, int end_bit) {
    return end_bit <= 16 && radix_sort_supports_device_count(n, begin_bit, end_bit);
}

// The words radix_sort_pairs(n, bits) needs zeroed beforehand — the group-sum tables of its passes — so that a kernel running
// earlier on the stream can clear them and the sort can be called with pre_zeroed = true (one launch less).
bool radix_sort_zero_region(int64_t n, int begin_bit, int end_bit, char* scratch, uint32_t** ptr, size_t* words) {
    if (n <= 0) return false;
    const int passes = passes_for(begin_bit, end_bit);
    if (passes > 4) return false;
    const GroupGeom G(n);
    const SortScratch L(n);
    *ptr = reinterpret_cast<uint32_t*>(scratch + L.hist) + (size_t)256 * G.nb;
    *words = G.zero_words(passes);
    return true;
}

namespace {
__global__ void store_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }

// one radix pass with ITEMS keys per thread: block histograms + group sums, [the one-block scan of the group sums for inputs of
// >= 4096 blocks,] then a stable scatter through LDS that derives its own bases.
// drop / n_ptr / n_out: compaction (radix_hist_kernel)
template <int ITEMS, typename KeyT>
void launch_pass(const KeyT* src_k, const uint32_t* src_v, KeyT* dst_k, uint32_t* dst_v, int64_t n, int shift, uint32_t mask,
                 const GroupGeom& G, uint32_t* hist, uint32_t* gs, bool drop, const uint32_t* n_ptr, uint32_t* n_out,
                 hipStream_t s) {
    const dim3 grid((unsigned)G.nb), block(SORT_THREADS);
    if (drop) {
        hipLaunchKernelGGL((radix_hist_kernel<ITEMS, true, KeyT>), grid, block, 0, s, src_k, n, shift, mask, G.nb, hist, gs, G.gsize,
                           G.ngroups, (const uint32_t*)nullptr);
        hipLaunchKernelGGL((radix_scatter_kernel<ITEMS, true, KeyT>), grid, block, 0, s, src_k, src_v, dst_k, dst_v, n, shift,
                           mask, G.nb, hist, gs, G.gsize, G.ngroups, false, (const uint32_t*)nullptr, n_out);
        return;
    }
    hipLaunchKernelGGL((radix_hist_kernel<ITEMS, false, KeyT>), grid, block, 0, s, src_k, n, shift, mask, G.nb, hist, gs, G.gsize,
                       G.ngroups, n_ptr);
    if (G.scanned)
        hipLaunchKernelGGL(group_scan_kernel, dim3((unsigned)G.ngroups + 1), dim3(256), 0, s, gs, G.ngroups, hist, G.gsize, G.nb, n_ptr,
                           SORT_THREADS * ITEMS);
    hipLaunchKernelGGL((radix_scatter_kernel<ITEMS, false, KeyT>), grid, block, 0, s, src_k, src_v, dst_k, dst_v, n, shift, mask,
                       G.nb, hist, gs, G.gsize, G.ngroups, G.scanned, n_ptr, (uint32_t*)nullptr);
}

// all passes of one sort over KeyT key arrays (uint16_t: the tile sort whenever the tile ids and the sentinel fit 16 bits — the
// caller's 4-byte-per-key buffers are then simply half used).  Ping-pong so that the LAST pass writes keys_out / vals_out.
// Digit widths are balanced over the passes (13 tile bits -> 7 + 6, not 8 + 5): a block's keys of one digit leave as one run,
// and the run length — hence the write coalescing of the scatter — is set by the pass with the MOST digits.
template <typename KeyT>
hipError_t sort_passes(KeyT* keys_in, uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out, int64_t n, int passes,
                       int begin_bit, int end_bit, const GroupGeom& G, KeyT* keys_alt, uint32_t* vals_alt, uint32_t* hist,
                       uint32_t* gsum_all, bool compact, uint32_t* n_valid_dev, const uint32_t* n_dev, hipStream_t s) {
    const KeyT* src_k = keys_in;
    const uint32_t* src_v = vals_in;
    int next_shift = begin_bit;
    for (int p = 0; p < passes; ++p) {
        const int shift = next_shift;
        const int left = end_bit - shift;
        const int bits = (left + (passes - p) - 1) / (passes - p);
        next_shift = shift + bits;
        const uint32_t mask = bits >= 8 ? 0xFFu : ((1u << (bits > 0 ? bits : 1)) - 1u);
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        KeyT* dst_k = to_out ? keys_out : keys_alt;
        uint32_t* dst_v = to_out ? vals_out : vals_alt;
        uint32_t* gs = gsum_all + (size_t)p * 256 * G.ngroups;
        const bool drop = compact && p == 0;
        const uint32_t* np = n_dev ? n_dev : (compact && p > 0 ? n_valid_dev : nullptr);
        uint32_t* no = drop ? n_valid_dev : nullptr;
        if (G.big) launch_pass<16, KeyT>(src_k, src_v, dst_k, dst_v, n, shift, mask, G, hist, gs, drop, np, no, s);
        else if (G.mid) launch_pass<8, KeyT>(src_k, src_v, dst_k, dst_v, n, shift, mask, G, hist, gs, drop, np, no, s);
        else launch_pass<SORT_ITEMS, KeyT>(src_k, src_v, dst_k, dst_v, n, shift, mask, G, hist, gs, drop, np, no, s);
        src_k = dst_k;
        src_v = dst_v;
    }
    return hipGetLastError();
}
}  // namespace

// n_valid_dev (optional, device word): COMPACTING sort — pairs whose key is 0xFFFFFFFF are dropped by the first pass, the
// number of survivors V is written to *n_valid_dev, the remaining passes run over V pairs, and keys_out / vals_out hold the V
// sorted survivors (the tail beyond V is unspecified).  Inputs of >= 4096 blocks (16.8 M pairs) sort all n pairs (the dropped
// keys sort last) and report V = n: every consumer of *n_valid_dev stays correct either way.
// n_dev (optional, device word): the element count when the host only knows an upper bound n (speculative stage 2): every pass
// takes its count from it, grids and group geometry are those of n.
hipError_t radix_sort_pairs(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                            int64_t n, int begin_bit, int end_bit, char* scratch, hipStream_t s, bool pre_zeroed,
                            uint32_t* n_valid_dev, const uint32_t* n_dev, bool keys16) {
    if (n <= 0) {
        if (n_valid_dev) hipLaunchKernelGGL(store_u32_kernel, dim3(1), dim3(1), 0, s, n_valid_dev, 0u);
        return hipSuccess;
    }
    const int passes = passes_for(begin_bit, end_bit);
    if (passes > 4) return hipErrorInvalidValue;            // keys of up to 32 bits
    const SortScratch L(n);
    uint32_t* keys_alt = reinterpret_cast<uint32_t*>(scratch + L.keys_alt);
    uint32_t* vals_alt = reinterpret_cast<uint32_t*>(scratch + L.vals_alt);
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch + L.hist);
    // 16 keys per thread for big inputs (longer digit runs per block -> better write coalescing; measured on the
    // tile sort: 94 -> 81 us at 4.1M pairs, 1.6 -> 1.2 ms at 55M); 8 below that, where 16 would leave CUs idle
    const GroupGeom G(n);
    uint32_t* gsum_all = hist + (size_t)256 * G.nb;         // group sums live behind the block-histogram table
    if (!pre_zeroed) {
        hipError_t e = launch_zero(gsum_all, sizeof(uint32_t) * G.zero_words(passes), s);
        if (e != hipSuccess) return e;
    }
    if (keys16) {          // (the caller asked radix_sort_keys16_ok first; no compaction)
        if (n_valid_dev || end_bit > 16) return hipErrorInvalidValue;
        return sort_passes<uint16_t>(reinterpret_cast<uint16_t*>(keys_in), vals_in, reinterpret_cast<uint16_t*>(keys_out), vals_out,
                                     n, passes, begin_bit, end_bit, G, reinterpret_cast<uint16_t*>(keys_alt), vals_alt, hist,
                                     gsum_all, false, nullptr, n_dev, s);
    }
    const bool compact = n_valid_dev != nullptr && !G.scanned;
    if (n_valid_dev && !compact) hipLaunchKernelGGL(store_u32_kernel, dim3(1), dim3(1), 0, s, n_valid_dev, (uint32_t)n);
    return sort_passes<uint32_t>(keys_in, vals_in, keys_out, vals_out, n, passes, begin_bit, end_bit, G, keys_alt, vals_alt, hist,
                                 gsum_all, compact, n_valid_dev, n_dev, s);
}

}  // namespace msgs
