// NOT PART OF THE PRODUCT (tools/experiments): the one-launch depth sort + scan of round 6, measured SLOWER than the ten-launch
// chain it was to replace (C2, 78 k keys: 161 us against 52 + 16 us — 32 resident workgroups walk 2-3 tiles per pass one after the
// other, every step a global round trip; profiles/r6_notes.md).  Kept for the record; not compiled by the Makefile.
// smallsort.hip — depth sort + exclusive scan of stage 1 in ONE launch, for views with few Gaussians (round 6).
//
// The radix sort of sort.hip is built for millions of keys: two launches per pass, every launch a grid of independent
// workgroups.  A view with 80 k visible Gaussians (BASELINE C2; every pyramid level k >= 1 of a trained MS-GS model renders
// 2 k - 110 k) spends its stage 1 on launch latency instead: eight sort launches + two scan launches = 52 + 16 us for 0.6 MB of
// keys, and the host — which waits for the instance count at the end of stage 1 — pays every one of them.  Here SS_GRID
// workgroups stay resident and walk the four 8-bit passes and the scan together, one grid barrier per step:
//     [histogram of the first digit] | pass 0 .. 3: bases from the histogram table, stable scatter, and — while an element is
//     scattered — its NEXT digit is counted for the workgroup that will own its new position | block sums of the tile counts |
//     exclusive scan, publication of the instance count (device words, the polled pinned host words: what scan_apply_fused_kernel
//     of sort.hip publishes)
// Same result as the large path, bit for bit: a stable LSD sort on the same key bits ((depth bits, Gaussian index) order), the
// compaction of the first pass (keys 0xFFFFFFFF = not rendered leave the sort), the same offsets.
// Every word one workgroup writes and another reads inside the launch travels through agent-scope (write-through / L2-bypassing)
// atomic stores and loads, so the barriers need no cache maintenance (cdna_hip_programming.md G16: per-XCD L2s are not coherent
// with each other); the barrier waits are bounded, and an expired wait raises the stage's error flag (MSGS_ERR_INTERNAL).
#include "msgs_internal.h"

namespace msgs {

namespace {

constexpr int SS_GRID = 32;                     // resident workgroups (of 256 CUs)
constexpr int SS_THREADS = 256;
constexpr int SS_ITEMS = 4;
constexpr int SS_TILE = SS_THREADS * SS_ITEMS;  // elements a workgroup ranks at a time
constexpr int SS_MAX_SPINS = 1 << 20;

#define SS_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct SmallSortState {                         // in the stage-1 scratch; zeroed by preprocess_kernel's zero job
    uint32_t bar[12];                           // arrival counters, one per barrier of the launch
    uint32_t watchdog;                          // non-zero: a barrier wait expired
    uint32_t pad[3];
    uint32_t hist[4][SS_GRID][256];             // [pass][workgroup that owns the INPUT range][digit]
    unsigned long long partial[SS_GRID];        // block sums of the scan
};

__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) { return __hip_atomic_load(p, SS_RLX); }
__device__ __forceinline__ void st_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, SS_RLX); }

// all waves drain what they issued, the workgroup meets, one lane arrives and polls; no cache maintenance (see the header)
__device__ __forceinline__ bool ss_barrier(SmallSortState* st, int k, uint32_t* s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&st->bar[k], 1u, SS_RLX);
        uint32_t ok = 1u;
        for (int spins = 0; __hip_atomic_load(&st->bar[k], SS_RLX) < gridDim.x; ++spins) {
            if (spins > SS_MAX_SPINS || __hip_atomic_load(&st->watchdog, SS_RLX) != 0u) {
                __hip_atomic_store(&st->watchdog, 1u, SS_RLX);
                ok = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0u;
}

__device__ __forceinline__ uint32_t block_excl_256(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < wv; ++k) base += s_w[k];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + inc - v;
}

// elements per workgroup for n elements: whole tiles, contiguous ranges in workgroup order
__device__ __forceinline__ uint32_t ss_chunk(uint32_t n, uint32_t g) {
    const uint32_t per = (n + g - 1) / g;
    return ((per + SS_TILE - 1) / SS_TILE) * SS_TILE;
}

struct SmallSortArgs {
    const uint32_t* keys_first;     // input of the first pass executed: geom.key (first_pass = 0) or the compacted keys of a DROP pass
    const uint32_t* vals_first;     // its values (nullptr: the element index)
    uint32_t* keys_out;             // geom.skey  (sorted depth keys)
    uint32_t* vals_out;             // geom.order (Gaussian ids in depth order)
    uint32_t* keys_alt;             // ping-pong partner (stage-1 scratch)
    uint32_t* vals_alt;
    uint32_t P;                     // elements of pass 0 when first_pass = 0
    uint32_t* n_valid;              // geom.nvalid: V — written here when first_pass = 0, read otherwise
    int first_pass;                 // 0: all four passes (the first one compacting); 1: passes 1 .. 3 behind a DROP pass of sort.hip
    SmallSortState* st;
    const uint32_t* tiles;          // per-Gaussian instance counts (index order)
    uint32_t* offs;                 // out: exclusive scan of tiles[order[r]]
    // publication of the total (exclusive_scan_u32's contract)
    uint64_t* total_out;
    uint64_t* status;
    volatile uint64_t* host;
    uint64_t ticket;
    uint32_t* clamped_total;
    uint64_t clamp;
    const uint32_t* extra;
    uint32_t* zero_word;
    uint32_t* err_flag;             // raised when a barrier wait expired
};

__global__ __launch_bounds__(SS_THREADS) void small_sort_scan_kernel(SmallSortArgs a) {
    __shared__ uint32_t s_cnt[4][256];          // per-wave digit counts of the current tile
    __shared__ uint32_t s_wbase[4][256];        // global position of each wave's first element of a digit
    __shared__ uint32_t s_base[256];            // running global position per digit of this workgroup
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_w[4], s_ok;
    SmallSortState* st = a.st;
    const uint32_t g = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    int bar = 0;
    // a barrier wait expired: raise the stage's error flag and release the host, which polls for the count (MSGS_ERR_INTERNAL)
    auto fail = [&]() {
        if (tid != 0) return;
        if (a.err_flag) st_u32(a.err_flag, 1u);
        if (a.status) { a.status[0] = 0; a.status[1] = 1; a.status[2] = 0; }
        if (a.host) { a.host[0] = 0; a.host[1] = 1; a.host[3] = 0; __threadfence_system(); a.host[2] = a.ticket; }
    };

    // ---- histogram of the first pass executed, over this workgroup's input range
    uint32_t n = a.first_pass == 0 ? a.P : *a.n_valid;                 // (n_valid: written by an earlier kernel)
    {
        const bool drop = a.first_pass == 0;
        const int shift = 8 * a.first_pass;
        const uint32_t C = ss_chunk(n, G), lo = min(n, g * C), hi = min(n, lo + C);
        s_hist[tid] = 0u;
        __syncthreads();
        for (uint32_t i = lo + tid; i < hi; i += SS_THREADS) {
            const uint32_t k = a.keys_first[i];
            if (!drop || k != 0xFFFFFFFFu) atomicAdd(&s_hist[(k >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        st_u32(&st->hist[a.first_pass][g][tid], s_hist[tid]);
    }
    if (!ss_barrier(st, bar++, &s_ok)) { fail(); return; }

    const uint32_t* src_k = a.keys_first;
    const uint32_t* src_v = a.vals_first;
    bool src_shared = false;                    // the source was written inside this launch (read it past the caches)
    for (int p = a.first_pass; p < 4; ++p) {
        const bool drop = p == 0;
        const int shift = 8 * p;
        const bool to_out = ((3 - p) % 2) == 0;
        uint32_t* dst_k = to_out ? a.keys_out : a.keys_alt;
        uint32_t* dst_v = to_out ? a.vals_out : a.vals_alt;
        // ---- bases: digit d of this workgroup starts behind every smaller digit and behind digit d of the workgroups in front
        uint32_t tot = 0, before = 0;
        for (uint32_t q = 0; q < G; ++q) {
            const uint32_t v = ld_u32(&st->hist[p][q][tid]);
            tot += v;
            before += q < g ? v : 0u;
        }
        uint32_t kept;
        const uint32_t digit_base = block_excl_256(tot, s_w, &kept);
        s_base[tid] = digit_base + before;
        const uint32_t n_next = drop ? kept : n;                        // elements of the next pass (the first pass compacts)
        if (drop && g == 0 && tid == 0) *a.n_valid = kept;              // V (read by later kernels)
        const uint32_t C = ss_chunk(n, G), lo = min(n, g * C), hi = min(n, lo + C);
        const uint32_t C_next = max(ss_chunk(n_next, G), 1u);
        __syncthreads();
        // ---- stable scatter, tile by tile; the next digit of every element is counted for the owner of its new position
        for (uint32_t t0 = lo; t0 < hi; t0 += SS_TILE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_cnt[k][tid] = 0u;
            __syncthreads();
            uint32_t key[SS_ITEMS], val[SS_ITEMS], rank[SS_ITEMS];
            bool valid[SS_ITEMS];
#pragma unroll
            for (int r = 0; r < SS_ITEMS; ++r) {
                const uint32_t i = t0 + (uint32_t)(wv * 64 * SS_ITEMS + r * 64 + lane);      // waves, rounds, lanes in key order
                key[r] = i < hi ? (src_shared ? ld_u32(src_k + i) : src_k[i]) : 0xFFFFFFFFu;
                valid[r] = i < hi && (!drop || key[r] != 0xFFFFFFFFu);
                val[r] = valid[r] ? (src_v ? (src_shared ? ld_u32(src_v + i) : src_v[i]) : i) : 0u;
                const uint32_t d = (key[r] >> shift) & 0xFFu;
                uint64_t peers = __ballot(valid[r]);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const uint64_t m = __ballot((d >> b) & 1u);
                    peers &= ((d >> b) & 1u) ? m : ~m;
                }
                const uint32_t prev = s_cnt[wv][d];
                const uint32_t bef = (uint32_t)__popcll(peers & lt_mask);
                rank[r] = prev + bef;
                if (valid[r] && bef == 0) s_cnt[wv][d] = prev + (uint32_t)__popcll(peers);   // (LDS operations of a wave are in order)
            }
            __syncthreads();
            {
                uint32_t run = s_base[tid];
#pragma unroll
                for (int k = 0; k < 4; ++k) { s_wbase[k][tid] = run; run += s_cnt[k][tid]; }
                s_base[tid] = run;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < SS_ITEMS; ++r) {
                if (!valid[r]) continue;
                const uint32_t pos = s_wbase[wv][(key[r] >> shift) & 0xFFu] + rank[r];
                st_u32(dst_k + pos, key[r]);
                st_u32(dst_v + pos, val[r]);
                if (p < 3) atomicAdd(&st->hist[p + 1][pos / C_next][(key[r] >> (shift + 8)) & 0xFFu], 1u);
            }
            __syncthreads();
        }
        n = n_next;
        src_k = dst_k;
        src_v = dst_v;
        src_shared = true;
        if (!ss_barrier(st, bar++, &s_ok)) { fail(); return; }
    }

    // ---- exclusive scan of the instance counts in depth order (n = V; the order sits in a.vals_out)
    const uint32_t C = ss_chunk(n, G), lo = min(n, g * C), hi = min(n, lo + C);
    {
        unsigned long long sum = 0;
        for (uint32_t i = lo + tid; i < hi; i += SS_THREADS) sum += a.tiles[ld_u32(a.vals_out + i)];
        for (int off = 32; off > 0; off >>= 1)
            sum += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(sum >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)sum, off);
        __shared__ unsigned long long s_sum[4];
        if (lane == 0) s_sum[wv] = sum;
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&st->partial[g], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3], SS_RLX);
    }
    if (!ss_barrier(st, bar++, &s_ok)) { fail(); return; }
    unsigned long long base = 0, total = 0;
    for (uint32_t q = 0; q < G; ++q) {
        const unsigned long long v = __hip_atomic_load(&st->partial[q], SS_RLX);
        total += v;
        base += q < g ? v : 0ull;
    }
    if (g == 0 && tid == 0) {       // publication (scan_apply_fused_kernel's contract, sort.hip)
        if (a.total_out) *a.total_out = total;
        if (a.zero_word) *a.zero_word = 0u;
        if (a.clamped_total) *a.clamped_total = (uint32_t)(total < a.clamp ? total : a.clamp);
        const uint64_t info = a.extra ? ((uint64_t)a.extra[0] | ((uint64_t)a.extra[1] << 32)) : 0ull;
        if (a.status) { a.status[0] = total; a.status[1] = 0; a.status[2] = info; }
        if (a.host) {
            a.host[0] = total;
            a.host[1] = 0;
            a.host[3] = info;
            __threadfence_system();
            a.host[2] = a.ticket;
        }
    }
    uint32_t run = (uint32_t)base;
    for (uint32_t t0 = lo; t0 < hi; t0 += SS_THREADS) {
        const uint32_t i = t0 + tid;
        const uint32_t c = i < hi ? a.tiles[ld_u32(a.vals_out + i)] : 0u;
        uint32_t tile_total;
        const uint32_t ex = block_excl_256(c, s_w, &tile_total);
        if (i < hi) a.offs[i] = run + ex;
        run += tile_total;
    }
}

}  // namespace

static_assert(sizeof(SmallSortState) == SMALL_SORT_STATE_BYTES, "Stage1Scratch::small_state is sized with SMALL_SORT_STATE_BYTES");
size_t small_sort_state_bytes() { return sizeof(SmallSortState); }

// Depth sort (keys geom.key, compacting, stable) + exclusive scan of tiles[order[r]] in one launch.  first_pass = 1: a DROP pass
// of sort.hip already left the compacted pairs in keys_alt / vals_alt and V in *n_valid.
hipError_t launch_small_sort_scan(const uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, uint32_t* keys_alt,
                                  uint32_t* vals_alt, uint32_t P, uint32_t* n_valid, int first_pass, void* state,
                                  const uint32_t* tiles, uint32_t* offs, uint64_t* total_out, uint64_t* status,
                                  uint64_t* host_mapped, uint64_t ticket, uint32_t* clamped_total, uint64_t clamp,
                                  const uint32_t* extra, uint32_t* zero_word, uint32_t* err_flag, hipStream_t s) {
    SmallSortArgs a;
    a.keys_first = first_pass == 0 ? keys_in : keys_alt;
    a.vals_first = first_pass == 0 ? nullptr : vals_alt;
    a.keys_out = keys_out; a.vals_out = vals_out; a.keys_alt = keys_alt; a.vals_alt = vals_alt;
    a.P = P; a.n_valid = n_valid; a.first_pass = first_pass;
    a.st = reinterpret_cast<SmallSortState*>(state);
    a.tiles = tiles; a.offs = offs;
    a.total_out = total_out; a.status = status; a.host = (volatile uint64_t*)host_mapped; a.ticket = ticket;
    a.clamped_total = clamped_total; a.clamp = clamp; a.extra = extra; a.zero_word = zero_word; a.err_flag = err_flag;
    hipLaunchKernelGGL(small_sort_scan_kernel, dim3(SS_GRID), dim3(SS_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace msgs
