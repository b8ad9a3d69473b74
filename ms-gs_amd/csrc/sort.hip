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
// One pass = three kernels: per-block digit histogram -> exclusive scan of the digit-major
// histogram table -> stable scatter.  Stability inside a block comes from wave-level match-any
// (8 ballots) ranking with each wave owning a contiguous key segment, waves ordered by id.
// Every kernel is HBM-streaming: 4 B (hist) + 8 B read + 8 B write per pair per pass.
#include "msgs_internal.h"

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

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                                   const uint32_t* __restrict__ gather,
                                                                   int64_t n, uint64_t* __restrict__ partials) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i < n) sum += gather ? in[gather[i]] : in[i];
    }
    uint32_t total;
    block_exclusive_scan(sum, s_wave, &total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// single block: exclusive scan of the per-block totals (u64), grand total to partials[nb]
__global__ __launch_bounds__(256) void scan_partials_kernel(uint64_t* __restrict__ partials, int64_t nb,
                                                            uint64_t* __restrict__ total_out) {
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t start = 0; start < nb; start += 256) {
        const int64_t i = start + threadIdx.x;
        const uint64_t v = i < nb ? partials[i] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, off);
            const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), off);
            if (lane >= off) inc += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint64_t base = s_carry;
        for (int k = 0; k < w; ++k) base += s_w[k];
        if (i < nb) partials[i] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[nb] = s_carry;
        if (total_out) *total_out = s_carry;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t* __restrict__ in,
                                                                  const uint32_t* __restrict__ gather,
                                                                  uint32_t* __restrict__ out, int64_t n,
                                                                  const uint64_t* __restrict__ partials) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + k;
        v[k] = i < n ? (gather ? in[gather[i]] : in[i]) : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, s_wave, &total) + (uint32_t)partials[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// radix pass
// ---------------------------------------------------------------------------------------------
// Element layout inside a block's chunk: wave w owns [w*64*ITEMS, (w+1)*64*ITEMS); in round r lane l
// handles element w*64*ITEMS + r*64 + l.  Waves, rounds and lanes are therefore all in key order.
__device__ __forceinline__ int64_t elem_index(int64_t chunk_base, int w, int r, int lane) {
    return chunk_base + (int64_t)w * (64 * SORT_ITEMS) + r * 64 + lane;
}

__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                                  int shift, uint32_t mask, int64_t nblocks,
                                                                  uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t chunk_base = (int64_t)blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = elem_index(chunk_base, w, r, lane);
        if (i < n) atomicAdd(&s_hist[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = s_hist[threadIdx.x];
}

__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, int64_t n,
                                                                     int shift, uint32_t mask, int64_t nblocks,
                                                                     const uint32_t* __restrict__ hist_scanned) {
    __shared__ uint32_t s_cnt[4][256];   // per-wave digit counters, later per-wave global bases
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t chunk_base = (int64_t)blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_cnt[k][threadIdx.x] = 0;
    __syncthreads();

    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], rank[SORT_ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = elem_index(chunk_base, w, r, lane);
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
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
    {   // digit d = threadIdx.x: turn per-wave counts into per-wave global bases
        const uint32_t d = threadIdx.x;
        uint32_t base = hist_scanned[(int64_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = s_cnt[k][d];
            s_cnt[k][d] = base;
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = elem_index(chunk_base, w, r, lane);
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = s_cnt[w][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = val[r];
        }
    }
}

}  // namespace

hipError_t exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n,
                              uint64_t* partials, uint64_t* total, hipStream_t s) {
    const int64_t nb = scan_blocks(n > 0 ? n : 1);
    if (n <= 0) {
        hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(256), 0, s, partials, (int64_t)0, total);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, gather, n, partials);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(256), 0, s, partials, nb, total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, gather, out, n, partials);
    return hipGetLastError();
}

hipError_t radix_sort_pairs(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                            int64_t n, int begin_bit, int end_bit, char* scratch, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const SortScratch L(n);
    uint32_t* keys_alt = reinterpret_cast<uint32_t*>(scratch + L.keys_alt);
    uint32_t* vals_alt = reinterpret_cast<uint32_t*>(scratch + L.vals_alt);
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch + L.hist);
    uint64_t* partials = reinterpret_cast<uint64_t*>(scratch + L.partials);
    const int64_t nb = sort_blocks(n);
    int passes = (end_bit - begin_bit + 7) / 8;
    if (passes < 1) passes = 1;
    // ping-pong so that the LAST pass writes keys_out/vals_out
    const uint32_t* src_k = keys_in;
    const uint32_t* src_v = vals_in;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + 8 * p;
        const int bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const uint32_t mask = bits >= 8 ? 0xFFu : ((1u << (bits > 0 ? bits : 1)) - 1u);
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        uint32_t* dst_k = to_out ? keys_out : keys_alt;
        uint32_t* dst_v = to_out ? vals_out : vals_alt;
        hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nb), dim3(SORT_THREADS), 0, s, src_k, n, shift, mask, nb, hist);
        hipError_t e = exclusive_scan_u32(hist, nullptr, hist, 256 * nb, partials, nullptr, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nb), dim3(SORT_THREADS), 0, s, src_k, src_v, dst_k,
                           dst_v, n, shift, mask, nb, hist);
        src_k = dst_k;
        src_v = dst_v;
    }
    return hipGetLastError();
}

}  // namespace msgs
