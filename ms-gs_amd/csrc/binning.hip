// binning.hip — instance emission and tile ranges (App. A.2).
//
// emit_kernel walks the Gaussians in DEPTH ORDER (order[] from the depth sort) and writes one
// (tile id, Gaussian id) pair per tile that the Gaussian's alpha >= 1/255 level set reaches (per-row extents
// with a slightly smaller margin than the count, see msgs_internal.h), at the slot given by the exclusive scan of the per-Gaussian counts.  The subsequent
// stable sort by tile id (sort.hip) then produces, inside every tile, the reference order
// (depth bits ascending, ties by Gaussian index).
//
// ranges_kernel finds the [start, end) slice of every tile in the sorted key array.
// Both are HBM-streaming: 8 B written per instance (emit), 4 B read per instance (ranges).
#include "msgs_internal.h"

#include <algorithm>
#include <cstdlib>

namespace msgs {

namespace {

// Block b handles ranks [256 b, 256 b + 256): their output slots form ONE contiguous range
// [offs[first], offs[last] + tiles[last]).  When that range fits the LDS stage (the normal case: ~4 tiles
// per Gaussian) every thread deposits its pairs in LDS and the block streams the stage out with fully
// coalesced stores; otherwise (huge Gaussians) threads store straight to global memory.
// LDS stage in pairs: 3072 for light scenes, 12288 when a block of 256 Gaussians emits more than the small stage on average
// (D > 8 P; C5: 11 tiles per Gaussian, 20 per rendered one).  The kernel is latency-bound and its occupancy is set by the
// stage, so a staged pair is 3 bytes, not 8: the tile id as KeyT (16 bits whenever the grid has fewer than 65535 tiles) and
// the owner's thread number (8 bits; the 256 Gaussian ids of the block sit in LDS once).

// Depth-slab binning (msgs_view_t.slab_fraction): the same kernel emits slab A — the ranks below SlabArgs::V_dev, whose slots
// end at *D_dev = offs[rA] — and, with SLAB_B, slab B: EVERY rank, but only into the tiles whose bit is set in open_bits (the tiles
// slab A left open), at the slots of the second scan (offs_b).  The tile set of a Gaussian is computed by the same code from the
// same inputs in all three uses, so a tile's slab-A list is exactly the head of its complete list.
struct SlabArgs {
    const uint32_t* V_dev;        // number of ranks to emit (nullptr: GeomLayout::nvalid)
    const uint32_t* offs;         // exclusive scan of the per-rank instance counts (nullptr: GeomLayout::offs)
    const uint32_t* open_bits;    // SLAB_B: bit t = tile t receives instances
};
__device__ __forceinline__ bool tile_open(const uint32_t* __restrict__ bits, uint32_t t) { return (bits[t >> 5] >> (t & 31u)) & 1u; }
// set bits of `bits` in [a, b] (inclusive)
__device__ __forceinline__ uint32_t open_in_range(const uint32_t* __restrict__ bits, uint32_t a, uint32_t b) {
    uint32_t c = 0;
    for (uint32_t w = a >> 5; w <= (b >> 5); ++w) {
        uint32_t m = bits[w];
        if (w == (a >> 5)) m &= 0xFFFFFFFFu << (a & 31u);
        if (w == (b >> 5)) m &= 0xFFFFFFFFu >> (31u - (b & 31u));
        c += (uint32_t)__popc(m);
    }
    return c;
}

// OutT: the element type of the key array in HBM — uint16_t when the tile sort runs on 16-bit keys (radix_sort_keys16_ok)
// emit_rank_block: the work of ONE block of 256 consecutive depth ranks (rank block `bx`); every `return` is workgroup-uniform.
template <int EMIT_STAGE, typename KeyT, typename OutT, bool SLAB_B>
__device__ __forceinline__ void emit_rank_block(const int bx, const ViewParams& vp, int P, const char* __restrict__ geom,
                                                OutT* __restrict__ keys, uint32_t* __restrict__ ids,
                                                int64_t D, const uint32_t* __restrict__ D_dev,
                                                uint32_t* __restrict__ heavy_q, const SlabArgs& sl) {
    // ranks 0 .. V-1 of the depth order are the Gaussians that stayed in the compacting depth sort (GeomLayout::nvalid)
    __shared__ KeyT s_keys[EMIT_STAGE];
    __shared__ uint8_t s_own[EMIT_STAGE];
    __shared__ uint32_t s_gi[256];
    __shared__ int64_t s_range[2];
    const GeomLayout L(P);
    const uint32_t* order = reinterpret_cast<const uint32_t*>(geom + L.order);
    const uint32_t* offs = sl.offs ? sl.offs : reinterpret_cast<const uint32_t*>(geom + L.offs);
    const uint32_t* __restrict__ obits = sl.open_bits;
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + L.binrec);
    // occlusion cut-off (occlusion.hip): a tile whose cut-off depth key lies in front of this Gaussian's key receives no
    // instance of it — the same predicate the recount applied to the counts the scan summed
    const OccHeader* occ = reinterpret_cast<const OccHeader*>(geom + L.occ_hdr);
    const bool occ_on = occ->enabled != 0u && occ->any_closed != 0u;             // block-uniform (scalar loads)
    __shared__ OccTable T;                                 // cut-off key per cover block (loaded only when something closed)
    const int occ_lb = (int)occ->block_log2, occ_nbx = (int)occ->nbx;

    if (D_dev) D = (int64_t)*D_dev;      // speculative stage 2: min(instance count, capacity), from the scan
    const int V = (int)(sl.V_dev ? *sl.V_dev : *reinterpret_cast<const uint32_t*>(geom + L.nvalid));
    const int r0 = bx * (int)blockDim.x;
    if (r0 >= V) return;
    if (SLAB_B) {       // normally a handful of tiles are open: most workgroups have nothing to emit and learn it from two words
        const int64_t lo = offs[r0], hi = r0 + (int)blockDim.x < V ? (int64_t)offs[r0 + blockDim.x] : D;
        if (hi <= lo) return;
    }
    uint32_t cut_min = 0xFFFFFFFFu;
    if (occ_on) {
        occ_table_load(T, reinterpret_cast<const uint32_t*>(geom + L.occ_cut), occ_nbx, (int)occ->nby);
        cut_min = T.cut_min;
    }
    const int r = r0 + threadIdx.x;
    const int rlast = min(r0 + (int)blockDim.x, V) - 1;
    uint32_t gi = 0, count = 0;
    int64_t off = 0;
    // the count of rank r is the difference of consecutive scanned offsets (coalesced; no gather of tiles[order[r]])
    uint32_t kmine = 0;
    if (r < V) { gi = order[r]; off = offs[r]; count = (uint32_t)((r + 1 < V ? (int64_t)offs[r + 1] : D) - off); }
    if (cut_min != 0xFFFFFFFFu && r < V) kmine = occ_bucket(reinterpret_cast<const uint32_t*>(geom + L.skey)[r]);   // depth bucket
    const bool cut_check = kmine > cut_min;       // some tile may be closed in front of this Gaussian
    s_gi[threadIdx.x] = gi;
    if (threadIdx.x == 0) s_range[0] = off;
    if (r == rlast) s_range[1] = min((int64_t)off + count, D);
    __syncthreads();
    const int64_t blk_lo = s_range[0], blk_hi = s_range[1];
    const int64_t blk_len = blk_hi - blk_lo;
    if (blk_len <= 0) return;
    const bool staged = blk_len <= EMIT_STAGE;

    // A Gaussian with many instances is emitted by its whole WAVE (below): one thread looping over thousands of tiles while 63
    // lanes wait made the emit 4.3 ms for the 427 M instances of a multi-scale model rendered without its filters
    // (render.py's defaults; 145 k Gaussians wider than 256 px), i.e. 0.8 TB/s of stores.
    constexpr uint32_t HEAVY_MIN = EMIT_HEAVY_MIN;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (count) { q0 = binrec[gi].q0; q1 = binrec[gi].q1; }
    // a Gaussian behind some cut-off with a very large rect walks its tiles with the whole wave as well, however few of them
    // are left; up to OCC_LIGHT_RECT tiles of rect its own thread does it, skipping closed rows of blocks and closed blocks
    // whole (the wave path handles ONE Gaussian at a time: a wave of 64 medium footprints took 64 turns, 0.48 ms per view)
    const uint32_t rect_area = ((__float_as_uint(q1.w) & 0xFFFFu) - (__float_as_uint(q1.z) & 0xFFFFu)) *
                               ((__float_as_uint(q1.w) >> 16) - (__float_as_uint(q1.z) >> 16));
    // In a block that writes straight to HBM the Gaussians with more than HEAVY_MIN instances are not emitted here: their ranks
    // go into a queue and emit_heavy_kernel gives each a workgroup of its own.  (The nearest ranks of a view that is cut off by a
    // few opaque covers are ALL such Gaussians: one wave taking 11 of them in turn, 25 us each, was 0.28 ms for 1.4 M instances.)
    const bool queued = heavy_q != nullptr && !staged && count > HEAVY_MIN;
    const bool heavy = !queued && (count > HEAVY_MIN || (count > 0 && cut_check && rect_area > (uint32_t)OCC_LIGHT_RECT));
    {
        const uint64_t qm = __ballot(queued);
        if (qm) {
            const int lane_ = threadIdx.x & 63;
            uint32_t qbase = 0;
            if (lane_ == 0) qbase = atomicAdd(heavy_q, (uint32_t)__popcll(qm));
            qbase = lane_bcast(qbase, 0);
            const uint32_t qi = qbase + (uint32_t)__popcll(qm & ((1ull << lane_) - 1ull));
            if (queued && qi < (uint32_t)(D / EMIT_HEAVY_MIN + 64)) heavy_q[1 + qi] = (uint32_t)r;
        }
    }
    auto put = [&](int64_t at, uint32_t k, uint32_t g_id, uint32_t owner) {
        if (staged) { s_keys[at - blk_lo] = (KeyT)k; s_own[at - blk_lo] = (uint8_t)owner; }
        else { keys[at] = (OutT)k; ids[at] = g_id; }
    };
    if (count && !heavy && !queued) {
        const int64_t end = min((int64_t)off + count, D);
        const uint32_t rcx = __float_as_uint(q1.z), rcy = __float_as_uint(q1.w);
        const int minx = rcx & 0xFFFF, miny = rcx >> 16, maxx = rcy & 0xFFFF, maxy = rcy >> 16;
        const float conC = q1.x;
        const float tau2 = q1.y;
        const bool test = tau2 > -1.0e38f;
        const LevelSetRows ls = test ? levelset_rows_setup(q0.z, q0.w, conC, tau2) : LevelSetRows{};
        for (int ty = miny; ty < maxy && off < end; ++ty) {
            if (cut_check && T.rowmax[ty >> occ_lb] < kmine) continue;          // this row of blocks is closed at this depth
            int tlo = minx, thi = maxx - 1;
            if (test && !levelset_row_interval(ls, q0.x, q0.y, ty, minx, maxx, LEVELSET_MARGIN_EMIT, tlo, thi)) continue;
            if (!cut_check) {
                for (int tx = tlo; tx <= thi && off < end; ++tx) {
                    if (SLAB_B && !tile_open(obits, (uint32_t)(ty * vp.gx + tx))) continue;
                    put(off, (uint32_t)(ty * vp.gx + tx), gi, threadIdx.x);
                    ++off;
                }
            } else {                                       // cover block by cover block: a closed one is skipped whole
                const int brow = (ty >> occ_lb) * occ_nbx;
                int tx = tlo;
                while (tx <= thi && off < end) {
                    const int bend = min(thi, (((tx >> occ_lb) + 1) << occ_lb) - 1);
                    if (T.cut[brow + (tx >> occ_lb)] >= kmine) {
                        for (; tx <= bend && off < end; ++tx) {
                            if (SLAB_B && !tile_open(obits, (uint32_t)(ty * vp.gx + tx))) continue;
                            put(off, (uint32_t)(ty * vp.gx + tx), gi, threadIdx.x);
                            ++off;
                        }
                    } else {
                        tx = bend + 1;
                    }
                }
            }
        }
        // count >= emitted by construction (larger margin in the count): park the surplus slots on the sentinel tile
        for (; off < end; ++off) put(off, (uint32_t)(vp.gx * vp.gy), gi, threadIdx.x);
    }
    // heavy Gaussians, one after the other, by all 64 lanes of their wave: lane <-> tile row for the row intervals (the same
    // levelset_row_interval as above and as the count: same bits), then row by row with the lanes on consecutive tiles —
    // consecutive output slots, so the stores coalesce
    {
        const int lane = threadIdx.x & 63;
        uint64_t hv = __ballot(heavy);
        while (hv) {
            const int src = __ffsll((long long)hv) - 1;
            hv &= hv - 1;
            const uint32_t h_gi = lane_bcast(gi, src);
            const uint32_t h_owner = (uint32_t)((threadIdx.x & ~63) + src);
            const int64_t h_off = ((int64_t)lane_bcast((uint32_t)(off >> 32), src) << 32) | lane_bcast((uint32_t)off, src);
            const uint32_t h_count = lane_bcast(count, src);
            const uint32_t h_key = lane_bcast(kmine, src);
            const bool h_check = h_key > cut_min;
            const int64_t h_end = min(h_off + (int64_t)h_count, D);
            const float gx_ = lane_bcast(q0.x, src), gy_ = lane_bcast(q0.y, src), cA = lane_bcast(q0.z, src), cBh = lane_bcast(q0.w, src);
            const float cC = lane_bcast(q1.x, src), tau2 = lane_bcast(q1.y, src);
            const uint32_t rcx = __float_as_uint(lane_bcast(q1.z, src)), rcy = __float_as_uint(lane_bcast(q1.w, src));
            const int minx = rcx & 0xFFFF, miny = rcx >> 16, maxx = rcy & 0xFFFF, maxy = rcy >> 16;
            const bool test = tau2 > -1.0e38f;
            const LevelSetRows ls = test ? levelset_rows_setup(cA, cBh, cC, tau2) : LevelSetRows{};
            int64_t at = h_off;                                        // wave-uniform write position
            for (int row0 = miny; row0 < maxy && at < h_end; row0 += 64) {
                const int ty = row0 + lane;
                int tlo = minx, thi = maxx - 1;
                bool hit = ty < maxy;
                if (hit && h_check) hit = T.rowmax[ty >> occ_lb] >= h_key;   // rows of blocks that are closed at this depth
                if (hit && test) hit = levelset_row_interval(ls, gx_, gy_, ty, minx, maxx, LEVELSET_MARGIN_EMIT, tlo, thi);
                const int n_row = hit ? thi - tlo + 1 : 0;
                uint64_t rows = __ballot(n_row > 0);
                while (rows && at < h_end) {                           // wave-uniform: the rows that receive something
                    const int r = __ffsll((long long)rows) - 1;
                    rows &= rows - 1;
                    const int n_r = lane_bcast(n_row, r);
                    const int tlo_r = lane_bcast(tlo, r);
                    const uint32_t kbase = (uint32_t)((row0 + r) * vp.gx + tlo_r);
                    if (!h_check && !SLAB_B) {
                        for (int j = lane; j < n_r; j += 64)
                            if (at + j < h_end) put(at + j, kbase + (uint32_t)j, h_gi, h_owner);
                        at += n_r;
                    } else {                                           // only the tiles still open at this depth, compacted
                        for (int j0 = 0; j0 < n_r; j0 += 64) {
                            const int j = j0 + lane;
                            const bool keep = j < n_r &&
                                              (!h_check || T.cut[((row0 + r) >> occ_lb) * occ_nbx + ((tlo_r + j) >> occ_lb)] >= h_key) &&
                                              (!SLAB_B || tile_open(obits, kbase + (uint32_t)j));
                            const uint64_t km = __ballot(keep);
                            const int64_t pos = at + __popcll(km & ((1ull << lane) - 1ull));
                            if (keep && pos < h_end) put(pos, kbase + (uint32_t)j, h_gi, h_owner);
                            at += __popcll(km);
                        }
                    }
                }
            }
            for (int64_t a = at + lane; a < h_end; a += 64) put(a, (uint32_t)(vp.gx * vp.gy), h_gi, h_owner);   // surplus -> sentinel
        }
    }
    if (!staged) return;
    __syncthreads();
    for (int i = threadIdx.x; i < (int)blk_len; i += blockDim.x) {
        keys[blk_lo + i] = (OutT)s_keys[i];
        ids[blk_lo + i] = s_gi[s_own[i]];
    }
}

template <int EMIT_STAGE, typename KeyT, typename OutT = uint32_t, bool SLAB_B = false>
__global__ __launch_bounds__(256) void emit_kernel(ViewParams vp, int P, const char* __restrict__ geom,
                                                   OutT* __restrict__ keys, uint32_t* __restrict__ ids,
                                                   int64_t D, ZeroJob zj, const uint32_t* __restrict__ D_dev,
                                                   uint32_t* __restrict__ heavy_q, SlabArgs sl) {
    {   // housekeeping for the launches that follow: the tile sort's group-sum table and the tile-range array
        const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
        for (size_t t = t0; t < zj.n0; t += nt) zj.p0[t] = 0u;
        for (size_t t = t0; t < zj.n1; t += nt) zj.p1[t] = 0u;
    }
    // (slab B on a capped grid that walks the rank blocks was tried: 2048 workgroups x 9.5 blocks 50 us against 40 us for 19 531
    //  workgroups that mostly learn from two words that they have nothing to emit — fewer workgroups hide less latency)
    emit_rank_block<EMIT_STAGE, KeyT, OutT, SLAB_B>((int)blockIdx.x, vp, P, geom, keys, ids, D, D_dev, heavy_q, sl);
}

// One workgroup per queued Gaussian (emit_kernel above): all four waves compute the Gaussian's row extents (lane <-> tile row,
// 64 rows at a time) and the exclusive scan of the rows' instance counts, so every wave knows where every row starts without
// talking to the others; wave w then writes the rows r with r % 4 == w, lanes on consecutive tiles.  Behind an occlusion cut-off
// a row's count is its OPEN tiles (cover block by cover block, from the table in LDS) and the stores are compacted by ballot.
template <typename OutT, bool SLAB_B = false>
__global__ __launch_bounds__(256) void emit_heavy_kernel(ViewParams vp, int P, const char* __restrict__ geom,
                                                         OutT* __restrict__ keys, uint32_t* __restrict__ ids, int64_t D,
                                                         const uint32_t* __restrict__ D_dev,
                                                         const uint32_t* __restrict__ heavy_q, SlabArgs sl) {
    __shared__ OccTable T;
    // (never more than D / EMIT_HEAVY_MIN + 1 entries: the queued Gaussians start below D and lie more than EMIT_HEAVY_MIN apart)
    const uint32_t n = min(heavy_q[0], (uint32_t)(D / EMIT_HEAVY_MIN + 64));
    if (blockIdx.x >= n) return;
    const GeomLayout L(P);
    const uint32_t* order = reinterpret_cast<const uint32_t*>(geom + L.order);
    const uint32_t* offs = sl.offs ? sl.offs : reinterpret_cast<const uint32_t*>(geom + L.offs);
    const uint32_t* __restrict__ obits = sl.open_bits;
    const uint32_t* skey = reinterpret_cast<const uint32_t*>(geom + L.skey);
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + L.binrec);
    const OccHeader* occ = reinterpret_cast<const OccHeader*>(geom + L.occ_hdr);
    const bool occ_on = occ->enabled != 0u && occ->any_closed != 0u;
    const int lb = (int)occ->block_log2, nbx = (int)occ->nbx;
    if (occ_on) occ_table_load(T, reinterpret_cast<const uint32_t*>(geom + L.occ_cut), nbx, (int)occ->nby);
    if (D_dev) D = (int64_t)*D_dev;
    const int V = (int)(sl.V_dev ? *sl.V_dev : *reinterpret_cast<const uint32_t*>(geom + L.nvalid));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t sentinel = (uint32_t)(vp.gx * vp.gy);
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const int r = (int)heavy_q[1 + q];
        const uint32_t gi = order[r];
        const int64_t off = offs[r];
        const int64_t end = min(off + (int64_t)(uint32_t)((r + 1 < V ? (int64_t)offs[r + 1] : D) - off), D);
        const BinRec b = binrec[gi];
        const uint32_t kb = occ_on ? occ_bucket(skey[r]) : 0u;
        const bool check = occ_on && kb > T.cut_min;
        const uint32_t rcx = __float_as_uint(b.q1.z), rcy = __float_as_uint(b.q1.w);
        const int minx = rcx & 0xFFFF, miny = rcx >> 16, maxx = rcy & 0xFFFF, maxy = rcy >> 16;
        const float tau2 = b.q1.y;
        const bool test = tau2 > -1.0e38f;
        const LevelSetRows ls = test ? levelset_rows_setup(b.q0.z, b.q0.w, b.q1.x, tau2) : LevelSetRows{};
        int64_t at = off;                                                  // the same in all four waves
        for (int row0 = miny; row0 < maxy && at < end; row0 += 64) {
            const int ty = row0 + lane;
            int tlo = minx, thi = maxx - 1;
            bool hit = ty < maxy;
            if (hit && check) hit = T.rowmax[ty >> lb] >= kb;
            if (hit && test) hit = levelset_row_interval(ls, b.q0.x, b.q0.y, ty, minx, maxx, LEVELSET_MARGIN_EMIT, tlo, thi);
            int n_row = 0;
            if (hit && !check) n_row = SLAB_B ? (int)open_in_range(obits, (uint32_t)(ty * vp.gx + tlo), (uint32_t)(ty * vp.gx + thi))
                                              : thi - tlo + 1;
            else if (hit) {
                const int brow = (ty >> lb) * nbx;
                for (int tx = tlo; tx <= thi;) {
                    const int bend = min(thi, (((tx >> lb) + 1) << lb) - 1);
                    if (T.cut[brow + (tx >> lb)] >= kb)
                        n_row += SLAB_B ? (int)open_in_range(obits, (uint32_t)(ty * vp.gx + tx), (uint32_t)(ty * vp.gx + bend))
                                        : bend - tx + 1;
                    tx = bend + 1;
                }
            }
            int inc = n_row;                                               // row starts: exclusive scan over the lanes
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(inc, o);
                if (lane >= o) inc += up;
            }
            const int chunk_total = __shfl(inc, 63);
            const int excl = inc - n_row;
            uint64_t rows = __ballot(n_row > 0);
            while (rows) {
                const int rr = __ffsll((long long)rows) - 1;
                rows &= rows - 1;
                if ((rr & 3) != wv) continue;
                const int64_t start = at + lane_bcast(excl, rr);
                const int tlo_r = lane_bcast(tlo, rr), thi_r = lane_bcast(thi, rr);
                const uint32_t kbase = (uint32_t)((row0 + rr) * vp.gx);
                if (!check && !SLAB_B) {
                    for (int j = lane; tlo_r + j <= thi_r; j += 64) {
                        const int64_t pos = start + j;
                        if (pos < end) { keys[pos] = (OutT)(kbase + (uint32_t)(tlo_r + j)); ids[pos] = gi; }
                    }
                } else {
                    const int brow = ((row0 + rr) >> lb) * nbx;
                    int64_t p = start;
                    for (int j0 = 0; tlo_r + j0 <= thi_r; j0 += 64) {
                        const int tx = tlo_r + j0 + lane;
                        const bool keep = tx <= thi_r && (!check || T.cut[brow + (tx >> lb)] >= kb) &&
                                          (!SLAB_B || tile_open(obits, kbase + (uint32_t)tx));
                        const uint64_t km = __ballot(keep);
                        const int64_t pos = p + __popcll(km & lt);
                        if (keep && pos < end) { keys[pos] = (OutT)(kbase + (uint32_t)tx); ids[pos] = gi; }
                        p += __popcll(km);
                    }
                }
            }
            at += chunk_total;
        }
        // count >= emitted (larger margin in the count): the surplus slots go to the sentinel tile
        for (int64_t a = at + threadIdx.x; a < end; a += 256) { keys[a] = (OutT)sentinel; ids[a] = gi; }
    }
}

// four consecutive keys per thread (one 16-byte load); the neighbours across thread boundaries come from the adjacent lanes, across
// wave boundaries from memory.  (One key per thread with three 4-byte loads ran at 1.5 TB/s: 14 us at C3, 150 us at C5.)
template <typename KeyT>
__global__ __launch_bounds__(256) void ranges_kernel(const KeyT* __restrict__ keys, int64_t D,
                                                     uint2* __restrict__ ranges, int num_tiles,
                                                     const uint32_t* __restrict__ D_dev, uint32_t base) {
    if (D_dev) D = (int64_t)*D_dev;
    // (capped grid: a launch sized for a worst-case capacity whose device count is small costs its workgroups' exits otherwise —
    //  53 k workgroups, 16 us, for slab B's 10 k instances at BASELINE C5)
    for (int64_t blk = blockIdx.x; blk * (4 * (int64_t)blockDim.x) < D; blk += gridDim.x) {
    const int64_t i0 = 4 * (blk * blockDim.x + threadIdx.x);
    const int lane = threadIdx.x & 63;
    const bool full = i0 + 3 < D;
    uint32_t k[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (full && sizeof(KeyT) == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(keys + i0);
        k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    } else if (full) {                 // four 16-bit keys in one 8-byte load
        const uint2 v = *reinterpret_cast<const uint2*>(keys + i0);
        k[0] = v.x & 0xFFFFu; k[1] = v.x >> 16; k[2] = v.y & 0xFFFFu; k[3] = v.y >> 16;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j < D) k[j] = (uint32_t)keys[i0 + j];
    }
    // key in front of k[0] / behind k[3] (every lane takes part in the shuffles)
    uint32_t prev = (uint32_t)__shfl_up((int)k[3], 1), next = (uint32_t)__shfl_down((int)k[0], 1);
    if (i0 >= D) continue;
    if (lane == 0) prev = i0 > 0 ? (uint32_t)keys[i0 - 1] : 0xFFFFFFFFu;
    if (lane == 63) next = i0 + 4 < D ? (uint32_t)keys[i0 + 4] : 0xFFFFFFFFu;
    const uint32_t nt = (uint32_t)num_tiles;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i >= D) break;
        const uint32_t t = k[j];
        if (t >= nt) continue;
        const uint32_t before = j == 0 ? prev : k[j - 1];
        const uint32_t after = (j == 3 || i + 1 >= D) ? (j == 3 ? next : 0xFFFFFFFFu) : k[j + 1];
        if (i == 0 || before != t) ranges[t].x = base + (uint32_t)i;           // base: where this key array's ids start in `ids`
        if (i == D - 1 || after != t) ranges[t].y = base + (uint32_t)(i + 1);
    }
    }
}

// ---------------------------------------------------------------------------------------------
// depth-slab binning: split of the depth order and the second count
// ---------------------------------------------------------------------------------------------
// One workgroup.  Slab A = the ranks [0, rA) with rA the smallest rank whose offset reaches thr = max(1, fraction * D):
// DA = offs[rA] <= fraction * D + one Gaussian's instances (<= number of tiles).  A 256-ary search over offs[0 .. V) (offs[V] := D).
// Also clears what the forward blend of slab A fills: the open-tile bitmap and the list's counter.
__global__ __launch_bounds__(256) void slab_split_kernel(int P, char* __restrict__ geom, int64_t D_host,
                                                         const uint32_t* __restrict__ D_dev, float fraction,
                                                         uint32_t* __restrict__ open_bits, int n_bit_words) {
    __shared__ uint32_t s_lo, s_hi;
    const GeomLayout L(P);
    const uint32_t* offs = reinterpret_cast<const uint32_t*>(geom + L.offs);
    SlabHeader* hdr = reinterpret_cast<SlabHeader*>(geom + L.slab_hdr);
    const uint32_t V = *reinterpret_cast<const uint32_t*>(geom + L.nvalid);
    const uint64_t D = D_dev ? (uint64_t)*D_dev : (uint64_t)D_host;
    for (int k = threadIdx.x; k < n_bit_words; k += 256) open_bits[k] = 0u;
    uint64_t thr = (uint64_t)((double)fraction * (double)D);
    thr = thr < 1 ? 1 : (thr > D ? D : thr);
    // invariant: value(lo) < thr <= value(hi), value(r) = r < V ? offs[r] : D, value(-1) = -inf; start lo = -1 (stored + 1), hi = V
    if (threadIdx.x == 0) { s_lo = 0u; s_hi = V + 1u; }       // both stored + 1
    __syncthreads();
    for (int round = 0; round < 5; ++round) {
        const uint32_t lo = s_lo, hi = s_hi;                  // candidates strictly between: lo < c < hi (in + 1 coordinates)
        __syncthreads();
        if (hi - lo <= 1u) break;
        const uint64_t span = (uint64_t)(hi - lo - 1u);       // number of interior candidates
        // thread t probes candidate c_t = lo + 1 + floor(t * span / 256) (distinct while span >= 256, repeated below)
        const uint32_t c = lo + 1u + (uint32_t)(((uint64_t)threadIdx.x * span) >> 8);
        const uint32_t r = c - 1u;                            // rank
        const uint64_t v = r < V ? (uint64_t)offs[r] : D;
        // offs is non-decreasing: the largest probed candidate with value < thr raises lo, the smallest with value >= thr lowers hi
        if (v < thr) atomicMax(&s_lo, c); else atomicMin(&s_hi, c);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (D == 0: thr = 0 is clamped to 1 > value(V) = 0 -> lo rises to V + 1, hi stays: rA = V, DA = 0)
        const uint32_t rA = min(s_hi - 1u, V);
        hdr->rA = rA;
        hdr->DA = rA < V ? offs[rA] : (uint32_t)D;
        hdr->DB = 0u;
        hdr->active = 1u;
        hdr->n_open = 0u;
        hdr->pad0 = 0u;               // (overflow flag: a speculative stage 2 on buffers the view outgrew may have raised it)
        hdr->total_b = 0ull;
    }
}

// Slab B's count: per depth rank, the instances that fall into OPEN tiles — the same per-row extents and margin as the count in
// preprocess_kernel (hence >= what the emit will write there), the same occlusion predicate as recount / emit.  One thread per
// rank.  Normally a handful of tiles are open and the kernel exists to PROVE that two or three million ranks have nothing in
// them: the open-tile bitmap sits in LDS (4 KB at 4K; read per tile row of every rect from global memory, 64 lanes on up to 32
// different lines, the loads were what the kernel waited for: 118 us at BASELINE C5) and in front of it a 64-bit mask of the
// 8 x 8 coarse screen cells that hold an open tile at all — most rects are answered by four shifts and an AND.
constexpr int SLAB_LDS_WORDS = 4096;              // bitmap words held in LDS (131 072 tiles); larger grids read global memory
constexpr int SLAB_FEW_OPEN = 64;                 // up to this many open tiles a Gaussian tests the tiles themselves
constexpr int SLAB_RECOUNT_GRID = 1024;           // persistent workgroups (four per CU): the bitmap is staged once per workgroup,
                                                  // not once per 256 ranks (19 531 workgroups at 5 M Gaussians: their prologues —
                                                  // dependent loads, 4 KB of bitmap, three barriers — were most of the kernel)
__global__ __launch_bounds__(256) void slab_recount_kernel(ViewParams vp, int P, const char* __restrict__ geom,
                                                           const uint32_t* __restrict__ open_bits,
                                                           const uint32_t* __restrict__ open_list, int64_t D_host,
                                                           const uint32_t* __restrict__ D_dev, uint32_t* __restrict__ cnt_b) {
    __shared__ OccTable T;
    __shared__ uint32_t s_open[SLAB_LDS_WORDS];
    __shared__ uint32_t s_cells[2];
    __shared__ uint32_t s_list[SLAB_FEW_OPEN];
    const GeomLayout L(P);
    const uint32_t* order = reinterpret_cast<const uint32_t*>(geom + L.order);
    const uint32_t* offs = reinterpret_cast<const uint32_t*>(geom + L.offs);
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + L.binrec);
    const SlabHeader* hdr = reinterpret_cast<const SlabHeader*>(geom + L.slab_hdr);
    const OccHeader* occ = reinterpret_cast<const OccHeader*>(geom + L.occ_hdr);
    const int V = (int)*reinterpret_cast<const uint32_t*>(geom + L.nvalid);
    if ((int)(blockIdx.x * blockDim.x) >= V) return;
    const uint32_t n_open = hdr->n_open;                                       // final: blend A has completed
    if (n_open == 0u) {                                                        // nothing left open: slab B is empty
        for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < V; r += gridDim.x * blockDim.x) cnt_b[r] = 0u;
        return;
    }
    const int n_tiles = vp.gx * vp.gy;
    const int n_words = (n_tiles + 31) / 32;
    const bool in_lds = n_words <= SLAB_LDS_WORDS;                             // workgroup-uniform
    // coarse cells: 2^cs x 2^ct tiles each, at most 8 x 8 of them (ViewParams: the same shifts as preprocess_kernel's ranges)
    const bool ranged = vp.cell_sx >= 0 && hdr->pad[0] == 1u;       // (pad[0]: the scan of THIS forward delivered the ranges)
    const int cs = vp.cell_sx, ct = vp.cell_sy;
    const bool use_cells = ranged && in_lds && n_open <= 1024u;                // (many open tiles: every cell is set anyway)
    if (threadIdx.x < 2) s_cells[threadIdx.x] = use_cells ? 0u : 0xFFFFFFFFu;
    // a handful of open tiles (the rule): a Gaussian that passes the cell test walks THEM, not the rows of its rect — the
    // multi-scale levels' giants have rects of a hundred rows, one lane of their wave walked them alone (80 us at BASELINE C5)
    const bool few = n_open <= (uint32_t)SLAB_FEW_OPEN && open_list != nullptr;
    if (few && threadIdx.x < n_open) { const uint32_t t = open_list[threadIdx.x]; s_list[threadIdx.x] = (t % (uint32_t)vp.gx) | ((t / (uint32_t)vp.gx) << 16); }
    __syncthreads();
    if (in_lds) {
        for (int w = threadIdx.x; w < n_words; w += blockDim.x) {
            uint32_t m = open_bits[w];
            if (w == n_words - 1 && (n_tiles & 31)) m &= 0xFFFFFFFFu >> (32 - (n_tiles & 31));
            s_open[w] = m;
            while (use_cells && m) {
                const int t = 32 * w + __ffs((int)m) - 1;
                m &= m - 1u;
                const int cell = ((t / vp.gx) >> ct) * 8 + ((t % vp.gx) >> cs);
                atomicOr(&s_cells[cell >> 5], 1u << (cell & 31));
            }
        }
    }
    const bool occ_on = occ->enabled != 0u && occ->any_closed != 0u;
    const int occ_lb = (int)occ->block_log2, occ_nbx = (int)occ->nbx;
    uint32_t cut_min = 0xFFFFFFFFu;
    if (occ_on) {
        occ_table_load(T, reinterpret_cast<const uint32_t*>(geom + L.occ_cut), occ_nbx, (int)occ->nby);
        cut_min = T.cut_min;
    }
    __syncthreads();
    const uint64_t cells = (uint64_t)s_cells[0] | ((uint64_t)s_cells[1] << 32);
    auto open_count = [&](uint32_t a, uint32_t b) { return in_lds ? open_in_range(s_open, a, b) : open_in_range(open_bits, a, b); };
    const int64_t D = D_dev ? (int64_t)*D_dev : D_host;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < V; r += gridDim.x * blockDim.x) {
        const uint32_t full = (uint32_t)((r + 1 < V ? (int64_t)offs[r + 1] : D) - (int64_t)offs[r]);
        uint32_t c = 0;
        // the coarse cells this rank's rect touches came with the scan (depth order, in cnt_b itself): only a Gaussian that
        // touches a cell with an open tile fetches its record — a random 32-byte gather per rank otherwise
        if (full && (!ranged || (cell_range_mask(cnt_b[r]) & cells))) {
            const uint32_t gi = order[r];
            const float4 q1 = binrec[gi].q1, q0 = binrec[gi].q0;         // (one 32-byte line)
            const uint32_t rcx = __float_as_uint(q1.z), rcy = __float_as_uint(q1.w);
            const int minx = rcx & 0xFFFF, miny = rcx >> 16, maxx = rcy & 0xFFFF, maxy = rcy >> 16;
            const uint32_t kmine = cut_min != 0xFFFFFFFFu ? occ_bucket(reinterpret_cast<const uint32_t*>(geom + L.skey)[r]) : 0u;
            const bool cut_check = kmine > cut_min;
            const float tau2 = q1.y;
            const bool test = tau2 > -1.0e38f;
            LevelSetRows ls{};
            bool ls_ready = false;
            if (few) {
                for (uint32_t k = 0; k < n_open; ++k) {
                    const int tx = (int)(s_list[k] & 0xFFFFu), ty = (int)(s_list[k] >> 16);
                    if (tx < minx || tx >= maxx || ty < miny || ty >= maxy) continue;
                    if (cut_check && T.cut[(ty >> occ_lb) * occ_nbx + (tx >> occ_lb)] < kmine) continue;
                    if (test && !ls_ready) { ls = levelset_rows_setup(q0.z, q0.w, q1.x, tau2); ls_ready = true; }
                    int tlo = minx, thi = maxx - 1;
                    if (test && !levelset_row_interval(ls, q0.x, q0.y, ty, minx, maxx, LEVELSET_MARGIN_COUNT, tlo, thi)) continue;
                    c += (tx >= tlo && tx <= thi) ? 1u : 0u;
                }
            } else
            for (int ty = miny; ty < maxy; ++ty) {
                if (cut_check && T.rowmax[ty >> occ_lb] < kmine) continue;
                const uint32_t row = (uint32_t)(ty * vp.gx);
                // (no open tile in this row of the rect -> no extent to compute)
                if (open_count(row + (uint32_t)minx, row + (uint32_t)(maxx - 1)) == 0u) continue;
                if (test && !ls_ready) { ls = levelset_rows_setup(q0.z, q0.w, q1.x, tau2); ls_ready = true; }
                int tlo = minx, thi = maxx - 1;
                if (test && !levelset_row_interval(ls, q0.x, q0.y, ty, minx, maxx, LEVELSET_MARGIN_COUNT, tlo, thi)) continue;
                if (tlo > thi) continue;
                if (!cut_check) {
                    c += open_count(row + (uint32_t)tlo, row + (uint32_t)thi);
                } else {
                    const int brow = (ty >> occ_lb) * occ_nbx;
                    for (int tx = tlo; tx <= thi;) {
                        const int bend = min(thi, (((tx >> occ_lb) + 1) << occ_lb) - 1);
                        if (T.cut[brow + (tx >> occ_lb)] >= kmine) c += open_count(row + (uint32_t)tx, row + (uint32_t)bend);
                        tx = bend + 1;
                    }
                }
            }
            c = min(c, full);    // (a subset of the first count's tiles; the clamp only guards the scan against a rounding surprise)
        }
        cnt_b[r] = c;
    }
}

}  // namespace

hipError_t launch_emit(const ViewParams& vp, int P, const char* geom, uint32_t* keys, uint32_t* ids, int64_t D,
                       hipStream_t s, ZeroJob zj, const uint32_t* D_dev, bool keys16, uint32_t* heavy_q, int slab,
                       const uint32_t* open_bits, int64_t density_D) {
    if (P == 0 || D == 0) return hipSuccess;     // (callers fold a ZeroJob in only when D > 0)
    const bool narrow = vp.gx * vp.gy < 65535;        // tile ids and the sentinel (= number of tiles) fit 16 bits
    const dim3 grid((P + 255) / 256), block(256);
    if (keys16 && !narrow) return hipErrorInvalidValue;
    if (slab == 2 && !open_bits) return hipErrorInvalidValue;
    const GeomLayout L(P);
    const SlabHeader* hdr = reinterpret_cast<const SlabHeader*>(geom + L.slab_hdr);
    SlabArgs sl{nullptr, nullptr, nullptr};
    if (slab == 1) sl.V_dev = &hdr->rA;
    if (slab == 2) { sl.offs = reinterpret_cast<const uint32_t*>(geom + L.offs_b); sl.open_bits = open_bits; }
    // a block of 256 Gaussians emits more than the small stage on average (slab A: the ranks it covers are as dense as the whole
    // view's; slab B: normally a handful of tiles — the small stage keeps more of its workgroups, most of which only learn that
    // they have nothing to emit, resident per CU: 64 -> ~15 us on an empty slab B at C5)
    const bool wide = slab == 2 ? false : (density_D > 0 ? density_D : D) > 8 * (int64_t)P;
#define MSGS_EMIT(STAGE, KEYT, OUTT, KPTR)                                                                                          \
    do {                                                                                                                            \
        if (slab == 2) hipLaunchKernelGGL((emit_kernel<STAGE, KEYT, OUTT, true>), grid, block, 0, s, vp, P, geom, KPTR, ids, D, zj,  \
                                          D_dev, heavy_q, sl);                                                                      \
        else hipLaunchKernelGGL((emit_kernel<STAGE, KEYT, OUTT, false>), grid, block, 0, s, vp, P, geom, KPTR, ids, D, zj, D_dev,    \
                                heavy_q, sl);                                                                                       \
    } while (0)
    uint16_t* k16 = reinterpret_cast<uint16_t*>(keys);
    if (keys16) {
        if (wide) MSGS_EMIT(12288, uint16_t, uint16_t, k16); else MSGS_EMIT(3072, uint16_t, uint16_t, k16);
    } else if (wide) {
        if (narrow) MSGS_EMIT(12288, uint16_t, uint32_t, keys); else MSGS_EMIT(6144, uint32_t, uint32_t, keys);
    } else {
        if (narrow) MSGS_EMIT(3072, uint16_t, uint32_t, keys); else MSGS_EMIT(3072, uint32_t, uint32_t, keys);
    }
#undef MSGS_EMIT
    if (heavy_q) {
        // at most D / EMIT_HEAVY_MIN Gaussians can be queued; workgroups beyond the queue's length leave at once
        const unsigned hb = (unsigned)std::min<int64_t>(2048, D / EMIT_HEAVY_MIN + 1);
        if (keys16 && slab == 2)
            hipLaunchKernelGGL((emit_heavy_kernel<uint16_t, true>), dim3(hb), block, 0, s, vp, P, geom, k16, ids, D, D_dev,
                               (const uint32_t*)heavy_q, sl);
        else if (keys16)
            hipLaunchKernelGGL((emit_heavy_kernel<uint16_t, false>), dim3(hb), block, 0, s, vp, P, geom, k16, ids, D, D_dev,
                               (const uint32_t*)heavy_q, sl);
        else if (slab == 2)
            hipLaunchKernelGGL((emit_heavy_kernel<uint32_t, true>), dim3(hb), block, 0, s, vp, P, geom, keys, ids, D, D_dev,
                               (const uint32_t*)heavy_q, sl);
        else
            hipLaunchKernelGGL((emit_heavy_kernel<uint32_t, false>), dim3(hb), block, 0, s, vp, P, geom, keys, ids, D, D_dev,
                               (const uint32_t*)heavy_q, sl);
    }
    return hipGetLastError();
}

hipError_t launch_slab_split(int P, char* geom, int64_t D, const uint32_t* D_dev, float fraction, uint32_t* open_bits,
                             int num_tiles, hipStream_t s) {
    hipLaunchKernelGGL(slab_split_kernel, dim3(1), dim3(256), 0, s, P, geom, D, D_dev, fraction, open_bits, (num_tiles + 31) / 32 + 1);
    return hipGetLastError();
}

hipError_t launch_slab_recount(const ViewParams& vp, int P, char* geom, const uint32_t* open_bits, const uint32_t* open_list, int64_t D,
                               const uint32_t* D_dev, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const GeomLayout L(P);
    hipLaunchKernelGGL(slab_recount_kernel, dim3(std::min((P + 255) / 256, SLAB_RECOUNT_GRID)), dim3(256), 0, s, vp, P, (const char*)geom, open_bits, open_list, D, D_dev,
                       reinterpret_cast<uint32_t*>(geom + L.offs_b));
    return hipGetLastError();
}

hipError_t launch_ranges(const uint32_t* keys, int64_t D, uint2* ranges, int num_tiles, hipStream_t s,
                         bool pre_zeroed, const uint32_t* D_dev, bool keys16, uint32_t base) {
    if (!pre_zeroed) {
        hipError_t e = launch_zero(ranges, sizeof(uint2) * (size_t)num_tiles, s);
        if (e != hipSuccess) return e;
    }
    if (D == 0) return hipSuccess;
    const dim3 grid((unsigned)std::min<int64_t>((D + 1023) / 1024, 8192)), block(256);
    if (keys16)
        hipLaunchKernelGGL(ranges_kernel<uint16_t>, grid, block, 0, s, reinterpret_cast<const uint16_t*>(keys), D, ranges, num_tiles,
                           D_dev, base);
    else
        hipLaunchKernelGGL(ranges_kernel<uint32_t>, grid, block, 0, s, keys, D, ranges, num_tiles, D_dev, base);
    return hipGetLastError();
}

}  // namespace msgs
