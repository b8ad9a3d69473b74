// occlusion.hip — exact per-tile occlusion cut-off in front of the depth sort and the emit (round 5).
//
// The reference emits, sorts and walks every (tile, Gaussian) instance of a view; a pixel stops walking its tile's list at the
// first entry where T (1 - alpha) < 1e-4.  When a multi-scale MS-GS model is rendered WITHOUT its
// pixel-size filters (/root/reference/render.py:32 and the evaluation loop /root/reference/train.py:488-496 call render() with
// the default flags) the coarse-level Gaussians — scaled x4 .. x64 — are all drawn at level 0: 427 M instances at 1080p for
// 1 M Gaussians, of which a few million are ever walked, because a handful of opaque giants in front terminates every pixel.
//
// What this pass proves and uses.  Let a Gaussian's alpha >= 1/255 level set contain a whole block of tiles.  alpha is a
// concave function of the pixel offset in the log domain, so its minimum over the block's pixel centres is at one of the four
// corner pixels: alpha_min.  EVERY pixel of the block then blends the Gaussian with alpha >= alpha_min (it is not skipped: the
// skip rule is alpha < 1/255), i.e. its transmittance behind that entry is at most (1 - alpha_min) times the one in front.
// With the covers of a block taken front to back, the first depth at which  prod (1 - alpha_min) < 0.5e-4  (the rule's 1e-4
// with a factor 2 of slack for float32 rounding on either side) is a depth behind which NO pixel of the block evaluates
// anything: every one of them has met the termination test at or before that entry.  Instances behind it are dropped from the
// tile counts (here, before the depth sort and the scan) and from the emit (binning.hip).  The lists every pixel actually walks
// are unchanged, entry for entry: image, n_contrib, final_T and every gradient are bit-identical to the uncut path
// (tests/test_occlusion_gpu.py).  Any SUBSET of the covers gives a valid (later) cut-off, so the pass may ignore what it
// likes: it only looks at Gaussians with more than OCC_HEAVY_MIN tile instances, only at blocks they cover completely, and it
// accumulates per depth BUCKET (1/16 octave of view depth) instead of per rank — the cut-off is the far end of the bucket in
// which the product crosses.  Integer (fixed-point) sums: the result does not depend on the order the candidates arrive in.
//
// Four launches between preprocess_kernel and the depth sort:
//   occ_hist_kernel      depth histogram of the candidates preprocess_kernel left behind, per wave slot
//   occ_gather_kernel    keeps the nearest OCC_MAX_CAND of them (whole depth buckets) and gathers their records
//   occ_cover_kernel     one workgroup per block of tiles: bucketed sums of -log2(1 - alpha_min), prefix, cut-off key per tile
//   occ_recount_kernel   Gaussians behind the nearest cut-off recount their tile instances (index order, tiles[] / key[] in
//                        place; a Gaussian left without instances leaves the depth sort: key 0xFFFFFFFF)
// On a view where nothing closes (the BASELINE C3 headline: four candidates) they cost 16-21 us; the Python wrapper then skips the
// pass for that kind of view and probes again every 32nd call (msgs_view_t.skip_occlusion, msgs_forward_info).
#include "msgs_internal.h"

#include <atomic>
#include <cstdlib>

namespace msgs {

namespace {

std::atomic<int> g_occlusion{[] { const char* e = getenv("MSGS_NO_OCCLUSION"); return (e && e[0] == '1') ? 0 : 1; }()};

constexpr float OCC_LOG2_T = 13.287712f + 1.0f;        // -log2(1e-4) + one bit of slack (factor 2 on the product)
constexpr float OCC_FIX = 2048.0f;                     // fixed-point scale of the bucket sums (2^-11 bits)
constexpr uint32_t OCC_THRESHOLD = (uint32_t)(OCC_LOG2_T * OCC_FIX) + 1u;
// (sums stay below 2^32: a cover adds at most -log2(0.01) * 2048 = 13 607, and a view has fewer than 2^31 / 13 607 candidates
//  per bucket in any scene this library accepts — P < 2^31 — while the prefix saturates below)

// Cover candidates.  preprocess_kernel left, per wave slot, {id, depth key} of its heavy Gaussians and their number.  Only the
// NEAREST candidates matter (the product crosses within the first few dozen covers of a block), so at most OCC_MAX_CAND are kept,
// chosen by depth:
//   occ_hist_kernel    every workgroup (256 slots, one per thread) counts its candidates per depth bucket in LDS and adds the
//                      non-empty buckets and its total to the view's histogram / counter (cleared by preprocess_kernel)
//   occ_gather_kernel  every workgroup prefix-sums the histogram itself, finds the deepest bucket kappa up to which the candidates
//                      still fit, keeps its candidates with bucket <= kappa (positions: one atomic per workgroup — the ORDER of the
//                      records is irrelevant, the cover sums are integers; the SET is deterministic) and gathers their records.
//                      When the nearest non-empty bucket alone holds more than fit (thousands of covers at one depth), every
//                      stride-th candidate in index order is kept instead: positions by formula from all slot counts.
constexpr int OCC_GATHER_SLOTS = 256;       // wave slots per workgroup (one per thread)

__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t v, uint32_t* s_w, uint32_t* block_total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < wv; ++k) base += s_w[k];
    *block_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void occ_hist_kernel(const uint32_t* __restrict__ heavy_list,
                                                       const uint32_t* __restrict__ heavy_count, int n_slots,
                                                       OccHeader* __restrict__ hdr, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[OCC_BUCKETS];
    __shared__ uint32_t s_w[4];
    const int slot = blockIdx.x * OCC_GATHER_SLOTS + threadIdx.x;
    const uint32_t cnt = slot < n_slots ? heavy_count[slot] : 0u;
    uint32_t total;
    block_exclusive_256(cnt, s_w, &total);
    if (total == 0) return;                                             // (block-uniform)
    for (int k = threadIdx.x; k < OCC_BUCKETS; k += 256) s_h[k] = 0u;
    __syncthreads();
    const uint2* e = reinterpret_cast<const uint2*>(heavy_list) + (size_t)slot * 64;
    for (uint32_t j = 0; j < cnt; ++j) atomicAdd(&s_h[occ_bucket(e[j].y)], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < OCC_BUCKETS; k += 256)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
    if (threadIdx.x == 0) atomicAdd(&hdr->n_heavy, total);
}

__global__ __launch_bounds__(256) void occ_gather_kernel(int P, const char* __restrict__ geom,
                                                         const uint32_t* __restrict__ heavy_list,
                                                         const uint32_t* __restrict__ heavy_count, int n_slots,
                                                         OccHeader* __restrict__ hdr, const uint32_t* __restrict__ hist,
                                                         OccCand* __restrict__ cand,
                                                         uint32_t block_log2, uint32_t nbx, uint32_t nby) {
    __shared__ uint16_t s_kept[OCC_GATHER_SLOTS * 64];     // (local slot << 6) | index in the slot
    __shared__ uint32_t s_prefix[OCC_GATHER_SLOTS];
    __shared__ uint32_t s_red[2][4], s_scan[4], s_kappa, s_keep, s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot0 = blockIdx.x * OCC_GATHER_SLOTS;
    const uint32_t total = hdr->n_heavy;                    // final: occ_hist_kernel has completed
    // deepest bucket kappa with (candidates in buckets <= kappa) <= OCC_MAX_CAND; thread t owns 8 consecutive buckets
    constexpr int PER = OCC_BUCKETS / 256;
    uint32_t hv[PER], hsum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { hv[k] = hist[threadIdx.x * PER + k]; hsum += hv[k]; }
    if (threadIdx.x == 0) { s_kappa = 0xFFFFFFFFu; s_keep = 0u; }
    uint32_t dummy;
    uint32_t run = block_exclusive_256(hsum, s_scan, &dummy);
    {
        uint32_t best = 0xFFFFFFFFu, keep = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            run += hv[k];
            if (hv[k] && run <= (uint32_t)OCC_MAX_CAND) { best = (uint32_t)(threadIdx.x * PER + k); keep = run; }
        }
        if (best != 0xFFFFFFFFu) { atomicMax(&s_keep, keep); }          // cumulative counts grow with the bucket index: the
        __syncthreads();                                                //   largest admissible cumulative count marks kappa
        if (best != 0xFFFFFFFFu && keep == s_keep) s_kappa = best;
        __syncthreads();
    }
    const uint32_t kappa = s_kappa, keep_total = s_keep;
    const bool by_depth = total <= (uint32_t)OCC_MAX_CAND || kappa != 0xFFFFFFFFu;
    const uint32_t limit = total <= (uint32_t)OCC_MAX_CAND ? (uint32_t)(OCC_BUCKETS - 1) : kappa;
    uint32_t stride = 1u, before = 0u;
    if (!by_depth) {
        // the nearest bucket alone is too full: every stride-th candidate in index order; the candidates in front of this
        // workgroup's slots from ALL slot counts (62 KB at 1 M Gaussians, four counts per load)
        stride = (total + OCC_MAX_CAND - 1) / OCC_MAX_CAND;
        const int n4 = n_slots >> 2;                        // n_slots is a multiple of 4
        const uint4* c4 = reinterpret_cast<const uint4*>(heavy_count);
        for (int q = threadIdx.x; q < n4 && 4 * q < slot0; q += 256) {
            const uint4 v = c4[q];
            before += v.x + v.y + v.z + v.w;
        }
        for (int off = 32; off > 0; off >>= 1) before += (uint32_t)__shfl_xor((int)before, off);
        if (lane == 0) s_red[1][wv] = before;
        __syncthreads();
        before = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // (the header was zeroed by preprocess_kernel)
        hdr->n_cand = by_depth ? (total <= (uint32_t)OCC_MAX_CAND ? total : keep_total) : (total + stride - 1) / stride;
        hdr->depth_limit = by_depth ? limit : 0xFFFFFFFFu;
        hdr->enabled = 1u;
        hdr->block_log2 = block_log2;
        hdr->nbx = nbx;
        hdr->nby = nby;
    }
    if (total == 0) return;
    const int slot = slot0 + threadIdx.x;
    const uint32_t cnt = slot < n_slots ? heavy_count[slot] : 0u;
    const uint2* ent = reinterpret_cast<const uint2*>(heavy_list);
    uint32_t kept = 0, first = 0, g0 = 0;
    if (by_depth) {
        for (uint32_t j = 0; j < cnt; ++j) kept += occ_bucket(ent[(size_t)slot * 64 + j].y) <= limit ? 1u : 0u;
    } else {
        // position (index order) of this slot's first candidate; positions g with g % stride == 0 are kept
        first = before + block_exclusive_256(cnt, s_scan, &dummy);
        g0 = ((first + stride - 1) / stride) * stride;
        kept = g0 < first + cnt ? (first + cnt - 1 - g0) / stride + 1 : 0u;
    }
    s_prefix[threadIdx.x] = first;
    uint32_t kept_total;
    uint32_t at = block_exclusive_256(kept, s_scan, &kept_total);       // <= 256 * 64 entries
    if (by_depth) {
        for (uint32_t j = 0; j < cnt; ++j)
            if (occ_bucket(ent[(size_t)slot * 64 + j].y) <= limit) s_kept[at++] = (uint16_t)((threadIdx.x << 6) | j);
        if (threadIdx.x == 0) s_base = kept_total ? atomicAdd(&hdr->n_written, kept_total) : 0u;
    } else {
        for (uint32_t g = g0; g < first + cnt; g += stride) s_kept[at++] = (uint16_t)((threadIdx.x << 6) | (g - first));
    }
    __syncthreads();
    const GeomLayout L(P);
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + L.binrec);
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom + L.rec);
    for (uint32_t e = threadIdx.x; e < kept_total; e += 256) {
        const uint32_t code = s_kept[e];
        const uint32_t ls = code >> 6, j = code & 63u;
        const uint2 ge = ent[(size_t)(slot0 + ls) * 64 + j];
        const BinRec b = binrec[ge.x];
        OccCand c;
        c.c0 = b.q0;                                                                  // px, py, kA, kB(half)
        c.c1 = make_float4(b.q1.x, rec[ge.x].r1.y, __uint_as_float(ge.y), __uint_as_float(ge.x));   // kC, log2 o, depth key, id
        c.rect_lo = __float_as_uint(b.q1.z);                                          // minx | miny << 16   (tiles)
        c.rect_hi = __float_as_uint(b.q1.w);                                          // maxx | maxy << 16   (exclusive)
        c.pad0 = c.pad1 = 0u;
        cand[by_depth ? s_base + e : (s_prefix[ls] + j) / stride] = c;
    }
}

// fixed-point weight -log2(1 - alpha_min) of candidate c over the pixel-centre rectangle [x0, x1] x [y0, y1] of the block of
// tiles [tx0, tx1) x [ty0, ty1); 0 when the level set does not contain the rectangle — or when the block is not inside the
// candidate's tile RECT: count, emit and recount clip every Gaussian to its rect (radius = ceil(3 sqrt(lambda_max)), Q4/Q5),
// while the alpha >= 1/255 level set of an opaque Gaussian reaches up to 3.33 sigma along the major axis.  A tile outside the
// rect holds no instance of the Gaussian, so none of its pixels loses transmittance to it (a giant centred off-screen whose
// rect ends inside a block would otherwise add a phantom weight to tiles it is never blended in).
__device__ __forceinline__ uint32_t cover_weight(const OccCand& c, float x0, float x1, float y0, float y1, int tx0, int tx1,
                                                 int ty0, int ty1) {
#pragma clang fp contract(off)
    {
        const int minx = (int)(c.rect_lo & 0xFFFFu), miny = (int)(c.rect_lo >> 16);
        const int maxx = (int)(c.rect_hi & 0xFFFFu), maxy = (int)(c.rect_hi >> 16);
        if (tx0 < minx || tx1 > maxx || ty0 < miny || ty1 > maxy) return 0u;
    }
    const float px = c.c0.x, py = c.c0.y, A = c.c0.z, Bh = c.c0.w, Cc = c.c1.x, l2o = c.c1.y;
    const float dxa = px - x0, dxb = px - x1, dya = py - y0, dyb = py - y1;
    // f(d) = A dx^2 + 2 Bh dx dy + C dy^2 = log2 G(d) <= 0, concave: its minimum over the rectangle is at a corner
    const float fa = A * dxa * dxa, fb = A * dxb * dxb, ga = Cc * dya * dya, gb = Cc * dyb * dyb;
    const float f00 = fa + ga + 2.0f * Bh * dxa * dya, f01 = fa + gb + 2.0f * Bh * dxa * dyb;
    const float f10 = fb + ga + 2.0f * Bh * dxb * dya, f11 = fb + gb + 2.0f * Bh * dxb * dyb;
    const float fmin = fminf(fminf(f00, f01), fminf(f10, f11));
    // log2 of the smallest alpha any pixel of the rectangle sees, pushed DOWN by more than the kernels' float32 evaluation of
    // the same quantity can differ: 4e-3 absolute (0.3 % on alpha) + 1e-6 of the magnitude of the terms that cancel in it
    // (an elongated, rotated footprint far from its centre: the three monomials are large and of mixed sign)
    const float mag = fmaxf(fabsf(fa), fabsf(fb)) + fmaxf(fabsf(ga), fabsf(gb)) +
                      2.0f * fabsf(Bh) * fmaxf(fabsf(dxa), fabsf(dxb)) * fmaxf(fabsf(dya), fabsf(dyb));
    const float la = fmin + l2o - (1e-6f * mag + 4e-3f);
    if (!(la >= -7.99f)) return 0u;                     // alpha_min must clear 1/255 (log2 = -7.9944) or a pixel may SKIP it
    const float amin = fminf(0.99f, exp2f(la));
    const float w = -log2f(1.0f - amin);                // >= 0.0057
    return (uint32_t)(w * (OCC_FIX * 0.999f));          // rounded down
}

// one workgroup per block of B x B tiles
constexpr int OCC_COVER_THREADS = 256;
__global__ __launch_bounds__(OCC_COVER_THREADS) void occ_cover_kernel(ViewParams vp, int B, int nbx, OccHeader* __restrict__ hdr,
                                                                      const OccCand* __restrict__ cand,
                                                                      uint32_t* __restrict__ occ_cut) {
    __shared__ uint32_t s_b[OCC_BUCKETS];
    __shared__ uint32_t s_wave[OCC_COVER_THREADS / 64];
    __shared__ uint32_t s_cross;
    const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const int tx0 = bx * B, ty0 = by * B;
    const int tx1 = min(tx0 + B, vp.gx), ty1 = min(ty0 + B, vp.gy);
    const uint32_t n = hdr->n_cand;
    uint32_t cut = 0xFFFFu;                             // depth bucket behind which the block is dead (0xFFFF: open)
    bool closed = false;
    if (n >= 3) {                                       // (a cover weighs at most 6.65 bits: fewer than three cannot close anything)
        for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_COVER_THREADS) s_b[k] = 0u;
        if (threadIdx.x == 0) s_cross = 0xFFFFFFFFu;
        __syncthreads();
        const float x0 = (float)(tx0 * TILE), y0 = (float)(ty0 * TILE);
        const float x1 = (float)(min(tx1 * TILE, vp.W) - 1), y1 = (float)(min(ty1 * TILE, vp.H) - 1);   // pixels inside the image
        auto add = [&](const OccCand& cc) {
            const uint32_t w = cover_weight(cc, x0, x1, y0, y1, tx0, tx1, ty0, ty1);
            if (w) atomicAdd(&s_b[occ_bucket(__float_as_uint(cc.c1.z))], w);
        };
        uint32_t c = threadIdx.x;
        for (; c + OCC_COVER_THREADS < n; c += 2 * OCC_COVER_THREADS) {      // two records in flight per thread
            const OccCand ca = cand[c], cb = cand[c + OCC_COVER_THREADS];
            add(ca);
            add(cb);
        }
        if (c < n) add(cand[c]);
        __syncthreads();
        // front-to-back prefix over the buckets: thread t owns PER consecutive buckets
        constexpr int PER = OCC_BUCKETS / OCC_COVER_THREADS;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = min(s_b[threadIdx.x * PER + k], 0x00FFFFFFu); sum += v[k]; }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int k = 0; k < wv; ++k) run += s_wave[k];
        uint32_t cross = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            run += v[k];
            if (cross == 0xFFFFFFFFu && run >= OCC_THRESHOLD) cross = (uint32_t)(threadIdx.x * PER + k);
        }
        if (cross != 0xFFFFFFFFu) atomicMin(&s_cross, cross);
        __syncthreads();
        const uint32_t q = s_cross;
        // everything up to and including the crossing bucket stays; the last bucket also holds the keys clamped into it, so a
        // crossing there closes nothing
        closed = q < (uint32_t)(OCC_BUCKETS - 1);
        if (closed) cut = q;
    }
    if (threadIdx.x == 0) {
        occ_cut[blockIdx.x] = cut;
        if (closed) hdr->any_closed = 1u;               // plain store (every writer writes the same value): no atomics here — the
    }                                                   // readers reduce the table themselves (occ_table_load)
}

// Gaussians behind the nearest cut-off count their instances again: the same per-row level-set extents and margin as the
// count in preprocess_kernel, restricted to the tiles whose block's cut-off they are in front of.  Index order (before the
// depth sort).  The block table sits in LDS.  Footprints with a rect of at most OCC_LIGHT_RECT tiles are recounted by their own
// thread — rows of blocks and blocks that are closed at this depth are skipped whole; larger ones by their whole wave: lane <-> tile row for the row extents, then only the
// rows whose row of blocks is still open at this depth, with the lanes on consecutive tiles.
constexpr int OCC_RECOUNT_THREADS = 1024;     // (few, fat workgroups: on a view where nothing closed every one of them only reads
                                              //  the header and leaves)
__global__ __launch_bounds__(OCC_RECOUNT_THREADS) void occ_recount_kernel(ViewParams vp, int P, char* __restrict__ geom) {
    __shared__ OccTable T;
    const GeomLayout L(P);
    OccHeader* hdr = reinterpret_cast<OccHeader*>(geom + L.occ_hdr);
    if (hdr->any_closed == 0u) return;                                  // nothing closed in this view
    const int lb = (int)hdr->block_log2, nbx = (int)hdr->nbx;
    occ_table_load(T, reinterpret_cast<const uint32_t*>(geom + L.occ_cut), nbx, (int)hdr->nby);
    const uint32_t cut_min = T.cut_min, cut_max = T.cut_max;
    uint32_t* tiles = reinterpret_cast<uint32_t*>(geom + L.tiles);
    uint32_t* key = reinterpret_cast<uint32_t*>(geom + L.key);
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + L.binrec);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0, k = 0;
    if (i < P) { cnt = tiles[i]; k = occ_bucket(key[i]); }             // k: this Gaussian's depth bucket
    const bool affected = cnt > 0 && k > cut_min;
    const bool all_behind = affected && k > cut_max;                    // behind the cut-off of EVERY block (none stayed open)
    const bool work = affected && !all_behind;
    uint32_t newcnt = all_behind ? 0u : cnt;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (work) {
        q0 = binrec[i].q0; q1 = binrec[i].q1;
        const uint32_t rcx = __float_as_uint(q1.z), rcy = __float_as_uint(q1.w);
        minx = rcx & 0xFFFF; miny = rcx >> 16; maxx = rcy & 0xFFFF; maxy = rcy >> 16;
    }
    const bool light = work && (maxx - minx) * (maxy - miny) <= OCC_LIGHT_RECT;
    if (light) {
        const float tau2 = q1.y;
        const bool test = tau2 > -1.0e38f;
        const LevelSetRows ls = test ? levelset_rows_setup(q0.z, q0.w, q1.x, tau2) : LevelSetRows{};
        uint32_t c = 0;
        for (int ty = miny; ty < maxy; ++ty) {
            if (T.rowmax[ty >> lb] < k) continue;
            int tlo = minx, thi = maxx - 1;
            if (test && !levelset_row_interval(ls, q0.x, q0.y, ty, minx, maxx, LEVELSET_MARGIN_COUNT, tlo, thi)) continue;
            const int brow = (ty >> lb) * nbx;
            for (int tx = tlo; tx <= thi;) {                             // cover block by cover block
                const int bend = min(thi, (((tx >> lb) + 1) << lb) - 1);
                if (T.cut[brow + (tx >> lb)] >= k) c += (uint32_t)(bend - tx + 1);
                tx = bend + 1;
            }
        }
        newcnt = c;
    }
    uint64_t todo = __ballot(work && !light);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t h_key = lane_bcast(k, src);
        const float gx_ = lane_bcast(q0.x, src), gy_ = lane_bcast(q0.y, src), cA = lane_bcast(q0.z, src), cBh = lane_bcast(q0.w, src);
        const float cC = lane_bcast(q1.x, src), tau2 = lane_bcast(q1.y, src);
        const int h_minx = lane_bcast(minx, src), h_miny = lane_bcast(miny, src), h_maxx = lane_bcast(maxx, src), h_maxy = lane_bcast(maxy, src);
        const bool test = tau2 > -1.0e38f;
        const LevelSetRows ls = test ? levelset_rows_setup(cA, cBh, cC, tau2) : LevelSetRows{};
        uint32_t c = 0;                                                 // wave-uniform
        for (int row0 = h_miny; row0 < h_maxy; row0 += 64) {
            const int ty = row0 + lane;
            int tlo = h_minx, thi = h_maxx - 1;
            bool hit = ty < h_maxy && T.rowmax[min(ty, vp.gy - 1) >> lb] >= h_key;
            if (hit && test) hit = levelset_row_interval(ls, gx_, gy_, ty, h_minx, h_maxx, LEVELSET_MARGIN_COUNT, tlo, thi);
            const int n_row = hit ? thi - tlo + 1 : 0;
            uint64_t rows = __ballot(n_row > 0);                        // only the rows that can still receive something
            while (rows) {
                const int r = __ffsll((long long)rows) - 1;
                rows &= rows - 1;
                const int n_r = lane_bcast(n_row, r), tlo_r = lane_bcast(tlo, r);
                const int brow = ((row0 + r) >> lb) * nbx;
                for (int j0 = 0; j0 < n_r; j0 += 64) {
                    const int j = j0 + lane;
                    const bool keep = j < n_r && T.cut[brow + ((tlo_r + j) >> lb)] >= h_key;
                    c += (uint32_t)__popcll(__ballot(keep));
                }
            }
        }
        if (lane == src) newcnt = c;
    }
    if (affected) {
        tiles[i] = newcnt;
        if (newcnt == 0) key[i] = 0xFFFFFFFFu;                          // leaves the (compacting) depth sort
    }
}

}  // namespace

int set_occlusion(int on) { return g_occlusion.exchange(on ? 1 : 0); }
int get_occlusion() { return g_occlusion.load(); }
int occlusion_block_log2(int gx, int gy) {
    static const int lb_min = [] {
        const char* e = getenv("MSGS_OCC_BLOCK");
        const int v = e ? atoi(e) : 4;
        return v >= 16 ? 4 : v >= 8 ? 3 : v >= 4 ? 2 : v >= 2 ? 1 : 0;
    }();
    int lb = lb_min;
    while ((int64_t)((gx + (1 << lb) - 1) >> lb) * ((gy + (1 << lb) - 1) >> lb) > OCC_MAX_BLOCKS ||
           ((gy + (1 << lb) - 1) >> lb) > OCC_MAX_BLOCK_ROWS)
        ++lb;
    return lb;
}

hipError_t launch_occlusion(const ViewParams& vp, int P, char* geom, const uint32_t* heavy_list, const uint32_t* heavy_count,
                            OccCand* cand, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const GeomLayout L(P);
    OccHeader* hdr = reinterpret_cast<OccHeader*>(geom + L.occ_hdr);
    const int n_slots = 4 * ((P + 255) / 256);
    const int lb = occlusion_block_log2(vp.gx, vp.gy), B = 1 << lb;
    const int nbx = (vp.gx + B - 1) / B, nby = (vp.gy + B - 1) / B;
    uint32_t* hist = reinterpret_cast<uint32_t*>(geom + L.occ_hdr + sizeof(OccHeader));
    const dim3 gslots((n_slots + OCC_GATHER_SLOTS - 1) / OCC_GATHER_SLOTS);
    hipLaunchKernelGGL(occ_hist_kernel, gslots, dim3(256), 0, s, heavy_list, heavy_count, n_slots, hdr, hist);
    hipLaunchKernelGGL(occ_gather_kernel, gslots, dim3(256), 0, s, P, (const char*)geom, heavy_list, heavy_count, n_slots, hdr,
                       (const uint32_t*)hist, cand, (uint32_t)lb, (uint32_t)nbx, (uint32_t)nby);
    hipLaunchKernelGGL(occ_cover_kernel, dim3(nbx * nby), dim3(OCC_COVER_THREADS), 0, s, vp, B, nbx, hdr, (const OccCand*)cand,
                       reinterpret_cast<uint32_t*>(geom + L.occ_cut));
    hipLaunchKernelGGL(occ_recount_kernel, dim3((P + OCC_RECOUNT_THREADS - 1) / OCC_RECOUNT_THREADS), dim3(OCC_RECOUNT_THREADS), 0, s, vp,
                       P, geom);
    return hipGetLastError();
}

}  // namespace msgs
