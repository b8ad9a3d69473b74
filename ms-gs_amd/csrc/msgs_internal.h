// msgs_internal.h — layouts and device helpers shared by the HIP translation units of
// libmsgs_hip.so (gfx950 only; wave64 is hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include "../../include/msgs.h"

namespace msgs {

constexpr int TILE = MSGS_TILE;
constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------
// HBM layouts.  All sub-arrays start on 256-byte boundaries.
// ---------------------------------------------------------------------------------------------
// Per-Gaussian record read by the blend kernels: 3 x float4 = 48 B, one aligned AoS record so a
// gather costs three 16-B loads from one or two 64-B lines.
// The conic is stored pre-scaled into the log2 domain (k = -1/2 log2 e):
//   r0 = { px, py, k*conic.A, k*conic.B }
//   r1 = { k*conic.C, log2(opacity_eff), r, g }
//   r2 = { b, depth, pixel_size, tau2 }     tau2 = -log2(255*opacity_eff) - margin (exact-cull bound;
//                                           > 0 if the Gaussian can never reach alpha 1/255, -3e38 if the
//                                           form is not negative definite and cannot be bounded)
struct __attribute__((aligned(16))) GaussRec { float4 r0, r1, r2; };

// Per-Gaussian record read by emit_kernel through the depth order (a random gather): everything it needs in one
// aligned 32-byte line instead of pieces of four arrays —
//   q0 = { px, py, k*conic.A, k*conic.B },  q1 = { k*conic.C, tau2, bits(minx | miny<<16), bits(maxx | maxy<<16) }
struct __attribute__((aligned(32))) BinRec { float4 q0, q1; };

// Per-Gaussian 2-D gradient record accumulated by the blend backward: 10 accumulators.
//   [0] sum q dx  [1] sum q dy  [2] sum q dx^2  [3] sum q dx dy  [4] sum q dy^2  [5] sum q  [6..8] dL/drgb
//   [9] pad, with q = alpha_raw dL/dalpha; preprocess_backward_kernel turns [0..5] into
//   dL/dmean2D (NDC-ish units), dL/dconic (A, B-half, C) and dL/dopacity with per-Gaussian factors.
// The accumulators are DOUBLES (80-byte records, global_atomic_add_f64): the per-tile float32 sums of a Gaussian are
// then added exactly, whatever the order the tiles' atomics arrive in — the default backward is reproducible to the
// last float bit after the final rounding, like the CPU oracle (double accumulators, one cast), and the order noise of
// float atomics (up to 1.5e-3 of the dL/dscale max-norm on a C4 view through K8's amplification) is gone.
// -DMSGS_GRAD_REC_F32 builds the float32 variant (48-byte records) for A/B measurements.
#if defined(MSGS_GRAD_REC_F32)
typedef float grad_acc_t;
constexpr int GRAD_REC_FLOATS = 12;     // accumulators per record (9 used)
#else
typedef double grad_acc_t;
constexpr int GRAD_REC_FLOATS = 10;     // accumulators per record (9 used)
#endif
constexpr uint32_t TILE_ORDER_MAGIC = 0x4C505431u;
constexpr size_t GRAD_REC_BYTES = sizeof(grad_acc_t) * GRAD_REC_FLOATS;
constexpr int DET_INST_FLOATS = 9;      // per tile entry in deterministic mode: the nine K7 sums (DOUBLES)

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// ---------------------------------------------------------------------------------------------
// Per-tile occlusion cut-off (occlusion.hip; round 5).  A Gaussian whose alpha >= 1/255 level set contains a whole block of
// tiles blends into EVERY pixel of it with at least alpha_min (the value at the far corner); once the product of
// (1 - alpha_min) over such covers, front to back, has fallen below 1e-4 — with a factor 2 of slack — every pixel of the block
// has met the termination rule T (1 - alpha) < 1e-4, and no entry behind that depth is ever evaluated there:
// those instances are not counted, emitted or sorted.  Outputs, n_contrib and all gradients are bit-identical to the uncut
// path.  What it buys: the multi-scale model rendered without its filters (render.py's defaults) has 427 M instances of which a
// few million are traversed.
// Region inside `geom` (fixed size: msgs_geom_bytes(P) does not know the image): header + one cut-off depth key per cover
// BLOCK of B x B tiles, B = the smallest power of two from 4 up for which the grid has at most
// OCC_MAX_BLOCKS blocks — the table then fits the LDS of the two kernels that look tiles up in it (emit, recount).
// ---------------------------------------------------------------------------------------------
constexpr int OCC_MAX_BLOCKS = 2048;
constexpr int OCC_MAX_CAND = 32768;         // cover candidates kept per view: the NEAREST ones, whole depth buckets (or, when the
                                            // nearest bucket alone holds more, every stride-th of all in index order)
constexpr uint32_t OCC_HEAVY_MIN = 96;      // tile instances a Gaussian needs to be a cover candidate (= emit's wave path)
constexpr int OCC_BUCKETS = 2048;           // depth buckets of the front-to-back accumulation: key >> 19, i.e. 1/16 octave
constexpr int OCC_KEY_SHIFT = 19;
constexpr uint32_t OCC_KEY_BASE = 0x3E4CCCCDu >> OCC_KEY_SHIFT;   // bucket of depth 0.2 (the near cull, Q1)
struct OccHeader {
    uint32_t n_cand;          // cover candidates the gather kept (<= OCC_MAX_CAND)
    uint32_t any_closed;      // non-zero when some cover block received a finite cut-off (plain stores of 1 by the cover kernel)
    uint32_t n_heavy;         // Gaussians with more than OCC_HEAVY_MIN tile instances (n_cand of them are sampled as covers)
    uint32_t enabled;         // 1 when the pass ran for this view
    uint32_t block_log2;      // log2 of the tiles per side of a cover block
    uint32_t nbx;             // cover blocks per row of blocks
    uint32_t nby;             // rows of cover blocks (nbx * nby <= OCC_MAX_BLOCKS)
    uint32_t n_written;       // candidate records appended so far (positions of the depth-selected gather; ends at n_cand)
    uint32_t depth_limit;     // depth bucket up to which candidates were kept (OCC_BUCKETS - 1: all; 0xFFFFFFFF: stride sample)
    uint32_t bar[3];          // arrival counters of the pass's three grid barriers (occlusion.hip)
    uint32_t watchdog;        // non-zero: a barrier wait expired (the pass stopped early; the state is valid, the cut weaker)
    uint32_t pad[3];          // (no instance statistics: one atomic per wave on a shared word cost 0.4 ms on a view that
                              //  drops 420 M instances; run the view with msgs_set_occlusion(0) to learn the uncut count)
};
static_assert(sizeof(OccHeader) == 64, "OccHeader layout");
// candidate record: { px, py, kA, kB | kC, log2 o, key bits, gaussian id | tile rect (minx | miny << 16, maxx | maxy << 16), -, - }
// (the rect travels with the candidate: tile instances exist only inside it, and the alpha >= 1/255 level set of an opaque
//  Gaussian — up to 3.33 sigma along the major axis — reaches past the 3-sigma rect: a block outside the rect is never covered)
constexpr int SLAB_SCAN_CHUNK = 4096;       // = SCAN_CHUNK (scan_blocks), needed by GeomLayout ahead of its definition
struct __attribute__((aligned(16))) OccCand { float4 c0, c1; uint32_t rect_lo, rect_hi, pad0, pad1; };
static_assert(sizeof(OccCand) == 48, "OccCand layout");

// Depth-slab binning (round 6; msgs_view_t.slab_fraction).  Device words of one forward:
struct SlabHeader {
    uint32_t rA;              // slab A = depth ranks [0, rA)                                   (slab_split_kernel)
    uint32_t DA;              // its tile instances = offs[rA]
    uint32_t DB;              // instances slab B emits: every instance of the view that falls into an OPEN tile (B's scan)
    uint32_t active;          // 1 when this forward ran in slab mode
    uint32_t n_open;          // tiles in which some pixel was still blending at the end of its slab-A list (blend A, atomics)
    uint32_t pad0;
    uint64_t total_b;         // B's scan total (64-bit twin of DB)
    uint32_t pad[8];
};
static_assert(sizeof(SlabHeader) == 64, "SlabHeader layout");

struct GeomLayout {
    size_t rec, binrec, tiles, key, flags, weight, order, offs, nvalid, skey, slab_hdr, occ_hdr, occ_cut, offs_b, scan_b, litrec, total;
    __host__ __device__ explicit GeomLayout(int64_t P) {
        size_t o = 0;
        rec = o;    o = align256(o + sizeof(GaussRec) * P);
        binrec = o; o = align256(o + sizeof(BinRec) * P);   // what emit needs, in ONE 32-byte line (rendered Gaussians)
        tiles = o;  o = align256(o + 4 * P);        // exact tile-overlap count
        key = o;    o = align256(o + 4 * P);        // depth sort key (float bits / 0xFFFFFFFF)
        flags = o;  o = align256(o + 4 * P);        // bit0..2 colour clamped, bit3 rendered
        weight = o; o = align256(o + 4 * P);        // multi-scale fade weight
        order = o;  o = align256(o + 4 * P);        // Gaussian ids in depth order
        offs = o;   o = align256(o + 4 * P);        // exclusive scan of tiles[order[r]]
        nvalid = o; o += 256;                       // one word: V = Gaussians that stayed in the (compacting) depth sort —
                                                    // order[0..V) / offs[0..V) are the ranks the scan and the emit cover
        skey = o;   o = align256(o + 4 * P);        // depth keys in depth order (the sort's key output): emit's cut-off test
        slab_hdr = o; o += 256;                     // SlabHeader, directly in front of the occlusion header: K1 clears both with
                                                    // one range
        occ_hdr = o; o = align256(o + sizeof(OccHeader) + 8 * (size_t)OCC_BUCKETS);   // header + depth histogram + fill counters of the
                                                                                        // cover candidates (both cleared by K1)
        occ_cut = o; o = align256(o + 4 * (size_t)OCC_MAX_BLOCKS);  // cut-off depth bucket per cover block (0xFFFF = open)
        offs_b = o; o = align256(o + 4 * P);        // slab B: per depth rank, instances in open tiles -> their exclusive scan
        scan_b = o; o = align256(o + 8 * (size_t)((P + SLAB_SCAN_CHUNK - 1) / SLAB_SCAN_CHUNK + 2));   // block totals of that scan
        litrec = o; o = align256(o + 16 * P);       // verification mode (literal.hip): raw conic A, B, C and effective opacity
        total = o;
    }
};

// radix sort geometry: 256 threads x ITEMS keys per block; ITEMS = 4 below SORT_MID_N keys (more, smaller blocks: 100k keys
// 59 -> 50 us), 8 from there (at 1M keys 8 per thread is faster, 76 vs 88 us), 16 from SORT_BIG_N (longer digit runs per block ->
// better write coalescing of the scatter: 94 -> 81 us at 4.1M pairs, 1.6 -> 1.2 ms at 55M)
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 4;
constexpr int64_t SORT_BIG_N = 2'000'000;
constexpr int64_t SORT_MID_N = 400'000;
constexpr int SORT_MAX_GROUPS = 128;             // group sums per digit and pass below SORT_SCANNED_MIN_BLOCKS blocks
constexpr int64_t SORT_SCANNED_MIN_BLOCKS = 4096;    // from here on (16.8 M pairs) a one-block kernel turns the group sums into bases
constexpr int SORT_SCANNED_GSIZE = 32;
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;
static_assert(SCAN_CHUNK == SLAB_SCAN_CHUNK, "GeomLayout::scan_b is sized with SLAB_SCAN_CHUNK");

__host__ __device__ inline int64_t scan_blocks(int64_t n) { return (n + SCAN_CHUNK - 1) / SCAN_CHUNK; }

// Geometry of one radix pass over n pairs: every block leaves a digit histogram (hist[block][digit]) and adds it to the sums of
// its GROUP of gsize blocks (gsum[group][digit], integer atomics); a scatter block derives its output bases from the group sums
// and the histograms of the earlier blocks of its own group.  Shared by the sort and by the callers that size and clear its tables.
struct SortGeom {
    bool big, mid, scanned;
    int items, gsize, ngroups;
    int64_t nb;
    __host__ __device__ explicit SortGeom(int64_t n) {
        if (n < 1) n = 1;
        big = n >= SORT_BIG_N;
        mid = !big && n >= SORT_MID_N;
        items = big ? 16 : (mid ? 8 : SORT_ITEMS);
        nb = (n + (int64_t)SORT_THREADS * items - 1) / ((int64_t)SORT_THREADS * items);
        scanned = big && nb >= SORT_SCANNED_MIN_BLOCKS;
        if (scanned) {
            gsize = SORT_SCANNED_GSIZE;  // short in-group walks; the number of groups is no longer what a block reads
        } else {
            // a scatter block reads `ngroups` group sums + on average gsize/2 block histograms per digit: balance them
            gsize = 8;
            while ((int64_t)gsize * gsize < nb) gsize += 8;
            const int64_t floor_g = (nb + SORT_MAX_GROUPS - 1) / SORT_MAX_GROUPS;
            if (floor_g > gsize) gsize = (int)floor_g;
        }
        ngroups = (int)((nb + gsize - 1) / gsize);
    }
    // words behind the block-histogram table that have to be zero before the first pass: the group sums of every pass
    __host__ __device__ size_t zero_words(int passes) const { return (size_t)passes * 256 * ngroups; }
};

// scratch needed by one radix_sort_pairs call over n pairs: the alternate key / value buffers of the ping-pong, the block
// histograms and the group sums of up to four passes
struct SortScratch {
    size_t keys_alt, vals_alt, hist, total;
    __host__ __device__ explicit SortScratch(int64_t n) {
        size_t o = 0;
        const size_t nn = (size_t)(n > 0 ? n : 1);
        const SortGeom G(n);
        keys_alt = o; o = align256(o + 4 * nn);
        vals_alt = o; o = align256(o + 4 * nn);
        hist = o;     o = align256(o + 4 * ((size_t)256 * G.nb + G.zero_words(4)));
        total = o;
    }
};

struct Stage1Scratch {
    size_t sort, scan_partials, total_out, heavy_list, heavy_count, heavy_blk, occ_cand, total;
    __host__ __device__ explicit Stage1Scratch(int64_t P) {
        size_t o = 0;
        const size_t Pn = (size_t)(P > 0 ? P : 1);
        const size_t waves = 4 * ((Pn + 255) / 256);          // wave slots of preprocess_kernel's grid
        sort = o;          o = align256(o + SortScratch(P).total);
        scan_partials = o; o = align256(o + 8 * (size_t)(scan_blocks(P > 0 ? P : 1) + 2));
        total_out = o;     o = align256(o + 64);    // {scan total} and the collected status block
        // occlusion cut-off: cover candidates as preprocess_kernel leaves them (per wave: up to 64 Gaussian ids + a count),
        // and their gathered records (one per candidate)
        heavy_list = o;    o = align256(o + 8 * 64 * waves);     // {Gaussian id, depth key} pairs
        heavy_count = o;   o = align256(o + 4 * waves);
        heavy_blk = o;     o = align256(o + 2 * waves);          // ... and per workgroup of preprocess_kernel (four waves): {count, sum
                                                                 //     of the candidates' largest possible cover weights}
        occ_cand = o;      o = align256(o + sizeof(OccCand) * OCC_MAX_CAND);
        total = o;
    }
};

// (the tile ranges come FIRST: the position of every part is then independent of D, so that stage 2 can be launched on
//  capacity-sized buffers before the host knows D, and the backward finds the parts whatever capacity the forward used)
constexpr int DTRAV_SLOTS = 64;                  // accumulators of the traversed-entries count (feedback), one 8-byte word per slot
struct BinningLayout {
    size_t ids, ranges, dtrav, total;
    __host__ __device__ BinningLayout(int64_t D, int64_t tiles) {
        size_t o = 0;
        ranges = o; o = align256(o + 8 * (size_t)tiles);
        dtrav = o;  o = align256(o + 8 * (size_t)DTRAV_SLOTS);    // directly behind the ranges: the emit's zero job clears both at once
        ids = o;    o = align256(o + 4 * (size_t)(D > 0 ? D : 1));
        total = o;
    }
    // words from the start of `ranges` to the end of the D_trav accumulators
    __host__ __device__ size_t ranges_and_dtrav_words() const { return (dtrav + 8 * (size_t)DTRAV_SLOTS - ranges) / 4; }
};

// Depth-slab binning: how the instance array `ids` (capacity cb) and the stage-2 scratch (capacity cs) are shared.
//   slab A's sorted ids         ids[0, nA)        nA <= n_a_max = fraction * D + tiles (+ alignment)
//   slab B's sorted ids         ids[n_a_max, n_a_max + DB)
// DB counts every instance of the view that falls into an open tile: DB <= D up to the handful of tiles by which two
// evaluations of a row extent may differ (msgs_internal.h, levelset_row_interval) — SLAB_SLACK instances are kept free for
// them in both buffers (one such tile per 5.5e7 instances was observed), and a count beyond the capacity raises
// SlabHeader::pad0 (reported with the feedback publication; asserted zero by the tests).
constexpr int64_t SLAB_SLACK = 65536;
constexpr int SLAB_MIN_TILES = 2048;             // below: the blend kernels are latency-bound, slabs only add launches
struct SlabGeom {
    int64_t cap_d, n_a_max, cap_b;               // largest D served; start of slab B's ids; capacity of slab B (instances)
    bool ok;
    // cap_d_limit: the largest D the caller wants served (the exact D when it is known, else a huge number)
    __host__ SlabGeom(int64_t cb, int64_t cs, double fraction, int64_t tiles, int64_t cap_d_limit) {
        const int64_t by_ids = (int64_t)((double)(cb - tiles - SLAB_SLACK - 256) / (1.0 + fraction));
        const int64_t by_scratch = cs - SLAB_SLACK;
        cap_d = by_ids < by_scratch ? by_ids : by_scratch;
        if (cap_d > cap_d_limit) cap_d = cap_d_limit;
        if (cap_d < 0) cap_d = 0;
        n_a_max = (((int64_t)(fraction * (double)cap_d) + tiles + 64) + 63) & ~(int64_t)63;
        cap_b = cap_d + SLAB_SLACK;
        ok = cap_d > 0 && n_a_max + cap_b <= cb && cap_b <= cs && n_a_max + cap_b <= 0xFFFFFFFFll;
    }
    // instance capacity of `ids` that serves D instances
    __host__ static int64_t ids_needed(int64_t D, double fraction, int64_t tiles) {
        return (int64_t)((double)(D + 1) * (1.0 + fraction)) + tiles + SLAB_SLACK + 512;
    }
};

// EMIT_HEAVY_MIN: instances from which a Gaussian is emitted cooperatively (by its wave inside emit_kernel, or — in the blocks
// that write straight to HBM — by a workgroup of emit_heavy_kernel that takes it from the queue)
constexpr uint32_t EMIT_HEAVY_MIN = 96;
struct Stage2Scratch {
    size_t keys_a, ids_a, sort, heavy_q, total;
    __host__ __device__ explicit Stage2Scratch(int64_t D) {
        size_t o = 0;
        int64_t n = D > 0 ? D : 1;
        keys_a = o; o = align256(o + 4 * (size_t)n);
        ids_a = o;  o = align256(o + 4 * (size_t)n);
        sort = o;   o = align256(o + SortScratch(n).total);
        // queue of the depth ranks whose Gaussian has more than EMIT_HEAVY_MIN instances: word 0 = their number (cleared by
        // the scan that precedes a speculative stage 2, or by msgs_forward_stage2 itself), then at most n / EMIT_HEAVY_MIN ranks
        heavy_q = o; o = align256(o + 4 * ((size_t)n / EMIT_HEAVY_MIN + 66));
        total = o;
    }
};

struct ImageLayout {
    size_t final_T, n_contrib, tile_last, tile_order, open_bits, open_list, total;
    __host__ __device__ ImageLayout(int64_t W, int64_t H) {
        size_t o = 0;
        const size_t tiles = (size_t)((W + TILE - 1) / TILE) * (size_t)((H + TILE - 1) / TILE);
        final_T = o;    o = align256(o + 4 * (size_t)(W * H));
        n_contrib = o;  o = align256(o + 4 * (size_t)(W * H));
        tile_last = o;  o = align256(o + 4 * tiles);   // per tile: position behind the last entry any of its pixels blended
                                                       // (written by the forward blend; = the entries the backward traverses)
        tile_order = o; o = align256(o + 4 * (tiles + 1));   // backward launch order: heaviest tiles first inside every XCD's
                                                       // run; word [tiles] = TILE_ORDER_MAGIC when valid (written by the
                                                       // forward's order kernel, cleared by every forward blend)
        open_bits = o;  o = align256(o + 4 * ((tiles + 31) / 32 + 1));   // slab mode: bit t = tile t is open after slab A
        open_list = o;  o = align256(o + 4 * tiles);   // ... and the open tiles as a list (SlabHeader::n_open entries, any order)
        total = o;
    }
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
// One Adam step of ONE float, term by term torch.optim.Adam's single-tensor formulation (torch/optim/adam.py,
// _single_tensor_adam; amsgrad = False, weight_decay = 0, maximize = False) with the roundings of the ATen kernels:
//     m <- fma(1 - beta1, g - m, m)                      (Tensor.lerp_, weight < 0.5 branch)
//     v <- fma((1 - beta2) * g, g, v * beta2)            (mul_ then addcmul_)
//     p <- p + ((-lr / (1 - beta1^t)) * m) / (sqrt(v) / sqrt(1 - beta2^t) + eps)      (addcdiv_)
// Shared by the multi-tensor optimizer kernel (epilogue.hip) and the per-Gaussian backward that steps on the spot
// (preprocess.hip, msgs_adam_in_backward_t): the two give the same bits.
struct AdamScalars {
    float w1;            // 1 - beta1
    float beta2, w2;     // beta2, 1 - beta2
    float bc2_sqrt, eps;
};
inline AdamScalars adam_scalars(int64_t step, double beta1, double beta2, double eps) {
    AdamScalars a;
    a.w1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    a.eps = (float)eps;
    return a;
}
// -lr / bias_correction1, rounded from double like torch's Python scalar
inline float adam_neg_step_size(double lr, int64_t step, double beta1) { return (float)(-(lr / (1.0 - pow(beta1, (double)step)))); }

// the optimizer step the per-Gaussian backward takes itself (msgs_grads_t::adam_in_backward): kernel-side table, tensors in the
// order means3D, features_dc, features_rest, opacities, scales, rotations
struct AdamInBackward {
    float* m[6];
    float* v[6];
    float nss[6];
    AdamScalars a;
};

#if defined(__HIPCC__)

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float nss, const AdamScalars& a) {
#pragma clang fp contract(off)
    m = __fmaf_rn(a.w1, g - m, m);
    v = __fmaf_rn(a.w2 * g, g, v * a.beta2);
    const float denom = __fsqrt_rn(v) / a.bc2_sqrt + a.eps;
    p = p + (nss * m) / denom;
}

// Exact-culling test in the log2 domain.  The record stores the NEGATIVE-definite form
//   f(d) = A dx^2 + 2 Bh dx dy + C dy^2  = log2 G(d)     (A, Bh, C = -1/2 log2(e) x conic)
// and alpha(d) >= 1/255  <=>  f(d) >= tau2 = -log2(255 o).  Does the level set {f >= tau2} of a Gaussian
// centred at (gx, gy) reach a pixel centre of the rectangle [x0,x1] x [y0,y1]?  f is concave with its
// maximum (0) at the centre, so the maximum over a rectangle that does not contain the centre lies on
// one of the (at most two) edges facing it.  Conservative by construction (tau2 carries a safety
// margin), written with explicit roundings so that every caller (count, emit, quadrant masks) gets
// bit-identical answers whatever the surrounding code is.
__device__ __forceinline__ bool levelset_hits_rect(float gx, float gy, float A, float Bh, float C,
                                                   float tau2, float x0, float x1, float y0, float y1) {
#pragma clang fp contract(off)   // HIP's __f*_rn are plain operators: pin contraction here, not per TU
    const float dxlo = __fsub_rn(gx, x1), dxhi = __fsub_rn(gx, x0);
    const float dylo = __fsub_rn(gy, y1), dyhi = __fsub_rn(gy, y0);
    const float cx = fminf(fmaxf(0.0f, dxlo), dxhi);
    const float cy = fminf(fmaxf(0.0f, dylo), dyhi);
    if (cx == 0.0f && cy == 0.0f) return true;          // centre inside the rectangle
    float fbest = -3.0e38f;
    if (cx != 0.0f) {                                    // facing vertical edge dx = cx
        float dy = __fdiv_rn(-__fmul_rn(Bh, cx), C);
        dy = fminf(fmaxf(dy, dylo), dyhi);
        const float u = __fmaf_rn(A, cx, __fmul_rn(Bh, dy)), w = __fmaf_rn(C, dy, __fmul_rn(Bh, cx));
        fbest = fmaxf(fbest, __fmaf_rn(cx, u, __fmul_rn(dy, w)));
    }
    if (cy != 0.0f) {                                    // facing horizontal edge dy = cy
        float dx = __fdiv_rn(-__fmul_rn(Bh, cy), A);
        dx = fminf(fmaxf(dx, dxlo), dxhi);
        const float u = __fmaf_rn(A, dx, __fmul_rn(Bh, cy)), w = __fmaf_rn(C, cy, __fmul_rn(Bh, dx));
        fbest = fmaxf(fbest, __fmaf_rn(dx, u, __fmul_rn(cy, w)));
    }
    return fbest >= tau2;
}

// Per-tile-ROW extent of the same level set, used by the overlap count in preprocess_kernel and by emit_kernel.
// The two inlined copies are NOT guaranteed to round identically (observed: one tile in 55 M at C5, where an
// extent ended exactly on a tile boundary), so the scheme does not rely on it: the COUNT is taken with a larger
// safety margin (LEVELSET_MARGIN_COUNT) than the EMIT (LEVELSET_MARGIN_EMIT), hence count >= emitted always; the
// emit is still a superset of the exact tile set, and the surplus slots are parked on the sentinel tile id
// (= number of tiles), which sorts behind every real tile and is ignored by ranges_kernel.  {f >= tau2} is an ellipse; its
// intersection with the horizontal band of one tile row is convex, so the tiles it reaches in that row
// form ONE interval [tlo, thi]: the x-range of (ellipse ∩ band) is bounded by the ellipse's extreme
// points when they fall inside the band and by the band edges' chords otherwise.  O(rows) work per
// Gaussian instead of O(tiles).  Conservative w.r.t. the per-pixel test (continuous rectangle + margin).
constexpr float LEVELSET_MARGIN_EMIT = 0.02f;    // pixels
constexpr float LEVELSET_MARGIN_COUNT = 0.03f;   // pixels (> emit)
struct LevelSetRows {
    float A, Bh, C, Atau, det, invA, dymax, dyR;
};
__device__ __forceinline__ LevelSetRows levelset_rows_setup(float A, float Bh, float C, float tau2) {
#pragma clang fp contract(off)   // count (preprocess.hip) and emit (binning.hip) must agree bit for bit
    LevelSetRows r;
    r.A = A; r.Bh = Bh; r.C = C;
    r.det = __fsub_rn(__fmul_rn(A, C), __fmul_rn(Bh, Bh));          // > 0 (negative definite form)
    r.Atau = __fmul_rn(A, tau2);                                     // > 0
    r.invA = __frcp_rn(A);
    r.dymax = __fsqrt_rn(__fdiv_rn(r.Atau, r.det));
    const float dx_ext = __fsqrt_rn(__fdiv_rn(__fmul_rn(C, tau2), r.det));
    r.dyR = __fdiv_rn(-__fmul_rn(Bh, dx_ext), C);                    // dy where dx is extreme (+dx_ext)
    return r;
}
// tile row ty (pixel rows 16 ty .. 16 ty + 15); returns false when the row is not reached
__device__ __forceinline__ bool levelset_row_interval(const LevelSetRows& r, float gx, float gy, int ty, int minx,
                                                      int maxx, float margin, int& tlo, int& thi) {
#pragma clang fp contract(off)
    const float y0 = (float)(ty * TILE);
    const float dym = r.dymax + margin;
    const float lo = fmaxf(__fsub_rn(gy, y0 + (float)(TILE - 1)), -dym);
    const float hi = fminf(__fsub_rn(gy, y0), dym);
    if (lo > hi) return false;
    const float dyr = fminf(fmaxf(r.dyR, lo), hi);                   // where dx is largest inside the band
    const float dyl = fminf(fmaxf(-r.dyR, lo), hi);                  // where dx is smallest
    const float sr = __fsqrt_rn(fmaxf(0.0f, __fmaf_rn(-r.det, __fmul_rn(dyr, dyr), r.Atau)));
    const float sl = __fsqrt_rn(fmaxf(0.0f, __fmaf_rn(-r.det, __fmul_rn(dyl, dyl), r.Atau)));
    const float dx_max = __fmul_rn(__fsub_rn(-__fmul_rn(r.Bh, dyr), sr), r.invA);   // A < 0
    const float dx_min = __fmul_rn(__fadd_rn(-__fmul_rn(r.Bh, dyl), sl), r.invA);
    const float xl = __fsub_rn(__fsub_rn(gx, dx_max), margin);       // pixel x = gx - dx
    const float xr = __fadd_rn(__fsub_rn(gx, dx_min), margin);
    tlo = max(minx, (int)ceilf(__fmul_rn(__fsub_rn(xl, (float)(TILE - 1)), 1.0f / TILE)));
    thi = min(maxx - 1, (int)floorf(__fmul_rn(xr, 1.0f / TILE)));
    return tlo <= thi;
}

// DPP move helper (gfx9 DPP controls: quad_perm 0x00-0xFF, row_shl 0x101-0x10F, row_shr 0x111-0x11F,
// row_ror 0x121-0x12F, row_mirror 0x140, row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143)
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = false>
__device__ __forceinline__ float dpp_mov(float v, float old = 0.0f) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK,
                                                      BANK_MASK, BOUND));
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = false>
__device__ __forceinline__ unsigned dpp_mov_u(unsigned v, unsigned old = 0u) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND);
}

// Position-preserving all-reduce across the four 16-lane rows of a wave64: lane l ends up with
// x[l%16] + x[l%16+16] + x[l%16+32] + x[l%16+48].  gfx950's v_permlane16_swap exchanges the odd rows of
// its first operand with the even rows of its second, v_permlane32_swap the upper half of the first
// with the lower half of the second; fed the same value twice, the two results add up to the pair sum
// in every lane.  (DPP row_bcast15/31 cannot be used here: it broadcasts ONE lane to a whole row.)
__device__ __forceinline__ float cross_row_allreduce(float x) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(a.x) + __uint_as_float(a.y);
    const u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b.x) + __uint_as_float(b.y);
}

// the same sum through the LDS crossbar: no VALU cycles beyond the two adds (addr16 / addr32 = (lane ^ 16) << 2,
// (lane ^ 32) << 2, computed once per kernel)
__device__ __forceinline__ float cross_row_allreduce_bperm(float x, int addr16, int addr32) {
    x += __int_as_float(__builtin_amdgcn_ds_bpermute(addr16, __float_as_int(x)));
    x += __int_as_float(__builtin_amdgcn_ds_bpermute(addr32, __float_as_int(x)));
    return x;
}

// broadcast of lane `src` (wave-uniform) without the LDS crossbar: __shfl with a run-time lane index compiles to ds_bpermute_b32
// (an LDS round trip per value); with a uniform index v_readlane_b32 does it in one VALU-to-SGPR move
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// every lane receives the wave total
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v += dpp_mov<0xB1>(v);            // xor 1
    v += dpp_mov<0x4E>(v);            // xor 2
    v += dpp_mov<0x124>(v);           // + quad (q-1)
    v += dpp_mov<0x128>(v);           // + quads (q-2, q-3): every lane = row sum
    return cross_row_allreduce(v);
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
    return v;
}

// The cover-block table of the occlusion cut-off in LDS, with what its two readers (recount, emit) ask first: the smallest
// and the largest cut-off and the largest per ROW of blocks (a Gaussian that lies deeper has nothing to emit in that row of
// blocks).  Cut-offs are depth BUCKETS (occ_bucket of a depth key; 0xFFFF = the block stayed open): a Gaussian reaches a block
// iff its own bucket is <= the block's.  4.6 KB of LDS.  Called by all threads of the block; ends with a barrier.
constexpr int OCC_MAX_BLOCK_ROWS = 256;
constexpr int OCC_LIGHT_RECT = 1024;        // tiles of rect up to which a Gaussian behind a cut-off is walked by its own thread
__device__ __forceinline__ uint32_t occ_bucket(uint32_t depth_key) {
    const uint32_t kb = depth_key >> OCC_KEY_SHIFT;
    return kb > OCC_KEY_BASE ? min(kb - OCC_KEY_BASE, (uint32_t)(OCC_BUCKETS - 1)) : 0u;
}
struct OccTable {
    uint16_t cut[OCC_MAX_BLOCKS];
    uint16_t rowmax[OCC_MAX_BLOCK_ROWS];
    uint32_t red[2][16];
    uint32_t cut_min, cut_max;
};
__device__ __forceinline__ void occ_table_load(OccTable& t, const uint32_t* __restrict__ occ_cut, int nbx, int nby) {
    const int n = nbx * nby;
    uint32_t lo = 0xFFFFu, hi = 0u;
    for (int q = threadIdx.x; q < n; q += blockDim.x) {
        const uint32_t c = min(occ_cut[q], 0xFFFFu);
        t.cut[q] = (uint16_t)c;
        lo = min(lo, c);
        hi = max(hi, c);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if ((threadIdx.x & 63) == 0) { t.red[0][threadIdx.x >> 6] = lo; t.red[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    for (int by = threadIdx.x; by < nby; by += blockDim.x) {
        uint32_t m = 0u;
        for (int bx = 0; bx < nbx; ++bx) m = max(m, (uint32_t)t.cut[by * nbx + bx]);
        t.rowmax[by] = (uint16_t)m;
    }
    if (threadIdx.x == 0) {
        uint32_t a = 0xFFFFu, b = 0u;
        for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) { a = min(a, t.red[0][w]); b = max(b, t.red[1][w]); }
        t.cut_min = a;
        t.cut_max = b;
    }
    __syncthreads();
}

#endif  // __HIPCC__

// ---------------------------------------------------------------------------------------------
// host-side launch entry points implemented in the .hip files
// ---------------------------------------------------------------------------------------------
struct ViewParams {            // by-value kernel argument
    int W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy;
    float scale_modifier, fade_size;
    int sh_degree, sh_coeffs;
    int filter_small, filter_large;
    const float* bg;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    int cell_sx, cell_sy;         // log2 of the tiles per coarse screen cell (at most 8 x 8 cells), -1: no cell ranges (below)
};

// Coarse cell range of a Gaussian's tile rect, packed into the 12 bits above its tile count in GeomLayout::tiles (a count is at
// most gx * gy < 2^20 whenever the ranges are used): { first column, last column, first row, last row } of the 8 x 8 coarse cells
// the rect touches, 3 bits each.  The scan of the counts gathers tiles[] through the depth order anyway; in slab mode it hands
// the ranges on in depth order, and slab B's recount answers "nothing of this Gaussian lies in a cell with an open tile" from
// them without fetching the Gaussian's record (binning.hip).
constexpr uint32_t TILE_COUNT_BITS = 20, TILE_COUNT_MASK = (1u << TILE_COUNT_BITS) - 1u;
__host__ __device__ inline uint32_t cell_range_pack(int minx, int miny, int maxx, int maxy, int sx, int sy) {   // rect: [min, max)
    return (uint32_t)(minx >> sx) | ((uint32_t)((maxx - 1) >> sx) << 3) | ((uint32_t)(miny >> sy) << 6) | ((uint32_t)((maxy - 1) >> sy) << 9);
}
__host__ __device__ inline uint64_t cell_range_mask(uint32_t packed) {
    const uint32_t c0 = packed & 7u, c1 = (packed >> 3) & 7u, r0 = (packed >> 6) & 7u, r1 = (packed >> 9) & 7u;
    const uint64_t cols = ((2ull << c1) - (1ull << c0)) & 0xFFull;
    uint64_t m = 0;
    for (uint32_t r = r0; r <= r1; ++r) m |= cols << (8u * r);
    return m;
}

inline ViewParams make_view_params(const msgs_view_t* v) {
    ViewParams p;
    p.W = v->image_width; p.H = v->image_height;
    p.gx = (p.W + TILE - 1) / TILE; p.gy = (p.H + TILE - 1) / TILE;
    p.tanfovx = v->tanfovx; p.tanfovy = v->tanfovy;
    p.fx = p.W / (2.0f * v->tanfovx); p.fy = p.H / (2.0f * v->tanfovy);
    p.scale_modifier = v->scale_modifier; p.fade_size = v->fade_size;
    p.sh_degree = v->sh_degree; p.sh_coeffs = v->sh_coeffs;
    p.filter_small = v->filter_small; p.filter_large = v->filter_large;
    p.bg = v->bg; p.viewmatrix = v->viewmatrix; p.projmatrix = v->projmatrix; p.campos = v->campos;
    p.cell_sx = p.cell_sy = -1;
    if ((int64_t)p.gx * p.gy < (int64_t)(1u << TILE_COUNT_BITS)) {
        p.cell_sx = p.cell_sy = 0;
        while ((p.gx + (1 << p.cell_sx) - 1) >> p.cell_sx > 8) ++p.cell_sx;
        while ((p.gy + (1 << p.cell_sy) - 1) >> p.cell_sy > 8) ++p.cell_sy;
    }
    return p;
}

// up to two word ranges a kernel clears on behalf of later launches (thread t clears word t of each range): saves the
// separate fill launches in front of the radix sorts and the tile-range pass
struct ZeroJob { uint32_t* p0; size_t n0; uint32_t* p1; size_t n1; };

// preprocess.hip
// heavy_list / heavy_count (nullable): per wave of the grid, the ids of its Gaussians with more than OCC_HEAVY_MIN tile
// instances (cover candidates of the occlusion cut-off) and their number
hipError_t launch_preprocess(const ViewParams& vp, const msgs_gaussians_t& g, int32_t* radii, float* pixel_sizes,
                             char* geom, hipStream_t s, ZeroJob zj = ZeroJob{nullptr, 0, nullptr, 0},
                             uint32_t* heavy_list = nullptr, uint32_t* heavy_count = nullptr, uint32_t* heavy_blk = nullptr,
                             bool write_litrec = false);     // verification mode: also GeomLayout::litrec (raw conic, opacity)
// occlusion.hip: gather the candidates, accumulate the covers per block of tiles front to back, recount the Gaussians behind a
// cut-off (tiles[] / key[] of `geom` are updated in place, before the depth sort)
hipError_t launch_occlusion(const ViewParams& vp, int P, char* geom, const uint32_t* heavy_list, const uint32_t* heavy_count,
                            const uint32_t* heavy_blk, OccCand* cand, hipStream_t s);
int set_occlusion(int on);                // occlusion.hip: process-wide switch (MSGS_NO_OCCLUSION=1 initially off); returns previous
int get_occlusion();
int occlusion_block_log2(int gx, int gy);  // log2(tiles per side of a cover block) for a gx x gy grid (4 x 4 tiles unless the grid has more than 2048 such blocks)
hipError_t launch_preprocess_backward(const ViewParams& vp, const msgs_gaussians_t& g, const int32_t* radii,
                                      const char* geom, const grad_acc_t* grad_rec, const msgs_grads_t& grads,
                                      hipStream_t s, bool textbook = false);
// literal.hip: the verification mode (msgs_set_deterministic) — the reference's blend loops restated literally
hipError_t launch_blend_forward_literal(const ViewParams& vp, const char* geom, int P, const uint32_t* ids, const uint2* ranges,
                                        float* out_color, float* out_ps, float* out_depth, float* final_T, uint32_t* n_contrib,
                                        uint32_t* order_flag, hipStream_t s);
hipError_t launch_blend_backward_literal(const ViewParams& vp, const char* geom, int P, const uint32_t* ids, const uint2* ranges,
                                         const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor, double* inst_grad,
                                         hipStream_t s);
hipError_t launch_sh_grad_from_views(int P, int n_views, int deg, const float* means3D, const float* campos,
                                     int64_t campos_stride, const float* drgb, int64_t drgb_stride, float scale,
                                     float* d_dc, float* d_rest, hipStream_t s);
hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                               uint8_t* present, hipStream_t s);
// sort.hip
// Stable LSD radix sort of n (u32 key, u32 val) pairs on bits [begin_bit, end_bit).  vals_in may be
// NULL (identity).  The result lands in keys_out/vals_out (which may alias neither input); the
// inputs are clobbered when more than one pass is needed.
hipError_t radix_sort_pairs(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                            int64_t n, int begin_bit, int end_bit, char* scratch /* SortScratch(n) */,
                            hipStream_t s, bool pre_zeroed = false, uint32_t* n_valid_dev = nullptr,
                            const uint32_t* n_dev = nullptr, bool keys16 = false);
// keys16: keys_in / keys_out hold uint16 keys (tile ids below 65536; 6 instead of 8 bytes per pair and pass); allowed when
// radix_sort_keys16_ok(n, begin_bit, end_bit)
bool radix_sort_keys16_ok(int64_t n, int begin_bit, int end_bit);
bool radix_sort_zero_region(int64_t n, int begin_bit, int end_bit, char* scratch, uint32_t** ptr, size_t* words);
bool radix_sort_supports_device_count(int64_t n, int begin_bit, int end_bit);

// out[r] = exclusive sum of in[gather ? gather[r] : r]; *total (device, u64) = grand total
// n_ptr (optional, device word): the number of elements actually scanned, <= n (n sizes the grids)
// in_mask / side_out (gathered scans only): the scanned value is in[gather[i]] & in_mask, and side_out[i] — when given —
// receives in[gather[i]] >> TILE_COUNT_BITS (the cell ranges of GeomLayout::tiles, in depth order)
hipError_t exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n,
                              uint64_t* partials /* scan_blocks(n)+2 */, uint64_t* total, hipStream_t s,
                              uint64_t* status = nullptr, uint64_t* host_mapped = nullptr, uint64_t ticket = 0,
                              const uint32_t* n_ptr = nullptr, uint32_t* clamped_total = nullptr, uint64_t clamp = 0,
                              const uint32_t* extra = nullptr,    // extra: two device words forwarded with the status
                              uint32_t* zero_word = nullptr,      // zero_word: one device word cleared on behalf of a later launch
                              uint32_t* overflow_flag = nullptr, uint32_t in_mask = 0xFFFFFFFFu, uint32_t* side_out = nullptr,
                              uint32_t* side_flag = nullptr);    // side_flag: set to 1 when side_out was written // set to 1 when the total exceeds `clamp`
// device-side fill with zeros (sort.hip): an ordinary kernel launch — hipMemsetAsync costs ~10 us of queue latency per
// call on this runtime (barrier packets around the fill), four of them per step were 3 % of the C3 step
hipError_t launch_zero(void* ptr, size_t bytes, hipStream_t s);     // ptr and bytes multiples of 4
// binning.hip
// D_dev (optional, device word): the instance count when the host does not know it yet (speculative stage 2); D is then the
// CAPACITY the grids and the geometry are sized for
// keys16: the key arrays hold uint16 tile ids (grids of fewer than 65535 tiles)
// slab: 0 = the whole view; 1 = slab A (ranks below SlabHeader::rA of `geom`; D_dev = &SlabHeader::DA); 2 = slab B (every rank,
// into the tiles of open_bits only, slots from GeomLayout::offs_b; D_dev = &SlabHeader::DB)
hipError_t launch_emit(const ViewParams& vp, int P, const char* geom, uint32_t* keys, uint32_t* ids,
                       int64_t D, hipStream_t s, ZeroJob zj = ZeroJob{nullptr, 0, nullptr, 0}, const uint32_t* D_dev = nullptr,
                       bool keys16 = false, uint32_t* heavy_q = nullptr, int slab = 0, const uint32_t* open_bits = nullptr,
                       int64_t density_D = 0);      // density_D: the WHOLE view's instance count (LDS stage choice of slab A)
// slab mode, between stage 1 and slab A's emit: SlabHeader {rA, DA, ...} from the scanned offsets; clears the open-tile bitmap
hipError_t launch_slab_split(int P, char* geom, int64_t D, const uint32_t* D_dev, float fraction, uint32_t* open_bits,
                             int num_tiles, hipStream_t s);
// slab mode, behind slab A's blend: GeomLayout::offs_b[r] = instances of rank r in open tiles (not yet scanned)
hipError_t launch_slab_recount(const ViewParams& vp, int P, char* geom, const uint32_t* open_bits, const uint32_t* open_list, int64_t D,
                               const uint32_t* D_dev, hipStream_t s);
// heavy_q (Stage2Scratch::heavy_q; word 0 must be zero when the launch starts): Gaussians with more than EMIT_HEAVY_MIN
// instances in blocks that write straight to HBM are queued and emitted by a second launch, one workgroup per Gaussian
// base: position in `ids` of the first id that belongs to keys[0] (slab B's ids start behind slab A's)
hipError_t launch_ranges(const uint32_t* keys, int64_t D, uint2* ranges, int num_tiles, hipStream_t s,
                         bool pre_zeroed = false, const uint32_t* D_dev = nullptr, bool keys16 = false, uint32_t base = 0);
int set_backward_generation(int gen);     // blend.hip: 0 = by tile count, 1 | 2 = forced; returns the previous value
int set_blend_granularity(int mode);      // blend.hip: 0 = by tile count, 1 = coarse, 2 = fine (16 waves per tile)
// blend.hip, deterministic backward: scratch = [grad_rec | inst_grad | sort buffers]
struct DetScratch {
    size_t grad_rec, inst_grad, keys, keys_s, entry, sort, total;
    DetScratch(int64_t P, int64_t D);
};
hipError_t launch_blend_backward_det(const ViewParams& vp, int P, const char* geom, const uint32_t* ids, int64_t D,
                                     const uint2* ranges, const float* final_T, const uint32_t* n_contrib,
                                     const float* dL_dcolor, char* scratch, hipStream_t s);

// epilogue.hip
hipError_t launch_adam(const msgs_adam_tensor_t* tensors, int n, int64_t step, double beta1, double beta2, double eps,
                       hipStream_t s);
hipError_t launch_densify_stats(const msgs_densify_stats_t& d, hipStream_t s);

// loss.hip
size_t loss_scratch_bytes(int C, int H, int W);
void ssim_window_host(float w[11]);
hipError_t launch_loss_forward(const float* img, const float* gt, int C, int H, int W, float lambda, float* out3,
                               char* scratch, int write_maps, hipStream_t s);
hipError_t launch_loss_backward(const float* img, const float* gt, int C, int H, int W, float lambda,
                                const float* upstream, const char* scratch, float* dL_dimg, hipStream_t s);

// knn.hip
size_t knn_scratch_bytes(int64_t P);
hipError_t knn_mean_dist2(const float* points, int64_t P, float* mean_dist2, char* scratch, hipStream_t s);

// voxel_pool.hip
size_t voxel_pool_scratch_bytes(int64_t M);
hipError_t voxel_pool_build(const float* positions, int64_t M, float voxel_size, uint32_t* order, uint32_t* seg_start,
                            int32_t* voxel_index, char* scratch, int64_t* num_voxels_host, hipStream_t s);
hipError_t voxel_pool_average(const float* features, int F, const uint32_t* order, const uint32_t* seg_start,
                              int64_t Mv, float* out, hipStream_t s);
// blend.hip
// slab (nullable): pass 0 = plain (only tile_len is used), 1 = slab A (publishes the tiles left open), 2 = slab B (blends the
// listed tiles only).  Passes 1 / 2 need forward_uses_quadrant_kernel(tiles).
struct FwdSlabArgs {
    int pass;
    uint32_t* open_bits;      // ImageLayout::open_bits
    uint32_t* open_list;      // ImageLayout::open_list
    uint32_t* n_open;         // &SlabHeader::n_open
    unsigned long long* dtrav;   // BinningLayout::dtrav (nullable: no feedback)
};
hipError_t launch_blend_forward(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                float* out_color, float* out_ps, float* out_depth, float* final_T,
                                uint32_t* n_contrib, uint32_t* tile_last, void* clear_ptr, size_t clear_bytes, hipStream_t s,
                                const FwdSlabArgs* slab = nullptr);
bool forward_uses_quadrant_kernel(int tiles);
// msgs_view_t.feedback_tag: {D_trav, tag, slab numbers, D} into words 4..7 of a pinned status block (host_mapped: its device address)
hipError_t launch_forward_feedback(const unsigned long long* dtrav, const SlabHeader* hdr, int slab_mode, int64_t D,
                                   const uint32_t* D_dev, uint32_t tag, uint64_t ticket, uint64_t* host_mapped, hipStream_t s);
hipError_t launch_blend_backward(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 grad_acc_t* grad_rec, hipStream_t s, const uint32_t* tile_order = nullptr);
// heaviest-first launch order of the one-wave-per-tile backward from the forward's per-tile traversal lengths
hipError_t launch_tile_order(const ViewParams& vp, const uint32_t* tile_last, uint32_t* tile_order, hipStream_t s);
hipError_t launch_blend_lane_stats(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                   unsigned long long* out3 /* device, zeroed inside */, hipStream_t s);
hipError_t launch_blend_backward_lane_stats(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                            const float* final_T, const uint32_t* n_contrib,
                                            unsigned long long* out4 /* device, zeroed inside */, hipStream_t s);
hipError_t launch_binning_stats(const ViewParams& vp, int P, const int32_t* radii, const uint32_t* n_contrib,
                                unsigned long long* out2 /* device, zeroed inside */, hipStream_t s);

}  // namespace msgs
