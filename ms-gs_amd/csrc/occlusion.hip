// occlusion.hip — exact per-tile occlusion cut-off in front of the depth sort and the emit (round 5; one launch since round 6).
//
// The reference emits, sorts and walks every (tile, Gaussian) instance of a view; a pixel stops walking its tile's list at the
// first entry where T (1 - alpha) < 1e-4.  When a multi-scale MS-GS model is rendered WITHOUT its
// pixel-size filters (/root/reference/render.py:32 and the evaluation loop /root/reference/train.py:488-496 call render() with
// the default flags) the coarse-level Gaussians — scaled x4 .. x64 — are all drawn at level 0: 427 M instances at 1080p for
// 1 M Gaussians, of which a few million are ever walked, because a handful of opaque giants in front terminates every pixel.
//
// What this pass proves and uses.  Let a Gaussian's alpha >= 1/255 level set contain a whole block of tiles that lies inside
// its tile rect.  alpha is a concave function of the pixel offset in the log domain, so its minimum over the block's pixel
// centres is at one of the four corner pixels: alpha_min.  EVERY pixel of the block then blends the Gaussian with
// alpha >= alpha_min (it is not skipped: the skip rule is alpha < 1/255), i.e. its transmittance behind that entry is at most
// (1 - alpha_min) times the one in front.  With the covers of a block taken front to back, the first depth at which
// prod (1 - alpha_min) < 0.5e-4  (the rule's 1e-4 with a factor 2 of slack for float32 rounding on either side) is a depth
// behind which NO pixel of the block evaluates anything: every one of them has met the termination test at or before that
// entry.  Instances behind it are dropped from the tile counts (here, before the depth sort and the scan) and from the emit
// (binning.hip).  The lists every pixel actually walks are unchanged, entry for entry: image, n_contrib, final_T and every
// gradient are bit-identical to the uncut path (tests/test_occlusion_gpu.py).  Any SUBSET of the covers gives a valid (later)
// cut-off, so the pass may ignore what it likes: it only looks at Gaussians with more than OCC_HEAVY_MIN tile instances, only
// at blocks they cover completely, and it accumulates per depth BUCKET (1/16 octave of view depth) instead of per rank — the
// cut-off is the far end of the bucket in which the product crosses.  Integer (fixed-point) sums: the result does not depend
// on the order the candidates arrive in.
//
// ONE launch between preprocess_kernel and the depth sort (four in round 5): occ_pass_kernel, a persistent grid of OCC_GRID
// workgroups.  Every workgroup first adds up the candidate counts preprocess_kernel left per workgroup (a few KB) — no barrier,
// no shared counter to clear:
//   fewer than three candidates in the whole view (a cover weighs at most 6.65 of the 14.3 bits needed; an ordinary training
//     view has none): every workgroup leaves — 4.6 us at BASELINE C3 including the launch;
//   up to OCC_SMALL candidates (BASELINE C3: four): every workgroup collects ALL of them into its LDS, ranks them by depth
//     bucket, and its waves take the tile blocks one each: lane <-> candidate, one 64-lane prefix sum, the bucket of the lane
//     at which the product crosses;
//   more: the phases of round 5 behind grid barriers — (only with more than OCC_MAX_CAND candidates: depth histogram of the
//     candidates -- barrier -- keep the nearest OCC_MAX_CAND, whole depth buckets) gather the records (then: in depth-bucket
//     order; otherwise every candidate adds its weight to the TOTAL of the blocks in its rect on the way, and when no total
//     reaches the threshold — BASELINE C5 — everybody leaves behind the next barrier) -- barrier -- per group of up to four blocks of tiles: one walk over the records, bucketed sums of
//     -log2(1 - alpha_min), prefix, cut-off bucket; over depth-ordered records the walk stops when every block of the group has
//     its 14.3 bits (the filters-off multi-scale model: 185 k heavy Gaussians, 32 k kept, closed within the nearest few hundred).
// Then ONE more grid barrier, behind which every workgroup knows whether anything closed: if not (the rule), it leaves; if so,
// the Gaussians behind the nearest cut-off recount their tile instances (index order, tiles[] / key[] in place; a Gaussian left
// without instances leaves the depth sort: key 0xFFFFFFFF).  The pass runs on EVERY forward, and the wrapper's adaptive skip
// policy of round 5 — with its cliff, a closing view inside the 31 skipped calls rendered uncut — is gone.
// Grid barriers: one counter per barrier in the header preprocess_kernel clears; the table, the flags and the candidate records
// are written with agent-scope (write-through) stores — no release fence anywhere: one would write back everything
// preprocess_kernel left dirty in the XCD's L2, 60 us —; every workgroup takes one agent-scope acquire before it reads another
// workgroup's data (cdna_hip_programming.md G16).  The waits are bounded.
// A workgroup whose wait expires raises OccHeader::watchdog and leaves, and whatever the others still do the state stays VALID:
// a table entry is either open or a proven cut-off; the cover phase starts only after EVERY workgroup has finished the gather;
// any subset of covers, of blocks and of recounted Gaussians is a valid (weaker) cut (count >= emitted, surplus slots go to the
// sentinel tile).  Never observed; msgs_occlusion_stats reports it.
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

constexpr int OCC_THREADS = 256;            // threads per workgroup of the pass
constexpr int OCC_GRID = 256;               // persistent workgroups: one per CU (1024 waves of the chip's 8192)
constexpr int OCC_GATHER_SLOTS = 256;       // wave slots per chunk of phases 1 / 2 (one per thread)
constexpr int OCC_MAX_SPINS = 1 << 20;      // bound of a barrier wait (~1 s)

#define OCC_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Grid barrier k of the pass: every wave drains its stores, the workgroup meets, ONE lane publishes (agent-scope release, then
// the arrival on the counter), polls the counter with relaxed agent-scope loads and takes ONE agent-scope acquire; the second
// __syncthreads() extends it to the workgroup.  Returns false when the wait expired (or somebody else's had).
// RELEASE = false: everything the workgroup published was written with agent-scope (write-through) atomic stores or atomics.
template <bool RELEASE>
__device__ __forceinline__ bool occ_grid_barrier(OccHeader* hdr, int k, uint32_t* s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (RELEASE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&hdr->bar[k], 1u, OCC_RLX_AGENT);
        uint32_t ok = 1u;
        for (int spins = 0; __hip_atomic_load(&hdr->bar[k], OCC_RLX_AGENT) < gridDim.x; ++spins) {
            if (spins > OCC_MAX_SPINS || __hip_atomic_load(&hdr->watchdog, OCC_RLX_AGENT) != 0u) {
                __hip_atomic_store(&hdr->watchdog, 1u, OCC_RLX_AGENT);
                ok = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        // (the acquire is the caller's: behind the last barrier only the workgroups that go on to read the table need one)
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0u;
}

__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t v, uint32_t* s_w, uint32_t* block_total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();                                   // (s_w may still be read from the previous use)
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < wv; ++k) base += s_w[k];
    *block_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + inc - v;
}

// LDS of the pass: the phases use it one after the other
struct OccGatherLds {
    uint16_t kept[OCC_GATHER_SLOTS * 64];     // (local slot << 6) | index in the slot
    uint32_t prefix[OCC_GATHER_SLOTS];
    uint32_t red[4], scan[4], kappa, keep, base;
    uint32_t start[OCC_BUCKETS];              // depth-ordered gather: first record position of every depth bucket (view-wide)
    uint32_t place[OCC_BUCKETS];              // ... and, per chunk, the chunk's count per bucket -> its next free position
};
constexpr int OCC_SMALL = 64;               // up to this many candidates every workgroup handles the whole view's covers itself
union OccLds {
    uint32_t hist[OCC_BUCKETS];               // depth histogram of a chunk
    uint32_t sums[4][OCC_BUCKETS];            // bucket sums of a group of cover blocks (OCC_GROUP = 4)
    OccGatherLds g;                           // selection + gather
    OccTable table;                           // recount
};

// ---- phase 1: one chunk of OCC_GATHER_SLOTS wave slots -------------------------------------------------------------------
// preprocess_kernel left, per wave slot, {id, depth key} of its heavy Gaussians and their number.  Every chunk counts its
// candidates per depth bucket in LDS and adds the non-empty buckets and its total to the view's histogram / counter (cleared
// by preprocess_kernel).
__device__ __forceinline__ void occ_hist_chunk(int chunk, const uint32_t* __restrict__ heavy_list,
                                               const uint32_t* __restrict__ heavy_count, int n_slots,
                                               uint32_t* hist, uint32_t* s_h, uint32_t* s_w) {
    const int slot = chunk * OCC_GATHER_SLOTS + threadIdx.x;
    const uint32_t cnt = slot < n_slots ? heavy_count[slot] : 0u;
    uint32_t total;
    block_exclusive_256(cnt, s_w, &total);
    if (total == 0) return;                                             // (workgroup-uniform)
    __syncthreads();
    for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_THREADS) s_h[k] = 0u;
    __syncthreads();
    const uint2* e = reinterpret_cast<const uint2*>(heavy_list) + (size_t)slot * 64;
    for (uint32_t j = 0; j < cnt; ++j) atomicAdd(&s_h[occ_bucket(e[j].y)], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_THREADS)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}

// ---- phase 2: selection of the candidates ---------------------------------------------------------------------------------
// Only the NEAREST candidates matter (the product crosses within the first few dozen covers of a block), so at most
// OCC_MAX_CAND are kept, chosen by depth: every workgroup prefix-sums the histogram itself and finds the deepest bucket kappa up
// to which the candidates still fit (OccSelect); a chunk keeps its candidates with bucket <= kappa (positions: one atomic per
// chunk — the ORDER of the records is irrelevant, the cover sums are integers; the SET is deterministic) and gathers their
// records.  When the nearest non-empty bucket alone holds more than fit (thousands of covers at one depth), every stride-th
// candidate in index order is kept instead: positions by formula from all slot counts.
struct OccSelect { uint32_t total, limit, stride, keep_total; bool by_depth, ordered, bound; };
__device__ __forceinline__ uint32_t cover_weight(const OccCand& c, float x0, float x1, float y0, float y1, int tx0, int tx1,
                                                 int ty0, int ty1);
constexpr int OCC_BOUND_BLOCKS = 64;        // blocks a candidate adds its weight to while it is gathered (S.bound, below) ...
constexpr int OCC_BOUND_WAVE_BLOCKS = 512;  // ... with its whole wave above that, and not at all beyond this (a flag instead)
__device__ __forceinline__ OccSelect occ_select(uint32_t total, const uint32_t* hist, OccGatherLds& L) {
    // deepest bucket kappa with (candidates in buckets <= kappa) <= OCC_MAX_CAND; thread t owns PER consecutive buckets
    constexpr int PER = OCC_BUCKETS / OCC_THREADS;
    uint32_t hv[PER], hsum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { hv[k] = hist[threadIdx.x * PER + k]; hsum += hv[k]; }
    if (threadIdx.x == 0) { L.kappa = 0xFFFFFFFFu; L.keep = 0u; }
    uint32_t dummy;
    uint32_t run = block_exclusive_256(hsum, L.scan, &dummy);           // (its barriers order the two stores above)
    uint32_t best = 0xFFFFFFFFu, keep = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        L.start[threadIdx.x * PER + k] = run;                           // exclusive prefix: where this bucket's records start
        run += hv[k];
        if (hv[k] && run <= (uint32_t)OCC_MAX_CAND) { best = (uint32_t)(threadIdx.x * PER + k); keep = run; }
    }
    if (best != 0xFFFFFFFFu) atomicMax(&L.keep, keep);                  // cumulative counts grow with the bucket index: the
    __syncthreads();                                                    //   largest admissible cumulative count marks kappa
    if (best != 0xFFFFFFFFu && keep == L.keep) L.kappa = best;
    __syncthreads();
    OccSelect S;
    S.total = total;
    S.keep_total = L.keep;
    S.by_depth = total <= (uint32_t)OCC_MAX_CAND || L.kappa != 0xFFFFFFFFu;
    S.limit = total <= (uint32_t)OCC_MAX_CAND ? (uint32_t)(OCC_BUCKETS - 1) : L.kappa;
    S.stride = S.by_depth ? 1u : (total + OCC_MAX_CAND - 1) / OCC_MAX_CAND;
    S.ordered = S.by_depth;      // a selection by depth places the records in depth-bucket order (occ_gather_chunk)
    S.bound = false;
    __syncthreads();
    return S;
}

__device__ __forceinline__ void occ_gather_chunk(int chunk, const OccSelect& S, int P, const char* __restrict__ geom,
                                                 const uint32_t* __restrict__ heavy_list,
                                                 const uint32_t* __restrict__ heavy_count, int n_slots, OccHeader* hdr,
                                                 OccCand* cand, OccGatherLds& L, uint32_t* __restrict__ fill,
                                                 const ViewParams& vp, int lb, int nbx) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot0 = chunk * OCC_GATHER_SLOTS;
    const int slot = slot0 + threadIdx.x;
    const uint32_t cnt = slot < n_slots ? heavy_count[slot] : 0u;
    uint32_t dummy, chunk_total;
    const uint32_t before_in_chunk = block_exclusive_256(cnt, L.scan, &chunk_total);
    if (chunk_total == 0) return;                                       // (workgroup-uniform)
    uint32_t before = 0u;
    if (!S.by_depth) {
        // the nearest bucket alone is too full: every stride-th candidate in index order; the candidates in front of this
        // chunk's slots from ALL slot counts (62 KB at 1 M Gaussians, four counts per load)
        const int n4 = n_slots >> 2;                        // n_slots is a multiple of 4
        const uint4* c4 = reinterpret_cast<const uint4*>(heavy_count);
        for (int q = threadIdx.x; q < n4 && 4 * q < slot0; q += OCC_THREADS) {
            const uint4 v = c4[q];
            before += v.x + v.y + v.z + v.w;
        }
        for (int off = 32; off > 0; off >>= 1) before += (uint32_t)__shfl_xor((int)before, off);
        __syncthreads();
        if (lane == 0) L.red[wv] = before;
        __syncthreads();
        before = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    }
    const uint2* ent = reinterpret_cast<const uint2*>(heavy_list);
    uint32_t kept = 0, first = 0, g0 = 0;
    if (S.by_depth) {
        for (uint32_t j = 0; j < cnt; ++j) kept += occ_bucket(ent[(size_t)slot * 64 + j].y) <= S.limit ? 1u : 0u;
    } else {
        // position (index order) of this slot's first candidate; positions g with g % stride == 0 are kept
        first = before + before_in_chunk;
        g0 = ((first + S.stride - 1) / S.stride) * S.stride;
        kept = g0 < first + cnt ? (first + cnt - 1 - g0) / S.stride + 1 : 0u;
    }
    L.prefix[threadIdx.x] = first;
    uint32_t kept_total;
    uint32_t at = block_exclusive_256(kept, L.scan, &kept_total);       // <= 256 * 64 entries
    if (S.ordered) {
        // depth-ordered placement: the chunk counts its kept candidates per bucket (LDS), reserves a run of positions in every
        // non-empty bucket with ONE atomic on the view's fill counter (cleared by preprocess_kernel), and its candidates take
        // the positions start[bucket] + run + rank.  The order INSIDE a bucket is arbitrary — the cover sums are per bucket
        for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_THREADS) L.place[k] = 0u;
        __syncthreads();
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t bk = occ_bucket(ent[(size_t)slot * 64 + j].y);
            if (bk <= S.limit) { L.kept[at++] = (uint16_t)((threadIdx.x << 6) | j); atomicAdd(&L.place[bk], 1u); }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_THREADS) {
            const uint32_t c = L.place[k];
            if (c) L.place[k] = L.start[k] + atomicAdd(&fill[k], c);
        }
    } else if (S.by_depth) {
        for (uint32_t j = 0; j < cnt; ++j)
            if (occ_bucket(ent[(size_t)slot * 64 + j].y) <= S.limit) L.kept[at++] = (uint16_t)((threadIdx.x << 6) | j);
        if (threadIdx.x == 0) L.base = kept_total ? atomicAdd(&hdr->n_written, kept_total) : 0u;
    } else {
        for (uint32_t g = g0; g < first + cnt; g += S.stride) L.kept[at++] = (uint16_t)((threadIdx.x << 6) | (g - first));
    }
    // S.bound: the chunk's contributions to the blocks' total weights are summed in LDS (L.place is free: no ordered placement)
    // and flushed with one atomic per non-zero block — thousands of candidates that each cover most of a dozen blocks (the
    // filters-off model at 240 x 135: 2.4 ms of atomics on twelve words when every candidate added to global memory itself)
    if (S.bound)
        for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_THREADS) L.place[k] = 0u;
    __syncthreads();
    const GeomLayout GL(P);
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + GL.binrec);
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom + GL.rec);
    for (uint32_t e = threadIdx.x; e - threadIdx.x < kept_total; e += OCC_THREADS) {       // (workgroup-uniform trip count)
        OccCand c;
        c.c0 = c.c1 = make_float4(0.f, 0.f, 0.f, 0.f);
        c.rect_lo = c.rect_hi = c.pad0 = c.pad1 = 0u;
        bool big = false;
        if (e < kept_total) {
        const uint32_t code = L.kept[e];
        const uint32_t ls = code >> 6, j = code & 63u;
        const uint2 ge = ent[(size_t)(slot0 + ls) * 64 + j];
        const BinRec b = binrec[ge.x];
        c.c0 = b.q0;                                                                  // px, py, kA, kB(half)
        c.c1 = make_float4(b.q1.x, rec[ge.x].r1.y, __uint_as_float(ge.y), __uint_as_float(ge.x));   // kC, log2 o, depth key, id
        c.rect_lo = __float_as_uint(b.q1.z);                                          // minx | miny << 16   (tiles)
        c.rect_hi = __float_as_uint(b.q1.w);                                          // maxx | maxy << 16   (exclusive)
        c.pad0 = c.pad1 = 0u;
        const uint32_t pos = S.ordered ? atomicAdd(&L.place[occ_bucket(ge.y)], 1u)
                                       : (S.by_depth ? L.base + e : (L.prefix[ls] + j) / S.stride);
        if (pos < (uint32_t)OCC_MAX_CAND) {
            // write-through (agent-scope) stores: the barrier behind this phase then needs no release fence — which would write
            // back everything preprocess_kernel left dirty in this XCD's L2 (60 us measured)
            uint64_t* d = reinterpret_cast<uint64_t*>(cand + pos);
            const uint64_t* w = reinterpret_cast<const uint64_t*>(&c);
#pragma unroll
            for (int k = 0; k < (int)(sizeof(OccCand) / 8); ++k) __hip_atomic_store(d + k, w[k], OCC_RLX_AGENT);
        }
        if (S.bound) {
            // every candidate is kept (at most OCC_MAX_CAND of them) and nobody needs the fill counters: while its record is in
            // registers the candidate adds its weight to the TOTAL of every block inside its rect — the same integer
            // cover_weight the cover phase would add to one of the block's buckets.  If afterwards no total reaches the 14.3
            // bits, no block can close and the cover phase — a walk over all records per group of blocks, 60 us at BASELINE C5,
            // whose 10 751 mid-sized candidates close nothing — is skipped.  (A rect of more than OCC_BOUND_BLOCKS blocks: below.)
            const int B = 1 << lb;
            const int minx = (int)(c.rect_lo & 0xFFFFu), miny = (int)(c.rect_lo >> 16);
            const int maxx = (int)(c.rect_hi & 0xFFFFu), maxy = (int)(c.rect_hi >> 16);
            const int bx0 = minx >> lb, bx1 = (maxx - 1) >> lb, by0 = miny >> lb, by1 = (maxy - 1) >> lb;
            big = (bx1 - bx0 + 1) * (by1 - by0 + 1) > OCC_BOUND_BLOCKS;
            if (!big) {
                for (int by = by0; by <= by1; ++by)
                    for (int bx = bx0; bx <= bx1; ++bx) {
                        const int tx0 = bx * B, ty0 = by * B;
                        const int tx1 = min(tx0 + B, vp.gx), ty1 = min(ty0 + B, vp.gy);
                        const uint32_t w = cover_weight(c, (float)(tx0 * TILE), (float)(min(tx1 * TILE, vp.W) - 1), (float)(ty0 * TILE),
                                                        (float)(min(ty1 * TILE, vp.H) - 1), tx0, tx1, ty0, ty1);
                        if (w) atomicAdd(&L.place[by * nbx + bx], w);
                    }
            }
        }
        }
        // ... and a candidate with a larger rect is taken by its whole wave, lanes on the blocks of the rect
        if (S.bound) {
            uint64_t bm = __ballot(big);
            while (bm) {
                const int src = __ffsll((long long)bm) - 1;
                bm &= bm - 1;
                OccCand g;
                g.c0 = make_float4(__shfl(c.c0.x, src), __shfl(c.c0.y, src), __shfl(c.c0.z, src), __shfl(c.c0.w, src));
                g.c1 = make_float4(__shfl(c.c1.x, src), __shfl(c.c1.y, src), __shfl(c.c1.z, src), __shfl(c.c1.w, src));
                g.rect_lo = (uint32_t)__shfl((int)c.rect_lo, src);
                g.rect_hi = (uint32_t)__shfl((int)c.rect_hi, src);
                g.pad0 = g.pad1 = 0u;
                const int B = 1 << lb;
                const int minx = (int)(g.rect_lo & 0xFFFFu), miny = (int)(g.rect_lo >> 16);
                const int maxx = (int)(g.rect_hi & 0xFFFFu), maxy = (int)(g.rect_hi >> 16);
                const int bx0 = minx >> lb, bx1 = (maxx - 1) >> lb, by0 = miny >> lb, by1 = (maxy - 1) >> lb;
                const int nbw = bx1 - bx0 + 1, nblk = nbw * (by1 - by0 + 1);
                if (nblk > OCC_BOUND_WAVE_BLOCKS) {     // a giant: where there are such, blocks do close — the cover phase decides
                    if (lane == 0) __hip_atomic_store(&hdr->pad[0], 1u, OCC_RLX_AGENT);
                    continue;
                }
                for (int q = lane; q < nblk; q += 64) {
                    const int bx = bx0 + q % nbw, by = by0 + q / nbw;
                    const int tx0 = bx * B, ty0 = by * B;
                    const int tx1 = min(tx0 + B, vp.gx), ty1 = min(ty0 + B, vp.gy);
                    const uint32_t w = cover_weight(g, (float)(tx0 * TILE), (float)(min(tx1 * TILE, vp.W) - 1), (float)(ty0 * TILE),
                                                    (float)(min(ty1 * TILE, vp.H) - 1), tx0, tx1, ty0, ty1);
                    if (w) atomicAdd(&L.place[by * nbx + bx], w);
                }
            }
        }
    }
    __syncthreads();
    if (S.bound)
        for (int k = threadIdx.x; k < OCC_BUCKETS; k += OCC_THREADS) {
            const uint32_t w = L.place[k];
            if (w) atomicAdd(&fill[k], w);
        }
    (void)dummy;
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

// ---- phase 3: a GROUP of up to OCC_GROUP consecutive blocks of B x B tiles -------------------------------------------------
// One walk over the candidate records serves the whole group (at 4K a workgroup owns eight blocks: two walks instead of eight),
// four records in flight per thread, and every group starts its walk at a different record: 256 workgroups reading the same
// records in lockstep queue up on the same L2 channels (measured at BASELINE C5, 10 751 candidates, 2040 blocks: 102 us of the
// pass's 175 in this phase, a microsecond per pair of dependent record loads).  The sums are integers: the order is free.
// Writes the cut-off bucket of every block of the group whose product crosses (the table started all-open; a crossing in the
// last bucket — which also holds the keys clamped into it — closes nothing).
constexpr int OCC_GROUP = 4;
__device__ __forceinline__ void occ_cover_group(int blk0, int nb, const ViewParams& vp, int B, int nbx, uint32_t n,
                                                const OccCand* cand, uint32_t (*s_b)[OCC_BUCKETS], uint32_t* s_wave,
                                                uint32_t* s_cross, uint32_t* s_tot, uint32_t* occ_cut, OccHeader* hdr,
                                                bool ordered) {
    __syncthreads();                                    // (LDS of the previous group consumed)
    for (int k = threadIdx.x; k < nb * OCC_BUCKETS; k += OCC_THREADS) s_b[0][k] = 0u;
    if (threadIdx.x < OCC_GROUP) { s_cross[threadIdx.x] = 0xFFFFFFFFu; s_tot[threadIdx.x] = 0u; }
    __syncthreads();
    // the group's blocks, and the tile range that holds them all (a candidate whose rect misses it is done after four compares)
    int tx0[OCC_GROUP], ty0[OCC_GROUP], tx1[OCC_GROUP], ty1[OCC_GROUP];
    float x0[OCC_GROUP], y0[OCC_GROUP], x1[OCC_GROUP], y1[OCC_GROUP];
    int gx0 = 0x7FFFFFFF, gy0 = 0x7FFFFFFF, gx1 = 0, gy1 = 0;
#pragma unroll
    for (int j = 0; j < OCC_GROUP; ++j) {
        const int blk = blk0 + min(j, nb - 1);
        const int bx = blk % nbx, by = blk / nbx;
        tx0[j] = bx * B; ty0[j] = by * B;
        tx1[j] = min(tx0[j] + B, vp.gx); ty1[j] = min(ty0[j] + B, vp.gy);
        x0[j] = (float)(tx0[j] * TILE); y0[j] = (float)(ty0[j] * TILE);
        x1[j] = (float)(min(tx1[j] * TILE, vp.W) - 1); y1[j] = (float)(min(ty1[j] * TILE, vp.H) - 1);   // pixels inside the image
        gx0 = min(gx0, tx0[j]); gy0 = min(gy0, ty0[j]); gx1 = max(gx1, tx1[j]); gy1 = max(gy1, ty1[j]);
    }
    auto add = [&](const OccCand& cc) {
        const int minx = (int)(cc.rect_lo & 0xFFFFu), miny = (int)(cc.rect_lo >> 16);
        const int maxx = (int)(cc.rect_hi & 0xFFFFu), maxy = (int)(cc.rect_hi >> 16);
        if (minx >= gx1 || maxx <= gx0 || miny >= gy1 || maxy <= gy0) return;       // (a block must lie INSIDE the rect to count)
        const uint32_t bucket = occ_bucket(__float_as_uint(cc.c1.z));
#pragma unroll
        for (int j = 0; j < OCC_GROUP; ++j) {
            if (j >= nb) break;
            const uint32_t w = cover_weight(cc, x0[j], x1[j], y0[j], y1[j], tx0[j], tx1[j], ty0[j], ty1[j]);
            if (w) { atomicAdd(&s_b[j][bucket], w); if (ordered) atomicAdd(&s_tot[j], min(w, 0x00FFFFFFu)); }
        }
    };
    // ordered: the records are in depth-bucket order (occ_gather_chunk) and the walk starts at the front and STOPS as soon as
    // every block of the group has collected its 14.3 bits: every bucket in front of the one the last chunk ended in is then
    // complete, and the crossing lies in one of them or in that last bucket itself — whose sum can only grow —, i.e. the
    // prefix below finds the same bucket as after a full walk.  (A view full of opaque giants closes its blocks within the
    // nearest few hundred of 32 768 records.)  Otherwise: any order, every group starts somewhere else.
    const uint32_t rot = (n && !ordered) ? (uint32_t)(((uint64_t)(uint32_t)blk0 * 2654435761ull) % n) : 0u;
    constexpr int FLY = 4;
    for (uint32_t c = threadIdx.x; c - threadIdx.x < n; c += FLY * OCC_THREADS) {
        if (ordered && c != threadIdx.x) {              // (workgroup-uniform: c - threadIdx.x is)
            __syncthreads();
            bool done = true;
            for (int j = 0; j < nb; ++j) done = done && s_tot[j] >= OCC_THRESHOLD;
            __syncthreads();                            // (nobody adds to s_tot before everybody has read it)
            if (done) break;
        }
        if (c >= n) continue;
        OccCand r[FLY];
#pragma unroll
        for (int u = 0; u < FLY; ++u) {
            uint32_t at = c + (uint32_t)(u * OCC_THREADS);
            at = at < n ? at : threadIdx.x;                                 // (tail: a valid record, not added)
            at += rot;
            r[u] = cand[at >= n ? at - n : at];
        }
#pragma unroll
        for (int u = 0; u < FLY; ++u)
            if (c + (uint32_t)(u * OCC_THREADS) < n) add(r[u]);
    }
    __syncthreads();
    // front-to-back prefix over the buckets of each block: thread t owns PER consecutive buckets
    constexpr int PER = OCC_BUCKETS / OCC_THREADS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = 0; j < nb; ++j) {
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = min(s_b[j][threadIdx.x * PER + k], 0x00FFFFFFu); sum += v[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wave[4 * j + wv] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int k = 0; k < wv; ++k) run += s_wave[4 * j + k];
        uint32_t cross = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            run += v[k];
            if (cross == 0xFFFFFFFFu && run >= OCC_THRESHOLD) cross = (uint32_t)(threadIdx.x * PER + k);
        }
        if (cross != 0xFFFFFFFFu) atomicMin(&s_cross[j], cross);
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nb) {
        const uint32_t q = s_cross[threadIdx.x];        // everything up to and including the crossing bucket stays
        if (q < (uint32_t)(OCC_BUCKETS - 1)) {
            __hip_atomic_store(&occ_cut[blk0 + threadIdx.x], q, OCC_RLX_AGENT);
            __hip_atomic_store(&hdr->any_closed, 1u, OCC_RLX_AGENT);
        }
    }
}

// ---- recount --------------------------------------------------------------------------------------------------------------
// Gaussians behind the nearest cut-off count their instances again: the same per-row level-set extents and margin as the
// count in preprocess_kernel, restricted to the tiles whose block's cut-off they are in front of.  Index order (before the
// depth sort).  The block table sits in LDS.  Footprints with a rect of at most OCC_LIGHT_RECT tiles are recounted by their own
// thread — rows of blocks and blocks that are closed at this depth are skipped whole; larger ones by their whole wave: lane <->
// tile row for the row extents, then only the rows whose row of blocks is still open at this depth, with the lanes on
// consecutive tiles.
__device__ __forceinline__ void occ_recount_chunk(int chunk, const ViewParams& vp, int P, char* __restrict__ geom, int lb, int nbx,
                                                  const OccTable& T) {
    const GeomLayout L(P);
    const uint32_t cut_min = T.cut_min, cut_max = T.cut_max;
    uint32_t* tiles = reinterpret_cast<uint32_t*>(geom + L.tiles);
    uint32_t* key = reinterpret_cast<uint32_t*>(geom + L.key);
    const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + L.binrec);
    const int i = chunk * OCC_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0, k = 0;
    uint32_t cells = 0;                                                  // (the coarse cell range above the count stays)
    const uint32_t cmask = vp.cell_sx >= 0 ? TILE_COUNT_MASK : 0xFFFFFFFFu;
    if (i < P) { cnt = tiles[i]; cells = cnt & ~cmask; cnt &= cmask; k = occ_bucket(key[i]); }   // k: this Gaussian's depth bucket
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
        tiles[i] = newcnt | cells;
        if (newcnt == 0) key[i] = 0xFFFFFFFFu;                          // leaves the (compacting) depth sort
    }
}

// ---- the pass ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OCC_THREADS) void occ_pass_kernel(ViewParams vp, int P, char* geom,
                                                               const uint32_t* __restrict__ heavy_list,
                                                               const uint32_t* __restrict__ heavy_count,
                                                               const uint32_t* __restrict__ heavy_blk, int n_slots,
                                                               OccCand* cand, int block_log2, int nbx, int nby) {
    __shared__ OccLds lds;
    __shared__ OccCand s_cand[OCC_SMALL];
    __shared__ uint8_t s_rank[OCC_SMALL];
    __shared__ uint32_t s_w[16], s_ws[4], s_ok, s_cross[4], s_tot[4], s_n, s_nz, s_nzq[OCC_SMALL];
    const GeomLayout GL(P);
    OccHeader* hdr = reinterpret_cast<OccHeader*>(geom + GL.occ_hdr);
    uint32_t* hist = reinterpret_cast<uint32_t*>(geom + GL.occ_hdr + sizeof(OccHeader));
    uint32_t* occ_cut = reinterpret_cast<uint32_t*>(geom + GL.occ_cut);
    const int n_blocks = nbx * nby;
    const int n_chunks = (n_slots + OCC_GATHER_SLOTS - 1) / OCC_GATHER_SLOTS;
    const int n_wg = n_slots >> 2;                          // workgroups of preprocess_kernel (four wave slots each)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    // ---- how many candidates does the view have?  Every workgroup adds up preprocess_kernel's per-workgroup counts itself
    // ... and the sum of the largest weights they could possibly have (a cover's weight over any block is at most the one at its
    // own centre, -log2(1 - min(0.99, opacity))): below the threshold no block can close (BASELINE C3: four heavy Gaussians of
    // ordinary opacity)
    // (two pairs per load, four loads in flight, and every workgroup starts somewhere else: as a plain strided loop this sum was
    //  5.6 of the pass's 6.7 us at BASELINE C3 — 15 dependent L2 round trips — and 29 us at C5.)  On the way the workgroups of
    //  preprocess_kernel that HAVE candidates are noted: the few-candidates path below fetches only their lists
    uint32_t total = 0, wsum = 0;
    const uint2* blk2 = reinterpret_cast<const uint2*>(heavy_blk);
    if (threadIdx.x == 0) { s_n = 0u; s_nz = 0u; }
    __syncthreads();
    {
        const uint4* blk4 = reinterpret_cast<const uint4*>(heavy_blk);
        const int n4 = n_wg >> 1;
        const int rot = n4 ? (int)(((uint32_t)blockIdx.x * 2654435761u) % (uint32_t)n4) : 0;
        for (int j = threadIdx.x; j < n4; j += 4 * OCC_THREADS) {
            uint4 v[4];
            int qs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u * OCC_THREADS;
                int q = (jj < n4 ? jj : j) + rot;
                q = q >= n4 ? q - n4 : q;
                qs[u] = q;
                v[u] = blk4[q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j + u * OCC_THREADS >= n4) continue;
                total += v[u].x + v[u].z;
                wsum = min(wsum + min(v[u].y + v[u].w, 0x1FFFFFFFu), 0x3FFFFFFFu);
                if (v[u].x) { const uint32_t at = atomicAdd(&s_nz, 1u); if (at < (uint32_t)OCC_SMALL) s_nzq[at] = (uint32_t)(2 * qs[u]); }
                if (v[u].z) { const uint32_t at = atomicAdd(&s_nz, 1u); if (at < (uint32_t)OCC_SMALL) s_nzq[at] = (uint32_t)(2 * qs[u] + 1); }
            }
        }
        if ((n_wg & 1) && threadIdx.x == 0) {
            const uint2 v = blk2[n_wg - 1];
            total += v.x;
            wsum = min(wsum + v.y, 0x3FFFFFFFu);
            if (v.x) { const uint32_t at = atomicAdd(&s_nz, 1u); if (at < (uint32_t)OCC_SMALL) s_nzq[at] = (uint32_t)(n_wg - 1); }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        total += (uint32_t)__shfl_xor((int)total, off);
        wsum = min(wsum + (uint32_t)__shfl_xor((int)wsum, off), 0x3FFFFFFFu);
    }
    if (lane == 0) { s_w[wv] = total; s_ws[wv] = wsum; }
    __syncthreads();
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    wsum = min(min(s_ws[0] + s_ws[1], 0x3FFFFFFFu) + min(s_ws[2] + s_ws[3], 0x3FFFFFFFu), 0x7FFFFFFFu);
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // (the header was zeroed by preprocess_kernel)
        hdr->enabled = 1u;
        hdr->n_heavy = total;
        hdr->block_log2 = (uint32_t)block_log2;
        hdr->nbx = (uint32_t)nbx;
        hdr->nby = (uint32_t)nby;
        hdr->depth_limit = (uint32_t)(OCC_BUCKETS - 1);
        if (total >= 3u && wsum >= OCC_THRESHOLD && total <= (uint32_t)OCC_SMALL) hdr->n_cand = total;
    }
    if (total < 3u || wsum < OCC_THRESHOLD) return;     // nothing can close: n_cand = 0 / any_closed = 0, nobody reads the table
    const int B = 1 << block_log2;
    if (total <= (uint32_t)OCC_SMALL) {
        // ---- few candidates: every workgroup collects ALL of them (the set, not the order, matters: the sums are integers) ...
        // (at most OCC_SMALL workgroups of preprocess_kernel hold them: total <= OCC_SMALL)
        for (int z = threadIdx.x; z < (int)min(s_nz, (uint32_t)OCC_SMALL); z += OCC_THREADS) {
            const int q = (int)s_nzq[z];
            for (int w4 = 0; w4 < 4; ++w4) {
                const int slot = 4 * q + w4;
                const uint32_t c = heavy_count[slot];
                if (c == 0u) continue;
                const uint32_t at = atomicAdd(&s_n, c);
                const uint2* e = reinterpret_cast<const uint2*>(heavy_list) + (size_t)slot * 64;
                const BinRec* binrec = reinterpret_cast<const BinRec*>(geom + GL.binrec);
                const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom + GL.rec);
                for (uint32_t j = 0; j < c && at + j < (uint32_t)OCC_SMALL; ++j) {
                    const uint2 ge = e[j];
                    const BinRec b = binrec[ge.x];
                    OccCand cc;
                    cc.c0 = b.q0;
                    cc.c1 = make_float4(b.q1.x, rec[ge.x].r1.y, __uint_as_float(ge.y), __uint_as_float(ge.x));
                    cc.rect_lo = __float_as_uint(b.q1.z);
                    cc.rect_hi = __float_as_uint(b.q1.w);
                    cc.pad0 = cc.pad1 = 0u;
                    s_cand[at + j] = cc;
                }
            }
        }
        __syncthreads();
        const uint32_t n = min(s_n, (uint32_t)OCC_SMALL);
        // ... ranks them front to back by (depth bucket, Gaussian id) — lane i of every wave then owns the i-th nearest ...
        if (threadIdx.x < n) {
            const uint32_t kb = occ_bucket(__float_as_uint(s_cand[threadIdx.x].c1.z)), id = __float_as_uint(s_cand[threadIdx.x].c1.w);
            uint32_t r = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t kj = occ_bucket(__float_as_uint(s_cand[j].c1.z)), ij = __float_as_uint(s_cand[j].c1.w);
                r += (kj < kb || (kj == kb && ij < id)) ? 1u : 0u;
            }
            s_rank[r] = (uint8_t)threadIdx.x;
        }
        __syncthreads();
        OccCand mine;
        uint32_t my_bucket = 0;
        if ((uint32_t)lane < n) { mine = s_cand[s_rank[lane]]; my_bucket = occ_bucket(__float_as_uint(mine.c1.z)); }
        // ... and the waves take the tile blocks one each: weight per lane, inclusive prefix over the lanes, the bucket of the
        // lane at which the sum reaches the threshold (= the bucket in which the bucketed sums of the large path cross)
        for (int b = blockIdx.x * 4 + wv; b < n_blocks; b += gridDim.x * 4) {
            const int bx = b % nbx, by = b / nbx;
            const int tx0 = bx * B, ty0 = by * B;
            const int tx1 = min(tx0 + B, vp.gx), ty1 = min(ty0 + B, vp.gy);
            const float x0 = (float)(tx0 * TILE), y0 = (float)(ty0 * TILE);
            const float x1 = (float)(min(tx1 * TILE, vp.W) - 1), y1 = (float)(min(ty1 * TILE, vp.H) - 1);
            uint32_t w = (uint32_t)lane < n ? cover_weight(mine, x0, x1, y0, y1, tx0, tx1, ty0, ty1) : 0u;
            uint32_t inc = w;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += o;
            }
            const uint64_t reached = __ballot(inc >= OCC_THRESHOLD);
            uint32_t q = 0xFFFFu;
            if (reached) {
                const uint32_t kq = (uint32_t)__shfl((int)my_bucket, __ffsll((long long)reached) - 1);
                if (kq < (uint32_t)(OCC_BUCKETS - 1)) q = kq;       // (a crossing in the last bucket closes nothing)
            }
            if (lane == 0) {
                __hip_atomic_store(&occ_cut[b], q, OCC_RLX_AGENT);
                if (q != 0xFFFFu) __hip_atomic_store(&hdr->any_closed, 1u, OCC_RLX_AGENT);
            }
        }
    } else {
        // ---- many candidates: histogram -- barrier -- selection + gather -- barrier -- covers
        for (int q = blockIdx.x * OCC_THREADS + threadIdx.x; q < n_blocks; q += gridDim.x * OCC_THREADS)
            __hip_atomic_store(&occ_cut[q], 0xFFFFu, OCC_RLX_AGENT);
        // (the depth histogram only decides WHICH candidates are kept when there are more than OCC_MAX_CAND: otherwise all are,
        //  and the first phase and its barrier are skipped — BASELINE C5: 10 751 candidates, 13 us)
        // (... and is only skipped while "nothing closes" is plausible — at most 16 candidates per block; a block needs at least
        //  three covers, in practice dozens: BASELINE C5 has 5 per block.  A dense view — the filters-off model at 240 x 135:
        //  2500 per block — takes the histogram and with it the depth-ordered records and the early-stopping cover walk)
        const bool all_kept = total <= (uint32_t)OCC_MAX_CAND && total <= 16u * (uint32_t)n_blocks;
        bool ordered = false;                              // (workgroup-uniform, the same in every workgroup)
        if (!all_kept) {
            for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) occ_hist_chunk(c, heavy_list, heavy_count, n_slots, hist, lds.hist, s_w);
            if (!occ_grid_barrier<false>(hdr, 0, &s_ok)) return;              // (the histogram: atomics only)
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
        }
        {
            OccSelect S;
            if (all_kept) {
                S.total = total; S.limit = (uint32_t)(OCC_BUCKETS - 1); S.stride = 1u; S.keep_total = total; S.by_depth = true;
                S.ordered = false; S.bound = true;
            }
            else S = occ_select(total, hist, lds.g);
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                __hip_atomic_store(&hdr->n_cand, S.by_depth ? (total <= (uint32_t)OCC_MAX_CAND ? total : S.keep_total)
                                                             : (total + S.stride - 1) / S.stride, OCC_RLX_AGENT);
                hdr->depth_limit = S.by_depth ? S.limit : 0xFFFFFFFFu;
            }
            for (int c = blockIdx.x; c < n_chunks; c += gridDim.x)
                occ_gather_chunk(c, S, P, geom, heavy_list, heavy_count, n_slots, hdr, cand, lds.g, hist + OCC_BUCKETS, vp, block_log2, nbx);
            ordered = S.ordered;
        }
        if (!occ_grid_barrier<false>(hdr, 1, &s_ok)) return;              // (the candidate records: write-through stores)
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (all_kept) {
            // the blocks' total weights came with the gather (S.bound): can anything close at all?  Every workgroup reads the
            // same final values and takes the same decision; "no" leaves any_closed = 0 and the table unread
            const uint32_t* tot = hist + OCC_BUCKETS;
            uint32_t mx = 0;
            for (int q = threadIdx.x; q < n_blocks; q += OCC_THREADS) mx = max(mx, __hip_atomic_load(&tot[q], OCC_RLX_AGENT));
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
            if (lane == 0) s_w[8 + wv] = mx;
            __syncthreads();
            mx = max(max(s_w[8], s_w[9]), max(s_w[10], s_w[11]));
            if (__hip_atomic_load(&hdr->pad[0], OCC_RLX_AGENT) == 0u && mx < OCC_THRESHOLD) return;
        }
        {
            const uint32_t n = min(__hip_atomic_load(&hdr->n_cand, OCC_RLX_AGENT), (uint32_t)OCC_MAX_CAND);
            // groups of consecutive blocks: as large as keeps every workgroup busy (1080p, 510 blocks: two; 4K, 2040 blocks: four)
            const int per = max(1, min(OCC_GROUP, (n_blocks + (int)gridDim.x - 1) / (int)gridDim.x));
            const int n_groups = (n_blocks + per - 1) / per;
            for (int g = blockIdx.x; g < n_groups; g += gridDim.x)
                occ_cover_group(g * per, min(per, n_blocks - g * per), vp, B, nbx, n, cand, lds.sums, s_w, s_cross, s_tot, occ_cut, hdr,
                                ordered);
        }
    }

    // ---- did anything close?  One more barrier (the table and the flag were written write-through: no release); if not — the
    // rule — every workgroup leaves
    if (!occ_grid_barrier<false>(hdr, 2, &s_ok)) return;
    if (__hip_atomic_load(&hdr->any_closed, OCC_RLX_AGENT) == 0u) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    occ_table_load(lds.table, occ_cut, nbx, nby);
    const int n_gchunks = (P + OCC_THREADS - 1) / OCC_THREADS;
    for (int c = blockIdx.x; c < n_gchunks; c += gridDim.x) occ_recount_chunk(c, vp, P, geom, block_log2, nbx, lds.table);
}

}  // namespace

int set_occlusion(int on) { return g_occlusion.exchange(on ? 1 : 0); }
int get_occlusion() { return g_occlusion.load(); }
int occlusion_block_log2(int gx, int gy) {
    int lb = 2;                                        // 4 x 4 tiles, larger when the grid has more than OCC_MAX_BLOCKS of them
    while ((int64_t)((gx + (1 << lb) - 1) >> lb) * ((gy + (1 << lb) - 1) >> lb) > OCC_MAX_BLOCKS ||
           ((gy + (1 << lb) - 1) >> lb) > OCC_MAX_BLOCK_ROWS)
        ++lb;
    return lb;
}

hipError_t launch_occlusion(const ViewParams& vp, int P, char* geom, const uint32_t* heavy_list, const uint32_t* heavy_count,
                            const uint32_t* heavy_blk, OccCand* cand, hipStream_t s) {
    if (P == 0) return hipSuccess;
    const int n_slots = 4 * ((P + 255) / 256);
    const int lb = occlusion_block_log2(vp.gx, vp.gy), B = 1 << lb;
    const int nbx = (vp.gx + B - 1) / B, nby = (vp.gy + B - 1) / B;
    hipLaunchKernelGGL(occ_pass_kernel, dim3(OCC_GRID), dim3(OCC_THREADS), 0, s, vp, P, geom, heavy_list, heavy_count, heavy_blk,
                       n_slots, cand, lb, nbx, nby);
    return hipGetLastError();
}

}  // namespace msgs
