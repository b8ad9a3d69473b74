// blend.hip — per-tile front-to-back alpha-composite forward (K6) and back-to-front gradient
// backward (K7) for gfx950.  Replaces renderCUDA forward/backward of the
// reference's un-vendored CUDA module (call site gaussian_renderer/__init__.py:94-108).
//
// Structure (both kernels): one 256-thread workgroup per 16x16 tile; the four wave64s each own one
// 8x8 pixel QUADRANT (lane l -> pixel (l & 7, l >> 3) of the quadrant).  The tile's depth-sorted
// Gaussian list is staged through LDS in batches of 256 records (coalesced 4-B id loads, then three
// 16-B gathers per record).  The loading thread also classifies its record against the four
// quadrants with the exact alpha >= 1/255 level-set test, and every wave compacts the batch to the
// entries that can touch ITS quadrant (64-bit ballot + popcount prefix), so the per-pixel loop only
// visits records that matter for that wave.  Skipping a record whose alpha is < 1/255 on every
// pixel of the quadrant is exactly what the per-pixel `continue` of the reference does, so results
// are unchanged.
//
// Both kernels are INSTRUCTION-ISSUE bound (a SIMD issues about one instruction of any kind per 2 cycles; compares, selects
// and DPP at half rate, v_exp / v_rcp at a quarter: tools/valu_calib.hip), so the per-(pixel, Gaussian)
// math is kept minimal: the record carries the conic pre-scaled into the log2 domain and log2(opacity), so
//     alpha_raw = exp2( A' dx^2 + 2B' dx dy + C' dy^2 + log2 o )
// is six FMA-class ops and one v_exp_f32; the three monomials double as the weights of the conic sums in
// the backward, the mean gradient is accumulated as (sum q dx, sum q dy), the colour recurrence is one
// scalar (see blend_backward_kernel), and every per-Gaussian constant factor (0.5 W, -1/2, 1/o, 2 ln 2,
// the conic in front of the mean sums) is applied once per Gaussian in preprocess_backward_kernel
// instead of once per pixel here.
//
// Backward reduction: per (wave, record) the nine partial sums are reduced across the 64 lanes with a
// hand-scheduled DPP reduce-scatter inside each 16-lane row (row_reduce_scatter9: 8 -> 4 -> 2 -> 1 values per
// lane, the ninth riding on the duplicate lanes), one cross-row all-reduce of the single remaining value, and ONE
// global_atomic_add_f64 instruction from nine lanes into the 80-byte per-Gaussian gradient record of double accumulators
// (the tiles' float32 sums add up exactly, so the default backward is reproducible whatever order the atomics arrive in:
// DESIGN.md 4.2) — one atomic per (quadrant or tile, Gaussian, component) instead of the reference's one per (pixel,
// Gaussian, component).
// (An LDS accumulator committed once per batch was measured slower and removed, profiles/r1_notes.md.)
//
// Roofline: HBM-bound by contract (BASELINE.json); algorithmic bytes K6 = 48*D_trav + 28*N + 8*tiles,
// K7 = 48*D_trav + 20*N + 36*V.
#include "msgs_internal.h"

#include <atomic>
#include <type_traits>

#include <algorithm>

namespace msgs {

namespace {

constexpr int BATCH = 256;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN = 0.0001f;

// Shared by forward and backward so both take bit-identical skip decisions: explicit FMA order.
//   p = A' dx^2 + B2' dx dy + C' dy^2 + lo     (log2 of the unclamped alpha; B2' = 2 Bh', doubled when the record is
// staged into LDS).  The three monomials double as the weights of the backward's conic sums, and the mean gradient is
// accumulated as (sum q dx, sum q dy) — preprocess_backward_kernel applies the per-Gaussian (A', Bh', C') to them —
// so the backward needs no product beyond these.
struct PairEval { float dxx, dxy, dyy, p; };
__device__ __forceinline__ PairEval eval_pair(float A, float B2, float C, float lo, float dx, float dy) {
#pragma clang fp contract(off)   // only the explicit FMAs below; nothing else may be fused differently per kernel
    PairEval e;
    e.dxx = __fmul_rn(dx, dx);
    e.dxy = __fmul_rn(dx, dy);
    e.dyy = __fmul_rn(dy, dy);
    e.p = __fmaf_rn(A, e.dxx, __fmaf_rn(B2, e.dxy, __fmaf_rn(C, e.dyy, lo)));
    return e;
}
__device__ __forceinline__ float4 doubled_w(const float4& r) { return make_float4(r.x, r.y, r.z, r.w + r.w); }
// Bound of the sign test of the exponent (SPEC Q11: an entry whose exponent comes out positive is skipped — a rounding guard, the
// conic being positive definite).  e.p is log2(alpha) = q + log2(opacity) from ONE FMA chain, and q > 0 is tested as p > bound.
// With bound = log2(opacity) itself (rounds 1-5) the sign of q was resolved to one ulp of log2(opacity) only: a cross term of
// +0.6 ulp rounds the chain up and two negative terms of 0.4 ulp each are lost — an entry with q = -1e-8 was skipped a hair from
// the centre of a giant Gaussian (found by the 60 000-configuration sweep, profiles/r5_parity.md 2.3).  The bound sits two to
// four ulps above log2(opacity): roundings of the chain never skip; a genuinely positive exponent (an indefinite conic from a
// degenerate covariance) still does.  One value per staged record, in a slot the walks read anyway.
__device__ __forceinline__ float sign_test_bound(float lo) { return __fmaf_rn(fabsf(lo), 2.4e-7f, lo); }
__device__ __forceinline__ float4 with_bound(const float4& r2, float lo) { return make_float4(r2.x, r2.y, r2.z, sign_test_bound(lo)); }

// quadrant hit mask of one record (bit q: quadrant q = qx + 2*qy of the tile at (tx0, ty0))
__device__ __forceinline__ uint32_t quadrant_mask(const float4& r0, float C, float tau2, float tx0, float ty0) {
    if (!(tau2 > -1.0e38f)) return 0xFu;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = tx0 + (float)((q & 1) * 8), y0 = ty0 + (float)((q >> 1) * 8);
        if (levelset_hits_rect(r0.x, r0.y, r0.z, r0.w, C, tau2, x0, x0 + 7.0f, y0, y0 + 7.0f)) m |= 1u << q;
    }
    return m;
}

// XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md).  Give
// every XCD a contiguous run of tiles so neighbouring tiles (which share Gaussians) share an L2.
__device__ __forceinline__ int swizzled_tile(int bid, int num_tiles) {
    const int per = num_tiles >> 3;            // tiles per XCD in the divisible part
    const int main = per << 3;
    if (bid >= main) return bid;               // ragged tail: identity
    return (bid & 7) * per + (bid >> 3);
}

// Side job of the forward blend kernels: clear the backward's per-Gaussian gradient records (msgs.h, grad_records).  They
// have to start from zero; done here the 80 bytes per Gaussian cost nothing — the kernels are instruction-issue bound with
// the memory pipe mostly idle, the stores are fire-and-forget and nothing in this kernel reads them — and the backward saves a
// fill launch.  Workgroup b clears slice b of the buffer (n16 sixteen-byte words in total).
constexpr int CLEAR_INLINE_MIN_TILES = 2048;
__device__ __forceinline__ void clear_slice(uint4* __restrict__ p, size_t n16) {
    if (p == nullptr) return;
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x;
    const size_t hi = lo + per < n16 ? lo + per : n16;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Per-tile traversal length for the backward's launch order (ImageLayout::tile_last): the position behind the last entry any
// pixel of the tile blended = the number of list entries the backward will walk for this tile — its work, to first order.
// Also invalidates the previous frame's launch order (launch_tile_order re-validates it after this kernel).
// The stored key estimates the backward's instruction count for the tile (/ 16): ~52 per traversed entry (record fetch, loop
// control, the 64-lane reduction, the atomics) + ~36 per (quadrant, entry) evaluation, the latter counted as the entries the
// forward's waves walked (`walked`, wave-uniform).
__device__ __forceinline__ void note_tile_last(uint32_t* s_wlast, int nwaves, uint32_t last, uint32_t walked, int tile, int w,
                                               int lane, uint32_t* __restrict__ tile_last,
                                               uint32_t* __restrict__ order_flag, unsigned long long* __restrict__ dtrav = nullptr) {
    if (tile_last == nullptr) return;
    const uint32_t wl = wave_max_u32(last);
    if (lane == 0) { s_wlast[w] = wl; s_wlast[nwaves + w] = walked; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0, q = 0;
        for (int k = 0; k < nwaves; ++k) { m = max(m, s_wlast[k]); q += s_wlast[nwaves + k]; }
        tile_last[tile] = (52u * m + 36u * q) >> 4;
        // entries of this tile's list that are traversed (sum over the tiles = D_trav), into one of DTRAV_SLOTS accumulators
        if (dtrav) atomicAdd(&dtrav[tile & (DTRAV_SLOTS - 1)], (unsigned long long)m);
        if (blockIdx.x == 0 && order_flag) *order_flag = 0u;
    }
}

// Per-pixel forward state and the walk over one wave's compacted entry list — shared by the quadrant-per-wave kernel
// and the fine-grained (4x4 sub-block per wave) kernel, so that both evaluate every pixel with the same instructions in
// the same order (bit-identical images, test_blend_granularities_agree).
struct FwdPix {
    float T, C0, C1, C2, aps, adp;
};
// depth-slab binning (msgs_view_t.slab_fraction) and feedback: what blend_forward_kernel does besides blending
struct FwdSlab {
    uint32_t* open_bits;          // slab A: publish the tiles that are still open here (bitmap) ...
    uint32_t* open_list;          // ... and here (list, any order)
    uint32_t* n_open;             // ... counted here
    const uint32_t* tile_list;    // slab B: blend only these tiles
    const uint32_t* n_tiles;      // ... this many
    unsigned long long* dtrav;    // D_trav accumulators (feedback; nullable)
};
// lp[0..cnt): BYTE offsets of the batch's 16-byte records this wave has to evaluate, in list order.  `alive` = lanes still
// blending, as a SCALAR mask: every predicate is the ballot of one direct comparison combined with scalar logic (a ballot
// of a derived bool costs two VALU instructions per use), and per-lane selects take their condition from the mask.
// Returns the byte offset of the last entry blended in this batch (0xFFFFFFFF: none).
// COUNT (diagnostic replica, msgs_blend_lane_stats): also counts, per wave, the entry evaluations (x 64 = evaluated lanes), the
// lanes still blending at each of them and the lanes that blended — scalar popcounts, no effect on the arithmetic.
struct LaneStats { uint32_t steps, alive, blended; };
// Entry lists of the forward kernels: BYTE offsets of the 16-byte records, four bytes each so that ONE 16-byte LDS read
// fetches four of them, and padded to a multiple of four with the offset of a SENTINEL record (slot BATCH of the record
// arrays: zero conic, log2-opacity -1e30 -> alpha = exp2(-1e30) = 0 < 1/255, never valid, every product with it is an exact
// zero).  The walk then takes four entries per trip with one list read, one "any pixel left?" test and one counter update for
// the four — 36 instead of 45 instructions per entry; a padded or post-termination evaluation changes no pixel.
// Measured at C3 (blend_forward_kernel incl. the clearing of the gradient records): two entries per trip 187 us; four per trip
// 216 us when the compiler is free to hoist all thirteen LDS reads of a trip (97 VGPRs -> 4 waves per SIMD) and 182 us
// when it is held to 64 VGPRs / 8 waves per SIMD (amdgpu_waves_per_eu below): issue-bound code wants the waves, not the
// hoisting.  (Folding the two selects of the update into one — alpha_b = blend ? alpha : 0, T = fma(-T, alpha_b, T) —
// measured the same time and was not kept; eight entries per trip: 184 us, no gain.)
constexpr int LIST_PAD = 4;
constexpr uint32_t SENTINEL_OFF = (uint32_t)BATCH << 4;

__device__ __forceinline__ void write_sentinel_record(float4* s_r0, float4* s_r1, float4* s_r2) {
    s_r0[BATCH] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_r1[BATCH] = make_float4(0.f, -1.0e30f, 0.f, 0.f);
    s_r2[BATCH] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// after the compaction: lanes 0..2 of the wave pad its list (cnt entries) up to the next multiple of four
__device__ __forceinline__ void pad_list(uint32_t* lp, int cnt, int lane) {
    if (lane < LIST_PAD - 1) lp[cnt + lane] = SENTINEL_OFF;
}

template <bool COUNT = false>
__device__ __forceinline__ uint32_t forward_walk(const uint32_t* lp, int cnt, const float4* s_r0, const float4* s_r1,
                                                 const float4* s_r2, float pxf, float pyf, FwdPix& st, uint64_t& alive_io,
                                                 uint32_t& walked, LaneStats* stats = nullptr) {
    // the state lives in LOCAL scalars while the list is walked (through the struct reference the compiler turned the
    // two selects of the update into EXEC-masked moves with duplicated loop-carried copies: +14 % VALU instructions)
    float T = st.T, C0 = st.C0, C1 = st.C1, C2 = st.C2, aps = st.aps, adp = st.adp;
    uint64_t alive = alive_io;
    uint32_t last_off = 0xFFFFFFFFu;
    // (a scalar entry index via readfirstlane was measured: 221 -> 257 us — the VALU->SALU->VALU hop in front of
    //  the LDS reads costs more than the two address instructions it saves)
    auto blend_entry = [&](uint32_t off) {
        const float4 r0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_r0) + off);
        const float4 r1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_r1) + off);
        const float4 r2 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_r2) + off);
        const float dx = r0.x - pxf, dy = r0.y - pyf;
        const PairEval ev = eval_pair(r0.z, r0.w, r1.x, r1.y, dx, dy);
        const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(ev.p));
        const float test_T = __fmaf_rn(-T, alpha, T);
        const uint64_t m_pow = __builtin_amdgcn_ballot_w64(ev.p <= r2.w);          // power <= 0 (r2.w: sign_test_bound)
        const uint64_t m_alpha = __builtin_amdgcn_ballot_w64(alpha >= ALPHA_MIN);   // alpha >= 1/255
        const uint64_t m_stop = __builtin_amdgcn_ballot_w64(test_T < T_MIN);
        const uint64_t validm = alive & m_pow & m_alpha;
        const uint64_t stopm = validm & m_stop;
        if (COUNT) {
            // (a trip on padding evaluates the sentinel record and can never blend: not counted)
            if (off != SENTINEL_OFF) {
                stats->steps += 1u; stats->alive += (uint32_t)__popcll(alive); stats->blended += (uint32_t)__popcll(validm & ~stopm);
            }
        }
        alive &= ~stopm;                                                            // terminated: NOT blended (Q7)
        const bool blend = __builtin_amdgcn_inverse_ballot_w64(validm & ~stopm);
        const float wgt = blend ? alpha * T : 0.0f;
        C0 = fmaf(r1.z, wgt, C0); C1 = fmaf(r1.w, wgt, C1); C2 = fmaf(r2.x, wgt, C2);
        adp = fmaf(r2.y, wgt, adp); aps = fmaf(r2.z, wgt, aps);
        T = blend ? test_T : T;
        last_off = blend ? off : last_off;
    };
    int j = 0;
    for (; j < cnt; j += LIST_PAD) {                       // the list is padded to a multiple of four
        if (alive == 0) break;
        const uint4 o = *reinterpret_cast<const uint4*>(lp + j);
        blend_entry(o.x); blend_entry(o.y); blend_entry(o.z); blend_entry(o.w);
    }
    walked += (uint32_t)min(j, cnt);                       // wave-uniform: entries this wave evaluated (backward work estimate)
    st.T = T; st.C0 = C0; st.C1 = C1; st.C2 = C2; st.aps = aps; st.adp = adp;
    alive_io = alive;
    return last_off;
}

__device__ __forceinline__ void forward_store(const FwdPix& st, uint32_t last, bool inside, int px, int py,
                                              const ViewParams& vp, float* out_color, float* out_ps, float* out_depth,
                                              float* final_T, uint32_t* n_contrib) {
    if (inside) {
        const size_t N = (size_t)vp.W * vp.H;
        const size_t pix = (size_t)py * vp.W + px;
        out_color[pix] = st.C0 + st.T * vp.bg[0];
        out_color[N + pix] = st.C1 + st.T * vp.bg[1];
        out_color[2 * N + pix] = st.C2 + st.T * vp.bg[2];
        out_ps[pix] = st.aps;
        out_depth[pix] = st.adp;
        final_T[pix] = st.T;
        n_contrib[pix] = last;
    }
}

// ---------------------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------------------
// (the kernel: blend_forward_kernel below — one call per workgroup, or, for slab B, one call per listed tile)
template <bool COUNT>
__device__ __forceinline__ void blend_forward_tile(int tile, const ViewParams& vp, const GaussRec* __restrict__ rec,
                                                            const uint32_t* __restrict__ ids,
                                                            const uint2* __restrict__ ranges,
                                                            float* __restrict__ out_color,
                                                            float* __restrict__ out_ps,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib,
                                                            unsigned long long* __restrict__ lane_stats,
                                                            uint32_t* __restrict__ tile_last,
                                                            uint32_t* __restrict__ order_flag, const FwdSlab& sb,
                                                            float4* s_r0, float4* s_r1, float4* s_r2, uint32_t* s_mask,
                                                            uint32_t* s_wlast, uint32_t (*s_list)[BATCH + LIST_PAD]) {
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < vp.W && py < vp.H;
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    FwdPix st = {1.0f, 0.f, 0.f, 0.f, 0.f, 0.f};
    LaneStats ls = {0u, 0u, 0u};
    uint32_t last = 0, walked = 0;
    // lanes still blending, as a SCALAR mask: every predicate below is the ballot of one direct comparison combined with
    // scalar logic (a ballot of a derived bool costs two VALU instructions per use; these kernels' time is their VALU
    // instruction count), and per-lane selects take their condition from the mask for free
    uint64_t alive = __builtin_amdgcn_ballot_w64(inside);

    for (int base = 0; base < len; base += BATCH) {
        if (__syncthreads_and(alive == 0)) break;    // barrier also protects the LDS batch
        const int n = min(BATCH, len - base);
        if (tid < n) {
            const uint32_t id = ids[range.x + base + tid];
            const float4 r0 = rec[id].r0, r1 = rec[id].r1, r2 = rec[id].r2;
            s_r0[tid] = doubled_w(r0); s_r1[tid] = r1; s_r2[tid] = with_bound(r2, r1.y);
            s_mask[tid] = quadrant_mask(r0, r1.x, r2.w, tx0, ty0);
        }
        __syncthreads();
        // per-wave compaction of the batch to this wave's quadrant; the list holds BYTE offsets of the 16-byte records
        // (one shift less per pair evaluation)
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < BATCH / 64; ++c) {
            const int e = c * 64 + lane;
            const bool hit = e < n && ((s_mask[e] >> w) & 1u);
            const uint64_t b = __ballot(hit);
            if (hit) s_list[w][cnt + __popcll(b & lt_mask)] = (uint32_t)(e << 4);
            cnt += __popcll(b);
        }
        pad_list(s_list[w], cnt, lane);
        const uint32_t last_off = forward_walk<COUNT>(s_list[w], cnt, s_r0, s_r1, s_r2, pxf, pyf, st, alive, walked, &ls);
        if (last_off != 0xFFFFFFFFu) last = (uint32_t)base + (last_off >> 4) + 1u;   // once per batch, not per pair
    }
    if (COUNT) {
        if (lane == 0) {
            atomicAdd(&lane_stats[0], (unsigned long long)ls.steps);
            atomicAdd(&lane_stats[1], (unsigned long long)ls.alive);
            atomicAdd(&lane_stats[2], (unsigned long long)ls.blended);
        }
    } else {
        forward_store(st, last, inside, px, py, vp, out_color, out_ps, out_depth, final_T, n_contrib);
        // slab A: a pixel that is still blending at the end of this list may blend entries of slab B — the tile stays OPEN and
        // is blended again over its complete list (which then also counts its traversed entries).  Where every pixel has
        // terminated (Q7) nothing behind the list is ever evaluated: the tile is done, bit for bit.
        const bool open = sb.open_bits != nullptr && __syncthreads_or(alive != 0);
        note_tile_last(s_wlast, 4, inside ? last : 0u, walked, tile, w, lane, tile_last, order_flag, open ? nullptr : sb.dtrav);
        if (open && threadIdx.x == 0) {
            atomicOr(&sb.open_bits[tile >> 5], 1u << (tile & 31));
            sb.open_list[atomicAdd(sb.n_open, 1u)] = (uint32_t)tile;
        }
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void blend_forward_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                            const uint32_t* __restrict__ ids,
                                                            const uint2* __restrict__ ranges,
                                                            float* __restrict__ out_color,
                                                            float* __restrict__ out_ps,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib,
                                                            unsigned long long* __restrict__ lane_stats,
                                                            uint4* __restrict__ clear_ptr, size_t clear_n16,
                                                            uint32_t* __restrict__ tile_last,
                                                            uint32_t* __restrict__ order_flag, FwdSlab sb) {
    __shared__ float4 s_r0[BATCH + 1], s_r1[BATCH + 1], s_r2[BATCH + 1];     // slot BATCH: the sentinel record
    __shared__ uint32_t s_mask[BATCH];
    __shared__ uint32_t s_wlast[8];
    __shared__ __attribute__((aligned(16))) uint32_t s_list[4][BATCH + LIST_PAD];
    clear_slice(clear_ptr, clear_n16);
    if (threadIdx.x == 0) write_sentinel_record(s_r0, s_r1, s_r2);          // ordered by the first barrier of the batch loop
    if (sb.tile_list == nullptr) {
        blend_forward_tile<COUNT>(swizzled_tile(blockIdx.x, vp.gx * vp.gy), vp, rec, ids, ranges, out_color, out_ps, out_depth, final_T,
                                  n_contrib, lane_stats, tile_last, order_flag, sb, s_r0, s_r1, s_r2, s_mask, s_wlast, s_list);
        return;
    }
    // slab B: the tiles slab A left open, in the order they were listed; a small grid walks the list (normally a handful of
    // tiles: a grid of one workgroup per tile of the image would spend 25 us at 4K on workgroups that only learn that)
    const uint32_t n = *sb.n_tiles;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        blend_forward_tile<COUNT>((int)sb.tile_list[k], vp, rec, ids, ranges, out_color, out_ps, out_depth, final_T, n_contrib,
                                  lane_stats, tile_last, order_flag, sb, s_r0, s_r1, s_r2, s_mask, s_wlast, s_list);
        __syncthreads();            // the staged batch and the per-wave slots are reused by the next tile
    }
}

// Feedback publication (msgs_view_t.feedback_tag): one wave sums the D_trav accumulators and writes
// {D_trav, tag | n_open, DA | DB, D | ticket} into words 4..7 of the forward's pinned status block (the last word last).
__global__ __launch_bounds__(64) void forward_feedback_kernel(const unsigned long long* __restrict__ dtrav,
                                                              const SlabHeader* __restrict__ hdr, int slab_mode,
                                                              int64_t D_host, const uint32_t* __restrict__ D_dev, uint32_t tag,
                                                              uint64_t ticket, volatile uint64_t* __restrict__ host) {
    static_assert(DTRAV_SLOTS == 64, "one accumulator per lane");
    unsigned long long acc = dtrav[threadIdx.x];
    for (int off = 32; off > 0; off >>= 1)
        acc += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(acc >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)acc, off);
    if (threadIdx.x == 0) {
        const uint64_t D = D_dev ? (uint64_t)*D_dev : (uint64_t)D_host;
        const bool overflow = slab_mode && hdr->pad0 != 0u;
        host[4] = (uint64_t)acc | (overflow ? (1ull << 63) : 0ull);
        host[5] = (uint64_t)tag | ((uint64_t)(slab_mode ? hdr->n_open : 0xFFFFFFFFu) << 32);
        host[6] = slab_mode ? ((uint64_t)hdr->DA | ((uint64_t)hdr->DB << 32)) : 0ull;
        __threadfence_system();
        host[7] = (D & 0xFFFFFFFFFFull) | ((ticket & 0xFFFFFFull) << 40);
    }
}

// Fine-grained variant for FEW tiles (low pyramid levels): one wave64 per 4x4 pixel sub-block (lanes 0..15) — sixteen per tile —
// in workgroups of WAVES waves, G = 16 / WAVES workgroups per tile.  With fewer than ~300 tiles the quadrant kernel is latency-bound
// — one wave per SIMD at best, and a pixel's list is inherently sequential — so its time is the length of the longest per-wave entry
// chain; a 4x4 block with its own exact hit list shortens that chain (an entry reaches it far less often than an 8x8 quadrant) at
// the price of idle lanes, which are free in that regime.  Round 3: (i) the sixteen waves of a tile used to share ONE workgroup,
// i.e. one CU did a whole tile's work while at 135 / 40 / 12 / 2 tiles most CUs had none; the launch now splits a tile over G
// workgroups (every one stages the tile's list for itself: redundant reads of a few MB against idle CUs) so that tiles x G fills
// the chip; (ii) every wave classifies the batch against its OWN block and compacts in the same pass (lanes = records) — no mask
// array, one barrier less per batch; (iii) the next batch's records travel in registers while the current one is walked.
// Same arithmetic per pixel in the same order: results are bit-identical to blend_forward_kernel.
// SB = side of the sub-block a wave owns: 4 (lanes 0..15, sixteen sub-blocks per tile) or 2 (lanes 0..3, sixty-four per tile — the
// entry chain of a wave is what bounds this regime, and a 2x2 block is reached by fewer entries still; the list is staged and
// classified per workgroup as before, so the waves per workgroup grow with the sub-block count).
template <int WAVES, int SB = 4>
__global__ __launch_bounds__(64 * WAVES) void blend_forward_fine_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                            const uint32_t* __restrict__ ids,
                                                            const uint2* __restrict__ ranges,
                                                            float* __restrict__ out_color,
                                                            float* __restrict__ out_ps,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib,
                                                            uint4* __restrict__ clear_ptr, size_t clear_n16,
                                                            uint32_t* __restrict__ order_flag) {
    constexpr int PER_ROW = TILE / SB, NSB = PER_ROW * PER_ROW, LANES = SB * SB;
    constexpr int T = 64 * WAVES, G = NSB / WAVES;
    constexpr int R = (BATCH + T - 1) / T;               // records a thread stages per batch
    __shared__ float4 s_r0[BATCH + 1], s_r1[BATCH + 1], s_r2[BATCH + 1];     // slot BATCH: the sentinel record
    __shared__ float s_tau[BATCH];                                           // (s_r2.w carries the sign-test bound)
    __shared__ __attribute__((aligned(16))) uint32_t s_list[WAVES][BATCH + LIST_PAD];
    clear_slice(clear_ptr, clear_n16);
    if (threadIdx.x == 0) {
        write_sentinel_record(s_r0, s_r1, s_r2);          // ordered by the first barrier of the batch loop
        if (blockIdx.x == 0 && order_flag) *order_flag = 0u;      // (no backward launch order in this regime)
    }
    const int tile = blockIdx.x / G, grp = blockIdx.x % G;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int sb = grp * WAVES + w;                       // this wave's SB x SB sub-block of the tile
    const int px = tx * TILE + (sb % PER_ROW) * SB + (lane % SB);
    const int py = ty * TILE + (sb / PER_ROW) * SB + ((lane / SB) % SB);
    const bool inside = lane < LANES && px < vp.W && py < vp.H;  // the other lanes idle: this variant buys latency, not throughput
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)(tx * TILE + (sb % PER_ROW) * SB), by0 = (float)(ty * TILE + (sb / PER_ROW) * SB);
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    FwdPix st = {1.0f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t last = 0, walked = 0;
    uint64_t alive = __builtin_amdgcn_ballot_w64(inside);

    float4 p0[R], p1[R], p2[R];
    auto fetch = [&](int base) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = tid + k * T;
            if (e < BATCH && base + e < len) {
                const uint32_t id = ids[range.x + base + e];
                p0[k] = rec[id].r0; p1[k] = rec[id].r1; p2[k] = rec[id].r2;
            }
        }
    };
    fetch(0);
    for (int base = 0; base < len; base += BATCH) {
        if (__syncthreads_and(alive == 0)) break;    // barrier also protects the LDS batch
        const int n = min(BATCH, len - base);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = tid + k * T;
            if (e < n) { s_r0[e] = doubled_w(p0[k]); s_r1[e] = p1[k]; s_r2[e] = with_bound(p2[k], p1[k].y); s_tau[e] = p2[k].w; }
        }
        fetch(base + BATCH);
        __syncthreads();
        // classification against this wave's block and compaction in one pass; the list holds BYTE offsets of the records
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < BATCH / 64; ++c) {
            const int e = c * 64 + lane;
            bool hit = false;
            if (e < n) {
                const float4 r0 = s_r0[e];
                const float C = s_r1[e].x, tau2 = s_tau[e];
                hit = !(tau2 > -1.0e38f) ||
                      levelset_hits_rect(r0.x, r0.y, r0.z, 0.5f * r0.w, C, tau2, bx0, bx0 + (float)(SB - 1), by0,
                                         by0 + (float)(SB - 1));
            }
            const uint64_t b = __ballot(hit);
            if (hit) s_list[w][cnt + __popcll(b & lt_mask)] = (uint32_t)(e << 4);
            cnt += __popcll(b);
        }
        pad_list(s_list[w], cnt, lane);
        const uint32_t last_off = forward_walk(s_list[w], cnt, s_r0, s_r1, s_r2, pxf, pyf, st, alive, walked);
        if (last_off != 0xFFFFFFFFu) last = (uint32_t)base + (last_off >> 4) + 1u;   // once per batch, not per pair
    }
    forward_store(st, last, inside, px, py, vp, out_color, out_ps, out_depth, final_T, n_contrib);
}

// ---------------------------------------------------------------------------------------------
// K7
// ---------------------------------------------------------------------------------------------
// 64 lanes x 9 partial sums -> ONE value per lane holding a 16-lane-row total, hand-scheduled DPP (the backward's
// time is its VALU instruction count; this is 21 instructions against 34 for selects + quad_perm exchanges):
//   * v0..v7 are reduce-scattered 8 -> 4 -> 2 -> 1 values per lane: across the half rows (row_ror:8) and across
//     neighbouring quads (row_half_mirror) the "keep mine / add the partner's" choice is the DPP bank mask — a bank is
//     a quad of lanes, so a second, bank-masked add overwrites the half of the lanes that keep the other operand and no
//     select is needed; only the last level (lane parity inside a quad) takes the two selects.
//   * v8 is all-reduced inside the row (4 adds) and rides on the lanes whose bit 1 is set, which would otherwise hold a
//     duplicate of their lane ^ 2 neighbour.
// Result: lane l holds the row total of component comp(l) = 4*((l>>3)&1) + 2*((l>>2)&1) + (l&1) when (l & 2) == 0, and
// the row total of component 8 when (l & 2) != 0.  The caller adds the four rows.
// All hazards (VALU write -> DPP read needs two wait states) are resolved by the instruction order inside the block;
// the leading s_nop covers whatever the compiler scheduled right before it.
template <class A> struct BwdSumsT { A v0, v1, v2, v3, v4, v5, v6, v7, v8; };
using BwdSums = BwdSumsT<float>;
__device__ __forceinline__ float row_reduce_scatter9(const BwdSums& v) {
    float a0, a1, a2, a3, b0, b1, t8, keep, send, c;
    const uint64_t odd = 0xAAAAAAAAAAAAAAAAull, bit1 = 0xCCCCCCCCCCCCCCCCull;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %[a0], %[v0], %[v0] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a1], %[v1], %[v1] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a2], %[v2], %[v2] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a3], %[v3], %[v3] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a0], %[v4], %[v4] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a1], %[v5], %[v5] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a2], %[v6], %[v6] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a3], %[v7], %[v7] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[t8], %[v8], %[v8] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[b0], %[a0], %[a0] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[b1], %[a1], %[a1] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[b0], %[a2], %[a2] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[b1], %[a3], %[a3] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[t8], %[t8], %[t8] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[send], %[b1], %[b0], %[odd]\n\t"
        "v_cndmask_b32_e64 %[keep], %[b0], %[b1], %[odd]\n\t"
        "v_add_f32_dpp %[t8], %[t8], %[t8] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[c], %[send], %[keep] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[t8], %[t8], %[t8] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[c], %[c], %[c] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[c], %[c], %[t8], %[bit1]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1),
          [t8] "=&v"(t8), [keep] "=&v"(keep), [send] "=&v"(send), [c] "=&v"(c)
        : [v0] "v"(v.v0), [v1] "v"(v.v1), [v2] "v"(v.v2), [v3] "v"(v.v3), [v4] "v"(v.v4), [v5] "v"(v.v5),
          [v6] "v"(v.v6), [v7] "v"(v.v7), [v8] "v"(v.v8), [odd] "s"(odd), [bit1] "s"(bit1));
    return c;
}
// component (0..8) that lane l of a row delivers after row_reduce_scatter9
__device__ __forceinline__ uint32_t row_reduce_component(int l) {
    return (l & 2) ? 8u : (uint32_t)(4 * ((l >> 3) & 1) + 2 * ((l >> 2) & 1) + (l & 1));
}
// Per-pixel backward state and the walk over one wave's compacted entry list (back to front) — shared by the
// four-waves-per-tile kernel (CROSS_ROW: the pixels of an 8x8 quadrant fill the wave, the four 16-lane rows are added
// with v_permlane16/32_swap) and the fine-grained kernel (the sixteen pixels of a 4x4 sub-block live in row 0: the row
// reduction is the whole reduction).  Same arithmetic per pixel in the same order in both.
struct BwdPix {
    float T, S, dL0, dL1, dL2;
    uint32_t last;
};
template <bool CROSS_ROW>
__device__ __forceinline__ void backward_walk(const uint16_t* lp, int cnt, int base, const float4* s_r0, const float4* s_r1,
                                              const float2* s_b, const uint32_t* s_id, float pxf, float pyf, BwdPix& st,
                                              bool alane, uint32_t aoff, grad_acc_t* __restrict__ grad_rec) {
    for (int j = cnt - 1; j >= 0; --j) {
        const int e = lp[j];
        const float4 r0 = s_r0[e], r1 = s_r1[e];
        const float2 bl = s_b[e];                              // {blue, sign_test_bound}
        const float cb = bl.x;
        const float dx = r0.x - pxf, dy = r0.y - pyf;
        const PairEval ev = eval_pair(r0.z, r0.w, r1.x, r1.y, dx, dy);
        const float a_raw = __builtin_amdgcn_exp2f(ev.p);
        const uint64_t validm = __builtin_amdgcn_ballot_w64((uint32_t)(base + e) < st.last) &
                                __builtin_amdgcn_ballot_w64(ev.p <= bl.y) &
                                __builtin_amdgcn_ballot_w64(a_raw >= ALPHA_MIN);   // <=> min(0.99, a_raw) >= 1/255
        if (validm == 0) continue;
        const bool valid = __builtin_amdgcn_inverse_ballot_w64(validm);
        const float a_m = valid ? a_raw : 0.0f;                // the one select: a masked lane is the identity below
        const float alpha_m = fminf(0.99f, a_m);
        const float inv = __builtin_amdgcn_rcpf(1.0f - alpha_m);
        const float Tn = st.T * inv;
        st.T = Tn;
        const float dch = alpha_m * Tn;
        const float sm = fmaf(cb, st.dL2, fmaf(r1.w, st.dL1, r1.z * st.dL0)) - st.S;
        const float dL_dalpha = sm * Tn;
        st.S = fmaf(alpha_m, sm, st.S);
        const float q = a_m * dL_dalpha;                       // Q6: gradient passes the 0.99 clamp
        const BwdSums v = {q * dx, q * dy, q * ev.dxx, q * ev.dxy, q * ev.dyy, q, dch * st.dL0, dch * st.dL1, dch * st.dL2};
        // rows by DPP; the four rows with v_permlane16/32_swap (in this latency-bound regime they beat ds_bpermute,
        // profiles/r1_notes.md); scalar record address, one atomic instruction from nine lanes
        const float rowv = row_reduce_scatter9(v);
        const float outv = CROSS_ROW ? cross_row_allreduce(rowv) : rowv;
        const uint32_t gid = __builtin_amdgcn_readfirstlane(s_id[e]);
        grad_acc_t* gdst = grad_rec + (size_t)gid * GRAD_REC_FLOATS;
        if (alane) unsafeAtomicAdd(gdst + aoff, (grad_acc_t)outv);
    }
}

// Gradient record components accumulated here (scaled to dL/d{mean2D, conic, opacity} per Gaussian
// in preprocess_backward_kernel):
//   [0] sum q dx  [1] sum q dy  [2] sum q dx dx   [3] sum q dx dy   [4] sum q dy dy   [5] sum q
//   [6..8] sum alpha T dL/dC_c                     with q = alpha_raw * dL/dalpha
__global__ __launch_bounds__(256) void blend_backward_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                             const uint32_t* __restrict__ ids,
                                                             const uint2* __restrict__ ranges,
                                                             const float* __restrict__ final_T,
                                                             const uint32_t* __restrict__ n_contrib,
                                                             const float* __restrict__ dL_dcolor,
                                                             grad_acc_t* __restrict__ grad_rec) {
    __shared__ float4 s_r0[BATCH], s_r1[BATCH];
    __shared__ float2 s_b[BATCH];                          // {blue, sign_test_bound}
    __shared__ uint32_t s_id[BATCH];
    __shared__ uint32_t s_mask[BATCH];
    __shared__ uint16_t s_list[4][BATCH];
    __shared__ uint32_t s_wmax[4];

    const int num_tiles = vp.gx * vp.gy;
    const int tile = swizzled_tile(blockIdx.x, num_tiles);
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int px = tx * TILE + (w & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < vp.W && py < vp.H;
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
    const uint2 range = ranges[tile];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const size_t N = (size_t)vp.W * vp.H;
    const size_t pix = (size_t)py * vp.W + px;

    BwdPix st;
    st.T = inside ? final_T[pix] : 1.0f;
    st.last = inside ? n_contrib[pix] : 0u;
    st.dL0 = st.dL1 = st.dL2 = 0.f;
    if (inside) { st.dL0 = dL_dcolor[pix]; st.dL1 = dL_dcolor[N + pix]; st.dL2 = dL_dcolor[2 * N + pix]; }

    const uint32_t wave_last = wave_max_u32(st.last);
    if (lane == 0) s_wmax[w] = wave_last;
    __syncthreads();
    const uint32_t tile_last = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    // S = sum_c dL/dC_c * (colour composited BEHIND the current entry, background included, normalised by the
    // transmittance in front of it).  dL/dalpha_i = T_i (g_i - S_i) with g_i = sum_c dL/dC_c colour_i,c, and
    // S_{i-1} = S_i + alpha_i (g_i - S_i): the three per-channel recurrences of the textbook form collapse into one
    // scalar, and starting it at bg . dL/dC absorbs the separate background term (-T_final bg.dL / (1 - alpha_i)).
    st.S = vp.bg[0] * st.dL0 + vp.bg[1] * st.dL1 + vp.bg[2] * st.dL2;
    const bool alane = lane < 16 && (!(lane & 2) || lane == 2);        // the nine lanes that issue the per-entry atomics
    const uint32_t aoff = row_reduce_component(lane);

    const int nb = ((int)tile_last + BATCH - 1) / BATCH;
    for (int b = nb - 1; b >= 0; --b) {
        __syncthreads();                              // previous batch fully consumed (and flushed)
        const int base = b * BATCH;
        const int n = min(BATCH, (int)tile_last - base);
        if (tid < n) {
            const uint32_t id = ids[range.x + base + tid];
            const float4 r0 = rec[id].r0, r1 = rec[id].r1;
            const float4 r2 = rec[id].r2;
            s_r0[tid] = doubled_w(r0); s_r1[tid] = r1; s_b[tid] = make_float2(r2.x, sign_test_bound(r1.y)); s_id[tid] = id;
            s_mask[tid] = quadrant_mask(r0, r1.x, r2.w, tx0, ty0);
        }
        __syncthreads();
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < BATCH / 64; ++c) {
            const int e = c * 64 + lane;
            const bool hit = e < n && (uint32_t)(base + e) < wave_last && ((s_mask[e] >> w) & 1u);
            const uint64_t bal = __ballot(hit);
            if (hit) s_list[w][cnt + __popcll(bal & lt_mask)] = (uint16_t)e;
            cnt += __popcll(bal);
        }
        backward_walk<true>(s_list[w], cnt, base, s_r0, s_r1, s_b, s_id, pxf, pyf, st, alane, aoff, grad_rec);
    }
}

// Fine-grained variant for FEW tiles (low pyramid levels), the counterpart of blend_forward_fine_kernel: one wave64 per 4x4 pixel
// sub-block (lanes 0..15), WAVES per workgroup, 16 / WAVES workgroups per tile; same per-pixel arithmetic in the same order as
// blend_backward_kernel; one atomic per (sub-block, Gaussian, component).  A workgroup walks the tile's list back from the last
// entry ITS pixels blended.
template <int WAVES, int SB = 4>
__global__ __launch_bounds__(64 * WAVES) void blend_backward_fine_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                             const uint32_t* __restrict__ ids,
                                                             const uint2* __restrict__ ranges,
                                                             const float* __restrict__ final_T,
                                                             const uint32_t* __restrict__ n_contrib,
                                                             const float* __restrict__ dL_dcolor,
                                                             grad_acc_t* __restrict__ grad_rec) {
    constexpr int PER_ROW = TILE / SB, NSB = PER_ROW * PER_ROW, LANES = SB * SB;
    constexpr int T = 64 * WAVES, G = NSB / WAVES;
    constexpr int R = (BATCH + T - 1) / T;               // records a thread stages per batch
    __shared__ float4 s_r0[BATCH], s_r1[BATCH];
    __shared__ float2 s_b[BATCH];                          // {blue, sign_test_bound}
    __shared__ float s_tau[BATCH];
    __shared__ uint32_t s_id[BATCH];
    __shared__ uint16_t s_list[WAVES][BATCH];
    __shared__ uint32_t s_wmax[WAVES];

    const int tile = blockIdx.x / G, grp = blockIdx.x % G;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int sb = grp * WAVES + w;                       // this wave's SB x SB sub-block of the tile
    const int px = tx * TILE + (sb % PER_ROW) * SB + (lane % SB);
    const int py = ty * TILE + (sb / PER_ROW) * SB + ((lane / SB) % SB);
    const bool inside = lane < LANES && px < vp.W && py < vp.H;  // the other lanes idle (see blend_forward_fine_kernel)
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)(tx * TILE + (sb % PER_ROW) * SB), by0 = (float)(ty * TILE + (sb / PER_ROW) * SB);
    const uint2 range = ranges[tile];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const size_t N = (size_t)vp.W * vp.H;
    const size_t pix = (size_t)py * vp.W + px;

    BwdPix st;
    st.T = inside ? final_T[pix] : 1.0f;
    st.last = inside ? n_contrib[pix] : 0u;
    st.dL0 = st.dL1 = st.dL2 = 0.f;
    if (inside) { st.dL0 = dL_dcolor[pix]; st.dL1 = dL_dcolor[N + pix]; st.dL2 = dL_dcolor[2 * N + pix]; }

    const uint32_t wave_last = wave_max_u32(st.last);
    if (lane == 0) s_wmax[w] = wave_last;
    __syncthreads();
    uint32_t grp_last = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) grp_last = max(grp_last, s_wmax[k]);

    // S = sum_c dL/dC_c * (colour composited BEHIND the current entry, background included, normalised by the
    // transmittance in front of it).  dL/dalpha_i = T_i (g_i - S_i) with g_i = sum_c dL/dC_c colour_i,c, and
    // S_{i-1} = S_i + alpha_i (g_i - S_i): the three per-channel recurrences of the textbook form collapse into one
    // scalar, and starting it at bg . dL/dC absorbs the separate background term (-T_final bg.dL / (1 - alpha_i)).
    st.S = vp.bg[0] * st.dL0 + vp.bg[1] * st.dL1 + vp.bg[2] * st.dL2;
    const bool alane = lane < 16 && (!(lane & 2) || lane == 2);        // the nine lanes that issue the per-entry atomics
    const uint32_t aoff = row_reduce_component(lane);

    const int nb = ((int)grp_last + BATCH - 1) / BATCH;
    // register prefetch of the next (nearer) batch, as in blend_forward_fine_kernel
    float4 p0[R], p1[R], p2[R];
    uint32_t pid[R];
    auto fetch = [&](int base) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = tid + k * T;
            if (e < BATCH && base + e < (int)grp_last) {
                pid[k] = ids[range.x + base + e];
                p0[k] = rec[pid[k]].r0; p1[k] = rec[pid[k]].r1; p2[k] = rec[pid[k]].r2;
            }
        }
    };
    if (nb > 0) fetch((nb - 1) * BATCH);
    for (int b = nb - 1; b >= 0; --b) {
        __syncthreads();                              // previous batch fully consumed (and flushed)
        const int base = b * BATCH;
        const int n = min(BATCH, (int)grp_last - base);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int e = tid + k * T;
            if (e < n) {
                s_r0[e] = doubled_w(p0[k]); s_r1[e] = p1[k]; s_b[e] = make_float2(p2[k].x, sign_test_bound(p1[k].y)); s_id[e] = pid[k];
                s_tau[e] = p2[k].w;
            }
        }
        if (b > 0) fetch(base - BATCH);
        __syncthreads();
        // classification against this wave's block and compaction in one pass
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < BATCH / 64; ++c) {
            const int e = c * 64 + lane;
            bool hit = false;
            if (e < n && (uint32_t)(base + e) < wave_last) {
                const float4 r0 = s_r0[e];
                const float C = s_r1[e].x, tau2 = s_tau[e];
                hit = !(tau2 > -1.0e38f) ||
                      levelset_hits_rect(r0.x, r0.y, r0.z, 0.5f * r0.w, C, tau2, bx0, bx0 + (float)(SB - 1), by0,
                                         by0 + (float)(SB - 1));
            }
            const uint64_t bal = __ballot(hit);
            if (hit) s_list[w][cnt + __popcll(bal & lt_mask)] = (uint16_t)e;
            cnt += __popcll(bal);
        }
        backward_walk<false>(s_list[w], cnt, base, s_r0, s_r1, s_b, s_id, pxf, pyf, st, alane, aoff, grad_rec);
    }
}

// ---------------------------------------------------------------------------------------------
// one-wave-per-tile backward: ONE wave64 per tile, four pixels per lane (lane l owns pixel (l & 7, l >> 3) of each of the
// four 8x8 quadrants).  Everything that is uniform over the tile — record fetch from LDS, loop control,
// the 64-lane reduction and the atomics — is paid once per (tile, Gaussian) instead of
// once per (quadrant, Gaussian); the quadrant hit mask becomes a wave-uniform branch per quadrant; there is
// no workgroup barrier (a single wave owns the tile) and the next batch of 64 records is prefetched into
// registers while the current one is consumed.  The four per-quadrant evaluations are independent, which
// gives the scheduler four-way ILP.
// ---------------------------------------------------------------------------------------------
constexpr int WB = 64;   // records per batch (one per lane)

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct BwdQuad {
    float T, S, dL0, dL1, dL2;       // S: see blend_backward_kernel
    uint32_t last;
};

// one (pixel, Gaussian) backward step accumulated into the per-lane partial sums; returns the lane's validity
// (no wave-level early-out here: every ballot-driven branch is a VALU -> SALU -> branch round trip, and the
// quadrant hit masks already removed the quadrants the record cannot touch)
// (returns the lanes that contributed as a scalar mask: three ballots of direct comparisons and scalar ANDs — a ballot
//  of a derived bool costs two VALU instructions, and the caller only needs "any lane?")
__device__ __forceinline__ uint64_t bwd_quad_step(BwdQuad& s, BwdSums& v, const float4& r0, const float4& r1, float cb,
                                                  float bound, float dx, float dy, uint32_t pos0) {
    const PairEval ev = eval_pair(r0.z, r0.w, r1.x, r1.y, dx, dy);
    const float a_raw = __builtin_amdgcn_exp2f(ev.p);
    // alpha = min(0.99, a_raw) >= 1/255  <=>  a_raw >= 1/255
    const uint64_t validm = __builtin_amdgcn_ballot_w64(pos0 < s.last) & __builtin_amdgcn_ballot_w64(ev.p <= bound) &
                            __builtin_amdgcn_ballot_w64(a_raw >= ALPHA_MIN);
    const bool valid = __builtin_amdgcn_inverse_ballot_w64(validm);
    // ONE select masks the lane: with a_m = 0 everything downstream is the identity (alpha 0, 1/(1-0) = 1 exactly,
    // T unchanged, zero contributions), so neither alpha nor T needs a select of its own
    const float a_m = valid ? a_raw : 0.0f;
    const float alpha_m = fminf(0.99f, a_m);
    const float inv = __builtin_amdgcn_rcpf(1.0f - alpha_m);
    const float Tn = s.T * inv;
    s.T = Tn;
    const float dch = alpha_m * Tn;
    const float sm = fmaf(cb, s.dL2, fmaf(r1.w, s.dL1, r1.z * s.dL0)) - s.S;
    const float dL_dalpha = sm * Tn;
    s.S = fmaf(alpha_m, sm, s.S);
    const float qq = a_m * dL_dalpha;                         // Q6: gradient passes the 0.99 clamp
    v.v0 = fmaf(qq, dx, v.v0); v.v1 = fmaf(qq, dy, v.v1);
    v.v2 = fmaf(qq, ev.dxx, v.v2); v.v3 = fmaf(qq, ev.dxy, v.v3); v.v4 = fmaf(qq, ev.dyy, v.v4);
    v.v5 += qq;
    v.v6 = fmaf(dch, s.dL0, v.v6); v.v7 = fmaf(dch, s.dL1, v.v7); v.v8 = fmaf(dch, s.dL2, v.v8);
    return validm;
}

#if defined(MSGS_TRACE_TILES)
// tool build (tools/trace_tiles.sh): per-tile {start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID, tile, traversal length} of the
// one-wave-per-tile backward, for the occupancy timeline in profiles/; never part of the product library
__device__ unsigned long long* g_tile_trace = nullptr;
__global__ void trace_set_kernel(unsigned long long* p) { g_tile_trace = p; }
#endif

// COUNT = true (diagnostic replica, msgs_blend_lane_stats): nothing is reduced or written; grad_out receives four counters —
// (tile, entry) visits, (quadrant, entry) evaluations (each 64 lanes), lanes that contributed, visits with a contribution.
// (The verification mode — msgs_set_deterministic — does not run this kernel: literal.hip restates the reference's loop.)
template <bool COUNT = false>
__global__ __launch_bounds__(64) void blend_backward_tile_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                                 const uint32_t* __restrict__ ids,
                                                                 const uint2* __restrict__ ranges,
                                                                 const float* __restrict__ final_T,
                                                                 const uint32_t* __restrict__ n_contrib,
                                                                 const float* __restrict__ dL_dcolor,
                                                                 void* __restrict__ grad_out,
                                                                 const uint32_t* __restrict__ tile_order) {
    __shared__ float4 s_r0[WB], s_r1[WB];
    __shared__ float4 s_bi[WB];                            // {blue, sign_test_bound, id bits, -}: 16-byte stride like s_r0 / s_r1, so one
                                                           // address register serves every LDS read of an entry
    const int num_tiles = vp.gx * vp.gy;
    const int lane = threadIdx.x;
    // launch order: heaviest tiles first inside every XCD's contiguous run (launch_tile_order) when the forward left a valid
    // order behind, the plain XCD swizzle otherwise
    int tile = swizzled_tile(blockIdx.x, num_tiles);
    if (tile_order != nullptr) {
        const uint32_t flag = tile_order[num_tiles];
        if (flag == TILE_ORDER_MAGIC) tile = (int)tile_order[tile];                       // per XCD run
    }
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int bx = tx * TILE + (lane & 7), by = ty * TILE + (lane >> 3);
    const float bxf = (float)bx, byf = (float)by;
    const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
    const uint2 range = ranges[tile];
    const size_t N = (size_t)vp.W * vp.H;

#if defined(MSGS_TRACE_TILES)
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t cnt_visits = 0, cnt_steps = 0, cnt_lanes = 0, cnt_hits = 0;       // COUNT only
    BwdQuad q0, q1, q2, q3;
    uint32_t ql0, ql1, ql2, ql3;                           // wave-uniform: last blended position per quadrant
    {
        auto init = [&](BwdQuad& s, int qi) -> uint32_t {
            const int px = bx + (qi & 1) * 8, py = by + (qi >> 1) * 8;
            const bool inside = px < vp.W && py < vp.H;
            const size_t pix = (size_t)py * vp.W + px;
            const float Tf = inside ? final_T[pix] : 1.0f;
            s.last = inside ? n_contrib[pix] : 0u;
            // (the counting replica does not read dL/dcolor: it only enters the sums, not the validity of a lane)
            s.dL0 = (inside && !COUNT) ? dL_dcolor[pix] : 0.f;
            s.dL1 = (inside && !COUNT) ? dL_dcolor[N + pix] : 0.f;
            s.dL2 = (inside && !COUNT) ? dL_dcolor[2 * N + pix] : 0.f;
            s.S = vp.bg[0] * s.dL0 + vp.bg[1] * s.dL1 + vp.bg[2] * s.dL2;
            s.T = Tf;
            return __builtin_amdgcn_readfirstlane(wave_max_u32(s.last));
        };
        ql0 = init(q0, 0); ql1 = init(q1, 1); ql2 = init(q2, 2); ql3 = init(q3, 3);
    }
    const uint32_t tile_last = max(max(ql0, ql1), max(ql2, ql3));
    const bool alane = lane < 16 && (!(lane & 2) || lane == 2);         // the nine lanes that issue the per-entry atomics
    const uint32_t aoff = row_reduce_component(lane);
    const int xrow16 = (lane ^ 16) << 2, xrow32 = (lane ^ 32) << 2;     // ds_bpermute byte addresses of the partner lanes

    const int nb = ((int)tile_last + WB - 1) / WB;
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
    uint32_t nid = 0;
    if (nb > 0) {
        const int i0 = (nb - 1) * WB + lane;
        if (i0 < (int)tile_last) { nid = ids[range.x + i0]; n0 = rec[nid].r0; n1 = rec[nid].r1; n2 = rec[nid].r2; }
    }
    for (int b = nb - 1; b >= 0; --b) {
        const int base = b * WB;
        const int n = min(WB, (int)tile_last - base);
        wave_fence();
        s_r0[lane] = doubled_w(n0); s_r1[lane] = n1; s_bi[lane] = make_float4(n2.x, sign_test_bound(n1.y), __uint_as_float(nid), 0.f);
        // quadrant hit masks of the batch as four 64-bit ballots; a record beyond the last blended entry of a
        // quadrant cannot matter to that quadrant
        const uint32_t mymask = lane < n ? quadrant_mask(n0, n1.x, n2.w, tx0, ty0) : 0u;
        const uint32_t mypos = (uint32_t)(base + lane);
        const uint64_t h0 = __ballot((mymask & 1u) && mypos < ql0), h1 = __ballot((mymask & 2u) && mypos < ql1),
                       h2 = __ballot((mymask & 4u) && mypos < ql2), h3 = __ballot((mymask & 8u) && mypos < ql3);
        wave_fence();
        if (b > 0) {                                      // prefetch the next (nearer) batch: always full
            nid = ids[range.x + base - WB + lane];
            n0 = rec[nid].r0; n1 = rec[nid].r1; n2 = rec[nid].r2;
        }
        uint64_t todo = h0 | h1 | h2 | h3;
        while (todo) {
            const int e = 63 - __builtin_clzll(todo);     // back to front
            const uint64_t bit = 1ull << e;
            todo &= ~bit;
            const uint32_t pos0 = (uint32_t)(base + e);   // 0-based position in the tile list
            const float4 r0 = s_r0[e], r1 = s_r1[e];
            const float2 bl = *reinterpret_cast<const float2*>(&s_bi[e]);      // {blue, sign_test_bound}: one 8-byte read
            const float cb = bl.x;
            const float dx = r0.x - bxf, dy = r0.y - byf;
            BwdSums v = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            uint64_t any = 0;
            if constexpr (COUNT) {
                uint64_t m;
                cnt_visits += 1u;
                if (h0 & bit) { m = bwd_quad_step(q0, v, r0, r1, cb, bl.y, dx, dy, pos0); any |= m; cnt_steps += 1u; cnt_lanes += (uint32_t)__popcll(m); }
                if (h1 & bit) { m = bwd_quad_step(q1, v, r0, r1, cb, bl.y, dx - 8.0f, dy, pos0); any |= m; cnt_steps += 1u; cnt_lanes += (uint32_t)__popcll(m); }
                if (h2 & bit) { m = bwd_quad_step(q2, v, r0, r1, cb, bl.y, dx, dy - 8.0f, pos0); any |= m; cnt_steps += 1u; cnt_lanes += (uint32_t)__popcll(m); }
                if (h3 & bit) { m = bwd_quad_step(q3, v, r0, r1, cb, bl.y, dx - 8.0f, dy - 8.0f, pos0); any |= m; cnt_steps += 1u; cnt_lanes += (uint32_t)__popcll(m); }
                if (any) cnt_hits += 1u;
                continue;
            }
            if (h0 & bit) any |= bwd_quad_step(q0, v, r0, r1, cb, bl.y, dx, dy, pos0);
            if (h1 & bit) any |= bwd_quad_step(q1, v, r0, r1, cb, bl.y, dx - 8.0f, dy, pos0);
            if (h2 & bit) any |= bwd_quad_step(q2, v, r0, r1, cb, bl.y, dx, dy - 8.0f, pos0);
            if (h3 & bit) any |= bwd_quad_step(q3, v, r0, r1, cb, bl.y, dx - 8.0f, dy - 8.0f, pos0);
            if (any == 0) continue;                        // no lane contributed: nothing to reduce
            // ---- one 64-lane reduction per (tile, Gaussian): rows by DPP (row_reduce_scatter9), then the four rows
            // through the LDS crossbar (ds_bpermute lane ^ 16, lane ^ 32: two adds on the VALU; v_permlane16/32_swap are
            // multi-cycle there).  Lanes 0,1,4,5,8,9,12,13 then hold components 0..7 and lane 2 component 8: one atomic
            // instruction; the record id is wave-uniform (scalar address arithmetic).
            {
                const float outv = cross_row_allreduce_bperm(row_reduce_scatter9(v), xrow16, xrow32);
                // (record id through v_readlane of a register copy and a scalar-base atomic — no 64-bit VALU multiply-add —
                //  were measured: no difference, 357..382 us for all four combinations)
                const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(s_bi[e].z));
                grad_acc_t* gdst = (grad_acc_t*)grad_out + (size_t)gid * GRAD_REC_FLOATS;
                if (alane) unsafeAtomicAdd(gdst + aoff, (grad_acc_t)outv);
            }
        }
    }
    if constexpr (COUNT) {
        if (lane == 0) {
            unsigned long long* o = (unsigned long long*)grad_out;
            atomicAdd(&o[0], (unsigned long long)cnt_visits); atomicAdd(&o[1], (unsigned long long)cnt_steps);
            atomicAdd(&o[2], (unsigned long long)cnt_lanes); atomicAdd(&o[3], (unsigned long long)cnt_hits);
        }
    }
#if defined(MSGS_TRACE_TILES)
    if (!COUNT && g_tile_trace && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g_tile_trace + 4ull * blockIdx.x;
        o[0] = trace_t0; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = ((unsigned long long)xcc << 32) | hw; o[3] = ((unsigned long long)tile << 32) | tile_last;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// statistics for the algorithmic-bytes formula: D_trav = sum_tiles max_pixels n_contrib, V = #radii>0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_stats_kernel(ViewParams vp, const uint32_t* __restrict__ n_contrib,
                                                         unsigned long long* __restrict__ out) {
    __shared__ uint32_t s_wmax[4];
    const int tile = blockIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int px = tx * TILE + (tid & 15), py = ty * TILE + (tid >> 4);
    const uint32_t v = (px < vp.W && py < vp.H) ? n_contrib[(size_t)py * vp.W + px] : 0u;
    const uint32_t m = wave_max_u32(v);
    if (lane == 0) s_wmax[w] = m;
    __syncthreads();
    if (tid == 0) atomicAdd(&out[0], (unsigned long long)max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));
}

__global__ __launch_bounds__(256) void visible_count_kernel(int P, const int32_t* __restrict__ radii,
                                                            unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool vis = i < P && radii[i] > 0;
    const uint64_t b = __ballot(vis);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&out[1], (unsigned long long)__popcll(b));
}

}  // namespace

// Forward: one 256-thread workgroup per tile, one 8x8 quadrant per wave (229 us at C3 in round 1 against 405 us for one wave per
// tile; the strip / 4x4-block list variants of rounds 3-4 measured slower as well — profiles/design_history_r1_r4.md, git history).
// Backward: one wave64 per tile, four pixels per lane, one reduction per (tile, Gaussian) from 4096 tiles up; four waves per tile
// below.  (A packed-fp32 forward with two pixels per lane was also built and measured: correct but 17 % slower — v_pk_*_f32 is
// not full rate on gfx950 — and removed; for the same reason this library is compiled with -fno-slp-vectorize, which alone took
// the backward from 558 to 495 us.)
// backward generation: MSGS_BWD_GEN = 1 | 2 forces one; default (0) picks by tile count — one wave per tile (gen 2) needs
// >= ~4000 tiles to occupy 1024 SIMDs, below that four waves per tile (gen 1) win (C3 scene, profiles/r1_notes.md:
// 8160 tiles 437 vs 667 us; 2040 tiles 334 vs 297; 510 tiles 481 vs 205; 135 tiles 834 vs 303; 2 tiles 1378 vs 431)
constexpr int BWD_GEN2_MIN_TILES = 4096;
// below these tile counts: the sixteen-waves-per-tile kernels (latency-bound regime).  C3 scene, profiles/r1_notes.md:
// forward 510 tiles 109 -> 100 us, 135 tiles 184 -> 112, 40 tiles 285 -> 161; backward 510 tiles 167 -> 184 (worse),
// 135 tiles 262 -> 185, 40 tiles 372 -> 236
constexpr int FINE_MAX_TILES_FWD = 300, FINE_MAX_TILES_BWD = 300;     // (round 3 sweep: 510 tiles forward 102 us coarse vs 127 fine;
                                                                       //  135 tiles 155 vs 104; backward 510 tiles 168 vs 229, 135 tiles 263 vs 182)
static std::atomic<int> g_bwd_gen{[] { const char* e = getenv("MSGS_BWD_GEN"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }()};
int set_backward_generation(int gen) { return g_bwd_gen.exchange(gen == 1 || gen == 2 ? gen : 0); }
// blend granularity: 0 = by tile count, 1 = coarse (quadrant / tile per wave), 2 = fine (4x4 sub-block per wave)
static std::atomic<int> g_granularity{[] { const char* e = getenv("MSGS_BLEND_GRANULARITY"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }()};
int set_blend_granularity(int mode) { return g_granularity.exchange(mode == 1 || mode == 2 ? mode : 0); }
// Shape of the fine-grained kernels: the sub-block side SB a wave owns (4 | 2 | 1 -> 16 | 64 | 256 waves per tile) and the
// number G of workgroups a tile's waves are split over (every workgroup stages and classifies the tile's whole list).
// Measured on the pyramid of the C3 scene, forward / backward us at 135, 40, 12, 2 tiles (profiles/r3_notes.md):
//   SB 4: G = 1: 105/184, 139/228, 146/241, 132/224;  G = 2: 87/152, 114/192, 115/194, 100/175;  G = 4: 87/140, 111/180, 108/182,
//         98/166;  G = 8: 112/151, 133/190, 138/188, 117/166;  G = 16: 152/164, 158/198, 149/176, 130/162
//   SB 2: G = 4: 156/251, 104/140, 100/133, 95/129;  G = 8: 128/181, 85/122, 83/117, 75/107;  G = 16: 122/168, 82/111, 82/107,
//         74/98;  G = 32: 167/207, 115/130, 104/108, 91/97
//   SB 1: G = 16: 329/368, 139/152, 82/94, 76/89;  G = 32: 277/330, 120/130, 73/86, 65/81;  G = 64: 246/304, 112/131, 71/84, 62/74
// i.e. the best shape keeps 2000-3000 waves in flight: 4x4 blocks down to ~64 tiles, 2x2 blocks down to ~16, single pixels below.
struct FineShape { int sb, g; };
static FineShape fine_shape(int tiles) {
    FineShape f;
    f.sb = tiles <= 16 ? 1 : (tiles <= 64 ? 2 : 4);
    f.g = f.sb == 1 ? 64 : (f.sb == 2 ? 16 : (tiles < 768 ? 4 : 1));   // (>= 768 tiles — forced granularity only — one workgroup per
    return f;                                                           //  tile fills the chip)
}
template <int WAVES, int SB = 4, class... Args>
static void launch_fine_fwd(int tiles, hipStream_t s, Args... args) {
    constexpr int NSB = (TILE / SB) * (TILE / SB);
    hipLaunchKernelGGL((blend_forward_fine_kernel<WAVES, SB>), dim3(tiles * (NSB / WAVES)), dim3(64 * WAVES), 0, s, args...);
}
template <int WAVES, int SB = 4, class... Args>
static void launch_fine_bwd(int tiles, hipStream_t s, Args... args) {
    constexpr int NSB = (TILE / SB) * (TILE / SB);
    hipLaunchKernelGGL((blend_backward_fine_kernel<WAVES, SB>), dim3(tiles * (NSB / WAVES)), dim3(64 * WAVES), 0, s, args...);
}
static bool use_fine(int tiles, int max_tiles) {
    const int g = g_granularity.load();
    return g == 2 || (g == 0 && tiles < max_tiles);
}
static bool bwd_v1(int tiles) {
    const int forced = g_bwd_gen.load();
    return forced ? forced == 1 : tiles < BWD_GEN2_MIN_TILES;
}

// ---------------------------------------------------------------------------------------------
// Launch order of the one-wave-per-tile backward.  A tile is ONE sequential wave there and a SIMD holds six of them, about eight
// tiles per SIMD in total at 1080p: with tiles dispatched in image order the SIMDs run dry at very different times (measured:
// the number of running waves falls linearly over the second half of the kernel).  Dispatching the heaviest tiles first
// leaves only light tiles for the end.  Key = the forward's per-tile traversal length (tile_last), NOT the list length: with
// early termination the two differ by 2.3x on average and by much more per tile.  One 256-thread block per XCD run of the
// swizzled order (the tiles of a run stay on their XCD): counting sort into 2048 bins of 8 entries, heaviest first.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int ORDER_BINS = 2048, ORDER_SHIFT = 3;
// (One global heaviest-first sequence dealt to the XCDs round-robin was measured too: no better, and it gives up the L2 locality
//  of the contiguous runs.)
// (256 threads per block since round 4: a 1024-thread block needs sixteen free wave slots on ONE CU, and with a second view in
//  flight — host/multi_view.py — it waited 30-45 us for them behind the other view's blend workgroups; four slots are found at once)
constexpr int ORDER_THREADS = 256, ORDER_PER_THREAD = ORDER_BINS / ORDER_THREADS;
__global__ __launch_bounds__(ORDER_THREADS) void tile_order_kernel(int num_tiles, const uint32_t* __restrict__ tile_last,
                                                                   uint32_t* __restrict__ order) {
    __shared__ uint32_t s_bin[ORDER_BINS];
    __shared__ uint32_t s_wave[ORDER_THREADS / 64];
    const int per = num_tiles >> 3, main = per << 3;
    const int x = blockIdx.x;                           // XCD run x: swizzled positions = tiles [x * per, (x + 1) * per)
    const int tid = threadIdx.x;
    for (int b = tid; b < ORDER_BINS; b += ORDER_THREADS) s_bin[b] = 0;
    __syncthreads();
    for (int j = tid; j < per; j += ORDER_THREADS) {
        const uint32_t key = min(tile_last[x * per + j] >> ORDER_SHIFT, (uint32_t)(ORDER_BINS - 1));
        atomicAdd(&s_bin[ORDER_BINS - 1 - key], 1u);    // bin 0 = heaviest
    }
    __syncthreads();
    // exclusive scan of the 2048 bins: ORDER_PER_THREAD consecutive bins per thread
    uint32_t c[ORDER_PER_THREAD], v = 0;
#pragma unroll
    for (int k = 0; k < ORDER_PER_THREAD; ++k) { c[k] = s_bin[ORDER_PER_THREAD * tid + k]; v += c[k]; }
    const int lane = tid & 63, w = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += n;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; ++k) base += s_wave[k];
    uint32_t run = base + inc - v;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORDER_PER_THREAD; ++k) { s_bin[ORDER_PER_THREAD * tid + k] = run; run += c[k]; }
    __syncthreads();
    for (int j = tid; j < per; j += ORDER_THREADS) {
        const uint32_t t = (uint32_t)(x * per + j);
        const uint32_t key = min(tile_last[t] >> ORDER_SHIFT, (uint32_t)(ORDER_BINS - 1));
        const uint32_t pos = atomicAdd(&s_bin[ORDER_BINS - 1 - key], 1u);
        order[x * per + pos] = t;
    }
    if (x == 0) {
        for (int t = main + tid; t < num_tiles; t += ORDER_THREADS) order[t] = (uint32_t)t;      // ragged tail: identity
        if (tid == 0) order[num_tiles] = TILE_ORDER_MAGIC;     // visible to the backward through the stream order
    }
}
}  // namespace

static bool bwd_uses_tile_kernel(int tiles) {
    return !(g_granularity.load() == 2 || (g_bwd_gen.load() == 0 && use_fine(tiles, FINE_MAX_TILES_BWD))) && !bwd_v1(tiles);
}

hipError_t launch_tile_order(const ViewParams& vp, const uint32_t* tile_last, uint32_t* tile_order, hipStream_t s) {
    const int tiles = vp.gx * vp.gy;
    if (tiles < 8 || !bwd_uses_tile_kernel(tiles)) return hipSuccess;
    // the fine-grained forward kernels do not write tile_last (and clear the order's validity word): no order from garbage
    if (use_fine(tiles, FINE_MAX_TILES_FWD)) return hipSuccess;
    hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(ORDER_THREADS), 0, s, tiles, tile_last, tile_order);
    return hipGetLastError();
}

hipError_t launch_blend_forward(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                float* out_color, float* out_ps, float* out_depth, float* final_T,
                                uint32_t* n_contrib, uint32_t* tile_last, void* clear_ptr, size_t clear_bytes, hipStream_t s,
                                const FwdSlabArgs* slab) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0) return clear_ptr && clear_bytes ? launch_zero(clear_ptr, clear_bytes, s) : hipSuccess;
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom);   // GeomLayout::rec == 0
    // the gradient records are cleared by the blend kernel itself when it has enough workgroups to spread the stores over the
    // chip; with few tiles (low pyramid levels: 2 .. 500 workgroups, measured 150 -> 410 us at 2 tiles for 80 MB) a fill
    // kernel in front of it is faster
    static_assert(GRAD_REC_BYTES % 16 == 0, "the forward clears the gradient records in 16-byte words");
    const bool inline_clear = clear_ptr && clear_bytes && tiles >= CLEAR_INLINE_MIN_TILES;
    if (clear_ptr && clear_bytes && !inline_clear) {
        hipError_t e = launch_zero(clear_ptr, clear_bytes, s);
        if (e != hipSuccess) return e;
    }
    uint4* const cp = inline_clear ? reinterpret_cast<uint4*>(clear_ptr) : nullptr;
    const size_t cn = inline_clear ? clear_bytes / 16 : 0;
    // tile_last sits in front of tile_order in the image state (ImageLayout): the order's validity word
    uint32_t* order_flag = tile_last ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tile_last) +
                                                                   align256(4 * (size_t)tiles)) + tiles : nullptr;
    auto fine = [&] {            // few tiles (low pyramid levels): one wave per 4x4 sub-block, the tile split over G workgroups
        const FineShape f = fine_shape(tiles);
#define MSGS_FINE_FWD(WAVES, SB) launch_fine_fwd<WAVES, SB>(tiles, s, vp, rec, ids, ranges, out_color, out_ps, out_depth, final_T, \
                                                            n_contrib, cp, cn, order_flag)
        if (f.sb == 1) {                      // single pixels: 256 waves per tile
            if (f.g >= 64) MSGS_FINE_FWD(4, 1); else if (f.g == 32) MSGS_FINE_FWD(8, 1); else MSGS_FINE_FWD(16, 1);
        } else if (f.sb == 2) {               // 2x2 sub-blocks: 64 waves per tile
            if (f.g >= 32) MSGS_FINE_FWD(2, 2); else if (f.g == 16) MSGS_FINE_FWD(4, 2); else if (f.g == 8) MSGS_FINE_FWD(8, 2);
            else MSGS_FINE_FWD(16, 2);
        } else {                              // 4x4 sub-blocks: 16 waves per tile
            if (f.g >= 16) MSGS_FINE_FWD(1, 4); else if (f.g == 8) MSGS_FINE_FWD(2, 4); else if (f.g == 4) MSGS_FINE_FWD(4, 4);
            else if (f.g == 2) MSGS_FINE_FWD(8, 4); else MSGS_FINE_FWD(16, 4);
        }
#undef MSGS_FINE_FWD
    };
    FwdSlab sb{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (slab) {
        sb.dtrav = slab->dtrav;
        if (slab->pass == 1) { sb.open_bits = slab->open_bits; sb.open_list = slab->open_list; sb.n_open = slab->n_open; }
        if (slab->pass == 2) { sb.tile_list = slab->open_list; sb.n_tiles = slab->n_open; }
    }
    if (use_fine(tiles, FINE_MAX_TILES_FWD)) {
        if (slab && slab->pass != 0) return hipErrorInvalidValue;        // (the caller engages slabs from SLAB_MIN_TILES tiles on)
        fine();
    } else {
        // slab B walks the list of open tiles with a grid that fills the chip once (normally a handful of tiles)
        const int grid = sb.tile_list ? std::min(tiles, 2048) : tiles;
        hipLaunchKernelGGL(blend_forward_kernel<false>, dim3(grid), dim3(256), 0, s, vp, rec, ids, ranges, out_color, out_ps,
                           out_depth, final_T, n_contrib, (unsigned long long*)nullptr, cp, cn, tile_last, order_flag, sb);
    }
    return hipGetLastError();
}

hipError_t launch_forward_feedback(const unsigned long long* dtrav, const SlabHeader* hdr, int slab_mode, int64_t D,
                                   const uint32_t* D_dev, uint32_t tag, uint64_t ticket, uint64_t* host_mapped, hipStream_t s) {
    if (!host_mapped || !dtrav) return hipSuccess;
    hipLaunchKernelGGL(forward_feedback_kernel, dim3(1), dim3(64), 0, s, dtrav, hdr, slab_mode, D, D_dev, tag, ticket,
                       (volatile uint64_t*)host_mapped);
    return hipGetLastError();
}

// does the forward of a `tiles`-tile image run the quadrant-list kernel (the one that can publish open tiles / tile lengths)?
bool forward_uses_quadrant_kernel(int tiles) { return tiles > 0 && !use_fine(tiles, FINE_MAX_TILES_FWD); }

hipError_t launch_blend_backward(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 grad_acc_t* grad_rec, hipStream_t s, const uint32_t* tile_order) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0) return hipSuccess;
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom);
    if (g_granularity.load() == 2 || (g_bwd_gen.load() == 0 && use_fine(tiles, FINE_MAX_TILES_BWD))) {
        const FineShape f = fine_shape(tiles);
#define MSGS_FINE_BWD(WAVES, SB) launch_fine_bwd<WAVES, SB>(tiles, s, vp, rec, ids, ranges, final_T, n_contrib, dL_dcolor, grad_rec)
        if (f.sb == 1) {
            if (f.g >= 64) MSGS_FINE_BWD(4, 1); else if (f.g == 32) MSGS_FINE_BWD(8, 1); else MSGS_FINE_BWD(16, 1);
        } else if (f.sb == 2) {
            if (f.g >= 32) MSGS_FINE_BWD(2, 2); else if (f.g == 16) MSGS_FINE_BWD(4, 2); else if (f.g == 8) MSGS_FINE_BWD(8, 2);
            else MSGS_FINE_BWD(16, 2);
        } else {
            if (f.g >= 16) MSGS_FINE_BWD(1, 4); else if (f.g == 8) MSGS_FINE_BWD(2, 4); else if (f.g == 4) MSGS_FINE_BWD(4, 4);
            else if (f.g == 2) MSGS_FINE_BWD(8, 4); else MSGS_FINE_BWD(16, 4);
        }
#undef MSGS_FINE_BWD
    }
    else if (!bwd_v1(tiles))
        hipLaunchKernelGGL(blend_backward_tile_kernel<false>, dim3(tiles), dim3(64), 0, s, vp, rec, ids, ranges, final_T,
                           n_contrib, dL_dcolor, grad_rec, tile_order);
    else
        hipLaunchKernelGGL(blend_backward_kernel, dim3(tiles), dim3(256), 0, s, vp, rec, ids, ranges, final_T,
                           n_contrib, dL_dcolor, grad_rec);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Deterministic backward (msgs_set_deterministic): no float atomics.  K7 stores the nine sums of every tile entry;
// the entries are then grouped by Gaussian with a STABLE sort of (Gaussian id, entry position) and each Gaussian's
// entries are added in ascending entry position (= ascending tile id) by one thread.  Bitwise reproducible run to run.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void det_reduce_kernel(const uint32_t* __restrict__ gid_sorted,
                                                         const uint32_t* __restrict__ entry_of, int64_t D,
                                                         const double* __restrict__ inst_grad,
                                                         grad_acc_t* __restrict__ grad_rec, int rec_stride) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= D) return;
    const uint32_t g = gid_sorted[q];
    if (q > 0 && gid_sorted[q - 1] == g) return;               // not the head of its segment
    // the per-tile sums of one Gaussian are added in DOUBLE (exact for any realistic tile count): the deterministic mode
    // is the verification mode, and a Gaussian that covers hundreds of tiles otherwise loses ~1e-6 of its sums to float32
    // accumulation, which the conic -> covariance chain of K8 amplifies by the squared aspect ratio
    double acc[DET_INST_FLOATS];
#pragma unroll
    for (int c = 0; c < DET_INST_FLOATS; ++c) acc[c] = 0.0;
    for (int64_t k = q; k < D && gid_sorted[k] == g; ++k) {
        const double* src = inst_grad + (size_t)entry_of[k] * DET_INST_FLOATS;
#pragma unroll
        for (int c = 0; c < DET_INST_FLOATS; ++c) acc[c] += src[c];
    }
    grad_acc_t* dst = grad_rec + (size_t)g * rec_stride;
#pragma unroll
    for (int c = 0; c < DET_INST_FLOATS; ++c) dst[c] = (grad_acc_t)acc[c];
}
}  // namespace

DetScratch::DetScratch(int64_t P, int64_t D) {
    const int64_t n = D > 0 ? D : 1;
    size_t o = 0;
    grad_rec = o;  o = align256(o + GRAD_REC_BYTES * (size_t)(P > 0 ? P : 1));
    inst_grad = o; o = align256(o + sizeof(double) * DET_INST_FLOATS * (size_t)n);
    keys = o;      o = align256(o + 4 * (size_t)n);
    keys_s = o;    o = align256(o + 4 * (size_t)n);
    entry = o;     o = align256(o + 4 * (size_t)n);
    sort = o;      o = align256(o + SortScratch(n).total);
    total = o;
}

hipError_t launch_blend_backward_det(const ViewParams& vp, int P, const char* geom, const uint32_t* ids, int64_t D,
                                     const uint2* ranges, const float* final_T, const uint32_t* n_contrib,
                                     const float* dL_dcolor, char* scratch, hipStream_t s) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0 || D <= 0) return hipSuccess;
    const DetScratch L(P, D);
    grad_acc_t* grad_rec = (grad_acc_t*)(scratch + L.grad_rec);
    double* inst = (double*)(scratch + L.inst_grad);
    uint32_t* keys = (uint32_t*)(scratch + L.keys);
    uint32_t* keys_s = (uint32_t*)(scratch + L.keys_s);
    uint32_t* entry = (uint32_t*)(scratch + L.entry);
    hipError_t e = launch_zero(inst, sizeof(double) * DET_INST_FLOATS * (size_t)D, s);
    if (e != hipSuccess) return e;
    // the reference's per-pixel backward restated literally (literal.hip): nine double sums per tile entry
    e = launch_blend_backward_literal(vp, geom, P, ids, ranges, final_T, n_contrib, dL_dcolor, inst, s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(keys, ids, 4 * (size_t)D, hipMemcpyDeviceToDevice, s);      // the sort clobbers its input
    if (e != hipSuccess) return e;
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < P) ++bits;
    e = radix_sort_pairs(keys, nullptr, keys_s, entry, D, 0, bits, scratch + L.sort, s);   // stable: entries ascending
    if (e != hipSuccess) return e;
    // ... added per Gaussian in ascending tile order: [P, 9] packed doubles, the textbook sums the per-Gaussian backward takes
    static_assert(sizeof(grad_acc_t) == 8, "the verification mode's sums are doubles");
    hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, s, keys_s, entry, D, inst,
                       grad_rec, DET_INST_FLOATS);
    return hipGetLastError();
}

// diagnostic: the quadrant-per-wave forward replayed with lane counters (no outputs written)
hipError_t launch_blend_backward_lane_stats(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                            const float* final_T, const uint32_t* n_contrib, unsigned long long* out4,
                                            hipStream_t s) {
    hipError_t e = hipMemsetAsync(out4, 0, 32, s);
    if (e != hipSuccess) return e;
    const int tiles = vp.gx * vp.gy;
    if (tiles)
        hipLaunchKernelGGL(blend_backward_tile_kernel<true>, dim3(tiles), dim3(64), 0, s, vp,
                           reinterpret_cast<const GaussRec*>(geom), ids, ranges, final_T, n_contrib, (const float*)nullptr, (void*)out4,
                           (const uint32_t*)nullptr);
    return hipGetLastError();
}

hipError_t launch_blend_lane_stats(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                   unsigned long long* out3, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out3, 0, 24, s);
    if (e != hipSuccess) return e;
    const int tiles = vp.gx * vp.gy;
    if (tiles)
        hipLaunchKernelGGL(blend_forward_kernel<true>, dim3(tiles), dim3(256), 0, s, vp, reinterpret_cast<const GaussRec*>(geom),
                           ids, ranges, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (uint32_t*)nullptr, out3, (uint4*)nullptr, (size_t)0, (uint32_t*)nullptr, (uint32_t*)nullptr,
                           FwdSlab{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
    return hipGetLastError();
}

hipError_t launch_binning_stats(const ViewParams& vp, int P, const int32_t* radii, const uint32_t* n_contrib,
                                unsigned long long* out2, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out2, 0, 16, s);
    if (e != hipSuccess) return e;
    const int tiles = vp.gx * vp.gy;
    if (tiles) hipLaunchKernelGGL(tile_stats_kernel, dim3(tiles), dim3(256), 0, s, vp, n_contrib, out2);
    if (P) hipLaunchKernelGGL(visible_count_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, out2);
    return hipGetLastError();
}

}  // namespace msgs

#if defined(MSGS_TRACE_TILES)
extern "C" int msgs_debug_set_tile_trace(void* p) {
    hipLaunchKernelGGL(msgs::trace_set_kernel, dim3(1), dim3(1), 0, 0, (unsigned long long*)p);
    return (int)hipDeviceSynchronize();
}
#endif
