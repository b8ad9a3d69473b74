// blend.hip — per-tile front-to-back alpha-composite forward (K6) and back-to-front gradient
// backward (K7) for gfx950.  Replaces renderCUDA forward/backward of the
// reference's un-vendored CUDA module (call site gaussian_renderer/__init__.py:94-108).
//
// Structure (both kernels): one 256-thread workgroup per 16x16 tile; the four wave64s each own one
// 8x8 pixel QUADRANT (lane l -> pixel (l & 7, l >> 3) of the quadrant).  The tile's depth-sorted
// Gaussian list is staged through LDS in batches of 256 records (coalesced 4-B id loads, then three
// 16-B gathers per record).  The loading thread also classifies its record against the four
// quadrants with the exact alpha >= 1/255 ellipse test, and every wave compacts the batch to the
// entries that can touch ITS quadrant (64-bit ballot + popcount prefix), so the per-pixel loop only
// visits records that matter for that wave.  Skipping a record whose alpha is < 1/255 on every
// pixel of the quadrant is exactly what the per-pixel `continue` of the reference does, so results
// are unchanged.
//
// Backward: per (wave, record) the nine partial gradients are reduced across the 64 lanes with a
// DPP reduce-scatter (8 values: 2 halving steps inside quads, then a row all-reduce with row_ror and a
// cross-row all-reduce with v_permlane16/32_swap on the remaining 2 values per lane; the 9th value with
// a plain reduction) and committed with two
// global_atomic_add_f32 instructions (4 + 5 lanes) into a 48-byte per-Gaussian gradient record —
// one atomic per (quadrant, Gaussian, component) instead of the reference's one per (pixel,
// Gaussian, component).
//
// Roofline: HBM-bound by contract (BASELINE.json); algorithmic bytes K6 = 48*D_trav + 28*N + 8*tiles,
// K7 = 48*D_trav + 20*N + 36*V.  In practice both are VALU/transcendental-bound.
#include "msgs_internal.h"

namespace msgs {

namespace {

constexpr int BATCH = 256;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_MIN = 0.0001f;

__device__ __forceinline__ float fast_exp(float x) {
    return __builtin_amdgcn_exp2f(__fmul_rn(x, 1.4426950408889634f));
}

// power = -1/2 (A dx^2 + C dy^2) - B dx dy with a fixed operation order, so that the forward and the
// backward kernel take bit-identical skip decisions (alpha < 1/255) whatever the compiler contracts.
__device__ __forceinline__ float gauss_power(float A, float B, float C, float dx, float dy) {
    const float s = __fmaf_rn(__fmul_rn(A, dx), dx, __fmul_rn(__fmul_rn(C, dy), dy));
    return __fmaf_rn(-0.5f, s, -__fmul_rn(__fmul_rn(B, dx), dy));
}
__device__ __forceinline__ float gauss_alpha(float opacity, float G) { return fminf(0.99f, __fmul_rn(opacity, G)); }

// quadrant hit mask of one record (bit q: quadrant q = qx + 2*qy of the tile at (tx0, ty0))
__device__ __forceinline__ uint32_t quadrant_mask(const float4& r0, float conC, float tau, float tx0, float ty0) {
    if (!(tau < 1.0e38f)) return 0xFu;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = tx0 + (float)((q & 1) * 8), y0 = ty0 + (float)((q >> 1) * 8);
        if (ellipse_hits_rect(r0.x, r0.y, r0.z, r0.w, conC, tau, x0, x0 + 7.0f, y0, y0 + 7.0f)) m |= 1u << q;
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

// ---------------------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_forward_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                            const uint32_t* __restrict__ ids,
                                                            const uint2* __restrict__ ranges,
                                                            float* __restrict__ out_color,
                                                            float* __restrict__ out_ps,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib) {
    __shared__ float4 s_r0[BATCH], s_r1[BATCH], s_r2[BATCH];
    __shared__ uint32_t s_mask[BATCH];
    __shared__ uint16_t s_list[4][BATCH];

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
    const int len = (int)(range.y - range.x);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, aps = 0.f, adp = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (int base = 0; base < len; base += BATCH) {
        if (__syncthreads_and(done)) break;          // barrier also protects the LDS batch
        const int n = min(BATCH, len - base);
        if (tid < n) {
            const uint32_t id = ids[range.x + base + tid];
            const float4 r0 = rec[id].r0, r1 = rec[id].r1, r2 = rec[id].r2;
            s_r0[tid] = r0; s_r1[tid] = r1; s_r2[tid] = r2;
            s_mask[tid] = quadrant_mask(r0, r1.x, r2.w, tx0, ty0);
        }
        __syncthreads();
        // per-wave compaction of the batch to this wave's quadrant
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < BATCH / 64; ++c) {
            const int e = c * 64 + lane;
            const bool hit = e < n && ((s_mask[e] >> w) & 1u);
            const uint64_t b = __ballot(hit);
            if (hit) s_list[w][cnt + __popcll(b & lt_mask)] = (uint16_t)e;
            cnt += __popcll(b);
        }
        for (int j = 0; j < cnt; ++j) {
            if (__ballot(!done) == 0) break;
            const int e = s_list[w][j];
            const float4 r0 = s_r0[e], r1 = s_r1[e], r2 = s_r2[e];
            const float dx = r0.x - pxf, dy = r0.y - pyf;
            const float power = gauss_power(r0.z, r0.w, r1.x, dx, dy);
            const float alpha = gauss_alpha(r1.y, fast_exp(power));
            const bool valid = !done && power <= 0.0f && alpha >= ALPHA_MIN;
            const float test_T = T * (1.0f - alpha);
            const bool stop = valid && test_T < T_MIN;
            done = done || stop;
            const bool blend = valid && !stop;
            const float wgt = blend ? alpha * T : 0.0f;
            C0 += r1.z * wgt; C1 += r1.w * wgt; C2 += r2.x * wgt;
            adp += r2.y * wgt; aps += r2.z * wgt;
            T = blend ? test_T : T;
            last = blend ? (uint32_t)(base + e + 1) : last;
        }
    }
    if (inside) {
        const size_t N = (size_t)vp.W * vp.H;
        const size_t pix = (size_t)py * vp.W + px;
        out_color[pix] = C0 + T * vp.bg[0];
        out_color[N + pix] = C1 + T * vp.bg[1];
        out_color[2 * N + pix] = C2 + T * vp.bg[2];
        out_ps[pix] = aps;
        out_depth[pix] = adp;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ---------------------------------------------------------------------------------------------
// K7
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_backward_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                             const uint32_t* __restrict__ ids,
                                                             const uint2* __restrict__ ranges,
                                                             const float* __restrict__ final_T,
                                                             const uint32_t* __restrict__ n_contrib,
                                                             const float* __restrict__ dL_dcolor,
                                                             float* __restrict__ grad_rec) {
    __shared__ float4 s_r0[BATCH], s_r1[BATCH];
    __shared__ float s_b[BATCH];
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

    const float T_final = inside ? final_T[pix] : 1.0f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
    if (inside) { dL0 = dL_dcolor[pix]; dL1 = dL_dcolor[N + pix]; dL2 = dL_dcolor[2 * N + pix]; }
    const float bg_dot = vp.bg[0] * dL0 + vp.bg[1] * dL1 + vp.bg[2] * dL2;
    const float ddelx_dx = 0.5f * vp.W, ddely_dy = 0.5f * vp.H;

    const uint32_t wave_last = wave_max_u32(last);
    if (lane == 0) s_wmax[w] = wave_last;
    __syncthreads();
    const uint32_t tile_last = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    float T = T_final;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;       // accum_rec
    float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;          // last_color
    float last_alpha = 0.f;
    const bool p0 = lane & 1, p1 = (lane >> 1) & 1;
    const int vbase = 4 * (int)p0 + 2 * (int)p1;

    const int nb = ((int)tile_last + BATCH - 1) / BATCH;
    for (int b = nb - 1; b >= 0; --b) {
        __syncthreads();                              // previous batch fully consumed
        const int base = b * BATCH;
        const int n = min(BATCH, (int)tile_last - base);
        if (tid < n) {
            const uint32_t id = ids[range.x + base + tid];
            const float4 r0 = rec[id].r0, r1 = rec[id].r1;
            const float4 r2 = rec[id].r2;
            s_r0[tid] = r0; s_r1[tid] = r1; s_b[tid] = r2.x; s_id[tid] = id;
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
        for (int j = cnt - 1; j >= 0; --j) {
            const int e = s_list[w][j];
            const float4 r0 = s_r0[e], r1 = s_r1[e];
            const float cb = s_b[e];
            const float dx = r0.x - pxf, dy = r0.y - pyf;
            const float power = gauss_power(r0.z, r0.w, r1.x, dx, dy);
            const float G = fast_exp(power);
            const float alpha = gauss_alpha(r1.y, G);
            const bool valid = (uint32_t)(base + e) < last && power <= 0.0f && alpha >= ALPHA_MIN;
            if (__ballot(valid) == 0) continue;
            const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
            const float Tn = T * inv;
            const float dch = alpha * Tn;
            const float a0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
            const float a1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
            const float a2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
            float dL_dalpha = ((r1.z - a0) * dL0 + (r1.w - a1) * dL1 + (cb - a2) * dL2) * Tn;
            dL_dalpha += (-T_final * inv) * bg_dot;
            const float dL_dG = r1.y * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
            const float dG_ddelx = -gdx * r0.z - gdy * r0.w;
            const float dG_ddely = -gdy * r1.x - gdx * r0.w;
            float v[9];
            v[0] = valid ? dL_dG * dG_ddelx * ddelx_dx : 0.f;
            v[1] = valid ? dL_dG * dG_ddely * ddely_dy : 0.f;
            v[2] = valid ? -0.5f * gdx * dx * dL_dG : 0.f;
            v[3] = valid ? -0.5f * gdx * dy * dL_dG : 0.f;
            v[4] = valid ? -0.5f * gdy * dy * dL_dG : 0.f;
            v[5] = valid ? G * dL_dalpha : 0.f;
            v[6] = valid ? dch * dL0 : 0.f;
            v[7] = valid ? dch * dL1 : 0.f;
            v[8] = valid ? dch * dL2 : 0.f;
            if (valid) {
                T = Tn;
                acc0 = a0; acc1 = a1; acc2 = a2;
                lc0 = r1.z; lc1 = r1.w; lc2 = cb;
                last_alpha = alpha;
            }
            // ---- 64-lane reduce-scatter of v[0..7], plain reduction of v[8] ----
            float a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float keep = p0 ? v[4 + k] : v[k];
                const float send = p0 ? v[k] : v[4 + k];
                a[k] = keep + dpp_mov<0xB1>(send);
            }
            float r[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float keep = p1 ? a[2 + k] : a[k];
                const float send = p1 ? a[k] : a[2 + k];
                r[k] = keep + dpp_mov<0x4E>(send);
                r[k] += dpp_mov<0x124>(r[k]);
                r[k] += dpp_mov<0x128>(r[k]);
                r[k] = cross_row_allreduce(r[k]);
            }
            const float v8 = wave_allreduce_sum(v[8]);
            // lanes 0..3 hold the totals of components vbase, vbase+1 (0,1 | 4,5 | 2,3 | 6,7); lane 4 adds #8
            float* gdst = grad_rec + (size_t)s_id[e] * GRAD_REC_FLOATS;
            if (lane < 4) unsafeAtomicAdd(gdst + vbase, r[0]);
            if (lane < 5) unsafeAtomicAdd(gdst + (lane == 4 ? 8 : vbase + 1), lane == 4 ? v8 : r[1]);
        }
    }
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

hipError_t launch_blend_forward(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                float* out_color, float* out_ps, float* out_depth, float* final_T,
                                uint32_t* n_contrib, hipStream_t s) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0) return hipSuccess;
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom);   // GeomLayout::rec == 0
    hipLaunchKernelGGL(blend_forward_kernel, dim3(tiles), dim3(256), 0, s, vp, rec, ids, ranges, out_color, out_ps,
                       out_depth, final_T, n_contrib);
    return hipGetLastError();
}

hipError_t launch_blend_backward(const ViewParams& vp, const char* geom, const uint32_t* ids, const uint2* ranges,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 float* grad_rec, hipStream_t s) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0) return hipSuccess;
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom);
    hipLaunchKernelGGL(blend_backward_kernel, dim3(tiles), dim3(256), 0, s, vp, rec, ids, ranges, final_T, n_contrib,
                       dL_dcolor, grad_rec);
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
