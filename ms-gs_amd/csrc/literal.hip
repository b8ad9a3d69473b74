// literal.hip — the VERIFICATION mode of the library (msgs_set_deterministic(1)): the reference algorithm evaluated literally.
//
// The product kernels re-associate the per-pixel arithmetic (log2-domain conic, fused exponent, one scalar colour recurrence,
// monomial sums: blend.hip) — faithful to the algorithm, but a DIFFERENT float32 evaluation of it, and the per-Gaussian
// backward (K8) amplifies a one-ulp difference in the nine per-Gaussian sums 100-850x: two float32 evaluations cannot agree to
// 1e-4 on dL/dscaling and dL/drotation (tests/test_k8_isolation_gpu.py).  This translation unit is compiled with
// -ffp-contract=off and restates the reference's expressions operation for operation, in the order of SURVEY App. A.1-A.3 as the
// CPU oracle restates them (oracle/msgs_oracle.cpp, the float32 checker, contraction off):
//   * preprocess_kernel / preprocess_backward_kernel (preprocess.hip) are compiled without contraction in every mode and mirror
//     the oracle's expressions already (radii, rects and depth keys are bit-equal to it; K8 + K9 fed the oracle's own sums agree
//     with it to 1e-6: tests/test_k8_isolation_gpu.py); in this mode the forward also leaves the RAW conic and the effective
//     opacity per Gaussian (GeomLayout::litrec) and the backward takes the nine textbook sums as they are;
//   * blend_forward_literal_kernel: one thread per pixel, power = -0.5 (A dx dx + C dy dy) - B dx dy, alpha = min(0.99, o G),
//     T (1 - alpha) < 1e-4 stops WITHOUT blending, C += rgb (alpha T);
//   * blend_backward_literal_kernel: the textbook backward — T /= (1 - alpha), the three accum_rec recurrences, the background
//     term, dL/dG, the nine per-(pixel, Gaussian) products in float32 — summed in DOUBLE over the tile's pixels in a fixed
//     order and stored per tile entry; the entries of a Gaussian are then added in ascending tile order (det_reduce_kernel,
//     blend.hip) and the per-Gaussian backward takes the nine textbook sums as they are.
// G = exp(power) is evaluated in DOUBLE and rounded to float once on both sides (msgs_oracle_set_exp_double): two float32
// exponentials differ by an ulp on a third of their arguments, two double exponentials rounded once practically never.
// With identical inputs the two sides then take the same float for every (pixel, Gaussian) term; what is left between them is the
// order of double additions (1e-16) and K8 + K9's own float32 arithmetic on identical sums (1e-6).  Asserted at 1e-4 flat — the
// north star's sentence as written — on all seven gradient tensors at C2 / C3 / C5 / C4 (tests/test_literal_gpu.py).
// Bitwise reproducible run to run by construction.  Slow (a verification mode): about 10x the default backward.
#include "msgs_internal.h"

#pragma clang fp contract(off)      // (and -ffp-contract=off on the command line, ms-gs_amd/Makefile)

namespace msgs {

namespace {

constexpr int LB = 256;             // threads per tile = pixels per tile = records per staged batch

__device__ __forceinline__ float literal_exp(float p) { return (float)exp((double)p); }

// one thread per pixel: thread t <-> pixel (t & 15, t >> 4) of the tile
__global__ __launch_bounds__(LB) void blend_forward_literal_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                                   const float4* __restrict__ litrec,
                                                                   const uint32_t* __restrict__ ids,
                                                                   const uint2* __restrict__ ranges,
                                                                   float* __restrict__ out_color, float* __restrict__ out_ps,
                                                                   float* __restrict__ out_depth, float* __restrict__ final_T,
                                                                   uint32_t* __restrict__ n_contrib,
                                                                   uint32_t* __restrict__ order_flag) {
    __shared__ float2 s_xy[LB];
    __shared__ float4 s_con[LB];            // conic A, B, C, effective opacity
    __shared__ float4 s_col[LB];            // r, g, b, depth
    __shared__ float s_ps[LB];
    const int tile = blockIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x;
    const int x = tx * TILE + (tid & 15), y = ty * TILE + (tid >> 4);
    const bool inside = x < vp.W && y < vp.H;
    const float pxf = (float)x, pyf = (float)y;
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    if (tile == 0 && tid == 0 && order_flag) *order_flag = 0u;          // (no backward launch order in this mode)

    float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, aps = 0.f, adp = 0.f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;
    for (int base = 0; base < len; base += LB) {
        if (__syncthreads_and(done)) break;
        const int n = min(LB, len - base);
        if (tid < n) {
            const uint32_t id = ids[range.x + base + tid];
            const float4 r0 = rec[id].r0, r1 = rec[id].r1, r2 = rec[id].r2;
            s_xy[tid] = make_float2(r0.x, r0.y);
            s_con[tid] = litrec[id];
            s_col[tid] = make_float4(r1.z, r1.w, r2.x, r2.y);
            s_ps[tid] = r2.z;
        }
        __syncthreads();
        for (int j = 0; j < n && !done; ++j) {
            ++contributor;
            const float4 con = s_con[j];
            const float dx = s_xy[j].x - pxf, dy = s_xy[j].y - pyf;
            const float power = -0.5f * (con.x * dx * dx + con.z * dy * dy) - con.y * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, con.w * literal_exp(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            const float wgt = alpha * T;
            const float4 col = s_col[j];
            C[0] += col.x * wgt; C[1] += col.y * wgt; C[2] += col.z * wgt;
            aps += s_ps[j] * wgt;
            adp += col.w * wgt;
            T = test_T;
            last = contributor;
        }
    }
    if (inside) {
        const size_t N = (size_t)vp.W * vp.H;
        const size_t pix = (size_t)y * vp.W + x;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C[0] + T * vp.bg[0];
        out_color[N + pix] = C[1] + T * vp.bg[1];
        out_color[2 * N + pix] = C[2] + T * vp.bg[2];
        out_ps[pix] = aps;
        out_depth[pix] = adp;
    }
}

// 64-lane sum of a double in a fixed tree (every lane receives the total)
__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, off), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), off);
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return v;
}

// The textbook backward per pixel; the nine float32 products of every (pixel, entry) pair enter DOUBLE sums over the tile's 256
// pixels — lanes of a wave in a fixed butterfly, the four waves in order — stored at inst_grad[range.x + position][0..8].
constexpr int LBB = 32;             // entries per backward batch (4 waves x 9 doubles each in LDS)
__global__ __launch_bounds__(LB) void blend_backward_literal_kernel(ViewParams vp, const GaussRec* __restrict__ rec,
                                                                    const float4* __restrict__ litrec,
                                                                    const uint32_t* __restrict__ ids,
                                                                    const uint2* __restrict__ ranges,
                                                                    const float* __restrict__ final_T,
                                                                    const uint32_t* __restrict__ n_contrib,
                                                                    const float* __restrict__ dL_dcolor,
                                                                    double* __restrict__ inst_grad) {
    __shared__ float2 s_xy[LBB];
    __shared__ float4 s_con[LBB];
    __shared__ float4 s_col[LBB];
    __shared__ double s_part[LBB][4][9];
    __shared__ uint32_t s_wmax[4];
    const int tile = blockIdx.x;
    const int tx = tile % vp.gx, ty = tile / vp.gx;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x = tx * TILE + (tid & 15), y = ty * TILE + (tid >> 4);
    const bool inside = x < vp.W && y < vp.H;
    const float pxf = (float)x, pyf = (float)y;
    const uint2 range = ranges[tile];
    const size_t N = (size_t)vp.W * vp.H;
    const size_t pix = (size_t)y * vp.W + x;
    const float T_final = inside ? final_T[pix] : 1.0f;
    float T = T_final;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float dL_dpixel[3] = {0.f, 0.f, 0.f};
    if (inside) { dL_dpixel[0] = dL_dcolor[pix]; dL_dpixel[1] = dL_dcolor[N + pix]; dL_dpixel[2] = dL_dcolor[2 * N + pix]; }
    float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f}, last_alpha = 0.f;
    const float ddelx_dx = 0.5f * vp.W, ddely_dy = 0.5f * vp.H;
    const float bg[3] = {vp.bg[0], vp.bg[1], vp.bg[2]};

    uint32_t m = last;
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if (lane == 0) s_wmax[wv] = m;
    __syncthreads();
    const int tile_last = (int)max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));

    for (int hi = tile_last; hi > 0; hi -= LBB) {
        const int lo = max(0, hi - LBB), n = hi - lo;
        __syncthreads();                                            // (previous batch written out)
        if (tid < n) {
            const uint32_t id = ids[range.x + lo + tid];
            const float4 r0 = rec[id].r0, r1 = rec[id].r1, r2 = rec[id].r2;
            s_xy[tid] = make_float2(r0.x, r0.y);
            s_con[tid] = litrec[id];
            s_col[tid] = make_float4(r1.z, r1.w, r2.x, 0.f);
        }
        __syncthreads();
        for (int e = n - 1; e >= 0; --e) {                          // back to front
            const uint32_t j = (uint32_t)(lo + e);                  // 0-based position in the tile's list
            float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (j < last) {
                const float4 con = s_con[e];
                const float dx = s_xy[e].x - pxf, dy = s_xy[e].y - pyf;
                const float power = -0.5f * (con.x * dx * dx + con.z * dy * dy) - con.y * dx * dy;
                if (!(power > 0.0f)) {
                    const float G = literal_exp(power);
                    const float alpha = fminf(0.99f, con.w * G);
                    if (!(alpha < 1.0f / 255.0f)) {
                        T = T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * T;
                        float dL_dalpha = 0.0f;
                        float dcol[3];
                        const float rgb[3] = {s_col[e].x, s_col[e].y, s_col[e].z};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            accum_rec[c] = last_alpha * last_color[c] + (1.f - last_alpha) * accum_rec[c];
                            last_color[c] = rgb[c];
                            dL_dalpha += (rgb[c] - accum_rec[c]) * dL_dpixel[c];
                            dcol[c] = dchannel_dcolor * dL_dpixel[c];
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) bg_dot += bg[c] * dL_dpixel[c];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                        const float dL_dG = con.w * dL_dalpha;      // Q6: through the clamp
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * con.x - gdy * con.y;
                        const float dG_ddely = -gdy * con.z - gdx * con.y;
                        v[0] = dL_dG * dG_ddelx * ddelx_dx;
                        v[1] = dL_dG * dG_ddely * ddely_dy;
                        v[2] = -0.5f * gdx * dx * dL_dG;
                        v[3] = -0.5f * gdx * dy * dL_dG;
                        v[4] = -0.5f * gdy * dy * dL_dG;
                        v[5] = G * dL_dalpha;
                        v[6] = dcol[0]; v[7] = dcol[1]; v[8] = dcol[2];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double t = wave_sum_f64((double)v[k]);
                if (lane == 0) s_part[e][wv][k] = t;
            }
        }
        __syncthreads();
        for (int q = tid; q < n * 9; q += LB) {
            const int e = q / 9, k = q - 9 * e;
            inst_grad[((size_t)range.x + lo + e) * DET_INST_FLOATS + k] = ((s_part[e][0][k] + s_part[e][1][k]) + s_part[e][2][k]) + s_part[e][3][k];
        }
    }
}

}  // namespace

hipError_t launch_blend_forward_literal(const ViewParams& vp, const char* geom, int P, const uint32_t* ids, const uint2* ranges,
                                        float* out_color, float* out_ps, float* out_depth, float* final_T, uint32_t* n_contrib,
                                        uint32_t* order_flag, hipStream_t s) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0) return hipSuccess;
    const GeomLayout L(P > 0 ? P : 1);
    hipLaunchKernelGGL(blend_forward_literal_kernel, dim3(tiles), dim3(LB), 0, s, vp, reinterpret_cast<const GaussRec*>(geom + L.rec),
                       reinterpret_cast<const float4*>(geom + L.litrec), ids, ranges, out_color, out_ps, out_depth, final_T, n_contrib,
                       order_flag);
    return hipGetLastError();
}

hipError_t launch_blend_backward_literal(const ViewParams& vp, const char* geom, int P, const uint32_t* ids, const uint2* ranges,
                                         const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor, double* inst_grad,
                                         hipStream_t s) {
    const int tiles = vp.gx * vp.gy;
    if (tiles == 0) return hipSuccess;
    const GeomLayout L(P > 0 ? P : 1);
    hipLaunchKernelGGL(blend_backward_literal_kernel, dim3(tiles), dim3(LB), 0, s, vp, reinterpret_cast<const GaussRec*>(geom + L.rec),
                       reinterpret_cast<const float4*>(geom + L.litrec), ids, ranges, final_T, n_contrib, dL_dcolor, inst_grad);
    return hipGetLastError();
}

}  // namespace msgs
