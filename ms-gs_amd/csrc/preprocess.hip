// preprocess.hip — per-Gaussian stages of the rasterizer for gfx950.
//
//   preprocess_kernel           K1: projection, EWA splat, SH->RGB, MS-GS pixel size and filters,
//                                   exact tile-overlap count            
//   preprocess_backward_kernel  K8+K9: 2-D covariance, projection, SH and 3-D covariance backward
//                                                                      
//   mark_visible_kernel         K10
//
// Replaces the per-Gaussian kernels of the reference's un-vendored CUDA module
// (/root/reference/.gitmodules:4-6; call site gaussian_renderer/__init__.py:94-108).
//
// Both kernels are HBM-streaming (309 B in / ~80 B out per Gaussian forward at SH degree 3).
// One thread per Gaussian; floating-point contraction is OFF in this file so that every float32
// operation is individually rounded and the discrete outputs (radii, tile rects, depth sort keys)
// are bit-reproducible against the CPU oracle.
#include "msgs_internal.h"

#pragma clang fp contract(off)

namespace msgs {

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Cam {
    float V[16];
    float M[16];
    float cam[3];
};

__device__ __forceinline__ void load_cam(const ViewParams& vp, Cam& c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { c.V[i] = vp.viewmatrix[i]; c.M[i] = vp.projmatrix[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) c.cam[i] = vp.campos[i];
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* cov) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    const float S[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * S[j];
    cov[0] = M[0][0] * M[0][0] + M[0][1] * M[0][1] + M[0][2] * M[0][2];
    cov[1] = M[0][0] * M[1][0] + M[0][1] * M[1][1] + M[0][2] * M[1][2];
    cov[2] = M[0][0] * M[2][0] + M[0][1] * M[2][1] + M[0][2] * M[2][2];
    cov[3] = M[1][0] * M[1][0] + M[1][1] * M[1][1] + M[1][2] * M[1][2];
    cov[4] = M[1][0] * M[2][0] + M[1][1] * M[2][1] + M[1][2] * M[2][2];
    cov[5] = M[2][0] * M[2][0] + M[2][1] * M[2][1] + M[2][2] * M[2][2];
}

struct Cov2D {
    float T[2][3];
    float a, b, c;
    float tx_c, ty_c, tz, x_mul, y_mul;
};

__device__ __forceinline__ void compute_cov2d(const float* t, const ViewParams& vp, const float* cov3D,
                                              const float* V, Cov2D& o) {
    const float limx = 1.3f * vp.tanfovx, limy = 1.3f * vp.tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    o.x_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    o.y_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    o.tx_c = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    o.ty_c = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    o.tz = t[2];
    const float J[2][3] = {{vp.fx / t[2], 0.f, -(vp.fx * o.tx_c) / (t[2] * t[2])},
                           {0.f, vp.fy / t[2], -(vp.fy * o.ty_c) / (t[2] * t[2])}};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            o.T[r][c] = J[r][0] * V[4 * c + 0] + J[r][1] * V[4 * c + 1] + J[r][2] * V[4 * c + 2];
    const float S[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]},
                           {cov3D[2], cov3D[4], cov3D[5]}};
    float ST[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i) ST[r][i] = S[i][0] * o.T[r][0] + S[i][1] * o.T[r][1] + S[i][2] * o.T[r][2];
    o.a = (o.T[0][0] * ST[0][0] + o.T[0][1] * ST[0][1] + o.T[0][2] * ST[0][2]) + 0.3f;
    o.b = o.T[0][0] * ST[1][0] + o.T[0][1] * ST[1][1] + o.T[0][2] * ST[1][2];
    o.c = (o.T[1][0] * ST[1][0] + o.T[1][1] * ST[1][1] + o.T[1][2] * ST[1][2]) + 0.3f;
}

__device__ __forceinline__ void view_point(const float* V, const float* p, float* t) {
    t[0] = ((V[0] * p[0] + V[4] * p[1]) + V[8] * p[2]) + V[12];
    t[1] = ((V[1] * p[0] + V[5] * p[1]) + V[9] * p[2]) + V[13];
    t[2] = ((V[2] * p[0] + V[6] * p[1]) + V[10] * p[2]) + V[14];
}

__device__ __forceinline__ void proj_point(const float* M, const float* p, float* h) {
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = ((M[k] * p[0] + M[4 + k] * p[1]) + M[8 + k] * p[2]) + M[12 + k];
}

// MS-GS pixel size, DESIGN.md SPEC M1
__device__ __forceinline__ float pixel_size_of(float opacity, float conA, float conC) {
    const float v = 255.0f * opacity;
    if (!(v > 1.0f) || !(conA > 0.f) || !(conC > 0.f)) return 0.f;
    const float ell = 2.0f * logf(v);
    const float sx = 2.0f * sqrtf(ell / conA);
    const float sy = 2.0f * sqrtf(ell / conC);
    return fminf(sx, sy);
}

// MS-GS filter weight, DESIGN.md SPEC M2-M4
__device__ __forceinline__ float filter_weight(const ViewParams& vp, float size, float minps, float maxps,
                                               bool base) {
    float w = 1.0f;
    if (vp.filter_small && !base && minps > 0.f && size < minps) {
        if (vp.fade_size > 0.f) {
            const float rel = minps / fmaxf(size, 1e-30f);
            w *= fminf(1.f, fmaxf(0.f, 1.f - (rel - 1.f) / vp.fade_size));
        } else w = 0.f;
    }
    if (vp.filter_large && maxps > 0.f && size > maxps) {
        if (vp.fade_size > 0.f) {
            const float rel = size / maxps;
            w *= fminf(1.f, fmaxf(0.f, 1.f - (rel - 1.f) / vp.fade_size));
        } else w = 0.f;
    }
    return w;
}

__device__ __forceinline__ float sh_channel(int deg, const float* sh, int c, float x, float y, float z) {
    auto S = [&](int k) { return sh[k * 3 + c]; };
    float r = SH_C0 * S(0);
    if (deg > 0) {
        r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
                SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
            if (deg > 2) {
                r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                    SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                    SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                    SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                    SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
            }
        }
    }
    return r;
}


// The same polynomial as sh_channel, factored as colour_c = sum_k basis[k] * sh[k][c] with the terms accumulated in
// the same order k = 0..15 (and the same products: (C * poly) * coefficient), so that the coefficients can arrive in
// two halves without changing a single bit of the result.
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float b[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = 0.f;
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -(SH_C1 * y); b[2] = SH_C1 * z; b[3] = -(SH_C1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.0f * xx - yy); b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy); b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative movement of 192-byte SH rows (K = 16 coefficients x 3 channels) between HBM and
// LDS.  A thread-per-Gaussian float4 access at a 192-B stride touches 64 different cache lines per
// instruction (TA-bound, measured 25 % of HBM rate); here n4 consecutive lanes move one row's n4
// float4 so every instruction covers 64/n4 whole rows.  LDS rows have a stride of 49 floats so the
// later one-thread-per-row 4-byte accesses are bank-conflict free.
// ---------------------------------------------------------------------------------------------
// LDS hand-off between lanes of ONE wave: the hardware executes a wave's LDS operations in order; this
// keeps the compiler from reordering across the phase boundary as well.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int ROW_F = 48;          // floats per SH row in HBM (K = 16)
constexpr int ROW_LDS = 49;        // floats per row in LDS
constexpr int K9_STAGE_ROWS = 32;  // Gaussians of a wave whose SH rows are in LDS at a time (preprocess_backward_kernel)

__device__ __forceinline__ int sh_row_float4s(int deg) { return deg == 0 ? 1 : deg == 1 ? 3 : deg == 2 ? 7 : 12; }

// rows listed in idx[0..nrow) (lane numbers inside the wave) are loaded from g_rows + lane*48 into LDS row (lane - row0)
__device__ __forceinline__ void coop_load_rows(float* lds_rows, const float* g_rows, const uint8_t* idx, int nrow,
                                               int n4, int lane, int row0 = 0) {
    const int rpi = 64 / n4;
    const int sub = lane / n4, c = lane - sub * n4;
    for (int it = 0; it * rpi < nrow; ++it) {
        const int slot = it * rpi + sub;
        if (sub < rpi && slot < nrow) {
            const int sl = idx[slot];
            const float4 v = *reinterpret_cast<const float4*>(g_rows + (size_t)sl * ROW_F + 4 * c);
            float* d = lds_rows + (sl - row0) * ROW_LDS + 4 * c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
}

// half rows: float4s [f4_first, f4_first + n4) of every listed row into LDS rows of HALF_LDS floats.  K1 keeps only
// 24 coefficients-floats per Gaussian in LDS at a time: 25.6 KB per workgroup instead of 50 KB, i.e. six workgroups per CU
// instead of three — the kernel is latency-bound and its time follows 1/occupancy (measured 87 -> 115 us with two)
constexpr int HALF_F4 = 6;             // float4s per half row
constexpr int HALF_LDS = 25;           // floats per half row in LDS (odd: conflict-free one-thread-per-row reads)
__device__ __forceinline__ void coop_load_rows_part(float* lds_rows, const float* g_rows, const uint8_t* idx, int nrow,
                                                    int f4_first, int n4, int lane) {
    const int rpi = 64 / n4;
    const int sub = lane / n4, c = lane - sub * n4;
    for (int it = 0; it * rpi < nrow; ++it) {
        const int slot = it * rpi + sub;
        if (sub < rpi && slot < nrow) {
            const int sl = idx[slot];
            const float4 v = *reinterpret_cast<const float4*>(g_rows + (size_t)sl * ROW_F + 4 * (f4_first + c));
            float* d = lds_rows + sl * HALF_LDS + 4 * c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
}

// rows 0..nrow of the wave are stored to g_rows (all 12 float4 of every row); a row whose bit in
// `live` is clear, and every float at or beyond nfloat, is written as zero.
// acc (msgs_grads_t::accumulate): live rows are ADDED to what g_rows holds, the other rows are not touched
__device__ __forceinline__ void coop_store_rows(const float* lds_rows, float* g_rows, int nrow, uint64_t live,
                                                int nfloat, int lane, bool acc) {
    constexpr int n4 = ROW_F / 4, rpi = 64 / n4;            // 12 float4 per row, 5 rows per instruction
    const int sub = lane / n4, c = lane - sub * n4;
    for (int it = 0; it * rpi < nrow; ++it) {
        const int row = it * rpi + sub;
        if (sub < rpi && row < nrow) {
            const bool on = (live >> row) & 1ull;
            if (acc && !on) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on) {
                const float* sp = lds_rows + row * ROW_LDS + 4 * c;
                v.x = 4 * c + 0 < nfloat ? sp[0] : 0.f;
                v.y = 4 * c + 1 < nfloat ? sp[1] : 0.f;
                v.z = 4 * c + 2 < nfloat ? sp[2] : 0.f;
                v.w = 4 * c + 3 < nfloat ? sp[3] : 0.f;
            }
            float4* dst = reinterpret_cast<float4*>(g_rows + (size_t)row * ROW_F + 4 * c);
            if (acc) { const float4 o = *dst; v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w; }
            *dst = v;
        }
    }
}

// ---- raw-parameter mode (msgs_gaussians_t::raw_params): the activations of the reference's GaussianModel getters
// (scene/gaussian_model.py:39-47,127-153), in the same float32 form torch evaluates them ----
constexpr int REST_F = 45;         // floats per features_rest row (15 coefficients x 3 channels)

__device__ __forceinline__ float act_opacity(const msgs_gaussians_t& g, int i) {
    const float x = g.opacities[i];
    return g.raw_params == 1 ? 1.0f / (1.0f + expf(-x)) : x;          // torch.sigmoid
}
__device__ __forceinline__ void act_scales(const msgs_gaussians_t& g, int i, float* s) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float x = g.scales[3 * i + k]; s[k] = g.raw_params == 1 ? expf(x) : x; }   // torch.exp
}
// q = raw / max(||raw||, 1e-12) (torch.nn.functional.normalize); returns the norm used
__device__ __forceinline__ float act_rotation(const msgs_gaussians_t& g, int i, float* q) {
    const float4 q4 = reinterpret_cast<const float4*>(g.rotations)[i];
    q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    if (g.raw_params == 0) return 1.0f;
    if (g.raw_params == 2) {         // chained mode: q is already normalised; the norm comes from the raw quaternion
        const float4 r4 = reinterpret_cast<const float4*>(g.rotations_raw)[i];
        return fmaxf(sqrtf(r4.x * r4.x + r4.y * r4.y + r4.z * r4.z + r4.w * r4.w), 1e-12f);
    }
    const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = q[k] / n;
    return n;
}

// SH rows of one wave from the split dc / rest parameters into LDS rows [dc(3) | rest(45)] (= the torch.cat of
// gaussian_model.py:144-149): the rest rows of 64 consecutive Gaussians are ONE contiguous, 16-byte aligned run of
// 64*45 floats, moved with full-width float4 loads.
// (lrow: the LDS row of THIS lane's Gaussian i — its lane number, or lane & 31 when 32 Gaussians are staged at a time;
//  wave_first / nrow then describe the staged run)
// Half rows of the SAME concatenated layout for a SPARSE set of the wave's Gaussians, straight from the split dc / rest leaves
// (K1 colours only the Gaussians that survived the culls, 56 % at C3): idx[0..nlisted) = wave-local row numbers; eight lanes serve
// one listed row with 12-byte loads (a rest row is 180 bytes, 4-byte aligned), eight rows per instruction.  half 0 = coefficients
// 0..7 = [dc | rest 0..20], half 1 = coefficients 8..15 = rest 21..44; LDS rows of HALF_LDS floats as in coop_load_rows_part.
typedef float msgs_f3 __attribute__((ext_vector_type(3)));
typedef msgs_f3 __attribute__((aligned(4))) msgs_f3_u;
__device__ __forceinline__ void coop_load_split_half_listed(float* lds_rows, const float* dc, const float* rest, int wave_first,
                                                            const uint8_t* idx, int nlisted, int half, int lane) {
    const int sub = lane >> 3, c = lane & 7;
    for (int it = 0; it * 8 < nlisted; ++it) {
        const int slot = it * 8 + sub;
        if (slot < nlisted) {
            const int sl = idx[slot];
            const size_t gi = (size_t)wave_first + sl;
            float* d = lds_rows + sl * HALF_LDS + 3 * c;
            // half 0: lane c holds coefficient c (c = 0: dc); half 1: coefficient 8 + c
            const float* src = (half == 0 && c == 0) ? dc + gi * 3 : rest + gi * 45 + 3 * (8 * half + c - 1);
            const msgs_f3 v = *reinterpret_cast<const msgs_f3_u*>(src);
            d[0] = v.x; d[1] = v.y; d[2] = v.z;
        }
    }
}

// Whole rows [dc(3) | rest(45)] of the listed Gaussians (K9: the rendered ones) into LDS rows of ROW_LDS floats: sixteen lanes
// serve one row with 12-byte loads (fifteen on the rest row, one on dc), four rows per instruction.  idx[0..nlisted) = wave-local
// row numbers, LDS row = number - row0.
__device__ __forceinline__ void coop_load_split_rows_listed(float* lds_rows, const float* dc, const float* rest, int wave_first,
                                                            const uint8_t* idx, int nlisted, int lane, int row0) {
    const int sub = lane >> 4, c = lane & 15;
    for (int it = 0; it * 4 < nlisted; ++it) {
        const int slot = it * 4 + sub;
        if (slot < nlisted) {
            const int sl = idx[slot];
            const size_t gi = (size_t)wave_first + sl;
            float* d = lds_rows + (sl - row0) * ROW_LDS + 3 * c;
            const float* src = c == 0 ? dc + gi * 3 : rest + gi * REST_F + 3 * (c - 1);
            const msgs_f3 v = *reinterpret_cast<const msgs_f3_u*>(src);
            d[0] = v.x; d[1] = v.y; d[2] = v.z;
        }
    }
}

__device__ __forceinline__ void coop_load_split_rows(float* lds_rows, const float* dc, const float* rest, int i,
                                                     bool in_range, int wave_first, int nrow, int lane, int lrow) {
    if (in_range) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lds_rows[lrow * ROW_LDS + c] = dc[3 * (size_t)i + c];
    }
    const float* src = rest + (size_t)wave_first * REST_F;
    const int nflat = nrow * REST_F;
    for (int base = 0; base < nflat; base += 256) {
        const int f0 = base + lane * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (f0 + 3 < nflat) {
            const float4 t = *reinterpret_cast<const float4*>(src + f0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (f0 + j < nflat) v[j] = src[f0 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = f0 + j;
            if (f < nflat) { const int row = f / REST_F; lds_rows[row * ROW_LDS + 3 + (f - row * REST_F)] = v[j]; }
        }
    }
}

// inverse: gradient rows from LDS to the split dc / rest gradient tensors (zeros for rows not in `live` and for
// coefficients beyond the active degree).  acc: live rows are ADDED to the tensors, the others not touched.
__device__ __forceinline__ void coop_store_split_rows(const float* lds_rows, float* d_dc, float* d_rest, int i,
                                                      bool in_range, int wave_first, int nrow, uint64_t live,
                                                      int nfloat, int lane, int lrow, bool acc = false) {
    if (in_range) {
        const bool on = (live >> lrow) & 1ull;
        if (!acc) {
#pragma unroll
            for (int c = 0; c < 3; ++c) d_dc[3 * (size_t)i + c] = on ? lds_rows[lrow * ROW_LDS + c] : 0.f;
        } else if (on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) d_dc[3 * (size_t)i + c] = d_dc[3 * (size_t)i + c] + lds_rows[lrow * ROW_LDS + c];
        }
    }
    float* dst = d_rest + (size_t)wave_first * REST_F;
    const int nflat = nrow * REST_F;
    for (int base = 0; base < nflat; base += 256) {
        const int f0 = base + lane * 4;
        float v[4];
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = min(f0 + j, nflat - 1);
            const int row = f / REST_F, k = f - row * REST_F;
            on[j] = (live >> row) & 1ull;
            v[j] = (on[j] && 3 + k < nfloat) ? lds_rows[row * ROW_LDS + 3 + k] : 0.f;
        }
        if (f0 + 3 < nflat) {
            float4* d4 = reinterpret_cast<float4*>(dst + f0);
            if (!acc) {
                *d4 = make_float4(v[0], v[1], v[2], v[3]);
            } else if (on[0] || on[3]) {             // four consecutive floats span at most two rows
                const float4 o = *d4;
                *d4 = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < nflat) {
                    if (!acc) dst[f0 + j] = v[j];
                    else if (on[j]) dst[f0 + j] = dst[f0 + j] + v[j];
                }
        }
    }
}

// msgs_adam_in_backward_t: the same walk over the run's dc / rest rows, but instead of storing the gradient rows the Adam step
// is taken on the spot — parameter and both moments of EVERY row of the run are read, updated with the row's gradient (LDS; zero
// for rows not in `live` and for coefficients beyond the active degree: exactly the values coop_store_split_rows would have
// stored for the optimizer kernel to read) and written back.  The run's rest rows are one contiguous, 16-byte aligned block
// (32 rows x 180 B from a multiple of 32 rows; the tensors' bases are aligned, msgs_backward checks).
__device__ __forceinline__ void coop_adam_split_rows(const float* lds_rows, float* p_dc, float* p_rest, const AdamInBackward& ad,
                                                     int i, bool in_range, int wave_first, int nrow, uint64_t live,
                                                     int nfloat, int lane, int lrow) {
    if (in_range) {
        const bool on = (live >> lrow) & 1ull;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t at = 3 * (size_t)i + c;
            float p = p_dc[at], m = ad.m[1][at], v = ad.v[1][at];
            adam_update(p, on ? lds_rows[lrow * ROW_LDS + c] : 0.f, m, v, ad.nss[1], ad.a);
            p_dc[at] = p; ad.m[1][at] = m; ad.v[1][at] = v;
        }
    }
    const size_t off = (size_t)wave_first * REST_F;
    float* __restrict__ P = p_rest + off;
    float* __restrict__ M = ad.m[2] + off;
    float* __restrict__ V = ad.v[2] + off;
    const float nss = ad.nss[2];
    const int nflat = nrow * REST_F;
    for (int base = 0; base < nflat; base += 256) {
        const int f0 = base + lane * 4;
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = min(f0 + j, nflat - 1);
            const int row = f / REST_F, k = f - row * REST_F;
            g[j] = (((live >> row) & 1ull) && 3 + k < nfloat) ? lds_rows[row * ROW_LDS + 3 + k] : 0.f;
        }
        if (f0 + 3 < nflat) {
            float4 p = *reinterpret_cast<const float4*>(P + f0), m = *reinterpret_cast<const float4*>(M + f0),
                   v = *reinterpret_cast<const float4*>(V + f0);
            adam_update(p.x, g[0], m.x, v.x, nss, ad.a);
            adam_update(p.y, g[1], m.y, v.y, nss, ad.a);
            adam_update(p.z, g[2], m.z, v.z, nss, ad.a);
            adam_update(p.w, g[3], m.w, v.w, nss, ad.a);
            *reinterpret_cast<float4*>(P + f0) = p;
            *reinterpret_cast<float4*>(M + f0) = m;
            *reinterpret_cast<float4*>(V + f0) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f0 + j < nflat) {
                    float p = P[f0 + j], m = M[f0 + j], v = V[f0 + j];
                    adam_update(p, g[j], m, v, nss, ad.a);
                    P[f0 + j] = p; M[f0 + j] = m; V[f0 + j] = v;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------
// SPLIT_ROWS = true: SH from the split dc / rest leaves without a concatenated tensor (raw mode 1, or mode 2 without
// `shs`): whole 48-float rows in LDS.  false: every other input form; a concatenated K = 16 `shs` goes through LDS in
// two 24-float halves.
template <bool SPLIT_ROWS>
__global__ __launch_bounds__(256) void preprocess_kernel(ViewParams vp, msgs_gaussians_t g,
                                                         int32_t* __restrict__ radii,
                                                         float* __restrict__ pixel_sizes,
                                                         char* __restrict__ geom, ZeroJob zj,
                                                         uint32_t* __restrict__ heavy_list,
                                                         uint32_t* __restrict__ heavy_count,
                                                         uint32_t* __restrict__ heavy_blk, int write_litrec) {
    __shared__ float s_rows[4][64 * HALF_LDS];
    __shared__ uint8_t s_idx[4][64];
    __shared__ uint32_t s_heavy[4], s_hvw[4];
    const int P = g.P;
    {   // housekeeping for the depth sort that follows: clear its group-sum table (one word per thread)
        const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
        for (size_t t = t0; t < zj.n0; t += nt) zj.p0[t] = 0u;
        for (size_t t = t0; t < zj.n1; t += nt) zj.p1[t] = 0u;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool in_range = i < P;
    const GeomLayout L(P);
    GaussRec* rec = reinterpret_cast<GaussRec*>(geom + L.rec);
    BinRec* binrec = reinterpret_cast<BinRec*>(geom + L.binrec);
    uint32_t* tiles = reinterpret_cast<uint32_t*>(geom + L.tiles);
    uint32_t* key = reinterpret_cast<uint32_t*>(geom + L.key);
    uint32_t* flags = reinterpret_cast<uint32_t*>(geom + L.flags);
    float* weight = reinterpret_cast<float*>(geom + L.weight);

    Cam cm;
    load_cam(vp, cm);

    int32_t out_radius = 0;
    float out_psize = 0.f;
    uint32_t out_tiles = 0, out_cells = 0, out_key = 0xFFFFFFFFu, out_flags = 0;
    float out_weight = 0.f, out_tau2 = -3.0e38f;

    // ---- phase A: geometry, pixel size, multi-scale filters (one lane per Gaussian) ----
    bool alive = false;
    float p[3] = {0.f, 0.f, 1.f}, t[3] = {0.f, 0.f, 0.f};
    float conA = 0.f, conB = 0.f, conC = 0.f, px = 0.f, py = 0.f, my_radius = 0.f, w = 0.f, o_eff = 0.f;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (in_range) {
        p[0] = g.means3D[3 * i]; p[1] = g.means3D[3 * i + 1]; p[2] = g.means3D[3 * i + 2];
        view_point(cm.V, p, t);
    }
    do {
        if (!in_range || t[2] <= 0.2f) break;                       // Q1 near-plane cull
        float h[4];
        proj_point(cm.M, p, h);
        const float pw = 1.0f / (h[3] + 0.0000001f);
        const float ndc_x = h[0] * pw, ndc_y = h[1] * pw;
        float cov3D[6];
        if (g.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) cov3D[k] = g.cov3D_precomp[6 * i + k];
        } else {
            float sc[3], q[4];
            act_scales(g, i, sc);
            act_rotation(g, i, q);
            cov3d_from_scale_rot(sc, vp.scale_modifier, q, cov3D);
        }
        Cov2D c2;
        compute_cov2d(t, vp, cov3D, cm.V, c2);
        const float det = c2.a * c2.c - c2.b * c2.b;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        conA = c2.c * det_inv; conB = -c2.b * det_inv; conC = c2.a * det_inv;
        const float mid = 0.5f * (c2.a + c2.c);
        const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lam1 = mid + root, lam2 = mid - root;
        my_radius = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
        px = ((ndc_x + 1.0f) * vp.W - 1.0f) * 0.5f;
        py = ((ndc_y + 1.0f) * vp.H - 1.0f) * 0.5f;
        const float o = act_opacity(g, i);
        out_psize = pixel_size_of(o, conA, conC);                    // SPEC M1 (before any filter)
        minx = min(vp.gx, max(0, (int)((px - my_radius) / TILE)));
        miny = min(vp.gy, max(0, (int)((py - my_radius) / TILE)));
        maxx = min(vp.gx, max(0, (int)((px + my_radius + TILE - 1) / TILE)));
        maxy = min(vp.gy, max(0, (int)((py + my_radius + TILE - 1) / TILE)));
        if ((maxx - minx) * (maxy - miny) == 0) break;
        w = filter_weight(vp, out_psize, g.min_pixel_sizes ? g.min_pixel_sizes[i] : -1.f,
                          g.max_pixel_sizes ? g.max_pixel_sizes[i] : -1.f,
                          g.base_mask ? g.base_mask[i] != 0 : false);
        if (!(w > 0.f)) break;                                       // SPEC M2/M3: dropped, radii stays 0
        o_eff = o * w;
        alive = true;
    } while (false);

    // ---- phase B + colour: SH rows of the surviving Gaussians, HBM -> LDS (coalesced per row) -> SH->RGB ----
    const bool raw = g.raw_params != 0;
    // chained mode may also hand over the concatenated `shs` the reference built: 16-byte aligned 192-byte rows, of
    // which only the surviving Gaussians' are fetched (the split leaves are read as one run per wave, all 64 rows)
    const bool split_in = SPLIT_ROWS && raw && g.shs == nullptr;
    const bool staged_sh = raw || (g.shs != nullptr && vp.sh_coeffs == 16);   // wave-uniform
    float rgb[3] = {0.f, 0.f, 0.f};
    float dirx = 0.f, diry = 0.f, dirz = 0.f;
    if (alive) {
        const float dx = p[0] - cm.cam[0], dy = p[1] - cm.cam[1], dz = p[2] - cm.cam[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dirx = dx / len; diry = dy / len; dirz = dz / len;
    }
    if (split_in || staged_sh) {
        const uint64_t need = __ballot(alive);
        if (alive) s_idx[wv][__popcll(need & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        const int wave_first = blockIdx.x * blockDim.x + wv * 64;
        const int ncoef = (vp.sh_degree + 1) * (vp.sh_degree + 1);
        float basis[16];
        sh_basis(vp.sh_degree, dirx, diry, dirz, basis);
        const int n4_total = sh_row_float4s(vp.sh_degree);
        for (int half = 0; half * HALF_F4 < n4_total; ++half) {
            wave_lds_fence();                              // s_idx visible / previous half consumed
            if (split_in) {
                if (need != 0)
                    coop_load_split_half_listed(s_rows[wv], g.features_dc, g.features_rest, wave_first, s_idx[wv], __popcll(need),
                                                half, lane);
            } else {
                coop_load_rows_part(s_rows[wv], g.shs + (size_t)wave_first * ROW_F, s_idx[wv], __popcll(need),
                                    half * HALF_F4, min(HALF_F4, n4_total - half * HALF_F4), lane);
            }
            wave_lds_fence();
            if (alive) {
                const float* row = (const float*)&s_rows[wv][lane * HALF_LDS];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int kk = 8 * half + k;
                    if (kk < ncoef) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] + basis[kk] * row[3 * k + c];
                    }
                }
            }
        }
    } else if (alive && !(g.colors_precomp && !raw)) {
        const float* sh = g.shs + (size_t)3 * vp.sh_coeffs * i;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sh_channel(vp.sh_degree, sh, c, dirx, diry, dirz);
    }

    // ---- phase C: clamp, exact tile-overlap count, record ----
    if (alive) {
        if (g.colors_precomp && !raw) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = g.colors_precomp[3 * i + c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float r = rgb[c] + 0.5f;
                if (r < 0.0f) { out_flags |= (1u << c); r = 0.0f; }
                rgb[c] = r;
            }
        }
        // exact tile-overlap count (this build's replacement for tiles_touched = rect area): a tile
        // is kept only if the alpha >= 1/255 level set reaches it (per-row analytic extents).
        constexpr float KLOG = -0.72134752044448170368f;            // -1/2 log2(e)
        const float sA = KLOG * conA, sBh = KLOG * conB, sC = KLOG * conC;
        const float v255 = 255.0f * o_eff;
        float tau2 = 1.0f;                                           // > 0: unreachable
        uint32_t count = 0;
        const bool nd = sA < 0.f && sC < 0.f && (sA * sC - sBh * sBh) > 0.f;
        if (v255 > 1.0f) {
            tau2 = -__log2f(v255);
            tau2 = tau2 - (1e-5f * fabsf(tau2) + 2e-3f);
            if (nd) {
                const LevelSetRows ls = levelset_rows_setup(sA, sBh, sC, tau2);
                for (int ty = miny; ty < maxy; ++ty) {
                    int tlo, thi;
                    if (levelset_row_interval(ls, px, py, ty, minx, maxx, LEVELSET_MARGIN_COUNT, tlo, thi))
                        count += (uint32_t)(thi - tlo + 1);
                }
            } else {
                tau2 = -3.0e38f;                                     // cannot bound: keep the whole rect
                count = (uint32_t)((maxx - minx) * (maxy - miny));
            }
        }
        if (write_litrec)       // verification mode: what the literal blend loops read (literal.hip)
            reinterpret_cast<float4*>(geom + L.litrec)[i] = make_float4(conA, conB, conC, o_eff);
        rec[i].r0 = make_float4(px, py, sA, sBh);
        rec[i].r1 = make_float4(sC, __log2f(o_eff), rgb[0], rgb[1]);
        rec[i].r2 = make_float4(rgb[2], t[2], out_psize, tau2);
        if (count) {
            binrec[i].q0 = make_float4(px, py, sA, sBh);
            binrec[i].q1 = make_float4(sC, tau2, __uint_as_float((uint32_t)minx | ((uint32_t)miny << 16)),
                                       __uint_as_float((uint32_t)maxx | ((uint32_t)maxy << 16)));
            if (vp.cell_sx >= 0)          // the coarse cells of the rect ride above the count (msgs_internal.h, cell_range_pack)
                out_cells = cell_range_pack(minx, miny, maxx, maxy, vp.cell_sx, vp.cell_sy) << TILE_COUNT_BITS;
        }
        out_radius = (int32_t)my_radius;
        out_tau2 = tau2;
        out_tiles = count;
        out_key = count ? __float_as_uint(t[2]) : 0xFFFFFFFFu;
        out_flags |= 8u;
        out_weight = w;
    }
    if (in_range) {
        radii[i] = out_radius;
        pixel_sizes[i] = out_psize;
        tiles[i] = out_tiles | out_cells;
        key[i] = out_key;
        flags[i] = out_flags;
        weight[i] = out_weight;
    }
    // cover candidates of the occlusion cut-off (occlusion.hip): the Gaussians with many tile instances, per wave — no
    // atomics, no counter to clear; occ_gather_kernel compacts the lists.  (A form that cannot be bounded, tau2 = -3e38, is
    // never a candidate: its level set is not an ellipse.)
    if (heavy_count) {
        const bool hv = out_tiles > OCC_HEAVY_MIN && out_tau2 > -1.0e38f;
        const uint64_t m = __ballot(hv);
        const size_t slot = (size_t)blockIdx.x * 4 + wv;
        if (hv) reinterpret_cast<uint2*>(heavy_list)[slot * 64 + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)i, out_key);
        // ... and per workgroup {candidates, sum of their largest possible cover weights}: what the occlusion pass adds up to
        // learn whether the view has candidates that could close anything at all.  A cover's weight over any block is at most
        // the one at its own centre, -log2(1 - min(0.99, opacity)) in the pass's fixed point (OCC_FIX = 2048), rounded UP
        uint32_t wmax = hv ? (uint32_t)(-__log2f(1.0f - fminf(0.99f, o_eff)) * 2048.0f) + 2u : 0u;
        for (int off = 32; off > 0; off >>= 1) wmax += (uint32_t)__shfl_xor((int)wmax, off);
        if (lane == 0) { heavy_count[slot] = (uint32_t)__popcll(m); s_heavy[wv] = (uint32_t)__popcll(m); s_hvw[wv] = wmax; }
        __syncthreads();
        if (threadIdx.x == 0)
            reinterpret_cast<uint2*>(heavy_blk)[blockIdx.x] = make_uint2(s_heavy[0] + s_heavy[1] + s_heavy[2] + s_heavy[3],
                                                                         min(s_hvw[0] + s_hvw[1] + s_hvw[2] + s_hvw[3], 0x0FFFFFFFu));
    }
}

// ---------------------------------------------------------------------------------------------
// K8 + K9
// ---------------------------------------------------------------------------------------------
// (held to 96 registers = 5 waves per SIMD: measured 116 us with the compiler's 102 registers / 4 waves, 112 us with 5 waves and
//  one spilled register, 125 us with 6 waves and 22 spills; 121 us before the rows were staged in two runs)
// TEXTBOOK = true (msgs_backward_per_gaussian, the K8 + K9 isolation entry of the parity tests): grad_rec is NOT this
// library's record of monomial sums but [P,9] doubles holding the textbook 2-D gradients {dL/dmean2D x, y (NDC-ish units),
// dL/dconic A, B, C, dL/dopacity_eff, dL/drgb[3]}; the per-Gaussian factors that turn the one into the other are skipped
// and everything behind them — the conic -> covariance -> scale / quaternion chain, projection, SH — is the same code.
// ADAM = true (msgs_grads_t::adam_in_backward; raw mode 1, staged SH rows, no accumulate, not factored): the raw-parameter
// gradients are not stored — every Gaussian's parameters and moments take their Adam step here, with the gradient still in
// registers / LDS (zero for a Gaussian that was not rendered).
template <bool TEXTBOOK, bool ADAM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void preprocess_backward_kernel(ViewParams vp, msgs_gaussians_t g,
                                                                  const int32_t* __restrict__ radii,
                                                                  const char* __restrict__ geom,
                                                                  const grad_acc_t* __restrict__ grad_rec,
                                                                  msgs_grads_t grads, AdamInBackward ad) {
    // 32 rows per wave: the SH rows of a wave's 64 Gaussians pass through LDS in two runs of 32 (below).  25 KB per workgroup
    // instead of 50: the kernel is latency-bound and its time follows the occupancy (measured at C3 with 1 / 2 / 3 workgroups
    // per CU: 252 / 147 / 121 us)
    __shared__ float s_rows[4][K9_STAGE_ROWS * ROW_LDS];
    __shared__ uint8_t s_idx[4][64];
    const int P = g.P;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool in_range = i < P;
    const GeomLayout L(P);
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(geom + L.flags);
    const int K = vp.sh_coeffs;
    const int deg = vp.sh_degree;

    float dmean[3] = {0.f, 0.f, 0.f};
    float g2x = 0.f, g2y = 0.f, dopac = 0.f;
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dcolr[3] = {0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool rendered = in_range && radii[i] > 0;
    // SH rows in / dSH rows out through LDS with coalesced wave-cooperative transfers (K == 16 only;
    // other layouts take the direct per-thread path)
    const bool raw = g.raw_params != 0;
    const bool split_in = raw && g.shs == nullptr;
    const bool staged_sh = raw || (g.shs != nullptr && grads.dL_dshs != nullptr && K == 16);   // wave-uniform
    // factored SH gradient (view-parallel exchange, msgs_sh_grad_from_views): dL/dSH of one view is the outer product
    // basis(direction) x dL/drgb, so only the (clamp-masked) dL/drgb is delivered and the 192-byte rows are not written
    const bool factored_sh = !ADAM && raw && grads.dL_dfeatures_dc == nullptr;
    // accumulate (several views of one optimizer step into one gradient bucket): add to what the gradient tensors hold and
    // leave the rows of Gaussians this view did not render alone — no zero rows, no separate accumulation pass
    const bool accum = !ADAM && grads.accumulate != 0;
    const int wave_first = blockIdx.x * blockDim.x + wv * 64;
    const uint64_t live = __ballot(rendered);
    if (staged_sh) {                         // list of the rendered lanes, ascending: the rows to fetch
        if (rendered) s_idx[wv][__popcll(live & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    }
    uint32_t fl = 0;
    float p[3] = {0.f, 0.f, 0.f};

    if (rendered) {
        Cam cm;
        load_cam(vp, cm);
        // the nine sums, accumulated in grad_acc_t (double by default) and rounded to float ONCE here — the CPU oracle's
        // structure (double accumulators, one cast)
        float4 ga, gb, gc;
        if constexpr (TEXTBOOK) {
            const double* gr = reinterpret_cast<const double*>(grad_rec) + (size_t)i * 9;
            ga = make_float4((float)gr[0], (float)gr[1], (float)gr[2], (float)gr[3]);
            gb = make_float4((float)gr[4], (float)gr[5], (float)gr[6], (float)gr[7]);
            gc = make_float4((float)gr[8], 0.f, 0.f, 0.f);
        } else {
            const grad_acc_t* gr = grad_rec + (size_t)i * GRAD_REC_FLOATS;
            grad_acc_t t[GRAD_REC_FLOATS];
            typedef grad_acc_t acc2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int k = 0; k < GRAD_REC_FLOATS / 2; ++k) {
                const acc2 v2 = reinterpret_cast<const acc2*>(gr)[k];
                t[2 * k] = v2.x; t[2 * k + 1] = v2.y;
            }
            ga = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
            gb = make_float4((float)t[4], (float)t[5], (float)t[6], (float)t[7]);
            gc = make_float4((float)t[8], 0.f, 0.f, 0.f);
        }
        // blend_backward_kernel accumulates [sum q dx, sum q dy, sum q dx^2, sum q dx dy, sum q dy^2, sum q]
        // with q = alpha_raw dL/dalpha; the per-Gaussian constant factors are applied here (blend.hip).
        const float gA = TEXTBOOK ? ga.z : -0.5f * ga.z, gBh = TEXTBOOK ? ga.w : -0.5f * ga.w,
                    gC = TEXTBOOK ? gb.x : -0.5f * gb.x;
        const float o_in = act_opacity(g, i);
        if constexpr (TEXTBOOK)
            dopac = reinterpret_cast<const float*>(geom + L.weight)[i] * gb.y;      // SPEC M4: dL/do = w dL/do_eff
        else
            dopac = o_in > 0.f ? gb.y / o_in : 0.f;                     // (q / (o w)) * w, SPEC M4
        if (raw) dopac = dopac * (o_in * (1.0f - o_in));                // through the sigmoid
        dcolr[0] = gb.z; dcolr[1] = gb.w; dcolr[2] = gc.x;
        fl = flags[i];
        p[0] = g.means3D[3 * i]; p[1] = g.means3D[3 * i + 1]; p[2] = g.means3D[3 * i + 2];

        float cov3D[6];
        float R[3][3], S[3] = {0.f, 0.f, 0.f};
        float qr = 0.f, qx = 0.f, qy = 0.f, qz = 0.f, qnorm = 1.0f;
        float sact[3] = {1.f, 1.f, 1.f};
        if (g.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) cov3D[k] = g.cov3D_precomp[6 * i + k];
        } else {
            float s[3], q[4];
            act_scales(g, i, s);
            qnorm = act_rotation(g, i, q);
            qr = q[0]; qx = q[1]; qy = q[2]; qz = q[3];
            sact[0] = s[0]; sact[1] = s[1]; sact[2] = s[2];
            cov3d_from_scale_rot(s, vp.scale_modifier, q, cov3D);
            S[0] = vp.scale_modifier * s[0]; S[1] = vp.scale_modifier * s[1]; S[2] = vp.scale_modifier * s[2];
            R[0][0] = 1.f - 2.f * (qy * qy + qz * qz); R[0][1] = 2.f * (qx * qy - qr * qz); R[0][2] = 2.f * (qx * qz + qr * qy);
            R[1][0] = 2.f * (qx * qy + qr * qz); R[1][1] = 1.f - 2.f * (qx * qx + qz * qz); R[1][2] = 2.f * (qy * qz - qr * qx);
            R[2][0] = 2.f * (qx * qz - qr * qy); R[2][1] = 2.f * (qy * qz + qr * qx); R[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
        }

        // ---- 2-D covariance backward ----
        float t[3];
        view_point(cm.V, p, t);
        Cov2D c2;
        compute_cov2d(t, vp, cov3D, cm.V, c2);
        const float ca = c2.a, cb = c2.b, cc = c2.c;
        const float denom = ca * cc - cb * cb;
        if constexpr (TEXTBOOK) {
            g2x = ga.x; g2y = ga.y;
        } else {
            // dL/dmean2D = sum q (u, w) with u = A' dx + Bh' dy, w = C' dy + Bh' dx and (A', Bh', C') the log2-scaled
            // conic exactly as preprocess_kernel built it for the record
            constexpr float KLOG = -0.72134752044448170368f;            // -1/2 log2(e)
            constexpr float LN2 = 0.69314718055994530942f;
            const float det_inv = 1.f / denom;
            const float sA = KLOG * (cc * det_inv), sBh = KLOG * (-cb * det_inv), sC = KLOG * (ca * det_inv);
            g2x = (sA * ga.x + sBh * ga.y) * (LN2 * vp.W);              // 2 ln2 * 0.5 W  (NDC-ish units)
            g2y = (sC * ga.y + sBh * ga.x) * (LN2 * vp.H);
        }
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * gA + 2 * cb * cc * gBh + (denom - ca * cc) * gC);
            dL_dc = denom2inv * (-ca * ca * gC + 2 * ca * cb * gBh + (denom - ca * cc) * gA);
            dL_db = denom2inv * 2 * (cb * cc * gA - (denom + 2 * cb * cb) * gBh + ca * cb * gC);
            const float(*T)[3] = c2.T;
            dcov[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
            dcov[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
            dcov[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
            dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        }
        {
            const float Sg[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]},
                                    {cov3D[2], cov3D[4], cov3D[5]}};
            float ST0[3], ST1[3], dT0[3], dT1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ST0[k] = Sg[k][0] * c2.T[0][0] + Sg[k][1] * c2.T[0][1] + Sg[k][2] * c2.T[0][2];
                ST1[k] = Sg[k][0] * c2.T[1][0] + Sg[k][1] * c2.T[1][1] + Sg[k][2] * c2.T[1][2];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dT0[k] = 2 * ST0[k] * dL_da + ST1[k] * dL_db;
                dT1[k] = 2 * ST1[k] * dL_dc + ST0[k] * dL_db;
            }
            const float* V = cm.V;   // Wr[k][c] = V[4*c + k]
            const float dJ00 = V[0] * dT0[0] + V[4] * dT0[1] + V[8] * dT0[2];
            const float dJ02 = V[2] * dT0[0] + V[6] * dT0[1] + V[10] * dT0[2];
            const float dJ11 = V[1] * dT1[0] + V[5] * dT1[1] + V[9] * dT1[2];
            const float dJ12 = V[2] * dT1[0] + V[6] * dT1[1] + V[10] * dT1[2];
            const float tz = 1.f / c2.tz, tz2 = tz * tz, tz3 = tz2 * tz;
            const float dtx = c2.x_mul * -vp.fx * tz2 * dJ02;                       // Q2
            const float dty = c2.y_mul * -vp.fy * tz2 * dJ12;
            const float dtz = -vp.fx * tz2 * dJ00 - vp.fy * tz2 * dJ11 + (2 * vp.fx * c2.tx_c) * tz3 * dJ02 +
                              (2 * vp.fy * c2.ty_c) * tz3 * dJ12;
#pragma unroll
            for (int j = 0; j < 3; ++j) dmean[j] += V[4 * j + 0] * dtx + V[4 * j + 1] * dty + V[4 * j + 2] * dtz;
        }
        // ---- projection backward ----
        {
            float h[4];
            proj_point(cm.M, p, h);
            const float m_w = 1.0f / (h[3] + 0.0000001f);
            const float mul1 = h[0] * m_w * m_w, mul2 = h[1] * m_w * m_w;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dmean[j] += (cm.M[4 * j + 0] * m_w - cm.M[4 * j + 3] * mul1) * g2x +
                            (cm.M[4 * j + 1] * m_w - cm.M[4 * j + 3] * mul2) * g2y;
        }
        // ---- 3-D covariance backward ----
        if (!g.cov3D_precomp) {
            const float Gm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dM[3][3], dR[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc += Gm[a][k] * (R[k][b] * S[b]);
                    dM[a][b] = 2.f * acc;
                }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float ds = dM[0][j] * R[0][j] + dM[1][j] * R[1][j] + dM[2][j] * R[2][j];
                dscale[j] = vp.scale_modifier * ds;
#pragma unroll
                for (int a = 0; a < 3; ++a) dR[a][j] = dM[a][j] * S[j];
            }
            const float r = qr, x = qx, y = qy, z = qz;
            dq[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            dq[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
            dq[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
            dq[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
            if (raw) {
                // through exp: d/d(log s) = s * d/ds;  through normalize: (g - q (q.g)) / ||raw||
#pragma unroll
                for (int j = 0; j < 3; ++j) dscale[j] = dscale[j] * sact[j];
                const float dotq = r * dq[0] + x * dq[1] + y * dq[2] + z * dq[3];
                dq[0] = (dq[0] - r * dotq) / qnorm; dq[1] = (dq[1] - x * dotq) / qnorm;
                dq[2] = (dq[2] - y * dotq) / qnorm; dq[3] = (dq[3] - z * dotq) / qnorm;
            }
        }
    }
    // ---- colour backward (the last contribution to dL/dmean) ----
    // sh: this Gaussian's 48 (or 3K) coefficients, dsh: where their gradient goes (the same LDS row when staged; nullptr = the
    // factored path, rows not formed)
    auto colour_backward = [&](const float* sh, float* dsh) {
        {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (fl & (1u << c)) dcolr[c] = 0.f;                                // Q8
            const float dox = p[0] - vp.campos[0], doy = p[1] - vp.campos[1], doz = p[2] - vp.campos[2];
            const float len = sqrtf(dox * dox + doy * doy + doz * doz);
            const float x = dox / len, y = doy / len, z = doz / len;
            float ddir[3] = {0.f, 0.f, 0.f};
            // k-th coefficient: basis value and its direction derivatives
            auto coef = [&](int k, float basis, float bx, float by, float bz) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float shv = sh[k * 3 + c];          // read first: sh and dsh share the LDS row
                    if (dsh) dsh[k * 3 + c] = basis * dcolr[c];
                    s += shv * dcolr[c];
                }
                ddir[0] += bx * s; ddir[1] += by * s; ddir[2] += bz * s;
            };
            coef(0, SH_C0, 0.f, 0.f, 0.f);
            if (deg > 0) {
                coef(1, -SH_C1 * y, 0.f, -SH_C1, 0.f);
                coef(2, SH_C1 * z, 0.f, 0.f, SH_C1);
                coef(3, -SH_C1 * x, -SH_C1, 0.f, 0.f);
                if (deg > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    coef(4, SH_C2[0] * xy, SH_C2[0] * y, SH_C2[0] * x, 0.f);
                    coef(5, SH_C2[1] * yz, 0.f, SH_C2[1] * z, SH_C2[1] * y);
                    coef(6, SH_C2[2] * (2.f * zz - xx - yy), SH_C2[2] * -2.f * x, SH_C2[2] * -2.f * y, SH_C2[2] * 4.f * z);
                    coef(7, SH_C2[3] * xz, SH_C2[3] * z, 0.f, SH_C2[3] * x);
                    coef(8, SH_C2[4] * (xx - yy), SH_C2[4] * 2.f * x, SH_C2[4] * -2.f * y, 0.f);
                    if (deg > 2) {
                        coef(9, SH_C3[0] * y * (3.f * xx - yy), SH_C3[0] * 6.f * xy, SH_C3[0] * (3.f * xx - 3.f * yy), 0.f);
                        coef(10, SH_C3[1] * xy * z, SH_C3[1] * yz, SH_C3[1] * xz, SH_C3[1] * xy);
                        coef(11, SH_C3[2] * y * (4.f * zz - xx - yy), SH_C3[2] * -2.f * xy, SH_C3[2] * (4.f * zz - xx - 3.f * yy), SH_C3[2] * 8.f * yz);
                        coef(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), SH_C3[3] * -6.f * xz, SH_C3[3] * -6.f * yz, SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy));
                        coef(13, SH_C3[4] * x * (4.f * zz - xx - yy), SH_C3[4] * (4.f * zz - 3.f * xx - yy), SH_C3[4] * -2.f * xy, SH_C3[4] * 8.f * xz);
                        coef(14, SH_C3[5] * z * (xx - yy), SH_C3[5] * 2.f * xz, SH_C3[5] * -2.f * yz, SH_C3[5] * (xx - yy));
                        coef(15, SH_C3[6] * x * (xx - 3.f * yy), SH_C3[6] * (3.f * xx - 3.f * yy), SH_C3[6] * -6.f * xy, 0.f);
                    }
                }
            }
            const int ncoef = (deg + 1) * (deg + 1);
            if (dsh && !staged_sh)
                for (int k = ncoef; k < K; ++k) { dsh[k * 3] = 0.f; dsh[k * 3 + 1] = 0.f; dsh[k * 3 + 2] = 0.f; }
            const float dotv = x * ddir[0] + y * ddir[1] + z * ddir[2];
            dmean[0] += (ddir[0] - x * dotv) / len;
            dmean[1] += (ddir[1] - y * dotv) / len;
            dmean[2] += (ddir[2] - z * dotv) / len;
        }
    };
    const bool do_colour = !g.colors_precomp;                 // wave-uniform
    if (do_colour && staged_sh) {
        // the wave's SH rows in two runs of K9_STAGE_ROWS = 32 Gaussians: load (cooperative, coalesced) -> the lanes of that
        // run work on their LDS row in place -> store (cooperative, coalesced)
        const int nfloat = 3 * (deg + 1) * (deg + 1);
#pragma unroll 1
        for (int h = 0; h < 64 / K9_STAGE_ROWS; ++h) {
            const int row0 = h * K9_STAGE_ROWS;
            const int first = wave_first + row0;
            const int nrow = min(K9_STAGE_ROWS, P - first);
            if (nrow <= 0) break;                                          // wave-uniform
            const uint64_t live_h = (live >> row0) & ((1ull << K9_STAGE_ROWS) - 1ull);
            const bool mine = (lane / K9_STAGE_ROWS) == h;
            const int lrow = lane - row0;
            wave_lds_fence();                                              // s_idx visible / previous run stored
            if (split_in) {
                if (live_h != 0)
                    coop_load_split_rows_listed(s_rows[wv], g.features_dc, g.features_rest, wave_first,
                                                s_idx[wv] + __popcll(live & ((1ull << row0) - 1ull)), __popcll(live_h), lane, row0);
            } else {
                coop_load_rows(s_rows[wv], g.shs + (size_t)wave_first * ROW_F,
                               s_idx[wv] + __popcll(live & ((1ull << row0) - 1ull)), __popcll(live_h), sh_row_float4s(deg),
                               lane, row0);
            }
            wave_lds_fence();
            if (rendered && mine) {
                float* row = &s_rows[wv][lrow * ROW_LDS];
                colour_backward(row, factored_sh ? nullptr : row);
            }
            if (factored_sh) continue;
            wave_lds_fence();
            if constexpr (ADAM)
                coop_adam_split_rows(s_rows[wv], const_cast<float*>(g.features_dc), const_cast<float*>(g.features_rest), ad, i,
                                     in_range && mine, first, nrow, live_h, nfloat, lane, lrow);
            else if (raw)
                coop_store_split_rows(s_rows[wv], grads.dL_dfeatures_dc, grads.dL_dfeatures_rest, i, in_range && mine, first,
                                      nrow, live_h, nfloat, lane, lrow, accum);
            else
                coop_store_rows(s_rows[wv], grads.dL_dshs + (size_t)first * ROW_F, nrow, live_h, nfloat, lane, accum);
        }
    } else if (do_colour) {
        // (direct per-thread rows, K != 16: accumulate mode is not offered on this path — msgs_backward refuses it)
        float* dsh = grads.dL_dshs && in_range ? grads.dL_dshs + (size_t)3 * K * i : nullptr;
        if (rendered) colour_backward(g.shs + (size_t)3 * K * i, dsh);
        else if (dsh) for (int k = 0; k < 3 * K; ++k) dsh[k] = 0.f;
    }
    if (!in_range) return;

    // the screen-space gradient is per view (viewspace_points.grad of THIS render, scene/gaussian_model.py:698-701): stored
    if (grads.dL_dmeans2D) { grads.dL_dmeans2D[3 * i] = g2x; grads.dL_dmeans2D[3 * i + 1] = g2y; grads.dL_dmeans2D[3 * i + 2] = 0.f; }
    if constexpr (ADAM) {
        // the small tensors: this thread's own rows (gradients are zero where the Gaussian was not rendered)
        auto step3 = [&](const float* param, int t, const float* gr) {
            float* P = const_cast<float*>(param) + 3 * (size_t)i;
            float* M = ad.m[t] + 3 * (size_t)i;
            float* V = ad.v[t] + 3 * (size_t)i;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float pp = P[c], mm = M[c], vv = V[c];
                adam_update(pp, gr[c], mm, vv, ad.nss[t], ad.a);
                P[c] = pp; M[c] = mm; V[c] = vv;
            }
        };
        step3(g.means3D, 0, dmean);
        {
            float* P = const_cast<float*>(g.opacities) + i;
            float pp = *P, mm = ad.m[3][i], vv = ad.v[3][i];
            adam_update(pp, dopac, mm, vv, ad.nss[3], ad.a);
            *P = pp; ad.m[3][i] = mm; ad.v[3][i] = vv;
        }
        step3(g.scales, 4, dscale);
        {
            float4* P = reinterpret_cast<float4*>(const_cast<float*>(g.rotations)) + i;
            float4* M = reinterpret_cast<float4*>(ad.m[5]) + i;
            float4* V = reinterpret_cast<float4*>(ad.v[5]) + i;
            float4 pp = *P, mm = *M, vv = *V;
            adam_update(pp.x, dq[0], mm.x, vv.x, ad.nss[5], ad.a);
            adam_update(pp.y, dq[1], mm.y, vv.y, ad.nss[5], ad.a);
            adam_update(pp.z, dq[2], mm.z, vv.z, ad.nss[5], ad.a);
            adam_update(pp.w, dq[3], mm.w, vv.w, ad.nss[5], ad.a);
            *P = pp; *M = mm; *V = vv;
        }
        return;
    }
    if (accum) {
        if (!rendered) return;
        if (grads.dL_dmeans3D) {
            float* d = grads.dL_dmeans3D + 3 * (size_t)i;
            d[0] = d[0] + dmean[0]; d[1] = d[1] + dmean[1]; d[2] = d[2] + dmean[2];
        }
        if (grads.dL_dopacities) grads.dL_dopacities[i] = grads.dL_dopacities[i] + dopac;
        if (grads.dL_dcolors && !factored_sh) {
            float* d = grads.dL_dcolors + 3 * (size_t)i;
            d[0] = d[0] + dcolr[0]; d[1] = d[1] + dcolr[1]; d[2] = d[2] + dcolr[2];
        }
        if (grads.dL_dscales) {
            float* d = grads.dL_dscales + 3 * (size_t)i;
            d[0] = d[0] + dscale[0]; d[1] = d[1] + dscale[1]; d[2] = d[2] + dscale[2];
        }
        if (grads.dL_drotations) {
            float4* d = reinterpret_cast<float4*>(grads.dL_drotations) + i;
            const float4 o = *d;
            *d = make_float4(o.x + dq[0], o.y + dq[1], o.z + dq[2], o.w + dq[3]);
        }
        if (grads.dL_dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; ++k) grads.dL_dcov3D[6 * (size_t)i + k] = grads.dL_dcov3D[6 * (size_t)i + k] + dcov[k];
        }
        return;
    }
    if (grads.dL_dmeans3D) { grads.dL_dmeans3D[3 * i] = dmean[0]; grads.dL_dmeans3D[3 * i + 1] = dmean[1]; grads.dL_dmeans3D[3 * i + 2] = dmean[2]; }
    if (grads.dL_dopacities) grads.dL_dopacities[i] = dopac;
    if (grads.dL_dcolors && !factored_sh) { grads.dL_dcolors[3 * i] = dcolr[0]; grads.dL_dcolors[3 * i + 1] = dcolr[1]; grads.dL_dcolors[3 * i + 2] = dcolr[2]; }
    if (grads.dL_dscales) { grads.dL_dscales[3 * i] = dscale[0]; grads.dL_dscales[3 * i + 1] = dscale[1]; grads.dL_dscales[3 * i + 2] = dscale[2]; }
    if (grads.dL_drotations) reinterpret_cast<float4*>(grads.dL_drotations)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
    if (grads.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) grads.dL_dcov3D[6 * i + k] = dcov[k];
    }
}

// The factors themselves: dL/drgb of one view (the three colour sums of the gradient record, zero where the colour was
// clamped, Q8, and for Gaussians that were not rendered) — a kernel of its own in front of K9, so that the exchange of
// the factors can overlap with K9 (msgs_grads_t::factors_ready).
__global__ __launch_bounds__(256) void sh_factor_kernel(int P, const int32_t* __restrict__ radii,
                                                        const char* __restrict__ geom,
                                                        const grad_acc_t* __restrict__ grad_rec, int rec_stride,
                                                        float* __restrict__ dL_dcolors) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (radii[i] > 0) {
        const GeomLayout L(P);
        const uint32_t fl = reinterpret_cast<const uint32_t*>(geom + L.flags)[i];
        const grad_acc_t* gr = grad_rec + (size_t)i * rec_stride;       // (components 6..8 in both record layouts)
        c0 = (fl & 1u) ? 0.f : (float)gr[6];
        c1 = (fl & 2u) ? 0.f : (float)gr[7];
        c2 = (fl & 4u) ? 0.f : (float)gr[8];
    }
    dL_dcolors[3 * (size_t)i] = c0; dL_dcolors[3 * (size_t)i + 1] = c1; dL_dcolors[3 * (size_t)i + 2] = c2;
}

// ---------------------------------------------------------------------------------------------
// Factored SH gradient of N views -> dL/dfeatures_dc, dL/dfeatures_rest  (view-parallel exchange)
//   dL/dSH[i][k][c] = scale * sum_v basis_k(normalize(p_i - campos_v)) * drgb[v][i][c]
// — exactly the products preprocess_backward_kernel forms for one view (same expressions, contraction off), added in
// view order, so every rank reconstructs bit-identical sums from the gathered [N,P,3] factors instead of exchanging
// 48 floats per Gaussian.  One thread per Gaussian, rows leave through LDS like K9's.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sh_grad_from_views_kernel(int P, int n_views, int deg,
                                                                 const float* __restrict__ means3D,
                                                                 const float* __restrict__ campos,   // view v at campos + v*cs
                                                                 int64_t cs,
                                                                 const float* __restrict__ drgb,     // view v: [P,3] at drgb + v*ds
                                                                 int64_t ds, float scale, float* __restrict__ d_dc,
                                                                 float* __restrict__ d_rest) {
    __shared__ float s_rows[4][64 * ROW_LDS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool in_range = i < P;
    const int wave_first = blockIdx.x * blockDim.x + wv * 64;
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.f;
    if (in_range) {
        const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
        for (int v = 0; v < n_views; ++v) {
            const float* d = drgb + (size_t)v * ds + (size_t)i * 3;
            const float* cp = campos + (size_t)v * cs;
            const float d0 = d[0], d1 = d[1], d2 = d[2];
            if (d0 == 0.f && d1 == 0.f && d2 == 0.f) continue;          // not rendered in this view (or fully clamped)
            const float dox = p[0] - cp[0], doy = p[1] - cp[1], doz = p[2] - cp[2];
            const float len = sqrtf(dox * dox + doy * doy + doz * doz);
            float b[16];
            sh_basis(deg, dox / len, doy / len, doz / len, b);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[3 * k] = acc[3 * k] + b[k] * d0;
                acc[3 * k + 1] = acc[3 * k + 1] + b[k] * d1;
                acc[3 * k + 2] = acc[3 * k + 2] + b[k] * d2;
            }
        }
    }
    float* row = &s_rows[wv][lane * ROW_LDS];
#pragma unroll
    for (int k = 0; k < 48; ++k) row[k] = acc[k] * scale;
    wave_lds_fence();
    const int nrow = min(64, P - wave_first);
    if (nrow > 0)
        coop_store_split_rows(s_rows[wv], d_dc, d_rest, i, in_range, wave_first, nrow, ~0ull, 48, lane, lane);
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ V,
                                    uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float Vl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Vl[k] = V[k];
    float t[3];
    view_point(Vl, p, t);
    present[i] = t[2] > 0.2f ? 1 : 0;
}

}  // namespace

hipError_t launch_preprocess(const ViewParams& vp, const msgs_gaussians_t& g, int32_t* radii, float* pixel_sizes,
                             char* geom, hipStream_t s, ZeroJob zj, uint32_t* heavy_list, uint32_t* heavy_count, uint32_t* heavy_blk,
                             bool write_litrec) {
    if (g.P == 0) return hipSuccess;
    if (g.raw_params != 0 && g.shs == nullptr)
        hipLaunchKernelGGL(preprocess_kernel<true>, dim3((g.P + 255) / 256), dim3(256), 0, s, vp, g, radii, pixel_sizes, geom, zj,
                           heavy_list, heavy_count, heavy_blk, write_litrec ? 1 : 0);
    else
        hipLaunchKernelGGL(preprocess_kernel<false>, dim3((g.P + 255) / 256), dim3(256), 0, s, vp, g, radii, pixel_sizes, geom, zj,
                           heavy_list, heavy_count, heavy_blk, write_litrec ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_preprocess_backward(const ViewParams& vp, const msgs_gaussians_t& g, const int32_t* radii,
                                      const char* geom, const grad_acc_t* grad_rec, const msgs_grads_t& grads,
                                      hipStream_t s, bool textbook) {
    if (g.P == 0) return hipSuccess;
    // textbook: grad_rec holds [P, 9] doubles — the TEXTBOOK 2-D gradients (msgs_backward_per_gaussian; the verification mode)
    static_assert(sizeof(grad_acc_t) == 8, "the textbook sums are doubles");
    const msgs_adam_in_backward_t* aib = grads.adam_in_backward;
    if (g.raw_params != 0 && grads.dL_dfeatures_dc == nullptr && !aib) {          // factored SH gradient: the factors first
        hipLaunchKernelGGL(sh_factor_kernel, dim3((g.P + 255) / 256), dim3(256), 0, s, g.P, radii, geom, grad_rec,
                           textbook ? 9 : GRAD_REC_FLOATS, grads.dL_dcolors);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (grads.factors_ready) {
            e = hipEventRecord((hipEvent_t)grads.factors_ready, s);
            if (e != hipSuccess) return e;
        }
    }
    // several views into one gradient bucket from different streams: this kernel reads and writes the bucket, so it
    // waits for the previous view's kernel (its `accumulated` event) and signals its own end
    if (grads.wait_before_accumulate) {
        hipError_t e = hipStreamWaitEvent(s, (hipEvent_t)grads.wait_before_accumulate, 0);
        if (e != hipSuccess) return e;
    }
    AdamInBackward ad{};
    if (aib) {
        for (int t = 0; t < 6; ++t) {
            ad.m[t] = aib->t[t].exp_avg;
            ad.v[t] = aib->t[t].exp_avg_sq;
            ad.nss[t] = adam_neg_step_size(aib->t[t].lr, aib->step, aib->beta1);
        }
        ad.a = adam_scalars(aib->step, aib->beta1, aib->beta2, aib->eps);
    }
    const dim3 grid((g.P + 255) / 256), block(256);
    if (textbook && aib) hipLaunchKernelGGL((preprocess_backward_kernel<true, true>), grid, block, 0, s, vp, g, radii, geom, grad_rec, grads, ad);
    else if (textbook) hipLaunchKernelGGL((preprocess_backward_kernel<true, false>), grid, block, 0, s, vp, g, radii, geom, grad_rec, grads, ad);
    else if (aib) hipLaunchKernelGGL((preprocess_backward_kernel<false, true>), grid, block, 0, s, vp, g, radii, geom, grad_rec, grads, ad);
    else hipLaunchKernelGGL((preprocess_backward_kernel<false, false>), grid, block, 0, s, vp, g, radii, geom, grad_rec, grads, ad);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && grads.accumulated) e = hipEventRecord((hipEvent_t)grads.accumulated, s);
    return e;
}

hipError_t launch_sh_grad_from_views(int P, int n_views, int deg, const float* means3D, const float* campos,
                                     int64_t campos_stride, const float* drgb, int64_t drgb_stride, float scale,
                                     float* d_dc, float* d_rest, hipStream_t s) {
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(sh_grad_from_views_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, n_views, deg, means3D,
                       campos, campos_stride, drgb, drgb_stride, scale, d_dc, d_rest);
    return hipGetLastError();
}

hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                               uint8_t* present, hipStream_t s) {
    (void)projmatrix;
    if (P == 0) return hipSuccess;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
    return hipGetLastError();
}

}  // namespace msgs
