/*
 * msgs_oracle.cpp — float32 CPU restatement of the multi-scale 3D-Gaussian rasterizer
 * (preprocess -> duplicate with (tile, depth) keys -> stable sort -> tile ranges -> per-tile
 * front-to-back blend; back-to-front blend backward -> 2-D covariance backward -> preprocess
 * backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, load or call this file; the product (ms-gs_amd/) never does.
 *
 * PARITY UNPINNED.  The reference's implementation of this path is the un-vendored submodule
 * submodules/diff-gaussian-rasterization -> https://github.com/JokerYan/MS-GS-rasterizer.git
 * (/root/reference/.gitmodules:4-6): the directory is empty and the pinned SHA is unrecoverable,
 * so the reference's arithmetic can be neither compiled nor imported here, and the reference holds
 * no tests / golden vectors for it.  This file therefore restates
 *   - the published tile-based EWA splatting algorithm of graphdeco-inria/diff-gaussian-rasterization
 *     (Kerbl et al., "3D Gaussian Splatting", 2023), which the MS-GS rasterizer forks
 *     (SURVEY.md App. A.1-A.3), in the same structure: per-Gaussian preprocess, 64-bit
 *     (tile << 32 | depth bits) keys, stable sort, per-tile ranges, per-pixel blend;
 *   - the MS-GS additions as they are visible at the reference's call sites:
 *       settings filter_small / filter_large / fade_size   gaussian_renderer/__init__.py:50-52
 *       inputs max/min_pixel_sizes, occ_multiplier, dc_delta, base_mask           :99-107
 *       5-tuple return (color, acc_pixel_size, depth, radii, pixel_sizes)         :94
 *       consumers: scene/gaussian_model.py:663-686 (update_pixel_sizes), :713-727 (acc_pixel_size,
 *       depth are [H,W]), train.py:244-250, 288-299
 *     with the semantics frozen in DESIGN.md §SPEC (M1-M6);
 *   - conventions pinned by importable reference helpers: SH polynomial utils/sh_utils.py:57-112,
 *     quaternion->R and covariance packing utils/general_utils.py:64-110, matrices
 *     utils/graphics_utils.py:38-71 + scene/cameras.py:54-57 (fixtures under tests/golden/).
 * It deliberately shares no code with ms-gs_amd/csrc: it keeps the reference's structure
 * (rect-based duplication, one 64-bit key sort) while the HIP path uses exact ellipse culling and a
 * two-level sort; agreement between the two is what the parity tests establish.
 *
 * Build: g++ -O2 -fopenmp -ffp-contract=off -shared -fPIC (oracle/Makefile).  -ffp-contract=off
 * keeps every float32 operation individually rounded so the per-Gaussian stage is bit-comparable
 * with the HIP preprocess kernel (which is compiled with contraction off for the same reason).
 */
#include "msgs_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <parallel/algorithm>
#include <omp.h>

// Precision of the arithmetic.  Default: float32 — THE checker of the parity tests (liboracle.so).  -DMSGS_ORACLE_F64 builds
// the SAME source with every computed quantity in double (liboracle64.so): the float64 "truth" of the three-way tests
// (HIP vs float32 oracle vs truth) at the full BASELINE sizes, where the autograd oracle (torch_oracle.py) takes minutes to
// hours; tests/test_oracle_cpu.py checks it against that autograd oracle.  In both builds the op's INPUTS are the float32
// arrays the HIP library receives (in_t), the tile lists are ordered by the FLOAT32 view depth (Q10), and the outputs /
// gradients are written in `real` (the float64 build is called with double buffers through the same C signatures).
#ifdef MSGS_ORACLE_F64
typedef double real;
#define RL(x) x
#else
typedef float real;
#define RL(x) x##f
#endif
typedef float in_t;

namespace {

constexpr int TILE = MSGS_TILE;
constexpr real SH_C0 = RL(0.28209479177387814);
constexpr real SH_C1 = RL(0.4886025119029199);
constexpr real SH_C2[5] = {RL(1.0925484305920792), RL(-1.0925484305920792), RL(0.31539156525252005),
                           RL(-1.0925484305920792), RL(0.5462742152960396)};
constexpr real SH_C3[7] = {RL(-0.5900435899266435), RL(2.890611442640554), RL(-0.4570457994644658),
                           RL(0.3731763325901154), RL(-0.4570457994644658), RL(1.445305721320277),
                           RL(-0.5900435899266435)};

struct Geom {
    float depth32;           // float32 view depth: the sort key (Q10) in both builds
    real depth, px, py;
    real con[3], opacity;   // conic (A,B,C), effective opacity (after the fade weight)
    real con_cond;          // (|a c| + b^2) / |a c - b^2| of the 2-D covariance: how much the conic's 1 / det amplifies float32
                            // rounding of (a, b, c) — 1 for a round footprint, ~ aspect^2 / 2 for a rotated needle
    real rgb[3];
    real cov3D[6];
    real pixel_size, weight;
    int32_t radius;
    int32_t rect[4];         // minx, miny, maxx, maxy (tiles)
    uint8_t clamped[3];
    uint8_t visible;
    uint8_t ghost;           // dropped by a multi-scale filter whose decision was within rounding of flipping: kept in the
                             // tile lists WITHOUT blending, only to flag the pixels it would have reached
    uint8_t filter_edge;     // pixel size within FILTER_EDGE (relative) of an active min / max threshold (fade_size == 0)
};

}  // namespace

// Relative half-width of the window around alpha = 1/255 inside which ANOTHER float32 implementation of the same algorithm may
// take the other branch.  d(alpha)/alpha = d(power), and a float32 evaluation of
//     power = -0.5 (A dx^2 + C dy^2) - B dx dy
// carries (i) the rounding of three products that cancel — 2^-24 of their magnitude sum M each, M >> |power| for a rotated
// elongated footprint — and (ii) the rounding of the conic itself, whose 1 / det amplifies float32 rounding of the 2-D covariance
// by con_cond (correlated over A, B, C, so it scales with |power|, not with M).  Measured (profiles/r5_parity.md): a needle of
// aspect 23 with M = 1179 at power -5.4 — the contraction-off build, the FMA build and the float64 build put alpha x 255 - 1 at
// -2.5e-5, -8.6e-5 and -3.4e-5, the HIP kernels at >= 0.  Round, well-conditioned footprints keep the 2e-5 of rounds 1-4.
static inline real alpha_window(const Geom& ge, real dx, real dy, real power) {
    const real M = RL(0.5) * (std::fabs(ge.con[0]) * dx * dx + std::fabs(ge.con[2]) * dy * dy) + std::fabs(ge.con[1] * dx * dy);
    return std::min(RL(0.1), RL(2e-5) + RL(1.8e-7) * (M + ge.con_cond * std::fabs(power)));      // 1.8e-7 = 3 x 2^-24
}

// Half-width of the window around power = 0 inside which ANOTHER float32 implementation may take the other branch of the
// reference's `if (power > 0) continue` guard.  For a positive-definite conic the exponent is never positive; the guard only
// catches roundings, i.e. pixels a hair from a Gaussian's centre, where the Gaussian has its LARGEST alpha: blended by one
// implementation and skipped by another it changes the pixel visibly.  Besides the rounding of the three products (2^-24 of
// their magnitude sum M each) an implementation may fold log2(opacity) into the exponent — this build's HIP kernels do: one
// FMA chain gives log2(alpha), and the sign test becomes `log2(alpha) > log2(opacity)` — which resolves the sign only to an ulp
// or two of log2(opacity).  Found by the 60 000-configuration sweep (profiles/r5_parity.md 2.3): a giant (conic 3e-4) whose
// centre sits 0.014 px from a pixel centre, power = -7.5e-9, blended by the three oracle builds, skipped by the HIP kernels.
static inline real power_sign_window(const Geom& ge, real dx, real dy) {
    const real M = RL(0.5) * (std::fabs(ge.con[0]) * dx * dx + std::fabs(ge.con[2]) * dy * dy) + std::fabs(ge.con[1] * dx * dy);
    const real l2o = std::fabs(std::log2(std::max(ge.opacity, RL(1e-30))));
    return RL(1.8e-7) * M + RL(3.3e-7) * std::max(RL(1.0), l2o);       // 3 x 2^-24 M  +  2^-21 ln 2 x max(1, |log2 opacity|)
}

// exp() of the blend kernels.  The float32 checker's default is glibc's expf — what rounds 1-5 were checked against.  With
// msgs_oracle_set_exp_double(1) the exponential is evaluated in double and rounded to float ONCE: the product's literal
// verification mode (msgs_set_deterministic, ms-gs_amd/csrc/literal.hip) does the same, so that the two sides take the SAME
// float for every alpha (two float32 exponentials — glibc's expf, the GPU's v_exp_f32 — differ by an ulp on about a third of
// their arguments; two double exponentials rounded once differ practically never).  No-op in the float64 build.
static int g_exp_double = 0;
extern "C" int msgs_oracle_set_exp_double(int on) { const int prev = g_exp_double; g_exp_double = on ? 1 : 0; return prev; }
static inline real oracle_exp(real p) {
#if !defined(MSGS_ORACLE_F64)
    if (g_exp_double) return (real)std::exp((double)p);
#endif
    return std::exp(p);
}

#if !defined(MSGS_ORACLE_F64)
// introspection for tests/test_oracle_cpu.py: the two decision windows for a conic (A, B, C), its conditioning, an opacity and a
// pixel offset — so that a test can bound them on well-conditioned footprints and they cannot grow silently
extern "C" void msgs_oracle_windows(float A, float B, float Cc, float con_cond, float opacity, float dx, float dy, float* out2) {
    Geom ge{};
    ge.con[0] = A; ge.con[1] = B; ge.con[2] = Cc; ge.con_cond = con_cond; ge.opacity = opacity;
    const real power = -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
    out2[0] = alpha_window(ge, dx, dy, power);
    out2[1] = power_sign_window(ge, dx, dy);
}
#endif

struct msgs_oracle_state {
    int P = 0, W = 0, H = 0, gx = 0, gy = 0;
    std::vector<Geom> geom;
    std::vector<uint32_t> list;           // sorted Gaussian ids
    std::vector<uint32_t> range_lo, range_hi;
    std::vector<real> final_T;
    std::vector<uint32_t> n_contrib;
    // flat copies for introspection
    std::vector<real> depths, conic_opacity, rgb, means2D, cov3D;
    std::vector<int32_t> rects;
    std::vector<uint8_t> borderline_gauss;  // Gaussian had an alpha within rounding distance of 1/255 on some pixel, or
                                            // shares a pixel with a Gaussian whose filter decision could flip
    std::vector<uint8_t> filter_edge;       // Gaussian's own filter decision is within rounding of flipping
    std::vector<uint8_t> shared_gauss;      // Gaussian reaches (alpha >= 1/255) a pixel at which SOME discrete decision could
                                            // flip — an alpha at 1/255, a transmittance at 1e-4, a filter-edge Gaussian: its
                                            // term at that pixel moves with that decision (T behind the entry scales by
                                            // 1 - 1/255, the colour composited behind the entries in front changes)
    int64_t ghost_instances = 0;
    int64_t traversed = 0;
    int64_t valid_pairs = 0;      // (pixel, Gaussian) evaluations with alpha >= 1/255 before the pixel terminated
    int64_t evaluated_pairs = 0;  // all evaluations of the reference algorithm (sum over pixels of n_contrib-ish)
};

namespace {

inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

/* SH -> RGB (utils/sh_utils.py:74-100; +0.5 and clamp as gaussian_renderer/__init__.py:86-87) */
inline void sh_to_rgb(int deg, int K, const in_t* sh, const in_t* p, const in_t* campos,
                      real* rgb, uint8_t* clamped) {
    real dx = (real)p[0] - (real)campos[0], dy = (real)p[1] - (real)campos[1], dz = (real)p[2] - (real)campos[2];
    real len = std::sqrt(dx * dx + dy * dy + dz * dz);
    real x = dx / len, y = dy / len, z = dz / len;
    (void)K;
    for (int c = 0; c < 3; ++c) {
        auto S = [&](int k) { return (real)sh[k * 3 + c]; };
        real r = SH_C0 * S(0);
        if (deg > 0) {
            r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) +
                    SH_C2[2] * (2.0f * zz - xx - yy) * S(6) + SH_C2[3] * xz * S(7) +
                    SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) +
                        SH_C3[5] * z * (xx - yy) * S(14) + SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
        r += 0.5f;
        clamped[c] = r < 0.0f;
        rgb[c] = r < 0.0f ? 0.0f : r;
    }
}

/* Sigma = R diag(mod s)^2 R^T, R(q) as utils/general_utils.py:85-98 (no normalisation) */
inline void cov3d_from_scale_rot(const in_t* s, real mod, const in_t* q, real* cov) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    real R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                     {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                     {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    real S[3] = {mod * s[0], mod * s[1], mod * s[2]};
    real M[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * S[j];
    auto dot = [&](int i, int j) { return M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2]; };
    cov[0] = dot(0, 0); cov[1] = dot(0, 1); cov[2] = dot(0, 2);
    cov[3] = dot(1, 1); cov[4] = dot(1, 2); cov[5] = dot(2, 2);
}

struct Cov2DCtx {
    real T[2][3];       // J * Wr
    real a, b, c;       // with the +0.3
    real tx_c, ty_c, tz;
    real x_mul, y_mul;
    real fx, fy;
};

inline void compute_cov2d(const real* t, real fx, real fy, real tanx, real tany,
                          const real* cov3D, const in_t* V, Cov2DCtx& o) {
    real limx = RL(1.3) * tanx, limy = RL(1.3) * tany;
    real txtz = t[0] / t[2], tytz = t[1] / t[2];
    o.x_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    o.y_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    o.tx_c = std::min(limx, std::max(-limx, txtz)) * t[2];
    o.ty_c = std::min(limy, std::max(-limy, tytz)) * t[2];
    o.tz = t[2];
    o.fx = fx; o.fy = fy;
    real J[2][3] = {{fx / t[2], 0.f, -(fx * o.tx_c) / (t[2] * t[2])},
                     {0.f, fy / t[2], -(fy * o.ty_c) / (t[2] * t[2])}};
    // Wr[k][c] = W2C rotation (row k, col c) = V[4*c + k]
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            o.T[r][c] = J[r][0] * (real)V[4 * c + 0] + J[r][1] * (real)V[4 * c + 1] + J[r][2] * (real)V[4 * c + 2];
    real S[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]},
                     {cov3D[2], cov3D[4], cov3D[5]}};
    real ST[2][3];   // ST[r][i] = sum_j S[i][j] T[r][j]
    for (int r = 0; r < 2; ++r)
        for (int i = 0; i < 3; ++i) ST[r][i] = S[i][0] * o.T[r][0] + S[i][1] * o.T[r][1] + S[i][2] * o.T[r][2];
    o.a = (o.T[0][0] * ST[0][0] + o.T[0][1] * ST[0][1] + o.T[0][2] * ST[0][2]) + RL(0.3);
    o.b = o.T[0][0] * ST[1][0] + o.T[0][1] * ST[1][1] + o.T[0][2] * ST[1][2];
    o.c = (o.T[1][0] * ST[1][0] + o.T[1][1] * ST[1][1] + o.T[1][2] * ST[1][2]) + RL(0.3);
}

inline void view_point(const in_t* V, const in_t* p, real* t) {
    const real x = p[0], y = p[1], z = p[2];
    t[0] = (((real)V[0] * x + (real)V[4] * y) + (real)V[8] * z) + (real)V[12];
    t[1] = (((real)V[1] * x + (real)V[5] * y) + (real)V[9] * z) + (real)V[13];
    t[2] = (((real)V[2] * x + (real)V[6] * y) + (real)V[10] * z) + (real)V[14];
}

inline void proj_point(const in_t* M, const in_t* p, real* h) {
    const real x = p[0], y = p[1], z = p[2];
    for (int k = 0; k < 4; ++k) h[k] = (((real)M[k] * x + (real)M[4 + k] * y) + (real)M[8 + k] * z) + (real)M[12 + k];
}

/* MS-GS pixel size (DESIGN.md SPEC M1): extent, through the centre and along the image axes, of
 * the alpha >= 1/255 level set of the low-passed 2-D Gaussian; the smaller of the two. */
inline real pixel_size_of(real opacity, real conA, real conC) {
    real v = 255.0f * opacity;
    if (!(v > 1.0f) || !(conA > 0.f) || !(conC > 0.f)) return 0.f;
    real ell = 2.0f * std::log(v);
    real sx = 2.0f * std::sqrt(ell / conA);
    real sy = 2.0f * std::sqrt(ell / conC);
    return std::min(sx, sy);
}

/* The hard filters (fade_size == 0) compare the pixel size with a threshold: two float32 implementations whose sizes differ
 * by the last bits of logf / sqrtf take different decisions when size / threshold is within FILTER_EDGE of 1 (the HIP kernel
 * and this file agree to 1e-4 relative on pixel_sizes).  Such a Gaussian is flagged, and — rendered or not — stays in the
 * tile lists so that every pixel it reaches (and every Gaussian blended there) is flagged too. */
constexpr double FILTER_EDGE = 1e-4;
inline bool filter_on_edge(const msgs_view_t* v, real size, real minps, real maxps, bool base) {
    if (v->fade_size > 0.f) return false;                 // a ramp: continuous in the size
    bool edge = false;
    if (v->filter_small && !base && minps > 0.f) edge |= std::fabs((double)size / (double)minps - 1.0) < FILTER_EDGE;
    if (v->filter_large && maxps > 0.f) edge |= std::fabs((double)size / (double)maxps - 1.0) < FILTER_EDGE;
    return edge;
}

/* MS-GS filter weight (DESIGN.md SPEC M2-M4) */
inline real filter_weight(const msgs_view_t* v, real size, real minps, real maxps, bool base) {
    real w = 1.0f;
    if (v->filter_small && !base && minps > 0.f && size < minps) {
        if (v->fade_size > 0.f) {
            real rel = minps / std::max<real>(size, RL(1e-30));
            w *= std::min<real>(1.f, std::max<real>(0.f, 1.f - (rel - 1.f) / (real)v->fade_size));
        } else w = 0.f;
    }
    if (v->filter_large && maxps > 0.f && size > maxps) {
        if (v->fade_size > 0.f) {
            real rel = size / maxps;
            w *= std::min<real>(1.f, std::max<real>(0.f, 1.f - (rel - 1.f) / (real)v->fade_size));
        } else w = 0.f;
    }
    return w;
}

}  // namespace

extern "C" int msgs_oracle_forward(const msgs_view_t* view, const msgs_gaussians_t* g,
                                   float* out_color_, float* out_acc_ps_, float* out_depth_,
                                   int32_t* radii, float* pixel_sizes_, uint8_t* borderline,
                                   msgs_oracle_state_t** state_out, int num_threads) {
    // (float64 build: the caller passes DOUBLE buffers through the float* parameters of the shared header)
    real* out_color = reinterpret_cast<real*>(out_color_);
    real* out_acc_ps = reinterpret_cast<real*>(out_acc_ps_);
    real* out_depth = reinterpret_cast<real*>(out_depth_);
    real* pixel_sizes = reinterpret_cast<real*>(pixel_sizes_);
    if (!view || !g) return MSGS_ERR_INVALID_ARG;
    if ((g->shs != nullptr) == (g->colors_precomp != nullptr)) return MSGS_ERR_INVALID_ARG;
    bool has_sr = g->scales != nullptr && g->rotations != nullptr;
    if (has_sr == (g->cov3D_precomp != nullptr)) return MSGS_ERR_INVALID_ARG;
    if (g->shs && (view->sh_degree < 0 || view->sh_degree > 3 ||
                   (view->sh_degree + 1) * (view->sh_degree + 1) > view->sh_coeffs))
        return MSGS_ERR_SH_DEGREE;
    if (num_threads > 0) omp_set_num_threads(num_threads);
    const int P = g->P, W = view->image_width, H = view->image_height;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    auto* st = new msgs_oracle_state();
    st->P = P; st->W = W; st->H = H; st->gx = gx; st->gy = gy;
    st->geom.assign(P, Geom{});
    const real fx = W / (2.0f * (real)view->tanfovx), fy = H / (2.0f * (real)view->tanfovy);
    const in_t* V = view->viewmatrix;
    const in_t* PM = view->projmatrix;

    // ---- K1 preprocess (App. A.1) ----
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        Geom& ge = st->geom[i];
        ge.visible = 0; ge.ghost = 0; ge.filter_edge = 0; ge.radius = 0; ge.pixel_size = 0.f;
        radii[i] = 0; pixel_sizes[i] = 0.f;
        const in_t* p = g->means3D + 3 * i;
        real t[3];
        view_point(V, p, t);
        ge.depth = t[2];
        // Q10: the tile lists are ordered by the FLOAT32 view depth in both builds
        ge.depth32 = ((V[2] * p[0] + V[6] * p[1]) + V[10] * p[2]) + V[14];
        if (t[2] <= RL(0.2)) continue;                                   // Q1
        real h[4];
        proj_point(PM, p, h);
        real pw = 1.0f / (h[3] + RL(0.0000001));                         // Q9
        real ndc_x = h[0] * pw, ndc_y = h[1] * pw;
        if (g->cov3D_precomp) { for (int c = 0; c < 6; ++c) ge.cov3D[c] = g->cov3D_precomp[6 * i + c]; }
        else cov3d_from_scale_rot(g->scales + 3 * i, view->scale_modifier, g->rotations + 4 * i, ge.cov3D);
        Cov2DCtx c2;
        compute_cov2d(t, fx, fy, view->tanfovx, view->tanfovy, ge.cov3D, V, c2);
        real det = c2.a * c2.c - c2.b * c2.b;
        if (det == 0.0f) continue;                                     // Q3
        real det_inv = 1.f / det;
        ge.con[0] = c2.c * det_inv; ge.con[1] = -c2.b * det_inv; ge.con[2] = c2.a * det_inv;
        ge.con_cond = (std::fabs(c2.a * c2.c) + c2.b * c2.b) / std::fabs(det);
        real mid = 0.5f * (c2.a + c2.c);
        real root = std::sqrt(std::max(RL(0.1), mid * mid - det));       // Q4
        real lam1 = mid + root, lam2 = mid - root;
        real my_radius = std::ceil(3.f * std::sqrt(std::max(lam1, lam2)));
        ge.px = ((ndc_x + 1.0f) * W - 1.0f) * 0.5f;
        ge.py = ((ndc_y + 1.0f) * H - 1.0f) * 0.5f;
        // MS-GS pixel size, written before any filtering (SPEC M1)
        real o = g->opacities[i];
        ge.pixel_size = pixel_size_of(o, ge.con[0], ge.con[2]);
        pixel_sizes[i] = ge.pixel_size;
        ge.rect[0] = std::min(gx, std::max(0, (int)((ge.px - my_radius) / TILE)));
        ge.rect[1] = std::min(gy, std::max(0, (int)((ge.py - my_radius) / TILE)));
        ge.rect[2] = std::min(gx, std::max(0, (int)((ge.px + my_radius + TILE - 1) / TILE)));
        ge.rect[3] = std::min(gy, std::max(0, (int)((ge.py + my_radius + TILE - 1) / TILE)));
        if ((ge.rect[2] - ge.rect[0]) * (ge.rect[3] - ge.rect[1]) == 0) continue;
        const real minps_i = g->min_pixel_sizes ? g->min_pixel_sizes[i] : -1.f;
        const real maxps_i = g->max_pixel_sizes ? g->max_pixel_sizes[i] : -1.f;
        const bool base_i = g->base_mask ? g->base_mask[i] != 0 : false;
        real w = filter_weight(view, ge.pixel_size, minps_i, maxps_i, base_i);
        ge.filter_edge = filter_on_edge(view, ge.pixel_size, minps_i, maxps_i, base_i) ? 1 : 0;
        ge.weight = w;
        if (!(w > 0.f)) {                                              // SPEC M2/M3: dropped
            if (ge.filter_edge) { ge.ghost = 1; ge.opacity = o; }      // ... by a decision that could flip: listed, not blended
            continue;
        }
        ge.opacity = o * w;
        if (g->colors_precomp) {
            for (int c = 0; c < 3; ++c) { ge.rgb[c] = g->colors_precomp[3 * i + c]; ge.clamped[c] = 0; }
        } else {
            sh_to_rgb(view->sh_degree, view->sh_coeffs, g->shs + (size_t)3 * view->sh_coeffs * i, p,
                      view->campos, ge.rgb, ge.clamped);
        }
        ge.radius = (int32_t)my_radius;
        ge.visible = 1;
        radii[i] = ge.radius;
    }

    // ---- K2/K3 duplicate with keys (App. A.2), K4 stable sort ----
    std::vector<uint64_t> offs(P + 1, 0);
    for (int i = 0; i < P; ++i) {
        const Geom& ge = st->geom[i];
        uint64_t n = (ge.visible || ge.ghost) ? (uint64_t)(ge.rect[2] - ge.rect[0]) * (ge.rect[3] - ge.rect[1]) : 0;
        offs[i + 1] = offs[i] + n;
        if (ge.ghost) st->ghost_instances += (int64_t)n;
    }
    const uint64_t D = offs[P];
    struct KV { uint64_t key; uint32_t val; };
    std::vector<KV> kv(D);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < P; ++i) {
        const Geom& ge = st->geom[i];
        if (!ge.visible && !ge.ghost) continue;
        uint64_t o = offs[i];
        for (int y = ge.rect[1]; y < ge.rect[3]; ++y)
            for (int x = ge.rect[0]; x < ge.rect[2]; ++x) {
                uint64_t key = (uint64_t)(y * gx + x);
                key = (key << 32) | float_bits(ge.depth32);               // Q10
                kv[o++] = KV{key, (uint32_t)i};
            }
    }
    __gnu_parallel::stable_sort(kv.begin(), kv.end(), [](const KV& a, const KV& b) { return a.key < b.key; });
    st->list.resize(D);
    st->range_lo.assign((size_t)gx * gy, 0);
    st->range_hi.assign((size_t)gx * gy, 0);
    for (uint64_t i = 0; i < D; ++i) {                                  // K5
        st->list[i] = kv[i].val;
        uint32_t tile = (uint32_t)(kv[i].key >> 32);
        if (i == 0 || (uint32_t)(kv[i - 1].key >> 32) != tile) st->range_lo[tile] = (uint32_t)i;
        if (i + 1 == D || (uint32_t)(kv[i + 1].key >> 32) != tile) st->range_hi[tile] = (uint32_t)(i + 1);
    }
    std::vector<KV>().swap(kv);

    // ---- K6 blend forward (App. A.2) ----
    st->final_T.assign((size_t)W * H, 1.0f);
    st->n_contrib.assign((size_t)W * H, 0);
    st->borderline_gauss.assign(P, 0);
    st->filter_edge.assign(P, 0);
    st->shared_gauss.assign(P, 0);
    for (int i = 0; i < P; ++i) st->filter_edge[i] = st->geom[i].filter_edge;
    const real bg[3] = {view->bg[0], view->bg[1], view->bg[2]};
    int64_t traversed = 0, valid_pairs = 0, evaluated_pairs = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : traversed, valid_pairs, evaluated_pairs)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t lo = st->range_lo[tile], hi = st->range_hi[tile];
        uint32_t tile_max = 0;
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int x = tx * TILE + lx, y = ty * TILE + ly;
                if (x >= W || y >= H) continue;
                const real pxf = (real)x, pyf = (real)y;
                real T = 1.0f, C[3] = {0, 0, 0}, aps = 0.f, adp = 0.f;
                uint32_t contributor = 0, last = 0;
                bool flag = false, taint = false;
                // T_hi: the largest transmittance ANY float32 implementation of the algorithm can hold at this point of the
                // list — the same product with every entry whose blending hinges on a borderline decision left out (an alpha
                // at 1/255, a filter-edge Gaussian).  Where the oracle terminates, another implementation may still be
                // blending as long as T_hi (1 - alpha) has not fallen below 1e-4: the flags cover the list that far.
                real T_hi = 1.0f;
                uint32_t k_end = lo;
                bool done = false;             // the ORACLE's pixel has terminated (Q7); the scan goes on in shadow
                for (uint32_t k = lo; k < hi; ++k) {
                    const Geom& ge = st->geom[st->list[k]];
                    if (!done) { k_end = k + 1; ++contributor; if (!ge.ghost) ++evaluated_pairs; }
                    real dx = ge.px - pxf, dy = ge.py - pyf;
                    real power = -0.5f * (ge.con[0] * dx * dx + ge.con[2] * dy * dy) - ge.con[1] * dx * dy;
                    // the sign of the exponent is undecided in float32 and the Gaussian would be blended (alpha ~ opacity there)
                    const bool power_edge = std::fabs(power) <= power_sign_window(ge, dx, dy) &&
                                            ge.opacity * 255.0f >= 1.0f - RL(2e-5);
                    if (power > 0.0f) {
                        if (power_edge && !done && !ge.ghost) {                  // skipped here, blended by another implementation
                            flag = true;
                            uint8_t* bg_flag = &st->borderline_gauss[st->list[k]];
#pragma omp atomic write
                            *bg_flag = 1;
                        }
                        continue;
                    }
                    real alpha = std::min(RL(0.99), ge.opacity * oracle_exp(power));        // Q6
                    const real win = alpha_window(ge, dx, dy, power);
                    const bool reaches = alpha * 255.0f >= 1.0f - win;
                    const bool own_edge = std::fabs(alpha * 255.0f - 1.0f) < win || power_edge;
                    if (done) {
                        // shadow: only how far another implementation can get
                        if (!reaches) continue;
                        const real th = T_hi * (1 - alpha);
                        if (th < RL(0.0001) - RL(2e-8)) break;
                        k_end = k + 1;
                        if (!ge.filter_edge && !own_edge) T_hi = th;
                        continue;
                    }
                    // a Gaussian whose filter decision could flip reaches this pixel: the pixel, and every Gaussian
                    // blended here, depends on that decision
                    if (ge.filter_edge && reaches) taint = true;
                    if (ge.ghost) continue;
                    if (own_edge) {
                        flag = true;
                        uint8_t* bg_flag = &st->borderline_gauss[st->list[k]];
#pragma omp atomic write
                        *bg_flag = 1;
                    }
                    if (alpha < RL(1.0) / RL(255.0)) continue;                                 // Q7
                    ++valid_pairs;
                    real test_T = T * (1 - alpha);
                    if (std::fabs(test_T - RL(0.0001)) < RL(2e-8)) flag = true;
                    if (test_T < RL(0.0001)) {                                              // Q7: not blended
                        done = true;
                        if (!(flag || taint)) break;          // nothing undecided at this pixel: no shadow needed
                        const real th = T_hi * (1 - alpha);
                        if (th < RL(0.0001) - RL(2e-8)) break;
                        if (!ge.filter_edge && !own_edge) T_hi = th;
                        continue;
                    }
                    real wgt = alpha * T;
                    for (int c = 0; c < 3; ++c) C[c] += ge.rgb[c] * wgt;
                    aps += ge.pixel_size * wgt;                                          // SPEC M6
                    adp += ge.depth * wgt;
                    T = test_T;
                    if (!ge.filter_edge && !own_edge) T_hi *= (1 - alpha);
                    last = contributor;
                }
                if (taint) flag = true;
                if (flag) {
                    // every Gaussian that reaches this pixel inside the range any implementation may traverse: its term here
                    // moves with the undecided entry.  filter-edge taint: tier 1 (borderline_gauss, excluded from the strict
                    // gradient check) as since round 4; an alpha / T decision: tier 2 (shared_gauss), which the strict checks
                    // still cover — it EXPLAINS an exceedance, it does not excuse one in advance (tests/parity_utils.py)
                    for (uint32_t k = lo; k < k_end; ++k) {
                        const uint32_t id = st->list[k];
                        const Geom& ge = st->geom[id];
                        real dx = ge.px - pxf, dy = ge.py - pyf;
                        real power = -0.5f * (ge.con[0] * dx * dx + ge.con[2] * dy * dy) - ge.con[1] * dx * dy;
                        if (power > 0.0f) continue;
                        if (std::min(RL(0.99), ge.opacity * oracle_exp(power)) * 255.0f < 1.0f - alpha_window(ge, dx, dy, power)) continue;
                        uint8_t* sh_flag = &st->shared_gauss[id];
#pragma omp atomic write
                        *sh_flag = 1;
                        if (taint) {
                            uint8_t* bg_flag = &st->borderline_gauss[id];
#pragma omp atomic write
                            *bg_flag = 1;
                        }
                    }
                }
                const size_t pix = (size_t)y * W + x;
                st->final_T[pix] = T;
                st->n_contrib[pix] = last;
                tile_max = std::max(tile_max, last);
                for (int c = 0; c < 3; ++c) out_color[(size_t)c * H * W + pix] = C[c] + T * bg[c];
                out_acc_ps[pix] = aps;
                out_depth[pix] = adp;
                if (borderline) borderline[pix] = flag ? 1 : 0;
            }
        traversed += tile_max;
    }
    st->traversed = traversed;
    st->valid_pairs = valid_pairs;
    st->evaluated_pairs = evaluated_pairs;

    // flat copies for the tests
    st->depths.resize(P); st->conic_opacity.resize((size_t)4 * P); st->rgb.resize((size_t)3 * P);
    st->means2D.resize((size_t)2 * P); st->cov3D.resize((size_t)6 * P); st->rects.resize((size_t)4 * P);
    for (int i = 0; i < P; ++i) {
        const Geom& ge = st->geom[i];
        st->depths[i] = ge.depth;
        for (int c = 0; c < 3; ++c) { st->conic_opacity[4 * i + c] = ge.con[c]; st->rgb[3 * i + c] = ge.rgb[c]; }
        st->conic_opacity[4 * i + 3] = ge.opacity;
        st->means2D[2 * i] = ge.px; st->means2D[2 * i + 1] = ge.py;
        for (int c = 0; c < 6; ++c) st->cov3D[6 * i + c] = ge.cov3D[c];
        for (int c = 0; c < 4; ++c) st->rects[4 * i + c] = ge.rect[c];
    }
    if (state_out) *state_out = st; else delete st;
    return MSGS_OK;
}

// sums2d_out (nullable, [P,9] doubles): the per-Gaussian 2-D gradient accumulators as they stand between the blend
// backward (K7) and the per-Gaussian backward (K8 + K9): {dL/dmean2D x, y (NDC-ish units), dL/dconic A, B, C,
// dL/dopacity_eff, dL/drgb[3]}.  The isolation test of tests/test_k8_isolation_gpu.py feeds exactly these to the HIP
// per-Gaussian backward (msgs_backward_per_gaussian) and compares the two K8 + K9 stages on identical inputs.
extern "C" int msgs_oracle_backward_ex(const msgs_oracle_state_t* st, const msgs_view_t* view,
                                       const msgs_gaussians_t* g, const float* dL_dcolor,
                                       const msgs_grads_t* grads, int num_threads, double* sums2d_out) {
    if (!st || !view || !g || !grads || !dL_dcolor) return MSGS_ERR_INVALID_ARG;
    if (num_threads > 0) omp_set_num_threads(num_threads);
    const int P = st->P, W = st->W, H = st->H, gx = st->gx, gy = st->gy;
    // per-Gaussian 2-D gradient accumulators.  Accumulated in double with atomics so that the
    // result does not depend on the thread schedule beyond 1e-16 relative.
    struct Acc { double mean2D[2], conic[3], opacity, color[3]; };
    std::vector<Acc> acc(P, Acc{});
    const real bg[3] = {view->bg[0], view->bg[1], view->bg[2]};
    const real ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // ---- K7 blend backward (App. A.3) ----
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t lo = st->range_lo[tile];
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                const int x = tx * TILE + lx, y = ty * TILE + ly;
                if (x >= W || y >= H) continue;
                const size_t pix = (size_t)y * W + x;
                const real pxf = (real)x, pyf = (real)y;
                const real T_final = st->final_T[pix];
                real T = T_final;
                const uint32_t last = st->n_contrib[pix];
                real dL_dpixel[3];
                for (int c = 0; c < 3; ++c) dL_dpixel[c] = dL_dcolor[(size_t)c * H * W + pix];
                real accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.f;
                for (uint32_t j = last; j-- > 0;) {
                    const uint32_t id = st->list[lo + j];
                    const Geom& ge = st->geom[id];
                    if (ge.ghost) continue;
                    real dx = ge.px - pxf, dy = ge.py - pyf;
                    real power = -0.5f * (ge.con[0] * dx * dx + ge.con[2] * dy * dy) - ge.con[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const real G = oracle_exp(power);
                    const real alpha = std::min(RL(0.99), ge.opacity * G);
                    if (alpha < RL(1.0) / RL(255.0)) continue;
                    T = T / (1.f - alpha);
                    const real dchannel_dcolor = alpha * T;
                    real dL_dalpha = 0.0f;
                    real dcol[3];
                    for (int c = 0; c < 3; ++c) {
                        accum_rec[c] = last_alpha * last_color[c] + (1.f - last_alpha) * accum_rec[c];
                        last_color[c] = ge.rgb[c];
                        dL_dalpha += (ge.rgb[c] - accum_rec[c]) * dL_dpixel[c];
                        dcol[c] = dchannel_dcolor * dL_dpixel[c];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    real bg_dot = 0.f;
                    for (int c = 0; c < 3; ++c) bg_dot += bg[c] * dL_dpixel[c];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    const real dL_dG = ge.opacity * dL_dalpha;                // Q6: through the clamp
                    const real gdx = G * dx, gdy = G * dy;
                    const real dG_ddelx = -gdx * ge.con[0] - gdy * ge.con[1];
                    const real dG_ddely = -gdy * ge.con[2] - gdx * ge.con[1];
                    Acc& a = acc[id];
                    const double v[9] = {(double)(dL_dG * dG_ddelx * ddelx_dx), (double)(dL_dG * dG_ddely * ddely_dy),
                                         (double)(-0.5f * gdx * dx * dL_dG), (double)(-0.5f * gdx * dy * dL_dG),
                                         (double)(-0.5f * gdy * dy * dL_dG), (double)(G * dL_dalpha),
                                         (double)dcol[0], (double)dcol[1], (double)dcol[2]};
                    double* dst[9] = {&a.mean2D[0], &a.mean2D[1], &a.conic[0], &a.conic[1], &a.conic[2],
                                      &a.opacity, &a.color[0], &a.color[1], &a.color[2]};
                    for (int k = 0; k < 9; ++k) {
#pragma omp atomic
                        *dst[k] += v[k];
                    }
                }
            }
    }

    if (sums2d_out) {
        for (int i = 0; i < P; ++i) {
            const Acc& a = acc[i];
            double* o = sums2d_out + (size_t)9 * i;
            o[0] = a.mean2D[0]; o[1] = a.mean2D[1]; o[2] = a.conic[0]; o[3] = a.conic[1]; o[4] = a.conic[2];
            o[5] = a.opacity; o[6] = a.color[0]; o[7] = a.color[1]; o[8] = a.color[2];
        }
    }

    // ---- K8 (2-D covariance backward) + K9 (preprocess backward), App. A.3 ----
    const real fx = W / (2.0f * (real)view->tanfovx), fy = H / (2.0f * (real)view->tanfovy);
    const in_t* V = view->viewmatrix;
    const in_t* PM = view->projmatrix;
    // gradient outputs in `real` (float64 build: double buffers behind the float* fields of msgs_grads_t)
    real* const G_means2D = reinterpret_cast<real*>(grads->dL_dmeans2D);
    real* const G_opac = reinterpret_cast<real*>(grads->dL_dopacities);
    real* const G_shs = reinterpret_cast<real*>(grads->dL_dshs);
    real* const G_colors = reinterpret_cast<real*>(grads->dL_dcolors);
    real* const G_scales = reinterpret_cast<real*>(grads->dL_dscales);
    real* const G_rot = reinterpret_cast<real*>(grads->dL_drotations);
    real* const G_cov3D = reinterpret_cast<real*>(grads->dL_dcov3D);
    real* const G_means3D = reinterpret_cast<real*>(grads->dL_dmeans3D);
    const int K = view->sh_coeffs;
    const int deg = view->sh_degree;
    const char* exact_env = std::getenv("MSGS_ORACLE_EXACT_DET");
    const bool exact_det = exact_env && exact_env[0] == '1';
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        const Geom& ge = st->geom[i];
        real dmean[3] = {0, 0, 0};
        if (G_means2D) { G_means2D[3 * i] = 0; G_means2D[3 * i + 1] = 0; G_means2D[3 * i + 2] = 0; }
        if (G_opac) G_opac[i] = 0.f;
        if (G_shs) std::memset(G_shs + (size_t)3 * K * i, 0, sizeof(real) * 3 * K);
        if (G_colors) std::memset(G_colors + 3 * i, 0, 3 * sizeof(real));
        if (G_scales) std::memset(G_scales + 3 * i, 0, 3 * sizeof(real));
        if (G_rot) std::memset(G_rot + 4 * i, 0, 4 * sizeof(real));
        if (G_cov3D) std::memset(G_cov3D + 6 * i, 0, 6 * sizeof(real));
        if (G_means3D) std::memset(G_means3D + 3 * i, 0, 3 * sizeof(real));
        if (!ge.visible) continue;
        const Acc& a = acc[i];
        const real g2x = (real)a.mean2D[0], g2y = (real)a.mean2D[1];
        const real gA = (real)a.conic[0], gBh = (real)a.conic[1], gC = (real)a.conic[2];
        const in_t* p = g->means3D + 3 * i;

        // -- 2-D covariance backward --
        real t[3];
        view_point(V, p, t);
        Cov2DCtx c2;
        compute_cov2d(t, fx, fy, view->tanfovx, view->tanfovy, ge.cov3D, V, c2);
        const real ca = c2.a, cb = c2.b, cc = c2.c;
        const real denom = ca * cc - cb * cb;
        // Q9.  (MSGS_ORACLE_EXACT_DET=1 drops the 1e-7 — the exact derivative of 1 / det, what autograd forms: only for the
        // test that checks this hand-derived backward against the autograd oracle to rounding, tests/test_oracle_cpu.py)
        const real denom2inv = 1.0f / ((denom * denom) + (exact_det ? RL(0.0) : RL(0.0000001)));
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real dcov[6] = {0, 0, 0, 0, 0, 0};
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * gA + 2 * cb * cc * gBh + (denom - ca * cc) * gC);
            dL_dc = denom2inv * (-ca * ca * gC + 2 * ca * cb * gBh + (denom - ca * cc) * gA);
            dL_db = denom2inv * 2 * (cb * cc * gA - (denom + 2 * cb * cb) * gBh + ca * cb * gC);
            const real(*T)[3] = c2.T;
            dcov[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
            dcov[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
            dcov[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
            dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        }
        {
            const real S[3][3] = {{ge.cov3D[0], ge.cov3D[1], ge.cov3D[2]}, {ge.cov3D[1], ge.cov3D[3], ge.cov3D[4]},
                                   {ge.cov3D[2], ge.cov3D[4], ge.cov3D[5]}};
            real ST0[3], ST1[3];
            for (int k = 0; k < 3; ++k) {
                ST0[k] = S[k][0] * c2.T[0][0] + S[k][1] * c2.T[0][1] + S[k][2] * c2.T[0][2];
                ST1[k] = S[k][0] * c2.T[1][0] + S[k][1] * c2.T[1][1] + S[k][2] * c2.T[1][2];
            }
            real dT0[3], dT1[3];
            for (int k = 0; k < 3; ++k) {
                dT0[k] = 2 * ST0[k] * dL_da + ST1[k] * dL_db;
                dT1[k] = 2 * ST1[k] * dL_dc + ST0[k] * dL_db;
            }
            // Wr[k][c] = V[4*c + k]
            auto Wr = [&](int k, int c) { return V[4 * c + k]; };
            const real dJ00 = Wr(0, 0) * dT0[0] + Wr(0, 1) * dT0[1] + Wr(0, 2) * dT0[2];
            const real dJ02 = Wr(2, 0) * dT0[0] + Wr(2, 1) * dT0[1] + Wr(2, 2) * dT0[2];
            const real dJ11 = Wr(1, 0) * dT1[0] + Wr(1, 1) * dT1[1] + Wr(1, 2) * dT1[2];
            const real dJ12 = Wr(2, 0) * dT1[0] + Wr(2, 1) * dT1[1] + Wr(2, 2) * dT1[2];
            const real tz = 1.f / c2.tz, tz2 = tz * tz, tz3 = tz2 * tz;
            const real dtx = c2.x_mul * -fx * tz2 * dJ02;                                  // Q2
            const real dty = c2.y_mul * -fy * tz2 * dJ12;
            const real dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * c2.tx_c) * tz3 * dJ02 +
                              (2 * fy * c2.ty_c) * tz3 * dJ12;
            // dL/dp_j = sum_i W2C[i][j] dL/dt_i = sum_i V[4*j + i] dt_i
            for (int j = 0; j < 3; ++j) dmean[j] += V[4 * j + 0] * dtx + V[4 * j + 1] * dty + V[4 * j + 2] * dtz;
        }

        // -- projection backward --
        {
            real h[4];
            proj_point(PM, p, h);
            const real m_w = 1.0f / (h[3] + RL(0.0000001));
            const real mul1 = h[0] * m_w * m_w, mul2 = h[1] * m_w * m_w;
            for (int j = 0; j < 3; ++j)
                dmean[j] += (PM[4 * j + 0] * m_w - PM[4 * j + 3] * mul1) * g2x +
                            (PM[4 * j + 1] * m_w - PM[4 * j + 3] * mul2) * g2y;
        }
        if (G_means2D) { G_means2D[3 * i] = g2x; G_means2D[3 * i + 1] = g2y; }
        if (G_opac) G_opac[i] = ge.weight * (real)a.opacity;      // SPEC M4

        // -- colour backward --
        real dcolr[3] = {(real)a.color[0], (real)a.color[1], (real)a.color[2]};
        if (g->colors_precomp) {
            if (G_colors) for (int c = 0; c < 3; ++c) G_colors[3 * i + c] = dcolr[c];
        } else {
            for (int c = 0; c < 3; ++c) if (ge.clamped[c]) dcolr[c] = 0.f;                      // Q8
            const in_t* sh = g->shs + (size_t)3 * K * i;
            real* dsh = G_shs ? G_shs + (size_t)3 * K * i : nullptr;
            const in_t* cam = view->campos;
            real dox = (real)p[0] - (real)cam[0], doy = (real)p[1] - (real)cam[1], doz = (real)p[2] - (real)cam[2];
            real len = std::sqrt(dox * dox + doy * doy + doz * doz);
            real x = dox / len, y = doy / len, z = doz / len;
            real basis[16], bdx[16], bdy[16], bdz[16];
            for (int k = 0; k < 16; ++k) basis[k] = bdx[k] = bdy[k] = bdz[k] = 0.f;
            basis[0] = SH_C0;
            if (deg > 0) {
                basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
                bdy[1] = -SH_C1; bdz[2] = SH_C1; bdx[3] = -SH_C1;
                if (deg > 1) {
                    real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2.f * zz - xx - yy);
                    basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
                    bdx[4] = SH_C2[0] * y; bdy[4] = SH_C2[0] * x;
                    bdy[5] = SH_C2[1] * z; bdz[5] = SH_C2[1] * y;
                    bdx[6] = SH_C2[2] * -2.f * x; bdy[6] = SH_C2[2] * -2.f * y; bdz[6] = SH_C2[2] * 4.f * z;
                    bdx[7] = SH_C2[3] * z; bdz[7] = SH_C2[3] * x;
                    bdx[8] = SH_C2[4] * 2.f * x; bdy[8] = SH_C2[4] * -2.f * y;
                    if (deg > 2) {
                        basis[9] = SH_C3[0] * y * (3.f * xx - yy); basis[10] = SH_C3[1] * xy * z;
                        basis[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                        basis[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                        basis[13] = SH_C3[4] * x * (4.f * zz - xx - yy); basis[14] = SH_C3[5] * z * (xx - yy);
                        basis[15] = SH_C3[6] * x * (xx - 3.f * yy);
                        bdx[9] = SH_C3[0] * 6.f * xy; bdy[9] = SH_C3[0] * (3.f * xx - 3.f * yy);
                        bdx[10] = SH_C3[1] * yz; bdy[10] = SH_C3[1] * xz; bdz[10] = SH_C3[1] * xy;
                        bdx[11] = SH_C3[2] * -2.f * xy; bdy[11] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); bdz[11] = SH_C3[2] * 8.f * yz;
                        bdx[12] = SH_C3[3] * -6.f * xz; bdy[12] = SH_C3[3] * -6.f * yz; bdz[12] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                        bdx[13] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); bdy[13] = SH_C3[4] * -2.f * xy; bdz[13] = SH_C3[4] * 8.f * xz;
                        bdx[14] = SH_C3[5] * 2.f * xz; bdy[14] = SH_C3[5] * -2.f * yz; bdz[14] = SH_C3[5] * (xx - yy);
                        bdx[15] = SH_C3[6] * (3.f * xx - 3.f * yy); bdy[15] = SH_C3[6] * -6.f * xy;
                    }
                }
            }
            const int ncoef = (deg + 1) * (deg + 1);
            real ddir[3] = {0, 0, 0};
            for (int k = 0; k < ncoef; ++k)
                for (int c = 0; c < 3; ++c) {
                    if (dsh) dsh[k * 3 + c] = basis[k] * dcolr[c];
                    const real s = (real)sh[k * 3 + c] * dcolr[c];
                    ddir[0] += bdx[k] * s; ddir[1] += bdy[k] * s; ddir[2] += bdz[k] * s;
                }
            // d normalize: (I - d d^T) / len
            const real dotv = x * ddir[0] + y * ddir[1] + z * ddir[2];
            dmean[0] += (ddir[0] - x * dotv) / len;
            dmean[1] += (ddir[1] - y * dotv) / len;
            dmean[2] += (ddir[2] - z * dotv) / len;
        }
        if (G_means3D) for (int j = 0; j < 3; ++j) G_means3D[3 * i + j] = dmean[j];

        // -- 3-D covariance backward --
        if (g->cov3D_precomp) {
            if (G_cov3D) for (int c = 0; c < 6; ++c) G_cov3D[6 * i + c] = dcov[c];
        } else {
            const in_t* q = g->rotations + 4 * i;
            const in_t* s = g->scales + 3 * i;
            const real mod = view->scale_modifier;
            real r = q[0], x = q[1], y = q[2], z = q[3];
            real R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                             {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                             {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
            real S[3] = {mod * s[0], mod * s[1], mod * s[2]};
            real Gm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                              {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                              {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            real dM[3][3], dR[3][3];
            for (int a2 = 0; a2 < 3; ++a2)
                for (int b2 = 0; b2 < 3; ++b2) {
                    real acc2 = 0.f;
                    for (int k = 0; k < 3; ++k) acc2 += Gm[a2][k] * (R[k][b2] * S[b2]);
                    dM[a2][b2] = 2.f * acc2;
                }
            for (int j = 0; j < 3; ++j) {
                real ds = dM[0][j] * R[0][j] + dM[1][j] * R[1][j] + dM[2][j] * R[2][j];
                if (G_scales) G_scales[3 * i + j] = mod * ds;
                for (int a2 = 0; a2 < 3; ++a2) dR[a2][j] = dM[a2][j] * S[j];
            }
            if (G_rot) {
                real* dq = G_rot + 4 * i;
                dq[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
                dq[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
                dq[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
                dq[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
            }
        }
    }
    return MSGS_OK;
}

extern "C" int msgs_oracle_backward(const msgs_oracle_state_t* st, const msgs_view_t* view,
                                    const msgs_gaussians_t* g, const float* dL_dcolor,
                                    const msgs_grads_t* grads, int num_threads) {
    return msgs_oracle_backward_ex(st, view, g, dL_dcolor, grads, num_threads, nullptr);
}

extern "C" int64_t msgs_oracle_num_instances(const msgs_oracle_state_t* s) { return (int64_t)s->list.size() - s->ghost_instances; }
extern "C" const uint8_t* msgs_oracle_filter_edge(const msgs_oracle_state_t* s) { return s->filter_edge.data(); }
extern "C" const uint8_t* msgs_oracle_shared_borderline_gaussians(const msgs_oracle_state_t* s) { return s->shared_gauss.data(); }
extern "C" int64_t msgs_oracle_traversed(const msgs_oracle_state_t* s) { return s->traversed; }
extern "C" int64_t msgs_oracle_valid_pairs(const msgs_oracle_state_t* s) { return s->valid_pairs; }
extern "C" int64_t msgs_oracle_evaluated_pairs(const msgs_oracle_state_t* s) { return s->evaluated_pairs; }
extern "C" const float* msgs_oracle_final_T(const msgs_oracle_state_t* s) { return reinterpret_cast<const float*>(s->final_T.data()); }
extern "C" const uint32_t* msgs_oracle_n_contrib(const msgs_oracle_state_t* s) { return s->n_contrib.data(); }
extern "C" const float* msgs_oracle_depths(const msgs_oracle_state_t* s) { return reinterpret_cast<const float*>(s->depths.data()); }
extern "C" const float* msgs_oracle_conic_opacity(const msgs_oracle_state_t* s) { return reinterpret_cast<const float*>(s->conic_opacity.data()); }
extern "C" const float* msgs_oracle_rgb(const msgs_oracle_state_t* s) { return reinterpret_cast<const float*>(s->rgb.data()); }
extern "C" const float* msgs_oracle_means2D(const msgs_oracle_state_t* s) { return reinterpret_cast<const float*>(s->means2D.data()); }
extern "C" const float* msgs_oracle_cov3D(const msgs_oracle_state_t* s) { return reinterpret_cast<const float*>(s->cov3D.data()); }
extern "C" const int32_t* msgs_oracle_rects(const msgs_oracle_state_t* s) { return s->rects.data(); }
extern "C" const uint8_t* msgs_oracle_borderline_gaussians(const msgs_oracle_state_t* s) { return s->borderline_gauss.data(); }
extern "C" void msgs_oracle_free(msgs_oracle_state_t* s) { delete s; }
