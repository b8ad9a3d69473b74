/*
 * msgs.h — C ABI of libmsgs_hip.so, the MI355X (gfx950) multi-scale 3D-Gaussian rasterizer.
 *
 * This is the drop-in boundary for the reference's un-vendored native module
 * `diff_gaussian_rasterization._C` (imported at /root/reference/gaussian_renderer/__init__.py:14,
 * constructed :37-55, called :94-108).  The reference binds that module with pybind11/torch; this
 * library is bound with ctypes from ms-gs_amd/diff_gaussian_rasterization/__init__.py (see
 * INTEGRATION.md for the stub a maintainer adds).  Plain pointers and sizes only: no torch types,
 * no C++ types, nothing thrown across the boundary.
 *
 *   reference entry point (upstream name)                 replaced by
 *   ---------------------------------------------------   ---------------------------------------
 *   _C.rasterize_gaussians(...)            (forward)      msgs_forward_stage1 + msgs_forward_stage2
 *   _C.rasterize_gaussians_backward(...)   (backward)     msgs_backward
 *   _C.mark_visible(...)                   (unused by the reference)
 *                                                          msgs_mark_visible
 *   resize-callback byte tensors (geom/binning/img state)  msgs_*_bytes size queries; the CALLER
 *                                                          allocates (torch caching allocator) and
 *                                                          keeps the buffers alive for backward
 *
 * Ownership: every pointer is owned by the caller; the library never allocates device memory,
 * never frees or retains a pointer past the call.  All `const float*` / `float*` arguments are
 * DEVICE pointers unless the name says `host`.  Calls are asynchronous on `stream` except that
 * msgs_forward_stage1 synchronises the stream once to return the instance count (the same D2H
 * `num_rendered` read the reference's forward performs).
 *
 * Threading: re-entrant (forward is called from the Python main thread, backward from PyTorch's
 * autograd thread).  Not stateless:
 *   - four PROCESS-WIDE mode flags: msgs_set_deterministic, msgs_set_backward_generation, msgs_set_blend_granularity,
 *     msgs_set_occlusion (the last three only choose between code paths that give the same results);
 *   - six environment variables latched on first use (process-wide): the initial values of the four flags
 *     (MSGS_DETERMINISTIC, MSGS_BWD_GEN, MSGS_BLEND_GRANULARITY, MSGS_NO_OCCLUSION) and MSGS_BLOCKING_SYNC,
 *     MSGS_NO_SPECULATIVE_STAGE2 — how the host learns the instance count; none changes a result.
 *     The A/B switches of rounds 1-5 (sort / scan / emit / forward-list variants, MSGS_DEPTH_SORT_BEGIN_BIT) are gone from the
 *     product together with the code paths they selected;
 *   - one 64-byte pinned status block per calling host thread for the calls that WAIT for the instance count
 *     (msgs_forward, msgs_forward_stage1), and one per msgs_status_t handle for msgs_forward_launch /
 *     msgs_forward_finish — any number of forwards may be in flight from one host thread on any streams, one per handle.
 *
 * Return value: 0 = MSGS_OK; < 0 = invalid argument / capacity (MSGS_ERR_*); > 0 = hipError_t.
 */
#ifndef MSGS_H_
#define MSGS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSGS_ABI_VERSION 11

#define MSGS_OK 0
#define MSGS_ERR_INVALID_ARG (-1)  /* NULL / inconsistent pointers (both or neither of shs|colors, ...) */
#define MSGS_ERR_CAPACITY (-2)     /* a caller-supplied buffer is smaller than the size query says   */
#define MSGS_ERR_TOO_MANY (-3)     /* more than 2^32-1 tile instances                                */
#define MSGS_ERR_SH_DEGREE (-4)    /* sh_degree outside 0..3 or (sh_degree+1)^2 > sh_coeffs          */
#define MSGS_ERR_INTERNAL (-5)     /* a bounded inter-workgroup wait expired (sort/scan look-back)     */

#define MSGS_TILE 16               /* 16x16 pixel tiles                     */

/* View / raster settings: the 15 fields of GaussianRasterizationSettings
 * (/root/reference/gaussian_renderer/__init__.py:37-53). */
typedef struct msgs_view {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    float fade_size;
    int32_t sh_degree;        /* active degree 0..3                                                 */
    int32_t sh_coeffs;        /* K: coefficients stored per Gaussian in `shs` ((max_degree+1)^2)    */
    int32_t filter_small;     /* bool */
    int32_t filter_large;     /* bool */
    int32_t prefiltered;      /* bool (accepted; the reference always passes False)                  */
    int32_t debug;            /* bool: synchronise + return the first HIP error after every stage   */
    int32_t no_heavy_queue;   /* bool (not a reference field), a performance hint only: "an ordinary view" — Gaussians with more than
                               * 96 tile instances are emitted by their wave inside the emit kernel instead of being queued for a
                               * second launch that gives each a workgroup (binning.hip).  The queue pays off when opaque covers
                               * closed blocks and the nearest depth ranks are all giants (0.28 -> 0.05 ms on the multi-scale model
                               * rendered without its filters); on an ordinary view it is one wasted launch (~3 us).  A caller that
                               * renders similar views in a loop sets it while msgs_forward_info says no block closed lately.
                               * Same outputs either way                                                                      */
    int32_t feedback_tag;     /* (not a reference field) non-zero: the forward ends with one small kernel that publishes what
                               * this view cost — instances D, traversed tile-list entries D_trav (sum over tiles of the last list
                               * position any pixel blended), and the slab numbers below — together with this tag in the pinned
                               * status block; msgs_forward_info returns the most recent publication, so a caller that renders
                               * similar views in a loop learns, one frame late and without any synchronisation, whether depth
                               * slabs pay off for that kind of view.  0: nothing is published (no extra launch)               */
    float slab_fraction;      /* (not a reference field) > 0: DEPTH-SLAB BINNING (round 6).  Stage 2 first bins and blends only the
                               * nearest depth ranks — up to this fraction of the D tile instances (slab A) —; the forward blend
                               * marks the tiles in which a pixel is still blending at the end of its slab-A list; the rest of
                               * the view (slab B) is then counted, emitted and sorted into THOSE tiles only (their complete
                               * lists, slab-A entries included) and they are blended again over the complete list.  A tile
                               * whose every pixel terminated inside slab A never evaluates an entry behind it (Q7), so every
                               * output, n_contrib, final_T and every gradient is bit-identical to the single-pass path; what
                               * shrinks is the work of emit / tile sort / ranges on instances nobody walks (BASELINE C5: 54.9 M
                               * instances, 4.35 M walked).  Needs binning >= msgs_binning_bytes_slab(D, W, H, fraction); ignored
                               * (single pass) with fewer than 2048 tiles, in deterministic mode, with fine blend granularity
                               * forced, or when the buffers are too small.  0: single pass.  With the two-call form pass the
                               * SAME value to msgs_forward_stage1: its scan then hands slab B's count the coarse screen-cell
                               * range of every Gaussian in depth order; without them (stage 1 called with 0) slab B still gives
                               * the same results but fetches every Gaussian's record to learn that it has nothing to add   */
    int32_t reserved1;        /* 0 */
    const float* bg;          /* [3]   device                                                        */
    const float* viewmatrix;  /* [16]  device; world_view_transform = W2C^T row-major (cameras.py:54) */
    const float* projmatrix;  /* [16]  device; full_proj_transform (cameras.py:55-56)                */
    const float* campos;      /* [3]   device; camera_center (cameras.py:57)                         */
} msgs_view_t;

/* Per-Gaussian inputs: the 13 kwargs of GaussianRasterizer.forward
 * (/root/reference/gaussian_renderer/__init__.py:95-107).  means2D carries no data forward (it is
 * the gradient sink, :27-31) and therefore only appears in msgs_backward. */
typedef struct msgs_gaussians {
    int32_t P;
    int32_t raw_params;            /* 0: activated inputs, the reference's op API.  1: RAW GaussianModel parameters
                                    * (opt-in): opacities = logits, scales = log-scales,
                                    * rotations = un-normalised quaternions, SH given as features_dc + features_rest
                                    * (shs / colors_precomp / cov3D_precomp must be NULL, sh_coeffs = 16); the
                                    * activations of scene/gaussian_model.py:39-47,127-153 (exp, sigmoid, normalize)
                                    * and the torch.cat of :144-149 are evaluated inside the kernels and msgs_backward
                                    * returns gradients w.r.t. the raw parameters.
                                    * 2: CHAINED gradients: opacities / scales / rotations are the ACTIVATED values
                                    * exactly as in mode 0 (the forward is bit-identical to mode 0), the SH
                                    * coefficients are read from features_dc + features_rest (whose concatenation the
                                    * reference passes as `shs`; `shs` may be given as well and is then the one read:
                                    * aligned rows, only the visible ones), rotations_raw holds the raw quaternions, and
                                    * msgs_backward applies the chain rule of sigmoid / exp / normalize / cat itself:
                                    * dL_dopacities, dL_dscales, dL_drotations, dL_dfeatures_dc/_rest are gradients
                                    * w.r.t. the RAW parameters — what autograd would produce by running the backward
                                    * of the reference's getters after the op */
    const float* means3D;          /* [P,3]                                                          */
    const float* shs;              /* [P,K,3]  xor colors_precomp                                    */
    const float* colors_precomp;   /* [P,3]                                                          */
    const float* opacities;        /* [P] (the reference passes [P,1])                               */
    const float* scales;           /* [P,3]    with rotations, xor cov3D_precomp                     */
    const float* rotations;        /* [P,4]    (w,x,y,z), pre-normalised (gaussian_model.py:132-133) */
    const float* cov3D_precomp;    /* [P,6]    xx,xy,xz,yy,yz,zz (general_utils.py:64-73)            */
    const float* max_pixel_sizes;  /* [P]      -1 = unset (gaussian_model.py:224)                    */
    const float* min_pixel_sizes;  /* [P]      -1 = unset (gaussian_model.py:225)                    */
    const float* occ_multiplier;   /* [P,4]    never read: the caller guarantees all ones (SPEC M5)   */
    const float* dc_delta;         /* [P,12]   never read: the caller guarantees all zeros (SPEC M5)  */
    const uint8_t* base_mask;      /* [P]      bool                                                  */
    const float* features_dc;      /* [P,1,3]  modes 1, 2    (gaussian_model.py:55)                          */
    const float* features_rest;    /* [P,15,3] modes 1, 2    (gaussian_model.py:56)                          */
    const float* rotations_raw;    /* [P,4]    mode 2 only: the un-normalised quaternions (gaussian_model.py:58)  */
} msgs_gaussians_t;

/* Optimizer step INSIDE the per-Gaussian backward (msgs_grads_t::adam_in_backward; ABI 11).  The reference's iteration is
 * loss.backward() -> optimizer.step() (/root/reference/train.py:216, :416-418) with torch.optim.Adam(lr = 0.0, eps = 1e-15) over
 * the leaf tensors (/root/reference/scene/gaussian_model.py:235-248): backward writes 59 gradient floats per Gaussian (236 B at
 * SH degree 3, zeros for Gaussians the view did not render) and the optimizer reads them back.  With this struct the kernel that
 * FORMS the raw-parameter gradients applies the Adam update to the parameter and its two moments on the spot — same arithmetic,
 * term by term, as msgs_adam_step (bit-identical parameters and moments) — and the gradient tensors are neither written nor
 * read.  Raw mode 1 only (the gradients must be those of the leaves), scales + rotations and SH given as features_dc /
 * features_rest with sh_coeffs == 16, no accumulate, no factored SH gradient.  The parameters are the tensors of
 * msgs_gaussians_t (means3D, features_dc, features_rest, opacities, scales, rotations — written through their const pointers),
 * every row is updated (a Gaussian that was not rendered has gradient zero: its moments decay and it moves, as under
 * torch.optim.Adam with dense zero gradients).  All 18 tensors 16-byte aligned.  dL_dmeans2D is still stored. */
typedef struct msgs_adam_moments {
    float* exp_avg;            /* same shape as the parameter                                        */
    float* exp_avg_sq;
    double lr;                 /* this tensor's learning rate (param_group["lr"])                    */
} msgs_adam_moments_t;
typedef struct msgs_adam_in_backward {
    int64_t step;              /* the step being taken, >= 1 (bias corrections 1 - beta^step)        */
    double beta1, beta2, eps;
    msgs_adam_moments_t t[6];  /* means3D, features_dc, features_rest, opacities, scales, rotations  */
} msgs_adam_in_backward_t;

/* Gradient outputs of msgs_backward.  Every non-NULL buffer is fully written (zeros for
 * Gaussians that were not rendered), so the caller may pass uninitialised memory. */
typedef struct msgs_grads {
    float* dL_dmeans3D;        /* [P,3]                                                              */
    float* dL_dmeans2D;        /* [P,3]   (x,y in the NDC-ish units of SURVEY App. A.3; z = 0)       */
    float* dL_dshs;            /* [P,K,3] when shs was given                                         */
    float* dL_dcolors;         /* [P,3]   when colors_precomp was given; also in the raw modes with both
                                *          dL_dfeatures_* NULL: the FACTORED SH gradient — the clamp-masked dL/drgb
                                *          of this view, from which msgs_sh_grad_from_views rebuilds dL/dSH      */
    float* dL_dopacities;      /* [P]                                                                */
    float* dL_dscales;         /* [P,3]   when scales/rotations were given                           */
    float* dL_drotations;      /* [P,4]                                                              */
    float* dL_dcov3D;          /* [P,6]   when cov3D_precomp was given                               */
    float* dL_dfeatures_dc;    /* [P,1,3]  raw mode only                                                     */
    float* dL_dfeatures_rest;  /* [P,15,3] raw mode only                                                     */
    void* factors_ready;       /* optional hipEvent_t, factored SH gradient only: dL_dcolors is written by a kernel of its
                                * own BEFORE the per-Gaussian backward and this event is recorded right behind it, so that
                                * the caller can start the all-gather of the factors while the rest of msgs_backward runs */
    int32_t scratch_is_clear;  /* non-zero: the first msgs_backward_scratch_bytes(P) bytes of `scratch` were handed to the
                                * forward of THIS view as grad_records (it cleared them during the blend) and nothing has
                                * written them since: msgs_backward skips its own fill launch.  0: msgs_backward clears them */
    int32_t accumulate;        /* non-zero: ADD this view's gradients to what the output tensors hold (dL_dmeans3D, dL_dshs /
                                * dL_dfeatures_*, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D), leaving the
                                * rows of Gaussians this view did not render untouched — the views of one optimizer step
                                * accumulate into ONE gradient bucket without zero rows and without a separate accumulation
                                * pass (the reference's autograd does `param.grad += view_grad`: 3 x 236 B per Gaussian and
                                * view at SH degree 3).  dL_dmeans2D stays per view (always stored).  The first view of a step
                                * is called with 0 (it initialises every row).  Not with sh_coeffs != 16 on the reference API. */
    void* wait_before_accumulate; /* optional hipEvent_t: the per-Gaussian kernel waits for it — the `accumulated` event of the
                                * previous view into the same tensors when that view ran on ANOTHER stream */
    void* accumulated;         /* optional hipEvent_t recorded behind the per-Gaussian kernel */
    const msgs_adam_in_backward_t* adam_in_backward;   /* NULL (default) or the optimizer step to take inside the per-Gaussian
                                * kernel (above): dL_dmeans3D, dL_dfeatures_*, dL_dopacities, dL_dscales, dL_drotations are then
                                * ignored and may be NULL */
} msgs_grads_t;

/* Optional per-kernel timing (bench.py's roofline leg).  The caller owns the events; the library
 * records ev[2*k] before and ev[2*k+1] after kernel class k on `stream`.  NULL = no timing. */
enum {
    MSGS_K_PREPROCESS = 0, MSGS_K_DEPTH_SORT = 1, MSGS_K_SCAN = 2, MSGS_K_EMIT = 3,
    MSGS_K_TILE_SORT = 4, MSGS_K_RANGES = 5, MSGS_K_BLEND_FWD = 6, MSGS_K_BLEND_BWD = 7,
    MSGS_K_PREPROCESS_BWD = 8,
    MSGS_K_SLAB_B = 9,        /* slab mode only: the whole second pass (count, scan, emit, tile sort, ranges, blend of the open tiles) */
    MSGS_K_COUNT = 10
};
typedef struct msgs_timing {
    void* ev[2 * MSGS_K_COUNT];   /* hipEvent_t handles created by the caller (msgs_timing_create) */
} msgs_timing_t;

int msgs_abi_version(void);
const char* msgs_error_string(int code);

/* ---- size queries (bytes) -------------------------------------------------------------------- */
/* per-Gaussian state written by stage 1, read by stage 2 and by backward */
size_t msgs_geom_bytes(int32_t P);
/* scratch of stage 1 (depth sort double buffers, scan partials); dead after stage 1 returns */
size_t msgs_stage1_scratch_bytes(int32_t P);
/* per-instance state written by stage 2 (sorted Gaussian ids + tile ranges), read by backward */
size_t msgs_binning_bytes(int64_t num_instances, int32_t width, int32_t height);
/* the same for a call with msgs_view_t.slab_fraction = fraction > 0: room for slab A's instances in front of the open tiles'
 * complete lists (<= (1 + fraction) D + one instance per tile) */
size_t msgs_binning_bytes_slab(int64_t num_instances, int32_t width, int32_t height, float slab_fraction);
size_t msgs_stage2_scratch_bytes_slab(int64_t num_instances, int32_t width, int32_t height);
/* scratch of stage 2 (tile sort double buffers, histograms); dead after stage 2 returns */
size_t msgs_stage2_scratch_bytes(int64_t num_instances, int32_t width, int32_t height);
/* per-pixel state written by stage 2 (final transmittance, last contributor), read by backward */
size_t msgs_image_bytes(int32_t width, int32_t height);
/* scratch of backward (per-Gaussian 2-D gradient records) */
size_t msgs_backward_scratch_bytes(int32_t P);

/* ---- forward --------------------------------------------------------------------------------- */
/* Stage 1: preprocess (projection, EWA splat, SH->RGB, pixel size, multi-scale filters, exact
 * tile-overlap count), depth sort of the Gaussians, exclusive scan of the overlap counts.
 * Writes radii[P] (int32, 0 = not rendered) and pixel_sizes[P]; returns the number of tile
 * instances through *num_instances_host after ONE stream synchronisation. */
int msgs_forward_stage1(const msgs_view_t* view, const msgs_gaussians_t* g,
                        int32_t* radii, float* pixel_sizes,
                        void* geom, size_t geom_bytes,
                        void* scratch, size_t scratch_bytes,
                        int64_t* num_instances_host,
                        const msgs_timing_t* timing, void* stream);

/* Stage 2: emit (tile, Gaussian) instances in depth order, stable radix sort by tile, tile ranges,
 * per-tile front-to-back alpha blend.  Writes out_color[3,H,W], out_acc_pixel_size[H,W],
 * out_depth[H,W]. */
int msgs_forward_stage2(const msgs_view_t* view, const msgs_gaussians_t* g,
                        const void* geom, size_t geom_bytes,
                        int64_t num_instances,
                        void* binning, size_t binning_bytes,
                        void* scratch, size_t scratch_bytes,
                        void* image_state, size_t image_bytes,
                        float* out_color, float* out_acc_pixel_size, float* out_depth,
                        void* grad_records, size_t grad_records_bytes, int32_t backward_follows,
                        const msgs_timing_t* timing, void* stream);
/* backward_follows: non-zero when msgs_backward will be called on this forward's state — the forward then also leaves the
 * backward's heaviest-first tile launch order behind (one small kernel).  Independent of grad_records.
 * grad_records (optional, NULL = none): the buffer the caller will pass to msgs_backward as `scratch`
 * (>= msgs_backward_scratch_bytes(P)).  The per-Gaussian gradient records in it have to start from zero; the blend
 * kernel of the forward clears them on the side (it is instruction-bound, the stores are free there), and a
 * msgs_backward called with msgs_grads_t.scratch_is_clear = 1 saves the fill launch. */

/* ---- backward -------------------------------------------------------------------------------- */
/* dL_dcolor is [3,H,W].  acc_pixel_size / depth / pixel_sizes carry no gradient (they never
 * enter the reference's loss, /root/reference/train.py:205-216). */
int msgs_backward(const msgs_view_t* view, const msgs_gaussians_t* g,
                  const int32_t* radii,
                  const void* geom, size_t geom_bytes,
                  int64_t num_instances,
                  const void* binning, size_t binning_bytes,
                  const void* image_state, size_t image_bytes,
                  const float* dL_dcolor,
                  void* scratch, size_t scratch_bytes,
                  const msgs_grads_t* grads,
                  const msgs_timing_t* timing, void* stream);

/* msgs_backward_per_gaussian: the per-Gaussian half of msgs_backward ALONE (2-D covariance backward, projection, SH,
 * scale / quaternion chain — upstream's computeCov2DCUDA + preprocessCUDA backward, SURVEY 2.2 K8 + K9) on per-Gaussian
 * 2-D gradients supplied by the caller instead of the blend backward's sums: sums2d [P,9] DOUBLES (device) =
 * {dL/dmean2D x, y (the NDC-ish units of dL_dmeans2D), dL/dconic A, B, C, dL/d(effective opacity), dL/drgb[3]}, rounded to
 * float once like the library's own records.  `geom` and `radii` are those of the forward of the same view.  For the
 * parity tests: fed the oracle's sums, the two per-Gaussian stages are compared on bit-identical inputs
 * (tests/test_k8_isolation_gpu.py); also usable by a caller that blends elsewhere.  Reference-API inputs only
 * (raw_params = 0); grads->accumulate / wait_before_accumulate / accumulated must be zero / NULL (MSGS_ERR_INVALID_ARG
 * otherwise: accumulation across views is msgs_backward's contract). */
int msgs_backward_per_gaussian(const msgs_view_t* view, const msgs_gaussians_t* g, const int32_t* radii,
                               const void* geom, size_t geom_bytes, const double* sums2d, const msgs_grads_t* grads,
                               void* stream);

/* ---- view-parallel gradient exchange (new relative to the single-GPU reference) -------------------
 * The SH part of one view's gradient (48 of the 59 floats per Gaussian) is the outer product of the 16 SH basis values
 * of the viewing direction and the 3 floats dL/drgb: msgs_backward delivers just dL/drgb (msgs_grads_t.dL_dcolors in
 * the raw modes), the ranks all-gather those [P,3] factors and every rank rebuilds
 *     dL/dfeatures_{dc,rest}[i] = scale * sum_v basis(normalize(means3D[i] - campos[v])) x drgb[v][i]
 * with the products msgs_backward itself would have formed (bit-identical per view), added in view order.
 * View v's camera centre is campos + v * campos_stride (3 floats), its factors drgb + v * drgb_stride ([P,3] floats,
 * zeros where the view did not render the Gaussian) — strides in floats, so the rows of one all-gathered buffer
 * {factors | camera centre} can be read in place.  Outputs are fully written. */
int msgs_sh_grad_from_views(int32_t P, int32_t n_views, int32_t sh_degree, const float* means3D, const float* campos,
                            int64_t campos_stride, const float* drgb, int64_t drgb_stride, float scale,
                            float* dL_dfeatures_dc, float* dL_dfeatures_rest, void* stream);

/* ---- optional -------------------------------------------------------------------------------- */
/* visibility mask only (upstream markVisible; present[P] uint8) */
int msgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present, void* stream);

/* Statistics of the last-built binning (for the algorithmic-bytes formula of DESIGN.md):
 * out_host[0] = sum over tiles of max-over-pixels last contributor (D_trav),
 * out_host[1] = number of Gaussians with radii > 0 (V).  Synchronises `stream`. */
int msgs_binning_stats(const msgs_view_t* view, int32_t P, const int32_t* radii,
                       const void* binning, size_t binning_bytes,
                       const void* image_state, size_t image_bytes,
                       void* scratch, size_t scratch_bytes,
                       int64_t* out_host, void* stream);

/* Diagnostic: lane efficiency of the blend kernels.  Replays the quadrant-list blend forward and — when
 * image_state is given — the one-wave-per-tile backward on the state a forward left behind (nothing is written to geom / binning /
 * image or to any output) with scalar counters:
 *   out_host[0] = forward (wave, entry) evaluations — each evaluates 64 lanes; [1] = lanes still blending summed over them;
 *   [2] = lanes that blended (alpha >= 1/255, power <= 0, before termination);
 *   [3] = backward (tile, entry) visits; [4] = backward (quadrant, entry) evaluations — each 64 lanes; [5] = lanes that
 *   contributed a gradient; [6] = visits with at least one contributing lane (= reductions + atomic instructions); -1 each
 *   without image_state.  out_host holds 7 values; scratch >= 64 bytes.  Synchronises `stream`. */
int msgs_blend_lane_stats(const msgs_view_t* view, const void* geom, size_t geom_bytes, int32_t P, int64_t D,
                          const void* binning, size_t binning_bytes, const void* image_state, size_t image_bytes,
                          void* scratch, size_t scratch_bytes, int64_t* out_host, void* stream);

/* Deterministic (verification) mode (process-wide switch; initial value from the environment variable
 * MSGS_DETERMINISTIC=1).  The forward is always bitwise reproducible.  The default backward adds the float32 sum each
 * tile delivers for a Gaussian into DOUBLE accumulators with float64 atomics: the sums are exact, so the result does not
 * depend on the order the atomics arrive in (reproducible in practice: a double total can move by 1e-16, which changes the
 * rounded float about once in 1e9 values).  With the switch on, msgs_backward accumulates the per-pixel float32 factors in
 * double inside the tile as well, stores the nine sums of every tile entry, groups them by Gaussian with a stable sort and
 * adds them in ascending tile order: reproducible by construction and the accumulation structure of the CPU oracle; about
 * 1.4 ms slower at C3, and the scratch buffer must hold msgs_backward_scratch_bytes_deterministic(P, D) bytes.
 * msgs_set_deterministic returns the previous value. */
int msgs_set_deterministic(int32_t on);
int msgs_get_deterministic(void);
size_t msgs_backward_scratch_bytes_deterministic(int32_t P, int64_t D);

/* Which of the two blend-backward kernels msgs_backward launches: 0 (default) = by tile count (one wave per tile from
 * 4096 tiles up, four waves per tile below), 1 | 2 = force one — for the parity tests and A/B measurements.  Initial
 * value from MSGS_BWD_GEN.  Returns the previous value. */
int msgs_set_backward_generation(int32_t gen);

/* Pixel granularity of the blend kernels' workgroups: 0 (default) = by tile count, 1 = always one wave64 per 8x8 pixel
 * quadrant (forward) / per quadrant or tile (backward, see above), 2 = always the fine-grained kernels — sixteen waves
 * per 16x16 tile, one per 4x4 pixel sub-block — which the default picks below 300 tiles (the low levels of the
 * resolution pyramid MS-GS trains on, utils/camera_utils.py:38-39), where the blend kernels are latency-bound and a
 * shorter per-wave entry list matters more than idle lanes.  Same per-pixel arithmetic in the same order: forward
 * results are bit-identical across granularities.  2 overrides msgs_set_backward_generation.  Initial value from
 * MSGS_BLEND_GRANULARITY.  Returns the previous value. */
int msgs_set_blend_granularity(int32_t mode);

/* Exact per-tile occlusion cut-off (round 5; process-wide switch, on by default, initial value off with MSGS_NO_OCCLUSION=1).
 * Between the per-Gaussian stage and the depth sort the forward finds, per block of tiles, the view depth behind which every
 * pixel of the block has provably met the reference's termination rule T (1 - alpha) < 1e-4 — from the Gaussians whose
 * alpha >= 1/255 level set contains the whole block, their smallest alpha over it, and the product of (1 - alpha_min) front to
 * back with a factor 2 of slack — and neither counts, emits nor sorts the tile instances behind it.  The lists the pixels
 * walk are unchanged entry for entry: every output and gradient is bit-identical to the uncut path; only the instance count D
 * shrinks (the multi-scale model rendered without its filters, /root/reference/render.py:32: 427 M -> a few million at
 * 1080p).  One launch (a persistent grid, four phases behind grid barriers); a view without cover candidates leaves it after
 * the first barrier (~5 us), so the pass runs on EVERY forward while the switch is on.  msgs_set_occlusion returns the
 * previous value.
 * msgs_occlusion_stats reads what the pass did for the forward that last wrote `geom`: out_host[8] = {ran (0 / 1; 2 = a barrier
 * wait of the pass expired: the cut is valid but weaker — never observed), Gaussians
 * with more than 96 tile instances, cover candidates kept from them (the nearest by depth, at most 32 768), cover blocks that received a cut-off,
 * cover blocks of the view, tiles per side of a cover block, smallest and largest cut-off (float32 bits of a view depth;
 * 0xFFFFFFFF = a block stayed open)}.  (The instances removed = the instance count of the same view with the pass switched
 * off minus the one with it on.)  Synchronises `stream`; not re-entrant (a diagnostic). */
int msgs_set_occlusion(int32_t on);
/* What the last forward that RETURNED ITS INSTANCE COUNT on the calling host thread (msgs_forward, msgs_forward_stage1,
 * msgs_forward_finish) learned besides the count — the words travel with it, no extra device access: out_host[0] = cover
 * candidates of its occlusion pass (0 when the pass did not run), out_host[1] = 1 when the pass closed at least one block.
 * out_host[2..7] = the most recent FEEDBACK publication that has landed in the status block of that forward (the thread's, or
 * the msgs_status_t handle's) — normally that of an EARLIER forward, its stage 2 having finished since:
 *   [2] its msgs_view_t.feedback_tag (0: none yet), [3] its instance count D, [4] its D_trav, [5] tiles left open by slab A
 *   (-1: it ran single-pass), [6] instances of slab A, [7] instances slab B emitted.  out_host holds 8 values. */
int msgs_forward_info(int64_t* out_host);
int msgs_occlusion_stats(const void* geom, size_t geom_bytes, int32_t P, int64_t* out_host, void* stream);
/* Diagnostic: what depth-slab binning did for the forward that last wrote `geom`: out_host[6] = {ran in slab mode (0 / 1), depth
 * ranks in slab A, instances of slab A, tiles slab A left open, instances slab B emitted, overflow flag (always 0)}.
 * Synchronises `stream`. */
int msgs_slab_stats(const void* geom, size_t geom_bytes, int32_t P, int64_t* out_host, void* stream);

/* msgs_forward: both stages in ONE call.  The caller passes `binning` and `scratch2` sized for a GUESS of the instance count
 * (the previous frame's D plus a margin).  Stage 1 is launched, stage 2 is launched right behind it on those buffers — sized for
 * their capacity, the kernels read min(D, capacity) from a device word the scan writes — and only then the host waits for D (one
 * poll of pinned host words): the GPU never idles while the host learns D.  D <= capacity (the normal case): *stage2_done = 1.
 * Otherwise (first frame of a shape, or the scene outgrew the margin) *stage2_done = 0 and *num_instances_host = D: grow the
 * buffers and call msgs_forward_stage2, which overwrites the truncated result (same stream).  The binning buffer's internal
 * layout depends neither on D nor on the capacity (tile ranges first).  view->debug, the look-back sort variants and
 * MSGS_NO_SPECULATIVE_STAGE2=1 take the sequential route (wait for D, then launch stage 2 if the buffers suffice). */
int msgs_forward(const msgs_view_t* view, const msgs_gaussians_t* gaussians, int32_t* radii, float* pixel_sizes,
                 void* geom, size_t geom_bytes, void* scratch1, size_t scratch1_bytes, void* binning,
                 size_t binning_bytes, void* scratch2, size_t scratch2_bytes, void* image, size_t image_bytes,
                 float* out_color, float* out_acc_pixel_size, float* out_depth, void* grad_records,
                 size_t grad_records_bytes, int32_t backward_follows, int64_t* num_instances_host, int32_t* stage2_done,
                 const msgs_timing_t* timing, void* stream);

/* ---- forward without the host wait: several views in flight from one host thread -------------------------------------
 * msgs_forward splits into msgs_forward_launch (everything msgs_forward enqueues: stage 1 and — on capacity-sized buffers
 * — stage 2, nothing waited for) and msgs_forward_finish (the wait for the instance count).  Between the two the caller
 * may launch other forwards (each on its own msgs_status_t and its own buffers, on the same or on other streams) and
 * backwards of finished views: the reference's multi-view loops (/root/reference/train.py:282-299,337-341 insertion sweeps,
 * :488-496 evaluation, render.py:37-49, render_traj.py:99-105) and the views of one optimizer step are independent, so two
 * views on two streams let the instruction-bound blend kernels of one run beside the HBM- / latency-bound kernels of the
 * other.  A handle owns one 64-byte pinned status block; create it once and reuse it launch after launch.
 * msgs_forward_finish: *stage2_done = 1 when stage 2 ran on the launch's buffers (D <= their capacity); 0 when the caller
 * has to call msgs_forward_stage2 on buffers sized for *num_instances_host (no guess yet, debug mode, or the scene outgrew
 * the guess) on the same stream — exactly msgs_forward's contract.  Every buffer passed to the launch stays owned by the
 * caller and must outlive the work enqueued on `stream`. */
typedef struct msgs_status msgs_status_t;
int msgs_status_create(msgs_status_t** out);
int msgs_status_destroy(msgs_status_t* status);
int msgs_forward_launch(const msgs_view_t* view, const msgs_gaussians_t* gaussians, int32_t* radii, float* pixel_sizes,
                        void* geom, size_t geom_bytes, void* scratch1, size_t scratch1_bytes, void* binning,
                        size_t binning_bytes, void* scratch2, size_t scratch2_bytes, void* image, size_t image_bytes,
                        float* out_color, float* out_acc_pixel_size, float* out_depth, void* grad_records,
                        size_t grad_records_bytes, int32_t backward_follows, msgs_status_t* status,
                        const msgs_timing_t* timing, void* stream);
int msgs_forward_finish(msgs_status_t* status, int64_t* num_instances_host, int32_t* stage2_done);

/* msgs_preprocess_only: the per-Gaussian stage alone (frustum cull, multi-scale filters, projection) — radii and
 * pixel_sizes exactly as msgs_forward_stage1 writes them, without sorting, binning or blending.  For the camera
 * sweeps of the reference that render a view only to read visibility_filter / pixel_sizes
 * (/root/reference/train.py:283-300 before insert_large_gaussians, :334-338 after it).  No host synchronisation. */
int msgs_preprocess_only(const msgs_view_t* view, const msgs_gaussians_t* gaussians, int32_t* radii,
                         float* pixel_sizes, void* geom, size_t geom_bytes, void* stream);

/* ---- voxel-average pooling (large-Gaussian insertion) ------------------------------------------------------
 * GPU replacement for the eleven CPU calls of open3d.ml.torch.layers.VoxelPooling(position_fn='center',
 * feature_fn='average') in /root/reference/scene/gaussian_model.py:802-816 (third-party, un-vendored; restated
 * from its published behaviour: voxel = floor(p / voxel_size), features = mean over the voxel's points).
 * Build the grouping once per position set, then average any number of [M,F] feature tensors with it.
 *   order[M]          point ids grouped by voxel (voxels ascending in (z,y,x) voxel index, points ascending by id)
 *   seg_start[M+1]    first num_voxels+1 entries valid: voxel v owns order[seg_start[v] .. seg_start[v+1])
 *   voxel_index[M,3]  (nullable) integer voxel coordinates, first num_voxels rows valid
 * msgs_voxel_pool_build synchronises the stream once to return num_voxels. */
size_t msgs_voxel_pool_scratch_bytes(int64_t M);
int msgs_voxel_pool_build(const float* positions, int64_t M, float voxel_size, uint32_t* order, uint32_t* seg_start,
                          int32_t* voxel_index, void* scratch, size_t scratch_bytes, int64_t* num_voxels_host,
                          void* stream);
int msgs_voxel_pool_average(const float* features, int32_t F, const uint32_t* order, const uint32_t* seg_start,
                            int64_t num_voxels, float* out, void* stream);

/* ---- train-step epilogue ------------------------------------------------------------------
 * msgs_adam_step replaces torch.optim.Adam(l, lr=0.0, eps=1e-15).step() over the model's leaf tensors
 * (/root/reference/scene/gaussian_model.py:235-248, /root/reference/train.py:416-418) with ONE launch; the state
 * tensors are the optimizer's own state["exp_avg"], state["exp_avg_sq"] (the reference's densification code slices
 * and concatenates them: gaussian_model.py:419-476), `step` is the 1-based step count after increment, `lr` the
 * group's current learning rate.  Semantics: torch Adam with amsgrad=False, weight_decay=0, maximize=False. */
#define MSGS_ADAM_MAX_TENSORS 8
typedef struct msgs_adam_tensor {
    float* param;            /* [n] updated in place */
    const float* grad;       /* [n] */
    float* exp_avg;          /* [n] updated in place */
    float* exp_avg_sq;       /* [n] updated in place */
    int64_t n;               /* number of floats */
    double lr;               /* double like the Python float in param_group["lr"]: scalars are rounded to float once */
} msgs_adam_tensor_t;
int msgs_adam_step(const msgs_adam_tensor_t* tensors, int32_t n_tensors, int64_t step, double beta1, double beta2,
                   double eps, void* stream);

/* msgs_densify_stats: every per-Gaussian statistic the reference updates between backward() and optimizer.step(),
 * in one launch, all masked by visibility_filter = radii > 0 (/root/reference/train.py:239-250):
 *   MSGS_STATS_BASE_MASK    base_mask |= visible                         (gaussian_model.py:702-704)
 *   MSGS_STATS_PIXEL_SIZES  update_pixel_sizes(visible, pixel_sizes, reso_lvl)   (gaussian_model.py:663-687)
 *   MSGS_STATS_DENSIFY      max_radii2D = max(max_radii2D, radii); xyz_gradient_accum[:, reso_lvl] += |grad_xy|;
 *                           denom[:, reso_lvl] += 1                      (train.py:247-250, gaussian_model.py:698-701)
 * Pointers of a group that is not selected may be NULL. */
#define MSGS_STATS_BASE_MASK 1
#define MSGS_STATS_PIXEL_SIZES 2
#define MSGS_STATS_DENSIFY 4
typedef struct msgs_densify_stats {
    int32_t P;
    int32_t flags;
    int32_t reso_lvl;                /* resolution level trained this iteration (train.py reso_idx) */
    int32_t reso_lvls;               /* number of levels (second dim of xyz_gradient_accum / denom) */
    const int32_t* radii;            /* [P] output of the forward */
    const float* pixel_sizes;        /* [P] output of the forward */
    const float* means2D_grad;       /* [P,3] viewspace_points.grad */
    const int64_t* target_reso_lvl;  /* [P] torch.long, gaussian_model.py:227 */
    float* xyz_gradient_accum;       /* [P, reso_lvls, 1] */
    float* denom;                    /* [P, reso_lvls, 1] */
    float* max_radii2D;              /* [P] */
    float* max_pixel_sizes;          /* [P] */
    float* min_pixel_sizes;          /* [P] */
    uint8_t* base_mask;              /* [P] torch.bool */
} msgs_densify_stats_t;
int msgs_densify_stats(const msgs_densify_stats_t* stats, void* stream);

/* ---- fused photometric loss ----------------------------------------------------------------
 * loss = (1 - lambda_dssim) * l1_loss(img, gt) + lambda_dssim * (1 - ssim(img, gt))   (/root/reference/train.py:209-211)
 * with l1_loss / ssim of /root/reference/utils/loss_utils.py:17-18,32-63 (window 11, sigma 1.5, zero padding, mean).
 * img, gt: [C,H,W] float32 device tensors.  msgs_loss_forward writes out3 = {loss, l1 mean, ssim mean} (device) and,
 * when keep_for_backward != 0, the derivative maps into scratch; msgs_loss_backward then writes
 * dL_dimg[C,H,W] = upstream * dloss/dimg (upstream: device scalar, NULL = 1) — the layout msgs_backward consumes.
 * The loss value is summed in a fixed order (run-to-run identical).  msgs_ssim_window returns the 11 window taps. */
size_t msgs_loss_scratch_bytes(int32_t C, int32_t H, int32_t W);
int msgs_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      float* out3, void* scratch, size_t scratch_bytes, int32_t keep_for_backward, void* stream);
int msgs_loss_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                       const float* upstream, const void* scratch, size_t scratch_bytes, float* dL_dimg, void* stream);
int msgs_ssim_window(float* taps11_host);

/* ---- distCUDA2 --------------------------------------------------------------------------------
 * mean_dist2[i] = mean of the squared distances from points[i] to its 3 nearest OTHER points (exact, float32) —
 * `distCUDA2(points)` of the un-vendored simple-knn submodule, used once at initialisation
 * (/root/reference/scene/gaussian_model.py:26,199).  points: [P,3] device, P >= 4. */
size_t msgs_knn_scratch_bytes(int64_t P);
int msgs_dist2_knn3(const float* points, int64_t P, float* mean_dist2, void* scratch, size_t scratch_bytes,
                    void* stream);

/* timing helpers: create/destroy the 2*MSGS_K_COUNT events and read elapsed ms per kernel class
 * (ms_host[MSGS_K_COUNT]; a class that was not recorded reads as -1).  The caller synchronises
 * the stream before msgs_timing_read. */
int msgs_timing_create(msgs_timing_t* t);
int msgs_timing_destroy(msgs_timing_t* t);
int msgs_timing_read(const msgs_timing_t* t, float* ms_host);

#ifdef __cplusplus
}
#endif
#endif /* MSGS_H_ */
