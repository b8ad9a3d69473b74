// api.hip — the extern "C" entry points of libmsgs_hip.so declared in include/msgs.h.
// Host-side orchestration only: argument checks, workspace carving, kernel launches on the caller's
// stream.  No allocation, no global mutable state, one stream synchronisation (stage 1's instance
// count) outside debug mode.
#include "msgs_internal.h"

#include <atomic>
#include <new>

#include <cstring>

using namespace msgs;

namespace {

struct Timer {
    const msgs_timing_t* t;
    hipStream_t s;
    void begin(int k) const { if (t && t->ev[2 * k]) (void)hipEventRecord((hipEvent_t)t->ev[2 * k], s); }
    void end(int k) const { if (t && t->ev[2 * k + 1]) (void)hipEventRecord((hipEvent_t)t->ev[2 * k + 1], s); }
};

inline int check_inputs(const msgs_view_t* v, const msgs_gaussians_t* g) {
    if (!v || !g) return MSGS_ERR_INVALID_ARG;
    if (g->P < 0 || v->image_width <= 0 || v->image_height <= 0) return MSGS_ERR_INVALID_ARG;
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos) return MSGS_ERR_INVALID_ARG;
    if (g->P > 0 && (!g->means3D || !g->opacities)) return MSGS_ERR_INVALID_ARG;
    if (g->raw_params) {      // 1: raw GaussianModel parameters; 2: activated inputs + chained gradients (msgs.h)
        if (g->raw_params != 1 && g->raw_params != 2) return MSGS_ERR_INVALID_ARG;
        if (!g->features_dc || !g->features_rest || !g->scales || !g->rotations) return MSGS_ERR_INVALID_ARG;
        if (g->raw_params == 2 && !g->rotations_raw) return MSGS_ERR_INVALID_ARG;
        if ((g->shs && g->raw_params != 2) || g->colors_precomp || g->cov3D_precomp) return MSGS_ERR_INVALID_ARG;
        if (v->sh_degree < 0 || v->sh_degree > 3 || v->sh_coeffs != 16) return MSGS_ERR_SH_DEGREE;
        if ((v->image_width + TILE - 1) / TILE > 65535 || (v->image_height + TILE - 1) / TILE > 65535)
            return MSGS_ERR_INVALID_ARG;
        return MSGS_OK;
    }
    // exactly one of shs / colors_precomp, exactly one of (scales, rotations) / cov3D_precomp
    // (upstream raises on both-or-neither)
    if ((g->shs != nullptr) == (g->colors_precomp != nullptr)) return MSGS_ERR_INVALID_ARG;
    const bool sr = g->scales != nullptr && g->rotations != nullptr;
    if ((g->scales != nullptr) != (g->rotations != nullptr)) return MSGS_ERR_INVALID_ARG;
    if (sr == (g->cov3D_precomp != nullptr)) return MSGS_ERR_INVALID_ARG;
    if (g->shs) {
        if (v->sh_degree < 0 || v->sh_degree > 3) return MSGS_ERR_SH_DEGREE;
        if ((v->sh_degree + 1) * (v->sh_degree + 1) > v->sh_coeffs) return MSGS_ERR_SH_DEGREE;
    }
    if ((v->image_width + TILE - 1) / TILE > 65535 || (v->image_height + TILE - 1) / TILE > 65535)
        return MSGS_ERR_INVALID_ARG;
    return MSGS_OK;
}

inline int debug_sync(const msgs_view_t* v, hipStream_t s) {
    if (!v->debug) return MSGS_OK;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    e = hipGetLastError();
    return e == hipSuccess ? MSGS_OK : (int)e;
}

inline int tile_bits(int num_tiles) {   // bits needed for tile ids 0..num_tiles (sentinel included)
    int b = 1;
    while ((1ll << b) <= (long long)num_tiles) ++b;
    return b;
}

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

}  // namespace

extern "C" {

int msgs_abi_version(void) { return MSGS_ABI_VERSION; }

const char* msgs_error_string(int code) {
    switch (code) {
        case MSGS_OK: return "ok";
        case MSGS_ERR_INVALID_ARG: return "invalid argument (NULL or inconsistent pointers: provide exactly one of shs/colors_precomp and exactly one of scales+rotations/cov3D_precomp)";
        case MSGS_ERR_CAPACITY: return "a caller-supplied buffer is smaller than its size query";
        case MSGS_ERR_TOO_MANY: return "more than 2^32-1 tile instances";
        case MSGS_ERR_SH_DEGREE: return "sh_degree must be 0..3 and (sh_degree+1)^2 <= sh_coeffs";
        case MSGS_ERR_INTERNAL: return "internal error: a look-back wait in the sort/scan kernels timed out";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

size_t msgs_geom_bytes(int32_t P) { return GeomLayout(P > 0 ? P : 1).total; }
size_t msgs_stage1_scratch_bytes(int32_t P) { return Stage1Scratch(P > 0 ? P : 1).total; }
size_t msgs_binning_bytes(int64_t D, int32_t W, int32_t H) {
    return BinningLayout(D, (int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE)).total;
}
size_t msgs_stage2_scratch_bytes(int64_t D, int32_t W, int32_t H) {
    (void)W; (void)H;
    const Stage2Scratch L(D);
    return L.total + align256(4 * (size_t)(D > 0 ? D : 1));   // + sorted key buffer
}
static int64_t tiles_of(int32_t W, int32_t H) { return (int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE); }
size_t msgs_binning_bytes_slab(int64_t D, int32_t W, int32_t H, float fraction) {
    if (!(fraction > 0.f)) return msgs_binning_bytes(D, W, H);
    return msgs_binning_bytes(SlabGeom::ids_needed(D > 0 ? D : 1, (double)fraction, tiles_of(W, H)), W, H);
}
size_t msgs_stage2_scratch_bytes_slab(int64_t D, int32_t W, int32_t H) {
    return msgs_stage2_scratch_bytes((D > 0 ? D : 1) + SLAB_SLACK + 256, W, H);
}
size_t msgs_image_bytes(int32_t W, int32_t H) { return ImageLayout(W, H).total; }
size_t msgs_backward_scratch_bytes(int32_t P) {
    return align256(GRAD_REC_BYTES * (size_t)(P > 0 ? P : 1));
}
size_t msgs_backward_scratch_bytes_deterministic(int32_t P, int64_t D) { return DetScratch(P, D).total; }

static std::atomic<int> g_deterministic{[] { const char* e = getenv("MSGS_DETERMINISTIC"); return (e && e[0] == '1') ? 1 : 0; }()};
int msgs_set_deterministic(int32_t on) { return g_deterministic.exchange(on ? 1 : 0); }
int msgs_get_deterministic(void) { return g_deterministic.load(); }
int msgs_set_backward_generation(int32_t gen) { return set_backward_generation(gen); }
int msgs_set_blend_granularity(int32_t mode) { return set_blend_granularity(mode); }
int msgs_set_occlusion(int32_t on) { return set_occlusion(on); }

}  // extern "C"

namespace {
// Speculative stage 2 (msgs_forward): everything stage 2 needs, handed to stage 1 so that it can launch stage 2 on the caller's
// capacity-sized buffers right behind its own last kernel — BEFORE the host waits for the instance count.  The stage-2 kernels then
// read min(D, capacity) from a device word the scan writes; the host checks D <= capacity afterwards (and the caller redoes
// stage 2 on exact buffers when the scene outgrew the guess).  Without it the GPU idled ~14 us per step between the scan and
// the emit while the host learned D and launched (profiles/r3_idle.txt).
struct SpecStage2 {
    int64_t capacity;
    void* binning; size_t binning_bytes;
    void* scratch2; size_t scratch2_bytes;
    void* image; size_t image_bytes;
    float* out_color; float* out_acc_ps; float* out_depth;
    void* grad_records; size_t grad_records_bytes;
    int backward_follows;
    int launched_rc;           // out: status of the speculative launch
};
int forward_stage2_impl(const msgs_view_t* view, const msgs_gaussians_t* g, const void* geom_v, size_t geom_bytes, int64_t D,
                        void* binning_v, size_t binning_bytes, void* scratch_v, size_t scratch_bytes, void* image_v,
                        size_t image_bytes, float* out_color, float* out_acc_ps, float* out_depth, void* grad_records,
                        size_t grad_records_bytes, int backward_follows, const msgs_timing_t* timing, void* stream,
                        const uint32_t* D_dev, uint64_t* fb_dev, uint64_t fb_ticket);

// The instance count D comes back through three pinned, device-mapped host words {D, flags, ticket} that a kernel writes
// itself and the host polls: no copy command, no interrupt-driven wait (a blocking hipStreamSynchronize wakes up tens of
// microseconds after the data landed, and until stage 2 is launched the GPU idles).  Block 0 of the scan's second kernel —
// which sums every block total — writes them first thing, so the host learns D while the last stage-1 kernel still runs.  MSGS_BLOCKING_SYNC=1, or a failed pinned allocation, falls back to a 16-byte copy +
// hipStreamSynchronize.  msgs_forward / msgs_forward_stage1 use one block per calling host thread (they wait before they
// return); msgs_forward_launch takes the block of the caller's msgs_status_t, so that any number of forwards can be in
// flight from one thread, one per handle.
struct StatusBlock {
    uint64_t* host = nullptr;
    uint64_t* dev = nullptr;
    uint64_t ticket = 0;
    bool tried = false;
    void ensure() {
        if (tried) return;
        tried = true;
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                host = (uint64_t*)h;
                dev = (uint64_t*)d;
                for (int k = 0; k < 8; ++k) host[k] = 0;
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    void release() {
        if (host) (void)hipHostFree(host);
        host = dev = nullptr;
        tried = false;
    }
};

// a stage 1 whose instance count has not been read back yet
struct PendingCount {
    bool active = false;
    bool polled = false;
    hipStream_t stream = nullptr;
    uint64_t ticket = 0;
    volatile uint64_t* host = nullptr;
    const uint64_t* status_dev = nullptr;   // device copy of {D, flags, info}: the fallback when the flag never lands
};

// {cover candidates, any block closed} of the last forward whose count this host thread collected (msgs_forward_info)
thread_local uint64_t t_last_info = 0;
// words 4..7 of the status block as they stood when that count was collected: the feedback publication of an earlier forward
thread_local uint64_t t_last_feedback[4] = {0, 0, 0, 0};

static bool blocking_sync() {
    static const bool blocking = [] { const char* e = getenv("MSGS_BLOCKING_SYNC"); return e && e[0] == '1'; }();
    return blocking;
}

// Depth-slab binning for this call?  (msgs_view_t.slab_fraction; include/msgs.h)  Independent of the buffers, which
// SlabGeom checks.
static bool slab_wanted(const msgs_view_t* view, const msgs_gaussians_t* g) {
    const float f = view->slab_fraction;
    if (!(f > 0.f) || !(f <= 0.5f) || g->P <= 0 || g_deterministic.load() != 0) return false;
    const int64_t tiles = (int64_t)((view->image_width + TILE - 1) / TILE) * ((view->image_height + TILE - 1) / TILE);
    return tiles >= SLAB_MIN_TILES && tiles < 65535 * 16 && forward_uses_quadrant_kernel((int)tiles);
}

// Launches K1, the depth sort and the scan — and, with `spec`, stage 2 right behind them — WITHOUT waiting for D.
int forward_stage1_launch(const msgs_view_t* view, const msgs_gaussians_t* g, int32_t* radii, float* pixel_sizes,
                          void* geom_v, size_t geom_bytes, void* scratch_v, size_t scratch_bytes,
                          const msgs_timing_t* timing, void* stream, SpecStage2* spec, StatusBlock& sb,
                          PendingCount& pend) {
    pend.active = false;
    int rc = check_inputs(view, g);
    if (rc) return rc;
    const int P = g->P;
    if (P == 0) return MSGS_OK;
    if (!radii || !pixel_sizes || !geom_v || !scratch_v) return MSGS_ERR_INVALID_ARG;
    if (geom_bytes < msgs_geom_bytes(P) || scratch_bytes < msgs_stage1_scratch_bytes(P)) return MSGS_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    char* geom = (char*)geom_v;
    char* scratch = (char*)scratch_v;
    const GeomLayout GL(P);
    const Stage1Scratch SL(P);
    const ViewParams vp = make_view_params(view);
    const Timer tm{timing, s};

    // K1 also clears the depth sort's group-sum table (saves a fill launch) and the header of the occlusion cut-off
    // (enabled = 0, no candidates: what the emit reads when the pass does not run)
    ZeroJob zj1{nullptr, 0, (uint32_t*)(geom + GL.slab_hdr), (GL.occ_hdr - GL.slab_hdr) / 4 + sizeof(OccHeader) / 4 + 2 * (size_t)OCC_BUCKETS};
    const bool sort1_prezeroed = radix_sort_zero_region(P, 0, 32, scratch + SL.sort, &zj1.p0, &zj1.n0);
    // exact per-tile occlusion cut-off (occlusion.hip): on unless switched off
    const bool occlusion = get_occlusion() != 0;
    uint32_t* heavy_list = occlusion ? (uint32_t*)(scratch + SL.heavy_list) : nullptr;
    uint32_t* heavy_count = occlusion ? (uint32_t*)(scratch + SL.heavy_count) : nullptr;
    uint32_t* heavy_blk = occlusion ? (uint32_t*)(scratch + SL.heavy_blk) : nullptr;
    tm.begin(MSGS_K_PREPROCESS);
    // (verification mode: the kernel also leaves the raw conic and the effective opacity for the literal blend loops)
    HIP_TRY(launch_preprocess(vp, *g, radii, pixel_sizes, geom, s, zj1, heavy_list, heavy_count, heavy_blk, g_deterministic.load() != 0));
    if (occlusion)      // (timed with K1: two launches that leave at once when the view has no cover candidates)
        HIP_TRY(launch_occlusion(vp, P, geom, heavy_list, heavy_count, heavy_blk, (OccCand*)(scratch + SL.occ_cand), s));
    tm.end(MSGS_K_PREPROCESS);
    if ((rc = debug_sync(view, s))) return rc;

    // depth order of the Gaussians (not rendered -> key 0xFFFFFFFF -> sorted last, zero tiles); the sorted keys stay in geom:
    // the emit compares them with the tiles' cut-off keys
    if (!blocking_sync()) sb.ensure();
    const bool polled = !blocking_sync() && sb.host != nullptr;
    const uint64_t ticket = ++sb.ticket;
    uint64_t* total_dev = (uint64_t*)(scratch + SL.total_out);
    uint64_t* status_dev = total_dev + 2;
    uint32_t* clamped_dev = (uint32_t*)(total_dev + 5);       // min(D, capacity) for a speculative stage 2
    // the speculative stage 2's queue of heavy Gaussians starts empty (binning.hip): cleared by whoever publishes the count
    uint32_t* heavy_q_word = spec ? (uint32_t*)((char*)spec->scratch2 + Stage2Scratch(spec->capacity).heavy_q) : nullptr;

    tm.begin(MSGS_K_DEPTH_SORT);
    HIP_TRY(radix_sort_pairs((uint32_t*)(geom + GL.key), nullptr, (uint32_t*)(geom + GL.skey),
                             (uint32_t*)(geom + GL.order), P, 0, 32, scratch + SL.sort, s, sort1_prezeroed,
                             (uint32_t*)(geom + GL.nvalid)));
    tm.end(MSGS_K_DEPTH_SORT);
    if ((rc = debug_sync(view, s))) return rc;
    tm.begin(MSGS_K_SCAN);
    HIP_TRY(exclusive_scan_u32((const uint32_t*)(geom + GL.tiles), (const uint32_t*)(geom + GL.order),
                               (uint32_t*)(geom + GL.offs), P, (uint64_t*)(scratch + SL.scan_partials), total_dev, s,
                               status_dev, polled ? sb.dev : nullptr, ticket,
                               (const uint32_t*)(geom + GL.nvalid), spec ? clamped_dev : nullptr,
                               spec ? (uint64_t)spec->capacity : 0, (const uint32_t*)(geom + GL.occ_hdr), heavy_q_word, nullptr,
                               // the counts sit below the Gaussians' coarse cell ranges; a view that may run in depth slabs gets the
                               // ranges in depth order (in offs_b: slab B's recount reads them and leaves its counts there)
                               vp.cell_sx >= 0 ? TILE_COUNT_MASK : 0xFFFFFFFFu,
                               vp.cell_sx >= 0 && slab_wanted(view, g) ? (uint32_t*)(geom + GL.offs_b) : nullptr,
                               &reinterpret_cast<SlabHeader*>(geom + GL.slab_hdr)->pad[0]));
    tm.end(MSGS_K_SCAN);
    pend.active = true;
    pend.polled = polled;
    pend.stream = s;
    pend.ticket = ticket;
    pend.host = sb.host;
    pend.status_dev = status_dev;
    if (spec)       // stage 2 goes out NOW, sized for the capacity; the GPU runs it while the host waits for D
        spec->launched_rc = forward_stage2_impl(view, g, geom_v, geom_bytes, spec->capacity, spec->binning, spec->binning_bytes,
                                                spec->scratch2, spec->scratch2_bytes, spec->image, spec->image_bytes,
                                                spec->out_color, spec->out_acc_ps, spec->out_depth, spec->grad_records,
                                                spec->grad_records_bytes, spec->backward_follows, timing, stream, clamped_dev,
                                                polled ? sb.dev : nullptr, ticket);
    return MSGS_OK;
}

// Waits until the count of `pend` has landed (polls the pinned words; falls back to a device copy + synchronise).
int forward_stage1_wait(PendingCount& pend, int64_t* num_instances_host) {
    *num_instances_host = 0;
    if (!pend.active) { t_last_info = 0; for (int k = 0; k < 4; ++k) t_last_feedback[k] = 0; return MSGS_OK; }      // P == 0
    pend.active = false;
    hipStream_t s = pend.stream;
    uint64_t host_status[3] = {0, 0, 0};
    if (pend.polled) {
        volatile uint64_t* hv = pend.host;
        const uint64_t ticket = pend.ticket;
        uint64_t spins = 0;
        while (hv[2] != ticket) {
            if ((++spins & 0xFFFF) == 0) {                 // every 65536 polls: has the stream failed or finished?
                const hipError_t q = hipStreamQuery(s);
                if (q != hipErrorNotReady && q != hipSuccess) return (int)q;
                if (q == hipSuccess && hv[2] != ticket) {  // finished without the flag: fall back to the device copy
                    HIP_TRY(hipMemcpyAsync(host_status, pend.status_dev, sizeof(host_status), hipMemcpyDeviceToHost, s));
                    HIP_TRY(hipStreamSynchronize(s));
                    break;
                }
            }
        }
        if (hv[2] == ticket) { host_status[0] = hv[0]; host_status[1] = hv[1]; host_status[2] = hv[3]; }
        // feedback words (written by forward_feedback_kernel of an EARLIER forward on this block, the last word last): a
        // consistent snapshot or nothing
        for (int k = 0; k < 4; ++k) t_last_feedback[k] = 0;
        for (int attempt = 0; attempt < 4; ++attempt) {
            const uint64_t w7 = hv[7], w4 = hv[4], w5 = hv[5], w6 = hv[6];
            if (hv[7] == w7) { t_last_feedback[0] = w4; t_last_feedback[1] = w5; t_last_feedback[2] = w6; t_last_feedback[3] = w7; break; }
        }
    } else {
        for (int k = 0; k < 4; ++k) t_last_feedback[k] = 0;
        HIP_TRY(hipMemcpyAsync(host_status, pend.status_dev, sizeof(host_status), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    const uint64_t total = host_status[0];
    t_last_info = host_status[2];
    if (host_status[1] != 0) return MSGS_ERR_INTERNAL;
    if (total > 0xFFFFFFFFull) return MSGS_ERR_TOO_MANY;
    *num_instances_host = (int64_t)total;
    return MSGS_OK;
}

StatusBlock& thread_status_block() {
    static thread_local StatusBlock sb;           // (never freed: 64 bytes of pinned memory per calling host thread)
    return sb;
}

int forward_stage1_impl(const msgs_view_t* view, const msgs_gaussians_t* g, int32_t* radii, float* pixel_sizes,
                        void* geom_v, size_t geom_bytes, void* scratch_v, size_t scratch_bytes,
                        int64_t* num_instances_host, const msgs_timing_t* timing, void* stream, SpecStage2* spec) {
    if (!num_instances_host) return MSGS_ERR_INVALID_ARG;
    *num_instances_host = 0;
    PendingCount pend;
    int rc = forward_stage1_launch(view, g, radii, pixel_sizes, geom_v, geom_bytes, scratch_v, scratch_bytes, timing, stream,
                                   spec, thread_status_block(), pend);
    if (rc) return rc;
    return forward_stage1_wait(pend, num_instances_host);
}

// largest D with bytes_of(D) <= bytes (bytes_of monotone); 0 if none
template <class F>
static int64_t max_instances_for(size_t bytes, F bytes_of) {
    if (bytes_of(1) > bytes) return 0;
    int64_t lo = 1, hi = 0xFFFFFFFFll;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo + 1) / 2;
        if (bytes_of(mid) <= bytes) lo = mid; else hi = mid - 1;
    }
    return lo;
}

}  // namespace

extern "C" {

int msgs_forward_stage1(const msgs_view_t* view, const msgs_gaussians_t* g, int32_t* radii, float* pixel_sizes,
                        void* geom_v, size_t geom_bytes, void* scratch_v, size_t scratch_bytes,
                        int64_t* num_instances_host, const msgs_timing_t* timing, void* stream) {
    return forward_stage1_impl(view, g, radii, pixel_sizes, geom_v, geom_bytes, scratch_v, scratch_bytes, num_instances_host,
                               timing, stream, nullptr);
}

// instance capacities of the caller's stage-2 buffers: ids of `binning`, and what the scratch serves
static void stage2_capacities(const msgs_view_t* view, size_t binning_bytes, size_t scratch2_bytes, int64_t& cb, int64_t& cs) {
    const int W0 = view->image_width, H0 = view->image_height;
    cb = max_instances_for(binning_bytes, [&](int64_t d) { return msgs_binning_bytes(d, W0, H0); });
    cs = max_instances_for(scratch2_bytes, [&](int64_t d) { return msgs_stage2_scratch_bytes(d, W0, H0); });
}

// speculative stage 2 on the caller's buffers: possible when they hold at least 4096 instances, the view is not in debug
// mode and the sort geometry can take its count from a device word (MSGS_NO_SPECULATIVE_STAGE2=1: never)
static bool prepare_spec(const msgs_view_t* view, const msgs_gaussians_t* g, void* binning, size_t binning_bytes, void* scratch2,
                         size_t scratch2_bytes, void* image, size_t image_bytes, float* out_color, float* out_acc_ps,
                         float* out_depth, void* grad_records, size_t grad_records_bytes, int backward_follows,
                         SpecStage2& spec) {
    static const bool no_spec = [] { const char* e = getenv("MSGS_NO_SPECULATIVE_STAGE2"); return e && e[0] == '1'; }();
    if (no_spec || view->debug || g->P <= 0 || !binning || !scratch2 || !image || !out_color || !out_acc_ps || !out_depth ||
        view->image_width <= 0 || view->image_height <= 0)
        return false;
    const int W0 = view->image_width, H0 = view->image_height;
    int64_t cb, cs;
    stage2_capacities(view, binning_bytes, scratch2_bytes, cb, cs);
    int64_t cap = cb < cs ? cb : cs;
    const int tiles0 = ((W0 + TILE - 1) / TILE) * ((H0 + TILE - 1) / TILE);
    if (slab_wanted(view, g)) {          // in slab mode the buffers serve fewer instances: slab A's ids sit in front of slab B's
        const SlabGeom SG(cb, cs, (double)view->slab_fraction, tiles0, (int64_t)1 << 40);
        if (SG.ok && SG.cap_d >= 4096) cap = SG.cap_d;      // (else: single pass on the whole capacity)
    }
    if (cap < 4096 || !radix_sort_supports_device_count(cap, 0, tile_bits(tiles0))) return false;
    spec.capacity = cap;
    spec.binning = binning; spec.binning_bytes = binning_bytes;
    spec.scratch2 = scratch2; spec.scratch2_bytes = scratch2_bytes;
    spec.image = image; spec.image_bytes = image_bytes;
    spec.out_color = out_color; spec.out_acc_ps = out_acc_ps; spec.out_depth = out_depth;
    spec.grad_records = grad_records; spec.grad_records_bytes = grad_records_bytes;
    spec.backward_follows = backward_follows;
    spec.launched_rc = MSGS_OK;
    return true;
}

}  // extern "C"

// the handle of msgs_forward_launch / msgs_forward_finish: its own pinned status block + the launch it is waiting for
struct msgs_status {
    StatusBlock sb;
    PendingCount pend;
    bool launched = false;       // a msgs_forward_launch has not been finished yet
    bool stage2_launched = false;
    int64_t capacity = 0;
    int launched_rc = MSGS_OK;
};

extern "C" {

int msgs_status_create(msgs_status_t** out) {
    if (!out) return MSGS_ERR_INVALID_ARG;
    *out = new (std::nothrow) msgs_status();
    if (!*out) return (int)hipErrorOutOfMemory;
    if (!blocking_sync()) (*out)->sb.ensure();
    return MSGS_OK;
}

int msgs_status_destroy(msgs_status_t* st) {
    if (!st) return MSGS_OK;
    if (st->launched && st->pend.active) (void)hipStreamSynchronize(st->pend.stream);   // a kernel may still write the block
    st->sb.release();
    delete st;
    return MSGS_OK;
}

int msgs_forward_launch(const msgs_view_t* view, const msgs_gaussians_t* g, int32_t* radii, float* pixel_sizes, void* geom,
                        size_t geom_bytes, void* scratch1, size_t scratch1_bytes, void* binning, size_t binning_bytes,
                        void* scratch2, size_t scratch2_bytes, void* image, size_t image_bytes, float* out_color,
                        float* out_acc_ps, float* out_depth, void* grad_records, size_t grad_records_bytes,
                        int32_t backward_follows, msgs_status_t* status, const msgs_timing_t* timing, void* stream) {
    if (!view || !g || !status) return MSGS_ERR_INVALID_ARG;
    if (status->launched) return MSGS_ERR_INVALID_ARG;          // one launch per handle at a time
    SpecStage2 spec{};
    const bool use_spec = prepare_spec(view, g, binning, binning_bytes, scratch2, scratch2_bytes, image, image_bytes, out_color,
                                       out_acc_ps, out_depth, grad_records, grad_records_bytes, backward_follows, spec);
    int rc = forward_stage1_launch(view, g, radii, pixel_sizes, geom, geom_bytes, scratch1, scratch1_bytes, timing, stream,
                                   use_spec ? &spec : nullptr, status->sb, status->pend);
    if (rc) return rc;
    status->launched = true;
    status->stage2_launched = use_spec;
    status->capacity = spec.capacity;
    status->launched_rc = status->stage2_launched ? spec.launched_rc : MSGS_OK;
    return MSGS_OK;
}

int msgs_forward_finish(msgs_status_t* status, int64_t* num_instances_host, int32_t* stage2_done) {
    if (!status || !num_instances_host || !stage2_done) return MSGS_ERR_INVALID_ARG;
    *stage2_done = 0;
    *num_instances_host = 0;
    if (!status->launched) return MSGS_ERR_INVALID_ARG;
    status->launched = false;
    int rc = forward_stage1_wait(status->pend, num_instances_host);
    if (rc) return rc;
    if (status->stage2_launched && status->launched_rc != MSGS_OK) return status->launched_rc;
    if (status->stage2_launched && *num_instances_host <= status->capacity) *stage2_done = 1;
    return MSGS_OK;
}

int msgs_forward(const msgs_view_t* view, const msgs_gaussians_t* g, int32_t* radii, float* pixel_sizes, void* geom,
                 size_t geom_bytes, void* scratch1, size_t scratch1_bytes, void* binning, size_t binning_bytes,
                 void* scratch2, size_t scratch2_bytes, void* image, size_t image_bytes, float* out_color,
                 float* out_acc_ps, float* out_depth, void* grad_records, size_t grad_records_bytes,
                 int32_t backward_follows, int64_t* num_instances_host, int32_t* stage2_done, const msgs_timing_t* timing,
                 void* stream) {
    if (!stage2_done) return MSGS_ERR_INVALID_ARG;
    *stage2_done = 0;
    if (!view || !g) return MSGS_ERR_INVALID_ARG;
    SpecStage2 spec{};
    bool use_spec = prepare_spec(view, g, binning, binning_bytes, scratch2, scratch2_bytes, image, image_bytes, out_color,
                                 out_acc_ps, out_depth, grad_records, grad_records_bytes, backward_follows, spec);
    int rc = forward_stage1_impl(view, g, radii, pixel_sizes, geom, geom_bytes, scratch1, scratch1_bytes, num_instances_host,
                                 timing, stream, use_spec ? &spec : nullptr);
    if (rc) return rc;
    const int64_t D = *num_instances_host;
    const int W = view->image_width, H = view->image_height;
    if (use_spec && spec.launched_rc != MSGS_OK) return spec.launched_rc;
    if (use_spec && D <= spec.capacity) {       // the normal case: stage 2 already ran (is running) on these buffers
        *stage2_done = 1;
        return MSGS_OK;
    }
    if (use_spec) return MSGS_OK;               // the scene outgrew the guess: the caller redoes stage 2 on exact buffers
    if (!binning || binning_bytes < msgs_binning_bytes(D, W, H)) return MSGS_OK;
    if (D > 0 && (!scratch2 || scratch2_bytes < msgs_stage2_scratch_bytes(D, W, H))) return MSGS_OK;
    rc = msgs_forward_stage2(view, g, geom, geom_bytes, D, binning, binning_bytes, scratch2, scratch2_bytes, image,
                             image_bytes, out_color, out_acc_ps, out_depth, grad_records, grad_records_bytes, backward_follows,
                             timing, stream);
    if (rc) return rc;
    *stage2_done = 1;
    return MSGS_OK;
}

int msgs_preprocess_only(const msgs_view_t* view, const msgs_gaussians_t* g, int32_t* radii, float* pixel_sizes,
                         void* geom_v, size_t geom_bytes, void* stream) {
    int rc = check_inputs(view, g);
    if (rc) return rc;
    const int P = g->P;
    if (P == 0) return MSGS_OK;
    if (!radii || !pixel_sizes || !geom_v) return MSGS_ERR_INVALID_ARG;
    if (geom_bytes < msgs_geom_bytes(P)) return MSGS_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_preprocess(make_view_params(view), *g, radii, pixel_sizes, (char*)geom_v, s));
    return debug_sync(view, s);
}

int msgs_forward_stage2(const msgs_view_t* view, const msgs_gaussians_t* g, const void* geom_v, size_t geom_bytes,
                        int64_t D, void* binning_v, size_t binning_bytes, void* scratch_v, size_t scratch_bytes,
                        void* image_v, size_t image_bytes, float* out_color, float* out_acc_ps, float* out_depth,
                        void* grad_records, size_t grad_records_bytes, int32_t backward_follows, const msgs_timing_t* timing,
                        void* stream) {
    uint64_t* fb_dev = nullptr;
    uint64_t fb_ticket = 0;
    if (view && view->feedback_tag != 0 && !blocking_sync()) {        // feedback goes to the calling thread's status block
        StatusBlock& sb = thread_status_block();
        sb.ensure();
        fb_dev = sb.dev;
        fb_ticket = sb.ticket;
    }
    return forward_stage2_impl(view, g, geom_v, geom_bytes, D, binning_v, binning_bytes, scratch_v, scratch_bytes, image_v,
                               image_bytes, out_color, out_acc_ps, out_depth, grad_records, grad_records_bytes, backward_follows,
                               timing, stream, nullptr, fb_dev, fb_ticket);
}

}  // extern "C"

namespace {
// D_dev != nullptr: speculative launch — D is the CAPACITY (layouts, grids, sort geometry), the kernels read the instance count
// min(D_true, capacity) from *D_dev.
// fb_dev / fb_ticket: pinned status block (device address) for the feedback publication, msgs_view_t.feedback_tag.
int forward_stage2_impl(const msgs_view_t* view, const msgs_gaussians_t* g, const void* geom_v, size_t geom_bytes, int64_t D,
                        void* binning_v, size_t binning_bytes, void* scratch_v, size_t scratch_bytes, void* image_v,
                        size_t image_bytes, float* out_color, float* out_acc_ps, float* out_depth, void* grad_records,
                        size_t grad_records_bytes, int backward_follows, const msgs_timing_t* timing, void* stream,
                        const uint32_t* D_dev, uint64_t* fb_dev, uint64_t fb_ticket) {
    int rc = check_inputs(view, g);
    if (rc) return rc;
    if (D < 0 || D > 0xFFFFFFFFll) return MSGS_ERR_TOO_MANY;
    const int P = g->P, W = view->image_width, H = view->image_height;
    if (!binning_v || !image_v || !out_color || !out_acc_ps || !out_depth) return MSGS_ERR_INVALID_ARG;
    if (P > 0 && (!geom_v || geom_bytes < msgs_geom_bytes(P))) return MSGS_ERR_CAPACITY;
    if (binning_bytes < msgs_binning_bytes(D, W, H) || image_bytes < msgs_image_bytes(W, H)) return MSGS_ERR_CAPACITY;
    if (D > 0 && (!scratch_v || scratch_bytes < msgs_stage2_scratch_bytes(D, W, H))) return MSGS_ERR_CAPACITY;
    if (grad_records && grad_records_bytes < msgs_backward_scratch_bytes(P)) return MSGS_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    char* geom = const_cast<char*>((const char*)geom_v);        // (slab mode writes its header and second scan into geom)
    char* binning = (char*)binning_v;
    char* scratch = (char*)scratch_v;
    char* image = (char*)image_v;
    const ViewParams vp = make_view_params(view);
    const int num_tiles = vp.gx * vp.gy;
    const ImageLayout IL(W, H);
    const Timer tm{timing, s};
    const int tbits = tile_bits(num_tiles);
    // (ranges first, then ids: BinningLayout's positions do not depend on the instance count)
    const BinningLayout BL(D, num_tiles);
    uint32_t* ids = (uint32_t*)(binning + BL.ids);
    uint2* ranges = (uint2*)(binning + BL.ranges);
    float* final_T = (float*)(image + IL.final_T);
    uint32_t* n_contrib = (uint32_t*)(image + IL.n_contrib);
    uint32_t* tile_last = (uint32_t*)(image + IL.tile_last);
    const bool literal = g_deterministic.load() != 0;
    const bool feedback = view->feedback_tag != 0 && fb_dev != nullptr && P > 0 && forward_uses_quadrant_kernel(num_tiles) && !literal;
    unsigned long long* dtrav = feedback ? (unsigned long long*)(binning + BL.dtrav) : nullptr;
    const size_t clear_bytes = grad_records ? GRAD_REC_BYTES * (size_t)P : 0;

    // ---- depth slabs?  The buffers must serve this D in slab mode (SlabGeom); the speculative caller sized `D` accordingly
    bool slab = D > 0 && slab_wanted(view, g) && radix_sort_supports_device_count(D, 0, tbits);
    int64_t n_a_max = 0, cap_b = 0;
    if (slab) {
        int64_t cb, cs;
        stage2_capacities(view, binning_bytes, scratch_bytes, cb, cs);
        const SlabGeom SG(cb, cs, (double)view->slab_fraction, num_tiles, D);
        slab = SG.ok && SG.cap_d == D && SortScratch(SG.n_a_max).total <= SortScratch(SG.cap_b).total;
        n_a_max = SG.n_a_max;
        cap_b = SG.cap_b;
    }

    if (!slab) {
        const Stage2Scratch SL(D);
        uint32_t* keys_sorted = D > 0 ? (uint32_t*)(scratch + SL.total) : nullptr;
        bool ranges_prezeroed = false, keys16 = false;
        if (D > 0) {
            uint32_t* keys_a = (uint32_t*)(scratch + SL.keys_a);
            uint32_t* ids_a = (uint32_t*)(scratch + SL.ids_a);
            // emit also clears the tile sort's group-sum table and the tile-range array (two fill launches less)
            ZeroJob zj2{nullptr, 0, (uint32_t*)ranges, BL.ranges_and_dtrav_words()};
            const bool sort2_prezeroed = radix_sort_zero_region(D, 0, tbits, scratch + SL.sort, &zj2.p0, &zj2.n0);
            ranges_prezeroed = true;
            // tile ids (and the sentinel id = number of tiles) below 65536: the emit writes, the tile sort moves and the range
            // search reads 16-bit keys — 6 instead of 8 bytes per pair and pass
            keys16 = num_tiles < 65535 && radix_sort_keys16_ok(D, 0, tbits);
            // the queue of heavy Gaussians (binning.hip: one more launch) — not on a view the caller marked as ordinary
            // (msgs_view_t.no_heavy_queue: no covers closed anything lately, hence no crowd of giants in the first ranks either;
            // the few Gaussians with many instances are then emitted by their wave inside emit_kernel, as before round 5)
            uint32_t* heavy_q = view->no_heavy_queue ? nullptr : (uint32_t*)(scratch + SL.heavy_q);
            if (heavy_q && !D_dev) HIP_TRY(launch_zero(heavy_q, 4, s));     // (a speculative launch had it cleared by stage 1's scan)
            tm.begin(MSGS_K_EMIT);
            HIP_TRY(launch_emit(vp, P, geom, keys_a, ids_a, D, s, zj2, D_dev, keys16, heavy_q));
            tm.end(MSGS_K_EMIT);
            if ((rc = debug_sync(view, s))) return rc;
            tm.begin(MSGS_K_TILE_SORT);
            HIP_TRY(radix_sort_pairs(keys_a, ids_a, keys_sorted, ids, D, 0, tbits, scratch + SL.sort, s, sort2_prezeroed, nullptr,
                                     D_dev, keys16));
            tm.end(MSGS_K_TILE_SORT);
            if ((rc = debug_sync(view, s))) return rc;
        }
        tm.begin(MSGS_K_RANGES);
        HIP_TRY(launch_ranges(keys_sorted, D, ranges, num_tiles, s, ranges_prezeroed, D_dev, keys16));
        tm.end(MSGS_K_RANGES);
        if ((rc = debug_sync(view, s))) return rc;

        tm.begin(MSGS_K_BLEND_FWD);
        if (literal) {          // verification mode: the reference's per-pixel loop restated literally (literal.hip)
            if (grad_records && clear_bytes) HIP_TRY(launch_zero(grad_records, clear_bytes, s));
            HIP_TRY(launch_blend_forward_literal(vp, geom, P, ids, ranges, out_color, out_acc_ps, out_depth, final_T, n_contrib,
                                                 (uint32_t*)(image + IL.tile_order) + num_tiles, s));
        } else {
            if (D == 0 && dtrav) HIP_TRY(launch_zero(dtrav, 8 * (size_t)DTRAV_SLOTS, s));      // (no emit ran: nobody cleared them)
            const FwdSlabArgs fa{0, nullptr, nullptr, nullptr, dtrav};
            HIP_TRY(launch_blend_forward(vp, geom, ids, ranges, out_color, out_acc_ps, out_depth, final_T, n_contrib, tile_last,
                                         grad_records, clear_bytes, s, dtrav ? &fa : nullptr));
        }
    } else {
        // ---------------- slab A: the nearest ranks, up to slab_fraction * D instances ----------------
        const GeomLayout GL(P);
        SlabHeader* hdr = reinterpret_cast<SlabHeader*>(geom + GL.slab_hdr);
        uint32_t* open_bits = (uint32_t*)(image + IL.open_bits);
        uint32_t* open_list = (uint32_t*)(image + IL.open_list);
        const Stage2Scratch SL(cap_b);                       // the scratch is laid out for slab B's capacity; A uses its head
        uint32_t* keys_a = (uint32_t*)(scratch + SL.keys_a);
        uint32_t* ids_a = (uint32_t*)(scratch + SL.ids_a);
        uint32_t* keys_sorted = (uint32_t*)(scratch + SL.total);
        uint32_t* heavy_q = view->no_heavy_queue ? nullptr : (uint32_t*)(scratch + SL.heavy_q);
        const bool keys16 = num_tiles < 65535 && radix_sort_keys16_ok(n_a_max, 0, tbits) && radix_sort_keys16_ok(cap_b, 0, tbits);
        tm.begin(MSGS_K_EMIT);
        HIP_TRY(launch_slab_split(P, geom, D, D_dev, view->slab_fraction, open_bits, num_tiles, s));
        if (heavy_q) HIP_TRY(launch_zero(heavy_q, 4, s));
        ZeroJob zjA{nullptr, 0, (uint32_t*)ranges, BL.ranges_and_dtrav_words()};
        const bool sortA_prezeroed = radix_sort_zero_region(n_a_max, 0, tbits, scratch + SL.sort, &zjA.p0, &zjA.n0);
        HIP_TRY(launch_emit(vp, P, geom, keys_a, ids_a, n_a_max, s, zjA, &hdr->DA, keys16, heavy_q, 1, nullptr, D));
        tm.end(MSGS_K_EMIT);
        if ((rc = debug_sync(view, s))) return rc;
        tm.begin(MSGS_K_TILE_SORT);
        HIP_TRY(radix_sort_pairs(keys_a, ids_a, keys_sorted, ids, n_a_max, 0, tbits, scratch + SL.sort, s, sortA_prezeroed, nullptr,
                                 &hdr->DA, keys16));
        tm.end(MSGS_K_TILE_SORT);
        if ((rc = debug_sync(view, s))) return rc;
        tm.begin(MSGS_K_RANGES);
        HIP_TRY(launch_ranges(keys_sorted, n_a_max, ranges, num_tiles, s, true, &hdr->DA, keys16, 0u));
        tm.end(MSGS_K_RANGES);
        if ((rc = debug_sync(view, s))) return rc;
        tm.begin(MSGS_K_BLEND_FWD);
        const FwdSlabArgs fa{1, open_bits, open_list, &hdr->n_open, dtrav};
        HIP_TRY(launch_blend_forward(vp, geom, ids, ranges, out_color, out_acc_ps, out_depth, final_T, n_contrib, tile_last,
                                     grad_records, clear_bytes, s, &fa));
        tm.end(MSGS_K_BLEND_FWD);
        if ((rc = debug_sync(view, s))) return rc;

        // ---------------- slab B: the complete lists of the tiles slab A left open ----------------
        tm.begin(MSGS_K_SLAB_B);
        uint32_t* offs_b = (uint32_t*)(geom + GL.offs_b);
        HIP_TRY(launch_slab_recount(vp, P, geom, open_bits, open_list, D, D_dev, s));
        HIP_TRY(exclusive_scan_u32(offs_b, nullptr, offs_b, P, (uint64_t*)(geom + GL.scan_b), &hdr->total_b, s, nullptr, nullptr, 0,
                                   (const uint32_t*)(geom + GL.nvalid), &hdr->DB, (uint64_t)cap_b, nullptr, heavy_q, &hdr->pad0));
        ZeroJob zjB{nullptr, 0, nullptr, 0};
        const bool sortB_prezeroed = radix_sort_zero_region(cap_b, 0, tbits, scratch + SL.sort, &zjB.p0, &zjB.n0);
        HIP_TRY(launch_emit(vp, P, geom, keys_a, ids_a, cap_b, s, zjB, &hdr->DB, keys16, heavy_q, 2, open_bits));
        if ((rc = debug_sync(view, s))) return rc;
        HIP_TRY(radix_sort_pairs(keys_a, ids_a, keys_sorted, ids + n_a_max, cap_b, 0, tbits, scratch + SL.sort, s, sortB_prezeroed,
                                 nullptr, &hdr->DB, keys16));
        if ((rc = debug_sync(view, s))) return rc;
        HIP_TRY(launch_ranges(keys_sorted, cap_b, ranges, num_tiles, s, true, &hdr->DB, keys16, (uint32_t)n_a_max));
        const FwdSlabArgs fb{2, open_bits, open_list, &hdr->n_open, dtrav};
        HIP_TRY(launch_blend_forward(vp, geom, ids, ranges, out_color, out_acc_ps, out_depth, final_T, n_contrib, tile_last,
                                     nullptr, 0, s, &fb));
        tm.end(MSGS_K_SLAB_B);
    }
    if (backward_follows && !literal)  // give the backward's one-wave-per-tile kernel a heaviest-first launch order
        HIP_TRY(launch_tile_order(vp, tile_last, (uint32_t*)(image + IL.tile_order), s));
    if (feedback) {
        const GeomLayout GL(P);
        HIP_TRY(launch_forward_feedback(dtrav, reinterpret_cast<const SlabHeader*>(geom + GL.slab_hdr), slab ? 1 : 0, D,
                                        D_dev, (uint32_t)view->feedback_tag, fb_ticket, fb_dev, s));
    }
    if (!slab) tm.end(MSGS_K_BLEND_FWD);
    return debug_sync(view, s);
}
}  // namespace

extern "C" {

int msgs_backward(const msgs_view_t* view, const msgs_gaussians_t* g, const int32_t* radii, const void* geom_v,
                  size_t geom_bytes, int64_t D, const void* binning_v, size_t binning_bytes, const void* image_v,
                  size_t image_bytes, const float* dL_dcolor, void* scratch_v, size_t scratch_bytes,
                  const msgs_grads_t* grads, const msgs_timing_t* timing, void* stream) {
    int rc = check_inputs(view, g);
    if (rc) return rc;
    if (!grads || !dL_dcolor) return MSGS_ERR_INVALID_ARG;
    const int P = g->P, W = view->image_width, H = view->image_height;
    if (P == 0) return MSGS_OK;
    if (!radii || !geom_v || !binning_v || !image_v || !scratch_v) return MSGS_ERR_INVALID_ARG;
    const bool det = g_deterministic.load() != 0;
    if (geom_bytes < msgs_geom_bytes(P) || binning_bytes < msgs_binning_bytes(D, W, H) ||
        image_bytes < msgs_image_bytes(W, H) ||
        scratch_bytes < (det ? msgs_backward_scratch_bytes_deterministic(P, D) : msgs_backward_scratch_bytes(P)))
        return MSGS_ERR_CAPACITY;
    if (g->shs && !g->raw_params && !grads->dL_dshs) return MSGS_ERR_INVALID_ARG;
    // raw modes: either both SH gradient tensors, or neither plus dL_dcolors (factored SH gradient, msgs.h)
    if (g->raw_params && (!grads->dL_dfeatures_dc != !grads->dL_dfeatures_rest)) return MSGS_ERR_INVALID_ARG;
    if (g->raw_params && !grads->dL_dfeatures_dc && !grads->dL_dcolors && !grads->adam_in_backward) return MSGS_ERR_INVALID_ARG;
    if (const msgs_adam_in_backward_t* ad = grads->adam_in_backward) {
        // the optimizer step inside the per-Gaussian kernel (msgs.h): gradients of the LEAVES, on the staged SH rows, one view
        if (g->raw_params != 1 || !g->features_dc || !g->features_rest || g->shs || g->colors_precomp || g->cov3D_precomp ||
            !g->scales || !g->rotations || view->sh_coeffs != 16 || grads->accumulate || grads->wait_before_accumulate ||
            ad->step < 1)
            return MSGS_ERR_INVALID_ARG;
        uintptr_t bits = (uintptr_t)g->means3D | (uintptr_t)g->features_dc | (uintptr_t)g->features_rest |
                         (uintptr_t)g->opacities | (uintptr_t)g->scales | (uintptr_t)g->rotations;
        for (int t = 0; t < 6; ++t) {
            if (!ad->t[t].exp_avg || !ad->t[t].exp_avg_sq) return MSGS_ERR_INVALID_ARG;
            bits |= (uintptr_t)ad->t[t].exp_avg | (uintptr_t)ad->t[t].exp_avg_sq;
        }
        if (bits & 15) return MSGS_ERR_INVALID_ARG;
    }
    // accumulate mode is implemented on the staged SH rows (K = 16) and not for the factored SH gradient
    if (grads->accumulate && ((g->shs && !g->raw_params && view->sh_coeffs != 16) ||
                              (g->raw_params && !grads->dL_dfeatures_dc)))
        return MSGS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const char* geom = (const char*)geom_v;
    const char* binning = (const char*)binning_v;
    const char* image = (const char*)image_v;
    const ViewParams vp = make_view_params(view);
    const BinningLayout BL(D, vp.gx * vp.gy);
    const ImageLayout IL(W, H);
    const Timer tm{timing, s};
    grad_acc_t* grad_rec = (grad_acc_t*)scratch_v;

    // deterministic mode writes its own scratch layout from scratch: a "cleared by the forward" claim cannot hold there
    if (det && grads->scratch_is_clear) return MSGS_ERR_INVALID_ARG;
    if (!grads->scratch_is_clear)           // (else: cleared by the blend kernel of this view's forward, msgs.h)
        HIP_TRY(launch_zero(grad_rec, GRAD_REC_BYTES * (size_t)P, s));
    tm.begin(MSGS_K_BLEND_BWD);
    if (det)      // grad_rec is the first region of the deterministic scratch layout
        HIP_TRY(launch_blend_backward_det(vp, P, geom, (const uint32_t*)(binning + BL.ids), D,
                                          (const uint2*)(binning + BL.ranges), (const float*)(image + IL.final_T),
                                          (const uint32_t*)(image + IL.n_contrib), dL_dcolor, (char*)scratch_v, s));
    else
        HIP_TRY(launch_blend_backward(vp, geom, (const uint32_t*)(binning + BL.ids), (const uint2*)(binning + BL.ranges),
                                      (const float*)(image + IL.final_T), (const uint32_t*)(image + IL.n_contrib),
                                      dL_dcolor, grad_rec, s, (const uint32_t*)(image + IL.tile_order)));
    tm.end(MSGS_K_BLEND_BWD);
    if ((rc = debug_sync(view, s))) return rc;

    tm.begin(MSGS_K_PREPROCESS_BWD);
    if (det)      // the nine TEXTBOOK sums per Gaussian ([P, 9] doubles) as they are
        HIP_TRY(launch_preprocess_backward(vp, *g, radii, geom, grad_rec, *grads, s, true));
    else
        HIP_TRY(launch_preprocess_backward(vp, *g, radii, geom, grad_rec, *grads, s));
    tm.end(MSGS_K_PREPROCESS_BWD);
    return debug_sync(view, s);
}

int msgs_backward_per_gaussian(const msgs_view_t* view, const msgs_gaussians_t* g, const int32_t* radii,
                               const void* geom_v, size_t geom_bytes, const double* sums2d, const msgs_grads_t* grads,
                               void* stream) {
    int rc = check_inputs(view, g);
    if (rc) return rc;
    if (!grads || g->raw_params || grads->adam_in_backward) return MSGS_ERR_INVALID_ARG;
    // the textbook branch of the per-Gaussian kernel neither waits for `wait_before_accumulate` nor records `accumulated`:
    // accumulation across views is msgs_backward's contract only
    if (grads->accumulate || grads->wait_before_accumulate || grads->accumulated) return MSGS_ERR_INVALID_ARG;
    const int P = g->P;
    if (P == 0) return MSGS_OK;
    if (!radii || !geom_v || !sums2d) return MSGS_ERR_INVALID_ARG;
    if (geom_bytes < msgs_geom_bytes(P)) return MSGS_ERR_CAPACITY;
    if (g->shs && !grads->dL_dshs) return MSGS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_preprocess_backward(make_view_params(view), *g, radii, (const char*)geom_v,
                                       reinterpret_cast<const grad_acc_t*>(sums2d), *grads, s, true));
    return debug_sync(view, s);
}

int msgs_sh_grad_from_views(int32_t P, int32_t n_views, int32_t sh_degree, const float* means3D, const float* campos,
                            int64_t campos_stride, const float* drgb, int64_t drgb_stride, float scale,
                            float* dL_dfeatures_dc, float* dL_dfeatures_rest, void* stream) {
    if (P < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3 || campos_stride < 0 || drgb_stride < 0)
        return MSGS_ERR_INVALID_ARG;
    if (P > 0 && (!means3D || !campos || !drgb || !dL_dfeatures_dc || !dL_dfeatures_rest)) return MSGS_ERR_INVALID_ARG;
    HIP_TRY(launch_sh_grad_from_views(P, n_views, sh_degree, means3D, campos, campos_stride, drgb, drgb_stride, scale,
                                      dL_dfeatures_dc, dL_dfeatures_rest, (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                      uint8_t* present, void* stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return MSGS_ERR_INVALID_ARG;
    HIP_TRY(launch_mark_visible(P, means3D, viewmatrix, projmatrix, present, (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_binning_stats(const msgs_view_t* view, int32_t P, const int32_t* radii, const void* binning,
                       size_t binning_bytes, const void* image_v, size_t image_bytes, void* scratch,
                       size_t scratch_bytes, int64_t* out_host, void* stream) {
    (void)binning; (void)binning_bytes;
    if (!view || !image_v || !scratch || !out_host || scratch_bytes < 16) return MSGS_ERR_INVALID_ARG;
    const int W = view->image_width, H = view->image_height;
    if (image_bytes < msgs_image_bytes(W, H)) return MSGS_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    const ViewParams vp = make_view_params(view);
    const ImageLayout IL(W, H);
    unsigned long long* dev = (unsigned long long*)scratch;
    HIP_TRY(launch_binning_stats(vp, P, radii, (const uint32_t*)((const char*)image_v + IL.n_contrib), dev, s));
    unsigned long long host[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(host, dev, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    out_host[0] = (int64_t)host[0];
    out_host[1] = (int64_t)host[1];
    return MSGS_OK;
}

int msgs_forward_info(int64_t* out_host) {
    if (!out_host) return MSGS_ERR_INVALID_ARG;
    out_host[0] = (int64_t)(t_last_info & 0xFFFFFFFFull);
    out_host[1] = (int64_t)(t_last_info >> 32) != 0 ? 1 : 0;
    // feedback publication (forward_feedback_kernel): {D_trav | overflow << 63, tag | n_open << 32, DA | DB << 32, D | ticket << 40}
    const uint64_t w4 = t_last_feedback[0], w5 = t_last_feedback[1], w6 = t_last_feedback[2], w7 = t_last_feedback[3];
    const uint32_t n_open = (uint32_t)(w5 >> 32);
    out_host[2] = (int64_t)(w5 & 0xFFFFFFFFull);                       // tag (0: nothing published yet)
    out_host[3] = (int64_t)(w7 & 0xFFFFFFFFFFull);                     // D
    out_host[4] = (int64_t)(w4 & ~(1ull << 63));                       // D_trav
    out_host[5] = n_open == 0xFFFFFFFFu ? -1 : (int64_t)n_open;        // tiles left open by slab A (-1: single pass)
    out_host[6] = (int64_t)(w6 & 0xFFFFFFFFull);                       // DA
    out_host[7] = (w4 >> 63) ? -1 : (int64_t)(w6 >> 32);               // DB (-1: slab B outgrew its buffers — never observed)
    return MSGS_OK;
}

int msgs_slab_stats(const void* geom_v, size_t geom_bytes, int32_t P, int64_t* out_host, void* stream) {
    if (!geom_v || !out_host || P <= 0) return MSGS_ERR_INVALID_ARG;
    if (geom_bytes < msgs_geom_bytes(P)) return MSGS_ERR_CAPACITY;
    const GeomLayout GL(P);
    SlabHeader h;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&h, (const char*)geom_v + GL.slab_hdr, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    out_host[0] = h.active;
    out_host[1] = h.rA;
    out_host[2] = h.DA;
    out_host[3] = h.n_open;
    out_host[4] = (int64_t)h.total_b;
    out_host[5] = h.pad0;          // 1: slab B outgrew its buffers (never observed)
    return MSGS_OK;
}

int msgs_occlusion_stats(const void* geom_v, size_t geom_bytes, int32_t P, int64_t* out_host, void* stream) {
    if (!geom_v || !out_host || P <= 0) return MSGS_ERR_INVALID_ARG;
    if (geom_bytes < msgs_geom_bytes(P)) return MSGS_ERR_CAPACITY;
    const GeomLayout GL(P);
    OccHeader h;
    hipStream_t s = (hipStream_t)stream;
    static uint32_t table[OCC_MAX_BLOCKS];               // (diagnostic entry: serialised by the synchronisation below)
    HIP_TRY(hipMemcpyAsync(&h, (const char*)geom_v + GL.occ_hdr, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 8; ++k) out_host[k] = 0;
    out_host[0] = h.enabled ? (h.watchdog ? 2 : 1) : 0;     // 2: a barrier wait of the pass expired (never observed)
    if (!h.enabled) return MSGS_OK;
    const int n_blocks = (int)(h.nbx * h.nby);
    if (n_blocks < 0 || n_blocks > OCC_MAX_BLOCKS) return MSGS_ERR_INTERNAL;
    if (h.any_closed) {       // (a view in which nothing closed leaves the table unwritten: nobody reads it)
        HIP_TRY(hipMemcpyAsync(table, (const char*)geom_v + GL.occ_cut, 4 * (size_t)n_blocks, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    } else {
        for (int k = 0; k < n_blocks; ++k) table[k] = 0xFFFFu;
    }
    // cut-offs are depth buckets (0xFFFF = open): reported as the depth key at the far end of the bucket
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    int closed = 0;
    for (int k = 0; k < n_blocks; ++k) {
        const uint32_t key = table[k] >= 0xFFFFu ? 0xFFFFFFFFu : (((table[k] + OCC_KEY_BASE + 1u) << OCC_KEY_SHIFT) - 1u);
        closed += key != 0xFFFFFFFFu;
        lo = key < lo ? key : lo;
        hi = key > hi ? key : hi;
    }
    out_host[1] = h.n_heavy;
    out_host[2] = h.n_cand;
    out_host[3] = closed;
    out_host[4] = n_blocks;
    out_host[5] = 1u << h.block_log2;
    out_host[6] = lo;
    out_host[7] = hi;
    return MSGS_OK;
}

int msgs_blend_lane_stats(const msgs_view_t* view, const void* geom, size_t geom_bytes, int32_t P, int64_t D,
                          const void* binning_v, size_t binning_bytes, const void* image_v, size_t image_bytes,
                          void* scratch, size_t scratch_bytes, int64_t* out_host, void* stream) {
    if (!view || !geom || !binning_v || !scratch || !out_host || scratch_bytes < 64 || P < 0) return MSGS_ERR_INVALID_ARG;
    const int W = view->image_width, H = view->image_height;
    if (geom_bytes < msgs_geom_bytes(P) || binning_bytes < msgs_binning_bytes(D, W, H)) return MSGS_ERR_CAPACITY;
    if (image_v && image_bytes < msgs_image_bytes(W, H)) return MSGS_ERR_CAPACITY;
    hipStream_t s = (hipStream_t)stream;
    const ViewParams vp = make_view_params(view);
    const BinningLayout BL(D, vp.gx * vp.gy);
    const ImageLayout IL(W, H);
    const char* binning = (const char*)binning_v;
    unsigned long long* dev = (unsigned long long*)scratch;
    HIP_TRY(launch_blend_lane_stats(vp, (const char*)geom, (const uint32_t*)(binning + BL.ids),
                                    (const uint2*)(binning + BL.ranges), dev, s));
    if (image_v)
        HIP_TRY(launch_blend_backward_lane_stats(vp, (const char*)geom, (const uint32_t*)(binning + BL.ids),
                                                 (const uint2*)(binning + BL.ranges),
                                                 (const float*)((const char*)image_v + IL.final_T),
                                                 (const uint32_t*)((const char*)image_v + IL.n_contrib), dev + 4, s));
    unsigned long long host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(host, dev, 64, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 3; ++k) out_host[k] = (int64_t)host[k];
    for (int k = 0; k < 4; ++k) out_host[3 + k] = image_v ? (int64_t)host[4 + k] : -1;
    return MSGS_OK;
}

int msgs_adam_step(const msgs_adam_tensor_t* tensors, int32_t n_tensors, int64_t step, double beta1, double beta2,
                   double eps, void* stream) {
    if (n_tensors < 0 || n_tensors > MSGS_ADAM_MAX_TENSORS || step < 1 || (n_tensors && !tensors))
        return MSGS_ERR_INVALID_ARG;
    for (int k = 0; k < n_tensors; ++k) {
        const msgs_adam_tensor_t& t = tensors[k];
        if (t.n < 0 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) return MSGS_ERR_INVALID_ARG;
        if (t.n > (int64_t)1 << 40) return MSGS_ERR_TOO_MANY;
    }
    HIP_TRY(launch_adam(tensors, n_tensors, step, beta1, beta2, eps, (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_densify_stats(const msgs_densify_stats_t* d, void* stream) {
    if (!d || d->P < 0 || d->reso_lvls < 1 || d->reso_lvl < 0 || d->reso_lvl >= d->reso_lvls) return MSGS_ERR_INVALID_ARG;
    if (d->P == 0 || d->flags == 0) return MSGS_OK;
    if (!d->radii) return MSGS_ERR_INVALID_ARG;
    if ((d->flags & MSGS_STATS_BASE_MASK) && !d->base_mask) return MSGS_ERR_INVALID_ARG;
    if ((d->flags & MSGS_STATS_PIXEL_SIZES) &&
        (!d->pixel_sizes || !d->target_reso_lvl || !d->max_pixel_sizes || !d->min_pixel_sizes)) return MSGS_ERR_INVALID_ARG;
    if ((d->flags & MSGS_STATS_DENSIFY) && (!d->means2D_grad || !d->xyz_gradient_accum || !d->denom || !d->max_radii2D))
        return MSGS_ERR_INVALID_ARG;
    HIP_TRY(launch_densify_stats(*d, (hipStream_t)stream));
    return MSGS_OK;
}

static int loss_args_ok(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float lambda) {
    if (!img || !gt || C < 1 || H < 1 || W < 1 || !(lambda >= 0.f && lambda <= 1.f)) return MSGS_ERR_INVALID_ARG;
    if ((int64_t)C * H * W > (int64_t)1 << 31 || C > 65535) return MSGS_ERR_TOO_MANY;
    return MSGS_OK;
}

size_t msgs_loss_scratch_bytes(int32_t C, int32_t H, int32_t W) {
    if (C < 1 || H < 1 || W < 1) return 0;
    return loss_scratch_bytes(C, H, W);
}

int msgs_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      float* out3, void* scratch, size_t scratch_bytes, int32_t keep_for_backward, void* stream) {
    if (int rc = loss_args_ok(img, gt, C, H, W, lambda_dssim)) return rc;
    if (!out3 || !scratch) return MSGS_ERR_INVALID_ARG;
    if (scratch_bytes < loss_scratch_bytes(C, H, W)) return MSGS_ERR_CAPACITY;
    HIP_TRY(launch_loss_forward(img, gt, C, H, W, lambda_dssim, out3, (char*)scratch, keep_for_backward != 0,
                                (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_loss_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                       const float* upstream, const void* scratch, size_t scratch_bytes, float* dL_dimg, void* stream) {
    if (int rc = loss_args_ok(img, gt, C, H, W, lambda_dssim)) return rc;
    if (!dL_dimg || !scratch) return MSGS_ERR_INVALID_ARG;
    if (scratch_bytes < loss_scratch_bytes(C, H, W)) return MSGS_ERR_CAPACITY;
    HIP_TRY(launch_loss_backward(img, gt, C, H, W, lambda_dssim, upstream, (const char*)scratch, dL_dimg,
                                 (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_ssim_window(float* taps11_host) {
    if (!taps11_host) return MSGS_ERR_INVALID_ARG;
    ssim_window_host(taps11_host);
    return MSGS_OK;
}

size_t msgs_knn_scratch_bytes(int64_t P) { return knn_scratch_bytes(P); }

int msgs_dist2_knn3(const float* points, int64_t P, float* mean_dist2, void* scratch, size_t scratch_bytes,
                    void* stream) {
    if (P < 4 || !points || !mean_dist2 || !scratch) return MSGS_ERR_INVALID_ARG;
    if (P > 0x7FFFFFFFll) return MSGS_ERR_TOO_MANY;
    if (scratch_bytes < knn_scratch_bytes(P)) return MSGS_ERR_CAPACITY;
    HIP_TRY(knn_mean_dist2(points, P, mean_dist2, (char*)scratch, (hipStream_t)stream));
    return MSGS_OK;
}

size_t msgs_voxel_pool_scratch_bytes(int64_t M) { return voxel_pool_scratch_bytes(M); }

int msgs_voxel_pool_build(const float* positions, int64_t M, float voxel_size, uint32_t* order, uint32_t* seg_start,
                          int32_t* voxel_index, void* scratch, size_t scratch_bytes, int64_t* num_voxels_host,
                          void* stream) {
    if (!num_voxels_host || M < 0 || !(voxel_size > 0.f)) return MSGS_ERR_INVALID_ARG;
    *num_voxels_host = 0;
    if (M == 0) return MSGS_OK;
    if (M > 0x7FFFFFFFll) return MSGS_ERR_TOO_MANY;
    if (!positions || !order || !seg_start || !scratch) return MSGS_ERR_INVALID_ARG;
    if (scratch_bytes < voxel_pool_scratch_bytes(M)) return MSGS_ERR_CAPACITY;
    HIP_TRY(voxel_pool_build(positions, M, voxel_size, order, seg_start, voxel_index, (char*)scratch, num_voxels_host,
                             (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_voxel_pool_average(const float* features, int32_t F, const uint32_t* order, const uint32_t* seg_start,
                            int64_t num_voxels, float* out, void* stream) {
    if (F < 0 || num_voxels < 0) return MSGS_ERR_INVALID_ARG;
    if (F == 0 || num_voxels == 0) return MSGS_OK;
    if (!features || !order || !seg_start || !out) return MSGS_ERR_INVALID_ARG;
    HIP_TRY(voxel_pool_average(features, F, order, seg_start, num_voxels, out, (hipStream_t)stream));
    return MSGS_OK;
}

int msgs_timing_create(msgs_timing_t* t) {
    if (!t) return MSGS_ERR_INVALID_ARG;
    std::memset(t, 0, sizeof(*t));
    for (int i = 0; i < 2 * MSGS_K_COUNT; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        t->ev[i] = (void*)e;
    }
    return MSGS_OK;
}

int msgs_timing_destroy(msgs_timing_t* t) {
    if (!t) return MSGS_ERR_INVALID_ARG;
    for (int i = 0; i < 2 * MSGS_K_COUNT; ++i)
        if (t->ev[i]) { (void)hipEventDestroy((hipEvent_t)t->ev[i]); t->ev[i] = nullptr; }
    return MSGS_OK;
}

int msgs_timing_read(const msgs_timing_t* t, float* ms_host) {
    if (!t || !ms_host) return MSGS_ERR_INVALID_ARG;
    for (int k = 0; k < MSGS_K_COUNT; ++k) {
        float ms = -1.0f;
        if (t->ev[2 * k] && t->ev[2 * k + 1]) {
            if (hipEventElapsedTime(&ms, (hipEvent_t)t->ev[2 * k], (hipEvent_t)t->ev[2 * k + 1]) != hipSuccess) {
                ms = -1.0f;
                (void)hipGetLastError();
            }
        }
        ms_host[k] = ms;
    }
    return MSGS_OK;
}

}  // extern "C"
