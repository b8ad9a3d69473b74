"""ctypes binding of lib/libmsgs_hip.so (C ABI: include/msgs.h).

This is the ONLY compute backend of the package: there is no CPU or PyTorch fallback.  If the
shared library is missing or cannot be loaded the import fails loudly.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  -- must be loaded first: libmsgs_hip.so has to bind to the SAME libamdhip64.so.7
#                     instance (matched by SONAME) that PyTorch-ROCm loaded, since streams cross the boundary

_PKG = os.path.dirname(os.path.abspath(__file__))
# MSGS_HIP_LIB, else the copy a `pip install` placed inside the package (ms-gs_amd/setup.py), else the source tree's lib/
_LIB_PATH = os.environ.get("MSGS_HIP_LIB") or next(
    (p for p in (os.path.join(_PKG, "libmsgs_hip.so"),) if os.path.exists(p)),
    os.path.join(os.path.dirname(_PKG), "lib", "libmsgs_hip.so"))

ABI_VERSION = 11

K_NAMES = ("preprocess", "depth_sort", "scan", "emit", "tile_sort", "ranges", "blend_fwd", "blend_bwd",
           "preprocess_bwd", "slab_b")
K_COUNT = len(K_NAMES)


class View(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("fade_size", C.c_float),
                ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
                ("filter_small", C.c_int32), ("filter_large", C.c_int32),
                ("prefiltered", C.c_int32), ("debug", C.c_int32),
                ("no_heavy_queue", C.c_int32), ("feedback_tag", C.c_int32),
                ("slab_fraction", C.c_float), ("reserved1", C.c_int32),
                ("bg", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p)]


class Gaussians(C.Structure):
    _fields_ = [("P", C.c_int32), ("raw_params", C.c_int32),
                ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
                ("cov3D_precomp", C.c_void_p), ("max_pixel_sizes", C.c_void_p),
                ("min_pixel_sizes", C.c_void_p), ("occ_multiplier", C.c_void_p),
                ("dc_delta", C.c_void_p), ("base_mask", C.c_void_p),
                ("features_dc", C.c_void_p), ("features_rest", C.c_void_p),
                ("rotations_raw", C.c_void_p)]


class Grads(C.Structure):
    _fields_ = [("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dshs", C.c_void_p),
                ("dL_dcolors", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p),
                ("dL_dfeatures_dc", C.c_void_p), ("dL_dfeatures_rest", C.c_void_p), ("factors_ready", C.c_void_p),
                ("scratch_is_clear", C.c_int32), ("accumulate", C.c_int32),
                ("wait_before_accumulate", C.c_void_p), ("accumulated", C.c_void_p), ("adam_in_backward", C.c_void_p)]


class AdamMoments(C.Structure):          # msgs_adam_moments_t
    _fields_ = [("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr", C.c_double)]


class AdamInBackward(C.Structure):       # msgs_adam_in_backward_t: means3D, features_dc, features_rest, opacities, scales, rotations
    _fields_ = [("step", C.c_int64), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("t", AdamMoments * 6)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_double)]


ADAM_MAX_TENSORS = 8
STATS_BASE_MASK, STATS_PIXEL_SIZES, STATS_DENSIFY = 1, 2, 4


class DensifyStats(C.Structure):
    _fields_ = [("P", C.c_int32), ("flags", C.c_int32), ("reso_lvl", C.c_int32), ("reso_lvls", C.c_int32),
                ("radii", C.c_void_p), ("pixel_sizes", C.c_void_p), ("means2D_grad", C.c_void_p),
                ("target_reso_lvl", C.c_void_p), ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p),
                ("max_radii2D", C.c_void_p), ("max_pixel_sizes", C.c_void_p), ("min_pixel_sizes", C.c_void_p),
                ("base_mask", C.c_void_p)]


class Timing(C.Structure):
    _fields_ = [("ev", C.c_void_p * (2 * K_COUNT))]


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"diff_gaussian_rasterization: HIP library not found at {_LIB_PATH}. Build it with "
            f"`make -C {os.path.dirname(_PKG)}` (hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = C.CDLL(_LIB_PATH)
    sz = C.c_size_t
    vp = C.c_void_p
    lib.msgs_abi_version.restype = C.c_int
    lib.msgs_error_string.restype = C.c_char_p
    lib.msgs_error_string.argtypes = [C.c_int]
    for name, args in (("msgs_geom_bytes", [C.c_int32]), ("msgs_stage1_scratch_bytes", [C.c_int32]),
                       ("msgs_binning_bytes", [C.c_int64, C.c_int32, C.c_int32]),
                       ("msgs_stage2_scratch_bytes", [C.c_int64, C.c_int32, C.c_int32]),
                       ("msgs_image_bytes", [C.c_int32, C.c_int32]),
                       ("msgs_binning_bytes_slab", [C.c_int64, C.c_int32, C.c_int32, C.c_float]),
                       ("msgs_stage2_scratch_bytes_slab", [C.c_int64, C.c_int32, C.c_int32]),
                       ("msgs_backward_scratch_bytes", [C.c_int32])):
        f = getattr(lib, name)
        f.restype = sz
        f.argtypes = args
    lib.msgs_forward_stage1.restype = C.c_int
    lib.msgs_forward_stage1.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, vp, vp, sz, vp, sz,
                                        C.POINTER(C.c_int64), C.POINTER(Timing), vp]
    lib.msgs_set_deterministic.restype = C.c_int
    lib.msgs_set_deterministic.argtypes = [C.c_int32]
    lib.msgs_get_deterministic.restype = C.c_int
    lib.msgs_get_deterministic.argtypes = []
    lib.msgs_backward_scratch_bytes_deterministic.restype = sz
    lib.msgs_backward_scratch_bytes_deterministic.argtypes = [C.c_int32, C.c_int64]
    lib.msgs_set_backward_generation.restype = C.c_int
    lib.msgs_set_backward_generation.argtypes = [C.c_int32]
    lib.msgs_set_blend_granularity.restype = C.c_int
    lib.msgs_set_blend_granularity.argtypes = [C.c_int32]
    lib.msgs_forward_info.restype = C.c_int
    lib.msgs_forward_info.argtypes = [C.POINTER(C.c_int64)]
    lib.msgs_set_occlusion.restype = C.c_int
    lib.msgs_set_occlusion.argtypes = [C.c_int32]
    lib.msgs_occlusion_stats.restype = C.c_int
    lib.msgs_occlusion_stats.argtypes = [vp, sz, C.c_int32, C.POINTER(C.c_int64), vp]
    lib.msgs_slab_stats.restype = C.c_int
    lib.msgs_slab_stats.argtypes = [vp, sz, C.c_int32, C.POINTER(C.c_int64), vp]
    lib.msgs_forward.restype = C.c_int
    lib.msgs_forward.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz,
                                 vp, vp, vp, vp, sz, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(Timing), vp]
    lib.msgs_status_create.restype = C.c_int
    lib.msgs_status_create.argtypes = [C.POINTER(vp)]
    lib.msgs_status_destroy.restype = C.c_int
    lib.msgs_status_destroy.argtypes = [vp]
    lib.msgs_forward_launch.restype = C.c_int
    lib.msgs_forward_launch.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, sz,
                                        vp, vp, vp, vp, sz, C.c_int32, vp, C.POINTER(Timing), vp]
    lib.msgs_forward_finish.restype = C.c_int
    lib.msgs_forward_finish.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.msgs_preprocess_only.restype = C.c_int
    lib.msgs_preprocess_only.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, vp, vp, sz, vp]
    lib.msgs_forward_stage2.restype = C.c_int
    lib.msgs_forward_stage2.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, sz, C.c_int64, vp, sz, vp, sz,
                                        vp, sz, vp, vp, vp, vp, sz, C.c_int32, C.POINTER(Timing), vp]
    lib.msgs_backward.restype = C.c_int
    lib.msgs_backward.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, vp, sz, C.c_int64, vp, sz, vp, sz,
                                  vp, vp, sz, C.POINTER(Grads), C.POINTER(Timing), vp]
    lib.msgs_backward_per_gaussian.restype = C.c_int
    lib.msgs_backward_per_gaussian.argtypes = [C.POINTER(View), C.POINTER(Gaussians), vp, vp, sz, vp, C.POINTER(Grads), vp]
    lib.msgs_sh_grad_from_views.restype = C.c_int
    lib.msgs_sh_grad_from_views.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, C.c_int64,
                                                C.c_float, vp, vp, vp]
    lib.msgs_mark_visible.restype = C.c_int
    lib.msgs_mark_visible.argtypes = [C.c_int32, vp, vp, vp, vp, vp]
    lib.msgs_binning_stats.restype = C.c_int
    lib.msgs_binning_stats.argtypes = [C.POINTER(View), C.c_int32, vp, vp, sz, vp, sz, vp, sz,
                                       C.POINTER(C.c_int64), vp]
    lib.msgs_blend_lane_stats.restype = C.c_int
    lib.msgs_blend_lane_stats.argtypes = [C.POINTER(View), vp, sz, C.c_int32, C.c_int64, vp, sz, vp, sz, vp, sz,
                                          C.POINTER(C.c_int64), vp]
    lib.msgs_voxel_pool_scratch_bytes.restype = sz
    lib.msgs_voxel_pool_scratch_bytes.argtypes = [C.c_int64]
    lib.msgs_voxel_pool_build.restype = C.c_int
    lib.msgs_voxel_pool_build.argtypes = [vp, C.c_int64, C.c_float, vp, vp, vp, vp, sz, C.POINTER(C.c_int64), vp]
    lib.msgs_voxel_pool_average.restype = C.c_int
    lib.msgs_voxel_pool_average.argtypes = [vp, C.c_int32, vp, vp, C.c_int64, vp, vp]
    lib.msgs_adam_step.restype = C.c_int
    lib.msgs_adam_step.argtypes = [C.POINTER(AdamTensor), C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, vp]
    lib.msgs_densify_stats.restype = C.c_int
    lib.msgs_densify_stats.argtypes = [C.POINTER(DensifyStats), vp]
    lib.msgs_loss_scratch_bytes.restype = sz
    lib.msgs_loss_scratch_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.msgs_loss_forward.restype = C.c_int
    lib.msgs_loss_forward.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp, vp, sz, C.c_int32, vp]
    lib.msgs_loss_backward.restype = C.c_int
    lib.msgs_loss_backward.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp, vp, sz, vp, vp]
    lib.msgs_knn_scratch_bytes.restype = sz
    lib.msgs_knn_scratch_bytes.argtypes = [C.c_int64]
    lib.msgs_dist2_knn3.restype = C.c_int
    lib.msgs_dist2_knn3.argtypes = [vp, C.c_int64, vp, vp, sz, vp]
    lib.msgs_ssim_window.restype = C.c_int
    lib.msgs_ssim_window.argtypes = [C.POINTER(C.c_float)]
    for name in ("msgs_timing_create", "msgs_timing_destroy"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(Timing)]
    lib.msgs_timing_read.restype = C.c_int
    lib.msgs_timing_read.argtypes = [C.POINTER(Timing), C.POINTER(C.c_float)]
    got = lib.msgs_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libmsgs_hip.so ABI version {got} != expected {ABI_VERSION}; rebuild the library")
    return lib


lib = _load()

EXPORTS = ("msgs_abi_version", "msgs_error_string", "msgs_geom_bytes", "msgs_stage1_scratch_bytes",
           "msgs_binning_bytes", "msgs_stage2_scratch_bytes", "msgs_image_bytes", "msgs_backward_scratch_bytes",
           "msgs_forward_stage1", "msgs_forward_stage2", "msgs_backward", "msgs_mark_visible",
           "msgs_binning_stats", "msgs_timing_create", "msgs_timing_destroy", "msgs_timing_read",
           "msgs_voxel_pool_scratch_bytes", "msgs_voxel_pool_build", "msgs_voxel_pool_average", "msgs_adam_step",
           "msgs_densify_stats", "msgs_loss_scratch_bytes", "msgs_loss_forward", "msgs_loss_backward",
           "msgs_ssim_window", "msgs_preprocess_only", "msgs_knn_scratch_bytes",
           "msgs_dist2_knn3", "msgs_forward", "msgs_set_deterministic",
           "msgs_get_deterministic", "msgs_backward_scratch_bytes_deterministic",
           "msgs_set_backward_generation", "msgs_set_blend_granularity", "msgs_sh_grad_from_views",
           "msgs_blend_lane_stats", "msgs_backward_per_gaussian",
           "msgs_status_create", "msgs_status_destroy", "msgs_forward_launch", "msgs_forward_finish",
           "msgs_set_occlusion", "msgs_occlusion_stats", "msgs_forward_info", "msgs_binning_bytes_slab",
           "msgs_stage2_scratch_bytes_slab", "msgs_slab_stats")


def check(rc, where):
    if rc == 0:
        return
    msg = lib.msgs_error_string(int(rc)).decode()
    if rc in (-1, -4):
        raise ValueError(f"{where}: {msg}")
    raise RuntimeError(f"{where}: error {rc}: {msg}")


# ---- optional per-kernel timing (used by bench.py; thread-local so backward picks it up too) ----
class KernelTimer:
    """Owns 2*K_COUNT HIP events; pass to set_timer() to have the next forward/backward record them.
    only: optional subset of K_NAMES — the library is handed just those event pairs (an event record costs ~10 us of queue
    latency on this runtime: all nine classes slow a 1.1 ms step by ~0.16 ms, two of them by ~0.04)."""

    def __init__(self, only=None):
        self.t = Timing()
        check(lib.msgs_timing_create(C.byref(self.t)), "msgs_timing_create")
        self.active = self.t
        if only is not None:
            self.active = Timing()
            for k, name in enumerate(K_NAMES):
                if name in only:
                    self.active.ev[2 * k], self.active.ev[2 * k + 1] = self.t.ev[2 * k], self.t.ev[2 * k + 1]

    def read_ms(self):
        out = (C.c_float * K_COUNT)()
        check(lib.msgs_timing_read(C.byref(self.active), out), "msgs_timing_read")
        return {K_NAMES[k]: float(out[k]) for k in range(K_COUNT)}

    def __del__(self):
        try:
            lib.msgs_timing_destroy(C.byref(self.t))
        except Exception:
            pass


_timer_lock = threading.Lock()
_active_timer = None


def set_timer(timer):
    """Process-wide (forward runs on the main thread, backward on autograd's)."""
    global _active_timer
    with _timer_lock:
        _active_timer = timer


def timer_ptr():
    t = _active_timer
    return C.byref(t.active) if t is not None else None
