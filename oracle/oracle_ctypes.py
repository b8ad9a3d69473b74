"""ctypes binding of oracle/liboracle.so (the float32 CPU restatement).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  ms-gs_amd/ never imports this.  PARITY UNPINNED (see msgs_oracle.cpp).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


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


_SO_NAMES = {"f32": "liboracle.so", "f64": "liboracle64.so", "fma": "liboracle_fma.so"}


def _variant(f64=False, fma=False):
    if f64 and fma:
        raise ValueError("the float64 build has no contraction variant")
    return "f64" if f64 else ("fma" if fma else "f32")


def build(force=False, f64=False, fma=False):
    name = _SO_NAMES[_variant(f64, fma)]
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "msgs_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return so


_LIBS = {}


def lib(f64=False, fma=False):
    """liboracle.so (float32, contraction off: THE checker), — f64 — liboracle64.so: the same source compiled with every computed
    quantity in double (msgs_oracle.cpp, MSGS_ORACLE_F64), the float64 "truth" at the full BASELINE sizes, or — fma —
    liboracle_fma.so: the float32 source compiled with FMA contraction (-ffp-contract=fast -mfma), i.e. the reference algorithm
    under the OTHER legal float32 rounding (what nvcc emits by default for the real CUDA reference): the yardstick of how far
    two faithful float32 evaluations of this algorithm sit from each other (tests/fuzz_cases.py, tools/parity_floor.py)"""
    global _LIB
    key = _variant(f64, fma)
    if key == "f32" and _LIB is not None and _LIBS.get("f32") is not _LIB:
        _LIBS["f32"] = _LIB                     # (tools/parity_floor.with_oracle_lib swaps the float32 build through _LIB)
    if key == "f32" and _LIB is None:
        _LIBS.pop("f32", None)
    if _LIBS.get(key) is None:
        so = os.path.join(_HERE, _SO_NAMES[key])
        if not os.path.exists(so):
            build(f64=f64, fma=fma)
        L = C.CDLL(so)
        L.msgs_oracle_forward.restype = C.c_int
        L.msgs_oracle_forward.argtypes = [C.POINTER(View), C.POINTER(Gaussians), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_void_p), C.c_int]
        L.msgs_oracle_backward.restype = C.c_int
        L.msgs_oracle_backward.argtypes = [C.c_void_p, C.POINTER(View), C.POINTER(Gaussians), C.c_void_p,
                                           C.POINTER(Grads), C.c_int]
        L.msgs_oracle_backward_ex.restype = C.c_int
        L.msgs_oracle_backward_ex.argtypes = [C.c_void_p, C.POINTER(View), C.POINTER(Gaussians), C.c_void_p,
                                              C.POINTER(Grads), C.c_int, C.c_void_p]
        L.msgs_oracle_num_instances.restype = C.c_int64
        L.msgs_oracle_num_instances.argtypes = [C.c_void_p]
        for name in ("traversed", "valid_pairs", "evaluated_pairs"):
            f = getattr(L, "msgs_oracle_" + name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p]
        for name in ("final_T", "n_contrib", "depths", "conic_opacity", "rgb", "means2D", "cov3D", "rects",
                     "borderline_gaussians", "filter_edge", "shared_borderline_gaussians"):
            f = getattr(L, "msgs_oracle_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p]
        L.msgs_oracle_free.restype = None
        L.msgs_oracle_free.argtypes = [C.c_void_p]
        _LIBS[key] = L
        if key == "f32":
            _LIB = L
    return _LIBS[key]


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous().cpu()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class OracleResult:
    """Holds the oracle's forward outputs + the native state needed for backward."""

    def __init__(self, f64=False, fma=False):
        self.state = C.c_void_p(None)
        self._keep = []
        self.f64 = bool(f64)
        self.fma = bool(fma)
        self._lib = lib(f64, fma)             # the build that owns `state` (freed through the same one)

    def __del__(self):
        try:
            if self.state:
                self._lib.msgs_oracle_free(self.state)
                self.state = C.c_void_p(None)
        except Exception:
            pass

    def _arr(self, name, shape, dtype):
        """(float arrays of the float64 build hold doubles: pass torch.float64 for them)"""
        p = getattr(self._lib, "msgs_oracle_" + name)(self.state)
        n = int(np.prod(shape))
        if n == 0:
            return torch.zeros(shape, dtype=dtype)
        ct = {torch.float32: C.c_float, torch.float64: C.c_double, torch.int32: C.c_int32, torch.uint8: C.c_uint8}[dtype]
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)).copy()
        return torch.from_numpy(a).view(*shape)

    @property
    def num_instances(self):
        return int(self._lib.msgs_oracle_num_instances(self.state))

    @property
    def traversed(self):
        return int(self._lib.msgs_oracle_traversed(self.state))

    @property
    def valid_pairs(self):
        return int(self._lib.msgs_oracle_valid_pairs(self.state))

    @property
    def evaluated_pairs(self):
        return int(self._lib.msgs_oracle_evaluated_pairs(self.state))


def _usable_cpus():
    """affinity mask ∩ cgroup CPU quota (a container may see 256 CPUs and be granted 16: OpenMP's default of one thread
    per visible CPU then spends the quota spinning and the kernel parks the process for the rest of every 100 ms period)"""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.floor(int(q) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                n = min(n, max(1, q // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _threads(num_threads):
    """0 = every CPU this process may use (not every CPU it can see)"""
    return int(num_threads) if num_threads and num_threads > 0 else _usable_cpus()


class exp_double:
    """`with exp_double():` — the float32 oracle evaluates exp() in double and rounds once (msgs_oracle_set_exp_double), like the
    product's literal verification mode; forward AND backward of a comparison belong inside the block"""

    def __enter__(self):
        L = lib()
        L.msgs_oracle_set_exp_double.restype = C.c_int
        L.msgs_oracle_set_exp_double.argtypes = [C.c_int]
        self.prev = L.msgs_oracle_set_exp_double(1)
        return self

    def __exit__(self, *exc):
        lib().msgs_oracle_set_exp_double(self.prev)
        return False


def rasterize(scene, cam, settings, bg, *, use_cov_precomp=False, use_colors_precomp=False,
              cov3D_precomp=None, colors_precomp=None, num_threads=0, scale_modifier=1.0, f64=False, fma=False):
    """Forward on a scenes.Scene.  Returns OracleResult with .color/.acc_pixel_size/.depth/.radii/
    .pixel_sizes/.borderline tensors (CPU).  f64: the float64 build — the same float32 inputs, every computed quantity and
    every output in double (the "truth" of the three-way tests).  fma: the float32 build with FMA contraction (lib())."""
    L = lib(f64, fma)
    rdt = torch.float64 if f64 else torch.float32
    W, H = cam.image_width, cam.image_height
    P = scene.P
    r = OracleResult(f64, fma)
    t = dict(means3D=_f32(scene.means3D), opac=_f32(scene.opacities.reshape(-1)),
             maxps=_f32(scene.max_pixel_sizes), minps=_f32(scene.min_pixel_sizes),
             occ=_f32(scene.occ_multiplier), dcd=_f32(scene.dc_delta),
             base=scene.base_mask.to(torch.uint8).contiguous().cpu(),
             bg=_f32(bg), vm=_f32(cam.world_view_transform), pm=_f32(cam.full_proj_transform),
             cp=_f32(cam.camera_center))
    if use_cov_precomp:
        t["cov"] = _f32(cov3D_precomp)
        t["scales"] = t["rot"] = None
    else:
        t["cov"] = None
        t["scales"], t["rot"] = _f32(scene.scales), _f32(scene.rotations)
    if use_colors_precomp:
        t["col"] = _f32(colors_precomp)
        t["shs"] = None
    else:
        t["col"] = None
        t["shs"] = _f32(scene.shs)
    K = scene.shs.shape[1]
    v = View(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), float(scale_modifier),
             float(settings.get("fade_size", 1.0)), int(scene.sh_degree), int(K),
             int(bool(settings.get("filter_small", False))), int(bool(settings.get("filter_large", False))),
             0, 0, 0, 0, 0.0, 0, _ptr(t["bg"]), _ptr(t["vm"]), _ptr(t["pm"]), _ptr(t["cp"]))
    g = Gaussians(P, 0, _ptr(t["means3D"]), _ptr(t["shs"]), _ptr(t["col"]), _ptr(t["opac"]),
                  _ptr(t["scales"]), _ptr(t["rot"]), _ptr(t["cov"]), _ptr(t["maxps"]), _ptr(t["minps"]),
                  _ptr(t["occ"]), _ptr(t["dcd"]), _ptr(t["base"]), None, None, None)
    r.color = torch.zeros(3, H, W, dtype=rdt)
    r.acc_pixel_size = torch.zeros(H, W, dtype=rdt)
    r.depth = torch.zeros(H, W, dtype=rdt)
    r.radii = torch.zeros(P, dtype=torch.int32)
    r.pixel_sizes = torch.zeros(P, dtype=rdt)
    r.borderline = torch.zeros(H, W, dtype=torch.uint8)
    rc = L.msgs_oracle_forward(C.byref(v), C.byref(g), _ptr(r.color), _ptr(r.acc_pixel_size), _ptr(r.depth),
                               _ptr(r.radii), _ptr(r.pixel_sizes), _ptr(r.borderline), C.byref(r.state),
                               _threads(num_threads))
    if rc != 0:
        raise RuntimeError(f"msgs_oracle_forward failed: {rc}")
    r._keep = [t, v, g]
    r.view, r.g = v, g
    r.W, r.H, r.P, r.K = W, H, P, K
    r.has_shs = not use_colors_precomp
    r.has_sr = not use_cov_precomp
    r.borderline_gaussians = r._arr("borderline_gaussians", (P,), torch.uint8).bool()
    # Gaussians whose multi-scale filter decision could flip in another float32 implementation (msgs_oracle.h)
    r.filter_edge = r._arr("filter_edge", (P,), torch.uint8).bool()
    r.borderline_gaussians |= r.filter_edge
    # tier 2: Gaussians that share a pixel with ANY undecided discrete decision (msgs_oracle.h).  They stay in the strict checks;
    # the flag explains an exceedance after the fact (parity_utils.classify_exceedance)
    r.shared_borderline_gaussians = r._arr("shared_borderline_gaussians", (P,), torch.uint8).bool() | r.borderline_gaussians
    return r


def backward(r, dL_dcolor, num_threads=0, want_sums2d=False):
    """Backward on an OracleResult; returns dict of CPU gradient tensors (float32; float64 for a result of the float64
    build).  want_sums2d: also "sums2d", the [P,9] float64 per-Gaussian 2-D gradient sums between the blend backward and the
    per-Gaussian backward (msgs_oracle.h, msgs_oracle_backward_ex)."""
    L = r._lib
    rdt = torch.float64 if r.f64 else torch.float32
    P, K = r.P, r.K
    dl = _f32(dL_dcolor)
    out = dict(means3D=torch.zeros(P, 3, dtype=rdt), means2D=torch.zeros(P, 3, dtype=rdt), opacities=torch.zeros(P, 1, dtype=rdt))
    if r.has_shs:
        out["shs"] = torch.zeros(P, K, 3, dtype=rdt)
    else:
        out["colors_precomp"] = torch.zeros(P, 3, dtype=rdt)
    if r.has_sr:
        out["scales"] = torch.zeros(P, 3, dtype=rdt)
        out["rotations"] = torch.zeros(P, 4, dtype=rdt)
    else:
        out["cov3D_precomp"] = torch.zeros(P, 6, dtype=rdt)
    gr = Grads(_ptr(out["means3D"]), _ptr(out["means2D"]), _ptr(out.get("shs")), _ptr(out.get("colors_precomp")),
               _ptr(out["opacities"]), _ptr(out.get("scales")), _ptr(out.get("rotations")),
               _ptr(out.get("cov3D_precomp")), None, None, None)
    sums = torch.zeros(P, 9, dtype=torch.float64) if want_sums2d else None
    rc = L.msgs_oracle_backward_ex(r.state, C.byref(r.view), C.byref(r.g), _ptr(dl), C.byref(gr), _threads(num_threads),
                                   _ptr(sums))
    if rc != 0:
        raise RuntimeError(f"msgs_oracle_backward failed: {rc}")
    if want_sums2d:
        out["sums2d"] = sums
    return out
