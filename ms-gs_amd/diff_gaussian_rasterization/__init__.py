"""MI355X-native drop-in for the `diff_gaussian_rasterization` package the reference imports at
/root/reference/gaussian_renderer/__init__.py:14 (the MS-GS fork of the 3DGS rasterizer, an
un-vendored CUDA submodule: /root/reference/.gitmodules:4-6).

Same public surface:
  * GaussianRasterizationSettings — NamedTuple with the 15 fields constructed at
    gaussian_renderer/__init__.py:37-53 (the 12 upstream fields + filter_small, filter_large,
    fade_size);
  * GaussianRasterizer(raster_settings=...) — nn.Module called by keyword with the 13 kwargs of
    gaussian_renderer/__init__.py:95-107, returning the 5-tuple
    (rendered_image [3,H,W], acc_pixel_size [H,W], depth [H,W], radii [P] int32, pixel_sizes [P]);
  * rasterize_gaussians(...) / GaussianRasterizer.markVisible(positions).

Host code is Python on PyTorch-ROCm; all device work is hand-written HIP for gfx950 behind the C
ABI of include/msgs.h (ms-gs_amd/csrc, loaded with ctypes by _backend.py).  PyTorch only provides
device memory (caching allocator), streams and autograd plumbing.  There is NO fallback: importing
this package without lib/libmsgs_hip.so raises ImportError, and calling it with CPU tensors raises.
"""
import ctypes as C
import os
import threading
import weakref
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _backend as _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_raw",
           "deferred_forward", "GradAccumulator", "set_grad_accumulator"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    filter_small: bool = False
    filter_large: bool = False
    fade_size: float = 1.0


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _opt(t):
    """None and empty tensors both mean 'not provided' (upstream passes torch.Tensor([]))."""
    if t is None or t.numel() == 0:
        return None
    return t


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(dev):
    """torch.cuda.device(dev) only when a switch is needed: entering and leaving that context costs ~15 us of host time,
    twice per step on the path between the forward's host synchronisation and the backward launches, where the GPU
    has only the forward's second stage queued."""
    return _NO_SWITCH if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _bytes(n, device):
    """a byte buffer of at least n bytes.  Large requests are rounded up to one of eight sizes per octave (<= 12.5 % more): the
    trainer changes the model size by a few per cent every densification interval (train.py:253-264) and the instance count
    drifts from view to view — with exact sizes every such step asks the caching allocator for blocks it has never seen
    (hipMalloc: ~1 ms in the first iteration after every densification); with size classes the cached blocks fit again"""
    n = max(int(n), 1)
    if n >= (1 << 20):
        q = 1 << (n.bit_length() - 4)
        n = (n + q - 1) & ~(q - 1)
    return torch.empty(n, dtype=torch.uint8, device=device)


# occ_multiplier / dc_delta: this build implements their identity configuration only (ones / zeros, DESIGN SPEC M5 — every
# published MS-GS configuration; the level-selection semantics of --multi_occ / --multi_dc live in the un-vendored CUDA
# kernels).  Anything else is REFUSED instead of rendered wrongly.  A leaf tensor (the reference passes the Parameter
# itself when multi_occ is off, scene/gaussian_model.py:156-164) is checked once per (tensor object, version): one tiny
# reduction + host read on the first call after creation / densification, nothing afterwards.  A computed tensor (e.g.
# sigmoid(_occ_multiplier) under --multi_occ, :158) is checked on every call.
_identity_checked = {}


def _require_identity(t, name, value, flag):
    if t is None:
        return
    leaf = t.grad_fn is None
    if leaf:
        hit = _identity_checked.get(id(t))
        # (object, version, storage address): `param.data = other` keeps the first two
        if hit is not None and hit[0]() is t and hit[1] == t._version and hit[2] == t.data_ptr():
            return
    if not bool((t.detach() == value).all().item()):
        raise NotImplementedError(
            f"diff_gaussian_rasterization (MI355X build): {name} must be all {value:g} — the {flag} semantics of the "
            "MS-GS rasterizer (scene/gaussian_model.py:156-164,203-213) are not implemented; refusing to render "
            "with them silently ignored")
    if leaf:
        if len(_identity_checked) > 64:
            for k in [k for k, v in _identity_checked.items() if v[0]() is None]:
                del _identity_checked[k]
        _identity_checked[id(t)] = (weakref.ref(t), t._version, t.data_ptr())


def _check_rows(name, t, P, tail):
    """per-Gaussian tensor `t` must be [P, *tail] (element count per row checked; a stale tensor after densify / prune
    would otherwise be read out of bounds by preprocess_kernel)"""
    if t is None:
        return
    if tail is None:                       # width not fixed by the ABI (never read: identity-checked): rows only
        if t.dim() == 0 or int(t.shape[0]) != P:
            raise ValueError(f"{name} must have {P} rows, got {tuple(t.shape)}")
        return
    n = 1
    for d in tail:
        n *= d
    if t.dim() == 0 or int(t.shape[0]) != P or t.numel() != P * n:
        raise ValueError(f"{name} must have shape [{P}, {', '.join(str(d) for d in tail)}], got {tuple(t.shape)}")


class _Call:
    """Marshals one (settings, tensors) pair into the C structs; keeps the tensors alive."""

    def __init__(self, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                 max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta, base_mask, raw_features=None,
                 rotations_raw=None):
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("diff_gaussian_rasterization (MI355X build): tensors must live on a HIP device "
                               "('cuda'); there is no CPU path")
        self.device = dev
        P = int(means3D.shape[0])
        self.P = P
        _check_rows("means3D", means3D, P, (3,))
        _check_rows("scales", scales, P, (3,))
        _check_rows("rotations", rotations, P, (4,))
        _check_rows("cov3D_precomp", cov3D_precomp, P, (6,))
        _check_rows("colors_precomp", colors_precomp, P, (3,))
        _check_rows("occ_multiplier", occ_multiplier, P, None)
        _check_rows("dc_delta", dc_delta, P, None)
        _check_rows("rotations_raw", rotations_raw, P, (4,))
        if sh is not None and (sh.dim() != 3 or int(sh.shape[0]) != P or int(sh.shape[2]) != 3):
            raise ValueError(f"shs must have shape [{P}, K, 3], got {tuple(sh.shape)}")
        _require_identity(occ_multiplier, "occ_multiplier", 1.0, "--multi_occ")
        _require_identity(dc_delta, "dc_delta", 0.0, "--multi_dc")
        self.means3D = _f32c(means3D)
        self.sh = _f32c(sh) if sh is not None else None
        self.colors = _f32c(colors_precomp) if colors_precomp is not None else None
        self.opac = _f32c(opacities).reshape(-1)
        self.scales = _f32c(scales) if scales is not None else None
        self.rot = _f32c(rotations) if rotations is not None else None
        self.cov = _f32c(cov3D_precomp) if cov3D_precomp is not None else None
        self.maxps = _f32c(max_pixel_sizes).reshape(-1) if max_pixel_sizes is not None else None
        self.minps = _f32c(min_pixel_sizes).reshape(-1) if min_pixel_sizes is not None else None
        self.occ = _f32c(occ_multiplier) if occ_multiplier is not None else None
        self.dcd = _f32c(dc_delta) if dc_delta is not None else None
        if base_mask is None:
            self.base = None
        elif base_mask.dtype == torch.bool and base_mask.is_contiguous():
            self.base = base_mask.view(torch.uint8)            # same bytes, no copy kernel
        else:
            self.base = base_mask.to(torch.uint8).contiguous()
        for name, t, n in (("opacities", self.opac, P), ("max_pixel_sizes", self.maxps, P),
                           ("min_pixel_sizes", self.minps, P), ("base_mask", self.base, P)):
            if t is not None and t.numel() != n:
                raise ValueError(f"{name} must have {n} elements, got {tuple(t.shape)}")
        self.K = int(self.sh.shape[1]) if self.sh is not None else 0
        self.fdc = self.frest = None
        if raw_features is not None:        # raw GaussianModel parameters (msgs_gaussians_t::raw_params = 1)
            fdc, frest = raw_features
            self.fdc, self.frest = _f32c(fdc), _f32c(frest)
            if self.fdc.numel() != 3 * P or self.frest.numel() != 45 * P:
                raise ValueError("raw mode needs features_dc [P,1,3] and features_rest [P,15,3]")
            self.K = 16
        on = lambda t: _f32c(t if t.device == dev else t.to(dev))
        self.bg, self.vm, self.pm, self.cp = on(rs.bg), on(rs.viewmatrix), on(rs.projmatrix), on(rs.campos)
        self.W, self.H = int(rs.image_width), int(rs.image_height)
        self.view = _C.View(self.H, self.W, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
                            float(rs.fade_size), int(rs.sh_degree), self.K,
                            int(bool(rs.filter_small)), int(bool(rs.filter_large)), int(bool(rs.prefiltered)),
                            int(bool(rs.debug)), 0, 0, 0.0, 0, _ptr(self.bg), _ptr(self.vm), _ptr(self.pm), _ptr(self.cp))
        # mode 1: raw parameters everywhere; mode 2: activated inputs + raw quaternions, gradients chained to the raw
        # parameters inside msgs_backward (include/msgs.h, msgs_gaussians_t::raw_params)
        self.rot_raw = _f32c(rotations_raw) if rotations_raw is not None else None
        mode = 0 if raw_features is None else (2 if rotations_raw is not None else 1)
        self.g = _C.Gaussians(P, mode, _ptr(self.means3D), _ptr(self.sh),
                              _ptr(self.colors), _ptr(self.opac), _ptr(self.scales), _ptr(self.rot), _ptr(self.cov),
                              _ptr(self.maxps), _ptr(self.minps), _ptr(self.occ), _ptr(self.dcd), _ptr(self.base),
                              _ptr(self.fdc), _ptr(self.frest), _ptr(self.rot_raw))
        self.view_ref, self.g_ref = C.byref(self.view), C.byref(self.g)


def _backward_scratch(P, D, dev):
    lib = _C.lib
    if lib.msgs_get_deterministic():
        return _bytes(lib.msgs_backward_scratch_bytes_deterministic(P, D), dev)
    return _bytes(lib.msgs_backward_scratch_bytes(P), dev)


# MSGS_NO_FORWARD_CLEAR=1: allocate the records in backward as before (a trainer that keeps MANY forward graphs alive before
# running their backwards would otherwise hold 80 bytes per Gaussian per graph)
_forward_clear = os.environ.get("MSGS_NO_FORWARD_CLEAR", "0") != "1"


# grad mode of the CALLER, noted right before .apply(): inside autograd.Function.forward grad mode is always off, and
# ctx.needs_input_grad is True under torch.no_grad() whenever the parameters require grad — every eval / viewer render of a
# trained model would otherwise allocate and clear 80 bytes per Gaussian for a backward that never comes
_caller = threading.local()


def _note_grad_mode():
    _caller.grad_enabled = torch.is_grad_enabled()


def _alloc_grad_records(ctx, P, dev):
    """The backward's per-Gaussian gradient records have to start from zero.  When a backward can follow, the buffer is
    allocated HERE and handed to the forward, whose blend kernel clears it on the side (include/msgs.h, grad_records: the
    kernel is instruction-bound, the stores are free) — the backward then skips its fill launch.  Not in the verification
    mode, whose scratch is sized by the instance count."""
    ctx.grad_rec = None
    ctx.det = bool(_C.lib.msgs_get_deterministic())
    # a backward can follow this forward (msgs_forward*: backward_follows — the tile launch order is prepared)
    ctx.backward_follows = bool(P > 0 and getattr(_caller, "grad_enabled", True) and any(ctx.needs_input_grad))
    if _forward_clear and ctx.backward_follows and not _C.lib.msgs_get_deterministic():
        ctx.grad_rec = _bytes(_C.lib.msgs_backward_scratch_bytes(P), dev)
    return ctx.grad_rec


def _take_backward_scratch(ctx, P, D, dev):
    """(scratch, is_clear): the buffer the forward cleared, once; any later backward through the same graph
    (retain_graph) gets a fresh one that msgs_backward clears itself"""
    if bool(_C.lib.msgs_get_deterministic()) != bool(getattr(ctx, "det", False)):
        raise RuntimeError("diff_gaussian_rasterization: set_deterministic() changed between this forward and its backward (the "
                           "verification mode's backward reads what its own forward left behind)")
    rec = getattr(ctx, "grad_rec", None)
    ctx.grad_rec = None
    if rec is not None and not _C.lib.msgs_get_deterministic():
        return rec, 1
    return _backward_scratch(P, D, dev), 0


def set_deterministic(on=True):
    """Process-wide switch: the VERIFICATION mode (ms-gs_amd/csrc/literal.hip).  Forward and backward blend loops
    restate the reference's per-pixel arithmetic literally (float32, no FMA contraction, exp evaluated in double and rounded
    once), the nine per-(pixel, Gaussian) products are summed in double in a fixed order (in-tile tree, per-entry stores, stable
    grouping by Gaussian): bitwise reproducible by construction, and — against the CPU checker of the tests evaluated the same
    way (exp in double) — every gradient tensor within 1e-4, the north star's sentence as written
    (tests/test_literal_gpu.py).  Several times slower than the default path.  Must not change between a forward and its
    backward.  Returns the previous setting.  Also: MSGS_DETERMINISTIC=1 in the environment."""
    return bool(_C.lib.msgs_set_deterministic(1 if on else 0))


# instance-count guess per (device, P, W, H, filters): the next call sizes its binning buffers from it (+12.5 %), and
# msgs_forward launches stage 2 on them speculatively, with grids sized for that capacity — so the guess has to follow the
# scene DOWN quickly as well (an outlier, e.g. one render of the same model without its multi-scale filters, would otherwise
# leave every following call with stage-2 grids and buffers many times too large): it halves its excess over the last count
# on every call; views whose counts differ by up to ~28 % alternate without outgrowing it.
class _LRU(dict):
    """a dict that forgets its least recently WRITTEN key beyond `cap` entries (the trainer's model changes size every hundred
    iterations — densify / prune, /root/reference/train.py:253-264 — and a wholesale clear() would throw away every other key's state)"""

    def __init__(self, cap=256):
        super().__init__()
        self.cap = cap

    def __setitem__(self, k, v):
        if k in self:
            super().__delitem__(k)                  # re-insert at the end: most recent
        super().__setitem__(k, v)
        while len(self) > self.cap:
            super().__delitem__(next(iter(self)))


_last_instances = _LRU()
# ... and per (device, W, H, filters) whatever the model size: (instances, P) of the latest forward.  The reference's training loop
# changes P every densification interval (train.py:253-264: clone / split / prune, a few per cent) and the pyramid level with
# every camera stack (:152-194): a key that has never been seen still gets a guess — the last count of the same view shape, scaled
# by P — so that only the very first forward of a view shape takes the non-speculative route (bench.py reference_schedule).
_instances_by_view = _LRU()
forward_stats = {"forwards": 0, "non_speculative": 0}        # since import (bench.py reference_schedule reads the difference)


def _instance_guess(key):
    g = _last_instances.get(key)
    if g is not None:
        return g
    v = _instances_by_view.get((key[0],) + key[2:])
    if v is None:
        return None
    D, P_ref = v
    ratio = key[1] / max(P_ref, 1)
    if not 0.8 <= ratio <= 1.25:        # another model altogether (D does not scale with P across scenes): no guess
        return None
    return int(D * ratio) + 1


def _note_instances(key, D, guess):
    _last_instances[key] = max(D, (guess + D) // 2) if guess is not None else D
    _instances_by_view[(key[0],) + key[2:]] = (D, key[1])


# Occlusion cut-off pass (include/msgs.h, msgs_set_occlusion): ONE launch between the per-Gaussian stage and the depth sort that a
# view without cover candidates leaves after its first grid barrier (~5 us) — it runs on EVERY forward; the adaptive skip policy
# of round 5 (and its cliff: a closing view among the skipped calls rendered uncut) is gone.  What is left per (device, P, W, H,
# filters) key is a performance HINT for one small launch: the queue that gives every Gaussian with more than 96 tile instances
# a workgroup of its own (msgs_view_t.no_heavy_queue) pays off when covers closed blocks — the nearest ranks are then all
# giants — and is a wasted ~3 us launch otherwise; it is kept on for HEAVY_QUEUE_MEMORY calls after the key last closed a block
# (msgs_forward_info: the answer travels with the instance count) and on the key's first call.  Outputs never depend on it.
HEAVY_QUEUE_MEMORY = 32
_occ_hot = _LRU(1024)        # key -> calls for which the queue stays on: refreshed by every call of the key that closes a block


def _heavy_queue_off(key):
    return _occ_hot.get(key, 1) == 0


# Depth-slab binning (include/msgs.h, msgs_view_t.slab_fraction).  Stage 2 bins and blends the nearest depth ranks
# first and then only the tiles in which a pixel is still blending: bit-identical outputs, and on a view whose pixels terminate
# early (BASELINE C5: 54.9 M instances, 4.35 M traversed) emit / tile sort / ranges shrink to a fraction.  On a view whose
# pixels walk most of their lists (C3: 44 %) it would only add launches, so the wrapper engages it per (device, P, W, H, filters)
# key from what the library publishes about the key's EARLIER frames (msgs_view_t.feedback_tag -> msgs_forward_info: instances D,
# traversed entries D_trav; one frame late, no synchronisation):
#   slab_policy "adaptive" (default)   on while D >= SLAB_MIN_INSTANCES and D >= SLAB_MIN_RATIO * D_trav (else the key asks for a
#                                      publication on every SLAB_RECHECK-th call only), with
#                                      slab_fraction = 2 * D_trav / D clamped to [0.04, 0.30]; off for SLAB_BACKOFF calls of the key
#                                      when a slab frame emitted more than 70 % of D anyway (the view changed)
#   "never" / a float                  never / always with that fraction (tests, A/B runs)
# Either way the result is exact; the worst case of a wrong guess is ~15 % of a binning pass.
slab_policy = os.environ.get("MSGS_SLAB_POLICY", "adaptive")
SLAB_MIN_INSTANCES = 2_000_000
SLAB_MIN_RATIO = 6.0
SLAB_BACKOFF = 64
SLAB_RECHECK = 16
_fb_stats = _LRU(1024)       # key -> {"D", "D_trav", "backoff", ...}: the latest publication of the key
_fb_tag_of = {}              # key -> tag (1 .. 2^31 - 1)
_fb_key_of = {}              # tag -> key
_fb_next_tag = [0]


def _slab_plan(key, guess, tiles):
    """(slab_fraction, feedback_tag) for the next forward of `key`.  The statistics are kept per VIEW SHAPE (device, W, H,
    filters), not per model size: D / D_trav is a property of the scene and the view, and the trainer changes P by a few per cent
    every densification interval"""
    key = (key[0],) + key[2:]
    pol = slab_policy
    if pol == "never":
        return 0.0, 0
    if pol != "adaptive":
        return float(pol), 0
    if tiles < 2048 or guess is None or guess < SLAB_MIN_INSTANCES // 2:
        return 0.0, 0                                   # small frames: no feedback launch either
    tag = _fb_tag_of.get(key)
    if tag is None:
        if len(_fb_tag_of) > 1024:
            _fb_tag_of.clear(); _fb_key_of.clear(); _fb_stats.clear()
        _fb_next_tag[0] = _fb_next_tag[0] % 0x7FFFFFF0 + 1        # never reused while an old publication may still sit in a status block
        tag = _fb_tag_of[key] = _fb_next_tag[0]
        _fb_key_of[tag] = key
    st = _fb_stats.get(key)
    if st is None:
        return 0.0, tag
    if st["D"] < SLAB_MIN_INSTANCES or st["D"] < SLAB_MIN_RATIO * max(st["D_trav"], 1):
        # not a view for slabs (BASELINE C3: D = 2.3 x D_trav): look again every SLAB_RECHECK-th call only — the publication is
        # one small kernel with a system-scope store, ~4 us per forward
        st["idle"] = (st.get("idle", 0) + 1) % SLAB_RECHECK
        return 0.0, (tag if st["idle"] == 0 else 0)
    if st.get("backoff", 0) > 0:
        st["backoff"] -= 1
        return 0.0, tag
    return min(0.30, max(0.04, 2.0 * st["D_trav"] / st["D"])), tag


def _note_info(key):
    """after the instance count of a forward has been collected on this thread: what travelled with it"""
    info = (C.c_int64 * 8)()
    _C.lib.msgs_forward_info(info)
    tag = int(info[2])
    if tag:
        k = _fb_key_of.get(tag)
        if k is not None:
            st = _fb_stats.setdefault(k, {})
            st["D"], st["D_trav"] = int(info[3]), int(info[4])
            if info[5] >= 0:                            # that frame ran in slab mode: did it pay?
                st["n_open"], st["DA"], st["DB"] = int(info[5]), int(info[6]), int(info[7])
                if info[7] < 0:
                    raise RuntimeError(f"diff_gaussian_rasterization: slab B outgrew its buffers (SlabHeader overflow flag) "
                                       f"[view {k}: D {int(info[3])}, D_trav {int(info[4])}, open tiles {int(info[5])}, slab A {int(info[6])}]")
                if int(info[6]) + int(info[7]) > 0.7 * max(int(info[3]), 1):
                    st["backoff"] = SLAB_BACKOFF
    _occ_hot[key] = HEAVY_QUEUE_MEMORY if info[1] else max(_occ_hot.get(key, 1) - 1, 0)


_size_cache = _LRU()


def _sizes(P, W, H):
    """(geom, stage-1 scratch, image) byte counts per (P, W, H): three ctypes calls saved per forward"""
    k = (P, W, H)
    v = _size_cache.get(k)
    if v is None:
        lib = _C.lib
        v = _size_cache[k] = (int(lib.msgs_geom_bytes(P)), int(lib.msgs_stage1_scratch_bytes(P)), int(lib.msgs_image_bytes(W, H)))
    return v


def _a256(n):
    return (int(n) + 255) & ~255


def _stage2_bytes(D, W, H, frac):
    """(binning, stage-2 scratch) byte counts that serve D instances — with room for slab A's ids in front of slab B's when the call
    asks for depth slabs"""
    lib = _C.lib
    if frac > 0.0:
        return int(lib.msgs_binning_bytes_slab(D, W, H, frac)), int(lib.msgs_stage2_scratch_bytes_slab(D, W, H))
    return int(lib.msgs_binning_bytes(D, W, H)), int(lib.msgs_stage2_scratch_bytes(D, W, H))


# ---- deferred forwards: several views in flight from one host thread (include/msgs.h msgs_forward_launch / _finish) ----
# Inside `with deferred_forward() as pending:` every forward of this package only LAUNCHES (stage 1 and, on buffers sized
# from the instance-count guess, stage 2) and returns its output tensors at once — valid in stream order, like any
# asynchronous kernel result — without waiting for the instance count.  The wait happens in _PendingForward.resolve():
# when the view's backward starts, when the caller resolves the entries of `pending`, or at the end of the `with` block,
# whichever comes first.  If a scene outgrew its guess, resolve() redoes stage 2 on exact buffers on the view's stream
# (the outputs are overwritten in place, stream-ordered behind the truncated result).  A consumer that reads an output
# on ANOTHER stream, or on the host, must resolve the view first.  host/multi_view.py builds the two-stream pipelines
# on top of this.
_deferred = threading.local()

_status_pool = []
_status_lock = threading.Lock()


def _take_status():
    with _status_lock:
        if _status_pool:
            return _status_pool.pop()
    h = C.c_void_p()
    _C.check(_C.lib.msgs_status_create(C.byref(h)), "msgs_status_create")
    return h


def _give_status(h):
    with _status_lock:
        _status_pool.append(h)


class deferred_forward:
    """Context manager; yields the list the forwards' _PendingForward objects are appended to (in call order)."""

    def __enter__(self):
        self.prev = getattr(_deferred, "pending", None)
        _deferred.pending = self.list = []
        return self.list

    def __exit__(self, *exc):
        _deferred.pending = self.prev
        first = None
        for p_ in self.list:                  # every launched view is waited for, also behind one that failed
            if exc[0] is None and first is None:
                try:
                    p_.resolve()
                except Exception as e:        # noqa: BLE001 - re-raised below, after the other handles are finished
                    first = e
            else:
                p_.abandon()
        if first is not None:
            raise first
        return False


class _PendingForward:
    """State of one launched forward: resolve() -> (geom, binning, image, D), waiting for the count if nobody has yet."""

    def __init__(self, call, status, stream, key, guess, geom, binning, image, outs, grad_rec, keep, backward_follows=False,
                 scratch1=None):
        self.backward_follows = bool(backward_follows)
        self.scratch1 = scratch1              # holds the device status words msgs_forward_finish may still copy from
        self.call, self.status, self.stream, self.key, self.guess = call, status, stream, key, guess
        self.geom, self.binning, self.image, self.outs, self.grad_rec, self.keep = geom, binning, image, outs, grad_rec, keep
        self.state = self.error = None
        self.lock = threading.Lock()

    def resolve(self):
        with self.lock:
            if self.state is not None:
                return self.state
            if self.status is None:                     # an earlier resolve() failed: the view has no result
                raise RuntimeError("this forward failed when its instance count was collected") from self.error
            lib, call = _C.lib, self.call
            D, done = C.c_int64(0), C.c_int32(0)
            status, self.status = self.status, None
            try:
                _C.check(lib.msgs_forward_finish(status, C.byref(D), C.byref(done)), "msgs_forward_finish")
            except Exception as e:
                self.error = e
                raise
            finally:
                _give_status(status)
                self.scratch1 = None
            D = int(D.value)
            guess = self.guess
            _note_instances(self.key, D, guess)
            _note_info(self.key)
            forward_stats["forwards"] += 1
            if not done.value:                          # first frame of this shape, or the scene grew past the margin
                forward_stats["non_speculative"] += 1
                dev, W, H = call.device, call.W, call.H
                color, acc_ps, depth = self.outs
                self.error = RuntimeError("stage 2 on exact buffers failed")     # cleared below
                with _on_device(dev), torch.cuda.stream(self.stream):
                    nb_, ns_ = _stage2_bytes(D, W, H, float(call.view.slab_fraction))
                    self.binning = _bytes(nb_, dev)
                    scratch2 = _bytes(ns_, dev)
                    grad_rec = self.grad_rec
                    _C.check(lib.msgs_forward_stage2(call.view_ref, call.g_ref, _ptr(self.geom), self.geom.numel(), D,
                                                     _ptr(self.binning), self.binning.numel(), _ptr(scratch2),
                                                     scratch2.numel(), _ptr(self.image), self.image.numel(), _ptr(color),
                                                     _ptr(acc_ps), _ptr(depth), _ptr(grad_rec),
                                                     grad_rec.numel() if grad_rec is not None else 0,
                                                     int(self.backward_follows), _C.timer_ptr(),
                                                     C.c_void_p(self.stream.cuda_stream)), "msgs_forward_stage2")
                    del scratch2
            self.state = (self.geom, self.binning, self.image, D)
            self.outs = self.grad_rec = self.error = None
            return self.state

    def abandon(self):
        """the caller's block raised: finish the wait (the handle must not be reused while a kernel can still write its
        status words) and swallow secondary errors"""
        try:
            self.resolve()
        except Exception:
            pass


def _resolve(state):
    return state.resolve() if isinstance(state, _PendingForward) else state


def _forward_impl(call, grad_rec=None, backward_follows=False):
    dev, P, W, H = call.device, call.P, call.W, call.H
    lib = _C.lib
    key = (dev.index, P, W, H, call.view.filter_small, call.view.filter_large)
    pending = getattr(_deferred, "pending", None)
    call.view.no_heavy_queue = int(_heavy_queue_off(key))
    frac, tag = _slab_plan(key, _instance_guess(key), ((W + 15) // 16) * ((H + 15) // 16))
    if frac > 0.0 and lib.msgs_get_deterministic():
        frac = 0.0
    call.view.slab_fraction, call.view.feedback_tag = frac, tag
    with _on_device(dev):
        cur = torch.cuda.current_stream(dev)
        stream = C.c_void_p(cur.cuda_stream)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        pixel_sizes = torch.empty(P, dtype=torch.float32, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        acc_ps = torch.empty(H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(H, W, dtype=torch.float32, device=dev)
        n_geom, n_s1, n_img = _sizes(P, W, H)
        guess = _instance_guess(key)
        # two allocations instead of five: what the backward needs again (geom | image | binning) and what dies with the
        # forward (stage-1 scratch | stage-2 scratch); every part starts on a 256-byte boundary
        n_bin = n_s2 = 0
        if guess is not None:
            cap = guess + (guess >> 3) + 4096
            n_bin, n_s2 = _stage2_bytes(cap, W, H, frac)
        keep = _bytes(_a256(n_geom) + _a256(n_img) + n_bin, dev)
        geom, image = keep[:n_geom], keep[_a256(n_geom):_a256(n_geom) + n_img]
        binning = keep[_a256(n_geom) + _a256(n_img):] if n_bin else None
        if pending is not None:
            # launch only.  The stage-2 temporaries go back to the caching allocator at once, which hands them out again in
            # THIS stream's order only (blocks are bound to the stream they were allocated on).  The stage-1 scratch holds the
            # device copy of the status words that msgs_forward_finish may still READ at resolve time (MSGS_BLOCKING_SYNC=1, a
            # failed pinned allocation, or a flag that never landed: a copy enqueued behind whatever this stream was given
            # since) — it stays referenced by the pending state until resolve()
            scratch1 = _bytes(n_s1, dev)
            scratch2 = _bytes(n_s2, dev) if n_s2 else None
            status = _take_status()
            try:
                _C.check(lib.msgs_forward_launch(call.view_ref, call.g_ref, _ptr(radii), _ptr(pixel_sizes),
                                                 _ptr(geom), n_geom, _ptr(scratch1), n_s1,
                                                 _ptr(binning), n_bin, _ptr(scratch2), n_s2,
                                                 _ptr(image), n_img, _ptr(color), _ptr(acc_ps), _ptr(depth),
                                                 _ptr(grad_rec), grad_rec.numel() if grad_rec is not None else 0,
                                                 int(backward_follows), status, _C.timer_ptr(), stream), "msgs_forward_launch")
            except Exception:
                _give_status(status)
                raise
            state = _PendingForward(call, status, cur, key, guess, geom, binning, image, (color, acc_ps, depth), grad_rec, keep,
                                    backward_follows, scratch1)
            pending.append(state)
            return color, acc_ps, depth, radii, pixel_sizes, state
        tmp = _bytes(_a256(n_s1) + n_s2, dev)
        scratch1 = tmp[:n_s1]
        scratch2 = tmp[_a256(n_s1):] if n_s2 else None
        D, done = C.c_int64(0), C.c_int32(0)
        _C.check(lib.msgs_forward(call.view_ref, call.g_ref, _ptr(radii), _ptr(pixel_sizes),
                                  _ptr(geom), n_geom, _ptr(scratch1), n_s1,
                                  _ptr(binning), n_bin, _ptr(scratch2), n_s2,
                                  _ptr(image), n_img, _ptr(color), _ptr(acc_ps), _ptr(depth),
                                  _ptr(grad_rec), grad_rec.numel() if grad_rec is not None else 0, int(backward_follows),
                                  C.byref(D), C.byref(done), _C.timer_ptr(), stream), "msgs_forward")
        D = int(D.value)
        _note_instances(key, D, guess)
        _note_info(key)
        del scratch1, scratch2, tmp
        forward_stats["forwards"] += 1
        if not done.value:                              # first frame of this shape, or the scene grew past the margin
            forward_stats["non_speculative"] += 1
            nb_, ns_ = _stage2_bytes(D, W, H, frac)
            binning = _bytes(nb_, dev)
            scratch2 = _bytes(ns_, dev)
            _C.check(lib.msgs_forward_stage2(call.view_ref, call.g_ref, _ptr(geom), n_geom, D,
                                             _ptr(binning), binning.numel(), _ptr(scratch2), scratch2.numel(),
                                             _ptr(image), n_img, _ptr(color), _ptr(acc_ps), _ptr(depth),
                                             _ptr(grad_rec), grad_rec.numel() if grad_rec is not None else 0,
                                             int(backward_follows), _C.timer_ptr(), stream), "msgs_forward_stage2")
    return color, acc_ps, depth, radii, pixel_sizes, (geom, binning, image, D)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta, base_mask, raster_settings):
        if means3D.shape[0] == 0:
            # nothing to rasterize: background image, no native call (every per-Gaussian tensor is empty,
            # which upstream's convention cannot tell apart from "not provided")
            rs = raster_settings
            dev = means3D.device
            if dev.type != "cuda":
                raise RuntimeError("diff_gaussian_rasterization (MI355X build): tensors must live on a HIP device")
            H, W = int(rs.image_height), int(rs.image_width)
            color = rs.bg.to(dev, torch.float32).view(3, 1, 1).expand(3, H, W).contiguous()
            ctx.empty = True
            ctx.in_shapes = [t.shape for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                               cov3Ds_precomp)]
            ctx.dev = dev
            outs = (color, torch.zeros(H, W, device=dev), torch.zeros(H, W, device=dev),
                    torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, device=dev))
            ctx.mark_non_differentiable(*outs[1:])
            ctx.set_materialize_grads(False)
            return outs
        ctx.empty = False
        call = _Call(raster_settings, means3D, _opt(sh), _opt(colors_precomp), opacities, _opt(scales),
                     _opt(rotations), _opt(cov3Ds_precomp), _opt(max_pixel_sizes), _opt(min_pixel_sizes),
                     _opt(occ_multiplier), _opt(dc_delta), _opt(base_mask))
        color, acc_ps, depth, radii, pixel_sizes, state = _forward_impl(call, _alloc_grad_records(ctx, call.P, call.device), ctx.backward_follows)
        ctx.call = call
        ctx.state = state
        ctx.radii = radii
        ctx.shapes = (means2D.shape, opacities.shape)
        _save_inputs(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        ctx.mark_non_differentiable(acc_ps, depth, radii, pixel_sizes)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the four non-differentiable outputs
        return color, acc_ps, depth, radii, pixel_sizes

    @staticmethod
    def backward(ctx, grad_color, grad_acc_ps, grad_depth, grad_radii, grad_pixel_sizes):
        if ctx.empty:
            return tuple(torch.zeros(s, device=ctx.dev) for s in ctx.in_shapes) + (None,) * 6
        _check_saved(ctx)
        call = ctx.call
        if grad_color is None:
            grad_color = torch.zeros(3, call.H, call.W, dtype=torch.float32, device=call.device)
        geom, binning, image, D = _resolve(ctx.state)
        dev, P, K = call.device, call.P, call.K
        lib = _C.lib
        with _on_device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            dL = _f32c(grad_color)
            g_means3D = torch.empty(P, 3, dtype=torch.float32, device=dev)
            g_means2D = torch.empty(P, 3, dtype=torch.float32, device=dev)
            g_opac = torch.empty(P, dtype=torch.float32, device=dev)
            g_sh = torch.empty(P, K, 3, dtype=torch.float32, device=dev) if call.sh is not None else None
            g_col = torch.empty(P, 3, dtype=torch.float32, device=dev) if call.colors is not None else None
            g_scales = torch.empty(P, 3, dtype=torch.float32, device=dev) if call.scales is not None else None
            g_rot = torch.empty(P, 4, dtype=torch.float32, device=dev) if call.rot is not None else None
            g_cov = torch.empty(P, 6, dtype=torch.float32, device=dev) if call.cov is not None else None
            scratch, is_clear = _take_backward_scratch(ctx, P, D, dev)
            grads = _C.Grads(_ptr(g_means3D), _ptr(g_means2D), _ptr(g_sh), _ptr(g_col), _ptr(g_opac),
                             _ptr(g_scales), _ptr(g_rot), _ptr(g_cov), None, None, None, is_clear)
            _C.check(lib.msgs_backward(call.view_ref, call.g_ref, _ptr(ctx.radii), _ptr(geom),
                                       geom.numel(), D, _ptr(binning), binning.numel(), _ptr(image),
                                       image.numel(), _ptr(dL), _ptr(scratch), scratch.numel(), C.byref(grads),
                                       _C.timer_ptr(), stream), "msgs_backward")
        m2_shape, op_shape = ctx.shapes
        # occ_multiplier / dc_delta / pixel-size inputs / masks receive no gradient (DESIGN.md SPEC M5)
        return (g_means3D, g_means2D.view(m2_shape) if g_means2D.shape == m2_shape else g_means2D,
                g_sh, g_col, g_opac.view(op_shape), g_scales, g_rot, g_cov,
                None, None, None, None, None, None)


# Gradient sinks (view-parallel training): a trainer that exchanges gradients through one flat bucket registers, per leaf
# parameter, the slice of the bucket its gradient belongs in.  The backward of the raw / chained entries then writes the
# gradient THERE and returns a fresh alias of that slice; with param.grad = None autograd adopts the alias as .grad (no
# copy), so neither a zero-fill of the bucket nor an accumulation pass over it is needed.  Keys: id() of the leaf tensor
# OBJECT, validated through a weak reference (an address-based key could match an unrelated tensor after densification
# re-allocates the parameters).
class _SinkRegistry(threading.local):
    """per host thread: set_grad_sinks and the forward that snapshots the sinks run on the training loop's thread; a second
    thread of the process (a viewer, an evaluation loop, another trainer) neither sees nor disturbs them"""

    def __init__(self):
        self.grad = {}
        self.sh_factor = [None, None]   # [destination tensor, optional torch.cuda.Event recorded once the factors are written]


_sinks = _SinkRegistry()


def _save_inputs(ctx, *tensors):
    """Everything the backward re-reads goes through save_for_backward, so that an in-place edit between forward and
    backward (optimizer step, reset_opacity-style edit) raises autograd's version-counter error instead of yielding
    silently wrong gradients.  (The ctypes structs on ctx.call point at the same storage.)"""
    ctx.save_for_backward(*[t for t in tensors if torch.is_tensor(t) and t.numel() > 0])


def _check_saved(ctx):
    if not getattr(ctx, "empty", False):
        ctx.saved_tensors              # raises "modified by an inplace operation" when a version changed


def set_grad_sinks(mapping, sh_factor=None, factors_ready=None):
    """mapping: {leaf parameter: destination tensor (float32, contiguous, same numel)} or None to clear.
    sh_factor: optional [P,3] float32 destination.  When given, the backward of the raw / chained entries does NOT form
    the 48-float SH gradient rows: it writes this view's clamp-masked dL/drgb there (the SH gradient is the outer product
    basis(direction) x dL/drgb, rebuilt by sh_grad_from_views after the ranks have exchanged the factors) and returns
    None for features_dc / features_rest.  factors_ready: optional torch.cuda.Event (created BEFORE the call so that its
    handle exists); the library records it on the backward's stream right behind the kernel that writes the factors, ahead
    of the per-Gaussian backward — a side stream that waits on it can start exchanging the factors while that kernel runs."""
    _grad_sinks, _sh_factor_sink = _sinks.grad, _sinks.sh_factor
    _grad_sinks.clear()
    _sh_factor_sink[0] = None
    _sh_factor_sink[1] = None
    if sh_factor is not None:
        if sh_factor.dtype != torch.float32 or not sh_factor.is_contiguous() or sh_factor.dim() != 2 or sh_factor.shape[1] != 3:
            raise ValueError("sh_factor sink must be a contiguous float32 [P,3] tensor")
        _sh_factor_sink[0] = sh_factor
        _sh_factor_sink[1] = factors_ready
    if mapping:
        for leaf, dest in mapping.items():
            if dest.dtype != torch.float32 or not dest.is_contiguous() or dest.numel() != leaf.numel():
                raise ValueError("grad sink must be a contiguous float32 tensor with the parameter's numel")
            if dest.data_ptr() % 16:
                raise ValueError("grad sink must be 16-byte aligned (the backward stores float4 rows)")
            _grad_sinks[id(leaf)] = (weakref.ref(leaf), dest)


class GradAccumulator:
    """One gradient bucket for ALL views of an optimizer step (include/msgs.h, msgs_grads_t::accumulate).

    While installed (set_grad_accumulator), the backward of the raw / chained entries writes the leaf gradients straight into
    this object's tensors — the first view of a step stores every row, the later ones ADD their rendered rows inside the
    per-Gaussian kernel — and returns None for the leaves, so autograd runs no `param.grad += g` passes (six kernels and
    3 x 236 bytes per Gaussian and view at SH degree 3) and the kernel writes no zero rows for Gaussians a view did not
    render.  The sums are formed in the order of the backward calls, like autograd's: same bits.  Views may run on
    different streams: each backward waits for the previous one's kernel through an event.  finish() hands the bucket to
    the parameters' .grad (call it on the stream that will consume the gradients)."""

    def __init__(self, leaves, dest=None):
        """dest: optional destination tensors, one per leaf (e.g. the slices of a flat exchange bucket): used for every
        step instead of fresh allocations; float32, contiguous, 16-byte aligned, the leaf's numel."""
        self.leaves = list(leaves)
        for t in self.leaves:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
                raise ValueError("GradAccumulator: leaves must be contiguous float32 tensors on a HIP device")
        self.fixed = None
        if dest is not None:
            dest = list(dest)
            if len(dest) != len(self.leaves):
                raise ValueError("GradAccumulator: one destination per leaf")
            for t, d in zip(self.leaves, dest):
                if d.dtype != torch.float32 or not d.is_contiguous() or d.numel() != t.numel() or d.device != t.device \
                        or d.data_ptr() % 16:
                    raise ValueError("GradAccumulator: a destination must be a contiguous, 16-byte aligned float32 tensor "
                                     "with its leaf's numel on the leaf's device")
            self.fixed = [d.view(t.shape) for t, d in zip(self.leaves, dest)]
        self.dest = None
        self.count = 0
        self.events = [torch.cuda.Event(), torch.cuda.Event()]
        for e in self.events:                       # torch creates the HIP event on the first record()
            e.record(torch.cuda.current_stream(self.leaves[0].device))
        self.last = None
        self.lock = threading.Lock()

    def begin_step(self):
        """fresh (uninitialised) tensors — or the caller's destinations: the first backward of the step writes every row"""
        self.dest = self.fixed if self.fixed is not None else [torch.empty_like(t) for t in self.leaves]
        self.count = 0
        self.last = None

    def _take(self, leaves):
        """(destinations in the order of `leaves`, accumulate flag, event to wait for or None, event to record)"""
        with self.lock:
            if self.dest is None:
                self.begin_step()
            by_id = {id(t): d for t, d in zip(self.leaves, self.dest)}
            try:
                dests = [by_id[id(t)] for t in leaves]
            except KeyError:
                raise ValueError("GradAccumulator: the call's parameters are not the tensors it was built for "
                                 "(densification re-creates them: build a new accumulator)") from None
            acc, wait = self.count > 0, self.last
            rec = self.events[self.count & 1]
            self.count += 1
            self.last = rec
            return dests, acc, wait, rec

    def finish(self):
        """param.grad = bucket (or += when a gradient is already there); the current stream waits for the last view"""
        if self.dest is None or self.count == 0:
            return
        if self.last is not None:
            torch.cuda.current_stream(self.leaves[0].device).wait_event(self.last)
        for t, d in zip(self.leaves, self.dest):
            if t.grad is None or t.grad.data_ptr() == d.data_ptr():
                t.grad = d
            else:
                t.grad += d
        self.dest = None
        self.count = 0
        self.last = None


_accumulator = threading.local()


def set_grad_accumulator(acc):
    """Install (or, with None, remove) the GradAccumulator the following forwards of the raw / chained entries OF THIS THREAD
    snapshot (the forward runs on the caller's thread; the backward finds the accumulator on its ctx).  Returns the previous one."""
    prev = getattr(_accumulator, "acc", None)
    _accumulator.acc = acc
    return prev


_step_in_backward = threading.local()


def set_optimizer_in_backward(optimizer):
    """Install (or, with None, remove) the optimizer whose step the following forwards of the RAW entry OF THIS THREAD hand to
    their backward (include/msgs.h, msgs_adam_in_backward_t): the per-Gaussian backward kernel then applies the Adam update to
    the six leaf tensors and their moments itself, where it forms their gradients — the 236 bytes of gradient per Gaussian are
    neither written nor read back, the leaves receive no .grad — and `optimizer.take_step_in_backward(leaves)` is called once
    per backward to deliver the table, `optimizer.commit_step_in_backward(leaves)` behind the launch to advance the step
    counters (train_epilogue.FusedAdam implements both).  One view per
    optimizer step, as in the reference's loop (/root/reference/train.py:203-216, :416-418); not with a GradAccumulator or the
    factored SH gradient.  Bit-identical parameters and moments to backward + FusedAdam.step() (tests/test_train_step_gpu.py).
    Returns the previous one."""
    prev = getattr(_step_in_backward, "opt", None)
    _step_in_backward.opt = optimizer
    return prev


def _snapshot_sinks(ctx, leaves):
    """Called in forward (the caller's thread): the sinks registered for THIS call travel on its ctx, so that the
    backward — which runs on autograd's worker thread — never reads module state another thread may be changing."""
    ctx.sinks = tuple(_sinks.grad.get(id(t)) for t in leaves)
    ctx.sh_factor = (_sinks.sh_factor[0], _sinks.sh_factor[1])
    ctx.accum = getattr(_accumulator, "acc", None)
    ctx.step_opt = getattr(_step_in_backward, "opt", None)
    ctx.leaves = leaves if (ctx.accum is not None or ctx.step_opt is not None) else None


def _grad_out(hit, shape, dev):
    if hit is not None and hit[0]() is not None and hit[1].device == dev:
        return hit[1].view(shape)                   # a NEW alias: autograd may adopt it as param.grad
    return torch.empty(shape, dtype=torch.float32, device=dev)


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Opt-in fused entry: takes the RAW GaussianModel parameters
    (/root/reference/scene/gaussian_model.py:53-58: _xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation);
    exp / sigmoid / normalize (:39-47) and the dc|rest concatenation (:144-149) run inside the HIP kernels and the
    backward returns gradients w.r.t. the raw parameters.  Same five outputs as the reference op."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta, base_mask, raster_settings):
        if xyz.shape[0] == 0:
            raise ValueError("rasterize_gaussians_raw: empty model")
        call = _Call(raster_settings, xyz, None, None, opacity_raw, scaling_raw, rotation_raw, None,
                     _opt(max_pixel_sizes), _opt(min_pixel_sizes), _opt(occ_multiplier), _opt(dc_delta),
                     _opt(base_mask), raw_features=(features_dc, features_rest))
        color, acc_ps, depth, radii, pixel_sizes, state = _forward_impl(call, _alloc_grad_records(ctx, call.P, call.device), ctx.backward_follows)
        ctx.call, ctx.state, ctx.radii = call, state, radii
        ctx.shapes = (means2D.shape, features_dc.shape, features_rest.shape, opacity_raw.shape)
        _snapshot_sinks(ctx, (xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw))
        _save_inputs(ctx, xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw)
        ctx.mark_non_differentiable(acc_ps, depth, radii, pixel_sizes)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the four non-differentiable outputs
        return color, acc_ps, depth, radii, pixel_sizes

    @staticmethod
    def backward(ctx, grad_color, grad_acc_ps, grad_depth, grad_radii, grad_pixel_sizes):
        _check_saved(ctx)
        call = ctx.call
        if grad_color is None:
            grad_color = torch.zeros(3, call.H, call.W, dtype=torch.float32, device=call.device)
        geom, binning, image, D = _resolve(ctx.state)
        dev, P = call.device, call.P
        lib = _C.lib
        m2_shape, dc_shape, rest_shape, op_shape = ctx.shapes
        with _on_device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            dL = _f32c(grad_color)
            kx, kdc, krest, kop, ksc, krot = ctx.sinks
            g_m2 = torch.empty(P, 3, dtype=torch.float32, device=dev)
            factor, ready = ctx.sh_factor
            if factor is not None and (factor.device != dev or factor.shape[0] != P):
                raise ValueError("sh_factor sink does not match this model (device / number of Gaussians)")
            accum = getattr(ctx, "accum", None)
            step_opt = getattr(ctx, "step_opt", None)
            acc_flag, ev_wait, ev_rec = 0, None, None
            adam = None
            if step_opt is not None:
                if accum is not None or factor is not None:
                    raise ValueError("set_optimizer_in_backward cannot be combined with a GradAccumulator or the factored SH gradient")
                if call.view.sh_coeffs != 16:
                    raise ValueError("set_optimizer_in_backward needs 16 SH coefficients per Gaussian")
                adam = step_opt.take_step_in_backward(ctx.leaves)        # (a _C.AdamInBackward; keeps its tensors alive itself)
                g_xyz = g_dc = g_rest = g_opac = g_scal = g_rot = None
            elif accum is not None:
                if factor is not None:
                    raise ValueError("GradAccumulator and the factored SH gradient cannot be combined")
                (g_xyz, g_dc, g_rest, g_opac, g_scal, g_rot), acc, wait, rec = accum._take(ctx.leaves)
                acc_flag = 1 if acc else 0
                ev_wait = C.c_void_p(wait.cuda_event) if wait is not None else None
                ev_rec = C.c_void_p(rec.cuda_event)
            else:
                g_xyz = _grad_out(kx, (P, 3), dev)
                g_dc = g_rest = None
                if factor is None:
                    g_dc, g_rest = _grad_out(kdc, dc_shape, dev), _grad_out(krest, rest_shape, dev)
                g_opac, g_scal, g_rot = _grad_out(kop, op_shape, dev), _grad_out(ksc, (P, 3), dev), _grad_out(krot, (P, 4), dev)
            scratch, is_clear = _take_backward_scratch(ctx, P, D, dev)
            grads = _C.Grads(_ptr(g_xyz), _ptr(g_m2), None, _ptr(factor), _ptr(g_opac), _ptr(g_scal), _ptr(g_rot), None,
                             _ptr(g_dc), _ptr(g_rest),
                             C.c_void_p(ready.cuda_event) if (factor is not None and ready is not None) else None,
                             is_clear, acc_flag, ev_wait, ev_rec, C.addressof(adam) if adam is not None else None)
            _C.check(lib.msgs_backward(call.view_ref, call.g_ref, _ptr(ctx.radii), _ptr(geom),
                                       geom.numel(), D, _ptr(binning), binning.numel(), _ptr(image),
                                       image.numel(), _ptr(dL), _ptr(scratch), scratch.numel(), C.byref(grads),
                                       _C.timer_ptr(), stream), "msgs_backward")
            if adam is not None:
                step_opt.commit_step_in_backward(ctx.leaves)
        if accum is not None or adam is not None:   # the leaf gradients live in the accumulator / were consumed by the step
            return (None, g_m2.view(m2_shape), None, None, None, None, None, None, None, None, None, None, None)
        return (g_xyz, g_m2.view(m2_shape), g_dc, g_rest, g_opac, g_scal, g_rot, None, None, None, None, None, None)


class _RasterizeGaussiansChained(torch.autograd.Function):
    """The reference-API call whose inputs are recognised as the reference's own getters applied to leaf parameters
    (see _match_reference_getters).  Forward: exactly the reference-API forward on the activated tensors (bit-identical
    outputs).  Backward: msgs_backward applies the chain rule of sigmoid / exp / normalize / cat itself and the gradients
    go straight to the leaf parameters, instead of autograd running the getters' backward kernels afterwards (at 1 M
    Gaussians: two strided 190 MB copies for the cat and ~15 small kernels)."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                shs, opacities, scales, rotations, max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta, base_mask,
                raster_settings):
        call = _Call(raster_settings, xyz, _opt(shs), None, opacities, scales, rotations, None,
                     _opt(max_pixel_sizes), _opt(min_pixel_sizes), _opt(occ_multiplier), _opt(dc_delta),
                     _opt(base_mask), raw_features=(features_dc, features_rest), rotations_raw=rotation_raw)
        color, acc_ps, depth, radii, pixel_sizes, state = _forward_impl(call, _alloc_grad_records(ctx, call.P, call.device), ctx.backward_follows)
        ctx.call, ctx.state, ctx.radii = call, state, radii
        ctx.shapes = (means2D.shape, features_dc.shape, features_rest.shape, opacity_raw.shape)
        _snapshot_sinks(ctx, (xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw))
        _save_inputs(ctx, xyz, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw, shs, opacities,
                     scales, rotations)
        ctx.mark_non_differentiable(acc_ps, depth, radii, pixel_sizes)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the four non-differentiable outputs
        return color, acc_ps, depth, radii, pixel_sizes

    @staticmethod
    def backward(ctx, grad_color, grad_acc_ps, grad_depth, grad_radii, grad_pixel_sizes):
        return _RasterizeGaussiansRaw.backward(ctx, grad_color, grad_acc_ps, grad_depth, grad_radii,
                                               grad_pixel_sizes)[:7] + (None,) * 10


# Recognition of the reference's getters (scene/gaussian_model.py:127-153) in the autograd graph of the arguments of
# GaussianRasterizer.forward.  Module-level switch; MSGS_NO_GETTER_CHAIN=1 in the environment turns it off.
chain_reference_getters = os.environ.get("MSGS_NO_GETTER_CHAIN", "0") != "1"
# the chained kernels read the SH rows from the concatenated tensor the reference built (aligned rows, visible ones
# only: K1 111 -> 89 us, K9 141 -> 126 us at C3) at the price of keeping that [P,16,3] tensor alive until backward
_chain_reads_cat = os.environ.get("MSGS_CHAIN_SPLIT_READS", "0") != "1"


_warned_chain = [False]


def _leaf(fn):
    return fn.variable if fn is not None and type(fn).__name__ == "AccumulateGrad" else None


def _match_reference_getters(means3D, sh, opacities, scales, rotations):
    """(features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw) when the four tensors are, provably from
    their grad_fn chains,  cat((dc, rest), dim=1), sigmoid(leaf), exp(leaf) and F.normalize(leaf)  of float32 CUDA leaf
    tensors with the reference's shapes; None otherwise (then autograd handles everything as usual)."""
    try:
        P = means3D.shape[0]
        f = sh.grad_fn
        if type(f).__name__ != "CatBackward0" or f._saved_dim != 1 or len(f.next_functions) != 2:
            return None
        dc, rest = _leaf(f.next_functions[0][0]), _leaf(f.next_functions[1][0])
        f = opacities.grad_fn
        if type(f).__name__ != "SigmoidBackward0":
            return None
        op = _leaf(f.next_functions[0][0])
        f = scales.grad_fn
        if type(f).__name__ != "ExpBackward0":
            return None
        sc = _leaf(f.next_functions[0][0])
        # F.normalize: x / x.norm(2, 1, keepdim=True).clamp_min(1e-12).expand_as(x)
        f = rotations.grad_fn
        if type(f).__name__ != "DivBackward0":
            return None
        rot = _leaf(f.next_functions[0][0])
        e = f.next_functions[1][0]
        if type(e).__name__ != "ExpandBackward0":
            return None
        c = e.next_functions[0][0]
        if type(c).__name__ != "ClampMinBackward0" or float(c._saved_min) != 1e-12:
            return None
        n = c.next_functions[0][0]
        if type(n).__name__ != "LinalgVectorNormBackward0" or float(n._saved_ord) != 2.0 or \
                tuple(n._saved_dim) != (1,) or not n._saved_keepdim or _leaf(n.next_functions[0][0]) is not rot:
            return None
        leaves = (dc, rest, op, sc, rot)
        want = ((P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4))
        for t, shape in zip(leaves, want):
            if t is None or tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() \
                    or t.device != means3D.device or not t.requires_grad:
                return None
        if tuple(sh.shape) != (P, 16, 3) or tuple(opacities.shape) != (P, 1):
            return None
        return leaves
    except (AttributeError, IndexError, TypeError):
        return None


def sh_grad_from_views(means3D, gathered, n_views, sh_degree, scale, out_dc, out_rest):
    """out_dc [P,1,3], out_rest [P,15,3] = scale * sum_v basis(normalize(means3D - campos_v)) x drgb_v
    (include/msgs.h msgs_sh_grad_from_views): the SH gradient of n_views views rebuilt from their factors.
    `gathered` is the all-gathered exchange buffer [n_views, 3P + 4]: row v = {drgb_v [P,3] | campos_v [3] | pad}."""
    P = int(means3D.shape[0])
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("sh_grad_from_views: tensors must live on a HIP device (no CPU path)")
    row = 3 * P + 4
    for t, name, numel in ((means3D, "means3D", 3 * P), (gathered, "gathered", n_views * row), (out_dc, "out_dc", 3 * P),
                           (out_rest, "out_rest", 45 * P)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != numel or t.device != dev:
            raise ValueError(f"sh_grad_from_views: {name} must be a contiguous float32 tensor of {numel} elements on {dev}")
    base = gathered.data_ptr()
    with _on_device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _C.check(_C.lib.msgs_sh_grad_from_views(P, int(n_views), int(sh_degree), _ptr(means3D),
                                                C.c_void_p(base + 4 * 3 * P), row, C.c_void_p(base), row, float(scale),
                                                _ptr(out_dc), _ptr(out_rest), stream), "msgs_sh_grad_from_views")
    return out_dc, out_rest


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta, base_mask, raster_settings):
    _note_grad_mode()
    return _RasterizeGaussiansRaw.apply(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw,
                                        rotation_raw, max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta,
                                        base_mask, raster_settings)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta, base_mask, raster_settings):
    _note_grad_mode()
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, max_pixel_sizes, min_pixel_sizes, occ_multiplier, dc_delta,
                                     base_mask, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of points in front of the near plane (upstream markVisible; unused by the
        reference's Python)."""
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions)
            dev = pos.device
            if dev.type != "cuda":
                raise RuntimeError("markVisible: positions must live on a HIP device")
            vm, pm = _f32c(rs.viewmatrix.to(dev)), _f32c(rs.projmatrix.to(dev))
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=dev)
            with _on_device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                _C.check(_C.lib.msgs_mark_visible(int(pos.shape[0]), _ptr(pos), _ptr(vm), _ptr(pm), _ptr(out),
                                                  stream), "msgs_mark_visible")
        return out.bool()

    def preprocess_only(self, means3D, opacities, scales=None, rotations=None, cov3D_precomp=None,
                        max_pixel_sizes=None, min_pixel_sizes=None, base_mask=None):
        """(radii, pixel_sizes) of this view exactly as forward() returns them, from the per-Gaussian kernel alone —
        no sort / binning / blend, no colour.  For camera sweeps that only read visibility_filter and pixel_sizes
        (/root/reference/train.py:283-300,334-338).  Same argument meaning as forward(); no gradients."""
        rs = self.raster_settings
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        with torch.no_grad():
            P = int(means3D.shape[0])
            dev = means3D.device
            call = _Call(rs, means3D, None, torch.zeros(P, 3, device=dev), opacities, scales, rotations,
                         cov3D_precomp, max_pixel_sizes, min_pixel_sizes, None, None, base_mask)
            radii = torch.zeros(P, dtype=torch.int32, device=dev)
            pixel_sizes = torch.zeros(P, dtype=torch.float32, device=dev)
            if P == 0:
                return radii, pixel_sizes
            lib = _C.lib
            with _on_device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                geom = torch.empty(int(lib.msgs_geom_bytes(P)), dtype=torch.uint8, device=dev)
                _C.check(lib.msgs_preprocess_only(C.byref(call.view), C.byref(call.g), _ptr(radii), _ptr(pixel_sizes),
                                                  _ptr(geom), geom.numel(), stream), "msgs_preprocess_only")
        return radii, pixel_sizes

    def forward_raw(self, xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                    max_pixel_sizes=None, min_pixel_sizes=None, occ_multiplier=None, dc_delta=None, base_mask=None):
        """Opt-in fused path on raw GaussianModel parameters (not part of the reference API)."""
        empty = torch.Tensor([])
        o = lambda t: t if t is not None else empty
        return rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw,
                                       rotation_raw, o(max_pixel_sizes), o(min_pixel_sizes), o(occ_multiplier),
                                       o(dc_delta), o(base_mask), self.raster_settings)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, max_pixel_sizes=None, min_pixel_sizes=None, occ_multiplier=None,
                dc_delta=None, base_mask=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        if chain_reference_getters and torch.is_grad_enabled() and shs is not None and cov3D_precomp is None \
                and means3D.shape[0] > 0 and means3D.device.type == "cuda":
            leaves = _match_reference_getters(means3D, shs, opacities, scales, rotations)
            if leaves is None and not _warned_chain[0] and all(
                    t is not None and t.grad_fn is not None for t in (shs, opacities, scales, rotations)):
                # all four inputs carry an autograd history but it is not the reference's getters (or PyTorch renamed
                # its autograd nodes): correct results through the plain path, but say so once — the chained backward
                # is ~15 % of the step at 1 M Gaussians
                _warned_chain[0] = True
                import warnings
                warnings.warn("diff_gaussian_rasterization: inputs were not recognised as the reference's getters "
                              "(cat / sigmoid / exp / normalize of leaf parameters); their backward runs in autograd")
            if leaves is not None:
                o = lambda t: t if t is not None else empty
                _note_grad_mode()
                return _RasterizeGaussiansChained.apply(
                    means3D, means2D, *leaves, shs.detach() if _chain_reads_cat else empty, opacities.detach(), scales.detach(),
                    rotations.detach(),
                    o(max_pixel_sizes), o(min_pixel_sizes), o(occ_multiplier), o(dc_delta), o(base_mask), rs)
        return rasterize_gaussians(
            means3D, means2D,
            shs if shs is not None else empty,
            colors_precomp if colors_precomp is not None else empty,
            opacities,
            scales if scales is not None else empty,
            rotations if rotations is not None else empty,
            cov3D_precomp if cov3D_precomp is not None else empty,
            max_pixel_sizes if max_pixel_sizes is not None else empty,
            min_pixel_sizes if min_pixel_sizes is not None else empty,
            occ_multiplier if occ_multiplier is not None else empty,
            dc_delta if dc_delta is not None else empty,
            base_mask if base_mask is not None else empty,
            rs)
