"""GPU train-step epilogue for MS-GS: what the reference runs over all P Gaussians between
loss.backward() and the next iteration, as two launches of libmsgs_hip.so instead of ~40 elementwise torch kernels.

  FusedAdam(param_groups, lr=0.0, eps=1e-15)
      stand-in for the torch.optim.Adam the reference builds in GaussianModel.training_setup
      (/root/reference/scene/gaussian_model.py:235-248).  Same param_groups / state layout (state[p]["step"],
      ["exp_avg"], ["exp_avg_sq"]), so the reference's densification code that slices, concatenates and re-keys the
      optimizer state (gaussian_model.py:419-476) and update_learning_rate() (:284-291) work unchanged.
      step() is one msgs_adam_step launch over every tensor that has a gradient.

  update_training_stats(model, viewspace_points, radii, pixel_sizes, reso_lvl, ...)
      the no_grad block of /root/reference/train.py:239-250 — update_base_gaussian_mask, update_pixel_sizes,
      max_radii2D, add_densification_stats (gaussian_model.py:663-704) — as one msgs_densify_stats launch on the
      model's own state tensors (attribute names of the reference's GaussianModel).

There is no CPU or torch fallback: CPU tensors raise.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _backend as _C


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require_gpu_f32(t, what):
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: tensor lives on {t.device}; the epilogue kernels are GPU-only (no CPU path)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{what}: need a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-15):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid betas: {betas}")
        if eps < 0.0 or lr < 0.0:
            raise ValueError("Invalid eps / lr")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # launches are keyed by (device, step, betas, eps): in the reference every tensor shares them
        batches = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue                                  # torch.optim.Adam skips these too
                _require_gpu_f32(p, "FusedAdam param")
                g = p.grad
                if g.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(torch.float32).contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                _require_gpu_f32(st["exp_avg"], "FusedAdam exp_avg")
                _require_gpu_f32(st["exp_avg_sq"], "FusedAdam exp_avg_sq")
                if st["exp_avg"].numel() != p.numel() or st["exp_avg_sq"].numel() != p.numel():
                    raise ValueError("FusedAdam: optimizer state and parameter sizes differ")
                st["step"] += 1
                key = (p.device, int(st["step"].item()), tuple(group["betas"]), float(group["eps"]))
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (dev, step, betas, eps), items in batches.items():
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for lo in range(0, len(items), _C.ADAM_MAX_TENSORS):
                    part = items[lo:lo + _C.ADAM_MAX_TENSORS]
                    arr = (_C.AdamTensor * len(part))()
                    for k, (p, g, m, v, lr) in enumerate(part):
                        arr[k] = _C.AdamTensor(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr)
                    _C.check(_C.lib.msgs_adam_step(arr, len(part), step, betas[0], betas[1], eps, stream),
                             "msgs_adam_step")
        return loss


    # ---- the step inside the rasterizer's backward (diff_gaussian_rasterization.set_optimizer_in_backward) ----
    def take_step_in_backward(self, leaves):
        """Called by the raw entry's backward with its six leaf tensors (xyz, features_dc, features_rest, opacity, scaling,
        rotation): returns the msgs_adam_in_backward_t table of the step that step() would take next — the per-Gaussian
        backward kernel then updates parameters and moments itself (include/msgs.h) — WITHOUT advancing the step counters:
        the backward calls commit_step_in_backward(leaves) once the library has accepted the launch.  The six tensors must
        be parameters of this optimizer with common betas / eps / step (the reference's setup, gaussian_model.py:235-248)."""
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        table = _C.AdamInBackward()
        common = None
        keep = []
        for k, p in enumerate(leaves):
            group = group_of.get(id(p))
            if group is None:
                raise ValueError("take_step_in_backward: a leaf of the render call is not a parameter of this optimizer "
                                 "(densification re-creates the tensors: install the optimizer again after it)")
            _require_gpu_f32(p, "FusedAdam param")
            if p.grad is not None:
                raise RuntimeError("take_step_in_backward: the parameter already holds a .grad — the step inside the backward "
                                   "would ignore it (call zero_grad(set_to_none=True) first)")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            _require_gpu_f32(st["exp_avg"], "FusedAdam exp_avg")
            _require_gpu_f32(st["exp_avg_sq"], "FusedAdam exp_avg_sq")
            if st["exp_avg"].numel() != p.numel() or st["exp_avg_sq"].numel() != p.numel():
                raise ValueError("FusedAdam: optimizer state and parameter sizes differ")
            key = (int(st["step"].item()) + 1, tuple(group["betas"]), float(group["eps"]))
            if common is None:
                common = key
            elif key != common:
                raise ValueError(f"take_step_in_backward: the six tensors must share step / betas / eps, got {common} and {key}")
            table.t[k] = _C.AdamMoments(_ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"]), float(group["lr"]))
            keep.append((st["exp_avg"], st["exp_avg_sq"]))
        table.step, (table.beta1, table.beta2), table.eps = common[0], common[1], common[2]
        table._keep = keep
        return table

    def commit_step_in_backward(self, leaves):
        """the kernel that takes the step has been launched: advance the six tensors' step counters"""
        for p in leaves:
            self.state[p]["step"] += 1
        self.steps_in_backward = getattr(self, "steps_in_backward", 0) + 1


def update_training_stats(model, viewspace_points, radii, pixel_sizes, reso_lvl=0, *, base_mask=False,
                          update_pixel_sizes=True, densify=True):
    """One launch for the reference's per-iteration statistics (train.py:239-250).  `model` carries the reference's
    attribute names: xyz_gradient_accum / denom [P, reso_lvls, 1], max_radii2D, max_pixel_sizes, min_pixel_sizes [P]
    float32, base_gaussian_mask [P] bool, target_reso_lvl [P] int64, reso_lvls.  `viewspace_points` is
    render_pkg["viewspace_points"] (its .grad is read), `radii` / `pixel_sizes` the render outputs.
      base_mask           train.py:239-241  (caller decides: preserve_large and past densify_until_iter, coarsest level)
      update_pixel_sizes  train.py:244-245
      densify             train.py:247-250  (iteration < densify_until_iter)"""
    flags = (_C.STATS_BASE_MASK if base_mask else 0) | (_C.STATS_PIXEL_SIZES if update_pixel_sizes else 0) | \
            (_C.STATS_DENSIFY if densify else 0)
    if flags == 0:
        return
    P = int(radii.shape[0])
    if radii.device.type != "cuda":
        raise RuntimeError("update_training_stats: tensors must live on the GPU (no CPU path)")
    if radii.dtype != torch.int32 or not radii.is_contiguous():
        raise ValueError("radii must be the contiguous int32 tensor returned by the rasterizer")
    d = _C.DensifyStats()
    d.P, d.flags, d.reso_lvl, d.reso_lvls = P, flags, int(reso_lvl), int(model.reso_lvls)
    d.radii = _ptr(radii)
    keep = []
    if base_mask:
        m = model.base_gaussian_mask
        if m.dtype != torch.bool or m.numel() != P or not m.is_contiguous():
            raise ValueError("base_gaussian_mask must be a contiguous [P] bool tensor")
        d.base_mask = _ptr(m)
    if update_pixel_sizes:
        lvl = model.target_reso_lvl
        if lvl.dtype != torch.int64 or lvl.numel() != P or not lvl.is_contiguous():
            raise ValueError("target_reso_lvl must be a contiguous [P] int64 tensor")
        for t, n in ((pixel_sizes, "pixel_sizes"), (model.max_pixel_sizes, "max_pixel_sizes"),
                     (model.min_pixel_sizes, "min_pixel_sizes")):
            _require_gpu_f32(t, n)
            if t.numel() != P:
                raise ValueError(f"{n} must have P elements")
        d.pixel_sizes, d.target_reso_lvl = _ptr(pixel_sizes), _ptr(lvl)
        d.max_pixel_sizes, d.min_pixel_sizes = _ptr(model.max_pixel_sizes), _ptr(model.min_pixel_sizes)
    if densify:
        g = viewspace_points.grad
        if g is None:
            raise RuntimeError("viewspace_points.grad is None: call loss.backward() first")
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        keep.append(g)
        for t, n, numel in ((g, "viewspace grad", 3 * P), (model.xyz_gradient_accum, "xyz_gradient_accum", P * d.reso_lvls),
                            (model.denom, "denom", P * d.reso_lvls), (model.max_radii2D, "max_radii2D", P)):
            _require_gpu_f32(t, n)
            if t.numel() != numel:
                raise ValueError(f"{n}: expected {numel} elements, got {t.numel()}")
        d.means2D_grad, d.xyz_gradient_accum, d.denom = _ptr(g), _ptr(model.xyz_gradient_accum), _ptr(model.denom)
        d.max_radii2D = _ptr(model.max_radii2D)
    with torch.cuda.device(radii.device):
        stream = C.c_void_p(torch.cuda.current_stream(radii.device).cuda_stream)
        _C.check(_C.lib.msgs_densify_stats(C.byref(d), stream), "msgs_densify_stats")
