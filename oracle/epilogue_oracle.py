"""CPU restatement of the train-step epilogue (Adam update + densification statistics).

TEST INFRASTRUCTURE ONLY — imported by tests/ (and bench.py's cpu leg); ms-gs_amd/ never imports this.

Parity status:
  * adam_step: PINNED.  The reference calls torch.optim.Adam (/root/reference/scene/gaussian_model.py:248); torch is
    present in this image, and tests/test_epilogue_cpu.py checks this numpy restatement against torch.optim.Adam on
    CPU (exp_avg / exp_avg_sq bit-for-bit, parameters within 1 ulp: ATen's CPU sqrt is not correctly rounded), and the GPU tests additionally compare the HIP kernel with
    torch.optim.Adam running on the same GPU.
  * training_stats: PINNED.  tests/golden/stats_*.npz hold inputs and outputs of the reference's OWN
    update_base_gaussian_mask / update_pixel_sizes / add_densification_stats (scene/gaussian_model.py:663-704, loaded by
    file path in the build container with empty placeholders for the third-party modules its import block names:
    tests/golden/make_stats_golden.py) plus train.py:249; tests/test_epilogue_cpu.py requires this numpy restatement to
    reproduce them bit for bit (the accumulated gradient norm to one ulp: torch.norm vs sqrt(x*x + y*y)).
"""
import numpy as np

f32 = np.float32


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """torch/optim/adam.py::_single_tensor_adam (amsgrad=False, weight_decay=0, maximize=False) in float32;
    scalars are formed in double and rounded once, as torch's Python-float scalars are.  In place."""
    w1 = f32(1.0 - beta1)
    # exp_avg.lerp_(grad, 1 - beta1): torch evaluates fma(weight, end - start, start) (ATen/native/Lerp.h, weight < 0.5)
    m[...] = (m.astype(np.float64) + (g - m).astype(np.float64) * np.float64(w1)).astype(f32)
    v *= f32(beta2)
    # addcmul_(grad, grad, value=1 - beta2): ATen's vectorised kernel evaluates fma(value * t1, t2, self)
    v[...] = (v.astype(np.float64) + (f32(1.0 - beta2) * g).astype(np.float64) * g.astype(np.float64)).astype(f32)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / f32(bc2 ** 0.5) + f32(eps)
    p += (f32(-(lr / bc1)) * m) / denom                            # addcdiv_(exp_avg, denom, value=-step_size)


def training_stats(radii, pixel_sizes, grad2d, target_reso_lvl, reso_lvl, reso_lvls, xyz_gradient_accum, denom,
                   max_radii2D, max_pixel_sizes, min_pixel_sizes, base_mask, *, do_base_mask, do_pixel_sizes,
                   do_densify):
    """All arrays numpy, updated in place.  visibility_filter = radii > 0 (gaussian_renderer/__init__.py:117)."""
    vis = radii > 0
    if do_base_mask:                                               # gaussian_model.py:702-704
        base_mask |= vis
    if do_pixel_sizes:                                             # gaussian_model.py:663-687
        mask = vis & (target_reso_lvl == reso_lvl)
        ps = pixel_sizes[mask]
        if reso_lvl > 0:
            max_pixel_sizes[mask] = np.maximum(max_pixel_sizes[mask] * f32(0.95), ps)
        if reso_lvl < reso_lvls - 1:
            grown = np.maximum(min_pixel_sizes[mask] * f32(1.05), f32(-1.0))
            valid = ps > 0
            fresh = grown < 0
            out = grown.copy()
            out[valid & fresh] = ps[valid & fresh]
            both = valid & ~fresh
            out[both] = np.minimum(grown[both], ps[both])
            min_pixel_sizes[mask] = out
    if do_densify:                                                 # train.py:247-250, gaussian_model.py:698-701
        max_radii2D[vis] = np.maximum(max_radii2D[vis], radii[vis].astype(f32))
        gx, gy = grad2d[vis, 0], grad2d[vis, 1]
        xyz_gradient_accum[vis, reso_lvl, 0] += np.sqrt(gx * gx + gy * gy)
        denom[vis, reso_lvl, 0] += f32(1.0)
