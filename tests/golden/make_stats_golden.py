"""Generates tests/golden/stats_*.npz by running the REFERENCE's own per-iteration statistics code on seeded inputs:
    python tests/golden/make_stats_golden.py      (needs /root/reference; build container only, never on the GPU box)

What runs is the reference's unmodified `GaussianModel.update_pixel_sizes`, `add_densification_stats` and
`update_base_gaussian_mask` (/root/reference/scene/gaussian_model.py:663-704) plus the two statements of
/root/reference/train.py:247-250 around them, on CPU tensors.  `scene/gaussian_model.py` is loaded BY FILE PATH (not through
the `scene` package, whose __init__ pulls in the dataset readers), and the third-party modules its import block names but
this image lacks (cv2, open3d, plyfile, simple_knn) are satisfied with EMPTY placeholder modules: none of their names is
touched by the three methods exercised here — the placeholders only let the `import` statements pass.
Stored per case: every input and every output array, so that oracle/epilogue_oracle.py::training_stats can be pinned to the
reference's behaviour (tests/test_epilogue_cpu.py) and, through it, the HIP kernel (tests/test_epilogue_gpu.py)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_placeholder("cv2")
_placeholder("open3d"); _placeholder("open3d.ml"); _placeholder("open3d.ml.torch")
_placeholder("plyfile", PlyData=None, PlyElement=None)
_placeholder("simple_knn"); _placeholder("simple_knn._C", distCUDA2=None)
# scene/cameras.py is importable as is, but only through the `scene` package; load both files by path
pkg = types.ModuleType("scene"); pkg.__path__ = [os.path.join(REF, "scene")]; sys.modules["scene"] = pkg
for mod in ("cameras", "gaussian_model"):
    spec = importlib.util.spec_from_file_location(f"scene.{mod}", os.path.join(REF, "scene", f"{mod}.py"))
    m = importlib.util.module_from_spec(spec); sys.modules[f"scene.{mod}"] = m; spec.loader.exec_module(m)
GaussianModel = sys.modules["scene.gaussian_model"].GaussianModel

CASES = {"a": dict(P=4000, L=4, lvl=0, seed=1, base=False), "b": dict(P=4000, L=4, lvl=2, seed=2, base=False),
         "c": dict(P=3000, L=4, lvl=3, seed=3, base=True), "d": dict(P=2500, L=1, lvl=0, seed=4, base=False),
         "e": dict(P=5000, L=7, lvl=6, seed=5, base=True)}

for name, c in CASES.items():
    P, L, lvl = c["P"], c["L"], c["lvl"]
    g = torch.Generator().manual_seed(c["seed"])
    radii = torch.where(torch.rand(P, generator=g) < 0.6, torch.randint(1, 40, (P,), generator=g), torch.zeros(P, dtype=torch.long)).to(torch.int32)
    pixel_sizes = 6.0 * torch.rand(P, generator=g)
    pixel_sizes[torch.rand(P, generator=g) < 0.15] = 0.0                      # invalid sizes
    grad = torch.zeros(P, 3)
    grad[:, :2] = 1e-3 * torch.randn(P, 2, generator=g)
    target = torch.randint(0, L, (P,), generator=g)
    m = GaussianModel.__new__(GaussianModel)
    m.reso_lvls = L
    m.target_reso_lvl = target.clone()
    m.xyz_gradient_accum = torch.rand(P, L, 1, generator=g)
    m.denom = torch.randint(0, 5, (P, L, 1), generator=g).float()
    m.max_radii2D = 30.0 * torch.rand(P, generator=g)
    m.max_pixel_sizes = torch.where(torch.rand(P, generator=g) < 0.3, -torch.ones(P), 5.0 * torch.rand(P, generator=g))
    m.min_pixel_sizes = torch.where(torch.rand(P, generator=g) < 0.4, -torch.ones(P), 2.0 * torch.rand(P, generator=g))
    m.base_gaussian_mask = torch.rand(P, generator=g) < 0.1
    inputs = dict(radii=radii.numpy(), pixel_sizes=pixel_sizes.numpy(), grad2d=grad.numpy(), target_reso_lvl=target.numpy(),
                  reso_lvl=lvl, reso_lvls=L, do_base_mask=c["base"],
                  in_xyz_gradient_accum=m.xyz_gradient_accum.numpy().copy(), in_denom=m.denom.numpy().copy(),
                  in_max_radii2D=m.max_radii2D.numpy().copy(), in_max_pixel_sizes=m.max_pixel_sizes.numpy().copy(),
                  in_min_pixel_sizes=m.min_pixel_sizes.numpy().copy(), in_base_mask=m.base_gaussian_mask.numpy().copy())
    # ---- the reference's block, train.py:239-250 ----
    visibility_filter = radii > 0                                             # gaussian_renderer/__init__.py:116
    viewspace_point_tensor = types.SimpleNamespace(grad=grad)
    with torch.no_grad():
        if c["base"]:
            m.update_base_gaussian_mask(visibility_filter)                    # train.py:239-241
        m.update_pixel_sizes(visibility_filter, pixel_sizes, lvl, 1000)       # train.py:244-245
        m.max_radii2D[visibility_filter] = torch.max(m.max_radii2D[visibility_filter], radii[visibility_filter])   # train.py:249
        m.add_densification_stats(viewspace_point_tensor, visibility_filter, lvl)                                 # train.py:250
    np.savez_compressed(os.path.join(HERE, f"stats_{name}.npz"), **inputs,
                        out_xyz_gradient_accum=m.xyz_gradient_accum.numpy(), out_denom=m.denom.numpy(),
                        out_max_radii2D=m.max_radii2D.numpy(), out_max_pixel_sizes=m.max_pixel_sizes.numpy(),
                        out_min_pixel_sizes=m.min_pixel_sizes.numpy(), out_base_mask=m.base_gaussian_mask.numpy())
    print(name, "visible", int(visibility_filter.sum()), "max_ps changed", int((m.max_pixel_sizes.numpy() != inputs["in_max_pixel_sizes"]).sum()),
          "min_ps changed", int((m.min_pixel_sizes.numpy() != inputs["in_min_pixel_sizes"]).sum()))
