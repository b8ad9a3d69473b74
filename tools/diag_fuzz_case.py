"""Diagnostic: one fuzz configuration — which Gaussian carries the worst dL/dfeatures_dc error against the oracle, and what is
special about it (filter weight, opacity, radius, pixel size vs its thresholds).  python tools/diag_fuzz_case.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from oracle import oracle_ctypes as oc
from parity_utils import hip_render, small_scene
c = dict(P=777, W=32, H=106, deg=1, ms=True, fade=1.0, seed=985935)
sc, cam = small_scene(c["P"], c["W"], c["H"], c["seed"], sh_degree=c["deg"], multiscale=c["ms"], scale_k=0.004 * 1920.0 / max(c["W"], 8) * 0.3)
st = dict(filter_small=True, filter_large=True, fade_size=c["fade"])
bg = torch.rand(3, generator=torch.Generator().manual_seed(c["seed"]))
dL = scenes.grad_seed(c["W"], c["H"], c["seed"] % 97)
out, pc, m2 = hip_render(sc, cam, st, bg, dL)
orc = oc.rasterize(pc.seen, cam, st, bg); og = oc.backward(orc, dL)
print("forward max abs diff", (out["render"].cpu() - orc.color).abs().max().item(), "radii equal", torch.equal(out["radii"].cpu(), orc.radii),
      "pixel_sizes max diff", (out["pixel_sizes"].cpu() - orc.pixel_sizes).abs().max().item())
gd = pc._features_dc.grad.cpu().reshape(-1, 3); rd = og["shs"].reshape(c["P"], -1, 3)[:, 0, :]
err = (gd - rd).abs().max(dim=1).values
print("dc: max ref", rd.abs().max().item(), "max err", err.max().item())
seen = pc.seen
for i in torch.topk(err, 5).indices.tolist():
    print(i, "err", err[i].item(), "hip", gd[i].tolist(), "ref", rd[i].tolist(), "radius", int(orc.radii[i]), "psize", orc.pixel_sizes[i].item(),
          "min/max ps", seen.min_pixel_sizes[i].item(), seen.max_pixel_sizes[i].item(), "opacity", seen.opacities[i].item(),
          "borderline", bool(orc.borderline_gaussians[i]), "mean", seen.means3D[i].tolist())
