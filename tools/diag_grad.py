"""Diagnostic (not a test): where do the HIP gradients deviate most from the oracle at C2?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
from parity_utils import hip_render
from oracle import oracle_ctypes as oc
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc, cam, st = scenes.config(cfg)
bg = torch.zeros(3)
W, H = cam.image_width, cam.image_height
dL = scenes.grad_seed(W, H, 1)
out, pc, m2 = hip_render(sc, cam, st, bg, dL)
r = oc.rasterize(sc, cam, st, bg); g = oc.backward(r, dL)
out2, pc2, m22 = hip_render(sc, cam, st, bg, dL)
print("run-to-run (atomics order) rel diff: scaling", ((pc._scaling.grad-pc2._scaling.grad).abs().max()/pc._scaling.grad.abs().max()).item(),
      "means2D", ((m2-m22).abs().max()/m2.abs().max()).item())
sg = (pc._scaling.grad.cpu().double() / torch.exp(pc._scaling.detach().cpu().double()))   # back to dL/dscale
ref = g["scales"].double()
err = (sg - ref).abs()
print("scales: max ref", ref.abs().max().item(), "max err", err.max().item())
top = torch.topk(err.max(dim=1).values, 8).indices
rad = out["radii"].cpu(); con = r._arr("conic_opacity", (sc.P, 4), torch.float32)
for i in top.tolist():
    print(i, "err", err[i].tolist(), "ref", ref[i].tolist(), "hip", sg[i].tolist(), "radius", rad[i].item(), "opac", sc.opacities[i].item(),
          "scale", sc.scales[i].tolist(), "z", sc.means3D[i, 2].item(), "conic", con[i].tolist())
m2r = g["means2D"].double(); e2 = (m2.cpu().double() - m2r).abs()
print("means2D: max ref", m2r.abs().max().item(), "max err", e2.max().item())
top = torch.topk(e2.max(dim=1).values, 5).indices
for i in top.tolist():
    print(i, "err", e2[i].tolist(), "ref", m2r[i].tolist(), "hip", m2[i].cpu().tolist(), "radius", rad[i].item(), "opac", sc.opacities[i].item())
# distribution of relative errors per Gaussian relative to own magnitude
own = (err.max(dim=1).values / ref.abs().max(dim=1).values.clamp_min(1e-12))
vis = rad > 0
print("per-Gaussian own-relative scale-grad error quantiles (visible):", torch.quantile(own[vis], torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64)).tolist())
