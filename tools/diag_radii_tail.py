import os, sys
ROOT = os.getcwd()
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
from parity_utils import PIPE
from gaussian_renderer import render
from synthetic_model import SyntheticGaussians
for cfg in ("C3", "C5"):
    sc, cam, st = scenes.config(cfg)
    pc = SyntheticGaussians(sc, "cuda", requires_grad=False)
    with torch.no_grad():
        out = render(cam.to("cuda"), pc, PIPE, torch.zeros(3, device="cuda"), **st)
    r = out["radii"].float(); v = r[r > 0]
    t = ((2 * v / 16) + 1) ** 2
    q = torch.quantile(t[:10_000_000] if t.numel() > 10_000_000 else t, torch.tensor([0.5, 0.9, 0.99, 0.999], device=t.device))
    print(cfg, "visible", v.numel(), "radius max", v.max().item(), "rect tiles median/p90/p99/p99.9", q.tolist(), "max", t.max().item(),
          "n(rect>256)", int((t > 256).sum()), "n(rect>1024)", int((t > 1024).sum()))
