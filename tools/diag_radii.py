import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
from parity_utils import hip_render
from oracle import oracle_ctypes as oc
sc, _, st = scenes.config("C3")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
cam = scenes.front_camera(W, H); bg = torch.zeros(3)
out, _, _ = hip_render(sc, cam, st, bg)
r = oc.rasterize(sc, cam, st, bg)
hr = out["radii"].cpu(); mis = (hr != r.radii).nonzero().flatten()
print("W,H", W, H, "mismatches", mis.numel(), "visible", (r.radii > 0).sum().item())
rects = r._arr("rects", (sc.P, 4), torch.int32); con = r._arr("conic_opacity", (sc.P, 4), torch.float32); m2d = r._arr("means2D", (sc.P, 2), torch.float32)
for i in mis[:10].tolist():
    print(i, "hip", hr[i].item(), "ora", r.radii[i].item(), "psz hip/ora", out["pixel_sizes"][i].item(), r.pixel_sizes[i].item(),
          "min/max ps", sc.min_pixel_sizes[i].item(), sc.max_pixel_sizes[i].item(), "rect", rects[i].tolist(), "mean2D", m2d[i].tolist(),
          "z", sc.means3D[i, 2].item(), "conic", con[i].tolist(), "lvl", sc.target_reso_lvl[i].item())
