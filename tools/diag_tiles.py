import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
from parity_utils import PIPE
from gaussian_renderer import render
from synthetic_model import SyntheticGaussians
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
out = render(cam.to("cuda"), pc, PIPE, torch.zeros(3, device="cuda"), **st)
ctx = out["render"].grad_fn
geom, binning, image, D = ctx.state
W, H = cam.image_width, cam.image_height
gx, gy = (W + 15) // 16, (H + 15) // 16
tiles = gx * gy
ranges = binning[:8 * tiles].view(torch.int32).view(tiles, 2).long()
lens = (ranges[:, 1] - ranges[:, 0]).float()
N = W * H
ncon = image[(4 * N + 255) // 256 * 256:][:4 * N].view(torch.int32).view(H, W).float()
pad = torch.zeros(gy * 16, gx * 16, device="cuda"); pad[:H, :W] = ncon
tmax = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(tiles, 256).max(dim=1).values
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device="cuda")
print("D", D, "tiles", tiles)
print("list length  : mean %.0f" % lens.mean().item(), "quantiles", torch.quantile(lens, q).tolist())
print("traversed max: mean %.0f" % tmax.mean().item(), "quantiles", torch.quantile(tmax, q).tolist(), "sum", tmax.sum().item())
