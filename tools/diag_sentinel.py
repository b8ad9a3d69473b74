import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes, numpy as np
from parity_utils import PIPE
from gaussian_renderer import render
from synthetic_model import SyntheticGaussians
sc, cam, st = scenes.config("C5")
pc = SyntheticGaussians(sc, "cuda", requires_grad=False)
with torch.no_grad():
    out = render(cam.to("cuda"), pc, PIPE, torch.zeros(3, device="cuda"), **st)
import diff_gaussian_rasterization as dgr
# re-run forward impl to get state
call = dgr._Call(dgr.GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=np.tan(cam.FoVx*0.5), tanfovy=np.tan(cam.FoVy*0.5),
        bg=torch.zeros(3, device="cuda"), scale_modifier=1.0, viewmatrix=cam.world_view_transform.cuda(), projmatrix=cam.full_proj_transform.cuda(), sh_degree=3,
        campos=cam.camera_center.cuda(), prefiltered=False, debug=False, filter_small=True, filter_large=True, fade_size=0.0),
        pc.get_xyz, pc.get_features, None, pc.get_opacity, pc.get_scaling, pc.get_rotation, None, pc.max_pixel_sizes, pc.min_pixel_sizes, None, None, pc.base_gaussian_mask)
color, aps, dep, radii, psz, (geom, binning, image, D) = dgr._forward_impl(call)
P = sc.P; W, H = cam.image_width, cam.image_height
gx, gy = (W + 15) // 16, (H + 15) // 16; tiles = gx * gy
ioff = (8 * tiles + 255) // 256 * 256
ids = binning[ioff:ioff + 4 * D].view(torch.int32).long()
ranges = binning[:8 * tiles].view(torch.int32).view(tiles, 2).long()
print("D", D, "sum", (ranges[:,1]-ranges[:,0]).sum().item(), "max hi", ranges[:,1].max().item())
g = ids[D - 1].item()
def al(x): return (x + 255) // 256 * 256
o_rec = 0; o_rect = al(48 * P); o_tiles = o_rect + al(8 * P)
rec = geom[:48 * P].view(torch.float32).view(P, 12)[g].cpu().numpy()
rect = geom[o_rect:o_rect + 8 * P].view(torch.int32).view(P, 2)[g].cpu().numpy()
cnt = geom[o_tiles:o_tiles + 4 * P].view(torch.int32)[g].item()
minx, miny, maxx, maxy = rect[0] & 0xFFFF, (rect[0] >> 16) & 0xFFFF, rect[1] & 0xFFFF, (rect[1] >> 16) & 0xFFFF
print("gaussian", g, "rec", rec.tolist(), "rect", minx, miny, maxx, maxy, "count", cnt, "instances of g in list", (ids == g).sum().item())
px, py, A, Bh, C, lo = [np.float32(v) for v in rec[:6]]; tau2 = np.float32(rec[11])
f = np.float32
det = f(f(A * C) - f(Bh * Bh)); Atau = f(A * tau2); invA = f(f(1) / A)
dymax = np.sqrt(f(Atau / det)); dx_ext = np.sqrt(f(f(C * tau2) / det)); dyR = f(-f(Bh * dx_ext) / C)
tot = 0
for ty in range(miny, maxy):
    y0 = f(ty * 16)
    lo_ = max(f(py - f(y0 + f(15))), -dymax); hi_ = min(f(py - y0), dymax)
    if lo_ > hi_: print("row", ty, "empty", lo_, hi_); continue
    dyr = min(max(dyR, lo_), hi_); dyl = min(max(-dyR, lo_), hi_)
    sr = np.sqrt(max(f(0), f(f(-det * f(dyr * dyr)) + Atau))); sl = np.sqrt(max(f(0), f(f(-det * f(dyl * dyl)) + Atau)))
    dx_max = f(f(f(-f(Bh * dyr)) - sr) * invA); dx_min = f(f(f(-f(Bh * dyl)) + sl) * invA)
    xl = f(f(px - dx_max) - f(0.02)); xr = f(f(px - dx_min) + f(0.02))
    tlo = max(minx, int(np.ceil(f(f(xl - f(15)) * f(1 / 16))))); thi = min(maxx - 1, int(np.floor(f(xr * f(1 / 16)))))
    n = max(0, thi - tlo + 1); tot += n
    print("row", ty, "xl", xl, "xr", xr, "tlo", tlo, "thi", thi, "n", n, "(xl-15)/16", f(f(xl - f(15)) * f(1/16)), "xr/16", f(xr * f(1/16)))
print("numpy total (no fma)", tot)
