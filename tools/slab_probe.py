"""Round-6 design probe for the depth-slab binning: how much of every tile's list does the forward walk at C5 / C3?
For a rank threshold at fraction f of D (slab A), a tile stays OPEN when its pixels walk past the slab; slab B then re-emits
the open tiles' whole lists.  Approximation: a tile's entries are uniform in rank, so (walked / length) is the rank fraction
the tile needs.  Prints, per f: open tiles, instances B would emit, A + B relative to D."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import scenes  # noqa: E402
from parity_utils import PIPE  # noqa: E402


def probe(name):
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    sc, cam, st = scenes.config(name)
    cam = cam.to("cuda")
    bg = torch.zeros(3, device="cuda")
    pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
    out = render(cam, pc, PIPE, bg, **st)
    torch.cuda.synchronize()
    ctx = out["render"].grad_fn
    geom, binning, image, D = ctx.state
    W, H = cam.image_width, cam.image_height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = gx * gy
    ranges = binning[:8 * tiles].view(torch.int32).view(tiles, 2).long()
    n4 = 4 * W * H
    a = (n4 + 255) & ~255
    ncon = image[a:a + n4].view(torch.int32).view(H, W).long()
    Hp, Wp = gy * 16, gx * 16
    pad = torch.zeros(Hp, Wp, dtype=torch.long, device="cuda")
    pad[:H, :W] = ncon
    walk = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(tiles, 256).max(dim=1).values
    ln = ranges[:, 1] - ranges[:, 0]
    frac = walk.double() / ln.clamp_min(1).double()
    print(f"[{name}] D={D} tiles={tiles} sum(len)={int(ln.sum())} D_trav={int(walk.sum())} "
          f"mean frac={float(frac.mean()):.3f} p50={float(frac.median()):.3f} p90={float(frac.quantile(0.9)):.3f} "
          f"p99={float(frac.quantile(0.99)):.3f} max={float(frac.max()):.3f}")
    # a tile whose pixels never all terminate walks its whole list: frac = 1
    for f in (0.05, 0.08, 0.1, 0.12, 0.15, 0.2, 0.25, 0.3):
        open_ = (frac > f) | (walk >= ln)
        DB = int(ln[open_].sum())
        DA = int(f * D)
        print(f"   f={f:.2f}: open tiles {int(open_.sum())} ({100.0 * int(open_.sum()) / tiles:.1f} %), B instances {DB} "
              f"({100.0 * DB / D:.1f} % of D), A + B = {100.0 * (DA + DB) / D:.1f} % of D, "
              f"re-blended walk {int(walk[open_].sum())} of {int(walk.sum())}")
    full = int((walk >= ln).sum())
    print(f"   tiles that walk their whole list: {full}")


if __name__ == "__main__":
    for n in sys.argv[1:] or ["C5", "C3"]:
        probe(n)
        torch.cuda.empty_cache()
