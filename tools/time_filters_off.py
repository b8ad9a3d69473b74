"""Per-kernel breakdown of the forward-only render of the C3 model WITHOUT its multi-scale filters (render.py's defaults on a
multi-scale model: the x4 .. x64 scaled coarse-level Gaussians are all rendered at level 0) — not a test.
usage: time_filters_off.py [k ...]   (pyramid levels, default 0 1 2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
sc, _, _ = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=False)
bg = torch.zeros(3, device="cuda")
plain = dict(filter_small=False, filter_large=False, fade_size=1.0)
on = dict(filter_small=True, filter_large=True, fade_size=1.0)
for k in [int(a) for a in sys.argv[1:]] or [0, 1, 2]:
    W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
    cam = scenes.front_camera(W, H).to("cuda")
    for name, st in (("filters_off", plain), ("filters_on", on)):
        with torch.no_grad():
            for _ in range(3):
                out = render(cam, pc, PIPE, bg, **st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tm = dgr._C.KernelTimer()
            for _ in range(5):
                dgr._C.set_timer(tm)
                out = render(cam, pc, PIPE, bg, **st)
            dgr._C.set_timer(None)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            V = int((out["radii"] > 0).sum())
            big = int((out["radii"] > 256).sum())
        key = (torch.device("cuda").index or 0, sc.P, W, H, int(st["filter_small"]), int(st["filter_large"]))
        D = dgr._last_instances.get(key)
        ms = {a: round(b * 1e3, 1) for a, b in tm.read_ms().items() if b >= 0}
        print(f"k={k} {W}x{H} {name}: {dt * 1e3:.3f} ms  rendered {V}  radius>256px {big}  instances(guess) {D}  us: {ms}", flush=True)
