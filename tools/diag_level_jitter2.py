"""Host-side time of every un-synchronised step of bench.py's pyramid loops at one level (diagnostic for an intermittent
~20-50 ms hiccup seen at k = 3): python tools/diag_level_jitter2.py [--no-gc]
With --no-gc Python's cyclic collector is paused; gc.callbacks report every collection either way (generation, duration)."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
_gc_t = [0.0]
def _gc_note(phase, info):
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    elif info["generation"] >= 1:
        print(f"    [gc gen {info['generation']} took {1e3 * (time.perf_counter() - _gc_t[0]):.2f} ms, collected {info['collected']}]")
gc.callbacks.append(_gc_note)
if "--no-gc" in sys.argv:
    gc.collect(); gc.disable()
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
plain = dict(filter_small=False, filter_large=False, fade_size=1.0)
cam0 = scenes.front_camera(1920, 1080).to("cuda")
with torch.no_grad():
    ps0 = render(cam0, pc, PIPE, bg, **plain)["pixel_sizes"]
keep = pc.min_pixel_sizes
trained = torch.where((pc.max_pixel_sizes < 0) & (ps0 > 0), 0.8 * ps0, keep).contiguous()
for rep in range(3):
    for name, mins in (("C3", keep), ("trained-like", trained)):
        pc.min_pixel_sizes = mins
        for k in (2, 3, 4):
            W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
            c = scenes.front_camera(W, H).to("cuda"); dL = scenes.grad_seed(W, H, 40 + k).to("cuda")
            host = []
            torch.cuda.synchronize(); T0 = time.perf_counter()
            for it in range(13):
                t0 = time.perf_counter()
                for p_ in pc.parameters(): p_.grad = None
                out = render(c, pc, PIPE, bg, **st); t1 = time.perf_counter()
                out["render"].backward(dL); t2 = time.perf_counter()
                host.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
            torch.cuda.synchronize(); tot = 1e3 * (time.perf_counter() - T0)
            D = out["render"].grad_fn.state[3]
            flag = "  <-- HICCUP" if max(max(h) for h in host) > 5 else ""
            print(f"rep {rep} {name} k={k} D={D}: total {tot:.1f} ms; fwd host " + " ".join(f"{h[0]:.2f}" for h in host) + " | bwd host " + " ".join(f"{h[1]:.2f}" for h in host) + flag)
