"""Per-step wall time of render() fwd+bwd (synchronised every step) while switching between pyramid levels of the C3
scene, as MS-GS training does per iteration (train.py:185-190): no allocation or warm-up penalty after the first visit.
Not a test."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from parity_utils import PIPE
from gaussian_renderer import render, render_fused
from synthetic_model import SyntheticGaussians
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
for k in (2, 3, 4, 3):
    W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
    c = scenes.front_camera(W, H).to("cuda")
    dL = scenes.grad_seed(W, H, 5).to("cuda")
    ts = []
    for i in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for p_ in pc.parameters(): p_.grad = None
        out = render(c, pc, PIPE, bg, **st); out["render"].backward(dL)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(k, W, H, " ".join("%.2f" % t for t in ts), "mem", torch.cuda.memory_reserved() >> 20, flush=True)
