"""Times render() vs render_fused() fwd+bwd at C3 (not a test)."""
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
camd = cam.to("cuda"); bg = torch.zeros(3, device="cuda")
dL = scenes.grad_seed(cam.image_width, cam.image_height, 2).to("cuda")
for name, fn in (("render", render), ("render_fused", render_fused)):
    def step(t=None):
        dgr._C.set_timer(t)
        for p_ in pc.parameters(): p_.grad = None
        out = fn(camd, pc, PIPE, bg, **st); out["render"].backward(dL); return out
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20; tm = dgr._C.KernelTimer()
    for _ in range(K): step(tm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(name, "ms/step %.3f" % (dt * 1e3), "Mpix/s %.1f" % (cam.image_width * cam.image_height / 1e6 / dt),
          {k: round(v, 3) for k, v in tm.read_ms().items() if k in ("preprocess", "preprocess_bwd")})
