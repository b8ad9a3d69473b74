"""Times render()/render_fused() fwd+bwd of the C3 scene at every pyramid level k (image 1920/2^k x 1080/2^k), the
resolutions MS-GS trains on (train.py:102-105, utils/camera_utils.py:38-39).  Not a test."""
import sys, os, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
if os.environ.get("MSGS_BENCH_MT_BACKWARD", "0") != "1":      # as bench.py: backward on the calling thread
    torch.autograd.set_multithreading_enabled(False)
import diff_gaussian_rasterization as dgr
from parity_utils import PIPE
from gaussian_renderer import render, render_fused
from synthetic_model import SyntheticGaussians
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
for k in range(7):
    W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
    c = scenes.front_camera(W, H).to("cuda")
    dL = scenes.grad_seed(W, H, 5).to("cuda")
    row = []
    for fn in (render, render_fused):
        def step(t=None):
            dgr._C.set_timer(t)
            for p_ in pc.parameters(): p_.grad = None
            out = fn(c, pc, PIPE, bg, **st); out["render"].backward(dL); return out
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10; tm = dgr._C.KernelTimer()
        for _ in range(K): out = step(tm)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / K * 1e3)
    vis = int((out["radii"] > 0).sum())
    print(f"k={k} {W}x{H} visible={vis} D={out['render'].grad_fn.state[3]} render {row[0]:.3f} ms fused {row[1]:.3f} ms",
          {n: round(v, 3) for n, v in tm.read_ms().items()})
