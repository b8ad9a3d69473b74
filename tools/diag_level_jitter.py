"""Per-step wall time of render()+backward() at one pyramid level of the C3 scene, every step synchronised (diagnostic):
python tools/diag_level_jitter.py 3"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
for k in [int(a) for a in sys.argv[1:]] or [2, 3, 4]:
    W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
    c = scenes.front_camera(W, H).to("cuda"); dL = scenes.grad_seed(W, H, 5).to("cuda")
    ts = []
    for it in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for p_ in pc.parameters(): p_.grad = None
        out = render(c, pc, PIPE, bg, **st); t1 = time.perf_counter()
        out["render"].backward(dL); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        ts.append((1e3 * (t3 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    s = torch.cuda.memory_stats()
    print(f"k={k}: " + " ".join(f"{t[0]:.2f}" for t in ts))
    worst = max(range(30), key=lambda i: ts[i][0])
    print(f"   worst step {worst}: total {ts[worst][0]:.2f} ms = fwd host {ts[worst][1]:.2f} + bwd host {ts[worst][2]:.2f} + drain {ts[worst][3]:.2f}; device allocs so far {s.get('num_device_alloc', 0)}")
