"""Per-step wall time of a BASELINE config with the caching allocator's device-allocation counters (diagnostic, not a test):
python tools/diag_step_jitter.py C5"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
cfg = sys.argv[1]
sc, cam, st = scenes.config(cfg)
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
camd = cam.to("cuda"); bg = torch.zeros(3, device="cuda")
dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).to("cuda")
def stats():
    s = torch.cuda.memory_stats()
    return s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("num_alloc_retries", 0), s.get("reserved_bytes.all.current", 0) >> 20
prev = stats()
for k in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for p_ in pc.parameters(): p_.grad = None
    out = render(camd, pc, PIPE, bg, **st); t1 = time.perf_counter()
    out["render"].backward(dL); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    cur = stats()
    print(f"step {k}: total {1e3*(t3-t0):.3f} ms (fwd host {1e3*(t1-t0):.3f}, bwd host {1e3*(t2-t1):.3f}, drain {1e3*(t3-t2):.3f}) "
          f"device allocs +{cur[0]-prev[0]} frees +{cur[1]-prev[1]} retries +{cur[2]-prev[2]} reserved {cur[3]} MiB")
    prev = cur
