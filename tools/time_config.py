"""Times render() fwd+bwd for a BASELINE config (not a test)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
if os.environ.get("MSGS_BENCH_MT_BACKWARD", "0") != "1":      # as bench.py: backward on the calling thread
    torch.autograd.set_multithreading_enabled(False)
import diff_gaussian_rasterization as dgr
from parity_utils import PIPE
from gaussian_renderer import render
from synthetic_model import SyntheticGaussians
cfg = sys.argv[1]
sc, cam, st = scenes.config(cfg)
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
camd = cam.to("cuda"); bg = torch.zeros(3, device="cuda")
dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).to("cuda")
def step(t=None):
    dgr._C.set_timer(t)
    for p_ in pc.parameters(): p_.grad = None
    out = render(camd, pc, PIPE, bg, **st); out["render"].backward(dL); return out
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tm = dgr._C.KernelTimer()
for _ in range(K): out = step(tm)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(cfg, "ms/step %.3f" % (dt * 1e3), "Mpix/s %.1f" % (cam.image_width * cam.image_height / 1e6 / dt), "D", out["render"].grad_fn.state[3],
      {k: round(v, 3) for k, v in tm.read_ms().items()}, "mem GB %.2f" % (torch.cuda.max_memory_allocated() / 1e9))
