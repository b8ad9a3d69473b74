"""One pyramid level of the C3 scene in a plain step loop, for rocprofv3 --kernel-trace + tools/idle_gaps.py:
python tools/level_loop.py <k> [render|fused|forward] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import time
import torch, scenes
if os.environ.get("MSGS_BENCH_MT_BACKWARD", "0") != "1":      # as bench.py: backward on the calling thread
    torch.autograd.set_multithreading_enabled(False)
from parity_utils import PIPE
from gaussian_renderer import render, render_fused
from synthetic_model import SyntheticGaussians
k = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "render"; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
c = scenes.front_camera(W, H).to("cuda"); dL = scenes.grad_seed(W, H, 5).to("cuda")
fn = render_fused if mode == "fused" else render
def step():
    if mode == "forward":
        with torch.no_grad():
            return render(c, pc, PIPE, bg, filter_small=True, filter_large=True, fade_size=1.0)
    for p_ in pc.parameters(): p_.grad = None
    out = fn(c, pc, PIPE, bg, **st); out["render"].backward(dL); return out
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); print(f"k={k} {mode}: {1e3 * (time.perf_counter() - t) / steps:.3f} ms/step")
