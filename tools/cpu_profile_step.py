"""cProfile of the host side of one fwd+bwd step on a tiny scene (GPU work negligible): where the CPU time goes."""
import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
if os.environ.get("MSGS_BENCH_MT_BACKWARD", "0") != "1":
    torch.autograd.set_multithreading_enabled(False)
from parity_utils import PIPE, small_scene
from gaussian_renderer import render, render_fused
from synthetic_model import SyntheticGaussians
fn = render_fused if (len(sys.argv) > 1 and sys.argv[1] == "fused") else render
sc, cam = small_scene(2000, 64, 64, 1)
pc = SyntheticGaussians(sc, "cuda")
camd = cam.to("cuda"); bg = torch.zeros(3, device="cuda"); dL = scenes.grad_seed(64, 64, 1).cuda()
def step():
    for p_ in pc.parameters(): p_.grad = None
    out = fn(camd, pc, PIPE, bg); out["render"].backward(dL)
for _ in range(20): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
