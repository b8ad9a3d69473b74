"""Single-GPU cost of the kernels the factored gradient exchange adds: the factor kernel inside msgs_backward
(sh_factor_kernel, in front of K9), K9 without its SH row stores, and msgs_sh_grad_from_views for 1 / 2 / 4 / 8 views at
1 M Gaussians.  HIP-event timing around each call, median of 20.  Not a test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians

def med(fn, n=20):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3

sc, cams, st = scenes.config_c4()
dev = "cuda"
pc = SyntheticGaussians(sc, dev)
P = sc.P
bg = torch.zeros(3, device=dev)
dL = scenes.grad_seed(cams[0].image_width, cams[0].image_height, 4).to(dev)
cam = cams[0].to(dev)
# backward with dense SH rows vs factored (factors + no SH row stores)
def bwd(factored):
    for p_ in pc.parameters(): p_.grad = None
    if factored:
        dgr.set_grad_sinks({}, sh_factor=factor)
    out = render(cam, pc, PIPE, bg, **st)
    tm = dgr._C.KernelTimer(); dgr._C.set_timer(tm)
    out["render"].backward(dL)
    dgr._C.set_timer(None); dgr.set_grad_sinks(None)
    torch.cuda.synchronize()
    return tm.read_ms()["preprocess_bwd"] * 1e3
factor = torch.empty(P, 3, device=dev)
for _ in range(3): bwd(False); bwd(True)
d = sorted(bwd(False) for _ in range(10))[5]; f = sorted(bwd(True) for _ in range(10))[5]
print(f"C4 view 0, 1 M Gaussians: per-Gaussian backward (K9 class) dense SH rows {d:.1f} us | factored (factor kernel + K9 without SH row stores) {f:.1f} us")
for n in (1, 2, 4, 8):
    rows = torch.randn(n, 3 * P + 4, device=dev)
    g_dc, g_rest = torch.empty(P, 1, 3, device=dev), torch.empty(P, 15, 3, device=dev)
    fn = lambda: dgr.sh_grad_from_views(pc._xyz.detach(), rows, n, 3, 1.0 / n, g_dc, g_rest)
    fn(); torch.cuda.synchronize()
    print(f"msgs_sh_grad_from_views, {n} views: {med(fn):.1f} us")
