"""Two-view pipeline vs the serial loop at a BASELINE config (not a test): ms per view, fwd+bwd and forward-only.
usage: time_two_view.py [C3|C4] [views] [rounds]    env: MSGS_TV_PRIO="0,-1" stream priorities"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
torch.autograd.set_multithreading_enabled(False)
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render, render_fused
from multi_view import ViewPipeline
from synthetic_model import SyntheticGaussians
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if cfg == "C4":
    sc, cams, st = scenes.config_c4()
    cams = [c.to("cuda") for c in cams][:n]
else:
    sc, cam, st = scenes.config(cfg)
    cams = [cam.to("cuda")] * n
W, H = cams[0].image_width, cams[0].image_height
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
dL = scenes.grad_seed(W, H, 5).to("cuda")
prio = [int(x) for x in os.environ["MSGS_TV_PRIO"].split(",")] if "MSGS_TV_PRIO" in os.environ else None
lanes = int(os.environ.get("MSGS_TV_LANES", "2"))
pipe = ViewPipeline("cuda", n_streams=len(prio) if prio else lanes, priorities=prio)
import gc


def timed(fn, what):
    gc.collect(); gc.disable()
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / rounds / n
    print(f"{cfg} {what:58s} {1e3 * dt:.4f} ms/view", flush=True)
    return dt


def zero():
    for p_ in pc.parameters():
        p_.grad = None


def serial(fn=render):
    zero()
    for c in cams:
        fn(c, pc, PIPE, bg, **st)["render"].backward(dL)


def piped(share, fn=render):
    zero()
    pipe.train_views(cams, pc, PIPE, bg, lambda i, pkg: pkg["render"].backward(dL), render_fn=fn, share_getters=share, **st)


timed(serial, "serial fwd+bwd (reference pattern)")
timed(lambda: piped(False), "two streams fwd+bwd, getters per view")
timed(lambda: piped(True), "two streams fwd+bwd, shared getters")
timed(lambda: serial(render_fused), "serial fwd+bwd, fused entry")
timed(lambda: piped(False, render_fused), "two streams fwd+bwd, fused entry")
with torch.no_grad():
    timed(lambda: [render(c, pc, PIPE, bg, **st) for c in cams], "serial forward-only")
    timed(lambda: pipe.render_views(cams, pc, PIPE, bg, share_getters=False, **st), "two streams forward-only, getters per view")
    timed(lambda: pipe.render_views(cams, pc, PIPE, bg, share_getters=True, **st), "two streams forward-only, shared getters")
    timed(lambda: pipe.render_views(cams, pc, PIPE, bg, render_fn=render_fused, share_getters=False, **st), "two streams forward-only, fused entry")
