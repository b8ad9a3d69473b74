"""Steady loop of one mode of the two-view comparison, for rocprofv3 --kernel-trace (not a test).
usage: two_view_loop.py <serial|piped|piped_shared|fwd_serial|fwd_piped|fwd_piped_shared> [rounds] [views] [config]
env: MSGS_TV_PRIO="0,-1" stream priorities, MSGS_TV_FUSED=1 raw-parameter entry"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
torch.autograd.set_multithreading_enabled(False)
from gaussian_renderer import PIPE, render, render_fused
from multi_view import ViewPipeline
from synthetic_model import SyntheticGaussians
mode = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = sys.argv[4] if len(sys.argv) > 4 else "C3"
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
fn = render_fused if os.environ.get("MSGS_TV_FUSED") == "1" else render
bwd = lambda i, pkg: pkg["render"].backward(dL)


def one():
    for p_ in pc.parameters():
        p_.grad = None
    if mode == "serial":
        for c in cams:
            fn(c, pc, PIPE, bg, **st)["render"].backward(dL)
    elif mode in ("piped", "piped_shared"):
        pipe.train_views(cams, pc, PIPE, bg, bwd, render_fn=fn, share_getters=mode.endswith("shared"), **st)
    else:
        with torch.no_grad():
            if mode == "fwd_serial":
                for c in cams:
                    fn(c, pc, PIPE, bg, **st)
            else:
                pipe.render_views(cams, pc, PIPE, bg, render_fn=fn, share_getters=mode.endswith("shared"), **st)


gc.collect(); gc.disable()
for _ in range(2):
    one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(rounds):
    one()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / rounds / n
print(f"{cfg} {mode} {1e3 * dt:.4f} ms/view", flush=True)
