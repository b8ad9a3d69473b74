"""Serial loop vs two-lane pipeline per pyramid level k of the C3 scene (8 views per sweep, training settings), ms per view.
usage: time_two_view_levels.py [k ...]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
torch.autograd.set_multithreading_enabled(False)
from gaussian_renderer import PIPE, render, render_fused
from multi_view import ViewPipeline
from synthetic_model import SyntheticGaussians
sc, _, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
lanes = int(os.environ.get("MSGS_TV_LANES", "2"))
pipe = ViewPipeline("cuda", n_streams=lanes)
n, rounds = 8, 5


def timed(fn):
    gc.collect(); gc.disable()
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / rounds / n


for k in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6]:
    W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
    cams = [scenes.front_camera(W, H).to("cuda")] * n
    dL = scenes.grad_seed(W, H, 40 + k).to("cuda")
    bwd = lambda i, pkg: pkg["render"].backward(dL)

    def zero():
        for p_ in pc.parameters():
            p_.grad = None

    def serial(fn=render):
        zero()
        for c in cams:
            fn(c, pc, PIPE, bg, **st)["render"].backward(dL)

    def piped(fn=render):
        zero()
        pipe.train_views(cams, pc, PIPE, bg, bwd, render_fn=fn, **st)
    a, b, c_, d = timed(serial), timed(piped), timed(lambda: serial(render_fused)), timed(lambda: piped(render_fused))
    with torch.no_grad():
        e = timed(lambda: [render(c, pc, PIPE, bg, **st) for c in cams])
        f = timed(lambda: pipe.render_views(cams, pc, PIPE, bg, **st))
    print(f"k={k} {W}x{H}: fwd+bwd serial {a:.3f} -> {lanes} lanes {b:.3f} | fused {c_:.3f} -> {d:.3f} | forward-only {e:.3f} -> {f:.3f} ms/view", flush=True)
