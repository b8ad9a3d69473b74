"""True GPU timeline of the two-stream pipeline without a profiler (rocprofv3 makes this loop host-bound and distorts it):
every view gets its own set of the library's per-kernel-class HIP events (msgs_timing_t), all read against ONE base event.
usage: two_view_timeline.py [piped|piped_shared|serial] [views] [config]   env MSGS_TV_PRIO, MSGS_TV_FUSED, MSGS_TV_AUTOGRAD_ACC=1"""
import ctypes as C, gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
torch.autograd.set_multithreading_enabled(False)
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render, render_fused
from multi_view import ViewPipeline
from synthetic_model import SyntheticGaussians
mode = sys.argv[1] if len(sys.argv) > 1 else "piped_shared"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = sys.argv[3] if len(sys.argv) > 3 else "C3"
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
base_fn = render_fused if os.environ.get("MSGS_TV_FUSED") == "1" else render
acc_kernel = os.environ.get("MSGS_TV_AUTOGRAD_ACC") != "1"
timers = [dgr._C.KernelTimer() for _ in range(n)]
cam_index = {}


def fn(c, *a, **k):
    i = fn.next
    fn.next += 1
    dgr._C.set_timer(timers[i] if fn.timed else None)
    return base_fn(c, *a, **k)


def bwd(i, pkg):
    dgr._C.set_timer(timers[i] if fn.timed else None)
    pkg["render"].backward(dL)


def one(timed):
    fn.next, fn.timed = 0, timed
    for p_ in pc.parameters():
        p_.grad = None
    if mode == "serial":
        for i, c in enumerate(cams):
            bwd(i, {"render": fn(c, pc, PIPE, bg, **st)["render"]})
    else:
        pipe.train_views(cams, pc, PIPE, bg, bwd, render_fn=fn, share_getters=mode.endswith("shared"),
                         accumulate_in_kernel=acc_kernel, **st)
    dgr._C.set_timer(None)


gc.collect(); gc.disable()
for _ in range(3):
    one(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    one(False)
torch.cuda.synchronize(); print(f"{cfg} {mode}: {(time.perf_counter() - t0) / 3 / n * 1e3:.4f} ms/view untimed")
torch.cuda.synchronize(); t0 = time.perf_counter()
one(True)
torch.cuda.synchronize(); print(f"{cfg} {mode}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/view with the events")
hip = C.CDLL("libamdhip64.so")
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
base = timers[0].t.ev[0]


def at(ev):
    ms = C.c_float(0)
    return ms.value * 1e3 if hip.hipEventElapsedTime(C.byref(ms), base, ev) == 0 else float("nan")


names = dgr._C.K_NAMES
print("times in us from the first kernel of view 0; per view: class [start, end)")
for i, t in enumerate(timers):
    row = []
    for k, nm in enumerate(names):
        a, b = at(t.t.ev[2 * k]), at(t.t.ev[2 * k + 1])
        row.append(f"{nm} {a:7.0f}-{b:7.0f}")
    print(f"view {i} (stream {i % len(pipe.streams)}): " + " | ".join(row))
