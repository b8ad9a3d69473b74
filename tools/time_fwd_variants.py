"""Times the blend-forward kernel variants on a BASELINE config and reports the lane efficiency of each (counting replica).
usage: python tools/time_fwd_variants.py C3"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc, cam, st = scenes.config(cfg)
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
camd = cam.to("cuda"); bg = torch.zeros(3, device="cuda")
dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).to("cuda")
lib = dgr._C.lib
def step(t=None):
    dgr._C.set_timer(t)
    for p_ in pc.parameters(): p_.grad = None
    out = render(camd, pc, PIPE, bg, **st); out["render"].backward(dL); return out
for v in (1, 3, 4, 5, 6, 1, 5, 6):
    lib.msgs_set_forward_variant(v)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tm = dgr._C.KernelTimer(); K = 10
    for _ in range(K): out = step(tm)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    ctx = out["render"].grad_fn
    geom, binning, image, D = ctx.state
    scratch = torch.empty(256, dtype=torch.uint8, device="cuda"); o3 = (C.c_int64 * 7)()
    dgr._C.check(lib.msgs_blend_lane_stats(C.byref(ctx.call.view), C.c_void_p(geom.data_ptr()), geom.numel(), sc.P, int(D),
                                           C.c_void_p(binning.data_ptr()), binning.numel(), C.c_void_p(image.data_ptr()), image.numel(),
                                           C.c_void_p(scratch.data_ptr()), scratch.numel(), o3, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stats")
    ms = tm.read_ms()
    print(f"{cfg} variant {v}: step {dt*1e3:.3f} ms blend_fwd {ms['blend_fwd']*1e3:.1f} us blend_bwd {ms['blend_bwd']*1e3:.1f} us "
          f"trips {o3[0]} lane_eff {o3[2] / max(64 * o3[0], 1):.4f} | bwd visits {o3[3]} quad steps {o3[4]} lane_eff "
          f"{o3[5] / max(64 * o3[4], 1):.4f} visits with contribution {o3[6]}")
lib.msgs_set_forward_variant(0)
