"""EXPERIMENT: phase times of occ_pass_kernel (needs the MSGS_X_OCC_TIMES hooks compiled in): python occ_times.py C5|C3|C3off"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["MSGS_X_OCC_TIMES"] = "1"
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
off = name.endswith("off")
scene, cam, st = scenes.config(name[:2])
if off:
    st = dict(filter_small=False, filter_large=False, fade_size=1.0)
dev = torch.device("cuda")
pc = SyntheticGaussians(scene, dev, requires_grad=True)
cam, bg = cam.to(dev), torch.zeros(3, device=dev)
for it in range(4):
    out = render(cam, pc, PIPE, bg, **st)
    torch.cuda.synchronize()
    ctx = out["render"].grad_fn
    geom = dgr._resolve(ctx.state)[0]
    o = (C.c_int64 * 8)()
    dgr._C.check(dgr._C.lib.msgs_occlusion_stats(C.c_void_p(geom.data_ptr()), geom.numel(), ctx.call.P, o,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "msgs_occlusion_stats")
    print(name, "heavy", int(o[1]), "candidates", int(o[2]), "closed blocks", int(o[3]), "of", int(o[4]))
    del out, ctx
