import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from oracle import oracle_ctypes as oc
from parity_utils import check_backward, check_forward, hip_render, small_scene
cfgs = [dict(P=9000, W=180, H=42, deg=1, ms=True, fade=0.5, gran=2, bwd_gen=1, fwd_var=4, seed=869634),
        dict(P=1500, W=69, H=121, deg=3, ms=False, fade=1.0, gran=2, bwd_gen=1, fwd_var=0, seed=892520),
        dict(P=9000, W=22, H=181, deg=0, ms=True, fade=0.5, gran=0, bwd_gen=1, fwd_var=0, seed=264900),
        dict(P=64, W=12, H=127, deg=3, ms=True, fade=0.5, gran=0, bwd_gen=0, fwd_var=4, seed=964987)]
for c in cfgs:
    for variant in ("as drawn", "round-2 kernels only"):
        gran, bwd_gen, fwd_var = (c["gran"], c["bwd_gen"], c["fwd_var"]) if variant == "as drawn" else (1, 1, 1)
        sc, cam = small_scene(c["P"], c["W"], c["H"], c["seed"], sh_degree=c["deg"], multiscale=c["ms"],
                              **({"scale_k": 0.004 * 1920.0 / max(c["W"], 8) * 0.3} if c["ms"] else {}))
        st = dict(filter_small=c["ms"], filter_large=c["ms"], fade_size=c["fade"])
        bg = torch.rand(3, generator=torch.Generator().manual_seed(c["seed"]))
        dL = scenes.grad_seed(c["W"], c["H"], c["seed"] % 97)
        dgr._C.lib.msgs_set_blend_granularity(gran); dgr._C.lib.msgs_set_backward_generation(bwd_gen); dgr._C.lib.msgs_set_forward_variant(fwd_var)
        out, pc, m2 = hip_render(sc, cam, st, bg, dL)
        orc = oc.rasterize(pc.seen, cam, st, bg); og = oc.backward(orc, dL)
        try:
            w = check_backward(pc, m2, og, "x", flagged=orc.borderline_gaussians, rtol=1.0)
        except AssertionError as e:
            w = str(e)[:200]
        print(c["seed"], variant, {k: f"{v:.2e}" for k, v in w.items()} if isinstance(w, dict) else w, "borderline px %.4f" % orc.borderline.float().mean().item())
