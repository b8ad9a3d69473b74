"""Accuracy of the three backward paths (deterministic, gen-1 atomic, gen-2 atomic) against the float32 C++ oracle on the
scene of tests/test_deterministic_gpu.py::test_bitwise_reproducible_and_close_to_atomic_mode, and against each other.
Not a test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from parity_utils import hip_render, small_scene, check_backward, rel_err
from oracle import oracle_ctypes as oc

seeds = [int(a) for a in sys.argv[1:]] or [31, 32, 33]
for seed in seeds:
    W, H = 160, 128
    sc, cam = small_scene(6000, W, H, seed)
    st = dict(filter_small=False, filter_large=False, fade_size=1.0)
    bg = torch.zeros(3)
    dL = scenes.grad_seed(W, H, 3)
    res = {}
    for name, det, gen in (("det", True, 0), ("gen1", False, 1), ("gen2", False, 2)):
        dgr.set_deterministic(det)
        dgr._C.lib.msgs_set_backward_generation(gen)
        out, pc, m2 = hip_render(sc, cam, st, bg, dL)
        if "orc" not in res:
            res["orc"] = oc.rasterize(pc.seen, cam, st, bg)
            res["og"] = oc.backward(res["orc"], dL)
        try:
            worst = check_backward(pc, m2, res["og"], name, rtol=1.0, flagged=res["orc"].borderline_gaussians)
        except AssertionError as e:
            worst = str(e)
        res[name] = {k: getattr(pc, k).grad.clone() for k in ("_xyz", "_scaling", "_rotation", "_opacity")}
        print(seed, name, "vs oracle:", {k: f"{v:.2e}" for k, v in worst.items()} if isinstance(worst, dict) else worst)
    for a, b in (("det", "gen1"), ("det", "gen2"), ("gen1", "gen2")):
        print(seed, a, b, {k: f"{rel_err(res[a][k], res[b][k]):.2e}" for k in res[a]})
dgr.set_deterministic(False); dgr._C.lib.msgs_set_backward_generation(0)
