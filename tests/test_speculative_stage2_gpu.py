"""msgs_forward launches stage 2 speculatively on buffers sized from the previous frame's instance count (include/msgs.h).
The cases around that guess: it suffices (normal), it is exceeded (the scene outgrew the margin: the truncated stage 2 must be
redone on exact buffers and leave no trace), there is none yet (first frame), and nothing is rendered at all."""
import pytest
import torch

import scenes
from parity_utils import PIPE, small_scene

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
ST = dict(filter_small=False, filter_large=False, fade_size=1.0)


def _run(sc, cam, dL, mod, bg):
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    pc = SyntheticGaussians(sc, "cuda")
    out = render(cam.to("cuda"), pc, PIPE, bg.cuda(), scaling_modifier=mod, **ST)
    out["render"].backward(dL.cuda())
    D = out["render"].grad_fn.state[3]
    return out, {n: getattr(pc, n).grad.clone() for n in LEAVES}, D


def test_guess_exceeded_then_redone_equals_a_fresh_render():
    import diff_gaussian_rasterization as dgr
    W, H = 400, 300
    sc, cam = small_scene(40000, W, H, 71, scale_k=0.004 * 1920.0 / W * 0.4)
    dL = scenes.grad_seed(W, H, 71)
    bg = torch.tensor([0.1, 0.0, 0.3])
    dgr._last_instances.clear()
    a, ga, Da = _run(sc, cam, dL, 0.3, bg)            # first frame: no guess, sequential route
    b, gb, Db = _run(sc, cam, dL, 0.3, bg)            # guess == D: speculative stage 2 stands
    assert Da == Db and torch.equal(a["render"], b["render"]) and all(torch.equal(ga[n], gb[n]) for n in LEAVES)
    c, gc, Dc = _run(sc, cam, dL, 1.5, bg)            # footprints 5x larger: D far beyond the 12.5 % margin -> redone
    assert Dc > 1.5 * Db
    dgr._last_instances.clear()
    d, gd, Dd = _run(sc, cam, dL, 1.5, bg)            # fresh: exact buffers from the start
    assert Dc == Dd
    for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
        assert torch.equal(c[k], d[k]), k
    assert all(torch.equal(gc[n], gd[n]) for n in LEAVES)
    e, ge, De = _run(sc, cam, dL, 0.3, bg)            # and back: a guess far too large is fine
    assert De == Da and torch.equal(e["render"], a["render"]) and all(torch.equal(ge[n], ga[n]) for n in LEAVES)


def test_nothing_rendered_with_a_stale_guess():
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    W, H = 128, 96
    sc, cam = small_scene(5000, W, H, 72)
    bg = torch.tensor([0.2, 0.4, 0.6])
    dgr._last_instances.clear()
    pc = SyntheticGaussians(sc, "cuda")
    render(cam.to("cuda"), pc, PIPE, bg.cuda(), **ST)                      # sets the guess
    with torch.no_grad():
        pc._xyz[:, 2] = -5.0                                                 # everything behind the camera
    out = render(cam.to("cuda"), pc, PIPE, bg.cuda(), **ST)
    assert int((out["radii"] > 0).sum()) == 0 and out["render"].grad_fn.state[3] == 0
    assert torch.equal(out["render"], bg.cuda().view(3, 1, 1).expand(3, H, W))
    out["render"].sum().backward()
    assert all(float(getattr(pc, n).grad.abs().max()) == 0.0 for n in LEAVES)

