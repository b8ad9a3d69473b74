"""Non-finite parameters (a diverging training run hands the rasterizer NaN / Inf before anyone notices): the op must neither hang nor
write outside its buffers.  A Gaussian whose POSITION, SCALE or ROTATION is NaN, or whose position is infinite, gets no tile (its
rect is empty: every float -> int conversion of a NaN bound gives 0) and radii 0 — so the image and the gradients of everything
else are bit-for-bit those of the scene without it.  NaN opacity / NaN SH / infinite scale propagate into the pixels they cover, as
in the reference's arithmetic; for those only "returns, with a sane instance count" is asserted."""
import pytest
import torch

import scenes
from parity_utils import PIPE, small_scene

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
ST = dict(filter_small=False, filter_large=False, fade_size=1.0)


def _run(fn, model, cam, bg, dL):
    out = fn(cam, model, PIPE, bg, **ST)
    out["render"].backward(dL)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("entry", ["reference", "raw"])
def test_gaussians_with_nonfinite_geometry_are_skipped(entry):
    from gaussian_renderer import render, render_fused
    from synthetic_model import SyntheticGaussians
    fn = render if entry == "reference" else render_fused
    W, H = 240, 136
    P = 6000
    sc, cam = small_scene(P, W, H, 31, sh_degree=2)
    dev = torch.device("cuda")
    camd, bg, dL = cam.to(dev), torch.tensor([0.2, 0.3, 0.4], device=dev), scenes.grad_seed(W, H, 31).to(dev)
    bad = torch.zeros(P, dtype=torch.bool)
    bad[torch.randperm(P, generator=torch.Generator().manual_seed(1))[:600]] = True
    idx = bad.nonzero().flatten()
    nan, inf = float("nan"), float("inf")
    poisoned = SyntheticGaussians(sc, dev)
    with torch.no_grad():
        kinds = idx.split(100)
        poisoned._xyz[kinds[0].to(dev), 0] = nan
        poisoned._xyz[kinds[1].to(dev), 2] = nan
        poisoned._xyz[kinds[2].to(dev), 1] = inf
        poisoned._xyz[kinds[3].to(dev), 2] = inf
        poisoned._scaling[kinds[4].to(dev), 1] = nan
        poisoned._rotation[kinds[5].to(dev), 2] = nan
    out = _run(fn, poisoned, camd, bg, dL)
    assert not out["radii"][bad.to(dev)].any()
    clean = SyntheticGaussians(sc.subset((~bad).nonzero().flatten()), dev)
    ref = _run(fn, clean, camd, bg, dL)
    good = (~bad).to(dev)
    assert torch.isfinite(out["render"]).all()
    for k in ("render", "acc_pixel_size", "depth"):
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(out["radii"][good], ref["radii"])
    for n in LEAVES:
        g, g_ref = getattr(poisoned, n).grad[good], getattr(clean, n).grad
        scale = g_ref.abs().max().item()
        assert torch.isfinite(g).all(), n
        assert (g - g_ref).abs().max().item() <= 1e-6 * scale, n


def test_nonfinite_appearance_or_infinite_scale_returns():
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    import diff_gaussian_rasterization as dgr
    W, H = 240, 136
    P = 6000
    sc, cam = small_scene(P, W, H, 32, sh_degree=1)
    dev = torch.device("cuda")
    camd, bg, dL = cam.to(dev), torch.zeros(3, device=dev), scenes.grad_seed(W, H, 32).to(dev)
    nan, inf = float("nan"), float("inf")
    m = SyntheticGaussians(sc, dev)
    with torch.no_grad():
        m._opacity[0:50] = nan
        m._opacity[50:100] = inf
        m._features_dc[100:150] = nan
        m._features_rest[150:200, 1] = inf
        m._scaling[200:220, 0] = inf           # exp(inf): an infinite covariance -> the whole screen, NaN conic
        m._scaling[220:240] = -inf             # exp(-inf) = 0: a point; the 0.3 px dilation keeps it a valid Gaussian
        m._rotation[240:260] = 0.0             # normalize(0) = 0: a zero covariance + dilation
        m._xyz[260:280] = 1e30
    out = _run(render, m, camd, bg, dL)
    D = [v for k, v in dgr._last_instances.items() if k[1:4] == (P, W, H)]
    assert D and 0 < D[-1] <= P * ((W + 15) // 16) * ((H + 15) // 16)
    assert out["render"].shape == (3, H, W) and int((out["radii"] > 0).sum()) > P // 4
    # and the next, clean call is unaffected
    m2 = SyntheticGaussians(sc, dev)
    a = _run(render, m2, camd, bg, dL)["render"].clone()
    b = _run(render, SyntheticGaussians(sc, dev), camd, bg, dL)["render"]
    assert torch.equal(a, b) and torch.isfinite(a).all()
