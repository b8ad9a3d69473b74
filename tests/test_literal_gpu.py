"""The VERIFICATION mode (msgs_set_deterministic; ms-gs_amd/csrc/literal.hip) against the float32 CPU oracle evaluated the same way
(oracle_ctypes.exp_double: exp() in double, rounded to float once — on both sides).  The mode restates the reference's blend
loops literally (float32, no FMA contraction, SURVEY App. A.2 / A.3 as the oracle restates them), sums the nine per-(pixel,
Gaussian) products in double in a fixed order and hands the textbook sums to the per-Gaussian backward.  Both sides then take the
same float for every alpha and every term, and the north star's sentence — "forward <= 1e-5 abs per pixel, backward grads <= 1e-4
rel on identical inputs" — is asserted AS WRITTEN: all seven gradient tensors, over ALL Gaussians and ALL pixels (no borderline
exclusions, no per-config ceilings), at every BASELINE config: C2, C3, C5 and the C4 ring views.  (The default mode, a different
float32 evaluation of the same algorithm, keeps its three-way property against the float64 truth: tests/test_fullsize_gpu.py.)"""
import pytest
import torch

import scenes
from parity_utils import hip_render, leaf_space, rel_err, report

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_BWD = 1e-5, 1e-4                    # the north star's sentence
GUARD_FWD, GUARD_BWD = 1e-6, 1e-5                # regression guards an order below it (measured: forward bit-identical at every
                                                 # config, gradients <= 2.9e-7; dL/dSH and dL/dmeans2D bit-identical)


@pytest.fixture()
def literal():
    import diff_gaussian_rasterization as dgr
    prev = dgr.set_deterministic(True)
    yield
    dgr.set_deterministic(prev)


def _compare(name, sc, cam, st, bg, dL):
    from oracle import oracle_ctypes as oc
    out, pc, m2 = hip_render(sc, cam, st, bg, dL)
    with oc.exp_double():
        orc = oc.rasterize(pc.seen, cam, st, bg)
        og = oc.backward(orc, dL)
    col = out["render"].detach().cpu()
    d = (col - orc.color).abs()
    n_diff = int((col != orc.color).sum().item())
    report(name, "literal forward: max |HIP - oracle| over ALL pixels", d.max().item())
    report(name, "literal forward: pixel channels that differ at all (fraction)", n_diff / col.numel())
    assert d.max().item() <= TOL_FWD, (name, d.max().item())
    assert d.max().item() <= GUARD_FWD, (name, d.max().item())
    for key, ref in (("acc_pixel_size", orc.acc_pixel_size), ("depth", orc.depth)):
        dd = (out[key].detach().cpu() - ref).abs().max().item()
        assert dd <= TOL_FWD * max(ref.abs().max().item(), 1.0), (name, key, dd)
    assert torch.equal(out["radii"].cpu(), orc.radii), name
    worst = {}
    for k, (got, ref) in leaf_space(pc, m2, og).items():
        worst[k] = rel_err(got, ref)                                   # every row
        report(name, f"literal grad {k}: max-norm rel err over ALL Gaussians", worst[k])
    for k, v in worst.items():
        assert v <= TOL_BWD, f"{name}: grad {k} rel err {v:.3e} > {TOL_BWD} ({worst})"
        assert v <= GUARD_BWD, f"{name}: grad {k} rel err {v:.3e} above its regression guard {GUARD_BWD} ({worst})"
    return worst


@pytest.mark.parametrize("ms,P,W,H,seed", [(False, 3000, 200, 136, 3), (True, 8000, 321, 203, 4), (True, 1, 64, 48, 5),
                                         (False, 20000, 640, 400, 6)])
def test_small_scenes_every_gaussian_every_pixel(literal, ms, P, W, H, seed):
    sc = scenes.frustum_scene(P, W, H, seed=seed, multiscale=ms, scale_k=0.004 * 1920.0 / W * 0.5)
    st = dict(filter_small=ms, filter_large=ms, fade_size=0.0 if seed % 2 else 0.5)
    _compare(f"literal small {seed}", sc, scenes.front_camera(W, H), st, torch.tensor([0.1, 0.4, 0.7]), scenes.grad_seed(W, H, seed))


def test_config_c2(literal):
    sc, cam, st = scenes.config("C2")
    _compare("literal C2", sc, cam, st, torch.zeros(3), scenes.grad_seed(cam.image_width, cam.image_height, 1))


def test_config_c3(literal):
    sc, cam, st = scenes.config("C3")
    _compare("literal C3", sc, cam, st, torch.zeros(3), scenes.grad_seed(cam.image_width, cam.image_height, 2))


def test_config_c5(literal):
    sc, cam, st = scenes.config("C5")
    _compare("literal C5", sc, cam, st, torch.zeros(3), scenes.grad_seed(cam.image_width, cam.image_height, 5))


@pytest.mark.parametrize("v", [0, 3, 6])
def test_config_c4_ring_views(literal, v):
    sc, cams, st = scenes.config_c4()
    cam = cams[v]
    _compare(f"literal C4v{v}", sc, cam, st, torch.tensor([0.0, 0.0, 0.0]), scenes.grad_seed(cam.image_width, cam.image_height, 10 + v))


def test_bitwise_reproducible_and_mode_must_not_change_under_a_graph(literal):
    import diff_gaussian_rasterization as dgr
    W, H = 300, 200
    sc = scenes.frustum_scene(5000, W, H, seed=9, scale_k=0.004 * 1920.0 / W * 0.5)
    cam, st, bg, dL = scenes.front_camera(W, H), dict(filter_small=False, filter_large=False, fade_size=1.0), torch.zeros(3), \
        scenes.grad_seed(W, H, 9)
    a = hip_render(sc, cam, st, bg, dL)
    b = hip_render(sc, cam, st, bg, dL)
    assert torch.equal(a[0]["render"], b[0]["render"]) and torch.equal(a[2], b[2])
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(a[1], n).grad, getattr(b[1], n).grad), n
    # a forward of one mode cannot be finished by the backward of the other
    from gaussian_renderer import render
    from parity_utils import PIPE
    from synthetic_model import SyntheticGaussians
    pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
    out = render(cam.to("cuda"), pc, PIPE, bg.cuda(), **st)
    dgr.set_deterministic(False)
    try:
        with pytest.raises(RuntimeError, match="changed between"):
            out["render"].backward(dL.cuda())
    finally:
        dgr.set_deterministic(True)


def test_the_double_exp_moves_the_oracle_by_an_ulp_at_most():
    """glibc's expf is not correctly rounded (0.06 % of arguments differ from exp() in double rounded once): the oracle under
    exp_double is the SAME algorithm with a better-defined exponential.  Its C3 image stays within float rounding of the
    default oracle's — the checker every other parity test of this tree uses is unchanged."""
    from oracle import oracle_ctypes as oc
    sc, cam, st = scenes.config("C3")
    bg = torch.zeros(3)
    a = oc.rasterize(sc, cam, st, bg)
    with oc.exp_double():
        b = oc.rasterize(sc, cam, st, bg)
    d = (a.color - b.color).abs()
    changed = (a.color != b.color).float().mean().item()
    report("oracle C3", "pixel channels the double exp changes (fraction)", changed)
    report("oracle C3", "max |change|", d.max().item())
    assert torch.equal(a.radii, b.radii)
    assert d.max().item() <= 3e-7 and changed < 0.05
