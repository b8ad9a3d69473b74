"""Recognition of the reference's getters (GaussianRasterizer.forward -> _RasterizeGaussiansChained): same forward bit
for bit, gradients of the leaf parameters equal to what autograd produces through the getters' own backward."""
import pytest
import torch

import scenes
from parity_utils import PIPE, rel_err, rel_err_reported, small_scene

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
# The rasterizer's gradient sums are exact across tiles (double accumulators), so the chained and the plain path differ only
# in the getter backward itself (sigmoid / normalize in the kernel vs in autograd): measured 0 on xyz / SH / scaling,
# <= 2.3e-7 on opacity / rotation, at every size including C3.
HIP_VS_HIP_RTOL = 2e-6
C3_RTOL = 2e-6


def _run(sc, cam, st, bg, dL, chain, pipe=PIPE):
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    prev = dgr.chain_reference_getters
    dgr.chain_reference_getters = chain
    try:
        pc = SyntheticGaussians(sc, "cuda")
        out = render(cam.to("cuda"), pc, pipe, bg.cuda(), **st)
        used = type(out["render"].grad_fn).__name__
        out["render"].backward(dL.cuda())
        torch.cuda.synchronize()
    finally:
        dgr.chain_reference_getters = prev
    return out, pc, used


@pytest.mark.parametrize("P,W,H,seed,deg,ms", [(500, 64, 48, 1, 3, False), (6000, 160, 128, 2, 3, True),
                                                 (4000, 130, 70, 3, 1, False), (3000, 96, 96, 4, 0, True)])
def test_chained_equals_autograd_through_the_getters(P, W, H, seed, deg, ms):
    sc, cam = small_scene(P, W, H, 300 + seed, sh_degree=deg, multiscale=ms,
                          **({"scale_k": 0.004 * 1920.0 / W * 0.2} if ms else {}))
    sc.rotations = sc.rotations * (0.5 + torch.rand(P, 1, generator=torch.Generator().manual_seed(seed)))   # un-normalised
    st = dict(filter_small=ms, filter_large=ms, fade_size=0.0 if ms else 1.0)
    bg = torch.tensor([0.2, 0.1, 0.6])
    dL = scenes.grad_seed(W, H, seed)
    a, pa, ua = _run(sc, cam, st, bg, dL, chain=False)
    b, pb, ub = _run(sc, cam, st, bg, dL, chain=True)
    assert ua == "_RasterizeGaussiansBackward" and ub == "_RasterizeGaussiansChainedBackward"
    for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
        assert torch.equal(a[k], b[k]), k                     # the forward is the same kernel on the same inputs
    for n in LEAVES:
        ga, gb = getattr(pa, n).grad, getattr(pb, n).grad
        assert gb.shape == ga.shape and gb.is_contiguous()
        assert rel_err_reported(f"chain P={P}", n, gb, ga) <= HIP_VS_HIP_RTOL, n
    assert rel_err_reported(f"chain P={P}", "means2D", b["viewspace_points"].grad, a["viewspace_points"].grad) <= HIP_VS_HIP_RTOL


def test_patterns_that_must_not_be_chained_take_the_plain_path():
    import types
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    sc, cam = small_scene(300, 48, 40, 9)
    bg = torch.zeros(3).cuda()
    camd = cam.to("cuda")
    pc = SyntheticGaussians(sc, "cuda")
    for pipe in (types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False),
                 types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=True, debug=False)):
        assert type(render(camd, pc, pipe, bg)["render"].grad_fn).__name__ == "_RasterizeGaussiansBackward"
    out = render(camd, pc, PIPE, bg, override_color=torch.rand(300, 3).cuda())
    assert type(out["render"].grad_fn).__name__ == "_RasterizeGaussiansBackward"
    with torch.no_grad():
        assert render(camd, pc, PIPE, bg)["render"].grad_fn is None
    # a getter that is not the reference's (scaled exp): plain path, and its own backward still runs
    class Scaled(SyntheticGaussians):
        @property
        def get_scaling(self):
            return torch.exp(self._scaling) * 1.0
    pc2 = Scaled(sc, "cuda")
    out = render(camd, pc2, PIPE, bg)
    assert type(out["render"].grad_fn).__name__ == "_RasterizeGaussiansBackward"
    out["render"].sum().backward()
    assert pc2._scaling.grad is not None


def test_c3_fullsize_chained_vs_plain():
    sc, cam, st = scenes.config("C3")
    bg = torch.zeros(3)
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 2)
    a, pa, _ = _run(sc, cam, st, bg, dL, chain=False)
    b, pb, ub = _run(sc, cam, st, bg, dL, chain=True)
    assert ub == "_RasterizeGaussiansChainedBackward"
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
    for n in LEAVES:
        assert rel_err_reported("chain C3", n, getattr(pb, n).grad, getattr(pa, n).grad) <= C3_RTOL, n


def test_gradient_sinks_deliver_into_the_exchange_bucket():
    """PipelinedGradExchange(direct=True): the backward writes the leaf gradients straight into the flat bucket and
    autograd adopts those aliases as .grad — same values as the ordinary path, no zero-fill / accumulation pass."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, render_fused
    from synthetic_model import SyntheticGaussians
    from view_parallel import PipelinedGradExchange
    W, H = 128, 96
    sc, cam = small_scene(3000, W, H, 41)
    bg = torch.zeros(3).cuda()
    camd = cam.to("cuda")
    dL = scenes.grad_seed(W, H, 41).cuda()
    ref = SyntheticGaussians(sc, "cuda")
    render(camd, ref, PIPE, bg)["render"].backward(dL)
    for fn in (render, render_fused):
        pc = SyntheticGaussians(sc, "cuda")
        ex = PipelinedGradExchange(pc.parameters(), world=2, direct=True)      # no process group: local average only
        for k in range(3):                                                    # both buckets, and reuse of the first
            ex.begin_view()
            assert all(p.grad is None for p in pc.parameters())
            fn(camd, pc, PIPE, bg)["render"].backward(dL)
            b = ex.end_view()
            for p, v in zip(pc.parameters(), b.views):
                assert p.grad.data_ptr() == v.data_ptr()                      # adopted, not copied
        ex.drain()
        assert not dgr._sinks.grad
        for n in LEAVES:
            got, want = getattr(pc, n).grad, getattr(ref, n).grad / 2
            # render_fused evaluates the activations in the kernels, 1 ulp from torch's (tests/test_fused_gpu.py)
            assert rel_err(got, want) <= (2e-4 if fn is render_fused else HIP_VS_HIP_RTOL), (fn.__name__, n)
    # the plain path cannot deliver into the bucket: loud failure instead of silently missing gradients
    pc = SyntheticGaussians(sc, "cuda")
    ex = PipelinedGradExchange(pc.parameters(), world=2, direct=True)
    dgr.chain_reference_getters = False
    try:
        ex.begin_view()
        render(camd, pc, PIPE, bg)["render"].backward(dL)
        with pytest.raises(RuntimeError, match="did not land in the bucket"):
            ex.end_view()
    finally:
        dgr.chain_reference_getters = True
        dgr.set_grad_sinks(None)
