"""Maximum sizes: a model whose tensors pass 2^31 ELEMENTS (48 M Gaussians x 48 SH floats = 2.3e9) and 2^32 BYTES
(SH rows 9.2 GB, gradient records 3.8 GB, geometry 3.8 GB), so that every row offset in the per-Gaussian kernels, the depth
sort's compaction and the index lists has to be 64-bit clean.  No oracle runs at this size; the property checked is
EMBEDDING: 30 000 live Gaussians placed at the first rows, around the middle and at the LAST rows of the 48 M (all other
rows behind the camera) must render bit-identically to the 30 000 alone — same (tile, depth, index) order, same per-Gaussian
arithmetic — and their gradient rows must agree (the float64 atomics of the blend backward make the last bits
order-dependent: 1e-6 of the tensor's max norm), with every other row of every gradient exactly zero.
Both entries: the reference API (getters chained by the op) and the raw-parameter entry."""
import pytest
import torch

import scenes
from parity_utils import PIPE, small_scene

pytestmark = pytest.mark.gpu

P_BIG = 48_000_000
BLOCK = 10_000
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
STATE = ("_occ_multiplier", "_dc_delta", "max_pixel_sizes", "min_pixel_sizes", "base_gaussian_mask")


def _embed(small, rows, P):
    """a model of P Gaussians whose rows `rows` are `small`'s and whose other rows lie behind the camera"""
    from synthetic_model import SyntheticGaussians
    big = object.__new__(SyntheticGaussians)
    big.max_sh_degree, big.active_sh_degree = small.max_sh_degree, small.active_sh_degree
    for name in LEAVES + STATE:
        src = getattr(small, name)
        t = torch.zeros((P,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        if name == "_xyz":
            t[:, 2] = -1.0                                     # p_view.z <= 0.2: culled by K1
        elif name in ("_occ_multiplier",):
            t.fill_(1.0)
        elif name in ("max_pixel_sizes", "min_pixel_sizes"):
            t.fill_(-1.0)
        t[rows] = src.detach()
        setattr(big, name, torch.nn.Parameter(t) if name in LEAVES else t)
    return big


def _run(fn, model, cam, bg, dL, st):
    for p in model.parameters():
        p.grad = None
    out = fn(cam, model, PIPE, bg, **st)
    out["render"].backward(dL)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("entry", ["reference", "raw"])
def test_rows_beyond_2_pow_31_elements(entry):
    from gaussian_renderer import render, render_fused
    from synthetic_model import SyntheticGaussians
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip(f"needs ~70 GB of free HBM, {free / 2 ** 30:.0f} GiB available")
    fn = render if entry == "reference" else render_fused
    W, H = 480, 270
    n = 3 * BLOCK
    sc, cam = small_scene(n, W, H, 77, sh_degree=3, multiscale=True, scale_k=0.004 * 1920.0 / W * 0.3)
    st = dict(filter_small=True, filter_large=True, fade_size=0.0)
    dev = torch.device("cuda")
    camd, bg, dL = cam.to(dev), torch.tensor([0.1, 0.2, 0.3], device=dev), scenes.grad_seed(W, H, 77).to(dev)
    small = SyntheticGaussians(sc, dev)
    ref = _run(fn, small, camd, bg, dL, st)
    ref_grads = [p.grad.clone() for p in small.parameters()]
    ref_m2 = ref["viewspace_points"].grad.clone()
    assert int((ref["radii"] > 0).sum()) > n // 4            # the property is about rendered rows

    rows = torch.cat([torch.arange(0, BLOCK), torch.arange(P_BIG // 2 - 5, P_BIG // 2 - 5 + BLOCK),
                      torch.arange(P_BIG - BLOCK, P_BIG)]).to(dev)
    big = _embed(small, rows, P_BIG)
    assert big._features_rest.numel() > 2 ** 31 and big._features_rest.numel() * 4 > 2 ** 32
    out = _run(fn, big, camd, bg, dL, st)
    for key in ("render", "acc_pixel_size", "depth"):
        assert torch.equal(out[key], ref[key]), key
    assert torch.equal(out["radii"][rows], ref["radii"]) and torch.equal(out["pixel_sizes"][rows], ref["pixel_sizes"])
    other = torch.ones(P_BIG, dtype=torch.bool, device=dev)
    other[rows] = False
    assert not out["radii"][other].any() and not out["pixel_sizes"][other].any()
    pairs = [(p.grad, g) for p, g in zip(big.parameters(), ref_grads)] + [(out["viewspace_points"].grad, ref_m2)]
    for k, (g, g_ref) in enumerate(pairs):
        assert g is not None and g.shape[0] == P_BIG
        scale = g_ref.abs().max().item()
        assert scale > 0
        assert (g[rows] - g_ref).abs().max().item() <= 1e-6 * scale, k
        nz = torch.count_nonzero(g.reshape(P_BIG, -1), dim=1)  # unrendered rows: exact zeros, also beyond 2^31 elements
        assert not nz[other].any(), k
    del big, out, pairs
    torch.cuda.empty_cache()


import contextlib


@contextlib.contextmanager
def _occlusion(on):
    """the occlusion cut-off (msgs_set_occlusion) sheds exactly the instances these scenes are made of: the index-width tests
    switch it off, and one test shows what it does to the same scene"""
    import diff_gaussian_rasterization as dgr
    prev = dgr._C.lib.msgs_set_occlusion(1 if on else 0)
    try:
        yield
    finally:
        dgr._C.lib.msgs_set_occlusion(prev)


def test_more_than_2_pow_32_instances_is_an_error_not_a_wrap():
    with _occlusion(False):
        _more_than_2_pow_32_instances_is_an_error_not_a_wrap()


def test_the_occlusion_cut_off_renders_the_scene_that_has_too_many_instances():
    """the same 4.5e9-instance scene with the cut-off on (the default): every pixel terminates after 14 of the 140 000 covers, the
    instances behind that depth are never counted, and the call SUCCEEDS — image and depth bit-equal to the 100 nearest alone"""
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    W, H = 3840, 2160
    n, near = 140_000, 100
    sc, cam = small_scene(n, W, H, 5, sh_degree=0)
    sc.means3D[:, 0] = 0.0
    sc.means3D[:, 1] = 0.0
    sc.means3D[:, 2] = torch.linspace(4.0, 6.0, n)
    sc.scales[:] = 50.0
    sc.opacities[:] = 0.5
    dev = torch.device("cuda")
    camd, bg = cam.to(dev), torch.zeros(3, device=dev)
    st = dict(filter_small=False, filter_large=False, fade_size=0.0)
    with _occlusion(True), torch.no_grad():
        ref = render(camd, SyntheticGaussians(sc.subset(torch.arange(near)), dev, requires_grad=False), PIPE, bg, **st)
        out = render(camd, SyntheticGaussians(sc, dev, requires_grad=False), PIPE, bg, **st)
    torch.cuda.synchronize()
    assert torch.equal(out["render"], ref["render"]) and torch.equal(out["depth"], ref["depth"])
    assert int((out["radii"] > 0).sum()) == n


def _more_than_2_pow_32_instances_is_an_error_not_a_wrap():
    """140 000 Gaussians that each cover all 32 400 tiles of a 3840x2160 image: 4.5e9 (tile, Gaussian) instances.  The scanned
    offsets are 32-bit (like the reference's); the grand total is 64-bit and the call must report it as an error
    ("more than 2^32-1 tile instances", MSGS_ERR_TOO_MANY) — after a speculative stage 2 that stayed inside its buffers —
    and the next, ordinary call on the same thread must be unaffected."""
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    W, H = 3840, 2160
    n = 140_000
    sc, cam = small_scene(n, W, H, 5, sh_degree=0)
    sc.means3D[:, 0] = 0.0
    sc.means3D[:, 1] = 0.0
    sc.means3D[:, 2] = torch.linspace(4.0, 6.0, n)
    sc.scales[:] = 50.0                                       # 3 sigma covers the screen many times over
    sc.opacities[:] = 0.5
    dev = torch.device("cuda")
    camd, bg = cam.to(dev), torch.zeros(3, device=dev)
    st = dict(filter_small=False, filter_large=False, fade_size=0.0)
    model = SyntheticGaussians(sc, dev, requires_grad=False)
    with torch.no_grad(), pytest.raises(RuntimeError, match="tile instances"):
        render(camd, model, PIPE, bg, **st)
    torch.cuda.synchronize()
    # ... and through the launch / finish pair of the view pipeline
    import diff_gaussian_rasterization as dgr
    with torch.no_grad(), pytest.raises(RuntimeError, match="tile instances"):
        with dgr.deferred_forward() as pending:
            render(camd, model, PIPE, bg, **st)
            for p in pending:
                p.resolve()
    torch.cuda.synchronize()
    sc2, cam2 = small_scene(2000, 96, 64, 6, sh_degree=0)
    ok_model, cam2 = SyntheticGaussians(sc2, dev, requires_grad=False), cam2.to(dev)
    # ... and when the failure surfaces at the END of a deferred block, with a healthy view launched behind the failing one:
    # the block raises, but the healthy view's wait is finished too (its handle goes back to the pool) and a later resolve()
    # of the failed one raises again instead of touching a handle it no longer owns
    with torch.no_grad(), pytest.raises(RuntimeError, match="tile instances"):
        with dgr.deferred_forward() as pending:
            render(camd, model, PIPE, bg, **st)
            healthy = render(cam2, ok_model, PIPE, bg, **st)
    assert pending[1].state is not None and pending[0].state is None
    with pytest.raises(RuntimeError):
        pending[0].resolve()
    torch.cuda.synchronize()
    with torch.no_grad():
        a = render(cam2, ok_model, PIPE, bg, **st)["render"].clone()
        b = render(cam2, ok_model, PIPE, bg, **st)["render"]
    assert torch.equal(a, b) and torch.isfinite(a).all() and a.abs().max() > 0
    assert torch.equal(healthy["render"], a)


def test_instance_count_between_2_pow_31_and_2_pow_32():
    with _occlusion(False):
        _instance_count_between_2_pow_31_and_2_pow_32()


def _instance_count_between_2_pow_31_and_2_pow_32():
    """80 000 screen-filling Gaussians at 3840x2160: 2.59e9 instances, past every signed 32-bit index in the emit, the tile sort
    (31 GB of key / id buffers) and the range search.  Opacity 0.5 everywhere: every pixel terminates after 14 Gaussians, so
    the image and the gradients must be those of the 100 nearest Gaussians alone — bit-equal image, i.e. the sort put the
    nearest first in every one of the 32 400 tile lists of 80 000 entries."""
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip(f"needs ~50 GB of free HBM, {free / 2 ** 30:.0f} GiB available")
    W, H = 3840, 2160
    n, near = 80_000, 100
    sc, cam = small_scene(n, W, H, 9, sh_degree=0)
    sc.means3D[:, 0] = 0.0
    sc.means3D[:, 1] = 0.0
    sc.means3D[:, 2] = torch.linspace(4.0, 6.0, n)
    sc.scales[:] = 50.0
    sc.opacities[:] = 0.5
    dev = torch.device("cuda")
    camd, bg, dL = cam.to(dev), torch.zeros(3, device=dev), scenes.grad_seed(W, H, 9).to(dev)
    st = dict(filter_small=False, filter_large=False, fade_size=0.0)

    few = sc.subset(torch.arange(near))
    small = SyntheticGaussians(few, dev)
    ref = _run(render, small, camd, bg, dL, st)
    assert int(ref["radii"].min()) > 0
    big = SyntheticGaussians(sc, dev)
    out = _run(render, big, camd, bg, dL, st)
    assert torch.equal(out["render"], ref["render"]) and torch.equal(out["depth"], ref["depth"])
    assert int((out["radii"] > 0).sum()) == n
    for p_big, p_small in zip(big.parameters(), small.parameters()):
        scale = p_small.grad.abs().max().item()
        assert (p_big.grad[:near] - p_small.grad).abs().max().item() <= 1e-6 * scale
        assert not p_big.grad[near:].any()
    del big, out
    torch.cuda.empty_cache()
