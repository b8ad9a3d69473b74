"""Guards of the drop-in Python boundary: inputs this build cannot honour are refused, never silently mis-rendered
(occ_multiplier / dc_delta away from their identity values, /root/reference/scene/gaussian_model.py:156-164,203-213),
mismatched per-Gaussian shapes raise before any kernel reads out of bounds, and an in-place edit of an input between
forward and backward raises autograd's version error instead of yielding wrong gradients."""
import math

import pytest
import torch

import scenes
from parity_utils import small_scene

pytestmark = pytest.mark.gpu


def _settings(cam, bg, deg=3):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width,
                                         tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
                                         scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                         projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center,
                                         prefiltered=False, debug=False)


def _inputs(P=500, W=64, H=48, seed=3):
    sc, cam = small_scene(P, W, H, seed)
    d = lambda t: t.to("cuda")
    kw = dict(means3D=d(sc.means3D), means2D=torch.zeros(P, 3, device="cuda"), opacities=d(sc.opacities), shs=d(sc.shs),
              scales=d(sc.scales), rotations=d(sc.rotations), max_pixel_sizes=d(sc.max_pixel_sizes),
              min_pixel_sizes=d(sc.min_pixel_sizes), occ_multiplier=d(sc.occ_multiplier), dc_delta=d(sc.dc_delta),
              base_mask=d(sc.base_mask))
    return kw, cam.to("cuda")


def test_non_identity_occ_multiplier_and_dc_delta_are_refused():
    from diff_gaussian_rasterization import GaussianRasterizer
    kw, cam = _inputs()
    r = GaussianRasterizer(_settings(cam, torch.zeros(3, device="cuda")))
    img = r(**kw)[0]                                           # identity values: accepted
    assert torch.isfinite(img).all()
    bad = dict(kw)
    bad["occ_multiplier"] = torch.sigmoid(torch.full_like(kw["occ_multiplier"], 4.6))     # --multi_occ: sigmoid(raw)
    with pytest.raises(NotImplementedError, match="occ_multiplier"):
        r(**bad)
    bad = dict(kw)
    bad["dc_delta"] = kw["dc_delta"].clone()
    bad["dc_delta"][7, 3, 0] = 0.01
    with pytest.raises(NotImplementedError, match="dc_delta"):
        r(**bad)
    # a leaf that passed is not re-read ... until it is modified in place
    occ = torch.nn.Parameter(kw["occ_multiplier"].clone())
    ok = dict(kw)
    ok["occ_multiplier"] = occ
    r(**ok)
    r(**ok)
    with torch.no_grad():
        occ[3, 1, 0] = 0.5
    with pytest.raises(NotImplementedError):
        r(**ok)


@pytest.mark.parametrize("name,shape", [("scales", (499, 3)), ("rotations", (500, 3)), ("shs", (501, 16, 3)),
                                        ("occ_multiplier", (499, 4, 1)), ("opacities", (499, 1)),
                                        ("min_pixel_sizes", (499,)), ("base_mask", (10,))])
def test_mismatched_per_gaussian_shapes_raise(name, shape):
    from diff_gaussian_rasterization import GaussianRasterizer
    kw, cam = _inputs()
    r = GaussianRasterizer(_settings(cam, torch.zeros(3, device="cuda")))
    fill = torch.ones if name == "occ_multiplier" else torch.zeros
    kw[name] = fill(*shape, device="cuda", dtype=torch.bool if name == "base_mask" else torch.float32)
    with pytest.raises(ValueError, match=name):
        r(**kw)


def test_inplace_edit_between_forward_and_backward_raises():
    from diff_gaussian_rasterization import GaussianRasterizer
    kw, cam = _inputs()
    r = GaussianRasterizer(_settings(cam, torch.zeros(3, device="cuda")))
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        kw[k] = kw[k].clone().requires_grad_(True)
    img = r(**kw)[0]
    with torch.no_grad():
        kw["opacities"].mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        img.sum().backward()
    # and through the getter-recognising path on leaf parameters
    from gaussian_renderer import render
    from parity_utils import PIPE
    from synthetic_model import SyntheticGaussians
    sc, cam2 = small_scene(400, 64, 48, 5)
    pc = SyntheticGaussians(sc, "cuda")
    out = render(cam2.to("cuda"), pc, PIPE, torch.zeros(3, device="cuda"), filter_small=False, filter_large=False,
                 fade_size=1.0)
    with torch.no_grad():
        pc._scaling.add_(0.1)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out["render"].sum().backward()


def test_direct_gradient_sinks_with_odd_gaussian_count():
    """P % 4 != 0: every bucket slice is still 16-byte aligned, the gradients written straight into the bucket equal
    the ones autograd accumulates the ordinary way."""
    from gaussian_renderer import render
    from parity_utils import PIPE
    from synthetic_model import SyntheticGaussians
    from view_parallel import PipelinedGradExchange
    P, W, H = 1003, 96, 64
    sc, cam = small_scene(P, W, H, 9)
    st = dict(filter_small=False, filter_large=False, fade_size=1.0)
    dL = scenes.grad_seed(W, H, 9).cuda()
    bg = torch.zeros(3, device="cuda")
    ref = SyntheticGaussians(sc, "cuda")
    render(cam.to("cuda"), ref, PIPE, bg, **st)["render"].backward(dL)
    pc = SyntheticGaussians(sc, "cuda")
    ex = PipelinedGradExchange(pc.parameters(), world=1, direct=True)
    assert all(v.data_ptr() % 16 == 0 for b in ex.buckets for v in b.views)
    ex.begin_view()
    render(cam.to("cuda"), pc, PIPE, bg, **st)["render"].backward(dL)
    b = ex.end_view()
    ex.drain()
    for p, q, v in zip(pc.parameters(), ref.parameters(), b.views):
        assert p.grad.data_ptr() == v.data_ptr()
        scale = q.grad.abs().max().clamp_min(1e-30)
        err = ((p.grad - q.grad).abs().max() / scale).item()
        print(f"[parity] direct sinks: {err:.3e}")
        assert err == 0.0         # same kernels, reproducible sums: the sink receives exactly what autograd would


@pytest.mark.parametrize("P,W,H,granularity", [(1003, 96, 64, 0), (1, 640, 480, 0), (2500, 80, 48, 2), (7, 16, 16, 0)])
def test_gradient_records_cleared_by_the_forward(P, W, H, granularity):
    """msgs.h grad_records: the forward's blend kernel clears the backward's per-Gaussian gradient records.  The buffer is
    poisoned beforehand (the caching allocator hands the freed block back), the first backward runs on the buffer the
    forward cleared, the second one (retain_graph) on a fresh buffer msgs_backward clears itself: identical bits, and both
    equal to a run without any pre-clearing (no_grad forward cannot be used for that, so: a poisoned second forward)."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import PIPE, render
    from synthetic_model import SyntheticGaussians
    sc, cam = small_scene(P, W, H, 21)
    st = dict(filter_small=False, filter_large=False, fade_size=1.0)
    dL = scenes.grad_seed(W, H, 21).cuda()
    bg = torch.zeros(3, device="cuda")
    nbytes = dgr._C.lib.msgs_backward_scratch_bytes(P)
    prev = dgr._C.lib.msgs_set_blend_granularity(granularity)
    try:
        grads = []
        for rep in range(2):
            poison = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda")      # NaN patterns in every record
            ptr = poison.data_ptr()
            del poison
            pc = SyntheticGaussians(sc, "cuda")
            out = render(cam.to("cuda"), pc, PIPE, bg, **st)
            ctx = out["render"].grad_fn
            assert ctx.grad_rec is not None and ctx.grad_rec.numel() == nbytes
            reused = ctx.grad_rec.data_ptr() == ptr
            out["render"].backward(dL, retain_graph=True)
            assert ctx.grad_rec is None                              # handed over once
            first = [p.grad.clone() for p in pc.parameters()]
            for p in pc.parameters():
                p.grad = None
            out["render"].backward(dL)                               # fresh scratch, cleared by msgs_backward
            second = [p.grad.clone() for p in pc.parameters()]
            for a, b in zip(first, second):
                assert torch.isfinite(a).all() and torch.equal(a, b)
            grads.append(first)
            print(f"[parity] grad_records P={P}: poisoned block reused by the forward: {reused}")
        for a, b in zip(*grads):
            assert torch.equal(a, b)
        # no backward can follow: nothing is allocated or cleared
        with torch.no_grad():
            pc = SyntheticGaussians(sc, "cuda")
            out = render(cam.to("cuda"), pc, PIPE, bg, **st)
            assert out["render"].grad_fn is None
    finally:
        dgr._C.lib.msgs_set_blend_granularity(prev)


@pytest.mark.parametrize("P,W,H", [(500, 64, 48), (333, 1024, 768), (3, 1920, 1080)])
def test_gradient_records_are_zero_after_the_forward_and_capacity_is_checked(P, W, H):
    """both clearing routes (a fill launch below 2048 tiles, the blend kernel's own stores above) leave every byte of the
    records zero, and only those bytes are written"""
    import diff_gaussian_rasterization as dgr
    kw, cam = _inputs(P, W, H)
    call = dgr._Call(_settings(cam, torch.zeros(3, device="cuda")), kw["means3D"], kw["shs"], None, kw["opacities"],
                     kw["scales"], kw["rotations"], None, kw["max_pixel_sizes"], kw["min_pixel_sizes"], None, None, None)
    nbytes = dgr._C.lib.msgs_backward_scratch_bytes(P)
    assert nbytes >= 80 * P
    buf = torch.full((nbytes + 256,), 0xFF, dtype=torch.uint8, device="cuda")
    for _ in range(2):                                           # first frame (two library calls) and steady state (one)
        buf.fill_(0xFF)
        dgr._forward_impl(call, buf[:nbytes])
        torch.cuda.synchronize()
        assert int(buf[:80 * P].max()) == 0
        assert int(buf[nbytes:].min()) == 0xFF                   # nothing behind the buffer was touched
    small = torch.empty(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="smaller"):
        dgr._forward_impl(call, small)


def test_per_gaussian_backward_entry_refuses_accumulation():
    """msgs_backward_per_gaussian (the isolation entry of tests/test_k8_isolation_gpu.py) runs the textbook branch of the
    per-Gaussian kernel, which neither waits for `wait_before_accumulate` nor records `accumulated`: accumulation across views
    is msgs_backward's contract, and asking for it here is an invalid argument instead of a race."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    kw, cam = _inputs()
    rs = _settings(cam, torch.zeros(3, device="cuda"))
    call = dgr._Call(rs, kw["means3D"], kw["shs"], None, kw["opacities"], kw["scales"], kw["rotations"], None,
                     kw["max_pixel_sizes"], kw["min_pixel_sizes"], None, None, kw["base_mask"])
    with torch.no_grad():
        _, _, _, radii, _, (geom, _, _, _) = dgr._forward_impl(call)
    P, K = call.P, call.K
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device="cuda")
    out = [e(P, 3), e(P, 3), e(P, K, 3), e(P), e(P, 3), e(P, 4)]
    p = lambda t: C.c_void_p(t.data_ptr())
    sums = torch.zeros(P, 9, dtype=torch.float64, device="cuda")
    ev = torch.cuda.Event()
    ev.record()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(acc=0, wait=None, rec=None):
        grads = dgr._C.Grads(p(out[0]), p(out[1]), p(out[2]), None, p(out[3]), p(out[4]), p(out[5]), None, None, None, None,
                             0, acc, wait, rec)
        return dgr._C.lib.msgs_backward_per_gaussian(C.byref(call.view), C.byref(call.g), p(radii), p(geom), geom.numel(),
                                                     p(sums), C.byref(grads), stream)
    assert run() == 0
    assert run(acc=1) == -1
    assert run(wait=C.c_void_p(ev.cuda_event)) == -1
    assert run(rec=C.c_void_p(ev.cuda_event)) == -1
    torch.cuda.synchronize()
    assert all(float(t.abs().max()) == 0.0 for t in out)       # zero sums in, zero gradients out: the valid call ran
