"""Two views in flight on two HIP streams (ms-gs_amd/host/multi_view.py over msgs_forward_launch / msgs_forward_finish) give
bit-identical results to the same views rendered one after the other on one stream: outputs, per-view screen-space gradients and
the gradients accumulated over the views of one optimizer step.  Also the corners of the deferred forward: no guess yet, a guess
that is exceeded (stage 2 redone on exact buffers at resolve time), an empty view, an exception inside the block."""
import pytest
import torch

import scenes
from parity_utils import PIPE

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
ST = dict(filter_small=False, filter_large=False, fade_size=1.0)
OUT_KEYS = ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes", "visibility_filter")


def _ball(P=60000, n_views=6, W=320, H=200):
    sc = scenes.ball_scene(P, seed=44, log_s=-3.0)
    cams = [scenes.ring_camera(v, n_views, W, H).to("cuda") for v in range(n_views)]
    dLs = [scenes.grad_seed(W, H, 90 + v).cuda() for v in range(n_views)]
    return sc, cams, dLs


def _serial_train(sc, cams, dLs, bg, st=ST):
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    pc = SyntheticGaussians(sc, "cuda")
    outs, m2 = [], []
    for cam, dL in zip(cams, dLs):
        o = render(cam, pc, PIPE, bg, **st)
        o["render"].backward(dL)
        outs.append(o)
        m2.append(o["viewspace_points"].grad.clone())
    torch.cuda.synchronize()
    return pc, outs, m2


@pytest.mark.parametrize("share_getters,accumulate_in_kernel", [(False, False), (True, True), (True, False), (False, True)])
@pytest.mark.parametrize("n_streams", [2, 3])
def test_train_views_equals_the_serial_loop_bit_for_bit(share_getters, accumulate_in_kernel, n_streams):
    import diff_gaussian_rasterization as dgr
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    sc, cams, dLs = _ball()
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    dgr._last_instances.clear()
    ref_pc, ref_outs, ref_m2 = _serial_train(sc, cams, dLs, bg)
    for attempt in range(2):                      # first: no guess for this shape yet; second: speculative stage 2
        if attempt == 0:
            dgr._last_instances.clear()
        pc = SyntheticGaussians(sc, "cuda")
        pipe = ViewPipeline("cuda", n_streams=n_streams)
        kept = []

        def bwd(i, pkg):
            pkg["render"].backward(dLs[i])
            kept.append(pkg)
            return pkg["viewspace_points"]
        vs = pipe.train_views(cams, pc, PIPE, bg, bwd, share_getters=share_getters, accumulate_in_kernel=accumulate_in_kernel,
                              **ST)
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(kept, ref_outs)):
            for k in OUT_KEYS:
                assert torch.equal(o[k], r[k]), (attempt, i, k)
            assert torch.equal(vs[i].grad, ref_m2[i]), (attempt, i, "means2D grad")
        for n in LEAVES:
            assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), (attempt, n)


def test_render_views_forward_only_and_consume():
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    sc, cams, _ = _ball(n_views=5)
    bg = torch.zeros(3, device="cuda")
    pc = SyntheticGaussians(sc, "cuda")
    with torch.no_grad():
        ref = [render(c, pc, PIPE, bg, **ST) for c in cams]
        pipe = ViewPipeline("cuda")
        outs = pipe.render_views(cams, pc, PIPE, bg, **ST)
        for o, r in zip(outs, ref):
            for k in OUT_KEYS:
                assert torch.equal(o[k], r[k]), k
        # a sweep that only reduces each view (train.py:282-299): minimum pixel size over the cameras that see a Gaussian
        mins = pipe.render_views(cams, pc, PIPE, bg, consume=lambda i, p: torch.where(p["radii"] > 0, p["pixel_sizes"],
                                                                                      torch.full_like(p["pixel_sizes"], 1e9)),
                                 **ST)
        got = torch.stack(mins).min(dim=0).values
        want = torch.stack([torch.where(r["radii"] > 0, r["pixel_sizes"], torch.full_like(r["pixel_sizes"], 1e9))
                            for r in ref]).min(dim=0).values
        assert torch.equal(got, want)
    assert dgr._status_pool, "status handles go back to the pool"


def test_guess_exceeded_inside_the_pipeline_is_redone_at_resolve_time():
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    sc, cams, dLs = _ball(n_views=4)
    bg = torch.tensor([0.3, 0.1, 0.0], device="cuda")
    big = dict(ST)
    dgr._last_instances.clear()
    ref_pc, ref_outs, _ = _serial_train(sc, cams, dLs, bg, big)
    D_true = [o["render"].grad_fn.state[3] for o in ref_outs]
    # a guess 20x too small: every view's speculative stage 2 is truncated and has to be redone
    for k in list(dgr._last_instances):
        dgr._last_instances[k] = max(4096, min(D_true) // 20)
    pc = SyntheticGaussians(sc, "cuda")
    kept = []

    def bwd(i, pkg):
        pkg["render"].backward(dLs[i])
        kept.append(pkg)
    ViewPipeline("cuda").train_views(cams, pc, PIPE, bg, bwd, **big)
    torch.cuda.synchronize()
    for o, r, D in zip(kept, ref_outs, D_true):
        assert o["render"].grad_fn.state.resolve()[3] == D
        for k in OUT_KEYS:
            assert torch.equal(o[k], r[k]), k
    for n in LEAVES:
        assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), n


def test_deferred_block_resolves_on_exit_and_survives_an_exception():
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    sc, cams, _ = _ball(P=20000, n_views=2)
    bg = torch.zeros(3, device="cuda")
    pc = SyntheticGaussians(sc, "cuda")
    with torch.no_grad():
        ref = render(cams[0], pc, PIPE, bg, **ST)
        with dgr.deferred_forward() as pending:
            a = render(cams[0], pc, PIPE, bg, **ST)
            b = render(cams[1], pc, PIPE, bg, **ST)
            assert len(pending) == 2 and all(p.state is None for p in pending)
        assert all(p.state is not None for p in pending)
        torch.cuda.synchronize()
        assert torch.equal(a["render"], ref["render"])
        with pytest.raises(ZeroDivisionError):
            with dgr.deferred_forward() as pending:
                render(cams[1], pc, PIPE, bg, **ST)
                1 / 0
        assert pending[0].state is not None           # the wait was finished, the handle is reusable
        c = render(cams[0], pc, PIPE, bg, **ST)       # and the ordinary (waiting) forward works afterwards
        assert torch.equal(c["render"], ref["render"])
        # an empty view (everything behind the camera) inside a deferred block
        pc._xyz[:, :] = 1e3
        with dgr.deferred_forward() as pending:
            e = render(cams[0], pc, PIPE, bg, **ST)
        assert pending[0].resolve()[3] == 0 and int((e["radii"] > 0).sum()) == 0


def test_multi_view_step_exchange_on_one_rank_equals_the_serial_mean():
    """view_parallel.MultiViewStepExchange without a process group: all views of the step through the pipeline into the flat
    bucket, averaged over the views — equal to the serial loop's accumulated gradients / views (same additions, one division)"""
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    from view_parallel import MultiViewStepExchange
    sc, cams, dLs = _ball(n_views=4)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    ref_pc, _, ref_m2 = _serial_train(sc, cams, dLs, bg)
    pc = SyntheticGaussians(sc, "cuda")
    ex = MultiViewStepExchange(pc, len(cams))
    pipe = ViewPipeline("cuda")
    for _ in range(2):                                    # two optimizer steps reuse the bucket
        vs = ex.step(pipe, cams, PIPE, bg, lambda i, pkg: (pkg["render"].backward(dLs[i]), pkg["viewspace_points"])[1], **ST)
        torch.cuda.synchronize()
        for n, v in zip(LEAVES, ex.bucket.views):
            assert getattr(pc, n).grad.data_ptr() == v.data_ptr(), n
            assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad / len(cams)), n
        for i in range(len(cams)):
            assert torch.equal(vs[i].grad, ref_m2[i])


def test_randomised_sweeps_stay_bit_identical():
    """tools/soak_pipeline.py in small: random views, pyramid levels, lane counts, entries and accumulation modes, first-frame and
    guess-exceeded paths mixed in — every sweep equal to the serial loop bit for bit (stream-ordering / allocator-reuse regressions
    show up here)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # in-process (a child process must not be started from a process that has initialised the GPU)
    sys.argv, argv = [os.path.join(root, "tools", "soak_pipeline.py"), "80", "80000"], sys.argv
    try:
        with pytest.raises(SystemExit) as e:
            exec(compile(open(sys.argv[0]).read(), sys.argv[0], "exec"), {"__name__": "__main__", "__file__": sys.argv[0]})
        assert e.value.code == 0
    finally:
        sys.argv = argv


def test_pipeline_with_the_plain_path_and_debug_mode():
    """views that do NOT take the chained / raw entry inside the pipeline: override_color (plain autograd path: the accumulator is
    not consulted, autograd accumulates as usual) and pipe.debug (msgs_view_t::debug: no speculative stage 2, every stage
    synchronised) — results equal to the serial loop"""
    import types
    from gaussian_renderer import render
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    sc, cams, dLs = _ball(P=20000, n_views=3)
    bg = torch.tensor([0.2, 0.2, 0.2], device="cuda")
    colors = torch.rand(sc.P, 3, generator=torch.Generator().manual_seed(3)).cuda().requires_grad_(True)
    dbg = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=True)
    for pipe_ns, kw in ((PIPE, dict(override_color=colors)), (dbg, {})):
        ref_pc = SyntheticGaussians(sc, "cuda")
        colors.grad = None
        ref = []
        for c, d in zip(cams, dLs):
            o = render(c, ref_pc, pipe_ns, bg, **kw, **ST)
            o["render"].backward(d)
            ref.append(o)
        ref_color_grad = None if colors.grad is None else colors.grad.clone()
        pc = SyntheticGaussians(sc, "cuda")
        colors.grad = None
        got = []

        def bwd(i, pkg):
            pkg["render"].backward(dLs[i])
            got.append(pkg)
        rf = lambda c, m, p, b, **s: render(c, m, p, b, **kw, **s)
        if "override_color" in kw:        # shared getters cannot serve a render that leaves their backward to autograd: said loudly
            with pytest.raises(RuntimeError, match="share_getters=False"):
                ViewPipeline("cuda").train_views(cams, SyntheticGaussians(sc, "cuda"), pipe_ns, bg, lambda i, pkg: None, render_fn=rf, **ST)
            colors.grad = None
        ViewPipeline("cuda").train_views(cams, pc, pipe_ns, bg, bwd, render_fn=rf, share_getters="override_color" not in kw, **ST)
        torch.cuda.synchronize()
        for o, r in zip(got, ref):
            for k in OUT_KEYS:
                assert torch.equal(o[k], r[k]), k
        for n in ("_xyz", "_opacity", "_scaling", "_rotation"):
            assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), n
        if "override_color" in kw:
            assert torch.equal(colors.grad, ref_color_grad)
        else:
            for n in ("_features_dc", "_features_rest"):
                assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), n


def test_a_view_that_raises_mid_step_leaves_the_pipeline_usable():
    """backward_fn raises on the third of six views (two forwards are in flight, an accumulator is installed, one view's gradients
    are already in the bucket): the exception reaches the caller, the lanes are joined, no accumulator stays installed, and the
    same pipeline then produces the serial loop's bits."""
    import diff_gaussian_rasterization as dgr
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    sc, cams, dLs = _ball(P=30000)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    ref_pc, _, _ = _serial_train(sc, cams, dLs, bg)
    pc = SyntheticGaussians(sc, "cuda")
    pipe = ViewPipeline("cuda")

    def failing(i, pkg):
        if i == 2:
            raise KeyError("view 2")
        pkg["render"].backward(dLs[i])
    with pytest.raises(KeyError):
        pipe.train_views(cams, pc, PIPE, bg, failing, **ST)
    torch.cuda.synchronize()
    assert dgr.set_grad_accumulator(None) is None
    for p in pc.parameters():
        p.grad = None
    pipe.train_views(cams, pc, PIPE, bg, lambda i, pkg: pkg["render"].backward(dLs[i]), **ST)
    torch.cuda.synchronize()
    for n in LEAVES:
        assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), n
