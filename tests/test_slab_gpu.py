"""Depth-slab binning (msgs_view_t.slab_fraction; ms-gs_amd/csrc/api.hip forward_stage2_impl, binning.hip):
stage 2 bins and blends the nearest depth ranks first (slab A), the forward blend marks the tiles in which a pixel is still
blending, and only THOSE tiles receive the rest of the view (slab B: their complete lists) and are blended again.  A tile whose
every pixel terminated inside slab A never evaluates an entry behind it, so nothing a pixel walks
changes.  Every test renders the same inputs in slab mode and single-pass and demands BIT-IDENTICAL results: image, depth,
acc_pixel_size, radii, pixel_sizes, the per-pixel state the backward reads (final_T, n_contrib) and every gradient — at the full
BASELINE size it was built for (C5), on the 4K multi-scale model without its filters (occlusion cut-off and slabs together),
on randomised dense scenes over fractions from "everything stays open" to "nothing does", through the speculative and the
exact-buffer routes, with two views in flight, and through the wrapper's adaptive policy."""
import ctypes as C

import pytest
import torch

import scenes
from parity_utils import PIPE

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
OUT_KEYS = ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes")
PLAIN = dict(filter_small=False, filter_large=False, fade_size=1.0)


def _slab_stats(ctx):
    import diff_gaussian_rasterization as dgr
    geom = dgr._resolve(ctx.state)[0]
    o = (C.c_int64 * 6)()
    dgr._C.check(dgr._C.lib.msgs_slab_stats(C.c_void_p(geom.data_ptr()), geom.numel(), ctx.call.P, o,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "msgs_slab_stats")
    return dict(active=int(o[0]), rA=int(o[1]), DA=int(o[2]), n_open=int(o[3]), DB=int(o[4]), overflow=int(o[5]))


def _run(sc, cam, st, bg, dL, policy, backward=True, fused=False, calls=1):
    """`calls` renders of the same view (the first sizes its stage-2 buffers exactly, the later ones take the speculative route
    on buffers sized from the previous count); returns the LAST"""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, render_fused
    from synthetic_model import SyntheticGaussians
    prev_slab, dgr.slab_policy = dgr.slab_policy, policy
    try:
        dgr._last_instances.clear()
        fn = render_fused if fused else render
        for _ in range(calls):
            pc = SyntheticGaussians(sc, "cuda", requires_grad=backward)
            if backward:
                out = fn(cam, pc, PIPE, bg, **st)
                out["render"].backward(dL)
            else:
                with torch.no_grad():
                    out = fn(cam, pc, PIPE, bg, **st)
        torch.cuda.synchronize()
        ctx = out["render"].grad_fn if backward else None
        stats = _slab_stats(ctx) if ctx is not None else None
        D = ctx.state[3] if ctx is not None else None
        W, H = cam.image_width, cam.image_height
        per_pixel = None
        if ctx is not None:          # final_T [N] f32 at offset 0, n_contrib [N] u32 at the next 256-byte boundary (ImageLayout)
            image = ctx.state[2]
            n4 = 4 * W * H
            a = (n4 + 255) & ~255
            per_pixel = (image[:n4].clone(), image[a:a + n4].clone())
        return out, pc, D, stats, per_pixel
    finally:
        dgr.slab_policy = prev_slab


def _assert_identical(a, b, what, backward=True):
    (oa, pa, Da, _, ppa), (ob, pb, Db, _, ppb) = a, b
    for k in OUT_KEYS:
        assert torch.equal(oa[k], ob[k]), (what, k)
    if backward:
        assert Da == Db, (what, "instance count", Da, Db)
        assert torch.equal(ppa[0], ppb[0]), (what, "final_T")
        assert torch.equal(ppa[1], ppb[1]), (what, "n_contrib")
        assert torch.equal(oa["viewspace_points"].grad, ob["viewspace_points"].grad), (what, "means2D grad")
        for n in LEAVES:
            assert torch.equal(getattr(pa, n).grad, getattr(pb, n).grad), (what, n)


def _dense_scene(P, W, H, seed, scale=1.0, opacity=None):
    """a frustum scene whose pixels terminate early: many mid-sized, fairly opaque Gaussians"""
    sc = scenes.frustum_scene(P, W, H, seed=seed, scale_k=0.004 * 1920.0 / W * scale)
    if opacity is not None:
        g = torch.Generator().manual_seed(seed + 7)
        sc.opacities[:, 0] = opacity[0] + (opacity[1] - opacity[0]) * torch.rand(P, generator=g)
    return sc


@pytest.mark.parametrize("calls", [1, 2])
def test_c5_is_bit_identical_and_sheds_the_instances_nobody_walks(calls):
    """BASELINE C5 (5 M Gaussians, 3840x2160, filters on): 54.9 M instances, 4.35 M traversed.  Slab A = 16 % of them; the
    handful of tiles it leaves open are re-binned completely.  calls = 1: exact buffers; 2: the speculative route."""
    sc, cam, st = scenes.config("C5")
    cam = cam.to("cuda")
    bg = torch.zeros(3, device="cuda")
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).cuda()
    one = _run(sc, cam, st, bg, dL, "never", calls=calls)
    slab = _run(sc, cam, st, bg, dL, "0.16", calls=calls)
    print(f"[slab] C5 calls={calls}: D {one[2]}, slab {slab[3]}")
    _assert_identical(slab, one, ("C5", calls))
    s = slab[3]
    assert s["active"] == 1 and one[3]["active"] == 0 and s["overflow"] == 0
    assert s["DA"] + s["DB"] < one[2] // 4, s                 # the point of the pass
    assert s["DA"] <= 0.16 * one[2] + 32400 and s["n_open"] < 32400 // 20, s


def test_4k_model_without_filters_occlusion_and_slabs_together():
    """every fifth Gaussian of the C5 model with render.py's flags: the occlusion cut-off removes 1.68 G instances, the slabs
    work on what is left; bit-identical to the run with both off... the cut-off alone (uncut needs 40 GB of buffers)"""
    sc, cam, _ = scenes.config("C5")
    sc = sc.subset(torch.arange(0, sc.P, 5))
    cam = cam.to("cuda")
    bg = torch.tensor([0.2, 0.3, 0.1], device="cuda")
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).cuda()
    one = _run(sc, cam, PLAIN, bg, dL, "never")
    for frac in ("0.05", "0.3"):
        slab = _run(sc, cam, PLAIN, bg, dL, frac, calls=2)
        print(f"[slab] 4K filters off, fraction {frac}: D {one[2]}, slab {slab[3]}")
        _assert_identical(slab, one, ("4K filters off", frac))
        assert slab[3]["active"] == 1 and slab[3]["overflow"] == 0


@pytest.mark.parametrize("bwd_gen", [0, 1])
def test_random_dense_scenes_over_all_fractions(bwd_gen):
    """24 random scenes x fractions from 'every tile stays open' to 'none does', filters on and off, fused and reference entry"""
    import diff_gaussian_rasterization as dgr
    lib = dgr._C.lib
    pb = lib.msgs_set_backward_generation(bwd_gen)
    try:
        seen_open, seen_closed, seen_partial = 0, 0, 0
        for seed in range(24):
            g = torch.Generator().manual_seed(5000 + seed)
            W = int(torch.randint(720, 1400, (1,), generator=g))          # >= 2048 tiles of 16 x 16
            H = int(torch.randint(736, 1100, (1,), generator=g))
            P = int(torch.randint(20_000, 160_000, (1,), generator=g))
            ms = seed % 3 == 0
            sc = scenes.frustum_scene(P, W, H, seed=seed, multiscale=ms,
                                      scale_k=0.004 * 1920.0 / W * float(0.6 + 1.6 * torch.rand(1, generator=g)))
            if seed % 4 != 3:           # mostly opaque: pixels terminate; every fourth scene keeps the generator's opacities
                sc.opacities[:, 0] = 0.55 + 0.44 * torch.rand(P, generator=g)
            st = dict(filter_small=True, filter_large=True, fade_size=0.0) if (ms and seed % 2 == 0) else PLAIN
            cam = scenes.front_camera(W, H).to("cuda")
            bg = torch.rand(3, generator=g).cuda()
            dL = scenes.grad_seed(W, H, seed).cuda()
            one = _run(sc, cam, st, bg, dL, "never", fused=seed % 5 == 0)
            for frac in ("0.04", "0.15", "0.45"):
                slab = _run(sc, cam, st, bg, dL, frac, fused=seed % 5 == 0, calls=1 + seed % 2)
                _assert_identical(slab, one, (seed, W, H, P, frac))
                s = slab[3]
                assert s["active"] == 1 and s["overflow"] == 0, s
                tiles = ((W + 15) // 16) * ((H + 15) // 16)
                seen_open += s["n_open"] > tiles // 2
                seen_closed += s["n_open"] == 0
                seen_partial += 0 < s["n_open"] <= tiles // 2
        print(f"[slab] random scenes: mostly open {seen_open}, partial {seen_partial}, all closed {seen_closed}")
        assert seen_open >= 3 and seen_partial >= 3          # the sweep exercises both ends
    finally:
        lib.msgs_set_backward_generation(pb)


def test_forward_only_and_empty_and_tiny_views_fall_back():
    """no backward (viewer / evaluation); fewer than 2048 tiles, the deterministic mode: single pass"""
    import diff_gaussian_rasterization as dgr
    W, H = 1024, 800
    sc = _dense_scene(60_000, W, H, 3, opacity=(0.6, 0.99))
    cam = scenes.front_camera(W, H).to("cuda")
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    dL = scenes.grad_seed(W, H, 3).cuda()
    a = _run(sc, cam, PLAIN, bg, dL, "never", backward=False)
    b = _run(sc, cam, PLAIN, bg, dL, "0.1", backward=False, calls=2)
    _assert_identical(b, a, "forward only", backward=False)
    # small image: fewer than 2048 tiles -> single pass whatever the policy says
    W2, H2 = 480, 320
    sc2 = _dense_scene(20_000, W2, H2, 4, opacity=(0.6, 0.99))
    cam2 = scenes.front_camera(W2, H2).to("cuda")
    dL2 = scenes.grad_seed(W2, H2, 4).cuda()
    c = _run(sc2, cam2, PLAIN, bg, dL2, "0.1", calls=2)
    assert c[3]["active"] == 0
    # deterministic mode: single pass
    prev = dgr.set_deterministic(True)
    try:
        d = _run(sc, cam, PLAIN, bg, dL, "0.1", calls=2)
        assert d[3]["active"] == 0
    finally:
        dgr.set_deterministic(prev)


def test_two_views_in_flight_in_slab_mode():
    """deferred forwards (msgs_forward_launch / _finish on two streams, host/multi_view.py) with slabs forced on: bit-identical
    to the serial single-pass loop — outputs, per-view means2D gradients, accumulated leaves"""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    W, H = 960, 720
    sc = _dense_scene(80_000, W, H, 9, opacity=(0.5, 0.99))
    cams = [scenes.front_camera(W, H).to("cuda") for _ in range(4)]
    dLs = [scenes.grad_seed(W, H, 20 + v).cuda() for v in range(4)]
    bg = torch.tensor([0.4, 0.1, 0.2], device="cuda")
    prev = dgr.slab_policy
    try:
        dgr.slab_policy = "never"
        dgr._last_instances.clear()
        ref_pc = SyntheticGaussians(sc, "cuda")
        ref, ref_m2 = [], []
        for cam, dL in zip(cams, dLs):
            o = render(cam, ref_pc, PIPE, bg, **PLAIN)
            o["render"].backward(dL)
            ref.append(o)
            ref_m2.append(o["viewspace_points"].grad.clone())
        torch.cuda.synchronize()
        dgr.slab_policy = "0.12"
        for attempt in range(2):                  # no guess yet / speculative stage 2
            if attempt == 0:
                dgr._last_instances.clear()
            pc = SyntheticGaussians(sc, "cuda")
            kept = []

            def bwd(i, pkg):
                pkg["render"].backward(dLs[i])
                kept.append(pkg)
                return pkg["viewspace_points"]
            vs = ViewPipeline("cuda", n_streams=2).train_views(cams, pc, PIPE, bg, bwd, **PLAIN)
            torch.cuda.synchronize()
            for i, (o, r) in enumerate(zip(kept, ref)):
                for k in OUT_KEYS:
                    assert torch.equal(o[k], r[k]), (attempt, i, k)
                assert torch.equal(vs[i].grad, ref_m2[i]), (attempt, i)
            for n in LEAVES:
                assert torch.equal(getattr(pc, n).grad, getattr(ref_pc, n).grad), (attempt, n)
            assert _slab_stats(kept[0]["render"].grad_fn)["active"] == 1
    finally:
        dgr.slab_policy = prev


def test_adaptive_policy_engages_on_a_view_that_terminates_early_and_not_otherwise():
    """slab_policy = "adaptive" (the default): the library publishes D and D_trav of every large frame (one small kernel), the
    wrapper reads the publication of EARLIER frames of the same (model, image, filters) key when it collects the next count,
    and asks for slabs while D >= 6 x D_trav.  A dense opaque scene engages from its third frame on; a hazy one (nothing
    terminates) never does.  Same image either way."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    W, H = 1600, 1200
    cam = scenes.front_camera(W, H).to("cuda")
    bg = torch.zeros(3, device="cuda")
    dense = _dense_scene(700_000, W, H, 11, scale=1.3, opacity=(0.7, 0.99))
    hazy = _dense_scene(700_000, W, H, 12, scale=1.3, opacity=(0.004, 0.02))
    prev, dgr.slab_policy = dgr.slab_policy, "adaptive"
    try:
        for sc, engages in ((dense, True), (hazy, False)):
            dgr._last_instances.clear()
            dgr._fb_stats.clear(); dgr._fb_tag_of.clear(); dgr._fb_key_of.clear()
            pc = SyntheticGaussians(sc, "cuda", requires_grad=False)
            imgs, active = [], []
            with torch.no_grad():
                for it in range(6):
                    # (the per-call state lives on the autograd ctx, which a no_grad render does not keep: ask the wrapper's plan)
                    key = (0, sc.P, W, H, 0, 0)
                    frac, tag = dgr._slab_plan(key, dgr._last_instances.get(key), ((W + 15) // 16) * ((H + 15) // 16))
                    active.append(frac > 0.0)
                    out = render(cam, pc, PIPE, bg, **PLAIN)
                    torch.cuda.synchronize()
                    imgs.append(out["render"])
            st = dgr._fb_stats.get((0, W, H, 0, 0))
            print(f"[slab] adaptive: engages={engages} plan per call {active} stats {st}")
            assert all(torch.equal(imgs[0], im) for im in imgs[1:])
            assert st is not None and st["D"] >= dgr.SLAB_MIN_INSTANCES, st
            if engages:
                assert not active[0] and all(active[3:]), active
                assert st["D"] >= 6 * st["D_trav"] and st["DA"] + st["DB"] < st["D"] // 2, st
            else:
                assert not any(active), active
    finally:
        dgr.slab_policy = prev
