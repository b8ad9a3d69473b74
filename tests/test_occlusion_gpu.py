"""Exact per-tile occlusion cut-off (ms-gs_amd/csrc/occlusion.hip; msgs_set_occlusion): instances behind the depth at which
every pixel of a block of tiles has provably terminated are neither counted, emitted nor sorted — and nothing a pixel walks
changes.  Every test renders the same inputs with the pass on and off and demands BIT-IDENTICAL results: image, depth,
acc_pixel_size, radii, pixel_sizes, the per-pixel state the backward reads (final_T, n_contrib) and every gradient.  Size
-independent property at the full BASELINE sizes (the multi-scale C3 model without its filters — render.py's defaults, 427 M
instances uncut — and C5), randomised small scenes with opaque giants in front, and the adversarial corners: covers whose
smallest alpha sits at the 1/255 skip threshold, elongated rotated covers, covers that end inside a tile block."""
import ctypes as C

import pytest
import torch

import scenes
from parity_utils import PIPE

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
OUT_KEYS = ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes")
PLAIN = dict(filter_small=False, filter_large=False, fade_size=1.0)


def _stats(ctx):
    import diff_gaussian_rasterization as dgr
    geom = ctx.state[0]
    o = (C.c_int64 * 8)()
    dgr._C.check(dgr._C.lib.msgs_occlusion_stats(C.c_void_p(geom.data_ptr()), geom.numel(), ctx.call.P, o,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "msgs_occlusion_stats")
    import struct
    depth = lambda bits: None if bits in (0, 0xFFFFFFFF) else round(struct.unpack("f", struct.pack("I", bits))[0], 4)
    return dict(ran=int(o[0]), heavy=int(o[1]), candidates=int(o[2]), closed_blocks=int(o[3]), blocks=int(o[4]), block=int(o[5]),
                cut_min_depth=depth(int(o[6])), cut_max_depth=depth(int(o[7])))


def _run(sc, cam, st, bg, dL, occlusion, backward=True, fused=False):
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render, render_fused
    from synthetic_model import SyntheticGaussians
    prev = dgr._C.lib.msgs_set_occlusion(1 if occlusion else 0)
    try:
        dgr._last_instances.clear()
        pc = SyntheticGaussians(sc, "cuda", requires_grad=backward)
        fn = render_fused if fused else render
        if backward:
            out = fn(cam, pc, PIPE, bg, **st)
            out["render"].backward(dL)
        else:
            with torch.no_grad():
                out = fn(cam, pc, PIPE, bg, **st)
        torch.cuda.synchronize()
        ctx = out["render"].grad_fn if backward else None
        D = ctx.state[3] if ctx is not None else None
        stats = _stats(ctx) if ctx is not None else None
        W, H = cam.image_width, cam.image_height
        per_pixel = None
        if ctx is not None:          # final_T [N] f32 at offset 0, n_contrib [N] u32 at the next 256-byte boundary (ImageLayout)
            image = ctx.state[2]
            n4 = 4 * W * H
            a = (n4 + 255) & ~255
            per_pixel = (image[:n4].clone(), image[a:a + n4].clone())
        return out, pc, D, stats, per_pixel
    finally:
        dgr._C.lib.msgs_set_occlusion(prev)


def _assert_identical(a, b, what, backward=True):
    (oa, pa, _, _, ppa), (ob, pb, _, _, ppb) = a, b
    for k in OUT_KEYS:
        assert torch.equal(oa[k], ob[k]), (what, k)
    if backward:
        assert torch.equal(ppa[0], ppb[0]) and torch.equal(ppa[1], ppb[1]), (what, "final_T / n_contrib")
        assert torch.equal(oa["viewspace_points"].grad, ob["viewspace_points"].grad), (what, "means2D grad")
        for n in LEAVES:
            assert torch.equal(getattr(pa, n).grad, getattr(pb, n).grad), (what, n)


def _giants_scene(P, W, H, seed, n_giants, giant_scale, giant_opacity=None, elongate=1.0):
    """a frustum scene with `n_giants` screen-filling Gaussians mixed into its depth range"""
    sc = scenes.frustum_scene(P, W, H, seed=seed, scale_k=0.004 * 1920.0 / W * 0.5)
    g = torch.Generator().manual_seed(seed + 1)
    idx = torch.randperm(P, generator=g)[:n_giants]
    z = sc.means3D[idx, 2].abs().clamp_min(0.6)
    sc.means3D[idx, 2] = z
    sc.means3D[idx, 0] *= 0.3
    sc.means3D[idx, 1] *= 0.3
    s = giant_scale * z[:, None] * (0.7 + 0.6 * torch.rand(n_giants, 3, generator=g))
    s[:, 0] *= elongate
    sc.scales[idx] = s
    if giant_opacity is not None:
        sc.opacities[idx, 0] = giant_opacity if torch.is_tensor(giant_opacity) else torch.full((n_giants,), float(giant_opacity))
    return sc


@pytest.mark.parametrize("fused", [False, True])
def test_filters_off_c3_model_is_bit_identical_and_sheds_its_dead_instances(fused):
    """BASELINE C3's multi-scale model rendered with render.py's default flags: 427 M instances uncut."""
    sc, cam, _ = scenes.config("C3")
    cam = cam.to("cuda")
    bg = torch.zeros(3, device="cuda")
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 2).cuda()
    on = _run(sc, cam, PLAIN, bg, dL, True, fused=fused)
    off = _run(sc, cam, PLAIN, bg, dL, False, fused=fused)
    _assert_identical(on, off, "C3 filters off")
    print(f"[occlusion] C3 filters off: D {off[2]} -> {on[2]}, {on[3]}")
    assert off[3]["ran"] == 0 and on[3]["ran"] == 1
    assert on[2] < off[2] // 10, (on[2], off[2])          # the point of the pass


def test_headline_c3_and_c5_are_bit_identical():
    for name, seed in (("C3", 2), ("C5", 5)):
        sc, cam, st = scenes.config(name)
        cam = cam.to("cuda")
        bg = torch.zeros(3, device="cuda")
        dL = scenes.grad_seed(cam.image_width, cam.image_height, seed).cuda()
        on = _run(sc, cam, st, bg, dL, True)
        off = _run(sc, cam, st, bg, dL, False)
        _assert_identical(on, off, name)
        print(f"[occlusion] {name}: D {off[2]} -> {on[2]}, {on[3]}")
        assert on[2] <= off[2]
        del on, off
        torch.cuda.empty_cache()


def test_4k_without_filters():
    """every fifth Gaussian of the C5 model at 3840x2160 with render.py's flags (the whole model has more than 2^32 - 1
    instances uncut: refused, tests/test_max_size_gpu.py)"""
    sc, cam, _ = scenes.config("C5")
    sc = sc.subset(torch.arange(0, sc.P, 5))
    cam = cam.to("cuda")
    bg = torch.tensor([0.2, 0.3, 0.1], device="cuda")
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).cuda()
    on = _run(sc, cam, PLAIN, bg, dL, True)
    off = _run(sc, cam, PLAIN, bg, dL, False)
    _assert_identical(on, off, "4K filters off")
    print(f"[occlusion] 4K, C5 / 5, filters off: D {off[2]} -> {on[2]}, {on[3]}")
    assert on[2] < off[2] // 10


@pytest.mark.parametrize("gran,bwd_gen", [(0, 0), (1, 1), (1, 2), (2, 0)])
def test_random_scenes_with_giants_in_front(gran, bwd_gen):
    import diff_gaussian_rasterization as dgr
    lib = dgr._C.lib
    pg, pb = lib.msgs_set_blend_granularity(gran), lib.msgs_set_backward_generation(bwd_gen)
    try:
        closed_somewhere = 0
        for seed in range(12):
            g = torch.Generator().manual_seed(1000 + seed)
            W = int(torch.randint(300, 900, (1,), generator=g))       # (a cover candidate needs more than 96 tile instances)
            H = int(torch.randint(220, 600, (1,), generator=g))
            P = int(torch.randint(300, 6000, (1,), generator=g))
            n_g = int(torch.randint(5, 120, (1,), generator=g))
            op = None if seed % 3 == 0 else (0.2 + 0.79 * torch.rand(n_g, generator=g))
            sc = _giants_scene(P, W, H, seed, n_g, giant_scale=float(0.2 + 1.5 * torch.rand(1, generator=g)), giant_opacity=op,
                               elongate=float(torch.tensor([1.0, 1.0, 4.0, 12.0])[seed % 4]))
            cam = scenes.front_camera(W, H).to("cuda")
            bg = torch.rand(3, generator=g).cuda()
            dL = scenes.grad_seed(W, H, seed).cuda()
            on = _run(sc, cam, PLAIN, bg, dL, True)
            off = _run(sc, cam, PLAIN, bg, dL, False)
            _assert_identical(on, off, (seed, W, H, P, n_g))
            assert on[2] <= off[2]
            closed_somewhere += 1 if on[3]["closed_blocks"] else 0
        assert closed_somewhere >= 4          # the scenes do exercise the cut
    finally:
        lib.msgs_set_blend_granularity(pg)
        lib.msgs_set_backward_generation(pb)


def test_covers_at_the_skip_threshold_and_partial_covers():
    """Adversarial: stacks of identical screen-filling covers whose alpha at the far corner of a block sits within 1 % of 1/255
    on either side (a pixel that SKIPS an entry does not lose transmittance to it: such a cover must not count), and covers
    that end inside a block.  20 opacities around the threshold x both sides."""
    W, H = 256, 192
    cam = scenes.front_camera(W, H).to("cuda")
    bg = torch.tensor([0.3, 0.6, 0.9], device="cuda")
    dL = scenes.grad_seed(W, H, 7).cuda()
    base = scenes.frustum_scene(1500, W, H, seed=77, scale_k=0.004 * 1920.0 / W * 0.5)
    n = 40
    for k in range(10):
        sc = scenes.Scene(**{f: (v.clone() if torch.is_tensor(v) else v) for f, v in base.__dict__.items()})
        idx = torch.arange(n)
        sc.means3D[idx] = torch.tensor([0.0, 0.0, 1.0]) + 0.001 * torch.arange(n)[:, None] * torch.tensor([0.0, 0.0, 1.0])
        sigma_px = 60.0 + 25.0 * k                    # footprint sigma in pixels: from "ends inside the image" to "fills it"
        f = 1000.0 * W / 1920.0
        sc.scales[idx] = torch.full((n, 3), sigma_px / f)
        sc.rotations[idx] = torch.tensor([1.0, 0.0, 0.0, 0.0])
        # alpha at the image corner = o * exp(-r^2 / (2 sigma^2)); choose o so that it straddles 1/255 over the stack
        r2 = (W / 2) ** 2 + (H / 2) ** 2
        o_thr = (1.0 / 255.0) / torch.exp(torch.tensor(-r2 / (2 * sigma_px ** 2)))
        sc.opacities[idx, 0] = (o_thr * (0.99 + 0.02 * torch.rand(n, generator=torch.Generator().manual_seed(k)))).clamp(1e-3, 0.99)
        on = _run(sc, cam, PLAIN, bg, dL, True)
        off = _run(sc, cam, PLAIN, bg, dL, False)
        _assert_identical(on, off, ("threshold covers", k))


def test_giants_centred_off_screen_whose_rect_ends_inside_a_block():
    """Round-5 advisor finding: a cover counts for a block only where tile instances of it EXIST, i.e. inside its 3-sigma tile
    rect (Q4 / Q5); the alpha >= 1/255 level set of an opaque Gaussian reaches 3.33 sigma.  3000 opaque giants centred 880 px
    left of a 1080p image, sigma 300 px: their rects end in tile column 1..2, their level sets contain all of block column 0
    (tiles 0..3).  Before the fix the blocks of column 0 closed on those phantom covers and the Gaussians behind them in tile
    columns 2..3 — which hold no instance of any giant — lost their instances: wrong image and gradients.  On / off bit-identical,
    and the Gaussians behind are really drawn."""
    W, H = 1920, 1080
    cam = scenes.front_camera(W, H).to("cuda")
    bg = torch.tensor([0.05, 0.1, 0.2], device="cuda")
    dL = scenes.grad_seed(W, H, 11).cuda()
    f = 1000.0
    sc = scenes.frustum_scene(6000, W, H, seed=61, scale_k=0.004 * 0.5)
    g = torch.Generator().manual_seed(62)
    n = 3000
    idx = torch.arange(n)
    z = 1.0 + 0.2 * torch.rand(n, generator=g)
    sigma = 300.0 + 10.0 * torch.rand(n, generator=g)                       # px
    px = 18.0 + 14.0 * torch.rand(n, generator=g) - 3.0 * sigma             # rect ends at px 18..32 -> tile column 1 or 2
    py = 540.0 + 80.0 * (2.0 * torch.rand(n, generator=g) - 1.0)
    sc.means3D[idx, 0] = (px - W / 2) * z / f
    sc.means3D[idx, 1] = (py - H / 2) * z / f
    sc.means3D[idx, 2] = z
    # flat discs facing the camera (no extent along z): off-axis the Jacobian's z column would otherwise widen the footprint
    sc.scales[idx] = torch.stack([sigma * z / f, sigma * z / f, torch.full((n,), 1e-4)], dim=1)
    sc.rotations[idx] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    sc.opacities[idx, 0] = 0.99
    # Gaussians BEHIND the giants in tile columns 2..3 of block column 0 (px 36..62), around the giants' rows
    m = 200
    jdx = torch.arange(n, n + m)
    zb = 4.0 + torch.rand(m, generator=g)
    pxb = 36.0 + 26.0 * torch.rand(m, generator=g)
    pyb = 540.0 + 120.0 * (2.0 * torch.rand(m, generator=g) - 1.0)
    sc.means3D[jdx, 0] = (pxb - W / 2) * zb / f
    sc.means3D[jdx, 1] = (pyb - H / 2) * zb / f
    sc.means3D[jdx, 2] = zb
    sc.scales[jdx] = (2.0 * zb / f)[:, None].expand(m, 3).clone()
    sc.opacities[jdx, 0] = 0.9
    on = _run(sc, cam, PLAIN, bg, dL, True)
    off = _run(sc, cam, PLAIN, bg, dL, False)
    print(f"[occlusion] off-screen giants: D {off[2]} -> {on[2]}, {on[3]}")
    _assert_identical(on, off, "off-screen giants")
    assert on[3]["ran"] == 1 and on[3]["candidates"] >= n // 2, on[3]
    # the scene is what it claims to be: the giants' rects end inside block column 0 and the Gaussians behind are rendered
    assert (off[0]["radii"][n:n + m] > 0).sum().item() > m // 2
    grad_behind = getattr(off[1], "_opacity").grad[n:n + m].abs().sum().item()
    print(f"[occlusion] off-screen giants: rendered behind {(off[0]['radii'][n:n + m] > 0).sum().item()} of {m}, "
          f"sum |dL/dopacity| behind {grad_behind:.3e}, giants' radii {off[0]['radii'][:n].min().item()}..{off[0]['radii'][:n].max().item()}")
    assert grad_behind > 0.0


def test_a_view_that_really_closes_blocks_matches_the_oracle():
    """the bit-identity tests compare the library with itself; here a giants scene whose blocks do close is compared with the CPU
    oracle directly (forward <= 1e-5 off the flagged pixels, gradients <= 1e-4 on the unflagged Gaussians) with the pass forced on"""
    import diff_gaussian_rasterization as dgr
    from oracle import oracle_ctypes as oc
    from parity_utils import check_backward, check_forward, hip_render
    W, H = 420, 300
    sc = _giants_scene(2500, W, H, 5, 60, giant_scale=1.2, giant_opacity=0.9)
    cam = scenes.front_camera(W, H)
    bg = torch.tensor([0.2, 0.5, 0.1])
    dL = scenes.grad_seed(W, H, 5)
    prev = dgr._C.lib.msgs_set_occlusion(1)
    try:
        dgr._last_instances.clear()
        out, pc, m2 = hip_render(sc, cam, PLAIN, bg, dL)
        stats = _stats(out["render"].grad_fn)
    finally:
        dgr._C.lib.msgs_set_occlusion(prev)
    assert stats["ran"] and stats["closed_blocks"] > 0, stats
    orc = oc.rasterize(pc.seen, cam, PLAIN, bg)
    og = oc.backward(orc, dL)
    check_forward(out, orc, "occlusion-closing")
    check_backward(pc, m2, og, "occlusion-closing", flagged=orc.borderline_gaussians)


def test_two_views_in_flight_on_a_scene_that_closes_blocks():
    """the deferred forward (msgs_forward_launch / _finish on two streams, host/multi_view.py) with the pass on, on views whose
    blocks close: bit-identical to the serial loop with the pass OFF — outputs, per-view means2D gradients, accumulated leaves"""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from multi_view import ViewPipeline
    from synthetic_model import SyntheticGaussians
    W, H = 480, 320
    sc = _giants_scene(4000, W, H, 9, 80, giant_scale=1.0, giant_opacity=0.95)
    cams = [scenes.front_camera(W, H).to("cuda") for _ in range(4)]
    dLs = [scenes.grad_seed(W, H, 20 + v).cuda() for v in range(4)]
    bg = torch.tensor([0.4, 0.1, 0.2], device="cuda")
    try:
        prev = dgr._C.lib.msgs_set_occlusion(0)
        dgr._last_instances.clear()
        ref_pc = SyntheticGaussians(sc, "cuda")
        ref, ref_m2 = [], []
        for cam, dL in zip(cams, dLs):
            o = render(cam, ref_pc, PIPE, bg, **PLAIN)
            o["render"].backward(dL)
            ref.append(o)
            ref_m2.append(o["viewspace_points"].grad.clone())
        torch.cuda.synchronize()
        D_off = ref[0]["render"].grad_fn.state[3]
        dgr._C.lib.msgs_set_occlusion(1)
        for attempt in range(2):                  # no guess yet / speculative stage 2 from the cut count
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
            D_on = dgr._resolve(kept[0]["render"].grad_fn.state)[3]
            assert D_on < D_off, (D_on, D_off)
    finally:
        dgr._C.lib.msgs_set_occlusion(prev)


def test_8k_image_uses_larger_cover_blocks():
    """7680x4320 = 480 x 270 tiles: more than 2048 blocks of 4 x 4 tiles, so the pass works on 8 x 8-tile blocks (60 x 34); forward
    and backward with giants in front, bit-identical to the uncut path"""
    W, H = 7680, 4320
    sc = _giants_scene(3000, W, H, 21, 60, giant_scale=1.0, giant_opacity=0.9)
    cam = scenes.front_camera(W, H).to("cuda")
    bg = torch.tensor([0.1, 0.1, 0.3], device="cuda")
    dL = scenes.grad_seed(W, H, 3).cuda()
    on = _run(sc, cam, PLAIN, bg, dL, True)
    off = _run(sc, cam, PLAIN, bg, dL, False)
    _assert_identical(on, off, "8K")
    print(f"[occlusion] 8K: D {off[2]} -> {on[2]}, {on[3]}")
    assert on[3]["block"] >= 8 and on[3]["closed_blocks"] > 0 and on[2] < off[2]       # (8 unless MSGS_OCC_BLOCK asks for more)


def test_pyramid_levels_and_filters_on_are_unchanged():
    """the training path (filters on, fade 0) at three pyramid levels of the C3 scene, and the switch really switches"""
    import diff_gaussian_rasterization as dgr
    sc, _, st = scenes.config("C3")
    sc = sc.subset(torch.arange(0, sc.P, 5))
    for k in (0, 2, 4):
        W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
        cam = scenes.front_camera(W, H).to("cuda")
        bg = torch.zeros(3, device="cuda")
        dL = scenes.grad_seed(W, H, 40 + k).cuda()
        on = _run(sc, cam, st, bg, dL, True)
        off = _run(sc, cam, st, bg, dL, False)
        _assert_identical(on, off, ("pyramid", k))
    assert dgr._C.lib.msgs_set_occlusion(1) == 1


def test_the_pass_always_runs_and_the_heavy_queue_hint_follows_what_the_views_close():
    """Round 6: the pass is one launch that a view without cover candidates leaves after its first grid barrier, so it runs on
    EVERY forward (the adaptive skip policy of round 5 and its cliff are gone).  A sweep that alternates a view that closes with
    one that looks away from everything — starting on the one that does not close — cuts the closing view every time.  What the
    wrapper still adapts per (model, image, filters) key is a hint for one small launch (msgs_view_t.no_heavy_queue): on after a
    call of the key closed a block, off after HEAVY_QUEUE_MEMORY calls that did not.  Same image either way."""
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    W, H = 480, 320
    cam = scenes.front_camera(W, H).to("cuda")
    away = scenes.ring_camera(0, 8, W, H, radius=50.0).to("cuda")        # far from the frustum scene, looking back at the origin
    bg = torch.zeros(3, device="cuda")
    quiet = scenes.frustum_scene(3000, W, H, seed=5, scale_k=0.004 * 1920.0 / W * 0.3)          # nothing to cut
    walls = _giants_scene(3000, W, H, 5, 60, giant_scale=1.5, giant_opacity=0.9)                # opaque covers in front
    info = (C.c_int64 * 8)()
    key = (0, 3000, W, H, 0, 0)
    # (1) alternating sweep starting on the view that closes nothing: the closing view is cut on every visit
    dgr._occ_hot.clear()
    dgr._last_instances.clear()
    pc = SyntheticGaussians(walls, "cuda", requires_grad=False)
    prev = dgr._C.lib.msgs_set_occlusion(0)
    with torch.no_grad():
        render(cam, pc, PIPE, bg, **PLAIN)
    D_uncut = dgr._last_instances[key]
    dgr._C.lib.msgs_set_occlusion(1)
    dgr._last_instances.clear()
    closed, counts, imgs = [], [], []
    with torch.no_grad():
        for it in range(12):
            facing = it % 2 == 1
            out = render(cam if facing else away, pc, PIPE, bg, **PLAIN)
            dgr._C.lib.msgs_forward_info(info)
            if facing:
                closed.append(int(info[1]))
                imgs.append(out["render"])
    dgr._C.lib.msgs_set_occlusion(prev)
    assert all(closed), closed
    assert all(torch.equal(imgs[0], im) for im in imgs[1:])
    assert dgr._occ_hot[key] >= dgr.HEAVY_QUEUE_MEMORY - 1            # the queue stays on for this key
    # (2) a quiet scene: the hint switches the queue off after HEAVY_QUEUE_MEMORY calls, the image does not change
    dgr._occ_hot.clear()
    dgr._last_instances.clear()
    pc = SyntheticGaussians(quiet, "cuda", requires_grad=False)
    imgs, off = [], []
    with torch.no_grad():
        for it in range(3):
            off.append(dgr._heavy_queue_off(key))
            imgs.append(render(cam, pc, PIPE, bg, **PLAIN)["render"])
            dgr._C.lib.msgs_forward_info(info)
            assert int(info[1]) == 0
    assert off == [False, True, True], off                             # (first call of a key: on; then the memory is 0)
    assert all(torch.equal(imgs[0], im) for im in imgs[1:])
    assert D_uncut > 0
