"""GPU parity tests: the HIP path (through the drop-in Python API and the C ABI) against the CPU
oracle on the same seeded inputs.  Run on the MI355X box with `pytest -m gpu`."""
import math

import pytest
import torch

import scenes
from parity_utils import (PIPE, check_backward, check_forward, hip_render, rel_err, small_scene)

pytestmark = pytest.mark.gpu

ST0 = dict(filter_small=False, filter_large=False, fade_size=1.0)


@pytest.fixture(autouse=True, params=[(1, 1, True), (2, 1, True), (1, 1, False), (2, 1, False), (0, 2, True), (0, 2, False)],
                ids=["bwd4-chained", "bwd1-chained", "bwd4-plain", "bwd1-plain", "fine-chained", "fine-plain"])
def backward_variants(request):
    """Every test runs with every blend kernel: the quadrant-per-wave forward with both coarse backward kernels (by
    default the library picks by tile count, which would leave the one-wave-per-tile kernel to the full-size tests
    only), and the fine-grained forward + backward (sixteen waves per tile; by default below 600 tiles, which would
    leave the coarse kernels to the larger tests only) — and with the recognition of the reference's getters on
    (gradients chained to the leaf parameters inside msgs_backward) and off (autograd runs the getters' backward)."""
    import diff_gaussian_rasterization as dgr
    gen, gran, chain = request.param
    prev_gen = dgr._C.lib.msgs_set_backward_generation(gen)
    prev_gran = dgr._C.lib.msgs_set_blend_granularity(gran)
    prev_chain = dgr.chain_reference_getters
    dgr.chain_reference_getters = chain
    yield request.param
    dgr._C.lib.msgs_set_backward_generation(prev_gen)
    dgr._C.lib.msgs_set_blend_granularity(prev_gran)
    dgr.chain_reference_getters = prev_chain


def _oracle(scene, cam, st, bg, dL=None, **kw):
    from oracle import oracle_ctypes as oc
    r = oc.rasterize(scene, cam, st, bg, **kw)
    g = oc.backward(r, dL) if dL is not None else None
    return r, g


def test_library_is_the_hip_one():
    import diff_gaussian_rasterization as dgr
    assert dgr._C._LIB_PATH.endswith("libmsgs_hip.so")
    assert dgr._C.lib.msgs_abi_version() == dgr._C.ABI_VERSION == 11


@pytest.mark.parametrize("P,W,H,seed,deg,bgv", [
    (300, 48, 40, 11, 3, (0.2, 0.5, 0.7)),
    (2000, 100, 75, 12, 3, (0.0, 0.0, 0.0)),
    (5000, 160, 128, 13, 2, (1.0, 1.0, 1.0)),
    (3000, 128, 128, 14, 1, (0.0, 0.0, 0.0)),
    (3000, 130, 70, 15, 0, (0.1, 0.1, 0.1)),
])
def test_forward_backward_vs_oracle(P, W, H, seed, deg, bgv):
    sc, cam = small_scene(P, W, H, seed, sh_degree=deg)
    bg = torch.tensor(bgv)
    dL = scenes.grad_seed(W, H, seed)
    out, pc, m2 = hip_render(sc, cam, ST0, bg, dL)
    orc, og = _oracle(pc.seen, cam, ST0, bg, dL)
    check_forward(out, orc, f"seed{seed}")
    check_backward(pc, m2, og, f"seed{seed}", flagged=orc.borderline_gaussians)


def test_config_c1_forward():
    """BASELINE.json configs[0]: 10k Gaussians, 256x256, SH degree 0, forward only."""
    sc, cam, st = scenes.config("C1")
    bg = torch.zeros(3)
    out, pc, _ = hip_render(sc, cam, st, bg)
    orc, _ = _oracle(pc.seen, cam, st, bg)
    check_forward(out, orc, "C1")


def test_multiscale_filters():
    W, H = 160, 96
    sc, cam = small_scene(4000, W, H, 21, multiscale=True, scale_k=0.004 * 1920.0 / W * 0.15)
    bg = torch.tensor([0.3, 0.3, 0.3])
    dL = scenes.grad_seed(W, H, 21)
    for st in (dict(filter_small=True, filter_large=True, fade_size=0.0),
               dict(filter_small=True, filter_large=True, fade_size=1.0),
               dict(filter_small=True, filter_large=False, fade_size=0.5),
               dict(filter_small=False, filter_large=True, fade_size=0.0)):
        out, pc, m2 = hip_render(sc, cam, st, bg, dL)
        orc, og = _oracle(pc.seen, cam, st, bg, dL)
        check_forward(out, orc, str(st))
        check_backward(pc, m2, og, str(st), flagged=orc.borderline_gaussians)
    # the filters must actually drop something in this scene
    out_nf, _, _ = hip_render(sc, cam, ST0, bg)
    out_f, _, _ = hip_render(sc, cam, dict(filter_small=True, filter_large=True, fade_size=0.0), bg)
    assert (out_f["radii"] > 0).sum() < (out_nf["radii"] > 0).sum()
    # pixel_sizes is reported for filtered Gaussians too (train.py:290-299 relies on it)
    dropped = (out_nf["radii"] > 0) & (out_f["radii"] == 0)
    assert dropped.any() and (out_f["pixel_sizes"][dropped] >= 0).all()
    assert torch.equal(out_f["pixel_sizes"], out_nf["pixel_sizes"])


def test_base_mask_exempts_small_filter():
    W, H = 96, 96
    sc, cam = small_scene(1500, W, H, 22, multiscale=True, scale_k=0.004 * 1920.0 / W * 0.1)
    sc.min_pixel_sizes[:] = 1.0e6         # everything is "too small"
    bg = torch.zeros(3)
    st = dict(filter_small=True, filter_large=False, fade_size=0.0)
    out, _, _ = hip_render(sc, cam, st, bg)
    assert (out["radii"] > 0).sum() == 0
    assert torch.allclose(out["render"], torch.zeros_like(out["render"]))
    sc.base_mask[::2] = True
    out2, pc2, _ = hip_render(sc, cam, st, bg)
    orc, _ = _oracle(pc2.seen, cam, st, bg)
    check_forward(out2, orc, "base_mask")
    assert (out2["radii"][1::2] == 0).all() and (out2["radii"][::2] > 0).any()


def test_python_side_cov_and_colors():
    """pipe.compute_cov3D_python / convert_SHs_python paths (render() :68-91)."""
    import types
    W, H = 96, 64
    sc, cam = small_scene(1500, W, H, 31)
    bg = torch.tensor([0.5, 0.1, 0.9])
    dL = scenes.grad_seed(W, H, 31)
    ref_out, ref_pc, ref_m2 = hip_render(sc, cam, ST0, bg, dL)
    for pipe in (types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=False, debug=False),
                 types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=True, debug=False),
                 types.SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=True, debug=True)):
        out, pc, m2 = hip_render(sc, cam, ST0, bg, dL, pipe=pipe)
        assert (out["render"] - ref_out["render"]).abs().max().item() <= 2e-5
        assert torch.equal(out["radii"], ref_out["radii"])
        # the Python-side activations differentiate through torch: end-to-end grads must agree with the
        # in-op path up to float32 rounding of two different formula orderings
        for n in ("_xyz", "_opacity", "_scaling", "_rotation", "_features_dc", "_features_rest"):
            e = rel_err(getattr(pc, n).grad, getattr(ref_pc, n).grad)
            print(f"[parity] python-side cov/colors {n}: {e:.3e}")
            assert e <= (1e-4 if n in ("_scaling", "_rotation") else 5e-6), n     # measured 2.2e-5 / 3e-6 / <= 5.2e-7


def test_precomputed_inputs_vs_oracle():
    from oracle import torch_oracle as to
    W, H = 80, 64
    sc, cam = small_scene(1200, W, H, 32)
    bg = torch.tensor([0.0, 0.3, 0.0])
    dL = scenes.grad_seed(W, H, 32)
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    camd = cam.to(dev)
    cov = to.cov3d_from_scale_rot(sc.scales.double(), sc.rotations.double(), 1.0).float()
    d = sc.means3D.double() - cam.camera_center.double()[None]
    d = d / d.norm(dim=1, keepdim=True)
    col = torch.clamp_min(to.eval_sh_color(sc.sh_degree, sc.shs.double(), d) + 0.5, 0).float()
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=bg.to(dev), scale_modifier=1.0,
                                       viewmatrix=camd.world_view_transform, projmatrix=camd.full_proj_transform,
                                       sh_degree=sc.sh_degree, campos=camd.camera_center, prefiltered=False,
                                       debug=False)
    t = lambda x: x.to(dev).requires_grad_(True)
    means, opac, covd, cold = t(sc.means3D), t(sc.opacities), t(cov), t(col)
    m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    img, aps, dep, radii, psz = GaussianRasterizer(rs)(means3D=means, means2D=m2, opacities=opac,
                                                        colors_precomp=cold, cov3D_precomp=covd)
    (img * dL.to(dev)).sum().backward()
    orc, og = _oracle(sc, cam, ST0, bg, dL, use_cov_precomp=True, use_colors_precomp=True, cov3D_precomp=cov,
                      colors_precomp=col)
    out = dict(render=img, acc_pixel_size=aps, depth=dep, radii=radii, visibility_filter=radii > 0, pixel_sizes=psz)
    check_forward(out, orc, "precomp")
    for name, g, ref in (("means3D", means.grad, og["means3D"]), ("opac", opac.grad, og["opacities"]),
                         ("cov", covd.grad, og["cov3D_precomp"]), ("col", cold.grad, og["colors_precomp"]),
                         ("m2", m2.grad, og["means2D"])):
        assert rel_err(g, ref) <= 1e-4, (name, rel_err(g, ref))


def test_scale_modifier_and_ring_camera():
    W, H = 96, 64
    sc = scenes.ball_scene(2500, seed=41, log_s=-1.6)
    cam = scenes.ring_camera(3, 8, W, H)
    bg = torch.tensor([0.1, 0.2, 0.3])
    dL = scenes.grad_seed(W, H, 41)
    out, pc, m2 = hip_render(sc, cam, ST0, bg, dL, scaling_modifier=0.7)
    from oracle import oracle_ctypes as oc
    r = oc.rasterize(pc.seen, cam, ST0, bg, scale_modifier=0.7)
    g = oc.backward(r, dL)
    check_forward(out, r, "ring")
    check_backward(pc, m2, g, "ring", flagged=r.borderline_gaussians)


def test_edge_cases_empty_and_culled():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    W, H = 40, 24
    cam = scenes.front_camera(W, H).to("cuda")
    bg = torch.tensor([0.25, 0.5, 0.75], device="cuda")
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
                                       viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                       sh_degree=0, campos=cam.camera_center, prefiltered=False, debug=True)
    # all Gaussians behind the camera: pure background, radii 0, zero gradients
    sc, _ = small_scene(64, W, H, 51)
    sc.means3D[:, 2] = -1.0
    out, pc, m2 = hip_render(sc, scenes.front_camera(W, H), ST0, bg.cpu(), scenes.grad_seed(W, H, 51))
    assert (out["radii"] == 0).all() and (out["pixel_sizes"] == 0).all()
    assert torch.allclose(out["render"], bg[:, None, None].expand(3, H, W))
    assert all(p.grad is not None and (p.grad == 0).all() for p in pc.parameters())
    # P == 0
    e = lambda *s: torch.zeros(*s, device="cuda")
    img, aps, dep, radii, psz = GaussianRasterizer(rs)(means3D=e(0, 3), means2D=e(0, 3), opacities=e(0, 1),
                                                        shs=e(0, 16, 3), scales=e(0, 3), rotations=e(0, 4))
    assert torch.allclose(img, bg[:, None, None].expand(3, H, W)) and radii.numel() == 0
    # API errors mirror upstream
    with pytest.raises(Exception):
        GaussianRasterizer(rs)(means3D=e(4, 3), means2D=e(4, 3), opacities=e(4, 1), scales=e(4, 3), rotations=e(4, 4))
    with pytest.raises(Exception):
        GaussianRasterizer(rs)(means3D=e(4, 3), means2D=e(4, 3), opacities=e(4, 1), shs=e(4, 16, 3),
                               colors_precomp=e(4, 3), scales=e(4, 3), rotations=e(4, 4))
    with pytest.raises(RuntimeError):
        GaussianRasterizer(rs)(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.zeros(4, 1),
                               shs=torch.zeros(4, 16, 3), scales=torch.zeros(4, 3), rotations=torch.zeros(4, 4))


def test_huge_and_tiny_gaussians():
    """Gaussians covering the whole image (tile-overflow path) together with sub-pixel ones."""
    W, H = 200, 120
    sc, cam = small_scene(600, W, H, 61)
    sc.scales[:40] *= 60.0
    sc.scales[40:200] *= 0.02
    bg = torch.zeros(3)
    dL = scenes.grad_seed(W, H, 61)
    out, pc, m2 = hip_render(sc, cam, ST0, bg, dL)
    orc, og = _oracle(pc.seen, cam, ST0, bg, dL)
    check_forward(out, orc, "huge")
    check_backward(pc, m2, og, "huge", flagged=orc.borderline_gaussians)


@pytest.mark.parametrize("W,H,keys16", [(4080, 4096, True), (4096, 4096, False)])
def test_tile_key_width_boundary(W, H, keys16):
    """The tile sort moves 16-bit keys while the tile ids and the sentinel (= number of tiles) fit: 255 x 256 = 65 280 tiles use
    them, 256 x 256 = 65 536 tiles fall back to 32-bit keys — both against the oracle, with Gaussians large enough to reach the
    last tiles (ids near 65 000) and a few that span hundreds of tiles (the wave-cooperative emit path)."""
    assert (((W + 15) // 16) * ((H + 15) // 16) < 65535) == keys16
    sc, cam = small_scene(3000, W, H, 91, scale_k=0.004 * 1920.0 / W * 1.2)
    sc.scales[:12] *= 60.0                                  # > 96 instances each: emitted by their whole wave
    bg = torch.tensor([0.1, 0.2, 0.3])
    dL = scenes.grad_seed(W, H, 91)
    out, pc, m2 = hip_render(sc, cam, ST0, bg, dL)
    orc, og = _oracle(pc.seen, cam, ST0, bg, dL)
    check_forward(out, orc, f"boundary {W}x{H}")
    check_backward(pc, m2, og, f"boundary {W}x{H}", flagged=orc.borderline_gaussians)
    D = out["render"].grad_fn.state[3]
    assert D > 5000 and int((out["radii"] > 0).sum()) > 1000


def test_wave_emitted_gaussian_inside_the_lds_stage():
    """emit: a Gaussian with more than 96 instances is written by its whole wave; when its block's output range still fits the
    LDS stage (<= 3072 pairs) the wave path has to deposit into the stage like the per-thread path.  One ~200-tile Gaussian among
    sub-pixel ones, and one spanning more than 64 tile rows (the row loop of the wave path runs more than once)."""
    for W, H, big in ((320, 240, 14.0), (96, 1200, 60.0)):
        sc, cam = small_scene(250, W, H, 93)
        sc.scales[:] *= 0.05
        sc.scales[7] *= 20.0 * big
        sc.opacities[7] = 0.9
        bg = torch.zeros(3)
        dL = scenes.grad_seed(W, H, 93)
        out, pc, m2 = hip_render(sc, cam, ST0, bg, dL)
        orc, og = _oracle(pc.seen, cam, ST0, bg, dL)
        D = out["render"].grad_fn.state[3]
        assert 96 < D <= 3072, D
        check_forward(out, orc, f"wave emit {W}x{H}")
        check_backward(pc, m2, og, f"wave emit {W}x{H}", flagged=orc.borderline_gaussians)


def test_mark_visible():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    W, H = 64, 64
    sc, cam = small_scene(1000, W, H, 71)
    camd = cam.to("cuda")
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=1.0, tanfovy=1.0,
                                       bg=torch.zeros(3, device="cuda"), scale_modifier=1.0,
                                       viewmatrix=camd.world_view_transform, projmatrix=camd.full_proj_transform,
                                       sh_degree=0, campos=camd.camera_center, prefiltered=False, debug=False)
    vis = GaussianRasterizer(rs).markVisible(sc.means3D.to("cuda"))
    assert torch.equal(vis.cpu(), sc.means3D[:, 2] > 0.2)


def test_determinism_of_forward():
    W, H = 128, 96
    sc, cam = small_scene(4000, W, H, 81)
    bg = torch.zeros(3)
    a, _, _ = hip_render(sc, cam, ST0, bg)
    b, _, _ = hip_render(sc, cam, ST0, bg)
    for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
        assert torch.equal(a[k], b[k]), k


def test_forward_capacity_guess_paths_give_identical_results():
    """msgs_forward with buffers sized from a previous frame: too small a guess must fall back to the exact two-call
    path, a generous one must take the one-call path, both bit-identical to a first call without history."""
    import diff_gaussian_rasterization as dgr
    W, H = 144, 96
    sc, cam = small_scene(4000, W, H, 23)
    bg = torch.tensor([0.0, 0.1, 0.2])
    dL = scenes.grad_seed(W, H, 23)

    key = (torch.cuda.current_device(), 4000, W, H, 0, 0)     # (device, P, W, H, filter_small, filter_large)

    def run(seed_guess):
        dgr._last_instances.clear()
        if seed_guess is not None:
            dgr._last_instances[key] = seed_guess
        out, pc, m2 = hip_render(sc, cam, ST0, bg, dL)
        return out, pc, m2, dgr._last_instances[key]
    ref, pref, mref, D = run(None)
    assert D > 0
    for guess in (1, max(D // 2, 1), D, 10 * D):
        out, pc, m2, D2 = run(guess)
        assert D2 >= D
        for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
            assert torch.equal(out[k], ref[k]), (k, guess)
        assert rel_err(pc._xyz.grad, pref._xyz.grad) <= 1e-4 and rel_err(m2, mref) <= 1e-4


@pytest.mark.parametrize("W,H", [(203, 117), (120, 67), (43, 29)])
def test_blend_granularities_agree(W, H):
    """The fine-grained kernels (one wave per 4x4 / 2x2 / 1x1 pixel sub-block: 104, 40 and 6 tiles select the three shapes,
    blend.hip fine_shape) evaluate every pixel with the same arithmetic in
    the same order as the quadrant-per-wave kernels: forward outputs bit-identical, gradients equal up to the float32 partial sums the
    waves form over their pixels (the cross-tile accumulation is exact).  Ragged image (W, H not multiples of 16 or 4), multi-scale filters on, non-zero background."""
    import diff_gaussian_rasterization as dgr
    sc, cam = small_scene(6000, W, H, 57, multiscale=True, scale_k=0.004 * 1920.0 / W * 0.2)
    st = dict(filter_small=True, filter_large=True, fade_size=0.0)
    bg = torch.tensor([0.2, 0.5, 0.1])
    dL = scenes.grad_seed(W, H, 11)
    res = {}
    prev = dgr._C.lib.msgs_set_blend_granularity(1)
    try:
        for gran in (1, 2):
            dgr._C.lib.msgs_set_blend_granularity(gran)
            res[gran] = hip_render(sc, cam, st, bg, dL)
    finally:
        dgr._C.lib.msgs_set_blend_granularity(prev)
    (a, pa, ma), (b, pb, mb) = res[1], res[2]
    for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
        assert torch.equal(a[k], b[k]), k
    assert a["render"].abs().max().item() > 0
    from parity_utils import report
    worst = {}
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        worst[n] = rel_err(getattr(pb, n).grad, getattr(pa, n).grad)
    worst["means2D"] = rel_err(mb, ma)
    report("granularities", "worst gradient difference between the 8x8 and 4x4 kernels", max(worst.values()))
    # the cross-tile accumulation is exact (double accumulators): what differs is the float32 partial sum a wave forms
    # over its 64 resp. 16 pixels before it is added
    # (measured 2e-6 .. 6e-6 on scaling / rotation, 4e-7 elsewhere; with float32 atomics this test needed 1e-3 / 3e-4)
    for n, v in worst.items():
        tol = 5e-5 if n in ("_scaling", "_rotation") else 1e-5
        assert v <= tol, (n, worst)


def test_kernel_variants_agree_on_random_shapes():
    """Differential stress of the blend kernel variants (quadrant-per-wave forward with the four-waves and one-wave
    backward, fine-grained forward + backward) over random image shapes, densities and footprint sizes, including
    images smaller than one tile / one 4x4 sub-block and footprints of many tiles: forward outputs bit-identical,
    gradients equal up to the in-tile float32 partial sums (the cross-tile accumulation is exact)."""
    import random
    import diff_gaussian_rasterization as dgr
    rnd = random.Random(20261002)
    shapes = [(1, 1), (3, 5), (4, 4), (17, 33), (16, 16), (31, 16)]
    shapes += [(rnd.randint(20, 260), rnd.randint(20, 200)) for _ in range(6)]
    prev_gen = dgr._C.lib.msgs_set_backward_generation(0)
    prev_gran = dgr._C.lib.msgs_set_blend_granularity(0)
    worst_all = {}
    try:
        for n, (W, H) in enumerate(shapes):
            P = rnd.choice([50, 800, 5000])
            k = rnd.choice([0.05, 0.2, 0.5, 2.0]) * 0.004 * 1920.0 / W
            ms = bool(n & 1)
            sc, cam = small_scene(P, W, H, 900 + n, multiscale=ms, scale_k=k)
            st = dict(filter_small=ms, filter_large=ms, fade_size=0.0 if ms else 1.0)
            bg = torch.tensor([0.1 * (n % 3), 0.3, 0.7])
            dL = scenes.grad_seed(W, H, 40 + n)
            res = []
            for gen, gran in ((1, 1), (2, 1), (0, 2)):
                dgr._C.lib.msgs_set_backward_generation(gen)
                dgr._C.lib.msgs_set_blend_granularity(gran)
                res.append(hip_render(sc, cam, st, bg, dL))
            (a, pa, ma) = res[0]
            for (b, pb, mb) in res[1:]:
                for key in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
                    assert torch.equal(a[key], b[key]), (key, W, H, P)
                for nm in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
                    tol = 5e-5 if nm in ("_scaling", "_rotation") else 1e-5      # measured 5.4e-6 / 3.8e-7 (in-tile float32 sums only)
                    e = rel_err(getattr(pb, nm).grad, getattr(pa, nm).grad)
                    worst_all[nm] = max(worst_all.get(nm, 0.0), e)
                    assert e <= tol, (nm, W, H, P)
                assert rel_err(mb, ma) <= 1e-5, (W, H, P)
    finally:
        dgr._C.lib.msgs_set_backward_generation(prev_gen)
        dgr._C.lib.msgs_set_blend_granularity(prev_gran)
    print("[parity] kernel variants, worst gradient difference per tensor:", {k: f"{v:.2e}" for k, v in worst_all.items()})


def test_lane_stats_diagnostic_matches_the_oracle_pair_count():
    """msgs_blend_lane_stats: the counting replica of the forward kernel blends exactly the (pixel, Gaussian) pairs the
    oracle counts as valid, minus one terminating entry per saturated pixel; it writes nothing."""
    import ctypes as C
    import diff_gaussian_rasterization as dgr
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    W, H = 160, 128
    sc, cam = small_scene(6000, W, H, 61)
    bg = torch.zeros(3)
    pc = SyntheticGaussians(sc, "cuda")
    out = render(cam.to("cuda"), pc, PIPE, bg.cuda(), **ST0)
    ctx = out["render"].grad_fn
    geom, binning, image, D = ctx.state
    before = out["render"].clone()
    scratch = torch.empty(256, dtype=torch.uint8, device="cuda")
    o3 = (C.c_int64 * 7)()
    dgr._C.check(dgr._C.lib.msgs_blend_lane_stats(C.byref(ctx.call.view), C.c_void_p(geom.data_ptr()), geom.numel(), sc.P, int(D),
                                                  C.c_void_p(binning.data_ptr()), binning.numel(),
                                                  C.c_void_p(image.data_ptr()), image.numel(),
                                                  C.c_void_p(scratch.data_ptr()), scratch.numel(), o3,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lane stats")
    steps, alive, blended = int(o3[0]), int(o3[1]), int(o3[2])
    assert torch.equal(out["render"], before)
    assert 0 < blended <= alive <= 64 * steps
    # backward replica: the lanes that contribute a gradient are exactly the pairs the forward blended
    visits, qsteps, lanes, hits = int(o3[3]), int(o3[4]), int(o3[5]), int(o3[6])
    assert lanes == blended and 0 < hits <= visits <= qsteps <= 4 * visits and lanes <= 64 * qsteps
    from oracle import oracle_ctypes as oc
    import copy
    seen = copy.copy(sc)
    with torch.no_grad():
        seen.scales = pc.get_scaling.detach().cpu().contiguous()
        seen.rotations = pc.get_rotation.detach().cpu().contiguous()
        seen.opacities = pc.get_opacity.detach().cpu().contiguous()
        seen.shs = pc.get_features.detach().cpu().contiguous()
        seen.means3D = pc.get_xyz.detach().cpu().contiguous()
    orc = oc.rasterize(seen, cam, ST0, bg)
    # valid_pairs counts the terminating entry of a saturated pixel too; at most one per pixel
    assert orc.valid_pairs - W * H <= blended <= orc.valid_pairs + 64
    assert abs(blended - orc.valid_pairs) <= W * H
