"""Full-size GPU tests on the BASELINE.json configurations: direct comparison with the CPU oracle at C2, C3, the C3 pyramid
and C5, plus size-independent properties (sortedness of every tile list, checksum of tile ranges, background and gradient
linearity, determinism).

Gradient tolerance at full size: the north star's 1e-4 max-norm on EVERY tensor at C2 and at every pyramid level, and on five
of the seven tensors (xyz, both SH tensors, opacity, means2D) everywhere else.  dL/dscaling and dL/drotation end the conic
-> covariance chain of K8, which multiplies whatever two float32 evaluations of the blend backward differ by with the
squared aspect ratio of the footprint; they carry per-config ceilings at ~1.5x the measured value (parity_utils.GRAD_CEILINGS,
profiles/r3_parity.md) instead of one flat number, so a regression at any config trips its own ceiling.  That this residual is
the blend's per-pixel rounding x conditioning — and not K8 / K9 — is established by test in tests/test_k8_isolation_gpu.py.
Also asserted: the 99th percentile of the per-Gaussian error relative to the Gaussian's own gradient <= 1e-4 (measured
8e-6) — all on the Gaussians without a borderline alpha decision, whose achieved fraction every test prints."""
TIGHT_RTOL = 1e-4
LINEARITY_RTOL = {"_scaling": 1e-3, "_rotation": 3e-4}       # measured 2.5e-4 / 5.9e-5; the other tensors <= 1.5e-6
LINEARITY_RTOL_DEFAULT = 1e-5
FULL_Q99 = 1e-4

import pytest
import torch

import scenes
from parity_utils import (PIPE, check_against_truth, check_backward, check_forward, grad_ceilings, hip_render, rel_err,
                          rel_err_reported)

pytestmark = pytest.mark.gpu


def _oracle(scene, cam, st, bg, dL):
    from oracle import oracle_ctypes as oc
    r = oc.rasterize(scene, cam, st, bg)
    return r, oc.backward(r, dL)


def test_config_c2_forward_backward_vs_oracle():
    """configs[1]: 100k Gaussians, 800x800, SH degree 3, fwd+bwd (tolerance check)."""
    sc, cam, st = scenes.config("C2")
    bg = torch.tensor([0.0, 0.0, 0.0])
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 1)
    out, pc, m2 = hip_render(sc, cam, st, bg, dL)
    orc, og = _oracle(pc.seen, cam, st, bg, dL)
    check_forward(out, orc, "C2")
    check_backward(pc, m2, og, "C2", flagged=orc.borderline_gaussians, rtol=TIGHT_RTOL, rtol_by_key=grad_ceilings("C2"),
                   q99_tol=FULL_Q99)
    check_against_truth("C2", pc.seen, cam, st, bg, dL, out, pc, m2, orc, og)


def test_config_c2_three_way_with_float64_truth():
    """C2 against the float64 autograd evaluation of the same pipeline (oracle/torch_oracle.py forward_backward_tiled):
    the HIP gradients are as close to the truth as the float32 reference algorithm is — the distance of BOTH from the
    truth (3e-4 .. 1.7e-3 per tensor) is what makes 1e-4 between two float32 evaluations a statement about rounding
    order, not about correctness.  The atomic mode's run-to-run noise is printed beside it."""
    import diff_gaussian_rasterization as dgr
    from oracle import torch_oracle as to
    from parity_utils import report
    sc, cam, st = scenes.config("C2")
    bg = torch.tensor([0.0, 0.0, 0.0])
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 1)
    out, pc, m2 = hip_render(sc, cam, st, bg, dL)
    _, pc_b, m2_b = hip_render(sc, cam, st, bg, dL)
    orc, og = _oracle(pc.seen, cam, st, bg, dL)
    t_out, tg = to.forward_backward_tiled(pc.seen, cam, st, bg, dL)
    flagged = orc.borderline_gaussians | (t_out["radii"] != orc.radii)
    okpx = ~(orc.borderline.bool() | t_out["borderline"])
    e_orc = (orc.color.double() - t_out["color"]).abs()[:, okpx].max().item()
    e_hip = (out["render"].detach().cpu().double() - t_out["color"]).abs()[:, okpx].max().item()
    report("C2", "forward, oracle_f32 vs float64 truth", e_orc)
    report("C2", "forward, HIP vs float64 truth", e_hip)
    assert e_hip <= 1.25 * e_orc + 1e-6
    # gradients: HIP vs the truth (the verification mode against the float32 oracle: tests/test_literal_gpu.py, 1e-4 flat)
    to_f32 = {k: v.float() for k, v in tg.items()}
    w_hip_truth = check_backward(pc, m2, to_f32, "C2 HIP vs truth", flagged=flagged, rtol=1.0)
    names = {"means3D": "_xyz", "features_dc": "_features_dc", "features_rest": "_features_rest", "opacity": "_opacity",
             "scaling": "_scaling", "rotation": "_rotation"}
    noise = {k: rel_err(getattr(pc_b, n).grad, getattr(pc, n).grad) for k, n in names.items()}
    report("C2", "atomic run-to-run noise, worst tensor", max(noise.values()))
    # the oracle's own distance from the truth (gradients w.r.t. the activated inputs, where both are defined)
    w_orc_truth = {}
    for k in ("means3D", "opacities", "scales", "rotations", "shs", "means2D"):
        ref = tg[k].double()
        scale = max(ref.abs().max().item(), 1e-30)
        d = (og[k].double().reshape(ref.shape) - ref).abs().reshape(ref.shape[0], -1).max(dim=1).values
        w_orc_truth[k] = (d[~flagged].max() / scale).item()
    report("C2", "oracle_f32 vs float64 truth, worst tensor (activated-input space)", max(w_orc_truth.values()))
    report("C2", "HIP vs float64 truth, worst tensor (leaf space)", max(w_hip_truth.values()))
    assert max(w_hip_truth.values()) <= 2.0 * max(w_orc_truth.values()) + 1e-4


@pytest.fixture(scope="module")
def c3():
    sc, cam, st = scenes.config("C3")
    return sc, cam, st


def test_config_c3_forward_backward_vs_oracle(c3):
    """configs[2]: 1M Gaussians, 1920x1080, multi-scale fields, filter_small + filter_large, fade 0."""
    sc, cam, st = c3
    bg = torch.tensor([0.0, 0.0, 0.0])
    dL = scenes.grad_seed(cam.image_width, cam.image_height, 2)
    out, pc, m2 = hip_render(sc, cam, st, bg, dL)
    orc, og = _oracle(pc.seen, cam, st, bg, dL)
    check_forward(out, orc, "C3")
    check_backward(pc, m2, og, "C3", flagged=orc.borderline_gaussians, rtol=TIGHT_RTOL, rtol_by_key=grad_ceilings("C3"),
                   q99_tol=FULL_Q99)
    assert (out["radii"] > 0).sum().item() == (orc.radii > 0).sum().item()
    # the ceiling above 1e-4 on dL/dscaling is a regression guard; the PROPERTY is: no farther from the float64 truth than the
    # float32 reference algorithm itself
    check_against_truth("C3", pc.seen, cam, st, bg, dL, out, pc, m2, orc, og)


def _forward_state(sc, cam, st, bg):
    """forward through the op, returning outputs + the raw (geom, binning, image, D) state of the ctx"""
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
    out = render(cam.to("cuda"), pc, PIPE, bg.to("cuda"), **st)
    ctx = out["render"].grad_fn
    return out, pc, ctx


def test_c3_tile_lists_are_depth_sorted_and_ranges_checksum(c3):
    """Properties of the binning at full size: every tile range is sorted by (depth bits, Gaussian index), ranges
    tile [0, D) exactly, and every listed Gaussian is a rendered one."""
    sc, cam, st = c3
    out, pc, ctx = _forward_state(sc, cam, st, torch.zeros(3))
    geom, binning, image, D = ctx.state
    P = sc.P
    W, H = cam.image_width, cam.image_height
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    ioff = (8 * tiles + 255) // 256 * 256 + 512                          # BinningLayout: tile ranges, 64 D_trav accumulators, then the ids
    ids = binning[ioff:ioff + 4 * D].view(torch.int32).long()
    ranges = binning[:8 * tiles].view(torch.int32).view(tiles, 2).long()
    lo, hi = ranges[:, 0], ranges[:, 1]
    nonempty = hi > lo
    n_sentinel = D - hi.max().item()                                     # surplus slots parked behind the last tile
    assert (hi - lo).sum().item() + n_sentinel == D                      # checksum of checksums
    assert 0 <= n_sentinel <= D // 200
    srt = torch.sort(lo[nonempty]).values
    assert srt[0].item() == 0 and torch.equal(torch.sort(hi[nonempty]).values[:-1], srt[1:])   # contiguous cover
    rec = geom[:48 * P].view(torch.float32).view(P, 12)
    depth_bits = rec[:, 9].contiguous().view(torch.int32).long()[ids]
    key = depth_bits * (2 ** 21) + ids                                   # (depth bits, index) as one integer
    # positions that start a tile are exempt from the "non-decreasing" check
    starts = torch.zeros(D, dtype=torch.bool, device=ids.device)
    starts[lo[nonempty]] = True
    real = torch.arange(D, device=ids.device) < hi.max()                # exclude the sentinel tail
    ok = (key[1:] > key[:-1]) | starts[1:] | ~real[1:]
    assert bool(ok.all()), f"{(~ok).sum().item()} out-of-order neighbours"
    assert bool((out["radii"][ids] > 0).all())


def test_c3_background_linearity_and_determinism(c3):
    sc, cam, st = c3
    a, _, _ = hip_render(sc, cam, st, torch.zeros(3))
    b, _, _ = hip_render(sc, cam, st, torch.tensor([1.0, 1.0, 1.0]))
    c, _, _ = hip_render(sc, cam, st, torch.tensor([0.25, 0.5, 0.75]))
    a2, _, _ = hip_render(sc, cam, st, torch.zeros(3))
    for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
        assert torch.equal(a[k], a2[k]), k                               # bitwise deterministic forward
    T = b["render"] - a["render"]                                        # = final transmittance, per channel
    assert (T[0] - T[1]).abs().max() <= 2e-6 and (T[0] - T[2]).abs().max() <= 2e-6
    assert T.min() >= -1e-6 and T.max() <= 1 + 1e-6
    want = a["render"] + T[0][None] * torch.tensor([0.25, 0.5, 0.75], device="cuda")[:, None, None]
    assert (c["render"] - want).abs().max() <= 3e-6
    assert torch.equal(a["acc_pixel_size"], b["acc_pixel_size"]) and torch.equal(a["depth"], b["depth"])


def test_c3_backward_is_linear_in_dL(c3):
    sc, cam, st = c3
    W, H = cam.image_width, cam.image_height
    bg = torch.tensor([0.2, 0.2, 0.2])
    d1, d2 = scenes.grad_seed(W, H, 7), scenes.grad_seed(W, H, 8)
    _, p1, m1 = hip_render(sc, cam, st, bg, d1)
    _, p2, m2 = hip_render(sc, cam, st, bg, d2)
    _, p3, m3 = hip_render(sc, cam, st, bg, 2.0 * d1 - 0.5 * d2)
    # every run is reproducible, so what is left is float32 rounding inside one pixel's chain (linear up to rounding) and K8's
    # amplification of it on dL/dscale, dL/drotation (the conic -> covariance cancellation, profiles/r2_parity_floor.md)
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        want = 2.0 * getattr(p1, n).grad - 0.5 * getattr(p2, n).grad
        assert rel_err_reported("C3 linearity", n, getattr(p3, n).grad, want) <= LINEARITY_RTOL.get(n, LINEARITY_RTOL_DEFAULT), n
    assert rel_err_reported("C3 linearity", "means2D", m3, 2.0 * m1 - 0.5 * m2) <= LINEARITY_RTOL_DEFAULT


def test_c3_multiscale_pyramid_levels(c3):
    """render at scale 2^k, k = 1..6 ((W,H) = (int(1920/2^k), int(1080/2^k)), utils/camera_utils.py:38-39) against
    the oracle: non-multiple-of-16 sizes and coarser levels where the filters bite.  Gaussians whose pixel size is
    within 1e-4 (relative) of their min / max threshold stay IN the scene: the filter decision hinges on the last bit of
    logf there, which differs between libm and the GPU, and the oracle flags them (filter_edge) together with every
    pixel they reach and every Gaussian blended at such a pixel (tests/test_oracle_cpu.py shows the flags cover everything
    a flipped decision changes)."""
    sc, _, st = c3
    for k in (1, 3, 6):
        W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
        cam = scenes.front_camera(W, H)
        bg = torch.zeros(3)
        dL = scenes.grad_seed(W, H, 30 + k)
        out, pc, m2 = hip_render(sc, cam, st, bg, dL)
        orc, og = _oracle(pc.seen, cam, st, bg, dL)
        assert orc.filter_edge.float().mean() < 1e-3
        check_forward(out, orc, f"C3@k={k}")
        check_backward(pc, m2, og, f"C3@k={k}", flagged=orc.borderline_gaussians, rtol=TIGHT_RTOL,
                       rtol_by_key=grad_ceilings("C3@k"), q99_tol=FULL_Q99)
        assert (out["radii"] > 0).sum() > 1000


def test_config_c5_stress_4k():
    """configs[4]: 5M Gaussians, 3840x2160, multi-scale fields (HBM-bound stress: tens of millions of instances,
    32400 tiles -> 15-bit tile ids).  Forward + backward against the oracle, binning checksum."""
    sc, cam, st = scenes.config("C5")
    bg = torch.tensor([0.0, 0.0, 0.0])
    W, H = cam.image_width, cam.image_height
    dL = scenes.grad_seed(W, H, 5)
    import diff_gaussian_rasterization as dgr
    prev, dgr.slab_policy = dgr.slab_policy, "never"     # the checksum below reads the single-pass list layout (an earlier test of
    try:                                                 # this process may have published D / D_trav for this view shape:
        out, pc, m2 = hip_render(sc, cam, st, bg, dL)    # tests/test_slab_gpu.py covers the slab layout)
    finally:
        dgr.slab_policy = prev
    ctx = out["render"].grad_fn
    geom, binning, image, D = ctx.state
    assert D > 10_000_000
    assert torch.isfinite(out["render"]).all() and all(torch.isfinite(p.grad).all() for p in pc.parameters())
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = binning[:8 * tiles].view(torch.int32).view(tiles, 2).long()                  # BinningLayout: ranges first
    n_sentinel = D - ranges[:, 1].max().item()
    assert (ranges[:, 1] - ranges[:, 0]).sum().item() + n_sentinel == D and 0 <= n_sentinel <= D // 200
    orc, og = _oracle(pc.seen, cam, st, bg, dL)
    check_forward(out, orc, "C5")
    check_backward(pc, m2, og, "C5", flagged=orc.borderline_gaussians, rtol=TIGHT_RTOL, rtol_by_key=grad_ceilings("C5"),
                   q99_tol=FULL_Q99)
    check_against_truth("C5", pc.seen, cam, st, bg, dL, out, pc, m2, orc, og)
