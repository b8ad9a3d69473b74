"""Shared helpers of the parity tests: run the HIP path through the drop-in Python API, run the CPU
oracle on the same seeded inputs, compare with the north-star tolerances.

Tolerances (BASELINE.json north_star):
  forward  <= 1e-5 abs per pixel
  backward <= 1e-4 rel per gradient tensor  (||d||_inf / max(||ref||_inf, eps))
Pixels on which the oracle reports a BORDERLINE discrete decision (an alpha within float32 rounding
of 1/255, or a transmittance within rounding of 1e-4: oracle/msgs_oracle.cpp) may legitimately flip
between two float32 implementations; they are excluded from the strict forward check, bounded by a
loose one (<= 2/255), and must stay rare (< 0.5 % of pixels).

The same holds for gradients: a rim pixel (alpha ~ 1/255) enters the conic gradients weighted by dx^2,
so ONE flipped skip decision moves that Gaussian's scale / rotation / mean gradients by about a pixel's
worth (~1 %).  Gaussians for which the oracle saw a borderline alpha (oracle: borderline_gaussians) are
excluded from the strict 1e-4 check, bounded by a loose one (5e-2 of the tensor's max norm) and must stay
rare (< 3 % of the Gaussians).
"""
import types

import torch

import scenes

FWD_ATOL = 1e-5
BWD_RTOL = 1e-4
BORDERLINE_PIXEL_BUDGET = 0.0045      # default ceiling on the pixels excluded from the strict forward check (small test scenes)
# Per-config ceilings: 1.2 x the ORACLE's own count on that config (the fraction is a property of the scene and of the
# oracle's flags, computed on the inputs: recount with oracle_ctypes.rasterize; round 5, with the conditioning-aware alpha window
# of msgs_oracle.cpp: C2 0.0203 %, C3 0.344 %, C3@k=1 0.348 %, k=3 0.824 %, k=6 0.417 %, C4 views 0 / 3 / 6 0.163 / 0.166 /
# 0.168 %, C5 0.342 %; with the fixed 2e-5 window of rounds 1-4: 0.0175 / 0.334 / 0.339 / 0.809 / 0.417 / 0.159 / 0.161 / 0.163 /
# 0.331 %) — so that a config whose exclusions grow trips its own bound instead of hiding under the largest one.
BORDERLINE_PIXEL_BUDGETS = {
    "C2": 2.45e-4, "C3": 4.15e-3, "C3@k=1": 4.2e-3, "C3@k=3": 9.9e-3, "C3@k=6": 5.0e-3,
    "C4v0": 1.96e-3, "C4v3": 1.99e-3, "C4v6": 2.02e-3, "C5": 4.1e-3,
}


def pixel_budget(name):
    return BORDERLINE_PIXEL_BUDGETS.get(name.split(" ")[0], BORDERLINE_PIXEL_BUDGET)

# Per-config ceilings on the two gradient tensors that end the conic -> 2-D covariance -> 3-D covariance chain of K8
# (dL/dscaling, dL/drotation), max-norm relative, at ~1.5x the measured value (profiles/r3_parity.md; the HIP numbers are
# reproducible run to run).  Everything NOT listed is asserted at the north star's 1e-4 — all seven tensors at C2 and at
# every pyramid level, five of seven elsewhere.  tests/test_k8_isolation_gpu.py shows by test where the listed residuals
# come from: K8 + K9 fed the oracle's own sums agree with the oracle to ~1e-6, dL/dcov3D meets 1e-4 at C2 and C3, and the
# printed amplification factors are what multiplies the blend backward's float32 rounding on these two tensors.
# (The VERIFICATION mode — msgs_set_deterministic, literal.hip — carries no ceilings: tests/test_literal_gpu.py asserts 1e-4 flat
#  on all seven tensors at every config, against the oracle evaluated the same way.  What follows is the DEFAULT mode, a different
#  float32 evaluation of the same algorithm.)
GRAD_CEILINGS = {
    "C2": {},
    "C3": {"scaling": 5e-4},
    "C3@k": {},
    "C5": {"scaling": 2.5e-4},
    "C4v0": {"scaling": 1.3e-4, "rotation": 2.2e-4}, "C4v3": {"scaling": 1.5e-3, "rotation": 4e-4},
    "C4v6": {"scaling": 1.4e-4},
}


def grad_ceilings(config, deterministic=False):
    assert not deterministic, "the verification mode has no ceilings (tests/test_literal_gpu.py)"
    return dict(GRAD_CEILINGS[config])
PIPE = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)   # = gaussian_renderer.PIPE


def hip_render(scene, cam, settings, bg, dL_dcolor=None, pipe=PIPE, device="cuda", override_color=None,
               scaling_modifier=1.0):
    """Forward (+ backward when dL_dcolor is given) through gaussian_renderer.render -> HIP library."""
    from gaussian_renderer import render
    from synthetic_model import SyntheticGaussians
    pc = SyntheticGaussians(scene, device, requires_grad=dL_dcolor is not None)
    # what the op actually receives: exp / sigmoid / normalize evaluated by torch ON THE GPU differ from the
    # CPU-generated activated values by ~1 ulp, enough to flip a ceil() or a filter comparison once per
    # million Gaussians.  The oracle must be fed these tensors (pc.seen), not the original scene.
    import copy
    with torch.no_grad():
        seen = copy.copy(scene)
        seen.scales = pc.get_scaling.detach().cpu().contiguous()
        seen.rotations = pc.get_rotation.detach().cpu().contiguous()
        seen.opacities = pc.get_opacity.detach().cpu().contiguous()
        seen.shs = pc.get_features.detach().cpu().contiguous()
        seen.means3D = pc.get_xyz.detach().cpu().contiguous()
    pc.seen = seen
    camd = cam.to(device)
    bgd = bg.to(device)
    if dL_dcolor is None:
        with torch.no_grad():
            out = render(camd, pc, pipe, bgd, scaling_modifier=scaling_modifier, override_color=override_color,
                         **settings)
        return out, pc, None
    out = render(camd, pc, pipe, bgd, scaling_modifier=scaling_modifier, override_color=override_color, **settings)
    (out["render"] * dL_dcolor.to(device)).sum().backward()
    torch.cuda.synchronize()
    return out, pc, out["viewspace_points"].grad


def rel_err(a, ref, rows=None):
    """||a - ref||_inf over `rows` (all rows if None) / max(||ref||_inf over ALL rows, eps)"""
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    scale = max(ref.abs().max().item(), 1e-20) if ref.numel() else 1.0
    d = (a.reshape(ref.shape) - ref).abs()
    if rows is not None:
        d = d[rows]
    return (d.max().item() if d.numel() else 0.0) / scale


REPORT = []          # (name, what, value): achieved exclusion fractions / errors, printed by the tests that want them


def report(name, what, value):
    REPORT.append((name, what, value))
    print(f"[parity] {name}: {what} = {value:.3e}")


def rel_err_reported(name, what, a, ref, rows=None):
    """rel_err() that also prints / records the achieved value (pytest -s), so asserted tolerances can be read against it"""
    e = rel_err(a, ref, rows)
    report(name, what, e)
    return e


def check_forward(out, orc, name=""):
    ok = ~orc.borderline.bool()
    frac_bl = 1.0 - ok.float().mean().item()
    budget = pixel_budget(name)
    report(name, f"borderline pixel fraction (bound {budget:g})", frac_bl)
    assert frac_bl < budget, f"{name}: too many borderline pixels ({frac_bl:.5f} >= {budget:g})"
    col = out["render"].detach().cpu()
    d = (col - orc.color).abs()
    strict = d[:, ok].max().item() if ok.any() else 0.0
    assert strict <= FWD_ATOL, f"{name}: forward colour max abs diff {strict:.3e} > {FWD_ATOL}"
    assert d.max().item() <= 2.0 / 255.0 + 1e-5, f"{name}: borderline pixel off by {d.max().item():.3e}"
    # acc_pixel_size / depth are sums of (value * weight): compare relative to the value scale
    for key, ref in (("acc_pixel_size", orc.acc_pixel_size), ("depth", orc.depth)):
        dd = (out[key].detach().cpu() - ref).abs()
        scale = max(ref.abs().max().item(), 1.0)
        m = dd[ok].max().item() if ok.any() else 0.0
        assert m <= FWD_ATOL * scale, f"{name}: {key} max abs diff {m:.3e} (scale {scale:.2f})"
    # radii: bit-equal, except that a Gaussian whose multi-scale filter decision sits within rounding of its threshold
    # (oracle: filter_edge) may be rendered by one implementation and dropped by the other — radius or 0, nothing else
    got_r = out["radii"].cpu()
    edge = getattr(orc, "filter_edge", None)
    edge = edge if edge is not None else torch.zeros_like(got_r, dtype=torch.bool)
    assert torch.equal(got_r[~edge], orc.radii[~edge]), f"{name}: radii differ"
    if edge.any():
        ge, oe = got_r[edge], orc.radii[edge]
        assert bool(((ge == oe) | (ge == 0) | (oe == 0)).all()), f"{name}: radii of filter-edge Gaussians differ"
        report(name, "filter-edge Gaussians (decision may flip), flipped", float(((ge == 0) != (oe == 0)).sum().item()))
    assert torch.equal(out["visibility_filter"].cpu(), got_r > 0)
    dps = (out["pixel_sizes"].cpu() - orc.pixel_sizes).abs()
    assert (dps <= 1e-4 * orc.pixel_sizes.abs().clamp_min(1.0)).all(), f"{name}: pixel_sizes differ {dps.max():.3e}"
    return strict


LOOSE_RTOL = 5e-2


def own_relative_quantile(got, ref, rows, q=0.99):
    """q-quantile over `rows` of ||d_i||_inf / max(||ref_i||_inf, 1e-3 ||ref||_inf) (per-Gaussian error relative
    to the Gaussian's own gradient magnitude)"""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    P = ref.shape[0]
    d = (got.reshape(ref.shape) - ref).abs().reshape(P, -1).max(dim=1).values
    own = ref.abs().reshape(P, -1).max(dim=1).values.clamp_min(1e-3 * max(ref.abs().max().item(), 1e-30))
    e = (d / own)[rows]
    if e.numel() == 0:
        return 0.0
    k = max(int(q * e.numel()) - 1, 0)
    return torch.sort(e).values[k].item()


def leaf_space(pc, m2grad, ograds):
    """{tensor name: (HIP gradient, reference gradient in float64)} in the space of the model's LEAF parameters.  pc holds
    RAW parameters; an oracle returns grads w.r.t. the ACTIVATED inputs, so they are pushed through the same torch
    activations (exp / sigmoid / normalize / cat) on CPU in float64."""
    dt = torch.float64
    pairs = {"means3D": (pc._xyz.grad, ograds["means3D"].to(dt))}
    if "shs" in ograds:
        pairs["features_dc"] = (pc._features_dc.grad, ograds["shs"][:, :1].to(dt))
        pairs["features_rest"] = (pc._features_rest.grad, ograds["shs"][:, 1:].to(dt))
    raw = pc._opacity.detach().cpu().to(dt)
    s = torch.sigmoid(raw)
    pairs["opacity"] = (pc._opacity.grad, ograds["opacities"].to(dt).view_as(raw) * s * (1 - s))
    if "scales" in ograds:
        pairs["scaling"] = (pc._scaling.grad, ograds["scales"].to(dt) * torch.exp(pc._scaling.detach().cpu().to(dt)))
        q = pc._rotation.detach().cpu().to(dt).requires_grad_(True)
        torch.nn.functional.normalize(q).backward(ograds["rotations"].to(dt))
        pairs["rotation"] = (pc._rotation.grad, q.grad)
    pairs["means2D"] = (m2grad, ograds["means2D"].to(dt))
    return pairs


def check_backward(pc, m2grad, ograds, name="", rtol=BWD_RTOL, flagged=None, q99_tol=None, rtol_by_key=None):
    """HIP leaf gradients against an oracle's (leaf_space).
    `flagged` [P] bool = oracle's borderline Gaussians (strict check on the others, loose on these).
    `rtol_by_key` {tensor name: tolerance} overrides `rtol` per tensor (names: means3D, features_dc, features_rest,
    opacity, scaling, rotation, means2D)."""
    P = pc._xyz.shape[0]
    if flagged is None:
        flagged = torch.zeros(P, dtype=torch.bool)
    flagged = flagged.cpu()
    report(name, "borderline Gaussian fraction (bound 3e-2)", flagged.float().mean().item())
    assert flagged.float().mean().item() < 0.03, f"{name}: {flagged.float().mean().item():.4f} of the Gaussians borderline"
    clean = ~flagged
    pairs = leaf_space(pc, m2grad, ograds)
    worst = {}
    for k, (got, ref) in pairs.items():
        worst[k] = rel_err(got, ref, clean)
        loose = rel_err(got, ref, flagged) if flagged.any() else 0.0
        assert loose <= LOOSE_RTOL, f"{name}: grad {k} on borderline Gaussians off by {loose:.3e}"
        if q99_tol is not None:
            q = own_relative_quantile(got, ref, clean)
            assert q <= q99_tol, f"{name}: grad {k}: 99th percentile of the per-Gaussian relative error {q:.3e}"
    report(name, "worst gradient max-norm rel err vs the float32 oracle", max(worst.values()))
    for k, v in worst.items():
        report(name, f"grad {k}", v)
    for k, v in worst.items():
        tol = (rtol_by_key or {}).get(k, rtol)
        assert v <= tol, f"{name}: grad {k} rel err {v:.3e} > {tol} ({worst})"
    return worst


TRUTH_FACTOR = 1.25      # HIP may be at most this much farther from the float64 truth than the float32 oracle is (+ 1e-6)


def check_against_truth(name, scene_seen, cam, st, bg, dL, out, pc, m2grad, orc, og, factor=TRUTH_FACTOR, enforce=True):
    """The three-way check that turns "1e-4 between two float32 evaluations is ill-posed for dL/dscaling / dL/drotation" into an
    asserted property: with the float64 build of the oracle (liboracle64.so: the same algorithm, the same float32 inputs, every
    computed quantity in double — checked against the autograd oracle in tests/test_oracle_cpu.py) as the truth, the HIP result
    is no farther from the truth than `factor` x the float32 oracle's own distance + 1e-6, per tensor (max-norm relative, on
    the Gaussians / pixels no oracle flags as borderline), forward and backward.  Returns {tensor: (hip, oracle)} distances
    (enforce=False: only measures)."""
    from oracle import oracle_ctypes as oc
    t = oc.rasterize(scene_seen, cam, st, bg, f64=True)
    tg = oc.backward(t, dL)
    flagged = (orc.borderline_gaussians | t.borderline_gaussians | (t.radii != orc.radii)).cpu()
    clean = ~flagged
    okpx = ~(orc.borderline.bool() | t.borderline.bool())
    e_orc = (orc.color.double() - t.color).abs()[:, okpx].max().item()
    e_hip = (out["render"].detach().cpu().double() - t.color).abs()[:, okpx].max().item()
    report(name, "forward, oracle_f32 vs float64 truth", e_orc)
    report(name, "forward, HIP vs float64 truth", e_hip)
    assert not enforce or e_hip <= factor * e_orc + 1e-6, \
        f"{name}: forward HIP-vs-truth {e_hip:.3e} > {factor} x oracle-vs-truth {e_orc:.3e}"
    hip_t = leaf_space(pc, m2grad, tg)                       # (HIP, truth) per tensor, leaf space
    orc_t = leaf_space(pc, m2grad, og)                       # (HIP, oracle) per tensor: only the oracle side is used
    dist_ = {}
    for k, (got, truth) in hip_t.items():
        d_hip = rel_err(got, truth, clean)
        d_orc = rel_err(orc_t[k][1], truth, clean)
        dist_[k] = (d_hip, d_orc)
        report(name, f"grad {k}: HIP vs truth", d_hip)
        report(name, f"grad {k}: oracle_f32 vs truth", d_orc)
    dist_["forward"] = (e_hip, e_orc)
    for k, (d_hip, d_orc) in dist_.items():
        assert not enforce or d_hip <= factor * d_orc + 1e-6, \
            f"{name}: grad {k}: HIP is {d_hip:.3e} from the float64 truth, the float32 oracle {d_orc:.3e} (allowed {factor}x + 1e-6)"
    return dist_


def small_scene(P, W, H, seed, **kw):
    """frustum scene whose footprints are a few pixels at this (small) resolution"""
    k = kw.pop("scale_k", 0.004 * 1920.0 / W * 0.5)
    return scenes.frustum_scene(P, W, H, seed=seed, scale_k=k, **kw), scenes.front_camera(W, H)
