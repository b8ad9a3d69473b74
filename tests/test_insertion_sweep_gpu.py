"""msgs_preprocess_only and the insertion sweeps built on it, against the full renders the reference performs."""
import copy
import math

import pytest
import torch

import scenes
from parity_utils import PIPE, small_scene

pytestmark = pytest.mark.gpu
ST = dict(filter_small=True, filter_large=True, fade_size=0.0)


def _model(sc):
    from synthetic_model import SyntheticGaussians
    pc = SyntheticGaussians(sc, "cuda", requires_grad=False)
    pc.training_setup(7, sc.target_reso_lvl)
    return pc


def _half(cam, k):
    c = copy.copy(cam)
    c.image_width, c.image_height = int(cam.image_width / 2 ** k), int(cam.image_height / 2 ** k)
    return c


@pytest.mark.parametrize("ms", [False, True])
def test_preprocess_only_equals_the_render_outputs(ms):
    from gaussian_renderer import render
    from insertion_sweep import view_visibility
    W, H = 200, 120
    sc, cam = small_scene(7000, W, H, 61, multiscale=ms, **({"scale_k": 0.004 * 1920.0 / W * 0.2} if ms else {}))
    pc = _model(sc)
    st = ST if ms else dict(filter_small=False, filter_large=False, fade_size=1.0)
    bg = torch.zeros(3).cuda()
    for k in (0, 2):
        c = _half(cam, k).to("cuda")
        with torch.no_grad():
            out = render(c, pc, PIPE, bg, **st)
        vis, ps, radii = view_visibility(c, pc, PIPE, bg, **st)
        assert torch.equal(radii, out["radii"]) and torch.equal(vis, out["visibility_filter"])
        assert torch.equal(ps, out["pixel_sizes"])                       # same kernel: bit-equal


def test_insertion_sweep_matches_the_render_based_sequence():
    from gaussian_renderer import render
    from insertion_sweep import refresh_pixel_sizes, select_insertion_sources
    W, H, V = 240, 136, 4
    sc = scenes.ball_scene(20000, seed=8, log_s=math.log(0.012))
    g = torch.Generator().manual_seed(8)
    sc.target_reso_lvl = torch.where(torch.rand(sc.P, generator=g) < 0.2, 2, 0)
    sc.min_pixel_sizes = torch.where(torch.rand(sc.P, generator=g) < 0.5, -torch.ones(sc.P), 0.5 + torch.rand(sc.P, generator=g))
    sc.max_pixel_sizes = torch.where(sc.target_reso_lvl > 0, 2 + 6 * torch.rand(sc.P, generator=g), -torch.ones(sc.P))
    cams = [scenes.ring_camera(v, V, W, H) for v in range(V)]
    base = [c.to("cuda") for c in cams]
    nxt = [_half(c, 2).to("cuda") for c in cams]
    bg = torch.zeros(3).cuda()
    pc, ref = _model(sc), _model(sc)

    sel, min_ps = select_insertion_sources(base, nxt, pc, PIPE, bg, **ST)
    ref_min = torch.ones_like(ref.min_pixel_sizes)                 # train.py:283-315 with full renders
    with torch.no_grad():
        for cb, cn in zip(base, nxt):
            bv = render(cb, ref, PIPE, bg, **ST)["visibility_filter"]
            o = render(cn, ref, PIPE, bg, **ST)
            ref_min = torch.where(torch.logical_and(o["pixel_sizes"] > 0, bv), torch.minimum(o["pixel_sizes"], ref_min), ref_min)
    ref_sel = torch.logical_and(ref_min < 1, ref.target_reso_lvl == 0)
    assert torch.equal(min_ps, ref_min) and torch.equal(sel, ref_sel)
    assert 0 < sel.sum().item() < sel.numel()
    # the same sweep with the base / next launches of every camera pair on two streams
    from multi_view import ViewPipeline
    sel2, min_ps2 = select_insertion_sources(base, nxt, pc, PIPE, bg, lanes=ViewPipeline("cuda"), **ST)
    assert torch.equal(min_ps2, ref_min) and torch.equal(sel2, ref_sel)

    refresh_pixel_sizes(nxt, pc, 2, PIPE, bg, **ST)
    with torch.no_grad():                                          # train.py:334-338 with full renders
        for cn in nxt:
            o = render(cn, ref, PIPE, bg, **ST)
            ps = o["pixel_sizes"]
            mask = o["visibility_filter"] & (ref.target_reso_lvl == 2)
            ref.max_pixel_sizes[mask] = torch.max(ref.max_pixel_sizes[mask] * 0.95, ps[mask])
            mn = torch.clip(ref.min_pixel_sizes[mask] * 1.05, -1)
            ref.min_pixel_sizes[mask] = torch.where(ps[mask] > 0, torch.where(mn < 0, ps[mask], torch.min(mn, ps[mask])), mn)
    assert torch.equal(pc.max_pixel_sizes, ref.max_pixel_sizes) and torch.equal(pc.min_pixel_sizes, ref.min_pixel_sizes)
    assert not torch.equal(pc.max_pixel_sizes, sc.max_pixel_sizes.cuda())
