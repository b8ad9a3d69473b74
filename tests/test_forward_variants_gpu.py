"""The list-walking blend-forward kernel variants (msgs_set_forward_variant: 1 quadrant lists, 3 strip lists with the y-extent
strip test, 4 strip lists with the exact strip test) evaluate every pixel with the same instructions in the same order on a
superset of the entries that can contribute: images, per-pixel state (hence every gradient) and per-Gaussian outputs are
bit-identical across them — on ragged small images, on scenes with huge and sub-pixel footprints, and at C3 size.  (Variant 2,
the round-1 one-wave-per-tile forward kept for A/B runs, rounds its accumulation differently: compared at 2e-6.)"""
import pytest
import torch

import scenes
from parity_utils import hip_render, small_scene

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def _all_variants(sc, cam, st, bg, dL):
    import diff_gaussian_rasterization as dgr
    lib = dgr._C.lib
    prev_v = lib.msgs_set_forward_variant(1)
    prev_g = lib.msgs_set_blend_granularity(1)       # coarse kernels at every size, so the variants are really exercised
    res = {}
    try:
        for v in (1, 2, 3, 4, 5, 6):
            lib.msgs_set_forward_variant(v)
            res[v] = hip_render(sc, cam, st, bg, dL)
    finally:
        lib.msgs_set_forward_variant(prev_v)
        lib.msgs_set_blend_granularity(prev_g)
    return res


def _assert_identical(res):
    a, pa, ma = res[1]
    assert a["render"].abs().max().item() > 0
    b2 = res[2][0]
    assert (a["render"] - b2["render"]).abs().max().item() <= 2e-6 and torch.equal(a["radii"], b2["radii"])
    for v in (3, 4):
        b, pb, mb = res[v]
        for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
            assert torch.equal(a[k], b[k]), (v, k)
        # same final_T / last contributor per pixel -> the backward sees identical inputs: identical gradients (the default
        # backward is reproducible to the bit)
        for n in LEAVES:
            assert torch.equal(getattr(pa, n).grad, getattr(pb, n).grad), (v, n)
        assert torch.equal(ma, mb), v


@pytest.mark.parametrize("P,W,H,seed,k", [(6000, 203, 117, 57, 0.2), (3000, 64, 48, 58, 1.5), (20000, 320, 200, 59, 0.05),
                                           (500, 17, 9, 60, 0.5)])
def test_forward_variants_bit_identical_small(P, W, H, seed, k):
    sc, cam = small_scene(P, W, H, seed, multiscale=True, scale_k=0.004 * 1920.0 / W * k)
    st = dict(filter_small=True, filter_large=True, fade_size=0.0)
    _assert_identical(_all_variants(sc, cam, st, torch.tensor([0.2, 0.5, 0.1]), scenes.grad_seed(W, H, seed)))


def test_forward_variants_bit_identical_c3():
    sc, cam, st = scenes.config("C3")
    _assert_identical(_all_variants(sc, cam, st, torch.zeros(3), scenes.grad_seed(cam.image_width, cam.image_height, 2)))
