"""Inputs that are contiguous but only 4-byte aligned (rows carved out of a flat parameter buffer at an odd float offset — a
packed-optimizer layout).  The kernels read rotations, SH rows and covariances with 16-byte global loads; gfx950 serves those from
any 4-byte aligned address (the KFD's unaligned access mode), and the wrapper hands the pointers through as they are — no copy.  This
test pins that assumption.  Property: bit-identical outputs and gradients to the same scene in freshly allocated (512-byte aligned)
tensors, on the reference API (chained getters), on the raw entry and with precomputed colours / covariances.  (Gradient OUTPUTS are
allocated by the wrapper, hence aligned; caller-owned accumulation targets must be 16-byte aligned and are checked:
GradAccumulator.)"""
import pytest
import torch

import scenes
from parity_utils import PIPE, small_scene

pytestmark = pytest.mark.gpu
LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
STATE = ("_occ_multiplier", "_dc_delta", "max_pixel_sizes", "min_pixel_sizes")


def _carve(t, offset_floats):
    """a contiguous view of t's values that starts `offset_floats` floats into a fresh flat buffer"""
    flat = torch.empty(t.numel() + offset_floats + 8, dtype=t.dtype, device=t.device)
    v = flat[offset_floats:offset_floats + t.numel()].view(t.shape)
    v.copy_(t.detach())
    assert v.is_contiguous() and v.data_ptr() % 16 == (4 * offset_floats) % 16
    return v


def _model(sc, dev, offset):
    from synthetic_model import SyntheticGaussians
    m = SyntheticGaussians(sc, dev)
    if offset:
        for k, name in enumerate(LEAVES):
            setattr(m, name, torch.nn.Parameter(_carve(getattr(m, name), offset + (k % 3))))
        for name in STATE:
            setattr(m, name, _carve(getattr(m, name), offset))
    return m


@pytest.mark.parametrize("entry", ["reference", "raw"])
def test_four_byte_aligned_parameters(entry):
    from gaussian_renderer import render, render_fused
    fn = render if entry == "reference" else render_fused
    W, H = 200, 120
    sc, cam = small_scene(5000, W, H, 21, sh_degree=3, multiscale=True, scale_k=0.004 * 1920.0 / W * 0.3)
    st = dict(filter_small=True, filter_large=True, fade_size=0.0)
    dev = torch.device("cuda")
    camd, bg, dL = cam.to(dev), torch.tensor([0.3, 0.1, 0.2], device=dev), scenes.grad_seed(W, H, 21).to(dev)
    res = []
    for offset in (0, 1):
        m = _model(sc, dev, offset)
        out = fn(camd, m, PIPE, bg, **st)
        out["render"].backward(dL)
        torch.cuda.synchronize()
        res.append((out, m))
    (a, ma), (b, mb) = res
    for k in ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["viewspace_points"].grad, b["viewspace_points"].grad)
    for n in LEAVES:
        assert torch.equal(getattr(ma, n).grad, getattr(mb, n).grad), n


def test_four_byte_aligned_precomputed_colours_and_covariances():
    import math
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from synthetic_model import SyntheticGaussians
    W, H = 160, 96
    sc, cam = small_scene(3000, W, H, 22, sh_degree=0)
    dev = torch.device("cuda")
    camd, dL = cam.to(dev), scenes.grad_seed(W, H, 22).to(dev)
    m = SyntheticGaussians(sc, dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                       viewmatrix=camd.world_view_transform, projmatrix=camd.full_proj_transform,
                                       sh_degree=0, campos=camd.camera_center, prefiltered=False, debug=False)
    with torch.no_grad():
        cov, col = m.get_covariance().contiguous(), torch.rand(sc.P, 3, device=dev)
        means, opac = m.get_xyz.detach().clone(), m.get_opacity.detach().clone()
    res = []
    for offset in (0, 1, 3):
        t = lambda x: (_carve(x, offset) if offset else x.clone()).requires_grad_(True)
        ins = dict(means3D=t(means), opacities=t(opac), cov3D_precomp=t(cov), colors_precomp=t(col))
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        img, *_ = GaussianRasterizer(rs)(means2D=m2, **ins)
        img.backward(dL)
        torch.cuda.synchronize()
        res.append((img, [v.grad for v in ins.values()] + [m2.grad]))
    for img, grads in res[1:]:
        assert torch.equal(img, res[0][0])
        for g, g0 in zip(grads, res[0][1]):
            assert torch.equal(g, g0)


def test_strided_inputs_are_rendered_like_their_contiguous_copies():
    """per-Gaussian tensors that are column slices of wider tensors (not contiguous): the wrapper copies, the gradients come back
    through the views into the wide leaves — same bits as with contiguous leaves"""
    import math
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from synthetic_model import SyntheticGaussians
    W, H = 160, 96
    sc, cam = small_scene(3000, W, H, 23, sh_degree=3)
    dev = torch.device("cuda")
    camd, dL = cam.to(dev), scenes.grad_seed(W, H, 23).to(dev)
    m = SyntheticGaussians(sc, dev, requires_grad=False)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                       viewmatrix=camd.world_view_transform, projmatrix=camd.full_proj_transform,
                                       sh_degree=3, campos=camd.camera_center, prefiltered=False, debug=False)
    vals = dict(means3D=m.get_xyz, opacities=m.get_opacity, scales=m.get_scaling, rotations=m.get_rotation, shs=m.get_features)
    res = []
    for strided in (False, True):
        leaves, ins = {}, {}
        for k, v in vals.items():
            if strided:
                flat = v.reshape(sc.P, -1)
                wide = torch.zeros(sc.P, flat.shape[1] + 3, device=dev)
                wide[:, 2:2 + flat.shape[1]] = flat
                wide.requires_grad_(True)
                leaves[k] = wide
                ins[k] = wide[:, 2:2 + flat.shape[1]].view(v.shape) if v.dim() == 2 else \
                    wide[:, 2:2 + flat.shape[1]].unflatten(1, v.shape[1:])
                assert not ins[k].is_contiguous() or ins[k].shape[1] == 1
            else:
                leaves[k] = ins[k] = v.clone().requires_grad_(True)
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        img, *_ = GaussianRasterizer(rs)(means2D=m2, **ins)
        img.backward(dL)
        torch.cuda.synchronize()
        grads = {k: (leaves[k].grad[:, 2:2 + vals[k].reshape(sc.P, -1).shape[1]].reshape(vals[k].shape) if strided
                     else leaves[k].grad) for k in vals}
        res.append((img, grads, m2.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    for k in vals:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
