"""Host-side model I/O (PLY layout, checkpoint tuple) and the kNN oracle — no GPU needed."""
import types

import numpy as np
import pytest
import torch

import model_io
from oracle import knn_oracle


def _model(P, seed=0, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    m = types.SimpleNamespace(active_sh_degree=2, max_sh_degree=3)
    m._xyz, m._features_dc, m._features_rest = r(P, 3), r(P, 1, 3), r(P, 15, 3)
    m._opacity, m._scaling, m._rotation = r(P, 1), r(P, 3), r(P, 4)
    m._occ_multiplier, m._dc_delta = torch.ones(P, 4, 1) + 0.1 * r(P, 4, 1), 0.1 * r(P, 12, 1)
    m.base_gaussian_mask = torch.rand(P, generator=g) < 0.3
    m.max_pixel_sizes, m.min_pixel_sizes = r(P).abs() * 4, -torch.ones(P)
    m.max_radii2D = r(P).abs()
    m.xyz_gradient_accum, m.denom = r(P, 3, 1).abs(), torch.ones(P, 3, 1)
    m.target_reso_lvl = torch.randint(0, 3, (P,), generator=g)
    return m


def test_ply_layout_and_round_trip(tmp_path):
    m = _model(37)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    model_io.save_ply(m, path)
    raw = open(path, "rb").read()
    header = raw[:raw.index(b"end_header\n")].decode().split("\n")
    assert header[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    props = [h.split() for h in header[3:] if h]
    names = [p[2] for p in props]
    assert names == model_io.attribute_names() and len(names) == 81
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[54] == "opacity" and names[55] == "occ_multiplier_0"
    assert names[59:62] == ["dc_delta_0_0", "dc_delta_0_1", "dc_delta_0_2"] and names[-3:] == ["base_gaussian_mask", "max_pixel_sizes", "min_pixel_sizes"]
    assert {p[2]: p[1] for p in props}["base_gaussian_mask"] == "uchar"
    assert all(p[1] == "float" for p in props if p[2] != "base_gaussian_mask")
    assert len(raw) == raw.index(b"end_header\n") + len(b"end_header\n") + 37 * (80 * 4 + 1)
    v = model_io.read_ply(path)
    # SH stored channel-major: f_rest_j = features_rest[:, j % 15, j // 15]
    np.testing.assert_array_equal(v["f_rest_16"], m._features_rest[:, 1, 1].numpy())
    np.testing.assert_array_equal(v["f_dc_2"], m._features_dc[:, 0, 2].numpy())
    np.testing.assert_array_equal(v["nx"], np.zeros(37, np.float32))
    back = model_io.load_ply(types.SimpleNamespace(), path, device="cpu")
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_occ_multiplier", "_dc_delta", "_scaling", "_rotation",
              "base_gaussian_mask", "max_pixel_sizes", "min_pixel_sizes"):
        assert torch.equal(getattr(back, k).detach(), getattr(m, k)), k
    assert back._xyz.requires_grad and not back._occ_multiplier.requires_grad and back.active_sh_degree == 3
    assert back._features_rest.shape == (37, 15, 3) and back._features_rest.is_contiguous()


def test_read_ply_accepts_ascii_and_rejects_garbage(tmp_path):
    p = tmp_path / "a.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty uchar m\n"
                 "end_header\n1.5 1\n-2 0\n")
    v = model_io.read_ply(str(p))
    assert v["x"].tolist() == [1.5, -2.0] and v["m"].tolist() == [1, 0]
    q = tmp_path / "b.ply"
    q.write_bytes(b"not a ply\n")
    with pytest.raises(ValueError):
        model_io.read_ply(str(q))
    with pytest.raises(ValueError, match="f_rest"):
        model_io.load_ply(types.SimpleNamespace(), _write_short(tmp_path), device="cpu")


def _write_short(tmp_path):
    m = _model(3)
    m._features_rest = m._features_rest[:, :8].contiguous()            # SH degree 2 file read as degree 3
    path = str(tmp_path / "short.ply")
    model_io.save_ply(m, path)
    return path


def test_checkpoint_tuple_round_trip_in_capture_order():
    m = _model(11, seed=3)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        setattr(m, k, torch.nn.Parameter(getattr(m, k)))
    mk = lambda mod: torch.optim.Adam([{"params": [mod._xyz], "lr": 1e-3, "name": "xyz"},
                                       {"params": [mod._opacity], "lr": 5e-2, "name": "opacity"}], lr=0.0, eps=1e-15)
    opt = mk(m)
    m._xyz.grad, m._opacity.grad = torch.ones_like(m._xyz), torch.ones_like(m._opacity)
    opt.step()
    tup = model_io.capture(m, opt, spatial_lr_scale=2.5)
    assert len(tup) == 18 and tup[0] == 2 and tup[9] is m.max_radii2D and tup[10] is m.base_gaussian_mask
    assert tup[11] is m.max_pixel_sizes and tup[12] is m.min_pixel_sizes and tup[15] is m.target_reso_lvl and tup[17] == 2.5
    fresh = types.SimpleNamespace(max_sh_degree=3)
    opt2, scale = model_io.restore(fresh, tup, mk)
    assert scale == 2.5 and fresh.active_sh_degree == 2                 # restored verbatim (gaussian_model.py:101-125)
    assert fresh.base_gaussian_mask is m.base_gaussian_mask and fresh.min_pixel_sizes is m.min_pixel_sizes
    assert fresh.max_pixel_sizes is m.max_pixel_sizes and fresh.denom is m.denom
    st = opt2.state[fresh._xyz]
    assert torch.equal(st["exp_avg"], opt.state[m._xyz]["exp_avg"]) and float(st["step"]) == 1
    with pytest.raises(ValueError):
        model_io.restore(fresh, tup[:-1], mk)


def test_knn_oracle_against_brute_force():
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(300, 3)).astype(np.float32)
    pts[10] = pts[11] = pts[12]                                         # duplicates count as neighbours at distance 0
    d = ((pts[:, None, :].astype(np.float64) - pts[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    ref = np.sort(d, axis=1)[:, :3].mean(1)
    got = knn_oracle.mean_dist2_knn3(pts)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-12)
    assert got[10] == np.float32(np.sort(d[10])[2] / 3)            # two of its three neighbours coincide with it
