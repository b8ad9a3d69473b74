"""distCUDA2 (msgs_dist2_knn3) against the exact 3-NN oracle; create_from_points on top of it."""
import types

import numpy as np
import pytest
import torch

from oracle import knn_oracle

pytestmark = pytest.mark.gpu


def _check(pts, rtol=2e-6):
    from simple_knn._C import distCUDA2
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    ref = knn_oracle.mean_dist2_knn3(pts)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=1e-30)
    return got


def test_uniform_and_tiny():
    rng = np.random.default_rng(1)
    _check(rng.random((4, 3), dtype=np.float32))
    _check(rng.random((65, 3), dtype=np.float32))
    _check(rng.random((5000, 3), dtype=np.float32) * 10 - 5)


def test_clustered_with_outliers_like_a_colmap_cloud():
    rng = np.random.default_rng(2)
    centres = rng.normal(size=(40, 3)) * 5
    pts = np.concatenate([c + rng.normal(size=(500, 3)) * rng.uniform(0.01, 0.5) for c in centres] +
                         [rng.normal(size=(300, 3)) * 200]).astype(np.float32)
    rng.shuffle(pts)
    _check(pts)


def test_degenerate_sets():
    rng = np.random.default_rng(3)
    line = np.zeros((3000, 3), np.float32)
    line[:, 0] = rng.random(3000, dtype=np.float32)                    # collinear: two Morton axes collapse
    _check(line)
    dup = rng.random((2000, 3), dtype=np.float32)
    dup[100:140] = dup[100]                                            # 40 coincident points: distance 0 neighbours
    got = _check(dup)
    assert (got[100:140] == 0).all()
    plane = rng.random((4000, 3), dtype=np.float32)
    plane[:, 2] = 7.0
    _check(plane)
    same = np.ones((100, 3), np.float32)                               # zero-extent bounding box
    assert (_check(same) == 0).all()


def test_large_cloud_and_argument_errors():
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(4)
    pts = (rng.normal(size=(300_000, 3)) * np.array([3.0, 1.0, 0.3])).astype(np.float32)
    _check(pts)
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(3, 3).cuda())                            # fewer than 4 points
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(10, 2).cuda())
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 3))


def test_create_from_points_and_ply_round_trip_on_gpu(tmp_path):
    import model_io
    rng = np.random.default_rng(5)
    pts, rgb = rng.normal(size=(3000, 3)).astype(np.float32), rng.random((3000, 3), dtype=np.float32)
    m = model_io.create_from_points(types.SimpleNamespace(), pts, rgb)
    d2 = np.maximum(knn_oracle.mean_dist2_knn3(pts), 1e-7)
    np.testing.assert_allclose(m._scaling.detach().cpu().numpy(), np.repeat(np.log(np.sqrt(d2))[:, None], 3, 1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(torch.sigmoid(m._opacity), torch.full_like(m._opacity, 0.1), atol=1e-6)
    assert torch.allclose(m._features_dc[:, 0, :] * 0.28209479177387814 + 0.5, torch.from_numpy(rgb).cuda(), atol=1e-6)
    path = str(tmp_path / "pc.ply")
    model_io.save_ply(m, path)
    back = model_io.load_ply(types.SimpleNamespace(), path)
    assert back._xyz.is_cuda and torch.equal(back._scaling.detach(), m._scaling.detach())
    assert torch.equal(back._features_rest.detach(), m._features_rest.detach())
