"""GPU voxel-average pooling (large-Gaussian insertion) against the numpy oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(pos, feats, vs):
    from voxel_pool import VoxelGrouping
    from oracle.voxel_pool_oracle import voxel_average_pool
    g = VoxelGrouping(pos.cuda(), vs)
    ref = voxel_average_pool(pos.numpy(), feats.numpy(), vs)
    assert g.num_voxels == ref["features"].shape[0]
    assert np.array_equal(g.voxel_index.cpu().numpy(), ref["voxel_index"])                 # same voxels, same order
    assert np.array_equal(g.counts.cpu().numpy(), ref["counts"])
    got = g.average(feats.cuda()).cpu().double().numpy()
    scale = max(np.abs(ref["features"]).max(), 1e-30)
    assert np.abs(got - ref["features"]).max() <= 2e-6 * scale
    assert np.allclose(g.centers.cpu().numpy(), ref["centers"], rtol=0, atol=1e-6 * max(1.0, abs(vs) * 2 ** 20 * 1e-3))
    # order[] is a permutation grouped by voxel with ascending point ids inside every voxel
    order = g.order[: g.M].cpu().long()
    assert torch.equal(torch.sort(order).values, torch.arange(g.M))
    inv = torch.from_numpy(ref["inverse"])
    seg = g.seg_start.cpu().long()
    vid_of_sorted = torch.repeat_interleave(torch.arange(g.num_voxels), seg[1:] - seg[:-1])
    assert torch.equal(inv[order], vid_of_sorted)
    same = vid_of_sorted[1:] == vid_of_sorted[:-1]
    assert bool((order[1:][same] > order[:-1][same]).all())
    return g


@pytest.mark.parametrize("M,F,vs,seed", [(1, 3, 0.5, 0), (257, 1, 0.1, 1), (5000, 7, 0.02, 2), (200_000, 45, 0.01, 3),
                                          (1_000_000, 3, 0.005, 4)])
def test_voxel_pool_matches_oracle(M, F, vs, seed):
    g = torch.Generator().manual_seed(seed)
    pos = (torch.rand(M, 3, generator=g) * 4 - 2)                # the reference maps positions into (-2, 2)
    pos[: M // 10] = pos[: M // 10].round(decimals=1)            # many coincident points -> multi-point voxels
    feats = torch.randn(M, F, generator=g)
    gr = _check(pos, feats, vs)
    assert gr.num_voxels <= M


def test_voxel_pool_edge_cases():
    from voxel_pool import VoxelGrouping, VoxelPooling
    g = VoxelGrouping(torch.zeros(0, 3).cuda(), 0.1)
    assert g.num_voxels == 0 and g.average(torch.zeros(0, 5).cuda()).shape == (0, 5)
    # all points in one voxel; negative coordinates floor towards -inf
    pos = torch.tensor([[-0.01, -0.01, -0.01], [-0.09, -0.02, -0.05], [0.01, 0.01, 0.01]])
    gr = VoxelGrouping(pos.cuda(), 0.1)
    assert gr.num_voxels == 2 and gr.voxel_index.cpu().tolist() == [[-1, -1, -1], [0, 0, 0]]
    assert torch.allclose(gr.average(torch.tensor([[1.0], [3.0], [10.0]]).cuda()).cpu(), torch.tensor([[2.0], [10.0]]))
    # open3d-compatible shim, CPU tensors in (as the reference passes them), GPU tensors out; grouping cached
    vp = VoxelPooling(position_fn="center", feature_fn="average")
    r1 = vp(pos, torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]), 0.1)
    r2 = vp(pos, torch.ones(3, 4), 0.1)
    assert r1.pooled_features.is_cuda and r1.pooled_features.cpu().tolist() == [[2.0, 3.0], [5.0, 6.0]]
    assert r2.pooled_features.shape == (2, 4)
    assert torch.allclose(r1.pooled_positions.cpu(), torch.tensor([[-0.05, -0.05, -0.05], [0.05, 0.05, 0.05]]))
    with pytest.raises(NotImplementedError):
        VoxelPooling(position_fn="center", feature_fn="max")


def test_pool_large_gaussians_mirrors_reference_formulas():
    """gaussian_model.py:789-848 on a synthetic model: rows aligned across tensors, scaling enlarged by
    2 / clip(min_pixel_size, 0.25, 2), sentinels for the pixel sizes, level tag."""
    from voxel_pool import pool_large_gaussians
    from oracle.voxel_pool_oracle import voxel_average_pool
    g = torch.Generator().manual_seed(9)
    N = 20_000
    xyz = torch.randn(N, 3, generator=g) * 3
    t = lambda *s: torch.randn(N, *s, generator=g)
    fdc, frest, opac, occ, dcd, rot, scal = t(1, 3), t(15, 3), t(1), torch.ones(N, 4, 1), torch.zeros(N, 12, 1), t(4), t(3) * 0.3 - 3
    mask = torch.rand(N, generator=g) < 0.4
    cur = torch.rand(N, generator=g) * 3
    dev = lambda x: x.cuda()
    out = pool_large_gaussians(dev(xyz), dev(fdc), dev(frest), dev(opac), dev(occ), dev(dcd), dev(rot), dev(scal),
                               -torch.ones(N).cuda(), -torch.ones(N).cuda(), dev(mask), dev(cur), reso_lvl=2,
                               scene_extent=4.0)
    rel = xyz[mask] / 4.0
    rel = torch.where(rel > 1, 2 - 1 / rel, rel)
    ref = voxel_average_pool(rel.numpy(), xyz[mask].numpy(), 0.02 * (2 / 4))
    M = ref["features"].shape[0]
    assert out["xyz"].shape == (M, 3) and out["features_rest"].shape == (M, 15, 3) and out["rotation"].shape == (M, 4)
    assert np.abs(out["xyz"].cpu().double().numpy() - ref["features"]).max() < 1e-5
    ref_s = voxel_average_pool(rel.numpy(), scal[mask].numpy(), 0.01)["features"]
    ref_c = np.clip(voxel_average_pool(rel.numpy(), cur[mask].numpy()[:, None], 0.01)["features"], 0.25, 2.0)
    want = np.log(np.exp(ref_s) * (2.0 / ref_c))
    assert np.abs(out["scaling"].cpu().double().numpy() - want).max() < 1e-4
    assert (out["max_pixel_sizes"] == -1).all() and (out["min_pixel_sizes"] == -1).all() and (out["target_reso_lvl"] == 2).all()
    assert torch.allclose(out["occ_multiplier"].cpu(), torch.ones(M, 4, 1)) and torch.allclose(out["dc_delta"].cpu(), torch.zeros(M, 12, 1))
