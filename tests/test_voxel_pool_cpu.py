"""CPU checks of the voxel-pool oracle itself (brute force on a tiny case) and of the ABI declarations."""
import numpy as np

from oracle.voxel_pool_oracle import voxel_average_pool


def test_oracle_against_brute_force():
    rng = np.random.RandomState(0)
    pos = (rng.rand(300, 3) * 2 - 1).astype(np.float32)
    feats = rng.randn(300, 4)
    vs = 0.25
    r = voxel_average_pool(pos, feats, vs)
    cells = {}
    for p, f in zip(pos, feats):
        k = tuple(np.floor(p * np.float32(1.0 / vs)).astype(int)[::-1])       # (z, y, x) for lexicographic order
        cells.setdefault(k, []).append(f)
    keys = sorted(cells)
    assert len(keys) == r["features"].shape[0]
    for row, k in enumerate(keys):
        assert tuple(r["voxel_index"][row][::-1]) == k
        assert np.allclose(r["features"][row], np.mean(cells[k], axis=0))
        assert r["counts"][row] == len(cells[k])


def test_voxel_pool_symbols_exported():
    import diff_gaussian_rasterization as dgr
    for n in ("msgs_voxel_pool_scratch_bytes", "msgs_voxel_pool_build", "msgs_voxel_pool_average"):
        assert hasattr(dgr._C.lib, n)
    assert dgr._C.lib.msgs_voxel_pool_scratch_bytes(1_000_000) >= 32 * 1_000_000
