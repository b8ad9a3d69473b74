"""numpy restatement of voxel-average pooling (test infrastructure only; never imported by ms-gs_amd/).

PARITY UNPINNED: the reference delegates to open3d.ml.torch.layers.VoxelPooling (third-party, not vendored, not
installed here, no version pinned in /root/reference/environment.yml); its published behaviour is restated:
voxel index = floor(p / voxel_size) per axis, pooled feature = arithmetic mean of the voxel's points, pooled position
= voxel centre ('center') or mean position ('average').  Call site: /root/reference/scene/gaussian_model.py:802-816.
Output rows are ordered by ascending (z, y, x) voxel index (open3d leaves the order unspecified)."""
import numpy as np


def voxel_average_pool(positions, features, voxel_size):
    pos = np.asarray(positions, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(voxel_size)
    idx = np.floor(pos * inv).astype(np.int64)                    # float32 product, like the kernel
    idx = np.clip(idx, -(1 << 20), (1 << 20) - 1)
    key = ((idx[:, 2] + (1 << 20)) << 42) | ((idx[:, 1] + (1 << 20)) << 21) | (idx[:, 0] + (1 << 20))
    uniq, inverse, counts = np.unique(key, return_inverse=True, return_counts=True)
    feats = np.asarray(features, dtype=np.float64).reshape(pos.shape[0], -1)
    sums = np.zeros((uniq.shape[0], feats.shape[1]), dtype=np.float64)
    np.add.at(sums, inverse, feats)
    means = sums / counts[:, None]
    vidx = np.stack([(uniq & 0x1FFFFF) - (1 << 20), ((uniq >> 21) & 0x1FFFFF) - (1 << 20),
                     ((uniq >> 42) & 0x1FFFFF) - (1 << 20)], axis=1)
    centers = (vidx.astype(np.float64) + 0.5) * float(voxel_size)
    return dict(features=means, counts=counts, voxel_index=vidx, centers=centers, inverse=inverse)
