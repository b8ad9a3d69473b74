"""CPU restatement of simple-knn's distCUDA2: exact 3-nearest-neighbour mean squared distance.

TEST INFRASTRUCTURE ONLY — imported by tests/; ms-gs_amd/ never imports this.

PARITY UNPINNED: simple-knn is an un-vendored CUDA submodule of the reference (/root/reference/.gitmodules,
`submodules/simple-knn`), absent here; its published behaviour — exact 3-NN among all OTHER points (excluded by
index, so duplicates count with distance 0), squared Euclidean distance, arithmetic mean of the three — is restated
with scipy's cKDTree.  Call site: /root/reference/scene/gaussian_model.py:199.
"""
import numpy as np
from scipy.spatial import cKDTree


def mean_dist2_knn3(points):
    pts = np.asarray(points, dtype=np.float32)
    P = len(pts)
    tree = cKDTree(pts.astype(np.float64))
    k = min(P, 8)
    _, idx = tree.query(pts.astype(np.float64), k=k)           # self (or a duplicate) comes first
    out = np.empty(P, dtype=np.float32)
    for i in range(P):
        nb = [j for j in idx[i] if j != i][:3]
        if len(nb) < 3 or (k == 8 and P > 8 and i not in idx[i]):
            # more than 7 duplicates of this point: fall back to brute force for it
            d = ((pts.astype(np.float64) - pts[i].astype(np.float64)) ** 2).sum(1)
            d[i] = np.inf
            nb = np.argsort(d, kind="stable")[:3]
        diff = pts[nb] - pts[i]                                  # float32 arithmetic like the kernel
        d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        d2 = np.sort(d2)
        out[i] = (d2[0] + d2[1] + d2[2]) / np.float32(3.0)
    return out
