"""GPU voxel-average pooling for MS-GS' large-Gaussian insertion.

The reference pools eleven per-Gaussian tensors with `open3d.ml.torch.layers.VoxelPooling(position_fn='center',
feature_fn='average')` on the CPU, with a .cpu() / .cuda() round trip around every call
(/root/reference/scene/gaussian_model.py:789-848, called from train.py:334).  Here the voxel grouping is built once on
the GPU by libmsgs_hip.so (msgs_voxel_pool_build: 64-bit voxel keys, two stable 32-bit radix stages, segment scan)
and every tensor is reduced with a deterministic segmented mean (msgs_voxel_pool_average).

  VoxelGrouping(positions, voxel_size)       build once; .average(features), .centers, .mean_positions(), .counts
  VoxelPooling(position_fn, feature_fn)      call-compatible stand-in for open3d.ml.torch.layers.VoxelPooling
                                             (result has .pooled_positions / .pooled_features)
  pool_large_gaussians(...)                  the attribute computation of GaussianModel.insert_large_gaussians

open3d is third-party and not vendored by the reference; its documented behaviour is restated (voxel index =
floor(p / voxel_size) per axis; 'average' = arithmetic mean; 'center' = voxel centre).  The ORDER of the pooled rows is
unspecified in open3d; here it is ascending (z, y, x) voxel index.  All eleven tensors share one grouping, so rows
stay aligned exactly as the reference requires.
"""
import ctypes as C
from types import SimpleNamespace

import torch

from diff_gaussian_rasterization import _backend as _C


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class VoxelGrouping:
    def __init__(self, positions: torch.Tensor, voxel_size: float):
        if positions.device.type != "cuda":
            positions = positions.cuda()
        pos = positions.detach().to(torch.float32).contiguous().view(-1, 3)
        self.device = pos.device
        self.voxel_size = float(voxel_size)
        self.M = int(pos.shape[0])
        lib = _C.lib
        M = self.M
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self.order = torch.empty(max(M, 1), dtype=torch.int32, device=self.device)
            seg = torch.empty(max(M, 1) + 1, dtype=torch.int32, device=self.device)
            vidx = torch.empty(max(M, 1), 3, dtype=torch.int32, device=self.device)
            scratch = torch.empty(max(int(lib.msgs_voxel_pool_scratch_bytes(M)), 1), dtype=torch.uint8, device=self.device)
            nv = C.c_int64(0)
            _C.check(lib.msgs_voxel_pool_build(_ptr(pos), M, self.voxel_size, _ptr(self.order), _ptr(seg), _ptr(vidx),
                                               _ptr(scratch), scratch.numel(), C.byref(nv), stream),
                     "msgs_voxel_pool_build")
        self.num_voxels = int(nv.value)
        self.seg_start = seg[: self.num_voxels + 1]
        self.voxel_index = vidx[: self.num_voxels]
        self._pos = pos

    @property
    def counts(self):
        return (self.seg_start[1:] - self.seg_start[:-1]).to(torch.int64)

    @property
    def centers(self):
        return (self.voxel_index.to(torch.float32) + 0.5) * self.voxel_size

    def average(self, features: torch.Tensor) -> torch.Tensor:
        """features [M, ...] -> [num_voxels, prod(...)] float32 means (row v = voxel v)"""
        if features.device != self.device:
            features = features.to(self.device)
        F = 1
        for d in features.shape[1:]:
            F *= int(d)
        f = features.detach().to(torch.float32).contiguous().view(self.M, F)
        out = torch.empty(self.num_voxels, F, dtype=torch.float32, device=self.device)
        if self.num_voxels and F:
            with torch.cuda.device(self.device):
                stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                _C.check(_C.lib.msgs_voxel_pool_average(_ptr(f), F, _ptr(self.order), _ptr(self.seg_start),
                                                        self.num_voxels, _ptr(out), stream), "msgs_voxel_pool_average")
        return out

    def mean_positions(self):
        return self.average(self._pos)


class VoxelPooling:
    """Stand-in for open3d.ml.torch.layers.VoxelPooling as the reference uses it
    (gaussian_model.py:802: position_fn='center', feature_fn='average'); also accepts position_fn='average'."""

    def __init__(self, position_fn="center", feature_fn="average"):
        if position_fn not in ("center", "average") or feature_fn != "average":
            raise NotImplementedError(f"VoxelPooling(position_fn={position_fn!r}, feature_fn={feature_fn!r}): only "
                                      "position_fn in ('center', 'average') with feature_fn='average' is provided")
        self.position_fn = position_fn
        self._cache = None      # (key, grouping): the reference calls with the same positions eleven times

    def __call__(self, positions, features, voxel_size):
        key = (positions.data_ptr(), tuple(positions.shape), positions._version, float(voxel_size), str(positions.device))
        if self._cache is None or self._cache[0] != key:
            self._cache = (key, VoxelGrouping(positions, voxel_size))
        g = self._cache[1]
        pooled_pos = g.centers if self.position_fn == "center" else g.mean_positions()
        return SimpleNamespace(pooled_positions=pooled_pos, pooled_features=g.average(features))


def pool_large_gaussians(xyz, features_dc, features_rest, opacity, occ_multiplier, dc_delta, rotation, scaling,
                         max_pixel_sizes, min_pixel_sizes, mask, cur_min_pixel_sizes, reso_lvl, scene_extent):
    """The attribute computation of GaussianModel.insert_large_gaussians (gaussian_model.py:789-848) on RAW parameters
    (log-scale, logit-opacity): returns the dict of new tensors the reference hands to densification_postfix."""
    rel_pos = xyz[mask] / scene_extent
    rel_pos = torch.where(rel_pos > 1, 2 - 1 / rel_pos, rel_pos)            # :793-794 (positive side only, as there)
    voxel_reso = 0.02 * (reso_lvl / 4)                                       # :800
    g = VoxelGrouping(rel_pos, voxel_reso)
    N = xyz.shape[0]
    pool = lambda t: g.average(t.reshape(N, -1)[mask])
    M = g.num_voxels
    out = dict(
        xyz=pool(xyz).reshape(M, *xyz.shape[1:]),
        features_dc=pool(features_dc).reshape(M, *features_dc.shape[1:]),
        features_rest=pool(features_rest).reshape(M, *features_rest.shape[1:]),
        opacity=pool(opacity).reshape(M, *opacity.shape[1:]),
        occ_multiplier=pool(occ_multiplier).reshape(M, *occ_multiplier.shape[1:]),
        dc_delta=pool(dc_delta).reshape(M, *dc_delta.shape[1:]),
        rotation=pool(rotation).reshape(M, *rotation.shape[1:]),
    )
    voxel_scaling = pool(scaling).reshape(M, *scaling.shape[1:])
    cur = torch.clip(pool(cur_min_pixel_sizes.float()).reshape(M, 1), min=0.25, max=2.0)      # :834
    out["scaling"] = torch.log(torch.exp(voxel_scaling) * (2.0 / cur))                          # :835-836
    out["opacity"] = torch.log(torch.sigmoid(out["opacity"]) / (1 - torch.sigmoid(out["opacity"])))   # :837 (identity round trip)
    out["max_pixel_sizes"] = -torch.ones(M, device=xyz.device)                                  # :840-841
    out["min_pixel_sizes"] = -torch.ones(M, device=xyz.device)
    out["target_reso_lvl"] = torch.full((M,), float(reso_lvl), device=xyz.device)
    out["grouping"] = g
    return out
