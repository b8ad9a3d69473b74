"""Camera sweeps around MS-GS' large-Gaussian insertion (second half), without rendering.

The reference renders every training camera twice — at the base and at the next resolution — only to read
`visibility_filter` and `pixel_sizes` (/root/reference/train.py:283-300), and once more after the insertion to refresh
the pixel sizes (:334-338).  Both quantities are outputs of the per-Gaussian kernel, so here each camera costs one
msgs_preprocess_only launch per resolution (~0.1 ms at 1 M Gaussians) instead of a full render.

  view_visibility(cam, pc, pipe, ...)                        -> (visibility_filter, pixel_sizes) of one camera
  select_insertion_sources(base_cams, next_cams, pc, ...)    -> (all_diff_vis_filter, min_pixel_sizes)  train.py:283-315
  refresh_pixel_sizes(next_cams, pc, next_reso_idx, ...)     train.py:334-338
"""
import torch

from diff_gaussian_rasterization import GaussianRasterizer
from gaussian_renderer import _settings
from train_epilogue import update_training_stats


def view_visibility(cam, pc, pipe, bg_color, scaling_modifier=1.0, filter_small=False, filter_large=False,
                    fade_size=1.0, activated=None):
    """`activated` = (opacity, scaling, rotation) evaluated once by the caller when sweeping many cameras."""
    rasterizer = GaussianRasterizer(raster_settings=_settings(cam, pc, pipe, bg_color, scaling_modifier, filter_small,
                                                              filter_large, fade_size))
    opacity, scaling, rotation = activated if activated is not None else (pc.get_opacity, pc.get_scaling, pc.get_rotation)
    radii, pixel_sizes = rasterizer.preprocess_only(
        pc.get_xyz, opacity, scales=scaling, rotations=rotation, max_pixel_sizes=pc.get_max_pixel_sizes,
        min_pixel_sizes=pc.get_min_pixel_sizes, base_mask=pc.get_base_mask)
    return radii > 0, pixel_sizes, radii


@torch.no_grad()
def select_insertion_sources(base_cams, next_cams, pc, pipe, bg_color, base_reso_idx=0, pixel_size_threshold=1.0,
                             lanes=None, **filters):
    """Which base-level Gaussians become too small at the next resolution (train.py:283-315): the minimum over the
    cameras of the next-resolution pixel size, restricted to Gaussians the same camera sees at the base resolution;
    selected = min < threshold and target_reso_lvl == base_reso_idx.
    lanes: an optional multi_view.ViewPipeline — the two per-Gaussian launches of a camera pair (base and next resolution)
    then run on its two streams side by side; the running minimum is folded on the caller's stream in camera order, so the
    result is the same bits."""
    act = (pc.get_opacity, pc.get_scaling, pc.get_rotation)
    min_ps = torch.full_like(pc.get_min_pixel_sizes, float(pixel_size_threshold))
    if lanes is None or len(lanes.streams) < 2:
        for cb, cn in zip(base_cams, next_cams):
            base_vis, _, _ = view_visibility(cb, pc, pipe, bg_color, activated=act, **filters)
            _, ps, _ = view_visibility(cn, pc, pipe, bg_color, activated=act, **filters)
            min_ps = torch.where((ps > 0) & base_vis, torch.minimum(ps, min_ps), min_ps)
    else:
        s0, s1 = lanes.streams[0], lanes.streams[1]
        cur = lanes._fork()
        for cb, cn in zip(base_cams, next_cams):
            with torch.cuda.stream(s0):
                base_vis, _, _ = view_visibility(cb, pc, pipe, bg_color, activated=act, **filters)
            with torch.cuda.stream(s1):
                _, ps, _ = view_visibility(cn, pc, pipe, bg_color, activated=act, **filters)
            cur.wait_stream(s0)
            cur.wait_stream(s1)
            base_vis.record_stream(cur)
            ps.record_stream(cur)
            min_ps = torch.where((ps > 0) & base_vis, torch.minimum(ps, min_ps), min_ps)
        lanes._join(cur)
    selected = (min_ps < pixel_size_threshold) & (pc.target_reso_lvl == base_reso_idx)
    return selected, min_ps


@torch.no_grad()
def refresh_pixel_sizes(next_cams, pc, next_reso_idx, pipe, bg_color, **filters):
    """train.py:334-338: update_pixel_sizes(vis_filter, pixel_sizes, next_reso_idx) once per camera of the new level."""
    act = (pc.get_opacity, pc.get_scaling, pc.get_rotation)
    for cam in next_cams:
        _, ps, radii = view_visibility(cam, pc, pipe, bg_color, activated=act, **filters)
        update_training_stats(pc, None, radii, ps, next_reso_idx, base_mask=False, update_pixel_sizes=True, densify=False)
