"""Host-side counterpart of the reference's `gaussian_renderer.render`
(/root/reference/gaussian_renderer/__init__.py:18-119): same signature, same argument meaning, same
seven-key result dict, so callers written against the reference (train.py:203, render.py:32,
viewer.py:71, render_traj.py:102) read the same.

The reference's own gaussian_renderer module runs unchanged on top of
ms-gs_amd/diff_gaussian_rasterization — THAT is the drop-in.  This module exists so the parity tests,
smoke() and bench.py can drive the operator through the reference's call pattern without importing
the reference, which does not travel to the GPU box.

Duck typing:
  pc    — the GaussianModel getters read at reference :57-64,71-75,82-89
  pipe  — PipelineParams (/root/reference/arguments/__init__.py:64-69)
  viewpoint_camera — Camera / MiniCam attributes (/root/reference/scene/cameras.py:17-76)
"""
import math
import types

import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

from .sh import eval_sh

# PipelineParams defaults of the reference (/root/reference/arguments/__init__.py:64-69)
PIPE = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)

RESULT_KEYS = ("render", "acc_pixel_size", "depth", "viewspace_points", "visibility_filter", "radii",
               "pixel_sizes")


def _settings(cam, pc, pipe, bg_color, scaling_modifier, filter_small, filter_large, fade_size):
    """The 15 raster settings (reference :37-53)."""
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(0.5 * cam.FoVx), tanfovy=math.tan(0.5 * cam.FoVy),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=cam.camera_center,
        prefiltered=False, debug=pipe.debug,
        filter_small=filter_small, filter_large=filter_large, fade_size=fade_size)


def _shape_inputs(pc, pipe, scaling_modifier):
    """Either the Python-side covariance or (scales, rotations) for the op (reference :66-75)."""
    if pipe.compute_cov3D_python:
        return dict(cov3D_precomp=pc.get_covariance(scaling_modifier), scales=None, rotations=None)
    return dict(cov3D_precomp=None, scales=pc.get_scaling, rotations=pc.get_rotation)


def _colour_inputs(cam, pc, pipe, override_color):
    """Override colour, Python-side SH->RGB, or raw SH coefficients for the op (reference :77-91)."""
    if override_color is not None:
        return dict(shs=None, colors_precomp=override_color)
    if not pipe.convert_SHs_python:
        return dict(shs=pc.get_features, colors_precomp=None)
    n_coeff = (pc.max_sh_degree + 1) ** 2
    per_channel = pc.get_features.transpose(1, 2).view(-1, 3, n_coeff)
    view_dir = pc.get_xyz - cam.camera_center.repeat(pc.get_features.shape[0], 1)
    view_dir = view_dir / view_dir.norm(dim=1, keepdim=True)
    rgb = eval_sh(pc.active_sh_degree, per_channel, view_dir)
    return dict(shs=None, colors_precomp=torch.clamp_min(rgb + 0.5, 0.0))


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
           filter_small=False, filter_large=False, fade_size=1.0):
    """Render `pc` from `viewpoint_camera`.  `bg_color` must already live on the GPU."""
    xyz = pc.get_xyz
    # gradient sink for the screen-space means (reference :27-31): a non-leaf zero tensor whose
    # .grad is retained so densification can read viewspace_points.grad[:, :2]
    # (/root/reference/scene/gaussian_model.py:698-701)
    viewspace = torch.zeros_like(xyz, requires_grad=True) + 0
    try:
        viewspace.retain_grad()
    except Exception:
        pass

    rasterizer = GaussianRasterizer(raster_settings=_settings(
        viewpoint_camera, pc, pipe, bg_color, scaling_modifier, filter_small, filter_large, fade_size))
    image, acc_pixel_size, depth, radii, pixel_sizes = rasterizer(
        means3D=xyz,
        means2D=viewspace,
        opacities=pc.get_opacity,
        max_pixel_sizes=pc.get_max_pixel_sizes,
        min_pixel_sizes=pc.get_min_pixel_sizes,
        occ_multiplier=pc.get_occ_multiplier,
        dc_delta=pc.get_dc_delta,
        base_mask=pc.get_base_mask,
        **_colour_inputs(viewpoint_camera, pc, pipe, override_color),
        **_shape_inputs(pc, pipe, scaling_modifier))
    # radii == 0 <=> frustum-culled, zero-area or filtered out: excluded from densification statistics
    # (reference :110-119)
    values = (image, acc_pixel_size, depth, viewspace, radii > 0, radii, pixel_sizes)
    return dict(zip(RESULT_KEYS, values))


def render_fused(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, filter_small=False,
                 filter_large=False, fade_size=1.0):
    """Opt-in variant of render(): hands the RAW parameters of the model (pc._xyz,
    pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation — the attribute names of the reference's
    GaussianModel, scene/gaussian_model.py:53-58) to the rasterizer, which evaluates the getters' activations
    (:127-153) and the SH concatenation (:144-149) inside its kernels.  Same result dict as render(); gradients land
    on the same leaf Parameters.  Not usable with override_color / convert_SHs_python / compute_cov3D_python."""
    xyz = pc._xyz
    # gradient sink for the screen-space means: the op never reads its values, so — unlike render(), which keeps the
    # reference's `zeros_like(...) + 0` — a leaf without the fill and add kernels will do (.grad is populated the same)
    viewspace = torch.empty_like(xyz, requires_grad=True)
    rasterizer = GaussianRasterizer(raster_settings=_settings(
        viewpoint_camera, pc, pipe, bg_color, scaling_modifier, filter_small, filter_large, fade_size))
    image, acc_pixel_size, depth, radii, pixel_sizes = rasterizer.forward_raw(
        xyz, viewspace, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation,
        max_pixel_sizes=pc.get_max_pixel_sizes, min_pixel_sizes=pc.get_min_pixel_sizes,
        occ_multiplier=pc.get_occ_multiplier, dc_delta=pc.get_dc_delta, base_mask=pc.get_base_mask)
    values = (image, acc_pixel_size, depth, viewspace, radii > 0, radii, pixel_sizes)
    return dict(zip(RESULT_KEYS, values))
