"""Seeded synthetic scenes for the parity tests, smoke() and bench.py.

Every tensor is drawn from a CPU ``torch.Generator`` so that every box sees the
same bits.  Camera matrices follow the recipe of the
reference's ``scene/cameras.py:54-57`` (world_view_transform = W2C^T,
full_proj_transform = world_view_transform @ P^T, camera_center =
inverse(world_view_transform)[3, :3]) with ``getWorld2View2`` /
``getProjectionMatrix`` of ``utils/graphics_utils.py:38-71`` restated below; the
restatement is pinned against the reference's own functions by
``tests/golden/camera_*.npz`` (generated with tests/golden/make_golden.py).

Frozen generator constants (do not change without regenerating tests/golden):
  SCALE_K = 0.004, SCALE_SIGMA = 0.6, OFFSCREEN = 1.15, NEAR_FRACTION = 0.02
"""
import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

try:                                   # every harness that draws a scene: thread pools sized for the container's CPU quota
    from hostinfo import limit_thread_pools
    limit_thread_pools(reserve=0)
except ImportError:                    # ms-gs_amd/host not on the path (the scene generator itself does not need it)
    pass

SCALE_K = 0.004
SCALE_SIGMA = 0.6
OFFSCREEN = 1.15
NEAR_FRACTION = 0.02


# ----------------------------------------------------------------------------
# camera helpers (reference: utils/graphics_utils.py:38-71, scene/cameras.py:48-57)
# ----------------------------------------------------------------------------
def world2view2(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """utils/graphics_utils.py:38-49 — R is the C2W rotation, t the W2C translation."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = np.asarray(t)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + np.asarray(translate)) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:51-71."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """The attributes gaussian_renderer.render() reads from a viewpoint camera
    (gaussian_renderer/__init__.py:34-46; scene/cameras.py:65-76 MiniCam)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device):
        return Camera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.znear, self.zfar)


def make_camera(R, T, FoVx, FoVy, width, height, znear=0.01, zfar=100.0):
    wvt = torch.tensor(world2view2(np.asarray(R, dtype=np.float64), np.asarray(T, dtype=np.float64))).transpose(0, 1)
    proj = projection_matrix(znear=znear, zfar=zfar, fovX=FoVx, fovY=FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return Camera(int(width), int(height), float(FoVx), float(FoVy),
                  wvt.contiguous(), full.contiguous(), center.contiguous(), znear, zfar)


def front_camera(width, height):
    """R = I, T = 0 looking down +z, f = 1000*W/1920 px."""
    f = 1000.0 * width / 1920.0
    fovx = 2.0 * math.atan(width / (2.0 * f))
    fovy = 2.0 * math.atan(height / (2.0 * f))
    return make_camera(np.eye(3), np.zeros(3), fovx, fovy, width, height)


def ring_camera(v, n_views, width, height, radius=8.0):
    """Camera v of n on a ring in the y=0 plane looking at the origin, up = -y
    (COLMAP convention, y down: scene/dataset_readers.py:202-203)."""
    az = 2.0 * math.pi * v / n_views
    C = np.array([radius * math.sin(az), 0.0, -radius * math.cos(az)])
    z = -C / np.linalg.norm(C)                       # forward (camera +z) towards the origin
    y = np.array([0.0, 1.0, 0.0])                    # camera +y = world +y (down in COLMAP)
    x = np.cross(y, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R_c2w = np.stack([x, y, z], axis=1)              # columns = camera axes in world
    T = -R_c2w.T @ C                                 # W2C translation
    f = 1000.0 * width / 1920.0
    fovx = 2.0 * math.atan(width / (2.0 * f))
    fovy = 2.0 * math.atan(height / (2.0 * f))
    return make_camera(R_c2w, T, fovx, fovy, width, height)


# ----------------------------------------------------------------------------
# Gaussian sets
# ----------------------------------------------------------------------------
@dataclass
class Scene:
    """Activated per-Gaussian tensors in the layout the rasterizer op takes
    (gaussian_renderer/__init__.py:57-64,94-108)."""
    means3D: torch.Tensor          # [P,3]
    scales: torch.Tensor           # [P,3]   (already exp-activated)
    rotations: torch.Tensor        # [P,4]   (w,x,y,z), normalised
    opacities: torch.Tensor        # [P,1]   (already sigmoid-activated)
    shs: torch.Tensor              # [P,16,3]
    max_pixel_sizes: torch.Tensor  # [P]
    min_pixel_sizes: torch.Tensor  # [P]
    occ_multiplier: torch.Tensor   # [P,4,1]
    dc_delta: torch.Tensor         # [P,12,1]
    base_mask: torch.Tensor        # [P] bool
    sh_degree: int = 3
    target_reso_lvl: Optional[torch.Tensor] = None
    meta: dict = field(default_factory=dict)

    @property
    def P(self):
        return self.means3D.shape[0]

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if torch.is_tensor(v) else v
        return Scene(**kw)

    def subset(self, idx):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v[idx].contiguous() if torch.is_tensor(v) else v
        return Scene(**kw)


def _common_attrs(g, P, sh_degree, n_coeffs=16):
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g))
    shs = torch.zeros(P, n_coeffs, 3)
    shs[:, 0, :] = torch.randn(P, 3, generator=g)
    if n_coeffs > 1:
        shs[:, 1:, :] = 0.15 * torch.randn(P, n_coeffs - 1, 3, generator=g)
    return rot, opac, shs


def frustum_scene(P, width, height, seed, sh_degree=3, multiscale=False, scale_k=SCALE_K):
    """Gaussians scattered through the view frustum of front_camera(width, height)
    (SURVEY §8(d) 'means/scales/rotations/opacity/SH'; MS fields when multiscale)."""
    g = torch.Generator().manual_seed(seed)
    f = 1000.0 * width / 1920.0
    tanx = width / (2.0 * f)
    tany = height / (2.0 * f)
    z = 0.5 + 11.5 * torch.rand(P, generator=g)
    x = (2.0 * torch.rand(P, generator=g) - 1.0) * OFFSCREEN * z * tanx
    y = (2.0 * torch.rand(P, generator=g) - 1.0) * OFFSCREEN * z * tany
    near = torch.rand(P, generator=g) < NEAR_FRACTION
    z_near = -2.0 + 2.2 * torch.rand(P, generator=g)
    z = torch.where(near, z_near, z)
    means = torch.stack([x, y, z], dim=1)
    zz = z.abs().clamp_min(0.2)
    # the pixel footprint is resolution independent: s/z*f = scale_k*1000*(W/1920) px
    logs = torch.log(scale_k * zz)[:, None] + SCALE_SIGMA * torch.randn(P, 3, generator=g)
    scales = torch.exp(logs)
    rot, opac, shs = _common_attrs(g, P, sh_degree)
    maxps = -torch.ones(P)
    minps = -torch.ones(P)
    lvl = torch.zeros(P, dtype=torch.long)
    if multiscale:
        u = torch.rand(P, generator=g)
        lvl = torch.zeros(P, dtype=torch.long)
        lvl[u >= 0.70] = 2
        lvl[u >= 0.82] = 4
        lvl[u >= 0.92] = 6
        scales = scales * (2.0 ** lvl.float())[:, None]
        has_min = (torch.rand(P, generator=g) < 0.5) & (lvl == 0)
        minps = torch.where(has_min, 0.5 + 1.5 * torch.rand(P, generator=g), minps)
        maxps = torch.where(lvl > 0, 2.0 + 6.0 * torch.rand(P, generator=g), maxps)
    return Scene(means3D=means.contiguous(), scales=scales.contiguous(), rotations=rot.contiguous(),
                 opacities=opac.contiguous(), shs=shs.contiguous(),
                 max_pixel_sizes=maxps.contiguous(), min_pixel_sizes=minps.contiguous(),
                 occ_multiplier=torch.ones(P, 4, 1), dc_delta=torch.zeros(P, 12, 1),
                 base_mask=torch.zeros(P, dtype=torch.bool), sh_degree=sh_degree,
                 target_reso_lvl=lvl,
                 meta=dict(kind="frustum", seed=seed, width=width, height=height, multiscale=multiscale,
                           scale_k=scale_k))


def ball_scene(P, seed, sh_degree=3, radius=4.0, log_s=math.log(0.02)):
    """World-space set for the view-parallel config C4."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(P, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    r = radius * torch.rand(P, generator=g) ** (1.0 / 3.0)
    means = d * r[:, None]
    scales = torch.exp(log_s + SCALE_SIGMA * torch.randn(P, 3, generator=g))
    rot, opac, shs = _common_attrs(g, P, sh_degree)
    return Scene(means3D=means.contiguous(), scales=scales.contiguous(), rotations=rot.contiguous(),
                 opacities=opac.contiguous(), shs=shs.contiguous(),
                 max_pixel_sizes=-torch.ones(P), min_pixel_sizes=-torch.ones(P),
                 occ_multiplier=torch.ones(P, 4, 1), dc_delta=torch.zeros(P, 12, 1),
                 base_mask=torch.zeros(P, dtype=torch.bool), sh_degree=sh_degree,
                 target_reso_lvl=torch.zeros(P, dtype=torch.long),
                 meta=dict(kind="ball", seed=seed))


def grad_seed(width, height, seed):
    """Fixed dL/dcolor ~ N(0,1)/N used by the kernel-parity backward tests."""
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.randn(3, height, width, generator=g) / float(width * height)


# BASELINE.json configs -----------------------------------------------------------------------
def config(name):
    """Returns (scene, camera, settings dict) for a BASELINE.json config."""
    if name == "C1":      # 10k / 256x256 / SH0 forward only
        W = H = 256
        return frustum_scene(10_000, W, H, seed=0, sh_degree=0), front_camera(W, H), \
            dict(filter_small=False, filter_large=False, fade_size=1.0)
    if name == "C2":      # 100k / 800x800 / SH3 fwd+bwd
        W = H = 800
        return frustum_scene(100_000, W, H, seed=1, sh_degree=3), front_camera(W, H), \
            dict(filter_small=False, filter_large=False, fade_size=1.0)
    if name == "C3":      # 1M / 1920x1080 / multi-scale filters, fade 0 (train.py:124-125)
        W, H = 1920, 1080
        return frustum_scene(1_000_000, W, H, seed=2, sh_degree=3, multiscale=True), front_camera(W, H), \
            dict(filter_small=True, filter_large=True, fade_size=0.0)
    if name == "C5":      # 5M / 4K / multi-scale
        W, H = 3840, 2160
        return frustum_scene(5_000_000, W, H, seed=5, sh_degree=3, multiscale=True), front_camera(W, H), \
            dict(filter_small=True, filter_large=True, fade_size=0.0)
    raise KeyError(name)


def config_c4(n_views=8, P=1_000_000, width=1920, height=1080):
    scene = ball_scene(P, seed=4)
    cams = [ring_camera(v, n_views, width, height) for v in range(n_views)]
    return scene, cams, dict(filter_small=False, filter_large=False, fade_size=1.0)
