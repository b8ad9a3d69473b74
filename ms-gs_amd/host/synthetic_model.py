"""A minimal stand-in for the reference's GaussianModel parameter container
(/root/reference/scene/gaussian_model.py:50-77,127-183) holding exactly the leaf Parameters and the
activated getters that render() reads.  Used by tests, smoke() and bench.py with synthetic scenes;
the optimiser / densification / PLY machinery of the reference class is out of scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def _strip_symmetric(S):
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def _build_rotation(r):
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


class SyntheticGaussians:
    """Leaf parameters in the reference's parametrisation (log-scale, logit-opacity, raw quaternion,
    dc/rest SH split) built from a scenes.Scene of activated values."""

    LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def __init__(self, scene, device, requires_grad=True):
        dev = torch.device(device)
        mk = lambda t: nn.Parameter(t.detach().to(dev, torch.float32).contiguous().clone(),
                                    requires_grad=requires_grad)
        self.max_sh_degree = 3 if scene.shs.shape[1] >= 16 else int(round(scene.shs.shape[1] ** 0.5)) - 1
        self.active_sh_degree = scene.sh_degree
        self._xyz = mk(scene.means3D)
        self._features_dc = mk(scene.shs[:, :1, :])
        self._features_rest = mk(scene.shs[:, 1:, :])
        self._scaling = mk(torch.log(scene.scales))
        self._rotation = mk(scene.rotations)
        op = scene.opacities.clamp(1e-6, 1 - 1e-6)
        self._opacity = mk(inverse_sigmoid(op))
        self._occ_multiplier = scene.occ_multiplier.to(dev).contiguous()
        self._dc_delta = scene.dc_delta.to(dev).contiguous()
        self.max_pixel_sizes = scene.max_pixel_sizes.to(dev).contiguous()
        self.min_pixel_sizes = scene.min_pixel_sizes.to(dev).contiguous()
        self.base_gaussian_mask = scene.base_mask.to(dev).contiguous()

    def parameters(self):
        return [getattr(self, n) for n in self.LEAVES]

    # per-group learning rates of OptimizationParams (/root/reference/arguments/__init__.py:73-79)
    LRS = dict(xyz=0.00016, f_dc=0.0025, f_rest=0.0025 / 20.0, opacity=0.05, scaling=0.005, rotation=0.001)

    def training_setup(self, reso_lvls=1, target_reso_lvl=None, spatial_lr_scale=1.0):
        """The state GaussianModel.training_setup / create_from_pcd allocate (gaussian_model.py:222-233) and the
        optimizer param groups of :235-246 (the two lr-0 groups without gradients are omitted).  Returns the groups."""
        P, dev = self._xyz.shape[0], self._xyz.device
        self.reso_lvls = int(reso_lvls)
        self.xyz_gradient_accum = torch.zeros((P, self.reso_lvls, 1), device=dev)
        self.denom = torch.zeros((P, self.reso_lvls, 1), device=dev)
        self.max_radii2D = torch.zeros((P,), device=dev)
        self.target_reso_lvl = (torch.zeros((P,), dtype=torch.long, device=dev) if target_reso_lvl is None
                                else target_reso_lvl.to(dev, torch.long).contiguous())
        names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
        return [{"params": [getattr(self, leaf)], "lr": self.LRS[n] * (spatial_lr_scale if n == "xyz" else 1.0), "name": n}
                for n, leaf in zip(names, self.LEAVES)]

    # --- getters with the reference's activations (gaussian_model.py:127-183) ---
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return F.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_occ_multiplier(self):
        return self._occ_multiplier

    @property
    def get_dc_delta(self):
        return self._dc_delta

    @property
    def get_max_pixel_sizes(self):
        return self.max_pixel_sizes

    @property
    def get_min_pixel_sizes(self):
        return self.min_pixel_sizes

    @property
    def get_base_mask(self):
        return self.base_gaussian_mask

    def get_covariance(self, scaling_modifier=1):
        """gaussian_model.py:33-37: L = R(q) diag(mod * s); Sigma = L L^T packed (xx,xy,xz,yy,yz,zz)."""
        L = _build_rotation(self._rotation) * (scaling_modifier * self.get_scaling)[:, None, :]
        return _strip_symmetric(L @ L.transpose(1, 2))
