"""Real spherical harmonics up to degree 3 for the convert_SHs_python path of render(): colour = sum_k Y_k(dir) * sh_k,
with the basis functions Y_k in the sign / ordering convention of the reference (/root/reference/utils/sh_utils.py:57-112),
evaluated here as one basis tensor contracted with the coefficients.  Pinned against the reference's own eval_sh by
tests/golden/sh_colors.npz."""
import torch

Y00 = 0.28209479177387814                      # 1 / (2 sqrt(pi))
Y1 = 0.4886025119029199                        # sqrt(3 / (4 pi))
Y2 = (1.0925484305920792, 0.31539156525252005, 0.5462742152960396)
Y3 = (0.5900435899266435, 2.890611442640554, 0.4570457994644658, 0.3731763325901154, 1.445305721320277)


def sh_basis(deg, dirs):
    """dirs [..., 3] unit vectors -> [..., (deg+1)^2] basis values."""
    if not 0 <= deg <= 3:
        raise ValueError(f"SH degree {deg} not in 0..3")
    x, y, z = dirs.unbind(-1)
    cols = [torch.full_like(x, Y00)]
    if deg >= 1:
        cols += [-Y1 * y, Y1 * z, -Y1 * x]
    if deg >= 2:
        x2, y2, z2 = x * x, y * y, z * z
        cols += [Y2[0] * x * y, -Y2[0] * y * z, Y2[1] * (2.0 * z2 - x2 - y2), -Y2[0] * x * z, Y2[2] * (x2 - y2)]
    if deg >= 3:
        r = 4.0 * z2 - x2 - y2
        cols += [-Y3[0] * y * (3.0 * x2 - y2), Y3[1] * x * y * z, -Y3[2] * y * r,
                 Y3[3] * z * (2.0 * z2 - 3.0 * x2 - 3.0 * y2), -Y3[2] * x * r, Y3[4] * z * (x2 - y2),
                 -Y3[0] * x * (x2 - 3.0 * y2)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg, sh, dirs):
    """sh [..., C, K >= (deg+1)^2], dirs [..., 3] unit -> [..., C]."""
    n = (deg + 1) ** 2
    if sh.shape[-1] < n:
        raise ValueError(f"{sh.shape[-1]} coefficients given, degree {deg} needs {n}")
    return (sh[..., :n] * sh_basis(deg, dirs).unsqueeze(-2)).sum(dim=-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / Y00


def SH2RGB(sh):
    return sh * Y00 + 0.5
