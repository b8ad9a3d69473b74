"""CPU restatement (numpy, float64 by default) of the photometric loss of the MS-GS train step and its gradient.

TEST INFRASTRUCTURE ONLY — imported by tests/; ms-gs_amd/ never imports this.

PINNED: utils/loss_utils.py of the reference is plain PyTorch and imports here; tests/golden/make_loss_golden.py runs
the reference's l1_loss / ssim and autograd on seeded inputs and commits inputs + outputs as
tests/golden/loss_*.npz; tests/test_loss_cpu.py checks this restatement against them.

Follows /root/reference/utils/loss_utils.py:17-18 (L1), :23-31 (window), :43-63 (SSIM) and train.py:209-211.
"""
import math

import numpy as np


def window_taps():
    """loss_utils.py:23-25: float32 taps, normalised in float32."""
    g = np.array([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=np.float32)
    # torch's float32 sum of these 11 values equals the correctly rounded exact sum (checked against the golden window)
    return g / np.float32(g.astype(np.float64).sum())


def window_2d():
    """loss_utils.py:28-29: the 2-D window is the float32 outer product of the taps (each entry rounded to float32)."""
    g = window_taps()
    return (g[:, None] * g[None, :]).astype(np.float32)


def _conv(a, w):
    """zero-padded 11x11 window over the last two axes (F.conv2d(..., padding=5, groups=C)).  w: [11] taps
    (separable evaluation, what the HIP kernels do) or [11,11] (the reference's float32 2-D window, 121 taps)."""
    H, W = a.shape[-2:]
    pad = np.zeros(a.shape[:-2] + (H + 10, W + 10), dtype=a.dtype)
    pad[..., 5:5 + H, 5:5 + W] = a
    if w.ndim == 2:
        return sum(w[i, j] * pad[..., i:i + H, j:j + W] for i in range(11) for j in range(11))
    tmp = sum(w[k] * pad[..., :, k:k + W] for k in range(11))
    return sum(w[k] * tmp[..., k:k + H, :] for k in range(11))


def l1_ssim(img, gt, lambda_dssim, dtype=np.float64, separable=False):
    """returns dict(loss, l1, ssim, grad) with grad = dloss/dimg.  separable=False uses the reference's exact 2-D
    float32 window (pinning); separable=True the product of the 1-D taps in `dtype` (differs by the float32 rounding
    of the 121 products: ~1e-7 relative on the window, up to ~1e-6 on SSIM of noise images)."""
    x, y = np.asarray(img, dtype=dtype), np.asarray(gt, dtype=dtype)
    w = (window_taps() if separable else window_2d()).astype(dtype)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m1, m2 = _conv(x, w), _conv(y, w)
    s1 = _conv(x * x, w) - m1 * m1
    s2 = _conv(y * y, w) - m2 * m2
    s12 = _conv(x * y, w) - m1 * m2
    a1, a2 = 2 * m1 * m2 + C1, 2 * s12 + C2
    b1, b2 = m1 * m1 + m2 * m2 + C1, s1 + s2 + C2
    S = a1 * a2 / (b1 * b2)
    n = x.size
    l1 = np.abs(x - y).mean()
    ssim = S.mean()
    # dS/d(moment) at every window centre, then the adjoint of the (symmetric) windowed sums
    dS_ds12 = 2 * a1 / (b1 * b2)
    dS_ds1 = -S / b2
    dS_dm1 = 2 * m2 * a2 / (b1 * b2) - 2 * m1 * S / b1
    A = dS_dm1 - 2 * m1 * dS_ds1 - m2 * dS_ds12
    dssim = (_conv(A, w) + 2 * x * _conv(dS_ds1, w) + y * _conv(dS_ds12, w)) / n
    grad = (1 - lambda_dssim) * np.sign(x - y) / n - lambda_dssim * dssim
    return dict(loss=(1 - lambda_dssim) * l1 + lambda_dssim * (1 - ssim), l1=l1, ssim=ssim, grad=grad)
