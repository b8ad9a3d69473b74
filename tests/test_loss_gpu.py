"""GPU parity of the fused L1 + SSIM loss (msgs_loss_forward / msgs_loss_backward) through the C ABI: against the
golden vectors produced by the reference's utils/loss_utils.py, against the float64 oracle on other shapes, and
against the same torch formulation running on the GPU at full size."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loss_oracle as lo

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "loss_*.npz")))
LOSS_ATOL = 1e-5            # forward tolerance of the north-star (abs)
GRAD_RTOL = 1e-4            # gradient tolerance of the north-star (rel. to the max-norm)


def _run(img, gt, lam, upstream=None):
    from loss_utils import l1_ssim_loss
    x = torch.from_numpy(np.asarray(img, dtype=np.float32)).cuda().requires_grad_(True)
    y = torch.from_numpy(np.asarray(gt, dtype=np.float32)).cuda()
    loss, l1 = l1_ssim_loss(x, y, lam)
    (loss if upstream is None else loss * upstream).backward()
    torch.cuda.synchronize()
    return loss.item(), l1.item(), x.grad.cpu().numpy()


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_loss_vs_reference_golden(path):
    z = np.load(path)
    lam = float(z["lambda_dssim"])
    loss, l1, grad = _run(z["img"], z["gt"], lam)
    assert abs(loss - float(z["loss_f64"])) <= LOSS_ATOL
    assert abs(l1 - float(z["l1_f64"])) <= LOSS_ATOL
    g = z["grad_f64"]
    # case f: the reference's own float32 gradient is 1.4e-4 from its float64 one (tests/test_loss_cpu.py)
    tol = 5e-4 if path.endswith("loss_f.npz") else GRAD_RTOL
    assert np.abs(grad - g).max() <= tol * np.abs(g).max()
    assert np.abs(grad - z["grad_f32"]).max() <= tol * np.abs(g).max()


@pytest.mark.parametrize("C,H,W,lam,seed", [(3, 1, 1, 0.2, 0), (3, 5, 3, 0.2, 1), (3, 31, 33, 0.2, 2), (3, 32, 32, 0.2, 3),
                                             (3, 33, 65, 0.7, 4), (2, 100, 7, 0.2, 5), (3, 135, 240, 0.2, 6)])
def test_loss_vs_oracle_ragged_shapes(C, H, W, lam, seed):
    rng = np.random.default_rng(seed)
    img, gt = rng.random((C, H, W), dtype=np.float32), rng.random((C, H, W), dtype=np.float32)
    r = lo.l1_ssim(img, gt, lam)
    loss, l1, grad = _run(img, gt, lam, upstream=0.1)                      # train.py:212-215 loss multiplier
    assert abs(loss - r["loss"]) <= LOSS_ATOL and abs(l1 - r["l1"]) <= LOSS_ATOL
    assert np.abs(grad - 0.1 * r["grad"]).max() <= GRAD_RTOL * np.abs(0.1 * r["grad"]).max()


def test_reference_signatures_ssim_l1_and_batch_dims():
    from loss_utils import l1_loss, ssim
    rng = np.random.default_rng(11)
    img, gt = rng.random((1, 3, 40, 56), dtype=np.float32), rng.random((1, 3, 40, 56), dtype=np.float32)
    r = lo.l1_ssim(img[0], gt[0], 1.0)
    x = torch.from_numpy(img).cuda().requires_grad_(True)
    y = torch.from_numpy(gt).cuda()
    s = ssim(x, y)                                 # metrics call it with a leading batch dim (metrics.py / train.py eval)
    assert abs(s.item() - r["ssim"]) <= LOSS_ATOL
    s.backward()
    assert x.grad.shape == x.shape
    assert np.abs(x.grad.cpu().numpy()[0] + r["grad"]).max() <= GRAD_RTOL * np.abs(r["grad"]).max()   # d ssim = -d(1-ssim)
    assert abs(l1_loss(x, y).item() - r["l1"]) <= LOSS_ATOL
    with torch.no_grad():                          # no-grad path: no maps kept, same value
        assert abs(ssim(x, y).item() - r["ssim"]) <= LOSS_ATOL


def test_identical_images_and_determinism():
    from loss_utils import l1_ssim_loss
    g = torch.Generator().manual_seed(2)
    y = torch.rand(3, 64, 80, generator=g).cuda()
    x = y.clone().requires_grad_(True)
    loss, l1 = l1_ssim_loss(x, y, 0.2)
    loss.backward()
    assert abs(loss.item()) <= 1e-6 and l1.item() == 0.0
    assert x.grad.abs().max().item() <= 1e-6
    x2 = torch.rand(3, 270, 480, generator=g).cuda()
    y2 = torch.rand(3, 270, 480, generator=g).cuda()
    vals = {l1_ssim_loss(x2, y2, 0.2)[0].item() for _ in range(5)}
    assert len(vals) == 1                          # fixed-order reduction: run-to-run identical


def test_full_size_vs_torch_formulation_on_gpu():
    """1080p: the same formula in float64 torch ops on the GPU (conv2d with the 2-D window, autograd)."""
    from loss_utils import l1_ssim_loss
    g = torch.Generator().manual_seed(7)
    gt = torch.rand(3, 1080, 1920, generator=g)
    img = (gt + 0.1 * torch.randn(3, 1080, 1920, generator=g)).clamp(0, 1)
    x = img.cuda().requires_grad_(True)
    loss, _ = l1_ssim_loss(x, gt.cuda(), 0.2)
    loss.backward()
    xd = img.cuda().double().requires_grad_(True)
    yd = gt.cuda().double()
    w = torch.from_numpy(lo.window_2d()).cuda().double().expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=5, groups=3)
    m1, m2 = conv(xd), conv(yd)
    s1, s2, s12 = conv(xd * xd) - m1 * m1, conv(yd * yd) - m2 * m2, conv(xd * yd) - m1 * m2
    S = ((2 * m1 * m2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((m1 * m1 + m2 * m2 + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))
    ref = 0.8 * (xd - yd).abs().mean() + 0.2 * (1 - S.mean())
    ref.backward()
    assert abs(loss.item() - ref.item()) <= LOSS_ATOL
    assert (x.grad.double() - xd.grad).abs().max().item() <= GRAD_RTOL * xd.grad.abs().max().item()


def test_loss_argument_errors():
    from loss_utils import l1_ssim_loss
    with pytest.raises(ValueError):
        l1_ssim_loss(torch.zeros(3, 8, 8).cuda(), torch.zeros(3, 8, 9).cuda())
    with pytest.raises(ValueError):
        l1_ssim_loss(torch.zeros(3, 8, 8).cuda(), torch.zeros(3, 8, 8).cuda(), 1.5)
