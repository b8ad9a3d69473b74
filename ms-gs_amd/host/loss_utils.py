"""GPU loss functions of the MS-GS train step, same names and argument meaning as the
reference's utils/loss_utils.py (/root/reference/utils/loss_utils.py:17-63), backed by libmsgs_hip.so:

  l1_ssim_loss(image, gt, lambda_dssim)   the whole photometric loss of /root/reference/train.py:209-211 in two launches
                                          forward (+ one backward): returns (loss, Ll1) — Ll1 detached, for logging
  ssim(img1, img2)                        reference signature; window_size 11 / size_average True only
  l1_loss(network_output, gt)             reference signature

Gradients flow to the FIRST argument only (the rendered image); the ground truth is data.  No CPU / torch fallback:
CPU tensors raise RuntimeError.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _backend as _C


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _as_planes(t, what):
    if t.device.type != "cuda":
        raise RuntimeError(f"{what}: tensor lives on {t.device}; the loss kernels are GPU-only (no CPU path)")
    if t.dim() < 2:
        raise ValueError(f"{what}: expected [..., H, W], got {tuple(t.shape)}")
    H, W = int(t.shape[-2]), int(t.shape[-1])
    planes = t.numel() // max(H * W, 1)                 # leading dims (batch, channel) are independent planes
    if planes < 1 or H < 1 or W < 1:
        raise ValueError(f"{what}: empty image {tuple(t.shape)}")
    return t.detach().to(torch.float32).contiguous(), planes, H, W


class _L1SsimLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        x, planes, H, W = _as_planes(image, "image")
        y, p2, H2, W2 = _as_planes(gt, "gt")
        if (planes, H, W) != (p2, H2, W2):
            raise ValueError(f"image {tuple(image.shape)} and gt {tuple(gt.shape)} differ in shape")
        lib = _C.lib
        need_grad = image.requires_grad
        with torch.cuda.device(x.device):
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            scratch = torch.empty(int(lib.msgs_loss_scratch_bytes(planes, H, W)), dtype=torch.uint8, device=x.device)
            out3 = torch.empty(3, dtype=torch.float32, device=x.device)
            _C.check(lib.msgs_loss_forward(_ptr(x), _ptr(y), planes, H, W, float(lambda_dssim), _ptr(out3),
                                           _ptr(scratch), scratch.numel(), int(need_grad), stream), "msgs_loss_forward")
        ctx.dims = (planes, H, W, float(lambda_dssim), image.shape, image.dtype)
        ctx.save_for_backward(x, y, scratch)
        ctx.mark_non_differentiable(out3)
        return out3[0], out3

    @staticmethod
    def backward(ctx, grad_loss, _grad_out3):
        x, y, scratch = ctx.saved_tensors
        planes, H, W, lam, shape, dtype = ctx.dims
        up = grad_loss.detach().to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _C.check(_C.lib.msgs_loss_backward(_ptr(x), _ptr(y), planes, H, W, lam, _ptr(up), _ptr(scratch),
                                               scratch.numel(), _ptr(grad), stream), "msgs_loss_backward")
        return grad.view(shape).to(dtype), None, None


def l1_ssim_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda_dssim) * l1_loss + lambda_dssim * (1 - ssim)  ->  (loss, Ll1).  train.py:209-211."""
    loss, out3 = _L1SsimLoss.apply(image, gt, lambda_dssim)
    return loss, out3[1]


def ssim(img1, img2, window_size=11, size_average=True):
    """loss_utils.py:32-63.  Only the configuration the reference uses (window 11, size_average) is implemented."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("ssim: only window_size=11, size_average=True (the values every MS-GS caller uses)")
    loss, _ = _L1SsimLoss.apply(img1, img2, 1.0)          # lambda = 1: loss = 1 - ssim
    return 1.0 - loss


def l1_loss(network_output, gt):
    """loss_utils.py:17-18."""
    loss, _ = _L1SsimLoss.apply(network_output, gt, 0.0)  # lambda = 0: loss = L1 mean
    return loss
