"""Generates tests/golden/loss_*.npz by running the REFERENCE's own loss functions (plain PyTorch, importable in the
build container) on seeded inputs:  python tests/golden/make_loss_golden.py   (needs /root/reference; never run on the
GPU box).  Stored: inputs, l1_loss, ssim, the train.py:209-211 loss and its autograd gradient w.r.t. the image."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from utils.loss_utils import l1_loss, ssim, gaussian  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"a": (3, 37, 53, 0.2, 1), "b": (3, 64, 96, 0.2, 2), "c": (1, 9, 7, 0.5, 3), "d": (3, 40, 33, 1.0, 4),
         "e": (3, 32, 32, 0.0, 5), "f": (3, 70, 45, 0.2, 6)}

for name, (C, H, W, lam, seed) in CASES.items():
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(C, H, W, generator=g, dtype=torch.float64)
    if name == "f":        # a render close to its target (late training): smooth target, small perturbation
        yy, xx = torch.meshgrid(torch.linspace(0, 3, H, dtype=torch.float64), torch.linspace(0, 3, W, dtype=torch.float64),
                                indexing="ij")
        gt = torch.stack([0.5 + 0.5 * torch.sin(xx * (c + 1) + yy) for c in range(C)])
        img = (gt + 0.02 * torch.randn(C, H, W, generator=g, dtype=torch.float64)).clamp(0, 1)
    else:
        img = torch.rand(C, H, W, generator=g, dtype=torch.float64)
    img, gt = img.float().double(), gt.float().double()      # float32-representable inputs for both precisions
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        x = img.to(dt).clone().requires_grad_(True)
        y = gt.to(dt)
        Ll1 = l1_loss(x, y)
        s = ssim(x, y)
        loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)
        loss.backward()
        out.update({f"l1_{tag}": Ll1.item(), f"ssim_{tag}": s.item(), f"loss_{tag}": loss.item(),
                    f"grad_{tag}": x.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, f"loss_{name}.npz"), img=img.float().numpy(), gt=gt.float().numpy(),
                        lambda_dssim=lam, window=gaussian(11, 1.5).numpy(), **out)
    print(name, out["loss_f32"], out["loss_f64"])
