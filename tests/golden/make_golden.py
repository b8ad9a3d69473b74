#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the authoring container only
(it imports /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

Two kinds of fixture (data only: inputs + expected outputs, never reference source):
  (A) PINNED BY THE REFERENCE — produced by importing the reference's own pure-torch helpers:
        sh_colors.npz     utils/sh_utils.py::eval_sh (deg 0..3) (+0.5, clamp as gaussian_renderer/__init__.py:86-87)
        cameras.npz       utils/graphics_utils.py::getWorld2View2 / getProjectionMatrix assembled exactly as
                          scene/cameras.py:54-57 (world_view_transform, full_proj_transform, camera_center)
        cov3d.npz         the arithmetic of utils/general_utils.py::build_scaling_rotation/strip_symmetric and
                          scene/gaussian_model.py:33-37; those functions hard-code device="cuda" and cannot run
                          here, so they are executed with torch.zeros/torch.device patched to CPU (same code
                          object, different allocation device)
  (B) SELF-GENERATED regression pins of the rasterizer itself (the reference's rasterizer is un-vendored:
      PARITY UNPINNED) — the float64 torch oracle's outputs and autograd gradients on small seeded scenes:
        raster_*.npz
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def gen_reference_pinned():
    sys.path.insert(0, REF)
    from utils import sh_utils, graphics_utils
    g = torch.Generator().manual_seed(1234)
    # (A1) SH colours
    N = 64
    sh = torch.randn(N, 3, 16, generator=g, dtype=torch.float64)          # reference layout [..., C, K]
    dirs = torch.randn(N, 3, generator=g, dtype=torch.float64)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"sh": sh.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        rgb = sh_utils.eval_sh(deg, sh, dirs)
        out[f"rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()
        out[f"raw_deg{deg}"] = rgb.numpy()
    out["rgb2sh"] = sh_utils.RGB2SH(torch.linspace(0, 1, 11, dtype=torch.float64)).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_colors.npz"), **out)

    # (A2) cameras
    cams = {}
    rng = np.random.RandomState(7)
    for i in range(4):
        A = rng.randn(3, 3)
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T = rng.randn(3) * 2.0
        fovx, fovy = 0.6 + 0.2 * i, 0.4 + 0.15 * i
        wvt = torch.tensor(graphics_utils.getWorld2View2(Q, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        cams[f"R{i}"] = Q
        cams[f"T{i}"] = T
        cams[f"fov{i}"] = np.array([fovx, fovy])
        cams[f"wvt{i}"] = wvt.numpy()
        cams[f"full{i}"] = full.numpy()
        cams[f"center{i}"] = center.numpy()
    np.savez_compressed(os.path.join(HERE, "cameras.npz"), **cams)

    # (A3) covariance packing: run the reference functions with CPU allocation
    from utils import general_utils
    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    torch.zeros = cpu_zeros
    try:
        s = torch.exp(torch.randn(32, 3, generator=g)) * 0.1
        q = torch.randn(32, 4, generator=g)
        L = general_utils.build_scaling_rotation(1.7 * s, q)          # gaussian_model.py:34 (modifier 1.7)
        cov = general_utils.strip_symmetric(L @ L.transpose(1, 2))     # :35-36
        R = general_utils.build_rotation(q)
    finally:
        torch.zeros = real_zeros
    np.savez_compressed(os.path.join(HERE, "cov3d.npz"), scales=s.numpy(), quats=q.numpy(), modifier=1.7,
                        cov=cov.numpy(), rot=R.numpy())
    sys.path.remove(REF)


def gen_raster_pins():
    import scenes
    from oracle import torch_oracle as to
    cases = {
        "raster_base": dict(P=250, W=48, H=32, seed=101, deg=3, ms=False, st=dict(filter_small=False, filter_large=False, fade_size=1.0), bg=(0.2, 0.5, 0.7)),
        "raster_ms": dict(P=300, W=56, H=40, seed=102, deg=2, ms=True, st=dict(filter_small=True, filter_large=True, fade_size=0.0), bg=(0.0, 0.0, 0.0)),
        "raster_fade": dict(P=300, W=40, H=40, seed=103, deg=1, ms=True, st=dict(filter_small=True, filter_large=True, fade_size=1.0), bg=(1.0, 1.0, 1.0)),
    }
    for name, c in cases.items():
        W, H = c["W"], c["H"]
        k = 0.004 * 1920.0 / W * (0.2 if c["ms"] else 0.5)
        sc = scenes.frustum_scene(c["P"], W, H, seed=c["seed"], sh_degree=c["deg"], multiscale=c["ms"], scale_k=k)
        cam = scenes.front_camera(W, H)
        bg = torch.tensor(c["bg"])
        dL = scenes.grad_seed(W, H, c["seed"])
        outs, grads = to.forward_backward(sc, cam, c["st"], bg, dL)
        d = dict(P=c["P"], W=W, H=H, seed=c["seed"], deg=c["deg"], ms=int(c["ms"]), scale_k=k,
                 filter_small=int(c["st"]["filter_small"]), filter_large=int(c["st"]["filter_large"]),
                 fade_size=c["st"]["fade_size"], bg=np.array(c["bg"], dtype=np.float32))
        for kk, v in outs.items():
            d["out_" + kk] = v.numpy()
        for kk, v in grads.items():
            d["grad_" + kk] = v.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)


if __name__ == "__main__":
    if os.path.isdir(REF):
        gen_reference_pinned()
    else:
        print("no /root/reference here: keeping the committed reference-pinned fixtures")
    gen_raster_pins()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
