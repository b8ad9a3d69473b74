"""Soak run (not a test): N steps of render() + backward() with the level, the view and the entry (reference pattern / fused) drawn at
random per step — the instance-count guess, the speculative stage 2, the kernel shapes and the allocator all keep changing — while
checking that (i) a repeated (level, view, entry) reproduces its image and its dL/dxyz bit for bit, (ii) everything stays finite,
(iii) device memory stops growing.  python tools/soak.py [steps=1500] [seed=0]"""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
torch.autograd.set_multithreading_enabled(False)
from parity_utils import PIPE
from gaussian_renderer import render, render_fused
from synthetic_model import SyntheticGaussians

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ball = SyntheticGaussians(scenes.ball_scene(int(os.environ.get("SOAK_P", "300000")), seed=4), "cuda", requires_grad=True)
frustum = SyntheticGaussians(scenes.frustum_scene(int(os.environ.get("SOAK_P", "300000")), 1920, 1080, seed=2, sh_degree=3, multiscale=True), "cuda",
                             requires_grad=True)
bg = torch.zeros(3, device="cuda")
seen, bad, t0 = {}, 0, time.perf_counter()
peak_after_warm = None
for it in range(steps):
    k, v, fused = rng.randint(0, 6), rng.randint(0, 8), rng.random() < 0.5
    filt = rng.random() < 0.7
    W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
    if v == 8:                       # the multi-scale frustum scene from the front
        pc, cam = frustum, scenes.front_camera(W, H).to("cuda")
    else:                            # the ball from ring camera v
        pc, cam = ball, scenes.ring_camera(v, 8, W, H).to("cuda")
    dL = scenes.grad_seed(W, H, 40 + k).to("cuda")
    for p_ in pc.parameters():
        p_.grad = None
    settings = dict(filter_small=filt, filter_large=filt, fade_size=0.0)
    out = (render_fused if fused else render)(cam, pc, PIPE, bg, **settings)
    out["render"].backward(dL)
    img, gx = out["render"].detach(), pc._xyz.grad.detach()
    if not (torch.isfinite(img).all() and torch.isfinite(gx).all()):
        bad += 1; print("non-finite at step", it, (k, v, fused, filt))
    key = (k, v, fused, filt)
    sig = (img.double().sum().item(), img.view(-1)[::97].clone(), gx.double().sum().item(), gx.view(-1)[::101].clone())
    if key in seen:
        a = seen[key]
        if not (a[0] == sig[0] and torch.equal(a[1], sig[1]) and a[2] == sig[2] and torch.equal(a[3], sig[3])):
            bad += 1; print("NOT reproduced at step", it, key, a[0], sig[0], a[2], sig[2])
    else:
        seen[key] = sig
    if it == 300:
        torch.cuda.synchronize(); peak_after_warm = torch.cuda.memory_reserved()
torch.cuda.synchronize()
grow = torch.cuda.memory_reserved() - (peak_after_warm or 0)
print(f"soak: {steps} steps, {len(seen)} distinct (level, view, entry, filters), {bad} problems, "
      f"{1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step incl. checks, reserved memory growth after step 300: {grow / 1e6:.1f} MB")
