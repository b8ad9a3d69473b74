"""hunt: slab mode forced on the giants scenes of tests/test_occlusion_gpu.py (occlusion on): stats + identity"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import test_occlusion_gpu as T
import test_slab_gpu as S
import diff_gaussian_rasterization as dgr

for seed in range(12):
    g = torch.Generator().manual_seed(1000 + seed)
    W = int(torch.randint(300, 900, (1,), generator=g)); H = int(torch.randint(220, 600, (1,), generator=g))
    P = int(torch.randint(300, 6000, (1,), generator=g)); n_g = int(torch.randint(5, 120, (1,), generator=g))
    op = None if seed % 3 == 0 else (0.2 + 0.79 * torch.rand(n_g, generator=g))
    sc = T._giants_scene(P, W, H, seed, n_g, giant_scale=float(0.2 + 1.5 * torch.rand(1, generator=g)), giant_opacity=op,
                         elongate=float(torch.tensor([1.0, 1.0, 4.0, 12.0])[seed % 4]))
    W2, H2 = W * 3, H * 3          # >= 2048 tiles
    sc = T._giants_scene(P, W2, H2, seed, n_g, giant_scale=float(0.2 + 1.5 * torch.rand(1, generator=g)), giant_opacity=op,
                         elongate=float(torch.tensor([1.0, 1.0, 4.0, 12.0])[seed % 4]))
    cam = scenes.front_camera(W2, H2).to("cuda")
    bg = torch.rand(3, generator=g).cuda(); dL = scenes.grad_seed(W2, H2, seed).cuda()
    for occ in (1, 0):
        prev = dgr._C.lib.msgs_set_occlusion(occ)
        try:
            one = S._run(sc, cam, T.PLAIN, bg, dL, "never")
            for frac in ("0.05", "0.3"):
                sl = S._run(sc, cam, T.PLAIN, bg, dL, frac, calls=2)
                same = all(torch.equal(sl[0][k], one[0][k]) for k in T.OUT_KEYS)
                print(seed, W2, H2, P, "occ", occ, frac, "D", one[2], sl[3], "identical" if same else "DIFFERENT", flush=True)
        finally:
            dgr._C.lib.msgs_set_occlusion(prev)
