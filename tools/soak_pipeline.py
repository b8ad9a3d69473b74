"""Soak of the view pipeline: sweeps over random subsets of ring cameras at random pyramid levels, lane counts, entries (reference
pattern / fused) and accumulation modes, each compared BIT FOR BIT with the same views rendered one after the other on one stream
(outputs, per-view means2D gradients, accumulated leaf gradients).  Stream-ordering and allocator-reuse bugs show up as mismatches.
usage: soak_pipeline.py [sweeps] [P]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
torch.autograd.set_multithreading_enabled(False)
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render, render_fused
from multi_view import ViewPipeline
from synthetic_model import SyntheticGaussians
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
LEAVES = SyntheticGaussians.LEAVES
sc = scenes.ball_scene(P, seed=9, log_s=-3.6)
pc_a, pc_b = SyntheticGaussians(sc, "cuda"), SyntheticGaussians(sc, "cuda")
bg = torch.tensor([0.05, 0.1, 0.2], device="cuda")
pipes = {n: ViewPipeline("cuda", n_streams=n) for n in (2, 3)}
rng = random.Random(5)
KEYS = ("render", "acc_pixel_size", "depth", "radii", "pixel_sizes")
bad = 0
t0 = time.time()
for it in range(sweeps):
    k = rng.choice([0, 0, 1, 2, 3, 5])
    W, H = int(1280 / 2 ** k), int(720 / 2 ** k)
    nv = rng.randint(1, 6)
    cams = [scenes.ring_camera(rng.randrange(16), 16, W, H).to("cuda") for _ in range(nv)]
    dLs = [scenes.grad_seed(W, H, rng.randrange(100)).cuda() for _ in range(nv)]
    fn = rng.choice([render, render_fused])
    st = rng.choice([dict(filter_small=False, filter_large=False, fade_size=1.0), dict(filter_small=True, filter_large=True, fade_size=0.0)])
    lanes, share, acc, train = rng.choice([2, 3]), rng.random() < 0.5, rng.random() < 0.6, rng.random() < 0.7
    if rng.random() < 0.1:
        dgr._last_instances.clear()                       # first-frame path inside the pipeline
    if rng.random() < 0.1:
        for key in list(dgr._last_instances):
            dgr._last_instances[key] = 4096               # guess exceeded inside the pipeline
    for m in (pc_a, pc_b):
        for p_ in m.parameters():
            p_.grad = None
    if train:
        ref = []
        for c, d in zip(cams, dLs):
            o = fn(c, pc_a, PIPE, bg, **st)
            o["render"].backward(d)
            ref.append((o, o["viewspace_points"].grad.clone()))
        got = []

        def bwd(i, pkg):
            pkg["render"].backward(dLs[i])
            got.append(pkg)
        pipes[lanes].train_views(cams, pc_b, PIPE, bg, bwd, render_fn=fn, share_getters=share and fn is render,
                                 accumulate_in_kernel=acc, **st)
        torch.cuda.synchronize()
        ok = all(torch.equal(g[k_], r[0][k_]) for g, r in zip(got, ref) for k_ in KEYS)
        ok = ok and all(torch.equal(g["viewspace_points"].grad, r[1]) for g, r in zip(got, ref))
        ok = ok and all(torch.equal(getattr(pc_a, n).grad, getattr(pc_b, n).grad) for n in LEAVES)
    else:
        with torch.no_grad():
            ref = [fn(c, pc_a, PIPE, bg, **st) for c in cams]
            got = pipes[lanes].render_views(cams, pc_b, PIPE, bg, render_fn=fn, share_getters=share and fn is render, **st)
        torch.cuda.synchronize()
        ok = all(torch.equal(g[k_], r[k_]) for g, r in zip(got, ref) for k_ in KEYS)
    if not ok:
        bad += 1
        print(f"MISMATCH sweep {it}: k={k} views={nv} lanes={lanes} fn={fn.__name__} share={share} acc={acc} train={train} st={st}", flush=True)
print(f"{sweeps} sweeps at P={P}: {bad} mismatches, {time.time() - t0:.1f} s, reserved {torch.cuda.memory_reserved() >> 20} MiB")
sys.exit(1 if bad else 0)
