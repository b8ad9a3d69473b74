"""Times the train-step epilogue at P = 1M (not a test): FusedAdam vs torch.optim.Adam (foreach default, and
fused=True), update_training_stats vs the reference's masked-indexing torch sequence."""
import sys, os, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from train_epilogue import FusedAdam, update_training_stats
from test_epilogue_cpu import _groups
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gen = torch.Generator().manual_seed(0)
base = _groups(P, gen)
def mk():
    return [{"params": [torch.nn.Parameter(g["params"][0].detach().clone().cuda())], "lr": g["lr"], "name": g["name"]} for g in base]
def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
nfl = sum(g["params"][0].numel() for g in base)
for name, ctor in (("FusedAdam (msgs_adam_step)", lambda g: FusedAdam(g, lr=0.0, eps=1e-15)),
                   ("torch.optim.Adam (reference call)", lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15)),
                   ("torch.optim.Adam(fused=True)", lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15, fused=True))):
    gs = mk()
    for g in gs: g["params"][0].grad = torch.randn_like(g["params"][0]) * 1e-3
    opt = ctor(gs)
    ms = bench(opt.step)
    print(f"{name}: {ms:.3f} ms/step  {28 * nfl / ms / 1e6:.0f} GB/s algorithmic")
L, lvl = 7, 0
m = types.SimpleNamespace(reso_lvls=L, xyz_gradient_accum=torch.zeros(P, L, 1).cuda(), denom=torch.zeros(P, L, 1).cuda(),
                          max_radii2D=torch.zeros(P).cuda(), max_pixel_sizes=torch.rand(P).cuda(), min_pixel_sizes=torch.rand(P).cuda(),
                          base_gaussian_mask=torch.zeros(P, dtype=torch.bool).cuda(), target_reso_lvl=torch.randint(0, L, (P,)).cuda())
radii = torch.where(torch.rand(P) < 0.45, torch.zeros(P, dtype=torch.int32), torch.randint(1, 60, (P,), dtype=torch.int32)).cuda()
ps = (torch.rand(P) * 10).cuda()
vsp = torch.zeros(P, 3, device="cuda", requires_grad=True); vsp.grad = torch.randn(P, 3).cuda()
print("update_training_stats: %.3f ms" % bench(lambda: update_training_stats(m, vsp, radii, ps, lvl, base_mask=True)))
def ref_stats():
    vis = radii > 0
    m.base_gaussian_mask = m.base_gaussian_mask | vis
    mask = vis & (m.target_reso_lvl == lvl)
    mn = torch.clip(m.min_pixel_sizes[mask] * 1.05, -1)
    m.min_pixel_sizes[mask] = torch.where(mn < 0, torch.where(ps[mask] > 0, ps[mask], mn), torch.where(ps[mask] > 0, torch.min(mn, ps[mask]), mn))
    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis])
    m.xyz_gradient_accum[:, lvl][vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
    m.denom[:, lvl][vis] += 1
print("reference torch sequence: %.3f ms" % bench(ref_stats))
