"""Large-size sanity run of the auxiliary kernels (not a test): 4K loss, 5M-Gaussian Adam + statistics, 5M-point kNN and
voxel pooling — checks against torch / sampled brute force."""
import sys, os, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from loss_utils import l1_ssim_loss
from train_epilogue import FusedAdam, update_training_stats
from simple_knn._C import distCUDA2
from voxel_pool import VoxelGrouping
from test_epilogue_cpu import _groups
torch.manual_seed(0)
# loss at 4K
gt = torch.rand(3, 2160, 3840, device="cuda"); img = (gt + 0.05 * torch.randn_like(gt)).clamp(0, 1).requires_grad_(True)
loss, l1 = l1_ssim_loss(img, gt, 0.2); loss.backward(); torch.cuda.synchronize()
print("loss 4K", loss.item(), l1.item(), "l1 check", (img.detach() - gt).abs().mean().item(), "grad finite", bool(torch.isfinite(img.grad).all()))
# Adam at 5M Gaussians (295M floats) vs torch fused
P = 5_000_000
g = torch.Generator().manual_seed(1)
base = _groups(1000, g)
def mk():
    return [{"params": [torch.nn.Parameter(torch.randn(P, *b["params"][0].shape[1:], device="cuda"))], "lr": b["lr"], "name": b["name"]} for b in base]
a = mk(); b = [{"params": [torch.nn.Parameter(x["params"][0].detach().clone())], "lr": x["lr"], "name": x["name"]} for x in a]
for x, y in zip(a, b):
    gr = torch.randn_like(x["params"][0]) * 1e-3; x["params"][0].grad = gr; y["params"][0].grad = gr.clone()
oa, ob = FusedAdam(a, lr=0.0, eps=1e-15), torch.optim.Adam(b, lr=0.0, eps=1e-15, fused=True)
oa.step(); ob.step(); torch.cuda.synchronize()
print("adam 5M max rel diff", max(((x["params"][0] - y["params"][0]).abs().max() / y["params"][0].abs().max()).item() for x, y in zip(a, b)))
del a, b, oa, ob; torch.cuda.empty_cache()
# kNN on 5M points: compare 2000 sampled queries with brute force
pts = torch.randn(5_000_000, 3, device="cuda") * torch.tensor([4.0, 2.0, 0.5], device="cuda")
t = time.perf_counter(); d2 = distCUDA2(pts); torch.cuda.synchronize(); print("knn 5M ms", (time.perf_counter() - t) * 1e3)
idx = torch.randint(0, pts.shape[0], (2000,), device="cuda")
dd = torch.cdist(pts[idx].double(), pts.double()) ** 2
dd[torch.arange(2000, device="cuda"), idx] = float("inf")
ref = dd.topk(3, dim=1, largest=False).values.mean(1)
print("knn 5M max rel err on 2000 samples", ((d2[idx].double() - ref).abs() / ref).max().item())
del dd
# voxel pooling 5M
vg = VoxelGrouping(pts, 0.05)
feat = torch.randn(5_000_000, 6, device="cuda")
out = vg.average(feat)
v0 = vg.voxel_index[:1]
sel = ((pts / 0.05).floor().int() == v0).all(1)
print("voxel 5M voxels", vg.num_voxels, "first voxel mean err", (out[0] - feat[sel].mean(0)).abs().max().item(), "members", int(sel.sum()))
