"""Runs N whole training iterations at C3 (render_fused + fused loss + backward + statistics + FusedAdam) and one
distCUDA2 call, for rocprofv3 --kernel-trace --stats / --pmc (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
from parity_utils import PIPE
from synthetic_model import SyntheticGaussians
from train_epilogue import FusedAdam
from train_step import fused_train_iteration
from simple_knn._C import distCUDA2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc, cam, st = scenes.config("C3")
model = SyntheticGaussians(sc, "cuda")
opt = FusedAdam(model.training_setup(7, sc.target_reso_lvl), lr=0.0, eps=1e-15)
gt = torch.rand(3, cam.image_height, cam.image_width, generator=torch.Generator().manual_seed(3)).cuda()
camd, bg = cam.to("cuda"), torch.zeros(3, device="cuda")
for _ in range(n):
    fused_train_iteration(model, opt, camd, gt, PIPE, bg, **st)
distCUDA2(model._xyz.detach())
torch.cuda.synchronize()
print("done", n)
