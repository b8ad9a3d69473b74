"""Times the photometric loss fwd+bwd at 1080p (not a test): msgs_loss_* vs the reference's torch formulation."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from loss_utils import l1_ssim_loss
from oracle import loss_oracle as lo
H, W = 1080, 1920
gt = torch.rand(3, H, W).cuda(); img = (gt + 0.1 * torch.randn(3, H, W).cuda()).clamp(0, 1)
def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def mine():
    x = img.requires_grad_(True); x.grad = None
    l, _ = l1_ssim_loss(x, gt, 0.2); l.backward()
w2 = torch.from_numpy(lo.window_2d()).cuda().expand(3, 1, 11, 11).contiguous()
def theirs():                                   # utils/loss_utils.py formulation (restated) + train.py:209-211
    x = img.requires_grad_(True); x.grad = None
    conv = lambda t: F.conv2d(t, w2, padding=5, groups=3)
    m1, m2 = conv(x), conv(gt)
    m1s, m2s, m12 = m1.pow(2), m2.pow(2), m1 * m2
    s1, s2, s12 = conv(x * x) - m1s, conv(gt * gt) - m2s, conv(x * gt) - m12
    S = ((2 * m12 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((m1s + m2s + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))
    l = 0.8 * torch.abs(x - gt).mean() + 0.2 * (1.0 - S.mean()); l.backward()
print("msgs_loss fwd+bwd: %.3f ms" % bench(mine))
print("torch formulation fwd+bwd: %.3f ms" % bench(theirs))
