"""Times distCUDA2 at 1M points (not a test)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host")):
    sys.path.insert(0, p)
import torch
from simple_knn._C import distCUDA2
for name, pts in (("uniform 1M", torch.rand(1_000_000, 3)), ("gaussian clusters 1M", torch.randn(1000, 1, 3) * 5 + torch.randn(1000, 1000, 3) * 0.2)):
    pts = pts.reshape(-1, 3).cuda()
    for _ in range(2): distCUDA2(pts)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): distCUDA2(pts)
    torch.cuda.synchronize(); print(name, "%.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
