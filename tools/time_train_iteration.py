"""Times the fused training iteration at C3 (not a test): FusedAdam.step() as its own launch against the Adam step taken
inside the per-Gaussian backward kernel (fused_train_iteration(step_in_backward=True)), interleaved A/B/A/B, median periods;
and the per-Gaussian backward kernel alone from the library's HIP events."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes, bench
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE
from synthetic_model import SyntheticGaussians
from train_epilogue import FusedAdam
from train_step import fused_train_iteration
torch.autograd.set_multithreading_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
scene, cam, settings = scenes.config(name)
W, H = cam.image_width, cam.image_height
dev = torch.device("cuda")
cam, bg = cam.to(dev), torch.zeros(3, device=dev)
gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
res = {}
for rnd in range(2):
    for mode in (False, True):
        model = SyntheticGaussians(scene, dev)
        opt = FusedAdam(model.training_setup(7, scene.target_reso_lvl), lr=0.0, eps=1e-15)
        fn = lambda: fused_train_iteration(model, opt, cam, gt, PIPE, bg, step_in_backward=mode, **settings)
        med, ts = bench.period_median(fn, 40, 10, torch.cuda.synchronize)
        timer = dgr._C.KernelTimer(only=("preprocess_bwd",))
        dgr._C.set_timer(timer)
        acc = []
        for _ in range(4):
            fn(); torch.cuda.synchronize(); acc.append(timer.read_ms()["preprocess_bwd"])
        dgr._C.set_timer(None)
        k = {"preprocess_bwd": round(sorted(acc)[1], 4)}
        print(f"round {rnd} step_in_backward={mode}: median {med:.4f} ms  min {ts[0]:.4f}  p90 {ts[int(0.9 * len(ts))]:.4f}   "
              f"preprocess_bwd {k.get('preprocess_bwd')}")
        del model, opt
