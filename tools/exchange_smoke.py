"""One GPU, one rank, real RCCL: the N>1 loop of bench.py (PipelinedGradExchange, direct sinks, async ncclAvg) with the
collectives forced on, checked against plain single-GPU gradients.  Not a test (needs the nccl backend)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ["MSGS_EXCHANGE_FORCE"] = "1"
import torch, torch.distributed as dist
import scenes
from parity_utils import PIPE
from gaussian_renderer import render
from synthetic_model import SyntheticGaussians
from view_parallel import PipelinedGradExchange
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
sc, cam, st = scenes.config("C3")
camd, bg = cam.to(dev), torch.zeros(3, device=dev)
dL = scenes.grad_seed(cam.image_width, cam.image_height, 2).to(dev)
ref = SyntheticGaussians(sc, dev); render(camd, ref, PIPE, bg, **st)["render"].backward(dL)
pc = SyntheticGaussians(sc, dev)
ex = PipelinedGradExchange(pc.parameters(), world=1, direct=True)
print("active", ex.active, "avg_op", ex.avg_op)
def step():
    ex.begin_view(); render(camd, pc, PIPE, bg, **st)["render"].backward(dL); return ex.end_view()
for _ in range(5): step()
ex.drain(); torch.cuda.synchronize(); t = time.perf_counter()
K = 20
for _ in range(K): b = step()
ex.drain(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
err = max(((p.grad - q.grad).abs().max() / q.grad.abs().max()).item() for p, q in zip(pc.parameters(), ref.parameters()))
print("ms/step with a 236 MB self-all-reduce per view in flight: %.3f   max rel grad diff vs plain: %.2e" % (dt * 1e3, err))
dist.barrier(); dist.destroy_process_group()
