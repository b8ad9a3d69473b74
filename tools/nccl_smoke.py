import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
print("backend", dist.get_backend())
t = torch.arange(10, device=dev, dtype=torch.float32)
w = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True); w.wait(); torch.cuda.synchronize(); print("AVG ok", t[:3].tolist())
big = torch.ones(59_000_000, device=dev)
w = dist.all_reduce(big, op=dist.ReduceOp.AVG, async_op=True)
x = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)     # compute while it is in flight
w.wait(); torch.cuda.synchronize(); print("async ok", big[0].item())
tm = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(tm, op=dist.ReduceOp.MAX); print("max ok", tm.item())
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group(); print("done")
