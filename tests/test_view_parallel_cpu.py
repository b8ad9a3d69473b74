"""World-size-2 `gloo` tests of the view-parallel gradient exchange (the N>1 path of bench.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ms-gs_amd", "host")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from view_parallel import FlatGradBucket, all_reduce_densification_stats, views_for_rank
    torch.manual_seed(0)                                    # identical replicas on every rank
    P = 257
    params = [torch.nn.Parameter(torch.randn(P, 3)), torch.nn.Parameter(torch.randn(P, 1, 3)),
              torch.nn.Parameter(torch.randn(P, 15, 3)), torch.nn.Parameter(torch.randn(P, 1)),
              torch.nn.Parameter(torch.randn(P, 3)), torch.nn.Parameter(torch.randn(P, 4))]
    bucket = FlatGradBucket(params)
    # 236 B per Gaussian; every slice padded to a 16-byte boundary (P = 257 is odd on purpose)
    assert 59 * P <= bucket.flat.numel() <= 59 * P + 3 * len(params)
    assert all(v.data_ptr() % 16 == 0 for v in bucket.views)
    views = views_for_rank(8, rank, world)
    # a per-view "loss" whose gradient is known in closed form: sum_v (v+1) * sum(p * c_k)
    bucket.zero()
    for v in views:
        loss = sum(((v + 1.0) * (k + 1)) * p.sum() for k, p in enumerate(params))
        loss.backward()
    assert all(p.grad.data_ptr() == w.data_ptr() for p, w in zip(params, bucket.views))   # accumulated in place
    bucket.all_reduce(average_over=8)
    want_scale = sum(v + 1.0 for v in range(8)) / 8.0
    ok = all(torch.allclose(p.grad, torch.full_like(p, want_scale * (k + 1))) for k, p in enumerate(params))
    # statistics
    gsum, cnt, rad = torch.full((P,), float(rank + 1)), torch.ones(P), torch.full((P,), float(10 * (rank + 1)))
    all_reduce_densification_stats(gsum, cnt, rad)
    ok = ok and bool((gsum == 3).all()) and bool((cnt == world).all()) and bool((rad == 20).all())
    # a second step reuses the same bucket without reallocation
    ptr = bucket.flat.data_ptr()
    bucket.zero()
    (params[0].sum()).backward()
    bucket.all_reduce()
    ok = ok and bucket.flat.data_ptr() == ptr and bool(torch.allclose(params[0].grad, torch.full_like(params[0], float(world))))
    # pipelined exchange: alternate buckets, exchange of view k in flight while view k+1 accumulates
    from view_parallel import PipelinedGradExchange
    ex = PipelinedGradExchange(params, world)
    seen = []
    for k in range(5):
        ex.begin_view()
        assert all(p.grad.data_ptr() == w.data_ptr() for p, w in zip(params, ex.current.views))
        (float(10 * k + rank + 1) * params[0].sum() + params[5].sum()).backward()
        seen.append(ex.end_view())
    ex.drain()
    # after the drain the two buckets hold the AVERAGED gradients of the last two views (k = 3 -> bucket 1, k = 4 -> 0)
    for k, b in ((4, ex.buckets[0]), (3, ex.buckets[1])):
        want0 = sum(10 * k + r + 1 for r in range(world)) / world
        ok = ok and bool(torch.allclose(b.views[0], torch.full_like(params[0], want0)))
        ok = ok and bool(torch.allclose(b.views[5], torch.ones_like(params[5])))
        ok = ok and bool((b.views[2] == 0).all())
    ok = ok and seen[0] is ex.buckets[0] and seen[1] is ex.buckets[1] and seen[2] is ex.buckets[0]
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_single_process_paths():
    import sys
    from view_parallel import FlatGradBucket, views_for_rank
    p = [torch.nn.Parameter(torch.ones(5, 3))]
    b = FlatGradBucket(p)
    (p[0] * 2).sum().backward()
    b.all_reduce(average_over=4)
    assert torch.allclose(p[0].grad, torch.full((5, 3), 0.5))
    assert views_for_rank(8, 1, 4) == [1, 5] and sum(len(views_for_rank(8, r, 8)) for r in range(8)) == 8
    from view_parallel import PipelinedGradExchange
    ex = PipelinedGradExchange(p, world=2)                      # no process group: local average only
    for k in range(3):
        ex.begin_view()
        (p[0] * (k + 1)).sum().backward()
        b = ex.end_view()
        assert torch.allclose(b.views[0], torch.full((5, 3), (k + 1) / 2.0))
    ex.drain()
