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


# ---------------------------------------------------------------------------------------------------------------------
# FactoredGradExchange (one view per GPU per optimizer step, BASELINE config C4) and the pixel-size statistics
# ---------------------------------------------------------------------------------------------------------------------
def _torch_sh_reconstruct(means3D, gathered, n_views, deg, scale, out_dc, out_rest):
    """host restatement of msgs_sh_grad_from_views (tests only): scale * sum_v basis(dir_v) x drgb_v"""
    from gaussian_renderer.sh import sh_basis
    P = means3D.shape[0]
    acc = torch.zeros(P, 16, 3, dtype=torch.float64)
    for v in range(n_views):
        drgb = gathered[v, :3 * P].view(P, 3).double()
        cam = gathered[v, 3 * P:3 * P + 3].double()
        d = means3D.double() - cam[None]
        b = torch.zeros(P, 16, dtype=torch.float64)
        b[:, :(deg + 1) ** 2] = sh_basis(deg, d / d.norm(dim=1, keepdim=True))
        acc += b[:, :, None] * drgb[:, None, :]
    acc *= scale
    out_dc.copy_(acc[:, :1].float())
    out_rest.copy_(acc[:, 1:].float())


def _factored_worker(rank, world, port, ret):
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ms-gs_amd", "host")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussian_renderer.sh import sh_basis
    from view_parallel import (FactoredGradExchange, batched_pixel_size_update, gather_pixel_size_observations)
    torch.manual_seed(0)                                    # identical replicas
    P, deg = 203, 2                                         # P % 4 != 0
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    shapes = ((P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4))
    model = types.SimpleNamespace(active_sh_degree=deg)
    for n, s in zip(names, shapes):
        setattr(model, n, torch.nn.Parameter(torch.randn(*s)))
    sinks = {}

    def set_sinks(mapping, sh_factor=None):                 # stands in for diff_gaussian_rasterization.set_grad_sinks
        sinks.clear()
        if mapping:
            sinks.update(mapping=mapping, sh_factor=sh_factor)

    ex = FactoredGradExchange(model, world, reconstruct=_torch_sh_reconstruct, set_sinks=set_sinks)
    ok = ex.active and all(v.data_ptr() % 16 == 0 for v in ex.small.views)

    def view_grads(v):                                      # the per-view gradients a rasterizer backward would deliver
        g = torch.Generator().manual_seed(100 + v)
        small = {n: torch.randn(*s, generator=g) for n, s in zip(names, shapes) if n in FactoredGradExchange.SMALL}
        drgb = torch.randn(P, 3, generator=g)
        drgb[torch.rand(P, generator=g) < 0.4] = 0.0        # not rendered in this view
        cam = torch.randn(3, generator=g) * 5.0
        return small, drgb, cam

    for it in range(2):                                     # two optimizer steps reuse the same buffers
        ex.begin_view()
        small, drgb, cam = view_grads(10 * it + rank)
        for p, dest in sinks["mapping"].items():            # "backward": write into the sinks, hand aliases to autograd
            name = next(n for n in names if getattr(model, n) is p)
            dest.copy_(small[name])
            p.grad = dest.view(p.shape)
        sinks["sh_factor"].copy_(drgb)
        ex.end_view(cam)
        ex.finish()
        # expectation: plain average over the ranks' dense per-view gradients
        want = {n: torch.zeros(*s, dtype=torch.float64) for n, s in zip(names, shapes)}
        for r in range(world):
            sm, dr, cm = view_grads(10 * it + r)
            for n in sm:
                want[n] += sm[n].double() / world
            d = model._xyz.detach().double() - cm.double()[None]
            b = torch.zeros(P, 16, dtype=torch.float64)
            b[:, :(deg + 1) ** 2] = sh_basis(deg, d / d.norm(dim=1, keepdim=True))
            dsh = b[:, :, None] * dr.double()[:, None, :] / world
            want["_features_dc"] += dsh[:, :1]
            want["_features_rest"] += dsh[:, 1:]
        for n in names:
            g = getattr(model, n).grad
            ok = ok and g is not None and bool(torch.allclose(g.double(), want[n], rtol=1e-5, atol=1e-6))
        ok = ok and bool((model._features_rest.grad[:, (deg + 1) ** 2 - 1:] == 0).all())    # inactive bands stay zero
    ok = ok and ex.bytes_per_step() == 4 * ((world - 1) * (3 * P + 4) + 2 * (world - 1) * ex.small.flat.numel() // world)

    # pixel-size observations: gathered rows in rank order, -1 where not visible
    vis = torch.arange(P) % (rank + 2) == 0
    ps = torch.full((P,), float(rank + 1)) + torch.arange(P) * 1e-3
    obs, lvls = gather_pixel_size_observations(vis, ps, reso_lvl=1)
    ok = ok and tuple(obs.shape) == (world, P) and lvls.tolist() == [1] * world
    for r in range(world):
        vr = torch.arange(P) % (r + 2) == 0
        pr = torch.full((P,), float(r + 1)) + torch.arange(P) * 1e-3
        ok = ok and bool(torch.equal(obs[r], torch.where(vr, pr, torch.full((P,), -1.0))))
    # batched update == the documented formula == the sequential single-view update wherever ONE view saw the Gaussian
    lvl = torch.ones(P, dtype=torch.long)
    lvl[::7] = 0
    mx, mn = torch.full((P,), 3.0), torch.full((P,), 1.2)
    mn[::5] = -1.0
    mx_b, mn_b = mx.clone(), mn.clone()
    batched_pixel_size_update(mx_b, mn_b, lvl, obs, reso_lvl=1, reso_lvls=4)
    nseen = (obs >= 0).sum(0)
    one = (nseen == 1) & (lvl == 1)
    o1 = obs.max(dim=0).values                              # the single observation where nseen == 1
    seq_max = torch.maximum(mx * 0.95, o1)
    grown = torch.clip(mn * 1.05, -1)
    seq_min = torch.where(grown < 0, o1, torch.minimum(grown, o1))
    ok = ok and bool(torch.equal(mx_b[one], seq_max[one])) and bool(torch.equal(mn_b[one], seq_min[one]))
    untouched = (nseen == 0) | (lvl != 1)
    ok = ok and bool(torch.equal(mx_b[untouched], mx[untouched])) and bool(torch.equal(mn_b[untouched], mn[untouched]))
    two = (nseen == 2) & (lvl == 1)
    ok = ok and bool(two.any()) and bool(torch.equal(mx_b[two], torch.maximum(mx * 0.95, obs.max(0).values)[two]))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_factored_exchange_and_pixel_size_stats_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_factored_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


# ---------------------------------------------------------------------------------------------------------------------
# visible-rows exchange: only the rows some rank rendered travel; equal to the dense exchange
# ---------------------------------------------------------------------------------------------------------------------
def _visible_rows_worker(rank, world, port, ret):
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ms-gs_amd", "host")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from view_parallel import FactoredGradExchange
    torch.manual_seed(0)
    P, deg = 517, 1
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    shapes = ((P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4))
    ok = True
    results = {}
    for mode, frac in (("dense", None), ("rows", 0.95), ("rows-fallback", 0.05)):
        torch.manual_seed(0)
        model = types.SimpleNamespace(active_sh_degree=deg)
        for n, s in zip(names, shapes):
            setattr(model, n, torch.nn.Parameter(torch.randn(*s)))
        sinks = {}

        def set_sinks(mapping, sh_factor=None):
            sinks.clear()
            if mapping:
                sinks.update(mapping=mapping, sh_factor=sh_factor)
        ex = FactoredGradExchange(model, world, reconstruct=_torch_sh_reconstruct, set_sinks=set_sinks,
                                  visible_rows=mode != "dense", visible_rows_max_fraction=frac or 0.85)
        for it in range(2):
            g = torch.Generator().manual_seed(1000 * it + rank)
            vis = torch.rand(P, generator=g) < (0.35 if world == 2 else 0.2)        # this rank's visibility_filter
            ex.begin_view(torch.randn(3, generator=g))
            for p_, dest in sinks["mapping"].items():
                val = torch.randn(*p_.shape, generator=g)
                val[~vis] = 0.0                                                    # K9 writes zeros for rows it did not render
                dest.copy_(val)
                p_.grad = dest.view(p_.shape)
            drgb = torch.randn(P, 3, generator=g)
            drgb[~vis] = 0.0
            sinks["sh_factor"].copy_(drgb)
            ex.end_view(visibility=vis)
            ex.finish()
            results[(mode, it)] = {n: getattr(model, n).grad.clone() for n in names}
            if mode == "rows":
                ok = ok and ex.bytes_last_step_visible_rows() is not None and ex.last_union_rows < P
                ok = ok and ex.bytes_last_step_visible_rows() < ex.bytes_per_step()
            if mode == "rows-fallback":
                ok = ok and ex.bytes_last_step_visible_rows() is None
    for it in range(2):
        for n in names:
            for mode in ("rows", "rows-fallback"):
                a, b = results[("dense", it)][n], results[(mode, it)][n]
                # two ranks: a + b is the same float whichever buffer it travels in; four: the ring reduces the chunks of
                # the dense and of the packed buffer in different rank orders (last-bit differences)
                ok = ok and (bool(torch.equal(a, b)) if world == 2 else bool(torch.allclose(a, b, rtol=1e-6, atol=1e-7)))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_visible_rows_exchange_equals_dense(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_visible_rows_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


# ---- an optimizer step covering several views per rank: MultiViewStepExchange (one all-reduce per step) ----
def _multi_view_worker(rank, world, port, ret):
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ms-gs_amd", "host")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from view_parallel import MultiViewStepExchange, views_for_rank
    torch.manual_seed(0)
    P = 131
    shapes = dict(_xyz=(P, 3), _features_dc=(P, 1, 3), _features_rest=(P, 15, 3), _opacity=(P, 1), _scaling=(P, 3),
                  _rotation=(P, 4))
    model = types.SimpleNamespace(**{n: torch.nn.Parameter(torch.randn(*s)) for n, s in shapes.items()})

    class FakeAccumulator:                       # stands in for the HIP GradAccumulator: same contract, CPU tensors
        def __init__(self, leaves, dest):
            self.leaves, self.dest, self.count = list(leaves), [d.view(t.shape) for t, d in zip(leaves, dest)], 0

        def begin_step(self):
            self.count = 0

        def add(self, grads):                    # first view of the step stores, later views accumulate (msgs_grads_t)
            for d, g in zip(self.dest, grads):
                if self.count == 0:
                    d.copy_(g)
                else:
                    d.add_(g)
            self.count += 1

        def finish(self):
            for t, d in zip(self.leaves, self.dest):
                t.grad = d

    class FakePipeline:                          # stands in for multi_view.ViewPipeline.train_views
        def train_views(self, cams, pc, pipe, bg, backward_fn, accumulator=None, **kw):
            accumulator.begin_step()
            out = [backward_fn(i, dict(cam=c, acc=accumulator)) for i, c in enumerate(cams)]
            accumulator.finish()
            return out

    ex = MultiViewStepExchange(model, 8, make_accumulator=FakeAccumulator)
    mine = views_for_rank(8, rank, world)
    leaves = [getattr(model, n) for n in MultiViewStepExchange.LEAVES]

    def backward_fn(i, pkg):                     # view v contributes (v + 1) * (k + 1) to every element of leaf k
        pkg["acc"].add([torch.full_like(t, (pkg["cam"] + 1.0) * (k + 1)) for k, t in enumerate(leaves)])
        return pkg["cam"]
    ok = True
    for _ in range(2):                           # two optimizer steps: the bucket is reused
        seen = ex.step(FakePipeline(), mine, None, None, backward_fn)
        want = sum(v + 1.0 for v in range(8)) / 8.0
        ok = ok and seen == mine
        ok = ok and all(torch.allclose(t.grad, torch.full_like(t, want * (k + 1))) for k, t in enumerate(leaves))
        ok = ok and all(t.grad.data_ptr() == v.data_ptr() for t, v in zip(leaves, ex.bucket.views))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_view_step_exchange(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_multi_view_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
