"""GPU parity of the train-step epilogue kernels (msgs_adam_step, msgs_densify_stats) through the C ABI:
against the numpy oracle, and against torch.optim.Adam — the implementation the reference calls — on the same GPU."""
import types

import numpy as np
import pytest
import torch

from oracle import epilogue_oracle as eo
from test_epilogue_cpu import _groups

pytestmark = pytest.mark.gpu


def _clone_groups(groups, dev):
    return [{"params": [torch.nn.Parameter(g["params"][0].detach().clone().to(dev))], "lr": g["lr"], "name": g["name"]}
            for g in groups]


@pytest.mark.parametrize("P,steps", [(1, 2), (1003, 5), (70001, 3)])
def test_fused_adam_vs_oracle_and_torch_adam(P, steps):
    from train_epilogue import FusedAdam
    gen = torch.Generator().manual_seed(100 + P)
    groups = _groups(P, gen)
    mine = _clone_groups(groups, "cuda")
    theirs = _clone_groups(groups, "cuda")
    opt = FusedAdam(mine, lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(theirs, lr=0.0, eps=1e-15)              # exactly the reference's constructor call
    orc = [dict(p=g["params"][0].detach().numpy().copy()) for g in groups]
    for s in orc:
        s["m"], s["v"] = np.zeros_like(s["p"]), np.zeros_like(s["p"])
    for step in range(1, steps + 1):
        if step == 2:
            for gs in (mine, theirs, groups):
                gs[0]["lr"] = 1.3e-4
        for k, g in enumerate(groups):
            grad = torch.randn(g["params"][0].shape, generator=gen) * (10.0 ** float(torch.randint(-7, 1, (1,), generator=gen)))
            if k == 3 and step == 2:
                grad.zero_()                                       # invisible-everywhere step: zero gradient
            mine[k]["params"][0].grad = grad.cuda()
            theirs[k]["params"][0].grad = grad.cuda()
            eo.adam_step(orc[k]["p"], grad.numpy(), orc[k]["m"], orc[k]["v"], step, groups[k]["lr"])
        opt.step()
        ref.step()
    torch.cuda.synchronize()
    for k, g in enumerate(groups):
        p, q = mine[k]["params"][0], theirs[k]["params"][0]
        st, sq = opt.state[p], ref.state[q]
        assert float(st["step"]) == steps
        # vs the oracle: moments bit-for-bit, parameters within an ulp per step (sqrt / divide roundings)
        np.testing.assert_array_equal(st["exp_avg"].cpu().numpy(), orc[k]["m"], err_msg=g["name"])
        np.testing.assert_array_equal(st["exp_avg_sq"].cpu().numpy(), orc[k]["v"], err_msg=g["name"])
        np.testing.assert_allclose(p.detach().cpu().numpy(), orc[k]["p"], rtol=2e-7, atol=2e-7, err_msg=g["name"])
        # vs torch.optim.Adam on this GPU
        scale = lambda t: max(t.abs().max().item(), 1e-30)
        assert (st["exp_avg"] - sq["exp_avg"]).abs().max().item() <= 1e-6 * scale(sq["exp_avg"]), g["name"]
        assert (st["exp_avg_sq"] - sq["exp_avg_sq"]).abs().max().item() <= 1e-6 * scale(sq["exp_avg_sq"]), g["name"]
        assert (p - q).abs().max().item() <= 1e-6 * scale(q), g["name"]


def test_fused_adam_survives_the_reference_state_surgery():
    """densification re-keys and concatenates optimizer state (gaussian_model.py:419-476): FusedAdam must keep
    working on tensors produced that way, and params without a gradient are skipped (occ_multiplier / dc_delta)."""
    from train_epilogue import FusedAdam
    gen = torch.Generator().manual_seed(5)
    groups = _clone_groups(_groups(50, gen), "cuda")
    frozen = torch.nn.Parameter(torch.ones(50, 7, device="cuda"))
    groups.append({"params": [frozen], "lr": 0, "name": "occ_multiplier"})
    opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    ref_groups = _clone_groups(groups[:6], "cuda")
    ref = torch.optim.Adam(ref_groups, lr=0.0, eps=1e-15)

    def grads(n):
        for k in range(6):
            g = torch.randn(groups[k]["params"][0].shape, generator=gen).cuda() * 1e-2
            groups[k]["params"][0].grad = g
            ref_groups[k]["params"][0].grad = g.clone()
    grads(50)
    opt.step(); ref.step()
    assert frozen not in opt.state and torch.equal(frozen, torch.ones_like(frozen))
    for o, gs in ((opt, groups), (ref, ref_groups)):                 # cat_tensors_to_optimizer, restated
        for g in gs[:6]:
            old = g["params"][0]
            ext = torch.zeros(13, *old.shape[1:], device="cuda")
            st = o.state.pop(old)
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            g["params"][0] = torch.nn.Parameter(torch.cat((old.detach(), ext + 0.5), dim=0))
            o.state[g["params"][0]] = st
    grads(63)
    opt.step(); ref.step()
    for k in range(6):
        p, q = groups[k]["params"][0], ref_groups[k]["params"][0]
        assert p.shape[0] == 63 and float(opt.state[p]["step"]) == 2
        assert (p - q).abs().max().item() <= 1e-6 * q.abs().max().item()


def _stats_model(P, L, gen, dev):
    m = types.SimpleNamespace(reso_lvls=L)
    m.xyz_gradient_accum = torch.rand(P, L, 1, generator=gen)
    m.denom = torch.randint(0, 5, (P, L, 1), generator=gen).float()
    m.max_radii2D = torch.randint(0, 30, (P,), generator=gen).float()
    m.max_pixel_sizes = torch.where(torch.rand(P, generator=gen) < 0.3, torch.full((P,), -1.0), torch.rand(P, generator=gen) * 8)
    m.min_pixel_sizes = torch.where(torch.rand(P, generator=gen) < 0.5, torch.full((P,), -1.0), torch.rand(P, generator=gen) * 3)
    m.base_gaussian_mask = torch.rand(P, generator=gen) < 0.2
    m.target_reso_lvl = torch.randint(0, L, (P,), generator=gen)
    cpu = {k: (v.numpy().copy() if torch.is_tensor(v) else v) for k, v in vars(m).items()}
    for k, v in list(vars(m).items()):
        if torch.is_tensor(v):
            setattr(m, k, v.to(dev).contiguous())
    return m, cpu


@pytest.mark.parametrize("P,L,lvl,flags", [(1, 1, 0, (True, True, True)), (5000, 4, 0, (True, True, True)),
                                            (5000, 4, 2, (False, True, True)), (5000, 4, 3, (True, True, False)),
                                            (100003, 7, 6, (False, False, True)), (100003, 7, 1, (False, True, False))])
def test_training_stats_vs_oracle(P, L, lvl, flags):
    from train_epilogue import update_training_stats
    gen = torch.Generator().manual_seed(P + 10 * lvl)
    model, cpu = _stats_model(P, L, gen, "cuda")
    radii = torch.where(torch.rand(P, generator=gen) < 0.4, torch.zeros(P, dtype=torch.int32),
                        torch.randint(1, 60, (P,), generator=gen, dtype=torch.int32))
    ps = torch.where(torch.rand(P, generator=gen) < 0.2, torch.full((P,), -1.0), torch.rand(P, generator=gen) * 10)
    vsp = torch.zeros(P, 3, device="cuda", requires_grad=True)
    grad = torch.randn(P, 3, generator=gen) * 1e-3
    vsp.grad = grad.cuda()
    update_training_stats(model, vsp, radii.cuda(), ps.cuda(), lvl, base_mask=flags[0], update_pixel_sizes=flags[1],
                          densify=flags[2])
    torch.cuda.synchronize()
    eo.training_stats(radii.numpy(), ps.numpy(), grad.numpy(), cpu["target_reso_lvl"], lvl, L, cpu["xyz_gradient_accum"],
                      cpu["denom"], cpu["max_radii2D"], cpu["max_pixel_sizes"], cpu["min_pixel_sizes"],
                      cpu["base_gaussian_mask"], do_base_mask=flags[0], do_pixel_sizes=flags[1], do_densify=flags[2])
    for k in ("denom", "max_radii2D", "max_pixel_sizes", "min_pixel_sizes", "base_gaussian_mask"):
        np.testing.assert_array_equal(getattr(model, k).cpu().numpy(), cpu[k], err_msg=k)          # bit-exact
    np.testing.assert_allclose(model.xyz_gradient_accum.cpu().numpy(), cpu["xyz_gradient_accum"], rtol=3e-7, atol=0)


def test_training_stats_match_the_reference_torch_sequence():
    """The same updates written as the masked-indexing torch sequence of train.py:239-250 on the GPU."""
    from train_epilogue import update_training_stats
    P, L, lvl = 20000, 4, 1
    gen = torch.Generator().manual_seed(9)
    a, _ = _stats_model(P, L, gen, "cuda")
    b = types.SimpleNamespace(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in vars(a).items()})
    radii = torch.where(torch.rand(P, generator=gen) < 0.4, torch.zeros(P, dtype=torch.int32),
                        torch.randint(1, 60, (P,), generator=gen, dtype=torch.int32)).cuda()
    ps = torch.where(torch.rand(P, generator=gen) < 0.2, torch.full((P,), -1.0), torch.rand(P, generator=gen) * 10).cuda()
    vsp = torch.zeros(P, 3, device="cuda", requires_grad=True)
    vsp.grad = (torch.randn(P, 3, generator=gen) * 1e-3).cuda()
    update_training_stats(a, vsp, radii, ps, lvl, base_mask=True, update_pixel_sizes=True, densify=True)
    vis = radii > 0
    b.base_gaussian_mask = b.base_gaussian_mask | vis
    mask = vis & (b.target_reso_lvl == lvl)
    b.max_pixel_sizes[mask] = torch.max(b.max_pixel_sizes[mask] * 0.95, ps[mask])
    mn = torch.clip(b.min_pixel_sizes[mask] * 1.05, -1)
    b.min_pixel_sizes[mask] = torch.where(ps[mask] > 0, torch.where(mn < 0, ps[mask], torch.min(mn, ps[mask])), mn)
    b.max_radii2D[vis] = torch.max(b.max_radii2D[vis], radii[vis])
    b.xyz_gradient_accum[:, lvl][vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
    b.denom[:, lvl][vis] += 1
    for k in ("denom", "max_radii2D", "max_pixel_sizes", "min_pixel_sizes", "base_gaussian_mask"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.allclose(a.xyz_gradient_accum, b.xyz_gradient_accum, rtol=1e-6, atol=0)


def test_epilogue_argument_errors():
    from diff_gaussian_rasterization import _backend as _C
    import ctypes as C
    assert _C.lib.msgs_adam_step(None, 9, 1, 0.9, 0.999, 1e-15, None) == -1
    assert _C.lib.msgs_adam_step(None, 0, 0, 0.9, 0.999, 1e-15, None) == -1
    assert _C.lib.msgs_adam_step(None, 0, 1, 0.9, 0.999, 1e-15, None) == 0
    d = _C.DensifyStats()
    d.P, d.flags, d.reso_lvl, d.reso_lvls = 4, _C.STATS_DENSIFY, 2, 2
    assert _C.lib.msgs_densify_stats(C.byref(d), None) == -1
    d.reso_lvl = 0
    assert _C.lib.msgs_densify_stats(C.byref(d), None) == -1          # NULL radii


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_training_stats_kernel_vs_reference_golden(case):
    """msgs_densify_stats against the outputs of the REFERENCE's own statistics methods (tests/golden/stats_*.npz,
    generated by make_stats_golden.py from /root/reference/scene/gaussian_model.py:663-704 + train.py:249): bit-exact."""
    import os
    import numpy as np
    from train_epilogue import update_training_stats
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"stats_{case}.npz"))
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    m = types.SimpleNamespace(reso_lvls=int(z["reso_lvls"]), xyz_gradient_accum=t(z["in_xyz_gradient_accum"]),
                              denom=t(z["in_denom"]), max_radii2D=t(z["in_max_radii2D"]),
                              max_pixel_sizes=t(z["in_max_pixel_sizes"]), min_pixel_sizes=t(z["in_min_pixel_sizes"]),
                              base_gaussian_mask=t(z["in_base_mask"]), target_reso_lvl=t(z["target_reso_lvl"]))
    vsp = types.SimpleNamespace(grad=t(z["grad2d"]))
    update_training_stats(m, vsp, t(z["radii"]), t(z["pixel_sizes"]), int(z["reso_lvl"]), base_mask=bool(z["do_base_mask"]),
                          update_pixel_sizes=True, densify=True)
    torch.cuda.synchronize()
    for k, attr in (("xyz_gradient_accum", "xyz_gradient_accum"), ("denom", "denom"), ("max_radii2D", "max_radii2D"),
                    ("max_pixel_sizes", "max_pixel_sizes"), ("min_pixel_sizes", "min_pixel_sizes"),
                    ("base_mask", "base_gaussian_mask")):
        got = getattr(m, attr).cpu().numpy()
        if k == "xyz_gradient_accum":     # one ulp of the gradient norm on ~1e-4 of the entries (torch.norm's CPU kernel)
            assert np.allclose(got, z["out_" + k], rtol=2e-7, atol=0) and (got != z["out_" + k]).mean() < 1e-3, f"{case}: {k}"
        else:
            assert np.array_equal(got, z["out_" + k]), f"{case}: {k}"
