"""CPU checks of the train-step epilogue oracle (oracle/epilogue_oracle.py) and of the host logic around the
GPU kernels.  The Adam restatement is pinned against torch.optim.Adam — the implementation the reference itself
calls (/root/reference/scene/gaussian_model.py:248) — running on CPU."""
import numpy as np
import pytest
import torch

from oracle import epilogue_oracle as eo


def _groups(P, gen):
    shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3),
              "rotation": (P, 4)}
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
    return [{"params": [torch.nn.Parameter(torch.randn(*s, generator=gen))], "lr": lrs[n], "name": n}
            for n, s in shapes.items()]


@pytest.mark.parametrize("P,steps", [(7, 1), (301, 6)])
def test_adam_oracle_matches_torch_adam(P, steps):
    gen = torch.Generator().manual_seed(11)
    groups = _groups(P, gen)
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15, foreach=False)
    mine = [{k: g["params"][0].detach().numpy().copy() for k in ("p",)} for g in groups]
    for s in mine:
        s["m"], s["v"] = np.zeros_like(s["p"]), np.zeros_like(s["p"])
    for step in range(1, steps + 1):
        if step == 3:
            groups[0]["lr"] = 1.1e-4              # update_learning_rate changes the xyz group per iteration
        for g, s in zip(groups, mine):
            grad = torch.randn(g["params"][0].shape, generator=gen) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=gen)))
            g["params"][0].grad = grad
            eo.adam_step(s["p"], grad.numpy(), s["m"], s["v"], step, g["lr"])
        opt.step()
    for g, s in zip(groups, mine):
        p = g["params"][0]
        st = opt.state[p]
        # exp_avg / exp_avg_sq bit-for-bit; the parameter within ~1 ulp per step (ATen's vectorised CPU sqrt is not correctly
        # rounded: 0.6 % of its results differ from IEEE sqrt by one ulp)
        np.testing.assert_allclose(s["p"], p.detach().numpy(), rtol=2e-7, atol=2e-7, err_msg=g["name"])
        np.testing.assert_array_equal(s["m"], st["exp_avg"].numpy())
        np.testing.assert_array_equal(s["v"], st["exp_avg_sq"].numpy())


def test_training_stats_known_answers():
    f = np.float32
    radii = np.array([0, 3, 5, 2, 0, 7], dtype=np.int32)
    ps = np.array([9, 2.0, -1.0, 4.0, 1.0, 0.5], dtype=f)
    lvl = np.array([0, 0, 0, 2, 2, 2], dtype=np.int64)
    grad = np.array([[3, 4, 9]] * 6, dtype=f)
    st = dict(xyz_gradient_accum=np.zeros((6, 3, 1), f), denom=np.zeros((6, 3, 1), f), max_radii2D=np.full(6, 4, f),
              max_pixel_sizes=np.array([1, 1, 1, 10, 10, 0.1], f), min_pixel_sizes=np.array([-1, -1, 3, 1, 1, 1], f),
              base_mask=np.zeros(6, bool))
    eo.training_stats(radii, ps, grad, lvl, 0, 3, **st, do_base_mask=True, do_pixel_sizes=True, do_densify=True)
    assert st["base_mask"].tolist() == [False, True, True, True, False, True]
    # level 0: max untouched; min: [1] uninitialised -> 2.0; [2] initialised, invalid pixel size -> 3 * 1.05
    assert st["max_pixel_sizes"].tolist() == [1, 1, 1, 10, 10, f(0.1)]
    np.testing.assert_array_equal(st["min_pixel_sizes"], np.array([-1, 2.0, f(3) * f(1.05), 1, 1, 1], f))
    assert st["max_radii2D"].tolist() == [4, 4, 5, 4, 4, 7]
    assert st["xyz_gradient_accum"][:, 0, 0].tolist() == [0, 5, 5, 5, 0, 5]
    assert st["denom"][:, 0, 0].tolist() == [0, 1, 1, 1, 0, 1] and st["denom"][:, 1:].sum() == 0
    # last level (2): max decays then takes the max; min untouched
    eo.training_stats(radii, ps, grad, lvl, 2, 3, **st, do_base_mask=False, do_pixel_sizes=True, do_densify=False)
    np.testing.assert_array_equal(st["max_pixel_sizes"], np.array([1, 1, 1, f(10) * f(0.95), 10, 0.5], f))
    np.testing.assert_array_equal(st["min_pixel_sizes"], np.array([-1, 2.0, f(3) * f(1.05), 1, 1, 1], f))


def test_fused_adam_host_logic_without_gpu():
    from train_epilogue import FusedAdam, update_training_stats
    gen = torch.Generator().manual_seed(3)
    groups = _groups(5, gen)
    opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    assert [g["name"] for g in opt.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert opt.param_groups[0]["betas"] == (0.9, 0.999) and opt.param_groups[0]["eps"] == 1e-15
    opt.step()                                             # no gradients anywhere: nothing to do, no library call
    assert len(opt.state) == 0
    groups[0]["params"][0].grad = torch.zeros(5, 3)
    with pytest.raises(RuntimeError, match="GPU-only"):
        opt.step()                                         # CPU tensors: loud failure, no fallback
    with pytest.raises(ValueError):
        FusedAdam(_groups(2, gen), betas=(1.5, 0.9))
    import types
    with pytest.raises(RuntimeError, match="GPU"):
        update_training_stats(types.SimpleNamespace(reso_lvls=1), None, torch.zeros(4, dtype=torch.int32),
                              torch.zeros(4))


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_training_stats_pinned_to_the_reference(case):
    """tests/golden/stats_*.npz hold inputs and outputs of the REFERENCE's own update_base_gaussian_mask /
    update_pixel_sizes / add_densification_stats (+ train.py:249) run on CPU tensors (make_stats_golden.py): the numpy
    restatement must reproduce them bit for bit."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"stats_{case}.npz"))
    st = dict(xyz_gradient_accum=z["in_xyz_gradient_accum"].copy(), denom=z["in_denom"].copy(),
              max_radii2D=z["in_max_radii2D"].copy(), max_pixel_sizes=z["in_max_pixel_sizes"].copy(),
              min_pixel_sizes=z["in_min_pixel_sizes"].copy(), base_mask=z["in_base_mask"].copy())
    eo.training_stats(z["radii"], z["pixel_sizes"], z["grad2d"], z["target_reso_lvl"], int(z["reso_lvl"]), int(z["reso_lvls"]),
                      **st, do_base_mask=bool(z["do_base_mask"]), do_pixel_sizes=True, do_densify=True)
    for k in st:
        if k == "xyz_gradient_accum":
            # torch.norm(grad[:, :2], dim=-1) (CPU kernel) and sqrt(gx*gx + gy*gy) differ by one ulp of the norm on
            # about 1 in 1e4 inputs; everything else is bit for bit
            np.testing.assert_allclose(st[k], z["out_" + k], rtol=2e-7, atol=0, err_msg=f"{case}: {k}")
            assert (st[k] != z["out_" + k]).mean() < 1e-3
        else:
            np.testing.assert_array_equal(st[k], z["out_" + k], err_msg=f"{case}: {k}")
