"""Generates tests/golden/getters.npz + getter_graphs.json from the REFERENCE's own GaussianModel getters
(/root/reference/scene/gaussian_model.py:127-153, activations set up at :39-47), evaluated on seeded CPU parameters:
    python tests/golden/make_getter_golden.py        (build container only)
getters.npz: the six raw parameter arrays and the activated outputs of get_scaling / get_rotation / get_opacity / get_features
(get_covariance allocates on "cuda" by name, general_utils.py:102, and cannot run here; its arithmetic is pinned by cov3d.npz).  getter_graphs.json: for each getter the autograd node class names of its output's
graph, depth-first (what diff_gaussian_rasterization._match_reference_getters keys on).  Data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_model_loader as L   # noqa: E402


def chain(t):
    out = []

    def rec(f, d):
        if f is None:
            return
        out.append([d, type(f).__name__])
        for nf, _ in f.next_functions:
            rec(nf, d + 1)
    rec(t.grad_fn, 0)
    return out


def main():
    GM = L.load_gaussian_model()
    m = GM(3)
    P = 257
    g = torch.Generator().manual_seed(11)
    mk = lambda *s, k=1.0: torch.nn.Parameter(k * torch.randn(*s, generator=g))
    m._xyz, m._features_dc, m._features_rest = mk(P, 3), mk(P, 1, 3), mk(P, 15, 3, k=0.2)
    m._opacity, m._scaling, m._rotation = mk(P, 1, k=2.0), mk(P, 3, k=0.7) - 3.0, mk(P, 4)
    m._scaling = torch.nn.Parameter(m._scaling.detach())
    with torch.no_grad():
        m._rotation[5] = 0.0                      # normalize's eps branch
    getters = ("get_features", "get_opacity", "get_scaling", "get_rotation")
    graphs = {name: chain(getattr(m, name)) for name in getters}
    arrays = {n: getattr(m, n).detach().numpy() for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}
    for name in getters:
        arrays[name] = getattr(m, name).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "getters.npz"), **arrays)
    with open(os.path.join(HERE, "getter_graphs.json"), "w") as f:
        json.dump(graphs, f, indent=1, sort_keys=True)
        f.write("\n")
    print({k: v.shape for k, v in arrays.items()})
    print(json.dumps(graphs))


if __name__ == "__main__":
    main()
