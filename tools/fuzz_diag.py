"""Diagnose one configuration of the randomised parity sweep (tests/fuzz_cases.py) whose FORWARD differs from the oracle on a pixel
no oracle flags: render it under every kernel variant, locate the worst pixel, rebuild that pixel's list from the float64 oracle's
per-Gaussian arrays and find the single entry whose removal (or insertion) reproduces the HIP colour.
usage: python tools/fuzz_diag.py '<cfg as JSON>'"""
import json
import math
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fuzz_cases  # noqa: E402
import scenes  # noqa: E402
from parity_utils import hip_render, small_scene  # noqa: E402


def hip_forward(cfg, sc, cam, st, bg, dL, occlusion=True):
    import diff_gaussian_rasterization as dgr
    lib = dgr._C.lib
    pg, pb = lib.msgs_set_blend_granularity(cfg["gran"]), lib.msgs_set_backward_generation(cfg["bwd_gen"])
    po = lib.msgs_set_occlusion(1 if occlusion else 0)
    pchain = dgr.chain_reference_getters
    dgr.chain_reference_getters = bool(cfg["chain"])
    try:
        smod = float(cfg.get("scale_mod", 1.0))
        if cfg["entry"] == "render":
            out, pc, _ = hip_render(sc, cam, st, bg, dL, scaling_modifier=smod)
            return out, pc.seen, {"scale_modifier": smod}
        use_col, use_cov = cfg["entry"] in ("precomp_col", "precomp_both"), cfg["entry"] in ("precomp_cov", "precomp_both")
        out, _, okw = fuzz_cases._hip_precomp(sc, cam, st, bg, dL, use_col, use_cov, smod)
        return out, sc, dict(okw, scale_modifier=smod)
    finally:
        lib.msgs_set_blend_granularity(pg)
        lib.msgs_set_backward_generation(pb)
        lib.msgs_set_occlusion(po)
        dgr.chain_reference_getters = pchain


def main():
    from oracle import oracle_ctypes as oc
    cfg = json.loads(sys.argv[1])
    P, W, H, seed, ms = cfg["P"], cfg["W"], cfg["H"], cfg["seed"], cfg["ms"]
    sc, cam = small_scene(P, W, H, seed, sh_degree=cfg["deg"], multiscale=ms,
                          **({"scale_k": 0.004 * 1920.0 / max(W, 8) * 0.3} if ms else {}))
    if not cfg.get("sh_full", True):
        sc.shs = sc.shs[:, :(cfg["deg"] + 1) ** 2, :].contiguous()
    sc, cam = fuzz_cases.posed(sc, cam, cfg.get("pose", "front"), cfg.get("focal", 1.0), seed)
    filt = bool(ms and cfg.get("filters", True))
    st = dict(filter_small=filt, filter_large=filt, fade_size=cfg["fade"])
    bg = torch.rand(3, generator=torch.Generator().manual_seed(seed))
    dL = scenes.grad_seed(W, H, seed % 97)
    base_out, seen, okw = hip_forward(cfg, sc, cam, st, bg, dL)
    orc = oc.rasterize(seen, cam, st, bg, **okw)
    tru = oc.rasterize(seen, cam, st, bg, f64=True, **okw)
    okpx = ~(orc.borderline.bool() | tru.borderline.bool())

    def worst(out):
        d = (out["render"].detach().cpu() - orc.color).abs().max(dim=0).values
        d = torch.where(okpx, d, torch.zeros_like(d))
        j = int(torch.argmax(d))
        return d.reshape(-1)[j].item(), j % W, j // W

    print("variant                           worst |HIP - oracle| on unflagged pixels   at (x, y)   radii equal")
    variants = [("as drawn", {})] + [(f"gran={g}", {"gran": g}) for g in (0, 1, 2)] + \
               [("occlusion off", {"_occ": False})]
    for name, kv in variants:
        c = dict(cfg, **{k: v for k, v in kv.items() if not k.startswith("_")})
        out, _, _ = hip_forward(c, sc, cam, st, bg, dL, occlusion=kv.get("_occ", True))
        e, x, y = worst(out)
        print(f"{name:32s}  {e:.3e}   ({x}, {y})   {torch.equal(out['radii'].cpu(), orc.radii)}")

    e, x, y = worst(base_out)
    print(f"\nworst pixel ({x}, {y}): HIP {base_out['render'][:, y, x].tolist()}  oracle {orc.color[:, y, x].tolist()}  "
          f"truth {tru.color[:, y, x].tolist()}")
    f64 = torch.float64
    co = tru._arr("conic_opacity", (P, 4), f64)
    m2 = tru._arr("means2D", (P, 2), f64)
    rgb = tru._arr("rgb", (P, 3), f64)
    dep32 = orc._arr("depths", (P,), torch.float32)
    rects = orc._arr("rects", (P, 4), torch.int32)
    radii = orc.radii
    ncon = orc._arr("n_contrib", (H, W), torch.int32)
    print(f"oracle n_contrib {int(ncon[y, x])}, HIP depth {base_out['depth'][y, x].item():.6f} oracle {orc.depth[y, x].item():.6f}")
    tx, ty = x // 16, y // 16
    inrect = (radii > 0) & (rects[:, 0] <= tx) & (rects[:, 2] > tx) & (rects[:, 1] <= ty) & (rects[:, 3] > ty)
    ids = torch.nonzero(inrect).squeeze(1)
    keys = [(struct.unpack("I", struct.pack("f", float(dep32[i])))[0], int(i)) for i in ids]
    keys.sort()
    entries = []
    T = 1.0
    for _, i in keys:
        dx, dy = m2[i, 0].item() - x, m2[i, 1].item() - y
        power = -0.5 * (co[i, 0].item() * dx * dx + co[i, 2].item() * dy * dy) - co[i, 1].item() * dx * dy
        if power > 0:
            continue
        a = min(0.99, co[i, 3].item() * math.exp(power))
        if a < 1.0 / 255.0:
            entries.append((i, a, None))
            continue
        if T * (1 - a) < 1e-4:
            break
        entries.append((i, a, T))
        T *= 1 - a

    def colour(skip=None, add=None):
        C_, T_ = [0.0, 0.0, 0.0], 1.0
        for i, a, t in entries:
            use = (t is not None and i != skip) or (t is None and i == add)
            if not use:
                continue
            if T_ * (1 - a) < 1e-4:
                break
            for c in range(3):
                C_[c] += rgb[i, c].item() * a * T_
            T_ *= 1 - a
        return [C_[c] + T_ * bg[c].item() for c in range(3)]
    hipc = base_out["render"][:, y, x].tolist()
    err = lambda col: max(abs(col[c] - hipc[c]) for c in range(3))
    print(f"rebuilt list: {sum(t is not None for _, _, t in entries)} blended entries, {sum(t is None for _, _, t in entries)} skipped "
          f"(alpha < 1/255); rebuilt colour vs oracle {max(abs(a - b) for a, b in zip(colour(), orc.color[:, y, x].tolist())):.2e}, vs HIP {err(colour()):.2e}")
    cands = [(err(colour(skip=i)), "without", i, a) for i, a, t in entries if t is not None] + \
            [(err(colour(add=i)), "with skipped", i, a) for i, a, t in entries if t is None]
    cands.sort()
    for e_, what, i, a in cands[:4]:
        A, B, Cc = co[i, 0].item(), co[i, 1].item(), co[i, 2].item()
        tr, det = A + Cc, A * Cc - B * B
        l1 = 0.5 * (tr + math.sqrt(max(tr * tr - 4 * det, 0)))
        l2 = det / l1 if l1 else 0.0
        print(f"  {what} Gaussian {i}: remaining error vs HIP {e_:.2e}; alpha here {a:.6f} (1/255 = {1/255:.6f}), radius {int(radii[i])}, "
              f"centre ({m2[i,0].item():.3f}, {m2[i,1].item():.3f}), conic ({A:.5g}, {B:.5g}, {Cc:.5g}) aspect {math.sqrt(l1 / l2) if l2 > 0 else float('inf'):.1f}, "
              f"opacity {co[i,3].item():.5f}, rect {rects[i].tolist()}, pixel in sub-block ({(x % 16) // 4}, {(y % 16) // 4}) of tile ({tx}, {ty})")


if __name__ == "__main__":
    main()
