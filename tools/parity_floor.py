"""Full-size gradient parity, separated into its sources (not a test; writes a markdown table).

For a BASELINE config (C2 / C3 / C5) this prints, per gradient tensor, the max-norm relative difference
(||a - b||_inf / ||ref||_inf over the Gaussians without a borderline alpha decision) between
  * the HIP deterministic backward and the float32 oracle            (formulation + per-pair rounding)
  * the HIP atomic backward and the float32 oracle                   (+ float-atomic order)
  * two runs of the HIP atomic backward                              (float-atomic order alone)
  * HIP deterministic and HIP atomic
  * [CPU only, --floor] the float32 oracle and the SAME oracle source compiled with FMA contraction
    (g++ -ffp-contract=fast -mfma): the reference algorithm's own sensitivity to float32 rounding — nvcc
    contracts to FMA by default, so the real CUDA reference is the contracted flavour
  * [--truth] each of them and the float64 autograd evaluation of the same pipeline
    (oracle/torch_oracle.py forward_backward_tiled)
and the borderline-pixel / borderline-Gaussian fractions.

usage: python tools/parity_floor.py C2 [--floor] [--truth] [--no-gpu] [--out file.md]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import scenes
from oracle import oracle_ctypes as oc

LEAVES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")


def leaf_grads_from_oracle(pc_raw, og):
    """oracle gradients w.r.t. the activated inputs pushed through the getters in float64 (as parity_utils.check_backward)"""
    dt = torch.float64
    out = {"_xyz": og["means3D"].to(dt), "_features_dc": og["shs"][:, :1].to(dt), "_features_rest": og["shs"][:, 1:].to(dt)}
    raw = pc_raw["_opacity"].to(dt)
    s = torch.sigmoid(raw)
    out["_opacity"] = og["opacities"].to(dt).view_as(raw) * s * (1 - s)
    out["_scaling"] = og["scales"].to(dt) * torch.exp(pc_raw["_scaling"].to(dt))
    q = pc_raw["_rotation"].to(dt).clone().requires_grad_(True)
    torch.nn.functional.normalize(q).backward(og["rotations"].to(dt))
    out["_rotation"] = q.grad
    out["means2D"] = og["means2D"].to(dt)
    return out


def maxnorm(a, b, ref, rows):
    a, b, ref = a.double().cpu(), b.double().cpu(), ref.double().cpu()
    P = ref.shape[0]
    d = (a.reshape(P, -1) - b.reshape(P, -1)).abs().max(dim=1).values
    scale = max(ref.abs().max().item(), 1e-30)
    dd = d[rows]
    i = int(torch.argmax(torch.where(rows, d, torch.zeros_like(d))).item())
    return (dd.max().item() if dd.numel() else 0.0) / scale, i


def build_fma_oracle():
    so = "/tmp/liboracle_fma.so"
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-ffp-contract=fast", "-mfma", "-fno-fast-math", "-std=c++17", "-fPIC",
                           "-shared", "-o", so, os.path.join(ROOT, "oracle", "msgs_oracle.cpp")])
    return so


def with_oracle_lib(path, fn):
    """run fn() with oracle_ctypes bound to another build of the same source"""
    saved, real = oc._LIB, C.CDLL
    oc._LIB = None
    C.CDLL = lambda p, *a, **k: real(path, *a, **k)
    try:
        oc.lib()
    finally:
        C.CDLL = real
    try:
        return fn()
    finally:
        oc._LIB = saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--floor", action="store_true")
    ap.add_argument("--truth", action="store_true")
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    sc, cam, st = scenes.config(a.config)
    W, H = cam.image_width, cam.image_height
    bg = torch.zeros(3)
    dL = scenes.grad_seed(W, H, {"C2": 1, "C3": 2, "C5": 5}.get(a.config, 9))
    lines = []

    def say(s=""):
        print(s, flush=True)
        lines.append(s)

    grads = {}
    seen = sc
    pc_raw = None
    if not a.no_gpu:
        import diff_gaussian_rasterization as dgr
        from parity_utils import hip_render
        for name, det in (("hip_det", True), ("hip_atomic_a", False), ("hip_atomic_b", False)):
            dgr.set_deterministic(det)
            out, pc, m2 = hip_render(sc, cam, st, bg, dL)
            grads[name] = {k: getattr(pc, k).grad.detach().cpu().clone() for k in LEAVES} | {"means2D": m2.detach().cpu().clone()}
            seen = pc.seen
            pc_raw = {k: getattr(pc, k).detach().cpu() for k in LEAVES}
            hip_out = {k: out[k].detach().cpu() for k in ("render", "radii")}
        dgr.set_deterministic(False)
    else:
        from synthetic_model import SyntheticGaussians
        pc = SyntheticGaussians(sc, "cpu", requires_grad=False)
        pc_raw = {k: getattr(pc, k).detach() for k in LEAVES}

    orc = oc.rasterize(seen, cam, st, bg)
    og = oc.backward(orc, dL)
    grads["oracle_f32"] = leaf_grads_from_oracle(pc_raw, og)
    flagged = orc.borderline_gaussians.clone()
    bl_px = orc.borderline.float().mean().item()
    truth_out = None
    if a.truth:
        from oracle import torch_oracle as to
        truth_out, og64 = to.forward_backward_tiled(seen, cam, st, bg, dL)
        grads["truth_f64"] = leaf_grads_from_oracle(pc_raw, og64)
        flagged |= truth_out["radii"] != orc.radii
    if a.floor:
        so = build_fma_oracle()

        def run():
            r = oc.rasterize(seen, cam, st, bg)
            return r, oc.backward(r, dL), r.borderline_gaussians.clone(), r.radii.clone(), r.color.clone(), r.borderline.clone()
        r2, og2, fl2, radii2, col2, blpx2 = with_oracle_lib(so, run)
        grads["oracle_f32_fma"] = leaf_grads_from_oracle(pc_raw, og2)
        flagged |= fl2 | (radii2 != orc.radii)
        okpx = ~(orc.borderline.bool() | blpx2.bool())
        say(f"oracle vs FMA-contracted oracle: radii differ on {(radii2 != orc.radii).sum().item()} Gaussians; forward max abs diff "
            f"{(col2 - orc.color).abs()[:, okpx].max().item():.3e} on non-borderline pixels, {(col2 - orc.color).abs().max().item():.3e} on all")
    clean = ~flagged
    V = (orc.radii > 0).sum().item()
    say(f"## {a.config}: P={sc.P} {W}x{H}  rendered V={V}  D_ref(rect)={orc.num_instances}  borderline pixels {bl_px:.5%}  "
        f"borderline Gaussians {flagged.float().mean().item():.4%} (excluded below)")
    if not a.no_gpu:
        d = (hip_out["render"] - orc.color).abs()
        okpx = ~orc.borderline.bool()
        say(f"forward: HIP vs oracle max abs diff {d[:, okpx].max().item():.3e} on non-borderline pixels ({d.max().item():.3e} on all); "
            f"radii equal: {torch.equal(hip_out['radii'], orc.radii)}")
    if truth_out is not None:
        okpx = ~(orc.borderline.bool() | truth_out["borderline"])
        say(f"forward vs float64 truth on non-borderline pixels: oracle_f32 {(orc.color.double() - truth_out['color']).abs()[:, okpx].max().item():.3e}"
            + ("" if a.no_gpu else f", HIP {(hip_out['render'].double() - truth_out['color']).abs()[:, okpx].max().item():.3e}")
            + f"; radii differ on {(truth_out['radii'] != orc.radii).sum().item()} Gaussians")
    ref = grads["truth_f64"] if a.truth else grads["oracle_f32"]
    pairs = []
    if not a.no_gpu:
        pairs += [("hip_det", "oracle_f32"), ("hip_atomic_a", "oracle_f32"), ("hip_atomic_a", "hip_atomic_b"), ("hip_det", "hip_atomic_a")]
    if a.floor:
        pairs += [("oracle_f32_fma", "oracle_f32")]
        if not a.no_gpu:
            pairs += [("hip_det", "oracle_f32_fma")]
    if a.truth:
        pairs += [("oracle_f32", "truth_f64")] + ([("oracle_f32_fma", "truth_f64")] if a.floor else [])
        if not a.no_gpu:
            pairs += [("hip_det", "truth_f64"), ("hip_atomic_a", "truth_f64")]
    keys = list(LEAVES) + ["means2D"]
    say("")
    say("| a vs b | " + " | ".join(k.lstrip("_") for k in keys) + " |")
    say("|---|" + "---|" * len(keys))
    worst_rows = {}
    for x, y in pairs:
        cells = []
        for k in keys:
            e, i = maxnorm(grads[x][k], grads[y][k], ref[k], clean)
            cells.append(f"{e:.2e}")
            worst_rows[(x, y, k)] = i
        say(f"| {x} vs {y} | " + " | ".join(cells) + " |")
    # the Gaussians behind the worst scale / rotation numbers
    say("")
    for (x, y, k), i in worst_rows.items():
        if k not in ("_scaling", "_rotation") or y not in ("oracle_f32", "truth_f64") or x == "hip_atomic_b":
            continue
        s = seen.scales[i].tolist()
        say(f"worst {k} {x} vs {y}: Gaussian {i} depth z={seen.means3D[i, 2].item():.3f} scales={[f'{v:.4g}' for v in s]} "
            f"aspect={max(s) / min(s):.1f} opacity={seen.opacities[i].item():.3f} radius={orc.radii[i].item()} "
            f"lvl={int(sc.target_reso_lvl[i])}")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "a") as f:
            f.write("\n".join(lines) + "\n\n")


if __name__ == "__main__":
    main()
