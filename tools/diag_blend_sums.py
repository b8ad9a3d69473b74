"""Where does a gradient difference come from?  For one fuzz configuration: the per-Gaussian 2-D sums the HIP blend backward leaves
in its gradient records (read back from the caller-owned scratch buffer of a direct msgs_backward call), converted to the textbook
sums, against the float32 oracle's and the float64 truth's (oracle_ctypes.backward(..., want_sums2d=True)); then the worst
Gaussian of dL/dscaling with its footprint.  usage: diag_blend_sums.py P W H deg ms fade gran bwd_gen fwd_var seed"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from oracle import oracle_ctypes as oc
from parity_utils import hip_render, leaf_space, small_scene
P, W, H, deg = (int(a) for a in sys.argv[1:5])
ms = sys.argv[5] in ("1", "True")
fade = float(sys.argv[6])
gran, bwd_gen, fwd_var, seed = (int(a) for a in sys.argv[7:11])
sc, cam = small_scene(P, W, H, seed, sh_degree=deg, multiscale=ms, **({"scale_k": 0.004 * 1920.0 / max(W, 8) * 0.3} if ms else {}))
st = dict(filter_small=ms, filter_large=ms, fade_size=fade)
bg = torch.rand(3, generator=torch.Generator().manual_seed(seed))
dL = scenes.grad_seed(W, H, seed % 97)
lib = dgr._C.lib
lib.msgs_set_blend_granularity(gran); lib.msgs_set_backward_generation(bwd_gen)
out, pc, m2 = hip_render(sc, cam, st, bg, dL)
ctx = out["render"].grad_fn
call = ctx.call
geom, binning, image, D = ctx.state
dev = call.device
# a direct msgs_backward with our own scratch: the records stay readable afterwards
scratch = torch.zeros(int(lib.msgs_backward_scratch_bytes(P)), dtype=torch.uint8, device=dev)
mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
g = dict(x=mk(P, 3), m2=mk(P, 3), op=mk(P), sc=mk(P, 3), ro=mk(P, 4), dc=mk(P, 1, 3), rest=mk(P, 15, 3))
ptr = lambda t: C.c_void_p(t.data_ptr())
grads = dgr._C.Grads(ptr(g["x"]), ptr(g["m2"]), None, None, ptr(g["op"]), ptr(g["sc"]), ptr(g["ro"]), None, ptr(g["dc"]), ptr(g["rest"]),
                     None, 0, 0, None, None)
dLd = dL.to(dev).contiguous()
dgr._C.check(lib.msgs_backward(call.view_ref, call.g_ref, ptr(ctx.radii), ptr(geom), geom.numel(), D, ptr(binning), binning.numel(),
                               ptr(image), image.numel(), ptr(dLd), ptr(scratch), scratch.numel(), C.byref(grads), None,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "msgs_backward")
torch.cuda.synchronize()
rec = scratch[:80 * P].view(torch.float64).view(P, 10).cpu()          # [sum q dx, q dy, q dx2, q dxdy, q dy2, q, drgb[3], pad]
orc = oc.rasterize(pc.seen, cam, st, bg); og = oc.backward(orc, dL, want_sums2d=True)
tru = oc.rasterize(pc.seen, cam, st, bg, f64=True); tg = oc.backward(tru, dL, want_sums2d=True)
# textbook sums from the records (preprocess.hip): dL/dconic = -0.5 * sums, dL/dopacity_eff = sum q / o_eff
con_o = orc._arr("conic_opacity", (P, 4), torch.float32).double()
hip = torch.zeros(P, 9, dtype=torch.float64)
hip[:, 2:5] = -0.5 * rec[:, 2:5]
hip[:, 5] = rec[:, 5] / con_o[:, 3].clamp_min(1e-30)
hip[:, 6:9] = rec[:, 6:9]
names = ["mean2D.x", "mean2D.y", "conic A", "conic B", "conic C", "opacity", "r", "g", "b"]
vis = orc.radii > 0
clean = vis & ~(orc.borderline_gaussians | tru.borderline_gaussians)
print(f"visible {int(vis.sum())}, clean {int(clean.sum())}")
for k in range(2, 9):
    t = tg["sums2d"][:, k]; s = t.abs().max().clamp_min(1e-300)
    eh = ((hip[:, k] - t).abs()[clean].max() / s).item(); eo = ((og["sums2d"][:, k] - t).abs()[clean].max() / s).item()
    print(f"sum {names[k]:9s}: HIP vs truth {eh:.3e}   oracle_f32 vs truth {eo:.3e}")
pairs_t, pairs_o = leaf_space(pc, m2, tg), leaf_space(pc, m2, og)
for key in ("scaling", "rotation"):
    got, truth = pairs_t[key]; orc_leaf = pairs_o[key][1]
    d = (got.detach().cpu().double() - truth).abs().reshape(P, -1).max(dim=1).values
    d[~clean] = 0
    i = int(d.argmax()); s = truth.abs().max().item()
    do = (orc_leaf - truth).abs().reshape(P, -1).max(dim=1).values[i].item()
    print(f"{key}: worst Gaussian {i}: HIP {d[i].item() / s:.3e} oracle {do / s:.3e} of the tensor max; radius {int(orc.radii[i])} px, "
          f"conic {con_o[i, :3].tolist()}, opacity {con_o[i, 3].item():.4f}, scales {pc.seen.scales[i].tolist()}")
    for k in range(2, 6):
        t = tg["sums2d"][i, k].item()
        print(f"      {names[k]:8s} truth {t:+.9e}  HIP {hip[i, k].item():+.9e} ({abs(hip[i, k].item() - t) / max(abs(t), 1e-300):.2e})  "
              f"oracle {og['sums2d'][i, k].item():+.9e} ({abs(og['sums2d'][i, k].item() - t) / max(abs(t), 1e-300):.2e})")
# does the worst Gaussian share a pixel with a borderline decision of ANOTHER Gaussian?  (the oracle flags the Gaussian whose
# own alpha sits at 1/255, and the pixel; a flip there changes T for everything behind it at that pixel and the colour behind for
# everything in front)
m2d = orc._arr("means2D", (P, 2), torch.float32).double()
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
for key in ("scaling", "rotation"):
    got, truth = pairs_t[key]
    d = (got.detach().cpu().double() - truth).abs().reshape(P, -1).max(dim=1).values
    d[~clean] = 0
    i = int(d.argmax())
    dx, dy = m2d[i, 0] - xs, m2d[i, 1] - ys
    power = -0.5 * (con_o[i, 0] * dx * dx + con_o[i, 2] * dy * dy) - con_o[i, 1] * dx * dy
    alpha = torch.clamp(con_o[i, 3] * torch.exp(power), max=0.99)
    foot = (power <= 0) & (alpha >= 1.0 / 255.0)
    bl = orc.borderline.bool() | tru.borderline.bool()
    print(f"{key}: Gaussian {i} reaches {int(foot.sum())} pixels with alpha >= 1/255, {int((foot & bl).sum())} of them carry a borderline "
          f"decision of some Gaussian (HIP forward vs oracle on those: max |dcolor| "
          f"{(out['render'].detach().cpu() - orc.color).abs()[:, foot & bl].max().item() if (foot & bl).any() else 0.0:.3e})")
