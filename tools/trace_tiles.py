"""Occupancy timeline of the one-wave-per-tile blend backward at C3 (tool build with -DMSGS_TRACE_TILES, tools/trace_tiles.sh):
per tile start / end timestamps and the hardware slot it ran on.  Prints the running-wave curve, per-SIMD busy spans and the
makespan decomposition; saves the raw trace."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians
out_dir = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
sc, cam, st = scenes.config(cfg)
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
camd = cam.to("cuda"); bg = torch.zeros(3, device="cuda")
dL = scenes.grad_seed(cam.image_width, cam.image_height, 5).to("cuda")
tiles = ((cam.image_width + 15) // 16) * ((cam.image_height + 15) // 16)
trace = torch.zeros(4 * tiles, dtype=torch.int64, device="cuda")
f = dgr._C.lib.msgs_debug_set_tile_trace
f.restype = C.c_int; f.argtypes = [C.c_void_p]
assert f(C.c_void_p(trace.data_ptr())) == 0
torch.cuda.synchronize()
for it in range(4):
    for p_ in pc.parameters(): p_.grad = None
    out = render(camd, pc, PIPE, bg, **st); out["render"].backward(dL)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(tiles, 4)
np.save(os.path.join(out_dir, "tile_trace.npy"), t)
t0, t1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0        # us (100 MHz)
hw = t[:, 2] & 0xFFFFFFFF; xcc = t[:, 2] >> 32
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
slot = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
work = t[:, 3] & 0xFFFFFFFF
print(f"tiles {tiles}: makespan {t1.max():.1f} us, mean tile {np.mean(t1 - t0):.1f} us (min {np.min(t1 - t0):.1f}, max {np.max(t1 - t0):.1f})")
print("last start %.1f us; tiles started at t<5us: %d" % (t0.max(), int((t0 < 5).sum())))
for q in np.arange(0, t1.max() + 20, 20):
    print("t=%4d us running %5d" % (q, int(((t0 <= q) & (t1 > q)).sum())))
u = np.unique(slot)
busy_end = np.array([t1[slot == s_].max() for s_ in u]); nt = np.array([(slot == s_).sum() for s_ in u])
wsum = np.array([work[slot == s_].sum() for s_ in u])
print(f"distinct SIMD slots {len(u)}; tiles per SIMD min/mean/max {nt.min()}/{nt.mean():.2f}/{nt.max()}")
print("per-SIMD last finish: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(busy_end, [10, 50, 90, 100])))
print("per-SIMD traversal-length sum: min %d mean %.0f max %d (max/mean %.2f)" % (wsum.min(), wsum.mean(), wsum.max(), wsum.max() / wsum.mean()))
print("corr(per-SIMD work sum, last finish) = %.3f" % np.corrcoef(wsum, busy_end)[0, 1])
print("per-XCC last finish:", {int(x): round(float(t1[xcc == x].max()), 1) for x in np.unique(xcc)})
print("per-XCC work:", {int(x): int(work[xcc == x].sum()) for x in np.unique(xcc)})
