"""Float64, differentiable, dense PyTorch restatement of the multi-scale Gaussian rasterizer.

TEST INFRASTRUCTURE ONLY.  Nothing under ms-gs_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.

PARITY UNPINNED: the reference's rasterizer (submodules/diff-gaussian-rasterization ->
https://github.com/JokerYan/MS-GS-rasterizer.git, /root/reference/.gitmodules:4-6) is an
un-vendored, empty submodule with an unrecoverable pinned SHA; the reference holds no tests,
fixtures or golden vectors for this path.  This restatement therefore
follows
  * the call-site contract            /root/reference/gaussian_renderer/__init__.py:18-119
  * quaternion / covariance packing   /root/reference/utils/general_utils.py:64-110,
                                      /root/reference/scene/gaussian_model.py:33-37
  * SH basis and evaluation           /root/reference/utils/sh_utils.py:26-112
  * camera / NDC conventions          /root/reference/utils/graphics_utils.py:38-71,
                                      /root/reference/scene/cameras.py:48-57
  * consumers of the MS outputs       /root/reference/scene/gaussian_model.py:663-686,713-727,
                                      /root/reference/train.py:203-250,269-345
and, for the tile-binned EWA splatting algorithm itself, the published algorithm of
graphdeco-inria/diff-gaussian-rasterization (3DGS, Kerbl et al. 2023) that the MS-GS rasterizer
forks, plus the MS-GS deltas as frozen in DESIGN.md §SPEC (App. A.4).
The pieces that CAN be pinned are pinned by tests/golden (SH colour and camera matrices generated
from the reference's own importable helpers).

Role in the test pyramid: this oracle supplies *autograd* gradients, so that the hand-derived
backward of oracle/msgs_oracle.cpp (float32) and of the HIP kernels is checked against a
mechanically differentiated forward, not against another hand derivation.

The upstream quirks reproduced on purpose:
  Q1 near-plane cull at view-z <= 0.2, no other frustum test
  Q2 t.x/t.z clamp to +-1.3 tanfov; a clamped coordinate is treated as a constant in backward
  Q3 +0.3 on the 2-D covariance diagonal; det == 0 -> skipped
  Q4 radius = ceil(3 sqrt(max eigenvalue)), eigenvalue root floored at 0.1
  Q5 rect from 16x16 tiles, Gaussians blend into every pixel of every tile their rect touches
  Q6 alpha = min(0.99, o G); gradient flows through the clamp as if unclamped
  Q7 alpha < 1/255 -> skipped;  T(1-alpha) < 1e-4 -> pixel terminates, entry NOT blended
  Q8 colour = SH + 0.5, clamped at 0 with zero gradient where clamped
  Q9 p_w = 1 / (p_hom.w + 1e-7); conic backward uses 1/(det^2 + 1e-7)  (the latter is not
     reproduced here: autograd differentiates 1/det exactly; the difference is < 1e-7 relative
     for det >= 0.09 which the +0.3 low-pass guarantees)
  Q10 sort key = (tile, float32 bits of view depth), ties by Gaussian index
"""
import math

import torch

TILE = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh_color(deg, sh, dirs):
    """sh [P,K,3], dirs [P,3] unit -> [P,3].  Same polynomial as utils/sh_utils.py:74-100."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


def quat_to_rot(q):
    """utils/general_utils.py:85-98 WITHOUT the normalisation (the op receives normalised q)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def cov3d_from_scale_rot(scales, rotations, mod):
    """scene/gaussian_model.py:33-37: L = R diag(mod*s); Sigma = L L^T, packed xx,xy,xz,yy,yz,zz."""
    R = quat_to_rot(rotations)
    L = R * (mod * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def depth_key_f32(means3D, viewmatrix):
    """Q10: the float32 view depth exactly as SPEC orders it: ((m2 x + m6 y) + m10 z) + m14."""
    m = viewmatrix.detach().to(torch.float32).reshape(-1)
    p = means3D.detach().to(torch.float32)
    d = ((m[2] * p[:, 0] + m[6] * p[:, 1]) + m[10] * p[:, 2]) + m[14]
    return d


def preprocess(means3D, opacities, view, *, scales=None, rotations=None, cov3D_precomp=None,
               shs=None, colors_precomp=None, max_pixel_sizes=None, min_pixel_sizes=None,
               base_mask=None):
    """Per-Gaussian stage.  `view` is a dict with image_width/height, tanfovx/y, viewmatrix [4,4],
    projmatrix [4,4], campos [3], sh_degree, scale_modifier, filter_small, filter_large, fade_size.
    Returns a dict of per-Gaussian tensors (float64; differentiable where meaningful)."""
    dt = torch.float64
    P = means3D.shape[0]
    W, H = int(view["image_width"]), int(view["image_height"])
    V = view["viewmatrix"].to(dt)
    PM = view["projmatrix"].to(dt)
    campos = view["campos"].to(dt)
    tanx, tany = float(view["tanfovx"]), float(view["tanfovy"])
    mod = float(view.get("scale_modifier", 1.0))
    p = means3D.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([p, ones], dim=1)
    p_view = ph @ V                       # row-vector convention: V = W2C^T (scene/cameras.py:54)
    p_hom = ph @ PM
    depth32 = depth_key_f32(means3D, view["viewmatrix"])
    in_front = depth32 > 0.2              # Q1 (decided on the float32 depth, like the kernels)
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)      # Q9
    p_proj = p_hom[:, :3] * p_w[:, None]

    if cov3D_precomp is not None:
        cov3D = cov3D_precomp.to(dt)
    else:
        cov3D = cov3d_from_scale_rot(scales.to(dt), rotations.to(dt), mod)

    # --- EWA splat (App. A.1 step 4) ---
    tx, ty, tz = p_view[:, 0], p_view[:, 1], p_view[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanx, 1.3 * tany
    txtz, tytz = tx / tz_safe, ty / tz_safe
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx_c = torch.where(cx, (txtz.clamp(-limx, limx) * tz_safe).detach(), tx)   # Q2
    ty_c = torch.where(cy, (tytz.clamp(-limy, limy) * tz_safe).detach(), ty)
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -fx * tx_c / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -fy * ty_c / (tz_safe * tz_safe)], dim=1).view(P, 2, 3)
    Wr = V[:3, :3].transpose(0, 1)        # W2C rotation
    S3 = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2],
                      cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                      cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], dim=1).view(P, 3, 3)
    M = J @ Wr
    cov2 = M @ S3 @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3               # Q3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_safe = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)
    mid = 0.5 * (a + c)
    root = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))      # Q4
    lam = torch.maximum(mid + root, mid - root)
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach().clamp_min(0)))
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def tile_lo(v, g):
        return torch.clamp(torch.trunc((v) / TILE), 0, g)

    pxd, pyd = px.detach(), py.detach()
    rmin_x = tile_lo(pxd - radius, gx)
    rmin_y = tile_lo(pyd - radius, gy)
    rmax_x = tile_lo(pxd + radius + TILE - 1, gx)
    rmax_y = tile_lo(pyd + radius + TILE - 1, gy)
    area = (rmax_x - rmin_x) * (rmax_y - rmin_y)

    # --- MS-GS pixel size + filters ---
    o = opacities.to(dt).reshape(P)
    ell = 2.0 * torch.log(torch.clamp_min(255.0 * o.detach(), 1e-300))
    qx = conic[:, 0].detach()
    qz = conic[:, 2].detach()
    size = torch.where((ell > 0) & ok & (qx > 0) & (qz > 0),
                       torch.minimum(2.0 * torch.sqrt(ell.clamp_min(0) / qx.clamp_min(1e-300)),
                                     2.0 * torch.sqrt(ell.clamp_min(0) / qz.clamp_min(1e-300))),
                       torch.zeros_like(ell))
    pixel_sizes = torch.where(ok, size, torch.zeros_like(size))
    wgt = torch.ones(P, dtype=dt)
    fade = float(view.get("fade_size", 1.0))
    if view.get("filter_small", False) and min_pixel_sizes is not None:
        mn = min_pixel_sizes.to(dt)
        bm = base_mask.to(torch.bool) if base_mask is not None else torch.zeros(P, dtype=torch.bool)
        active = (~bm) & (mn > 0) & (size < mn)
        if fade > 0:
            rel = mn / size.clamp_min(1e-30)
            ws = torch.clamp(1.0 - (rel - 1.0) / fade, 0.0, 1.0)
        else:
            ws = torch.zeros(P, dtype=dt)
        wgt = torch.where(active, wgt * ws, wgt)
    if view.get("filter_large", False) and max_pixel_sizes is not None:
        mx = max_pixel_sizes.to(dt)
        active = (mx > 0) & (size > mx)
        if fade > 0:
            rel = size / mx.clamp_min(1e-30)
            wl = torch.clamp(1.0 - (rel - 1.0) / fade, 0.0, 1.0)
        else:
            wl = torch.zeros(P, dtype=dt)
        wgt = torch.where(active, wgt * wl, wgt)
    visible = ok & (area > 0) & (wgt > 0)
    o_eff = o * wgt.detach()

    # --- colour (Q8) ---
    if colors_precomp is not None:
        rgb = colors_precomp.to(dt)
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        d = p - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        raw = eval_sh_color(int(view["sh_degree"]), shs.to(dt), d) + 0.5
        clamped = raw < 0
        rgb = torch.where(clamped, torch.zeros_like(raw), raw)

    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    return dict(visible=visible, depth32=depth32, depth=tz, px=px, py=py, conic=conic, opacity=o_eff,
                rgb=rgb, clamped=clamped, radii=radii, pixel_sizes=pixel_sizes,
                rect=(rmin_x.long(), rmin_y.long(), rmax_x.long(), rmax_y.long()),
                cov3D=cov3D, cov2=(a, b, c))


def rasterize(means3D, opacities, view, bg, **kw):
    """Full forward.  Returns (color [3,H,W], acc_pixel_size [H,W], depth [H,W], radii [P] i32,
    pixel_sizes [P], aux) with means2D-gradient support through aux['means2D'] (a [P,2] leaf that
    is added, as zeros, to the pixel centres: d L/d means2D is then in *pixel* units; the op's
    NDC-ish unit is pixel-grad * 0.5*W (resp. H), App. A.3)."""
    dt = torch.float64
    pre = preprocess(means3D, opacities, view, **kw)
    W, H = int(view["image_width"]), int(view["image_height"])
    P = means3D.shape[0]
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    means2D = torch.zeros(P, 2, dtype=dt, requires_grad=True)
    px = pre["px"] + means2D[:, 0]
    py = pre["py"] + means2D[:, 1]
    vis = pre["visible"]
    rminx, rminy, rmaxx, rmaxy = pre["rect"]
    bg = bg.to(dt)

    color = torch.zeros(3, H, W, dtype=dt)
    acc_ps = torch.zeros(H, W, dtype=dt)
    dep = torch.zeros(H, W, dtype=dt)
    final_T = torch.ones(H, W, dtype=dt)
    n_blended = torch.zeros(H, W, dtype=torch.int64)
    borderline = torch.zeros(H, W, dtype=torch.bool)
    # sort key (Q10): float32 depth bits, ties by index -> stable argsort on the float32 depth
    order_all = torch.argsort(pre["depth32"], stable=True)
    vis_sorted = order_all[vis[order_all]]
    color_rows = []
    for ty in range(gy):
        row_tiles = []
        for tx in range(gx):
            sel = vis_sorted[(rminx[vis_sorted] <= tx) & (tx < rmaxx[vis_sorted]) &
                             (rminy[vis_sorted] <= ty) & (ty < rmaxy[vis_sorted])]
            x0, y0 = tx * TILE, ty * TILE
            x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
            ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt),
                                    indexing="ij")
            npix = ys.numel()
            xs, ys = xs.reshape(-1), ys.reshape(-1)
            if sel.numel() == 0:
                row_tiles.append((x0, x1, y0, y1, bg[:, None].expand(3, npix).clone(),
                                  torch.zeros(npix, dtype=dt), torch.zeros(npix, dtype=dt),
                                  torch.ones(npix, dtype=dt), torch.zeros(npix, dtype=torch.int64),
                                  torch.zeros(npix, dtype=torch.bool)))
                continue
            dx = px[sel][:, None] - xs[None, :]
            dy = py[sel][:, None] - ys[None, :]
            con = pre["conic"][sel]
            power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
            G = torch.exp(torch.clamp(power, max=0.0))
            a_raw = pre["opacity"][sel][:, None] * G
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()          # Q6
            valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)                   # Q7
            near = (power.detach() <= 0) & ((alpha.detach() - 1.0 / 255.0).abs() < 2e-6)
            alpha_v = torch.where(valid, alpha, torch.zeros_like(alpha))
            one_m = 1.0 - alpha_v
            T_after = torch.cumprod(one_m, dim=0)
            T_before = torch.cat([torch.ones(1, npix, dtype=dt), T_after[:-1]], dim=0)
            fail = valid & (T_after.detach() < 1e-4)
            near_t = valid & ((T_after.detach() - 1e-4).abs() < 1e-9)
            any_fail = fail.any(dim=0)
            first_fail = torch.where(any_fail, fail.to(torch.int64).argmax(dim=0),
                                     torch.full((npix,), sel.numel(), dtype=torch.int64))
            idx = torch.arange(sel.numel())[:, None]
            blended = valid & (idx < first_fail[None, :])
            # borderline decisions only matter before termination
            bl = ((near | near_t) & (idx <= first_fail[None, :])).any(dim=0)
            wgt = torch.where(blended, alpha * T_before, torch.zeros_like(alpha))
            Cc = (wgt[:, None, :] * pre["rgb"][sel][:, :, None]).sum(dim=0)
            Tfin = torch.where(blended, one_m, torch.ones_like(one_m)).prod(dim=0)
            tile_color = Cc + Tfin[None, :] * bg[:, None]
            tile_ps = (wgt.detach() * pre["pixel_sizes"][sel][:, None]).sum(dim=0)
            tile_dep = (wgt.detach() * pre["depth"][sel].detach()[:, None]).sum(dim=0)
            row_tiles.append((x0, x1, y0, y1, tile_color, tile_ps, tile_dep, Tfin.detach(),
                              blended.sum(dim=0), bl))
        color_rows.append(row_tiles)
    # assemble (differentiable wrt color)
    rows = []
    for row_tiles in color_rows:
        parts = []
        for (x0, x1, y0, y1, tc, tps, tdp, tT, nb, bl) in row_tiles:
            parts.append(tc.view(3, y1 - y0, x1 - x0))
            acc_ps[y0:y1, x0:x1] = tps.view(y1 - y0, x1 - x0)
            dep[y0:y1, x0:x1] = tdp.view(y1 - y0, x1 - x0)
            final_T[y0:y1, x0:x1] = tT.view(y1 - y0, x1 - x0)
            n_blended[y0:y1, x0:x1] = nb.view(y1 - y0, x1 - x0)
            borderline[y0:y1, x0:x1] = bl.view(y1 - y0, x1 - x0)
        rows.append(torch.cat(parts, dim=2))
    color = torch.cat(rows, dim=1)
    aux = dict(pre=pre, means2D=means2D, final_T=final_T, n_blended=n_blended, borderline=borderline)
    return color, acc_ps, dep, pre["radii"], pre["pixel_sizes"], aux


def view_dict(cam, *, sh_degree, scale_modifier=1.0, filter_small=False, filter_large=False,
              fade_size=1.0):
    # tan(FoV / 2) rounded to float32: what the op receives (GaussianRasterizationSettings -> msgs_view_t::tanfovx is a float)
    f32 = lambda v: float(torch.tensor(v, dtype=torch.float64).to(torch.float32))
    return dict(image_width=cam.image_width, image_height=cam.image_height,
                tanfovx=f32(math.tan(cam.FoVx * 0.5)), tanfovy=f32(math.tan(cam.FoVy * 0.5)),
                viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                campos=cam.camera_center, sh_degree=sh_degree, scale_modifier=scale_modifier,
                filter_small=filter_small, filter_large=filter_large, fade_size=fade_size)


def forward_backward(scene, cam, settings, bg, dL_dcolor, use_cov_precomp=False, use_colors_precomp=False):
    """Convenience: float64 forward + autograd backward for a scenes.Scene.  Returns (outputs, grads)
    with grads in the op's units (means2D grad scaled to NDC-ish units, App. A.3)."""
    dt = torch.float64
    leaf = lambda t: t.detach().to(dt).clone().requires_grad_(True)
    means3D = leaf(scene.means3D)
    opac = leaf(scene.opacities)
    view = view_dict(cam, sh_degree=scene.sh_degree, **settings)
    kw = dict(max_pixel_sizes=scene.max_pixel_sizes, min_pixel_sizes=scene.min_pixel_sizes,
              base_mask=scene.base_mask)
    leaves = dict(means3D=means3D, opacities=opac)
    if use_cov_precomp:
        cov = cov3d_from_scale_rot(scene.scales.to(dt), scene.rotations.to(dt),
                                   float(settings.get("scale_modifier", 1.0)))
        cov = leaf(cov)
        kw["cov3D_precomp"] = cov
        leaves["cov3D_precomp"] = cov
    else:
        sc, ro = leaf(scene.scales), leaf(scene.rotations)
        kw["scales"], kw["rotations"] = sc, ro
        leaves["scales"], leaves["rotations"] = sc, ro
    if use_colors_precomp:
        d = scene.means3D.to(dt) - cam.camera_center.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp_min(eval_sh_color(scene.sh_degree, scene.shs.to(dt), d) + 0.5, 0.0)
        col = leaf(col)
        kw["colors_precomp"] = col
        leaves["colors_precomp"] = col
    else:
        sh = leaf(scene.shs)
        kw["shs"] = sh
        leaves["shs"] = sh
    color, acc_ps, dep, radii, psz, aux = rasterize(means3D, opac, view, bg, **kw)
    loss = (color * dL_dcolor.to(dt)).sum()
    loss.backward()
    W, H = cam.image_width, cam.image_height
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    g2 = aux["means2D"].grad
    if g2 is None:
        g2 = torch.zeros(scene.P, 2, dtype=dt)
    m2 = torch.zeros(scene.P, 3, dtype=dt)
    m2[:, 0] = g2[:, 0] * 0.5 * W
    m2[:, 1] = g2[:, 1] * 0.5 * H
    grads["means2D"] = m2
    outs = dict(color=color.detach(), acc_pixel_size=acc_ps, depth=dep, radii=radii,
                pixel_sizes=psz.detach(), final_T=aux["final_T"], borderline=aux["borderline"],
                n_blended=aux["n_blended"])
    return outs, grads


def forward_backward_tiled(scene, cam, settings, bg, dL_dcolor, progress=None):
    """The same float64 forward + autograd backward as forward_backward() (scales / rotations / SH inputs), organised so
    that it scales to the BASELINE configurations (1e5 .. 1e6 Gaussians): the per-Gaussian stage keeps its autograd
    graph ([P]-sized tensors only), every tile is blended and differentiated on its own (torch.autograd.grad w.r.t. the
    tile's gathered 2-D quantities, scatter-added into [P]-sized accumulators), and ONE backward through the
    per-Gaussian stage finishes the chain.  Mathematically identical to forward_backward(); this is the float64
    "truth" the full-size float32 results (CPU oracle, HIP kernels) are measured against in tools/parity_floor.py.
    Returns (outputs, grads) like forward_backward()."""
    dt = torch.float64
    leaf = lambda t: t.detach().to(dt).clone().requires_grad_(True)
    means3D, opac = leaf(scene.means3D), leaf(scene.opacities)
    sc, ro, sh = leaf(scene.scales), leaf(scene.rotations), leaf(scene.shs)
    view = view_dict(cam, sh_degree=scene.sh_degree, **settings)
    pre = preprocess(means3D, opac, view, scales=sc, rotations=ro, shs=sh, max_pixel_sizes=scene.max_pixel_sizes,
                     min_pixel_sizes=scene.min_pixel_sizes, base_mask=scene.base_mask)
    W, H = int(cam.image_width), int(cam.image_height)
    P = scene.P
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    bg = bg.to(dt)
    dL = dL_dcolor.to(dt)
    d_px, d_py = pre["px"].detach(), pre["py"].detach()
    d_con, d_op, d_rgb = pre["conic"].detach(), pre["opacity"].detach(), pre["rgb"].detach()
    g_px, g_py = torch.zeros(P, dtype=dt), torch.zeros(P, dtype=dt)
    g_con, g_op, g_rgb = torch.zeros(P, 3, dtype=dt), torch.zeros(P, dtype=dt), torch.zeros(P, 3, dtype=dt)
    color = torch.zeros(3, H, W, dtype=dt)
    borderline = torch.zeros(H, W, dtype=torch.bool)
    vis = pre["visible"]
    rminx, rminy, rmaxx, rmaxy = pre["rect"]
    order_all = torch.argsort(pre["depth32"], stable=True)           # Q10
    vis_sorted = order_all[vis[order_all]]
    rx0, rx1, ry0, ry1 = rminx[vis_sorted], rmaxx[vis_sorted], rminy[vis_sorted], rmaxy[vis_sorted]
    for ty in range(gy):
        in_row = (ry0 <= ty) & (ty < ry1)
        row_ids, row_x0, row_x1 = vis_sorted[in_row], rx0[in_row], rx1[in_row]
        for tx in range(gx):
            sel = row_ids[(row_x0 <= tx) & (tx < row_x1)]
            x0, y0 = tx * TILE, ty * TILE
            x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
            if sel.numel() == 0:
                color[:, y0:y1, x0:x1] = bg[:, None, None]
                continue
            ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
            npix = ys.numel()
            xs, ys = xs.reshape(-1), ys.reshape(-1)
            t_px, t_py = d_px[sel].requires_grad_(True), d_py[sel].requires_grad_(True)
            t_con, t_op, t_rgb = d_con[sel].requires_grad_(True), d_op[sel].requires_grad_(True), d_rgb[sel].requires_grad_(True)
            dx = t_px[:, None] - xs[None, :]
            dy = t_py[:, None] - ys[None, :]
            power = -0.5 * (t_con[:, 0:1] * dx * dx + t_con[:, 2:3] * dy * dy) - t_con[:, 1:2] * dx * dy
            G = torch.exp(torch.clamp(power, max=0.0))
            a_raw = t_op[:, None] * G
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()          # Q6
            valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)                   # Q7
            near = (power.detach() <= 0) & ((alpha.detach() - 1.0 / 255.0).abs() < 2e-6)
            alpha_v = torch.where(valid, alpha, torch.zeros_like(alpha))
            one_m = 1.0 - alpha_v
            T_after = torch.cumprod(one_m, dim=0)
            T_before = torch.cat([torch.ones(1, npix, dtype=dt), T_after[:-1]], dim=0)
            fail = valid & (T_after.detach() < 1e-4)
            near_t = valid & ((T_after.detach() - 1e-4).abs() < 1e-9)
            any_fail = fail.any(dim=0)
            first_fail = torch.where(any_fail, fail.to(torch.int64).argmax(dim=0),
                                     torch.full((npix,), sel.numel(), dtype=torch.int64))
            idx = torch.arange(sel.numel())[:, None]
            blended = valid & (idx < first_fail[None, :])
            bl = ((near | near_t) & (idx <= first_fail[None, :])).any(dim=0)
            wgt = torch.where(blended, alpha * T_before, torch.zeros_like(alpha))
            Cc = (wgt[:, None, :] * t_rgb[:, :, None]).sum(dim=0)
            Tfin = torch.where(blended, one_m, torch.ones_like(one_m)).prod(dim=0)
            tile_color = Cc + Tfin[None, :] * bg[:, None]
            loss = (tile_color * dL[:, y0:y1, x0:x1].reshape(3, npix)).sum()
            gs = torch.autograd.grad(loss, [t_px, t_py, t_con, t_op, t_rgb], allow_unused=True)
            for acc, g in zip((g_px, g_py, g_con, g_op, g_rgb), gs):
                if g is not None:
                    acc.index_add_(0, sel, g)
            color[:, y0:y1, x0:x1] = tile_color.detach().view(3, y1 - y0, x1 - x0)
            borderline[y0:y1, x0:x1] = bl.view(y1 - y0, x1 - x0)
        if progress is not None:
            progress(ty + 1, gy)
    torch.autograd.backward([pre["px"], pre["py"], pre["conic"], pre["opacity"], pre["rgb"]],
                            [g_px, g_py, g_con, g_op, g_rgb])
    zeros = lambda v: v.grad if v.grad is not None else torch.zeros_like(v)
    grads = dict(means3D=zeros(means3D), opacities=zeros(opac), scales=zeros(sc), rotations=zeros(ro), shs=zeros(sh))
    m2 = torch.zeros(P, 3, dtype=dt)
    m2[:, 0] = g_px * 0.5 * W
    m2[:, 1] = g_py * 0.5 * H
    grads["means2D"] = m2
    outs = dict(color=color, radii=pre["radii"], pixel_sizes=pre["pixel_sizes"].detach(), borderline=borderline)
    return outs, grads
