"""MS-GS model files: the PLY layout GaussianModel.save_ply / load_ply use
(/root/reference/scene/gaussian_model.py:293-344,358-417) and the checkpoint tuple of capture / restore (:79-125),
without the plyfile dependency (binary_little_endian PLY read and written with numpy structured arrays).

PLY vertex properties, in order: x y z nx ny nz | f_dc_0..2 | f_rest_0..44 | opacity | occ_multiplier_0..3 |
dc_delta_{0..3}_{0..2} | scale_0..2 | rot_0..3 | base_gaussian_mask (uchar; the reference passes a numpy bool, which
plyfile stores as uchar) | max_pixel_sizes | min_pixel_sizes; everything else float32.  SH coefficients are stored
channel-major (transpose(1, 2) of the [P, K, 3] parameter), as the reference does.

capture()/restore(): the reference's restore() unpacks (max_radii2D, min_pixel_sizes, base_gaussian_mask,
max_pixel_sizes) where capture() packed (max_radii2D, base_gaussian_mask, max_pixel_sizes, min_pixel_sizes)
(gaussian_model.py:89-92 vs :110-113) — restoring one of its own checkpoints permutes three tensors.  Here restore()
unpacks in capture() order, so checkpoints written by the reference load correctly.
"""
import os

import numpy as np
import torch
import torch.nn as nn


def attribute_names(n_dc=3, n_rest=45, n_lvl_occ=4, n_lvl_dc=4, n_scale=3, n_rot=4):
    """gaussian_model.py:293-313"""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
    names += [f"occ_multiplier_{i}" for i in range(n_lvl_occ)]
    names += [f"dc_delta_{i}_{j}" for i in range(n_lvl_dc) for j in range(3)]
    names += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names + ["base_gaussian_mask", "max_pixel_sizes", "min_pixel_sizes"]


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "<i2", "int16": "<i2",
              "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4",
              "float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8"}


def write_ply(path, columns):
    """columns: ordered {name: 1-D numpy array}; bool/uint8 -> uchar, everything else float32."""
    names = list(columns)
    n = len(next(iter(columns.values()))) if names else 0
    dt = np.dtype([(k, "u1" if columns[k].dtype in (np.bool_, np.uint8) else "<f4") for k in names])
    rec = np.empty(n, dtype=dt)
    for k in names:
        rec[k] = columns[k]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property {'uchar' if dt[k] == np.uint8 else 'float'} {k}" for k in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_ply(path):
    """First element of a binary_little_endian (or ascii) PLY as a numpy structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen_elements = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen_elements += 1
                in_first = seen_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None:
            raise ValueError(f"{path}: no element in header")
        dt = np.dtype(props)
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        elif fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            data = np.empty(count, dtype=dt)
            for j, (k, _) in enumerate(props):
                data[k] = rows[:, j]
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return data


def save_ply(model, path):
    """gaussian_model.py:315-344 on a model with the reference's attribute names."""
    cpu = lambda t: t.detach().cpu().numpy()
    P = model._xyz.shape[0]
    xyz = cpu(model._xyz)
    blocks = [xyz, np.zeros_like(xyz),
              cpu(model._features_dc.detach().transpose(1, 2).flatten(start_dim=1)),
              cpu(model._features_rest.detach().transpose(1, 2).flatten(start_dim=1)),
              cpu(model._opacity).reshape(P, -1), cpu(model._occ_multiplier.flatten(start_dim=1)),
              cpu(model._dc_delta.flatten(start_dim=1)), cpu(model._scaling), cpu(model._rotation)]
    flat = np.concatenate(blocks, axis=1).astype(np.float32)
    names = attribute_names(model._features_dc.shape[1] * model._features_dc.shape[2],
                            model._features_rest.shape[1] * model._features_rest.shape[2],
                            model._occ_multiplier.shape[1], model._dc_delta.shape[1] // 3,
                            model._scaling.shape[1], model._rotation.shape[1])
    cols = {k: flat[:, j] for j, k in enumerate(names[:-3])}
    cols["base_gaussian_mask"] = cpu(model.base_gaussian_mask).astype(np.bool_)
    cols["max_pixel_sizes"] = cpu(model.max_pixel_sizes).astype(np.float32)
    cols["min_pixel_sizes"] = cpu(model.min_pixel_sizes).astype(np.float32)
    write_ply(path, cols)


def load_ply(model, path, device="cuda", max_sh_degree=3, n_lvl_occ=4, n_lvl_dc=4, multi_occ=False, multi_dc=False):
    """gaussian_model.py:358-417: fills the reference's attribute names on `model`."""
    v = read_ply(path)
    names = v.dtype.names
    col = lambda k: np.asarray(v[k], dtype=np.float32)
    by_index = lambda prefix: sorted((k for k in names if k.startswith(prefix)), key=lambda k: int(k.split("_")[-1]))
    P = len(v)
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    f_dc = np.stack([col(f"f_dc_{i}") for i in range(3)], axis=1).reshape(P, 3, 1)
    rest_names = by_index("f_rest_")
    n_rest = 3 * (max_sh_degree + 1) ** 2 - 3
    if len(rest_names) != n_rest:
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* properties, expected {n_rest} for SH degree {max_sh_degree}")
    f_rest = np.stack([col(k) for k in rest_names], axis=1).reshape(P, 3, n_rest // 3) if n_rest else np.zeros((P, 3, 0), np.float32)
    occ = np.stack([col(f"occ_multiplier_{i}") for i in range(n_lvl_occ)], axis=1).reshape(P, n_lvl_occ, 1)
    dcd = np.stack([col(f"dc_delta_{i}_{j}") for i in range(n_lvl_dc) for j in range(3)], axis=1).reshape(P, n_lvl_dc * 3, 1)
    scales = np.stack([col(k) for k in by_index("scale_")], axis=1)
    rots = np.stack([col(k) for k in by_index("rot")], axis=1)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=device)
    model._xyz = nn.Parameter(t(xyz).requires_grad_(True))
    model._features_dc = nn.Parameter(t(f_dc).transpose(1, 2).contiguous().requires_grad_(True))
    model._features_rest = nn.Parameter(t(f_rest).transpose(1, 2).contiguous().requires_grad_(True))
    model._opacity = nn.Parameter(t(col("opacity")[:, None]).requires_grad_(True))
    model._occ_multiplier = nn.Parameter(t(occ), requires_grad=multi_occ)
    model._dc_delta = nn.Parameter(t(dcd), requires_grad=multi_dc)
    model._scaling = nn.Parameter(t(scales).requires_grad_(True))
    model._rotation = nn.Parameter(t(rots).requires_grad_(True))
    model.base_gaussian_mask = torch.from_numpy(np.asarray(v["base_gaussian_mask"]).astype(np.bool_)).to(device)
    model.max_pixel_sizes = t(col("max_pixel_sizes"))
    model.min_pixel_sizes = t(col("min_pixel_sizes"))
    model.active_sh_degree = max_sh_degree
    model.max_sh_degree = max_sh_degree
    return model


CAPTURE_FIELDS = ("active_sh_degree", "_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity",
                  "_occ_multiplier", "_dc_delta", "max_radii2D", "base_gaussian_mask", "max_pixel_sizes",
                  "min_pixel_sizes", "xyz_gradient_accum", "denom", "target_reso_lvl")


def capture(model, optimizer, spatial_lr_scale=1.0):
    """The 18-tuple of gaussian_model.py:79-99 (same order, so the reference can read it back... with its bug)."""
    return tuple(getattr(model, k) for k in CAPTURE_FIELDS) + (optimizer.state_dict(), spatial_lr_scale)


def restore(model, model_args, make_optimizer):
    """Inverse of capture() IN CAPTURE ORDER (see module docstring for the reference's permutation).
    make_optimizer(model) -> optimizer is called after the tensors are in place (training_setup, :117);
    returns (optimizer, spatial_lr_scale)."""
    if len(model_args) != len(CAPTURE_FIELDS) + 2:
        raise ValueError(f"checkpoint tuple has {len(model_args)} entries, expected {len(CAPTURE_FIELDS) + 2}")
    for k, val in zip(CAPTURE_FIELDS, model_args):
        setattr(model, k, val)
    opt_dict, spatial_lr_scale = model_args[-2], model_args[-1]
    xyz_gradient_accum, denom = model.xyz_gradient_accum, model.denom
    optimizer = make_optimizer(model)               # may re-allocate the accumulators like training_setup does
    model.xyz_gradient_accum, model.denom = xyz_gradient_accum, denom
    optimizer.load_state_dict(opt_dict)
    # active_sh_degree is restored verbatim (gaussian_model.py:101-125 has no SH bump; oneupSHdegree is only called
    # from the train loop, train.py)
    return optimizer, spatial_lr_scale


def create_from_points(model, points, colors, device="cuda", max_sh_degree=3, n_lvl_occ=4, n_lvl_dc=4):
    """GaussianModel.create_from_pcd (gaussian_model.py:186-227) on [P,3] points / [P,3] RGB in [0,1]: SH DC from the
    colours, log-scales from the 3-NN mean squared distance (distCUDA2, on the GPU), identity rotations, opacity 0.1."""
    from simple_knn._C import distCUDA2
    pts = torch.as_tensor(np.asarray(points), dtype=torch.float32).to(device)
    rgb = torch.as_tensor(np.asarray(colors), dtype=torch.float32).to(device)
    P = pts.shape[0]
    C0 = 0.28209479177387814
    features = torch.zeros((P, 3, (max_sh_degree + 1) ** 2), device=device)
    features[:, :3, 0] = (rgb - 0.5) / C0                                           # utils/sh_utils.py RGB2SH
    dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    rots = torch.zeros((P, 4), device=device)
    rots[:, 0] = 1
    opac = torch.full((P, 1), 0.1, device=device)
    model._xyz = nn.Parameter(pts.requires_grad_(True))
    model._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
    model._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
    model._scaling = nn.Parameter(scales.requires_grad_(True))
    model._rotation = nn.Parameter(rots.requires_grad_(True))
    model._opacity = nn.Parameter(torch.log(opac / (1 - opac)).requires_grad_(True))
    model._occ_multiplier = nn.Parameter(torch.ones((P, n_lvl_occ, 1), device=device), requires_grad=False)
    model._dc_delta = nn.Parameter(torch.zeros((P, n_lvl_dc * 3, 1), device=device), requires_grad=False)
    model.max_radii2D = torch.zeros(P, device=device)
    model.max_pixel_sizes = -torch.ones(P, device=device)
    model.min_pixel_sizes = -torch.ones(P, device=device)
    model.base_gaussian_mask = torch.zeros(P, dtype=torch.bool, device=device)
    model.target_reso_lvl = torch.zeros(P, dtype=torch.long, device=device)
    model.active_sh_degree, model.max_sh_degree = 0, max_sh_degree
    return model
