"""CPU checks of the drop-in boundary (no compute calls: there is no GPU here): the C-ABI library
loads, exports every symbol include/msgs.h declares, its host-only size queries behave, and the
Python surface mirrors the reference's (names, fields, defaults, error behaviour)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "msgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(msgs_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    import diff_gaussian_rasterization as dgr
    lib = dgr._C.lib
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msgs.h but not exported"
    assert set(dgr._C.EXPORTS) == set(names)
    assert lib.msgs_abi_version() == dgr._C.ABI_VERSION == 11
    assert b"exactly one" in lib.msgs_error_string(-1)


def test_struct_layouts_match_header():
    """ctypes mirrors of msgs_view_t / msgs_gaussians_t / msgs_grads_t (LP64 layout of include/msgs.h)."""
    import diff_gaussian_rasterization as dgr
    from oracle import oracle_ctypes as oc
    assert C.sizeof(dgr._C.View) == 16 * 4 + 4 * 8 == C.sizeof(oc.View)
    assert C.sizeof(dgr._C.Gaussians) == 8 + 15 * 8 == C.sizeof(oc.Gaussians)
    assert C.sizeof(dgr._C.Grads) == 15 * 8 == C.sizeof(oc.Grads) and dgr._C.Grads.adam_in_backward.offset == 112 and C.sizeof(dgr._C.AdamInBackward) == 32 + 6 * 24 and dgr._C.Grads.scratch_is_clear.offset == 88
    assert dgr._C.Grads.accumulate.offset == 92 and dgr._C.Grads.wait_before_accumulate.offset == 96
    assert dgr._C.View.bg.offset == 64 and dgr._C.View.slab_fraction.offset == 56 and dgr._C.Gaussians.means3D.offset == 8
    assert [f[0] for f in dgr._C.View._fields_] == [f[0] for f in oc.View._fields_]


def test_size_queries_are_monotone_and_aligned():
    import diff_gaussian_rasterization as dgr
    lib = dgr._C.lib
    prev = 0
    for P in (0, 1, 1000, 10**6, 5 * 10**6):
        g = lib.msgs_geom_bytes(P)
        assert g >= prev and g % 256 == 0
        prev = g
        assert lib.msgs_stage1_scratch_bytes(P) > 0 and lib.msgs_backward_scratch_bytes(P) >= 48 * P
    assert lib.msgs_geom_bytes(10**6) >= 76 * 10**6
    assert lib.msgs_image_bytes(1920, 1080) >= 8 * 1920 * 1080
    for D in (0, 1, 10**7, 10**8):
        assert lib.msgs_binning_bytes(D, 1920, 1080) >= 4 * D + 8 * 120 * 68
        assert lib.msgs_stage2_scratch_bytes(D, 3840, 2160) >= 20 * D


def test_python_surface_mirrors_reference():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    # the 15 fields constructed by keyword at gaussian_renderer/__init__.py:37-53
    want = ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
            "projmatrix", "sh_degree", "campos", "prefiltered", "debug", "filter_small", "filter_large", "fade_size"]
    assert list(GaussianRasterizationSettings._fields) == want
    rs = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                       scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                       sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False,
                                       filter_small=False, filter_large=False, fade_size=1.0)
    r = GaussianRasterizer(raster_settings=rs)
    import inspect
    params = list(inspect.signature(r.forward).parameters)
    for k in ("means3D", "means2D", "shs", "colors_precomp", "max_pixel_sizes", "min_pixel_sizes", "opacities",
              "occ_multiplier", "dc_delta", "scales", "rotations", "cov3D_precomp", "base_mask"):
        assert k in params                     # the 13 kwargs of gaussian_renderer/__init__.py:95-107
    z = lambda *s: torch.zeros(*s)
    # upstream error convention: plain Exception on both / neither
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 16, 3))
    # no silent CPU fallback: host tensors are refused loudly
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 16, 3), scales=z(2, 3), rotations=z(2, 4))


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/ (task rule ③)."""
    pkg = os.path.join(ROOT, "ms-gs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_ctypes" not in txt and "torch_oracle" not in txt and "liboracle" not in txt, (dp, f)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import subprocess
    import sys
    code = ("import os, sys; os.environ['MSGS_HIP_LIB']='/nonexistent/libmsgs_hip.so'; "
            f"sys.path.insert(0, {os.path.join(ROOT, 'ms-gs_amd')!r}); import diff_gaussian_rasterization")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode != 0 and "HIP library not found" in p.stderr


def test_render_mirror_signature_and_keys():
    import inspect
    import gaussian_renderer
    sig = inspect.signature(gaussian_renderer.render)
    assert list(sig.parameters) == ["viewpoint_camera", "pc", "pipe", "bg_color", "scaling_modifier",
                                    "override_color", "filter_small", "filter_large", "fade_size"]
    assert sig.parameters["fade_size"].default == 1.0 and sig.parameters["filter_small"].default is False
    assert gaussian_renderer.RESULT_KEYS == ("render", "acc_pixel_size", "depth", "viewspace_points",
                                             "visibility_filter", "radii", "pixel_sizes")


def test_reference_getter_recognition_on_cpu_graphs():
    """_match_reference_getters only accepts the exact autograd patterns of the reference's getters."""
    import torch.nn.functional as F
    import diff_gaussian_rasterization as dgr
    P = 10
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s))
    xyz, dc, rest, op, sc, rot = mk(P, 3), mk(P, 1, 3), mk(P, 15, 3), mk(P, 1), mk(P, 3), mk(P, 4)
    cat = lambda: torch.cat((dc, rest), dim=1)
    m = dgr._match_reference_getters(xyz, cat(), torch.sigmoid(op), torch.exp(sc), F.normalize(rot))
    assert m is not None and all(x is y for x, y in zip(m, (dc, rest, op, sc, rot)))
    M = dgr._match_reference_getters
    assert M(xyz, cat(), torch.sigmoid(op), torch.exp(sc) * 1.0, F.normalize(rot)) is None          # not a bare exp
    assert M(xyz, cat().clone(), torch.sigmoid(op), torch.exp(sc), F.normalize(rot)) is None          # not a bare cat
    assert M(xyz, torch.cat((rest[:, :1], rest), dim=1), torch.sigmoid(op), torch.exp(sc), F.normalize(rot)) is None
    assert M(xyz, cat(), torch.sigmoid(op.detach()), torch.exp(sc), F.normalize(rot)) is None         # no graph
    assert M(xyz, cat(), torch.sigmoid(op), torch.exp(sc), F.normalize(rot, eps=1e-6)) is None        # other eps
    assert M(xyz, cat(), torch.sigmoid(op), torch.exp(sc), F.normalize(rot, dim=0)) is None
    assert M(xyz, cat(), torch.tanh(op), torch.exp(sc), F.normalize(rot)) is None
    half = torch.nn.Parameter(torch.randn(P, 3).double())
    assert M(xyz, cat(), torch.sigmoid(op), torch.exp(half), F.normalize(rot)) is None                # dtype
    with torch.no_grad():
        assert M(xyz, cat(), torch.sigmoid(op), torch.exp(sc), F.normalize(rot)) is None


def test_header_is_plain_c():
    """include/msgs.h must be consumable from C (cgo / JNI / any FFI generator): C99, no C++isms, no torch types."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "msgs.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = open(hdr).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S).lower()
