"""Loads the reference's GaussianModel class in the build container (never on the GPU box): scene/gaussian_model.py BY FILE PATH,
with EMPTY placeholder modules for the third-party imports this image lacks (cv2, open3d, plyfile, simple_knn) — nothing of them is
touched by the code paths the fixtures exercise (getters, statistics); the placeholders only let the `import` statements pass.
Used by make_stats_golden.py-style generators and by the CPU tests that validate against the live reference when it is present."""
import importlib.util
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isfile(os.path.join(REF, "scene", "gaussian_model.py"))


_cached = None


def load_gaussian_model():
    """the reference's GaussianModel class; sys.modules and sys.path are left as they were found (the placeholders must not shadow
    this repository's own `simple_knn` for whoever imports it later in the same process)"""
    global _cached
    if _cached is not None:
        return _cached
    before, path_before = dict(sys.modules), list(sys.path)
    try:
        sys.path.append(REF)            # behind everything else: only the reference's own `utils` / `arguments` resolve here
        for name, attrs in (("cv2", {}), ("open3d", {}), ("open3d.ml", {}), ("open3d.ml.torch", {}),
                            ("plyfile", dict(PlyData=None, PlyElement=None)), ("simple_knn", {}),
                            ("simple_knn._C", dict(distCUDA2=None))):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        sys.modules["scene"] = pkg
        for mod in ("cameras", "gaussian_model"):
            full = f"scene.{mod}"
            spec = importlib.util.spec_from_file_location(full, os.path.join(REF, "scene", f"{mod}.py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[full] = m
            spec.loader.exec_module(m)
        _cached = sys.modules["scene.gaussian_model"].GaussianModel
    finally:
        for k in list(sys.modules):
            if k not in before:
                del sys.modules[k]
        for k, v in before.items():
            sys.modules[k] = v
        sys.path[:] = path_before
    return _cached
