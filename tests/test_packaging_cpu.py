"""`pip install ./ms-gs_amd` (the install step of the reference's README for its two extension submodules): the build half of it,
into a temporary directory — the package tree it produces carries libmsgs_hip.so, and that copy is the one the package loads."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_setup_py_build_ships_the_library_inside_the_package(tmp_path):
    base = str(tmp_path / "build")
    src = os.path.join(ROOT, "ms-gs_amd")
    p = subprocess.run([sys.executable, "setup.py", "-q", "build", "--build-base", base], cwd=src, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lib_dir = os.path.join(base, "lib")
    for rel in ("diff_gaussian_rasterization/__init__.py", "diff_gaussian_rasterization/_backend.py",
                "diff_gaussian_rasterization/libmsgs_hip.so", "simple_knn/__init__.py", "simple_knn/_C.py"):
        assert os.path.isfile(os.path.join(lib_dir, rel)), rel
    code = ("import diff_gaussian_rasterization as d, simple_knn._C as k; "
            "print(d._C._LIB_PATH); print(d._C.lib.msgs_abi_version()); print(callable(k.distCUDA2))")
    env = {k: v for k, v in os.environ.items() if k not in ("MSGS_HIP_LIB", "PYTHONPATH")}
    env["PYTHONPATH"] = lib_dir
    q = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert q.returncode == 0, q.stderr[-2000:]
    path, abi, ok = q.stdout.strip().splitlines()[-3:]
    assert os.path.realpath(path) == os.path.realpath(os.path.join(lib_dir, "diff_gaussian_rasterization", "libmsgs_hip.so"))
    assert int(abi) == 11 and ok == "True"
