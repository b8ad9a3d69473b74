"""pip-installable form of the drop-in, for the install step of the reference's README
(`pip install submodules/diff-gaussian-rasterization`, `pip install submodules/simple-knn`, /root/reference/README.md:22-28):

    pip install /path/to/this/repo/ms-gs_amd          # provides `diff_gaussian_rasterization` AND `simple_knn`

builds lib/libmsgs_hip.so with hipcc for gfx950 (the Makefile next to this file) and ships it inside the
`diff_gaussian_rasterization` package, where _backend.py looks first.  Running from the source tree
(`PYTHONPATH=.../ms-gs_amd`) needs no install and stays the way the tests and bench.py use it."""
import os
import shutil
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class build_py_with_hip(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", HERE])
        super().run()
        dst = os.path.join(self.build_lib, "diff_gaussian_rasterization")
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(os.path.join(HERE, "lib", "libmsgs_hip.so"), os.path.join(dst, "libmsgs_hip.so"))


setup(
    name="diff_gaussian_rasterization",
    version="0.4.0+msgs.gfx950",
    description="MI355X (gfx950) drop-in for the MS-GS diff_gaussian_rasterization / simple_knn extensions",
    packages=["diff_gaussian_rasterization", "simple_knn"],
    package_dir={"diff_gaussian_rasterization": "diff_gaussian_rasterization", "simple_knn": "simple_knn"},
    package_data={"diff_gaussian_rasterization": ["libmsgs_hip.so"]},
    cmdclass={"build_py": build_py_with_hip},
    python_requires=">=3.9",
    zip_safe=False,
)
