"""Distribution `dprox-mi355x` -> top-level package `dprox` (see pyproject.toml).  The build hook compiles the gfx950 HIP library
with __graft_entry__.build() and places it inside the package (dprox/lib/) for wheels; an editable / develop install keeps using
delta-prox_amd/lib/ in the tree.  Counterpart of the reference's setup.py (same package name; its I/O, plotting and RL
dependencies are not needed by the solver path)."""
import os
import shutil
import sys

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = "delta-prox_amd"


class build_py_with_hip(build_py):
    def run(self):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.build()                                              # hipcc -> delta-prox_amd/lib/libdpx_hip.so (raises on failure)
        super().run()
        if not getattr(self, "editable_mode", False):
            dst = os.path.join(self.build_lib, "dprox", "lib")
            os.makedirs(dst, exist_ok=True)
            shutil.copy2(ge.LIB, os.path.join(dst, os.path.basename(ge.LIB)))


setup(
    name="dprox-mi355x",
    version="0.4.0",
    description="MI355X-native (gfx950 HIP) backend for the ADMM / proximal-gradient hot path of Delta-Prox, behind the reference's dprox API",
    long_description=open(os.path.join(ROOT, "README.md")).read(),
    long_description_content_type="text/markdown",
    python_requires=">=3.10",
    install_requires=["torch", "numpy", "tqdm"],
    extras_require={"fast-hash": ["xxhash"]},      # 10 GB/s fingerprint of NumPy observations (proxfn/quadratic.py; blake2b otherwise)
    package_dir={"": SRC},
    packages=find_packages(where=os.path.join(ROOT, SRC), include=["dprox*"]),
    cmdclass={"build_py": build_py_with_hip},
    zip_safe=False,
)
