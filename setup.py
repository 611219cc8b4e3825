"""Build hook of torchpq_amd (metadata lives in pyproject.toml; the reference's counterpart is
/root/reference/setup.py:1-30, a pure-Python package whose CUDA text is compiled by NVRTC at run
time -- here the kernels are compiled AHEAD of time, for gfx950 only, by csrc/build.sh).

`build_py` first runs torchpq_amd/csrc/build.sh (hipcc must be on the host: /opt/rocm/bin/hipcc or
$HIPCC), then copies include/torchpq_amd.h next to the library so an installed package carries the
C ABI's header.  TPQ_SKIP_NATIVE_BUILD=1 packages an already built libtorchpq_amd.so as is.
"""
import os
import shutil
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "torchpq_amd")


def build_native():
    so = os.path.join(PKG, "libtorchpq_amd.so")
    if os.environ.get("TPQ_SKIP_NATIVE_BUILD") == "1" and os.path.exists(so):
        print("torchpq_amd: TPQ_SKIP_NATIVE_BUILD=1, packaging the existing", so)
    else:
        subprocess.check_call(["bash", os.path.join(PKG, "csrc", "build.sh")])
    if not os.path.exists(so):
        raise RuntimeError("torchpq_amd: csrc/build.sh did not produce libtorchpq_amd.so "
                           "(is hipcc installed? gfx950 is the only target)")
    os.makedirs(os.path.join(PKG, "include"), exist_ok=True)
    shutil.copy2(os.path.join(ROOT, "include", "torchpq_amd.h"), os.path.join(PKG, "include", "torchpq_amd.h"))


class BuildPy(build_py):
    def run(self):
        build_native()
        super().run()


class Develop(develop):
    def run(self):
        build_native()
        super().run()


def version():
    ns = {}
    exec(open(os.path.join(PKG, "_version.py")).read(), ns)
    return ns["__version__"]


setup(
    name="torchpq_amd",
    version=version(),
    description="MI355X-native IVFPQ train/add/search behind the TorchPQ IVFPQIndex API "
                "(hand-written HIP kernels for gfx950 behind a C ABI)",
    long_description=open(os.path.join(ROOT, "README.md"), encoding="utf-8").read(),
    long_description_content_type="text/markdown",
    license="MIT",
    keywords=["IVFPQ", "product quantization", "approximate nearest neighbors", "KMeans", "ROCm", "HIP", "MI355X"],
    python_requires=">=3.8",
    install_requires=["numpy", "torch"],
    packages=find_packages(include=["torchpq_amd", "torchpq_amd.*"], exclude=["torchpq_amd.variants*"]),
    package_data={"torchpq_amd": ["libtorchpq_amd.so", "include/torchpq_amd.h", "csrc/*.hip", "csrc/*.h",
                                  "csrc/*.cpp", "csrc/build.sh"]},
    include_package_data=False,
    zip_safe=False,
    cmdclass={"build_py": BuildPy, "develop": Develop},
)
