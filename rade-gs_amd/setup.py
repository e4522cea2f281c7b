"""pip-installable form of the MI355X drop-in (counterpart of DGR/setup.py:17-33, which builds the CUDA extension with nvcc).

    pip install --no-build-isolation ./rade-gs_amd          # builds libradegs_hip.so with hipcc for gfx950, installs it as package data
    pip install --no-build-isolation -e ./rade-gs_amd       # in-tree (what the repository's tests and bench use through sys.path)

Installs the package the reference's callers import (`diff_gaussian_rasterization`: GaussianRasterizationSettings, GaussianRasterizer),
`simple_knn` (distCUDA2) and the fused steps either side of the rasterizer (graphics_utils, loss_utils, gaussian_model_ops, fused_adam,
view_parallel).  The native library is built by build.py, the same recipe `__graft_entry__.build()` runs; there is no CPU fallback."""
import importlib.util
import os

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    spec = importlib.util.spec_from_file_location("radegs_build", os.path.join(HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = mod.build(verbose=True)
    if os.environ.get("RADEGS_TORCH_BINDING", "0") == "1":   # also upstream's compiled `_C` module over the C ABI (RADEGS_BINDING=torch)
        mod.build_torch_binding(verbose=True)
    return lib


class BuildPyWithHip(build_py):
    def run(self):
        _build_native()
        super().run()


class DevelopWithHip(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="rade-gs-amd",
    version="0.6.0",
    description="MI355X-native differentiable Gaussian-splat rasterizer behind RaDe-GS's diff_gaussian_rasterization API (HIP, gfx950)",
    packages=["diff_gaussian_rasterization", "simple_knn"],
    py_modules=["graphics_utils", "loss_utils", "gaussian_model_ops", "fused_adam", "view_parallel", "synth_scene"],
    package_data={"diff_gaussian_rasterization": ["libradegs_hip.so", "_C_torch*.so"]},
    include_package_data=True,
    python_requires=">=3.8",
    cmdclass={"build_py": BuildPyWithHip, "develop": DevelopWithHip},
    zip_safe=False,
)
