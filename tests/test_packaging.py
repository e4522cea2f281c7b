"""`pip install` of the package (counterpart of the reference's `pip install submodules/diff-gaussian-rasterization`, DGR/setup.py:17-33):
an editable install into a throw-away virtual environment must expose `diff_gaussian_rasterization` (+ `simple_knn` and the fused-step
modules) with the native library loadable -- from a working directory that is not the repository, without the repository on sys.path."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rade-gs_amd")


@pytest.mark.skipif(shutil.which(os.environ.get("HIPCC", "hipcc")) is None and not os.path.exists(os.path.join(PKG, "diff_gaussian_rasterization", "libradegs_hip.so")),
                    reason="neither hipcc nor a built library")
def test_pip_install_exposes_the_drop_in_package():
    tmp = tempfile.mkdtemp(prefix="radegs_venv_")
    try:
        venv = os.path.join(tmp, "venv")
        # --without-pip: this image has no ensurepip; the system pip is visible through --system-site-packages (as torch is)
        subprocess.check_call([sys.executable, "-m", "venv", "--without-pip", "--system-site-packages", venv])
        py = os.path.join(venv, "bin", "python")
        env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
        subprocess.check_call([py, "-m", "pip", "install", "--quiet", "--no-build-isolation", "--no-index", "--no-deps", "-e", PKG], env=env, cwd=tmp)
        code = ("import diff_gaussian_rasterization as d, simple_knn, graphics_utils, loss_utils, gaussian_model_ops, fused_adam, view_parallel\n"
                "import diff_gaussian_rasterization._C as C, ctypes\n"
                "assert d.GaussianRasterizer is not None and d.GaussianRasterizationSettings._fields[0] == 'image_height'\n"
                "L = ctypes.CDLL(C._LIB_PATH)\n"
                "assert all(hasattr(L, s) for s in C.EXPORTED_SYMBOLS)\n"
                "print(d.__file__)\n")
        out = subprocess.check_output([py, "-c", code], env=env, cwd=tmp).decode().strip()
        assert os.path.samefile(os.path.dirname(out), os.path.join(PKG, "diff_gaussian_rasterization")), out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        shutil.rmtree(os.path.join(PKG, "rade_gs_amd.egg-info"), ignore_errors=True)
