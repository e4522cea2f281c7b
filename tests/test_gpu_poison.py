"""Poison mode (RADEGS_DEBUG_POISON=1, diff_gaussian_rasterization/_C.py): every buffer the native side is about to fill -- the produced
maps, radii, the gradients, the geometry / binning / image / accumulation state -- is first overwritten with 0xFF bytes / NaN.  A kernel
that reads something this call has not written, or leaves an output element unwritten, then shows up in the result instead of hiding
behind whatever the caching allocator's recycled memory happened to hold (usually the previous call's identical data).  The golden
cases must pass unchanged."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_golden_cases_pass_with_poisoned_buffers():
    env = dict(os.environ, RADEGS_DEBUG_POISON="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_golden.py"), os.path.join(HERE, "test_gpu_parity.py"),
                        "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", "golden or blend_paths or empty or ragged"],
                       env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-500:]
