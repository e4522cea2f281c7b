"""Randomised parity sweep: scene density, image shapes that are not multiples of the tile, SH degree, 2D-filter size,
output modes, background colour, scale_modifier, camera pose, opacity regime -- forward indices exact, images within
tolerance, gradients by the criteria of test_gpu_parity.check_backward with widened noise factors (small random scenes have
short, cancelling per-Gaussian sums).  Seeds are fixed, so a failure reproduces."""
import numpy as np
import pytest
import torch

from synth_scene import make_scene
from test_gpu_parity import check_backward, check_forward

pytestmark = pytest.mark.gpu


def _config(seed):
    r = np.random.default_rng(1000 + seed)
    coord, depth = [(False, False), (False, True), (True, False), (True, True)][int(r.integers(0, 4))]
    kw = dict(P=int(r.integers(300, 12000)), W=int(r.integers(33, 420)), H=int(r.integers(17, 300)), sh_degree=int(r.integers(0, 4)),
              mu_px=float(r.choice([0.7, 1.5, 4.0, 12.0, 30.0])), seed=int(r.integers(0, 10_000)),
              kernel_size=float(r.choice([0.0, 0.1, 0.3])), require_coord=coord, require_depth=depth,
              low_opacity=bool(r.integers(0, 2)), pose=str(r.choice(["identity", "random"])),
              bg=tuple(float(v) for v in r.random(3)) if r.integers(0, 2) else (0.0, 0.0, 0.0),
              near_cull_frac=float(r.choice([0.0, 0.02, 0.3])), fovx_deg=float(r.choice([40.0, 60.0, 95.0])))
    return kw, float(r.choice([1.0, 1.0, 0.5, 1.7]))


import os  # noqa: E402

# RADEGS_FUZZ_SEEDS="a:b" (or "s1,s2,...") widens the sweep, e.g. 14:214 for a one-off bug hunt; the default 14 run in ~20 s
_SPEC = os.environ.get("RADEGS_FUZZ_SEEDS", "0:14")
_SEEDS = [int(v) for v in _SPEC.split(",")] if "," in _SPEC else list(range(*(int(v) for v in _SPEC.split(":"))))


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_configuration(seed):
    kw, scale_modifier = _config(seed)
    if kw["mu_px"] >= 12.0:
        kw["P"] = min(kw["P"], 2500)  # heavy overdraw: keep the oracle's backward in seconds
    s = make_scene(**kw)
    o, h = check_forward(s, scale_modifier=scale_modifier)
    # Scenes of a few hundred Gaussians make the statistical criteria noisy (the fp32-oracle's own error is one random draw of
    # rounding, the HIP path's another): the strict-fraction floor and the fp64-arbiter factors are widened accordingly.  What
    # the sweep is for -- NaNs, wrong indices, gross errors on odd shapes/modes -- is untouched by this.  (More entry streams per
    # wave also mean more, smaller fp32 partial sums per Gaussian: the forced 8-stream backward sits at ~2x the oracle's rms error
    # on 300-Gaussian scenes; the reference itself adds one atomic per pixel-Gaussian pair.)
    check_backward(s, o, seed=seed, min_strict=0.90, scale_modifier=scale_modifier, rms_factor=2.5, max_factor=5.0, band_factor=16.0)
