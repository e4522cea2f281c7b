"""Randomised parity sweep: scene density, image shapes that are not multiples of the tile, SH degree, 2D-filter size,
output modes, background colour, scale_modifier, camera pose, opacity regime -- forward indices exact, images within
tolerance, gradients by the criteria of test_gpu_parity.check_backward: the standard ones for scenes of at least 2 000 Gaussians,
widened noise factors below (short, cancelling per-Gaussian sums).  Seeds are fixed, so a failure reproduces."""
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


# Gradient criteria of the sweep (check_backward) as a FUNCTION OF SCENE SIZE.  Scenes of at least 2 000 Gaussians run the standard
# criteria of test_gpu_parity.py -- >= 99 % of every tensor's elements inside the strict 1e-5 / 1e-4 bar, the rest inside the fp32 noise
# band; against the fp64 oracle at most 1.1x the fp32 oracle's own rms error and 1.25x its max error.  Below that the statistics
# themselves are noisy (the fp32 oracle's error is ONE random draw of rounding over a few hundred short, cancelling sums, the HIP
# path's another): 0.97 / 4x band / 1.5x rms / 2x max.  The split is the measured one: of 40 seeds at the standard criteria the four
# that fail have 306, 627, 961 and 1 261 Gaussians, and they fail identically with a build whose blend backward uses the specified
# exponential and an IEEE division (profiles/r03_fuzz_table_*.txt).  RADEGS_FUZZ_THRESH overrides both (experiments).
_STANDARD, _SMALL_SCENE, _SMALL_BELOW = (0.99, 1.1, 1.25, 1.0), (0.97, 1.5, 2.0, 4.0), 2000
_FORCED = [float(v) for v in os.environ["RADEGS_FUZZ_THRESH"].split(",")] if os.environ.get("RADEGS_FUZZ_THRESH") else None


def _run(seed):
    kw, scale_modifier = _config(seed)
    if kw["mu_px"] >= 12.0:
        kw["P"] = min(kw["P"], 2500)  # heavy overdraw: keep the oracle's backward in seconds
    s = make_scene(**kw)
    o, h = check_forward(s, scale_modifier=scale_modifier)
    t = _FORCED or (_STANDARD if kw["P"] >= _SMALL_BELOW else _SMALL_SCENE)
    check_backward(s, o, seed=seed, min_strict=t[0], scale_modifier=scale_modifier, rms_factor=t[1], max_factor=t[2], band_factor=t[3])


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_configuration(seed):
    _run(seed)


@pytest.mark.parametrize("streams", [0, 1])
@pytest.mark.parametrize("seed", [14, 15, 16, 17, 18, 19])
def test_random_configuration_forced_blend_path(seed, streams, monkeypatch):
    """The launcher picks tile-wide kernels or sub-tile entry streams by splat size; here each is forced on scenes it would not
    have been picked for (big splats through the streams, tiny ones through the tile-wide walk)."""
    monkeypatch.setenv("RADEGS_STREAMS", str(streams))
    _run(seed)
