"""Randomised parity sweep: scene density, image shapes that are not multiples of the tile, SH degree, 2D-filter size,
output modes, background colour, scale_modifier, camera pose, opacity regime -- forward indices exact, images within
tolerance, gradients against the float64 oracle with the reference's own arithmetic as the yardstick (tests/arbiter.py): one
criterion for every scene size.  Seeds are fixed, so a failure reproduces."""
import numpy as np
import pytest
import torch

from synth_scene import make_scene
from test_gpu_parity import check_forward

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

# The default sweep: seeds 0..39 plus the scenes round 5's sweep of 400 singled out -- the sixteen scenes of at least 2 000 Gaussians that
# missed that round's "standard" criteria (almost all 0.7-px splats) and the three small ones with the largest deviations.  ~1.3 s each.
# RADEGS_FUZZ_SEEDS="a:b" (or "s1,s2,...") replaces the list, e.g. 0:400 for a one-off hunt.
_DEFAULT = list(range(40)) + [47, 75, 106, 118, 152, 211, 220, 241, 242, 250, 256, 280, 289, 310, 340, 378] + [44, 54, 329, 22]
_SPEC = os.environ.get("RADEGS_FUZZ_SEEDS", "")
_SEEDS = _DEFAULT if not _SPEC else ([int(v) for v in _SPEC.split(",")] if "," in _SPEC or ":" not in _SPEC else list(range(*(int(v) for v in _SPEC.split(":")))))


def _run(seed):
    """Forward: exact indices, maps at 1e-5 / 1e-4 (check_forward).  Backward: ONE criterion for every scene size and tensor -- the product
    is as close to the exact (float64) gradient as the reference's own arithmetic is, and agrees with the reference wherever the
    reference's fp32 value is itself well-conditioned (tests/arbiter.py: criteria A-D, constants from profiles/r06_arbiter_table_*.txt).
    Rounds 2-5 used ">= 99 % strict + noise band" with looser numbers below 2 000 Gaussians; the table showed why that cannot hold on
    sub-pixel scenes of ANY size: the reference's own fp32 sums miss the exact value by a quarter of the tolerance on up to 43 % of the
    elements there."""
    import arbiter
    from synth_scene import upstream_grads
    kw, scale_modifier = _config(seed)
    if kw["mu_px"] >= 12.0:
        kw["P"] = min(kw["P"], 2500)  # heavy overdraw: keep the oracle's backward in seconds
    s = make_scene(**kw)
    check_forward(s, scale_modifier=scale_modifier)
    info, rows = arbiter.evaluate(s, upstream_grads(s, seed), scale_modifier=scale_modifier)
    bad = {k: arbiter.failed_criteria(v, worst_element=info["same_decisions"]) for k, v in rows.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, (seed, kw, {k: [(c, float(m), float(a)) for c, m, a in v] for k, v in bad.items()})


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_configuration(seed):
    _run(seed)


@pytest.mark.parametrize("streams", [0, 1])
@pytest.mark.parametrize("seed", [14, 15, 16, 17, 18, 19, 106, 241])
def test_random_configuration_forced_blend_path(seed, streams, monkeypatch):
    """The launcher picks tile-wide kernels or sub-tile entry streams by splat size; here each is forced on scenes it would not
    have been picked for (big splats through the streams, tiny ones -- two sub-pixel scenes among them -- through the tile-wide walk)."""
    monkeypatch.setenv("RADEGS_STREAMS", str(streams))
    _run(seed)
