"""Pins the oracle's hand-derived backward: central finite differences of the float64 oracle forward
against its analytic float64 backward, in all four flag combinations.  (The reference's backward is
~600 lines of hand calculus -- backward.cu:145-488 -- with no test upstream.)"""
import numpy as np
import pytest

from synth_scene import make_scene, upstream_grads
from util import oracle_for

# (the conftest fixture `_gradient_mode` runs these calculus checks on the INTENDED derivative: oracle.set_opacity_slip(0))
PARAMS = {"means3D": "dL_dmeans3D", "opacities": "dL_dopacity", "shs": "dL_dsh", "scales": "dL_dscales", "rotations": "dL_drotations"}


def _loss(scene, arrs, g, grads=False):
    s = scene._replace(**{k: __import__("torch").from_numpy(v) for k, v in arrs.items()})
    o = oracle_for(s, precision=64, nthreads=1)
    o.forward()
    out = o.outputs()
    L = 0.0
    for k, i in (("color", 0), ("coord", 2), ("mcoord", 3), ("depth", 4), ("mdepth", 5), ("alpha", 6), ("normal", 7)):
        L += float((g[k].numpy().astype(np.float64) * out[i]).sum())
    gr = None
    if grads:
        o.backward(g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
        gr = o.grads()
    # every discrete decision of the forward: blend thresholds, depth order, tile membership
    nc = np.concatenate([o.get("n_contrib").astype(np.int64), o.get("point_list").astype(np.int64), o.get("radii").astype(np.int64),
                         np.array([o.stat_blended()], dtype=np.int64)])
    o.close()
    return L, gr, nc


@pytest.mark.parametrize("seed,coord,depth,ks,pose", [(1, False, False, 0.0, "identity"), (2, False, True, 0.0, "identity"),
                                                      (3, True, False, 0.1, "random"), (4, True, True, 0.1, "random")])
def test_backward_matches_finite_differences(seed, coord, depth, ks, pose):
    s = make_scene(16, 24, 24, sh_degree=3, mu_px=3.0, seed=seed, kernel_size=ks, require_coord=coord, require_depth=depth,
                   pose=pose, bg=(0.3, 0.1, 0.7))
    g = upstream_grads(s, seed)
    arrs = {k: getattr(s, k).numpy().astype(np.float64) for k in PARAMS}
    _, gr, nc0 = _loss(s, arrs, g, True)
    rng = np.random.default_rng(seed)
    checked = 0
    for k, gn in PARAMS.items():
        a = arrs[k]
        ana = gr[gn].reshape(a.shape)
        idxs = list(np.ndindex(a.shape))
        idxs = [idxs[i] for i in rng.choice(len(idxs), min(len(idxs), 40), replace=False)]
        for ix in idxs:
            old = a[ix]
            h = 1e-6 * max(1.0, abs(old)) if k != "scales" else 1e-6 * abs(old)
            a[ix] = old + h
            Lp, _, ncp = _loss(s, arrs, g)
            a[ix] = old - h
            Lm, _, ncm = _loss(s, arrs, g)
            a[ix] = old
            if not (ncp.shape == nc0.shape and ncm.shape == nc0.shape and np.array_equal(ncp, nc0) and np.array_equal(ncm, nc0)):
                continue  # the perturbation crossed a thresholded decision: not differentiable there
            fd = (Lp - Lm) / (2 * h)
            assert abs(fd - ana[ix]) <= 2e-3 * (abs(fd) + abs(ana[ix])) + 2e-7, (k, ix, fd, ana[ix])
            checked += 1
    assert checked > 100
