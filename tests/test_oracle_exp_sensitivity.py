"""How much can a differently-rounded expf change the result?  ("document the expected borderline-decision mismatch
rate".)  The blend loop's thresholded decisions (alpha < 1/255, T(1-alpha) < 1e-4, T > 0.5) consume exp(power); the CUDA
reference uses its own expf, which cannot be reproduced off-device, so this repo's oracle and kernels share a fully specified
exp instead (oracle/radegs_oracle.cpp exp_spec, csrc/rg_blend.h).  Here the ORACLE is re-run with stand-ins for "another expf":
the C library's, and the specification moved one ulp up / down.  Measured on the C1 shape (10k Gaussians, 256x256):
the fraction of pixels whose contributor count / last contributor changes and the size of the image change.  The bounds asserted
below are what DESIGN.md section 7 quotes as the expected mismatch against a CUDA build."""
import numpy as np
import pytest

from oracle import oracle as orc
from util import make_scene, oracle_for


def _run(scene):
    o = oracle_for(scene, nthreads=None)
    o.forward()
    out = o.outputs()
    res = dict(n_contrib=o.get("n_contrib").copy(), color=np.array(out[0], dtype=np.float64), alpha=np.array(out[6], dtype=np.float64),
               depth=np.array(out[4], dtype=np.float64))
    o.close()
    return res


@pytest.mark.parametrize("mu_px,seed", [(1.5, 11), (4.0, 12)])
def test_decisions_under_a_differently_rounded_exp(mu_px, seed):
    s = make_scene(10000, 256, 256, sh_degree=0, mu_px=mu_px, seed=seed)
    base = _run(s)
    try:
        report = {}
        for mode, name in ((1, "libm expf"), (2, "spec + 1 ulp"), (3, "spec - 1 ulp")):
            orc.set_exp_mode(mode)
            r = _run(s)
            flipped = float((r["n_contrib"] != base["n_contrib"]).mean())
            d_color = float(np.abs(r["color"] - base["color"]).max())
            d_alpha = float(np.abs(r["alpha"] - base["alpha"]).max())
            report[name] = (flipped, d_color, d_alpha)
            # a flipped alpha >= 1/255 decision adds or removes one contribution of weight <= T/255; a flipped termination one
            # of weight <= 1e-4: the images move by less than 1/255 even where a decision flips
            assert flipped < 1e-3, report
            assert d_color < 4.5e-3 and d_alpha < 4.5e-3, report
        print("exp sensitivity (flipped-pixel fraction, max |d color|, max |d alpha|):", report)
    finally:
        orc.set_exp_mode(0)
    again = _run(s)
    assert np.array_equal(again["n_contrib"], base["n_contrib"]) and np.array_equal(again["color"], base["color"])
