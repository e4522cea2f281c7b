#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference ships NO golden vector for this path and cannot be built or imported here (CUDA + glm,
SURVEY.md 8c), so these fixtures are outputs of the CPU oracle on small seeded scenes, frozen at the
commit where the oracle was pinned by its known-answer and finite-difference tests.  They guard the
oracle (and through it the HIP path) against silent drift; they are not reference outputs.
    python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from synth_scene import make_scene, upstream_grads  # noqa: E402
from util import grad_noise_floor, oracle_backward, oracle_for  # noqa: E402

CASES = {
    "g_depth": dict(P=400, W=64, H=48, sh_degree=3, mu_px=3.0, seed=101, kernel_size=0.1, require_coord=False, require_depth=True, pose="random"),
    "g_coord": dict(P=400, W=64, H=48, sh_degree=2, mu_px=3.0, seed=102, kernel_size=0.0, require_coord=True, require_depth=False, pose="random"),
    "g_all": dict(P=300, W=50, H=40, sh_degree=1, mu_px=5.0, seed=103, kernel_size=0.1, require_coord=True, require_depth=True, pose="identity"),
}


def run(case):
    s = make_scene(**CASES[case])
    o = oracle_for(s, nthreads=1)
    R = o.forward()
    out = o.outputs()
    g = upstream_grads(s, CASES[case]["seed"])
    gr = oracle_backward(o, g)
    floor, _ = grad_noise_floor(s, g)  # |fp32 - fp64| of the oracle: the fp32 conditioning of each gradient
    d = dict(num_rendered=np.int64(R), radii=out[1], point_list=o.get("point_list"), ranges=o.get("ranges"), n_contrib=o.get("n_contrib"))
    for k, i in (("color", 0), ("coord", 2), ("mcoord", 3), ("depth", 4), ("mdepth", 5), ("alpha", 6), ("normal", 7)):
        d[k] = out[i]
    d.update(gr)
    for k, v in floor.items():
        d["floor_" + k] = np.float64(v)
    return d


if __name__ == "__main__":
    for case in CASES:
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **run(case))
        print("wrote", case)
