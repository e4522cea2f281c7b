#!/usr/bin/env python3
"""Regenerates tests/golden/g_*.npz -- golden vectors PRODUCED BY THE REFERENCE'S OWN CODE.

Every array below except the `floor_*` scalars is an output of oracle/_ref: the reference's forward.cu / backward.cu /
rasterizer_impl.cu / auxiliary.h compiled for the host from /root/reference (oracle/build_ref.py) and driven through
CudaRasterizer::Rasterizer::forward / backward, single-threaded (fixed order of the float atomics), built with -ffp-contract=off,
`exp` routed to the specified exponential (CUDA's expf is not reproducible off-device; see DESIGN.md section 7 for what
a differently rounded exp / nvcc's fma contraction can move).  This script runs in the build container only (it needs
/root/reference); the .npz files are what travels to the GPU box, where tests/test_golden.py compares the HIP path with them.
`floor_*` = |oracle fp32 - oracle fp64| per gradient tensor (util.grad_noise_floor): the fp32 conditioning of the algorithm, used
as the width of the tolerance band for the few gradient elements outside 1e-5 / 1e-4.
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
    # BASELINE.json configs[0] (C1) at its named size: 10k Gaussians, 256x256, SH degree 0, depth mode
    "g_C1": dict(P=10000, W=256, H=256, sh_degree=0, mu_px=1.5, seed=0, kernel_size=0.0, require_coord=False, require_depth=True),
    # a C2-shaped slice (SH3, 1.5-px splats, 4 tiles per Gaussian) with both flags on -- the render.py mode
    "g_C2s": dict(P=8000, W=208, H=112, sh_degree=3, mu_px=1.5, seed=1, kernel_size=0.0, require_coord=True, require_depth=True),
}
MAPS = (("color", 0), ("coord", 2), ("mcoord", 3), ("depth", 4), ("mdepth", 5), ("alpha", 6), ("normal", 7))


def _pack(R, out, get, grads):
    d = dict(num_rendered=np.int64(R), radii=out[1], point_list=get("point_list"), ranges=get("ranges"), n_contrib=get("n_contrib"),
             tiles_touched=get("tiles_touched"))
    for k, i in MAPS:
        d[k] = out[i]
    d.update(grads)
    return d


def run_reference(case):
    """The compiled reference (oracle/_ref)."""
    from oracle import ref
    from test_ref_parity import ref_for
    ref.set_exp("spec")
    ref.set_num_threads(1)
    s = make_scene(**CASES[case])
    r = ref_for(s)
    R = r.forward()
    out = r.outputs()
    g = upstream_grads(s, CASES[case]["seed"])
    r.backward(g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
    d = _pack(R, out, r.get, r.grads())
    ref.set_exp("libm")
    # the reference's own order noise: the same backward with the per-Gaussian sums formed in another order (double accumulators)
    from oracle import oracle as orc
    orc.set_opacity_slip(1)
    o = oracle_for(s, nthreads=1)
    o.forward()
    other = oracle_backward(o, g)
    for k, v in r.grads().items():
        d["noise_" + k] = np.float64(np.abs(v.astype(np.float64) - other[k]).max()) if v.size else np.float64(0)
    fl = grad_noise_floor(s, g)      # None: the float64 run takes a different thresholded decision somewhere (larger scenes)
    for k in r.grads():
        d["floor_" + k] = np.float64(fl[0][k]) if fl is not None else np.float64("nan")
    return d


def run(case, ref_order=True):
    """The hand-written oracle on the same scene (summing the backward in the reference's host order, so that it can be compared
    bit for bit with the vectors above)."""
    from oracle import oracle as orc
    s = make_scene(**CASES[case])
    orc.set_ref_order(1 if ref_order else 0)
    try:
        o = oracle_for(s, nthreads=1)
        R = o.forward()
        out = o.outputs()
        gr = oracle_backward(o, upstream_grads(s, CASES[case]["seed"]))
    finally:
        orc.set_ref_order(0)
    return _pack(R, out, o.get, gr)


if __name__ == "__main__":
    for case in CASES:
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **run_reference(case))
        print("wrote", case, os.path.getsize(os.path.join(HERE, case + ".npz")) // 1024, "KiB")
