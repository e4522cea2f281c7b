#!/usr/bin/env python3
"""Is a HIP-vs-oracle gradient difference fp32 noise or a defect?  Arbiter: the float64 oracle.
Prints, per gradient tensor, the error of the HIP path and of the fp32 oracle against fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import make_golden
from gpu_util import HipRun
from synth_scene import make_scene, upstream_grads
from util import oracle_for, oracle_backward

def run(name, s, seed):
    g = upstream_grads(s, seed)
    h = HipRun(s, "cuda:0"); h.forward(); got = h.backward(g)
    o32 = oracle_for(s, nthreads=1); o32.forward(); g32 = oracle_backward(o32, g)
    o64 = oracle_for(s, precision=64, nthreads=1); o64.forward(); g64 = oracle_backward(o64, g)
    same = np.array_equal(o32.get("n_contrib"), o64.get("n_contrib"))
    print(f"== {name}: fp32 and fp64 oracle decisions identical: {same}")
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        a, b, c = got[k].astype(np.float64), g32[k].reshape(got[k].shape).astype(np.float64), g64[k].reshape(got[k].shape)
        sc = np.abs(c).max()
        print(f"  {k:14s} scale {sc:9.3e}  max|hip-f64| {np.abs(a-c).max():.3e}  max|o32-f64| {np.abs(b-c).max():.3e}  rms hip {np.sqrt(((a-c)**2).mean()):.3e}  rms o32 {np.sqrt(((b-c)**2).mean()):.3e}")

for case in make_golden.CASES:
    run(case, make_scene(**make_golden.CASES[case]), make_golden.CASES[case]["seed"])
run("small3000", make_scene(3000, 200, 136, sh_degree=3, mu_px=3.0, seed=21, kernel_size=0.1, require_coord=False, require_depth=True, pose="random"), 21)
