import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
import make_golden
from gpu_util import HipRun
from synth_scene import make_scene, upstream_grads
from util import oracle_for, oracle_backward, close
for case in make_golden.CASES:
    want = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    s = make_scene(**make_golden.CASES[case])
    g = upstream_grads(s, make_golden.CASES[case]["seed"])
    h2 = HipRun(s, "cuda:0"); h2.forward(); got = h2.backward(g)
    o = oracle_for(s, nthreads=1); o.forward(); live = oracle_backward(o, g)
    for k in ("dL_dmeans2D", "dL_dopacity"):
        b = want[k].reshape(got[k].shape); l = live[k].reshape(got[k].shape)
        d = np.abs(got[k] - b); i = np.unravel_index(d.argmax(), d.shape)
        print(case, k, "max|hip-golden|", d.max(), "at", i, got[k][i], b[i], "live", l[i], "max|live-golden|", np.abs(l - b).max())
