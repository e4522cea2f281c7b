import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "rade-gs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from gpu_util import HipRun
from synth_scene import make_scene, upstream_grads
from util import oracle_for, oracle_backward
for P, mu, seed in ((30000, 1.5, 1), (30000, 6.0, 2), (4000, 1.5, 3), (60000, 3.0, 4)):
    s = make_scene(P, 320, 200, sh_degree=1, mu_px=mu, seed=seed, require_depth=True)
    h = HipRun(s, 'cuda:0'); out = h.forward(); g = h.backward(upstream_grads(s, seed))
    o = oracle_for(s); o.forward(); ref = oracle_backward(o, upstream_grads(s, seed))
    for k, v in g.items():
        if v is None: continue
        bad = ~np.isfinite(v)
        if bad.any():
            rows = np.unique(np.nonzero(bad)[0])
            print(P, mu, seed, k, 'non-finite rows', rows[:10], 'count', len(rows))
            r = rows[0]
            print('  gpu', v[r], ' oracle', ref[k][r] if k in ref else None, 'radii', o.get('radii')[r], 'scales', s.scales[r].numpy(), 'op', s.opacities[r].numpy())
            print('  other grads at row:', {kk: vv[r] for kk, vv in g.items() if vv is not None and kk != k and vv.ndim == 2})
print('done')
