#!/usr/bin/env python3
"""CPU estimate (no GPU): how many (rectangle, list entry) pairs the tile-wide kernels' batch cull visits with its bounding-box test against an exact
ellipse-against-rectangle test, for rectangles of 16x16 (the 4-pixels-per-lane backward), 16x8 (forward), 16x4 and 8x4 pixels -- a scaled-down
BASELINE config through the oracle.     python tests/sim_cull.py C5 0.25   (test infrastructure: it drives the oracle)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/rade-gs_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np
from synth_scene import make_scene, CONFIGS
from util import oracle_for
cfg = sys.argv[1]
sc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
kw = dict(CONFIGS[cfg])
W0, H0, P0 = kw["W"], kw["H"], kw["P"]
W, H = int(W0 * sc) // 16 * 16, int(H0 * sc) // 16 * 16
kw.update(W=W, H=H, P=int(P0 * W * H / (W0 * H0)), sh_degree=0)
s = make_scene(**kw)
o = oracle_for(s)
R = o.forward()
m2 = o.get("means2D").reshape(-1, 2).astype(np.float64)
co = o.get("conic_opacity").reshape(-1, 4).astype(np.float64)
plist = o.get("point_list").astype(np.int64)[:R]
ranges = o.get("ranges").reshape(-1, 2).astype(np.int64)
gx, gy = W // 16, H // 16
print(cfg, "P", kw["P"], W, H, "R", R, "R/P", R / kw["P"], "per tile", R / (gx * gy))
tile_of = np.zeros(R, np.int64)
for t in range(gx * gy):
    a, b = ranges[t]
    tile_of[a:b] = t
ids = plist
mx, my = m2[ids, 0], m2[ids, 1]
cx, cy, cz, op = (co[ids, k] for k in range(4))
thr = np.log(1.0 / (255.0 * op))  # power >= thr needed
det = cx * cz - cy * cy
m = -2 * thr
ok = (thr <= 0) & (det > 0)
hx = np.sqrt(np.maximum(m * cz / det, 0)); hy = np.sqrt(np.maximum(m * cx / det, 0))
tx0 = (tile_of % gx) * 16.0; ty0 = (tile_of // gx) * 16.0
def rect_hits(x0, x1, y0, y1):
    bbox = ok & ~((mx + hx < x0) | (mx - hx > x1) | (my + hy < y0) | (my - hy > y1))
    # exact: min over rect of q(dx,dy) = cx dx^2 + 2 cy dx dy + cz dy^2 <= m ; convex -> check centre inside, else min over 4 edges
    def qmin_edge_h(yv, xa, xb):  # along horizontal edge y = yv: dy fixed
        dy = yv - my
        dxs = np.clip(-cy * dy / cx, xa - mx, xb - mx)
        return cx * dxs * dxs + 2 * cy * dxs * dy + cz * dy * dy
    def qmin_edge_v(xv, ya, yb):
        dx = xv - mx
        dys = np.clip(-cy * dx / cz, ya - my, yb - my)
        return cx * dx * dx + 2 * cy * dx * dys + cz * dys * dys
    inside = (mx >= x0) & (mx <= x1) & (my >= y0) & (my <= y1)
    q = np.minimum(np.minimum(qmin_edge_h(y0, x0, x1), qmin_edge_h(y1, x0, x1)), np.minimum(qmin_edge_v(x0, y0, y1), qmin_edge_v(x1, y0, y1)))
    exact = ok & (inside | (q <= m))
    return bbox, exact
for name, strips in (("16x16", [(0, 15, 0, 15)]), ("16x8", [(0, 15, 0, 7), (0, 15, 8, 15)]), ("16x4", [(0, 15, 4 * k, 4 * k + 3) for k in range(4)]),
                     ("8x4", [(8 * c, 8 * c + 7, 4 * k, 4 * k + 3) for k in range(4) for c in range(2)])):
    nb = ne = 0
    for (xa, xb, ya, yb) in strips:
        b, e = rect_hits(tx0 + xa, tx0 + xb, ty0 + ya, ty0 + yb)
        nb += b.sum(); ne += e.sum()
    n = len(strips) * R
    print(f"{name}: all {n}, bbox {nb} ({nb / n:.3f}), exact {ne} ({ne / n:.3f}), exact/bbox {ne / nb:.3f}")
