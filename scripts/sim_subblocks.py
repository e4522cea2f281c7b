#!/usr/bin/env python3
"""CPU estimate (no GPU) of the 4x4 sub-block walk with a per-tile LDS merge: a C2-shaped scene through the oracle,
contribution bits recomputed in numpy, then wave-iteration counts for
  today   8x4 blocks, 4 per wave (16 lanes x 2 px), blocks paired by list length inside 64-tile neighbourhoods
  new     4x4 sub-blocks, 8 per wave (8 lanes x 2 px), the 16 sub-blocks of a tile in ONE workgroup of two waves
          (a) no position windows (accumulators for the whole tile list), (b) windows of Wn tile-list positions with a barrier each.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/rade-gs_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np
from synth_scene import make_scene
from util import oracle_for

W, H = 384, 224
mu = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
P = int(round(1_000_000 / (1920 * 1080) * W * H))
s = make_scene(P=P, W=W, H=H, sh_degree=0, mu_px=mu, seed=1, require_coord=False, require_depth=True)
o = oracle_for(s)
R = o.forward()
m2 = o.get("means2D").reshape(-1, 2).astype(np.float32)
co = o.get("conic_opacity").reshape(-1, 4).astype(np.float32)
plist = o.get("point_list").astype(np.int64)[:R]
ranges = o.get("ranges").reshape(-1, 2).astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
print(f"scene: {P} Gaussians, {W}x{H}, num_rendered {R}, {R / P:.2f} tiles per Gaussian, {R / (gx * gy):.0f} entries per tile (C2: 3.9, 480)")

tiles = []   # per tile: (n, pos8[8] lists of positions, pos4[16] lists of positions, useful pairs)
for ty in range(gy):
    for tx in range(gx):
        a, b = ranges[ty * gx + tx]
        if b <= a:
            continue
        ids = plist[a:b]
        px = (tx * 16 + np.arange(16, dtype=np.float32))[None, None, :]
        py = (ty * 16 + np.arange(16, dtype=np.float32))[None, :, None]
        dx = m2[ids, 0][:, None, None] - px
        dy = m2[ids, 1][:, None, None] - py
        cx, cy, cz, op = (co[ids, k][:, None, None] for k in range(4))
        power = np.float32(-0.5) * (cx * dx * dx + cz * dy * dy) - cy * dx * dy
        alpha = np.minimum(np.float32(0.99), op * np.exp(power))
        ok = (power <= 0) & (alpha >= np.float32(1.0 / 255.0))
        T = np.ones((16, 16), np.float32)
        done = np.zeros((16, 16), bool)
        contrib = np.zeros(ok.shape, bool)
        inside = (py[0] < H) & (px[0] < W)
        for e in range(len(ids)):
            cand = ok[e] & ~done & inside
            tt = T * (1 - alpha[e])
            term = cand & (tt < 1e-4)
            done |= term
            act = cand & ~term
            contrib[e] = act
            T = np.where(act, tt, T)

        def lists(bw, bh):
            out = []
            for by in range(16 // bh):
                for bx in range(16 // bw):
                    sub = contrib[:, by * bh:(by + 1) * bh, bx * bw:(bx + 1) * bw].reshape(len(ids), -1)
                    reach = ok[:, by * bh:(by + 1) * bh, bx * bw:(bx + 1) * bw].reshape(len(ids), -1).any(1)
                    last = np.nonzero(sub.any(1))[0]
                    n_cons = (last[-1] + 1) if len(last) else 0
                    keep = reach.copy(); keep[n_cons:] = False
                    out.append(np.nonzero(keep)[0])
            return out
        tiles.append((len(ids), lists(8, 4), lists(4, 4), int(contrib.sum())))

import pickle
pickle.dump(tiles, open("/tmp/sim_tiles_%g.pkl" % mu, "wb"))
useful = sum(t[3] for t in tiles)
l8 = np.array([len(p) for t in tiles for p in t[1]], float)
l4 = np.array([len(p) for t in tiles for p in t[2]], float)
print(f"tiles {len(tiles)}, entries per tile {np.mean([t[0] for t in tiles]):.0f}; list per 8x4 block {l8.mean():.1f}, per 4x4 sub-block {l4.mean():.1f}; "
      f"(block, entry) pairs {int(l8.sum())} -> {int(l4.sum())} ({l4.sum() / l8.sum():.2f}x); (tile, entry) pairs with any contribution "
      f"{sum(len(np.unique(np.concatenate(t[2]))) if t[2] else 0 for t in tiles)}")
# today: sort blocks inside 64-tile neighbourhoods, quadruples
n = (l8.size // 512) * 512
it8 = np.sort(l8[:n].reshape(-1, 512), axis=1)[:, ::-1].reshape(-1, 4).max(1)
print(f"today: wave-iterations per 128 px {it8.mean():.1f} (unsorted strips {l8[:n].reshape(-1, 4).max(1).mean():.1f}); lane utilisation {useful / (32 * it8.sum() * 4 / 4 * 4 / 4):.3f}" if False else
      f"today: wave-iterations per wave (128 px) {it8.mean():.1f}; slots {it8.sum() * 128:.3g}, useful pairs {useful:.3g}, lane utilisation {useful / (it8.sum() * 128):.3f}")
base = it8.sum() * l8.size / n      # the neighbourhood sort covers whole groups of 512 blocks: scale to all blocks
# new (a): per tile two waves; the 8 longest sub-blocks in one wave, the 8 shortest in the other; also balanced across a 64-tile neighbourhood
per_tile = np.array([sorted((len(p) for p in t[2]), reverse=True) for t in tiles], float)   # [tiles, 16]
it_a = per_tile[:, 0] + per_tile[:, 8]
print(f"new (a) whole-list accumulators, per-tile waves: wave-iterations per tile {it_a.mean():.1f} = {it_a.sum() / base:.3f}x today's; "
      f"lane utilisation {useful / (it_a.sum() * 128):.3f}")
fl = l4[: (l4.size // 1024) * 1024]
it_n = np.sort(fl.reshape(-1, 1024), axis=1)[:, ::-1].reshape(-1, 8).max(1)
print(f"    (forward only: sub-blocks paired inside 64-tile neighbourhoods: {it_n.sum() / (base * fl.size / l4.size):.3f}x)")
# rounds of 16 entries: the forward / backward walk whole rounds only where a row still has entries; iterations are per entry (trip = min(16, nmax - 16 r)) -> same count
for Wn in (64, 128, 256, 512):
    tot = 0.0
    for (nt, _, p4, _) in tiles:
        order = np.argsort([-len(p) for p in p4])
        waves = (order[:8], order[8:])
        for w0 in range(0, nt, Wn):
            per_wave = []
            for wv in waves:
                per_wave.append(max(int(((p4[i] >= w0) & (p4[i] < w0 + Wn)).sum()) for i in wv))
            tot += 2 * max(per_wave)     # both waves wait at the window's barrier: the tile's two waves take max() each
    print(f"new (b) windows of {Wn} positions: {tot / base:.3f}x today's wave-iterations; LDS per tile {Wn * 64 / 1024:.0f} KB")
    tot2 = 0.0
    for (nt, _, p4, _) in tiles:
        order = np.argsort([-len(p) for p in p4])
        waves = (order[:8], order[8:])
        for wv in waves:
            for w0 in range(0, nt, Wn):
                tot2 += max(int(((p4[i] >= w0) & (p4[i] < w0 + Wn)).sum()) for i in wv)
    print(f"        (same, if waves did not wait for each other: {tot2 / base:.3f}x)")

# ---- a ring of `slots` half-size windows instead of barriers: a row may work in window w while no row is `slots` or more windows above it
def ring(WN, slots=2):
    tot = 0
    for (nt, p8, _, _) in tiles:
        lists = [p[::-1] for p in p8]          # rows walk back to front
        ptr = [0] * 8
        iters = [0, 0]
        while any(ptr[r] < len(lists[r]) for r in range(8)):
            wins = [lists[r][ptr[r]] // WN if ptr[r] < len(lists[r]) else -1 for r in range(8)]
            slowest = max(wins)
            adv = [wins[r] >= 0 and slowest - wins[r] < slots for r in range(8)]
            for wv in (0, 1):
                iters[wv] += any(adv[4 * wv:4 * wv + 4])
            for r in range(8):
                ptr[r] += adv[r]
        tot += iters[0] + iters[1]
    return tot / base
for WN in (32, 64):
    print(f"8x4 + per-tile workgroups + a ring of two windows of {WN} positions, no barriers: {ring(WN):.3f}x today's wave-iterations")
