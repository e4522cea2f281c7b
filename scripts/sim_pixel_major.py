#!/usr/bin/env python3
"""CPU estimate (no GPU) of what a pixel-major walk of per-pixel compacted lists would buy the blend backward.

A C2-shaped scene (same Gaussians per pixel, same splat size, smaller image) goes through the oracle's forward; from its splat
records, tile lists and the blend loop's own rules the contribution bit of every (pixel, tile-list entry) pair is recomputed in numpy.
Then, for every 8x4 block:
  today      one entry of the block's list per iteration for all 32 pixel slots:  iterations = list length (entries that reach the
             block: any pixel with alpha >= 1/255 -- the emission's masks keep 1.4 % more)
  pixel-major, window of W chunks of 16 list entries: every lane walks the contribution bits of its own pixel(s); a lane may run
             ahead of the slowest lane of its row by at most the window; iterations = steps until every lane is through.
Reported: iterations per block (mean), pair slots evaluated per useful pair, for 1 and 2 pixels per lane."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/rade-gs_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np
from synth_scene import make_scene
from util import oracle_for

W, H = 384, 224
P = int(round(1_000_000 / (1920 * 1080) * W * H))
s = make_scene(P=P, W=W, H=H, sh_degree=0, mu_px=1.5, seed=1, require_coord=False, require_depth=True)
o = oracle_for(s)
R = o.forward()
m2 = o.get("means2D").reshape(-1, 2).astype(np.float32)
co = o.get("conic_opacity").reshape(-1, 4).astype(np.float32)
plist = o.get("point_list").astype(np.int64)[:R]
ranges = o.get("ranges").reshape(-1, 2).astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
print(f"scene: {P} Gaussians, {W}x{H}, num_rendered {R}, {R / P:.2f} tiles per Gaussian, {R / (gx * gy):.0f} entries per tile (C2: 3.9, 480)")

blocks = []          # per 8x4 block: bool array [entries_reaching_block, 32 pixels]
pix_entries = []
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
        # transmittance / termination (forward.cu:552-573), sequential over the entries
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
        for by in range(4):
            for bx in range(2):
                sub = contrib[:, by * 4:by * 4 + 4, bx * 8:bx * 8 + 8].reshape(len(ids), 32)
                reach = ok[:, by * 4:by * 4 + 4, bx * 8:bx * 8 + 8].reshape(len(ids), 32).any(1)
                # the block's list stops where all its pixels have terminated (blk_consumed): drop the tail after the last contribution
                last = np.nonzero(sub.any(1))[0]
                n_cons = (last[-1] + 1) if len(last) else 0
                keep = reach.copy(); keep[n_cons:] = False
                blocks.append(sub[keep])
        pix_entries.append(contrib.reshape(len(ids), 256).sum(0))

useful = sum(int(b.sum()) for b in blocks)
lens = np.array([b.shape[0] for b in blocks])
print(f"blocks {len(blocks)}, mean list {lens.mean():.1f} entries, entries per pixel {np.concatenate(pix_entries).mean():.1f}, "
      f"lane utilisation today {useful / (32.0 * lens.sum()):.3f}")


def walk(bits, window_chunks):
    """bits [n, L] bool: L lanes, each follows its own set bits in order; a lane may only touch entries below
    16 * (chunk of the slowest unfinished lane + window_chunks).  Returns the number of steps."""
    n, L = bits.shape
    if n == 0:
        return 0
    pos = [np.nonzero(bits[:, l])[0] for l in range(L)]
    idx = np.zeros(L, int)
    cnt = np.array([len(p) for p in pos])
    steps = 0
    while True:
        live = idx < cnt
        if not live.any():
            return steps
        nxt = np.array([pos[l][idx[l]] if live[l] else n for l in range(L)])
        base = (nxt.min() // 16) * 16
        can = live & (nxt < base + 16 * window_chunks)
        idx[can] += 1
        steps += 1


rng = np.random.default_rng(0)
sample = rng.choice(len(blocks), size=min(1500, len(blocks)), replace=False)
tot_today = sum(blocks[i].shape[0] for i in sample)
use_s = sum(int(blocks[i].sum()) for i in sample)
print(f"sample of {len(sample)} blocks: today {tot_today / len(sample):.1f} iterations per block, {32 * tot_today / use_s:.2f} slots per useful pair")
for ppl, name in ((2, "16 lanes x 2 pixels (a lane walks the union of its two pixels' bits)"), (1, "32 lanes x 1 pixel")):
    for w in (1, 2, 4, 8, 1000):
        tot = 0
        for i in sample:
            b = blocks[i]
            bits = (b[:, :16] | b[:, 16:]) if ppl == 2 else b
            tot += walk(bits, w)
        slots = (32 if ppl == 2 else 32) * tot      # per step a row evaluates 32 slots (16 lanes x 2) or 32 lanes x 1
        print(f"  {name}, window {w if w < 1000 else 'inf'} chunks: {tot / len(sample):.1f} steps per block ({tot / tot_today:.2f}x today), "
              f"{slots / use_s:.2f} slots per useful pair")
