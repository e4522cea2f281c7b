#!/usr/bin/env python3
"""GPU diagnostic: HIP vs oracle, stage by stage, printing mismatch statistics instead of asserting.
Usage: python scripts/gpu_diag.py [P W H mu_px coord depth]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpu_util import HipRun  # noqa: E402
from synth_scene import make_scene, upstream_grads  # noqa: E402
from util import close, oracle_backward, oracle_for  # noqa: E402


def diag(P, W, H, mu, coord, depth, seed=21, ks=0.1, pose="random", deg=3):
    print(f"\n=== diag P={P} {W}x{H} mu={mu} coord={coord} depth={depth} seed={seed} ===", flush=True)
    s = make_scene(P, W, H, sh_degree=deg, mu_px=mu, seed=seed, kernel_size=ks, require_coord=coord, require_depth=depth, pose=pose,
                   bg=(0.2, 0.5, 0.9))
    o = oracle_for(s)
    R_ref = o.forward()
    ref = o.outputs()
    h = HipRun(s, "cuda:0")
    t = time.time()
    st = h.forward_native()
    torch.cuda.synchronize()
    print(f"forward ok in {time.time() - t:.3f}s  R={st[0]} (ref {R_ref})", flush=True)
    radii = st[8].cpu().numpy()
    print("radii mismatches:", int((radii != ref[1]).sum()), "of", P)
    tt = h.export("tiles_touched", torch.int32, P).view(np.uint32)
    print("tiles_touched mismatches:", int((tt != o.get("tiles_touched")).sum()))
    if st[0] == R_ref and R_ref:
        pl = h.export("point_list", torch.int32, R_ref).view(np.uint32)
        bad = pl != o.get("point_list")
        print("point_list mismatches:", int(bad.sum()), "first at", int(np.argmax(bad)) if bad.any() else -1)
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    rg = h.export("ranges", torch.int32, 2 * ntiles).view(np.uint32)
    print("ranges mismatches:", int((rg != o.get("ranges")).sum()))
    nc = h.export("n_contrib", torch.int32, 2 * H * W).view(np.uint32)
    ncr = o.get("n_contrib")
    print("n_contrib mismatches: last", int((nc[:H * W] != ncr[:H * W]).sum()), " median", int((nc[H * W:] != ncr[H * W:]).sum()), "of", H * W)
    got = [st[1], None, st[2], st[3], st[6], st[7], st[4], st[5]]
    for k, name in enumerate(["color", "radii", "coord", "mcoord", "depth", "mdepth", "alpha", "normal"]):
        if k == 1:
            continue
        a, b = got[k].cpu().numpy(), ref[k]
        ok = close(a, b)
        print(f"  {name:7s} max|diff| {np.abs(a - b).max():.3e}  outside tol: {int((~ok).sum())}  nan {int(np.isnan(a).sum())}")
    g = upstream_grads(s, seed)
    refg = oracle_backward(o, g)
    h2 = HipRun(s, "cuda:0")
    h2.forward()
    t = time.time()
    gg = h2.backward(g)
    print(f"backward ok in {time.time() - t:.3f}s", flush=True)
    for k, b in refg.items():
        a = gg[k]
        if a is None:
            continue
        b = b.reshape(a.shape)
        ok = close(a, b)
        scale = np.abs(b).max()
        print(f"  {k:14s} max|diff| {np.abs(a - b).max():.3e} (scale {scale:.3e})  frac within 1e-5/1e-4: {ok.mean():.5f}  nan {int(np.isnan(a).sum())}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        P, W, H, mu, c, d = sys.argv[1:7]
        diag(int(P), int(W), int(H), float(mu), bool(int(c)), bool(int(d)))
    else:
        diag(3000, 200, 136, 3.0, False, True)
        diag(3000, 200, 136, 3.0, True, True)
        diag(3000, 200, 136, 3.0, False, False)
        diag(6000, 203, 117, 14.0, True, False, seed=33, ks=0.0, pose="identity", deg=2)
        diag(100000, 960, 540, 1.5, False, True, seed=1, ks=0.0, pose="identity")
