#!/usr/bin/env python3
"""Stress the forward for run-to-run reproducibility: the golden scenes rendered over and over, interleaved, in one process; every
result must equal the first one BIT FOR BIT (the forward has no atomics) and match the golden maps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/rade-gs_amd", ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import numpy as np, torch
import make_golden
from gpu_util import HipRun
from synth_scene import make_scene
from util import close
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = ["g_depth", "g_coord", "g_all", "g_C1", "g_C2s"]
scenes = {c: make_scene(**make_golden.CASES[c]) for c in cases}
want = {c: np.load(f"{ROOT}/tests/golden/{c}.npz") for c in cases}
first, bad = {}, 0
for it in range(n):
    for c in cases:
        h = HipRun(scenes[c], "cuda:0")
        st = h.forward_native()
        torch.cuda.synchronize()
        maps = [t.cpu().numpy() for t in st[1:8]]
        if c not in first:
            first[c] = maps
            for k, i in (("color", 0), ("alpha", 3), ("depth", 5)):
                assert close(maps[i], want[c][k]).all(), (c, k)
        else:
            for i, (a, b) in enumerate(zip(maps, first[c])):
                if not np.array_equal(a, b):
                    d = np.argwhere(a != b)
                    bad += 1
                    print(f"iter {it} {c} map {i}: {len(d)} elements differ from the first run, max |diff| {np.abs(a - b).max():.3e}, at {d[:6].tolist()}", flush=True)
print(f"{n} x {len(cases)} forwards, {bad} maps differed from the first run")
