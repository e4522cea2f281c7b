"""Entry-stream statistics of a BASELINE config on the GPU: list lengths per block, iterations per wave (max over its 4 blocks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from synth_scene import make_config
from gpu_util import HipRun
os.environ["RADEGS_STREAMS"] = "1"
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
s = make_config(name)
h = HipRun(s, "cuda:0"); st = h.forward_native(); torch.cuda.synchronize()
nt = ((s.W + 15) // 16) * ((s.H + 15) // 16)
for arr in ("blk_count", "blk_consumed"):
    c = h.export(arr, torch.int32, nt * 8).reshape(nt, 2, 4).astype(np.float64)   # [tile][strip][group]
    rounds = np.ceil(c / 16) * 16
    print(name, arr, "mean per block %.1f | per wave: mean of max %.1f (rounded to 16-entry rounds %.1f), mean of mean %.1f | sorted-into-waves max-mean %.1f"
          % (c.mean(), c.max(2).mean(), rounds.max(2).mean(), c.mean(2).mean(),
             np.sort(c.reshape(-1))[::-1].reshape(-1, 4).max(1).mean()))
    # neighbourhood sort (64 tiles = 512 blocks)
    flat = c.reshape(-1)
    n = (flat.size // 512) * 512
    nb = np.sort(flat[:n].reshape(-1, 512), axis=1)[:, ::-1].reshape(-1, 4)
    print("   sorted inside 64-tile neighbourhoods: mean of max %.1f" % nb.max(1).mean())
