"""Times GaussianRasterizer.integrate on the C2 scene (1M Gaussians, 1920x1080) with query points scattered around the
Gaussians the way mesh extraction places tetrahedra vertices.  Prints per-stage HIP-event timings.  (GPU box only.)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "rade-gs_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from synth_scene import make_config, to_device
import diff_gaussian_rasterization._C as C
from diff_gaussian_rasterization import GaussianRasterizer
from gpu_util import settings_for

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--points-per-gaussian", type=int, default=4)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--max-points", type=int, default=0, help="truncate the point set (isolates the image pass)")
ap.add_argument("--check", action="store_true", help="compare with the oracle (slow)")
a = ap.parse_args()
s = make_config(a.config, kernel_size=0.0)
P = s.means3D.shape[0]
rng = np.random.default_rng(0)
k = a.points_per_gaussian
pts = (s.means3D.numpy()[:, None, :] + rng.normal(size=(P, k, 3)).astype(np.float32) * 1.5 * s.scales.numpy().max(1)[:, None, None])
pts = np.ascontiguousarray(pts.reshape(-1, 3), dtype=np.float32)
if a.max_points:
    pts = pts[rng.permutation(len(pts))[:a.max_points]].copy()
dev = torch.device("cuda:0")
d = to_device(s, dev)
r = GaussianRasterizer(settings_for(s, dev))
p = torch.from_numpy(pts).to(dev)
def run():
    return r.integrate(p, d.means3D, None, d.opacities, shs=d.shs, scales=d.scales, rotations=d.rotations)
out = run(); torch.cuda.synchronize()
ts = []
for _ in range(a.iters):
    t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{a.config}: P={P} PN={len(pts)} integrate {min(ts)*1e3:.2f} ms (min of {a.iters}), projected {(out[3].abs().sum(1) > 0).sum().item()}, "
      f"max pts/pixel {out[0][8].max().item():.0f}, mean alpha {out[1].mean().item():.4f}")
if a.check:
    from util import oracle_for, close
    t0 = time.perf_counter()
    o = oracle_for(s); ref = o.integrate(pts)
    print(f"oracle {time.perf_counter()-t0:.1f} s")
    for kk, name in ((0, "color9"), (1, "alpha"), (2, "color_i"), (3, "coord"), (4, "sdf"), (5, "radii")):
        g = out[kk].cpu().numpy()
        ok = close(g, ref[kk]) if g.dtype != np.int32 else (g == ref[kk])
        print(name, "all close" if ok.all() else f"MISMATCH {(~ok).sum()} of {ok.size}, max abs {np.abs(g-ref[kk]).max():.3e}")
