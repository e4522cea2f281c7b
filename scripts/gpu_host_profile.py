#!/usr/bin/env python3
"""Host-side cost of one bench step (C2): how long the Python / ctypes / allocator work takes next to the 1.46 ms of GPU work it queues."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/rade-gs_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
import diff_gaussian_rasterization._C as C
from synth_scene import make_config, to_device, upstream_grads
dev = torch.device("cuda:0")
sc = make_config("C2")
s = to_device(sc, dev)
g = {k: v.to(dev) for k, v in upstream_grads(sc, 1).items()}
e = torch.Tensor([])
H, W = s.H, s.W

def fwd():
    return C.rasterize_gaussians(s.bg, s.means3D, e, s.opacities, s.scales, s.rotations, 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                 s.tanfovy, s.kernel_size, H, W, s.shs, s.sh_degree, s.campos, False, s.require_coord, s.require_depth, False)

def bwd(fw):
    R, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geom, binning, img = fw
    return C.rasterize_gaussians_backward(s.bg, s.means3D, radii, e, s.scales, s.rotations, 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                          s.tanfovy, s.kernel_size, g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"],
                                          g["normal"], normal, s.shs, s.sh_degree, s.campos, geom, R, binning, img, alpha, s.require_coord,
                                          s.require_depth, False)

for _ in range(10):
    bwd(fwd())
torch.cuda.synchronize()
N = 200
tf = tb = 0.0
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); fw = fwd(); b = time.perf_counter(); bwd(fw); c = time.perf_counter()
    tf += b - a; tb += c - b
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"step {1e3 * (t1 - t0) / N:.3f} ms; host time inside rasterize_gaussians {1e3 * tf / N:.3f} ms (includes the wait for num_rendered), "
      f"inside rasterize_gaussians_backward {1e3 * tb / N:.3f} ms")
# the same with the GPU drained before every call: pure host cost of queueing
tf = tb = 0.0
for _ in range(50):
    torch.cuda.synchronize(); a = time.perf_counter(); fw = fwd(); b = time.perf_counter()
    torch.cuda.synchronize(); b2 = time.perf_counter(); bwd(fw); c = time.perf_counter()
    tf += b - a; tb += c - b2
print(f"GPU idle at call time: forward call {1e3 * tf / 50:.3f} ms (queue + its own binning on the GPU until num_rendered arrives), backward call {1e3 * tb / 50:.3f} ms (pure queueing)")
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    bwd(fwd())
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
