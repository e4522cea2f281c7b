"""Generates tests/golden/filter3d.npz from the REFERENCE's GaussianModel.get_scaling_n_opacity_with_3D_filter
(/root/reference/scene/gaussian_model.py:156-166) evaluated on the CPU with torch autograd.  Build container only.
The module's unavailable imports (plyfile, simple_knn, trimesh, cv2) are stubbed and the object is created without
running __init__ (which allocates CUDA tensors); only setup_functions() + the property under test are executed."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for name, attrs in {"plyfile": ("PlyData", "PlyElement"), "simple_knn": (), "simple_knn._C": ("distCUDA2",), "trimesh": (), "cv2": ()}.items():
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, None)
    sys.modules.setdefault(name, m)
sys.path.insert(0, "/root/reference")
pkg = types.ModuleType("scene")            # keep scene/__init__.py (dataset readers, PIL, ...) from running
pkg.__path__ = ["/root/reference/scene"]
sys.modules["scene"] = pkg
from scene.gaussian_model import GaussianModel  # noqa: E402

rng = np.random.default_rng(0)
P = 4096
gm = object.__new__(GaussianModel)
gm.setup_functions()
sc = (np.log(0.01) + 1.2 * rng.standard_normal((P, 3))).astype(np.float32)
sc[:64] -= 6.0                                   # Gaussians much smaller than the filter
op = (2.0 * rng.standard_normal((P, 1))).astype(np.float32)
f3 = (0.002 + 0.02 * rng.random((P, 1))).astype(np.float32)
f3[64:128] = 0.0                                 # reset_3D_filter() state: filter off
gm._scaling = torch.from_numpy(sc).requires_grad_(True)
gm._opacity = torch.from_numpy(op).requires_grad_(True)
gm.filter_3D = torch.from_numpy(f3)
scales, opacity = gm.get_scaling_n_opacity_with_3D_filter
cs = torch.from_numpy(rng.standard_normal((P, 3)).astype(np.float32))
co = torch.from_numpy(rng.standard_normal((P, 1)).astype(np.float32))
((scales * cs).sum() + (opacity * co).sum()).backward()
# ---- compute_3D_filter over a ring of cameras (scene/gaussian_model.py:179-232) ----
from collections import namedtuple  # noqa: E402
import math  # noqa: E402
Cam = namedtuple("Cam", "R T image_width image_height FoVx FoVy")
xyz = (rng.standard_normal((P, 3)) * 2.0).astype(np.float32)
cams, rows = [], []
for c in range(12):
    ang = 2 * math.pi * c / 12
    Rwc = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    T = np.array([0.1 * c, -0.05 * c, 5.0])
    W, H = (640, 480) if c % 2 else (800, 600)
    cams.append(Cam(Rwc.T.copy(), T, W, H, math.radians(50 + c), math.radians(40 + c)))
    rows.append(np.concatenate([cams[-1].R.reshape(9), T, [W, H, cams[-1].FoVx, cams[-1].FoVy]]))
gm._xyz = torch.from_numpy(xyz)
gm.compute_3D_filter(cams)
extra = dict(xyz=xyz, cams=np.stack(rows), filter_out=gm.filter_3D.numpy())
np.savez_compressed(os.path.join(HERE, "filter3d.npz"), scaling_raw=sc, opacity_raw=op, filter_3D=f3, scales=scales.detach().numpy(),
                    opacity=opacity.detach().numpy(), cot_scales=cs.numpy(), cot_opacity=co.numpy(), g_scaling_raw=gm._scaling.grad.numpy(),
                    g_opacity_raw=gm._opacity.grad.numpy(), **extra)
print("ok", scales.shape, opacity.shape, float(opacity.mean()))
