"""Generates tests/golden/normals_*.npz by running the REFERENCE's own Python functions
(/root/reference/utils/graphics_utils.py: depth_double_to_normal, point_double_to_normal) on the CPU, with torch
autograd for the gradients, plus the loss expression of train.py:146-155 evaluated on their outputs.
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_golden_normals.py
The reference hard-codes `.cuda()` and imports cv2; both are neutralised here (identity / empty stub)."""
import importlib.util
import math
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
torch.Tensor.cuda = lambda self, *a, **k: self
spec = importlib.util.spec_from_file_location("ref_graphics_utils", "/root/reference/utils/graphics_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

View = namedtuple("View", "image_width image_height FoVx FoVy")


def smooth_depth(rng, H, W, rough):
    y, x = np.mgrid[0:H, 0:W]
    d = 3.0 + 0.8 * np.sin(x / 7.0) * np.cos(y / 5.0) + 0.02 * x - 0.015 * y + rough * rng.standard_normal((H, W))
    return d.astype(np.float32)


def case(name, W, H, fovx_deg, seed, rough, mode):
    rng = np.random.default_rng(seed)
    fovx = math.radians(fovx_deg)
    fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    view = View(W, H, fovx, fovy)
    d1 = torch.from_numpy(smooth_depth(rng, H, W, rough)).reshape(1, H, W).requires_grad_(True)
    d2 = torch.from_numpy(smooth_depth(rng, H, W, rough) + 0.05).reshape(1, H, W).requires_grad_(True)
    rn = rng.standard_normal((3, H, W)).astype(np.float32)
    rn /= np.linalg.norm(rn, axis=0, keepdims=True)
    rn[:, : H // 4] *= 0.3                       # rendered normals are alpha-weighted: not unit length everywhere
    rendered_normal = torch.from_numpy(rn).requires_grad_(True)
    out = dict(W=W, H=H, fovx=fovx, fovy=fovy, depth1=d1.detach().numpy(), depth2=d2.detach().numpy(), rendered_normal=rn)
    if mode == "depth":
        nm = ref.depth_double_to_normal(view, d1, d2)
        leaves = (d1, d2)
    else:
        p1, p2 = ref.depths_double_to_points(view, d1.detach(), d2.detach())
        p1 = (p1 + 0.01 * torch.from_numpy(rng.standard_normal((3, H, W)).astype(np.float32))).requires_grad_(True)
        p2 = (p2 + 0.01 * torch.from_numpy(rng.standard_normal((3, H, W)).astype(np.float32))).requires_grad_(True)
        out.update(points1=p1.detach().numpy(), points2=p2.detach().numpy())
        nm = ref.point_double_to_normal(view, p1, p2)
        leaves = (p1, p2)
    # train.py:152-155
    depth_ratio = 0.6
    normal_error_map = (1 - (rendered_normal.unsqueeze(0) * nm).sum(dim=1))
    loss = (1 - depth_ratio) * normal_error_map[0].mean() + depth_ratio * normal_error_map[1].mean()
    loss.backward()
    out.update(normals=nm.detach().numpy(), loss=np.float32(loss.item()), g1=leaves[0].grad.numpy(), g2=leaves[1].grad.numpy(),
               g_rendered=rendered_normal.grad.numpy())
    # a second, generic cotangent on the normal maps alone (the functions are also used outside the loss)
    for t in leaves:
        t.grad = None
    cot = torch.from_numpy(rng.standard_normal(tuple(nm.shape)).astype(np.float32))
    nm2 = ref.depth_double_to_normal(view, d1, d2) if mode == "depth" else ref.point_double_to_normal(view, *leaves)
    (nm2 * cot).sum().backward()
    out.update(cot=cot.numpy(), c1=leaves[0].grad.numpy(), c2=leaves[1].grad.numpy())
    np.savez_compressed(os.path.join(HERE, f"normals_{name}.npz"), **out)
    print(name, "loss", out["loss"], "|g1|", np.abs(out["g1"]).max())


if __name__ == "__main__":
    case("depth_smooth", 48, 40, 60.0, 1, 0.0, "depth")
    case("depth_rough", 37, 29, 75.0, 2, 0.05, "depth")
    case("points", 40, 32, 50.0, 3, 0.02, "points")
