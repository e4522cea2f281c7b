"""tests/golden/pyfallback.npz: the two sub-steps of the rasterizer's preprocess that the REFERENCE also implements in
Python (its `pipe.compute_cov3D_python` / `pipe.convert_SHs_python` switches, gaussian_renderer/__init__.py:143-166):
    cov3D   = strip_symmetric(L @ L^T),  L = build_scaling_rotation(scaling_modifier * scaling, rotation)
              (scene/gaussian_model.py:30-34, utils/general_utils.py:66-117)
    colors  = clamp_min(eval_sh(degree, shs_view, normalized(xyz - camera_center)) + 0.5, 0)   (utils/sh_utils.py:57-118)
run on the CPU.  These pin the oracle's computeCov3D / computeColorFromSH restatements (forward.cu:23-74,270-304) to code
the reference itself ships.  Build container only.  The reference hard-codes device='cuda' in torch.zeros: neutralised."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "rade-gs_amd"))
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


gen = load("ref_general_utils", "/root/reference/utils/general_utils.py")
shu = load("ref_sh_utils", "/root/reference/utils/sh_utils.py")
from synth_scene import make_scene  # noqa: E402

out = {}
for deg, seed, mod in ((0, 11, 1.0), (1, 12, 1.0), (2, 13, 0.7), (3, 14, 1.0)):
    s = make_scene(2000, 160, 120, sh_degree=deg, mu_px=3.0, seed=seed, pose="random", require_depth=True)
    rot = s.rotations   # normalised, as gaussian_renderer passes pc.get_rotation (the CUDA computeCov3D does not normalise; the Python one does)
    L = gen.build_scaling_rotation(mod * s.scales, rot)
    cov = gen.strip_symmetric(L @ L.transpose(1, 2))
    shs_view = s.shs.transpose(1, 2).view(-1, 3, 16)
    d = s.means3D - s.campos.repeat(2000, 1)
    d = d / d.norm(dim=1, keepdim=True)
    colors = torch.clamp_min(shu.eval_sh(deg, shs_view, d) + 0.5, 0.0)
    out.update({f"rot_{deg}": rot.numpy(), f"cov3D_{deg}": cov.numpy(), f"colors_{deg}": colors.numpy(), f"seed_{deg}": seed, f"mod_{deg}": mod})
np.savez_compressed(os.path.join(HERE, "pyfallback.npz"), **out)
print("ok")
