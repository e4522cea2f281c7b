"""tests/golden/cameras.npz: the camera matrices the rasterizer's callers hand over, computed with the REFERENCE's own
getWorld2View2 / getProjectionMatrix exactly as scene/cameras.py:54-57 combines them (CPU).  Pins synth_scene.py's camera
conventions (transposed storage, full_proj = view @ proj, camera centre).  Build container only."""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
spec = importlib.util.spec_from_file_location("ref_graphics_utils", "/root/reference/utils/graphics_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
sys.path.insert(0, os.path.join(HERE, "..", "..", "rade-gs_amd"))
from synth_scene import make_scene  # noqa: E402

out = {}
for i, (W, H, fov, pose) in enumerate(((256, 256, 60.0, "identity"), (1920, 1080, 60.0, "random"), (320, 200, 95.0, "random"))):
    s = make_scene(8, W, H, seed=20 + i, pose=pose, fovx_deg=fov)
    w2c = s.viewmatrix.double().numpy().T                 # the generator's world->view matrix
    R, T = w2c[:3, :3].T.copy(), w2c[:3, 3].copy()        # scene/cameras.py stores R = c2w rotation, T = w2c translation
    fovx, fovy = 2 * math.atan(s.tanfovx), 2 * math.atan(s.tanfovy)
    world_view_transform = torch.tensor(ref.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    projection_matrix = ref.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (world_view_transform.float().unsqueeze(0).bmm(projection_matrix.unsqueeze(0))).squeeze(0)
    center = world_view_transform.float().inverse()[3, :3]
    out.update({f"args_{i}": np.array([W, H, fov, 20 + i, pose == "random"], dtype=np.float64), f"view_{i}": world_view_transform.float().numpy(),
                f"proj_{i}": full.numpy(), f"campos_{i}": center.numpy()})
np.savez_compressed(os.path.join(HERE, "cameras.npz"), **out)
print("ok")
