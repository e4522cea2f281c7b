"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of
GaussianModel.get_scaling_n_opacity_with_3D_filter (scene/gaussian_model.py:156-166, activations :36-41) and its
hand-derived backward.  Pinned against the reference's own property evaluated with torch autograd on the CPU:
tests/golden/make_golden_filter3d.py -> tests/golden/filter3d.npz -> tests/test_filter3d_oracle.py."""
import numpy as np


def forward(scaling_raw, opacity_raw, filter_3D):
    s = np.exp(scaling_raw)
    s2 = np.square(s)
    det1 = s2.prod(axis=1)
    a2 = s2 + np.square(filter_3D)
    det2 = a2.prod(axis=1)
    coef = np.sqrt(det1 / det2)
    op = 1.0 / (1.0 + np.exp(-opacity_raw))
    return np.sqrt(a2), op * coef[..., None]


def backward(scaling_raw, opacity_raw, filter_3D, g_scales, g_opacity):
    s2 = np.square(np.exp(scaling_raw))
    f2 = np.square(filter_3D)
    a2 = s2 + f2
    coef = np.sqrt(s2.prod(axis=1) / a2.prod(axis=1))[..., None]
    sg = 1.0 / (1.0 + np.exp(-opacity_raw))
    g_op_raw = g_opacity * coef * sg * (1 - sg)
    g_sc_raw = g_scales * s2 / np.sqrt(a2) + (g_opacity * sg * coef) * f2 / a2
    return g_sc_raw, g_op_raw


def compute_3D_filter(xyz, cameras):
    """scene/gaussian_model.py:179-232 in numpy float32.  cameras: objects with R, T, image_width, image_height, FoVx, FoVy."""
    import math
    xyz = xyz.astype(np.float32)
    distance = np.full(xyz.shape[0], 100000.0, dtype=np.float32)
    valid_points = np.zeros(xyz.shape[0], dtype=bool)
    focal_length = 0.0
    for cam in cameras:
        W, H = cam.image_width, cam.image_height
        fx, fy = W / (2 * math.tan(cam.FoVx / 2.)), H / (2 * math.tan(cam.FoVy / 2.))
        pc = xyz @ np.asarray(cam.R, dtype=np.float32) + np.asarray(cam.T, dtype=np.float32)[None, :]
        valid_depth = pc[:, 2] > 0.2
        z = np.maximum(pc[:, 2], np.float32(0.001))
        x = pc[:, 0] / z * np.float32(fx) + np.float32(W / 2.0)
        y = pc[:, 1] / z * np.float32(fy) + np.float32(H / 2.0)
        in_screen = (x >= -0.15 * W) & (x <= W * 1.15) & (y >= -0.15 * H) & (y <= 1.15 * H)
        valid = valid_depth & in_screen
        distance[valid] = np.minimum(distance[valid], z[valid])
        valid_points |= valid
        focal_length = max(focal_length, fx)
    distance[~valid_points] = distance[valid_points].max()
    return (distance / np.float32(focal_length) * np.float32(0.2 ** 0.5))[:, None]
