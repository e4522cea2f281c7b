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
