"""oracle/normal_oracle.py against golden vectors produced by the REFERENCE's own functions
(utils/graphics_utils.py depth_double_to_normal / point_double_to_normal + the train.py:146-155 loss, torch autograd on
the CPU; tests/golden/make_golden_normals.py).  This row of the oracle is therefore PINNED to the reference."""
import os

import numpy as np
import pytest

from oracle import normal_oracle as no

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["depth_smooth", "depth_rough", "points"]


def load(name):
    z = np.load(os.path.join(GOLD, f"normals_{name}.npz"))
    return {k: z[k] for k in z.files}


def inputs(g, dtype):
    W, H = int(g["W"]), int(g["H"])
    if "points1" in g:
        pts = np.stack([g["points1"], g["points2"]], 0).astype(dtype)
    else:
        p1, p2 = no.depths_to_points(g["depth1"].astype(dtype), g["depth2"].astype(dtype), W, H, float(g["fovx"]), float(g["fovy"]))
        pts = np.stack([p1, p2], 0)
    return W, H, pts


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_forward_matches_reference(name, dtype):
    g = load(name)
    W, H, pts = inputs(g, dtype)
    nm, _ = no.points_to_normal(pts)
    assert np.abs(nm - g["normals"]).max() < 2e-5    # unit vectors; cross products of fp32 differences
    assert (nm[..., 0, :] == 0).all() and (nm[..., -1, :] == 0).all() and (nm[..., :, 0] == 0).all() and (nm[..., :, -1] == 0).all()
    loss = no.consistency_loss(g["rendered_normal"].astype(dtype), nm)
    assert abs(loss - float(g["loss"])) < 2e-6


@pytest.mark.parametrize("name", CASES)
def test_backward_matches_reference_autograd(name):
    g = load(name)
    dtype = np.float64
    W, H, pts = inputs(g, dtype)
    fovx, fovy = float(g["fovx"]), float(g["fovy"])
    nm, _ = no.points_to_normal(pts)
    rn = g["rendered_normal"].astype(dtype)
    g_rendered, g_normals = no.consistency_loss_bwd(rn, nm)
    assert np.abs(g_rendered - g["g_rendered"]).max() < 1e-9 + 1e-5 * np.abs(g["g_rendered"]).max()
    for cot, k1, k2 in ((g_normals, "g1", "g2"), (g["cot"].astype(dtype), "c1", "c2")):
        if "points1" in g:
            gp = no.points_to_normal_bwd(pts, cot)
            got = (gp[0], gp[1])
        else:
            got = no.depth_double_to_normal_bwd(g["depth1"].astype(dtype), g["depth2"].astype(dtype), W, H, fovx, fovy, cot)
        for a, k in zip(got, (k1, k2)):
            ref = g[k].reshape(a.shape)
            scale = np.abs(ref).max()
            # the reference's gradients are fp32 autograd through normalize(cross(...)) of tiny differences
            assert np.abs(a - ref).max() < 2e-3 * scale, (k, np.abs(a - ref).max(), scale)
            assert np.median(np.abs(a - ref)) < 1e-5 * scale
