"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of the step that FOLLOWS the rasterizer in
every training iteration >= 15000:

    depths_double_to_points   utils/graphics_utils.py:97-112   (back-project two depth maps with the pinhole rays)
    point_double_to_normal    utils/graphics_utils.py:116-123  (central differences -> cross -> normalize, border = 0)
    depth_double_to_normal    utils/graphics_utils.py:125-127
    the normal-consistency loss of train.py:146-155:
        err_k = 1 - sum_c rendered_normal_c * N_k,c ;  loss = (1-r) mean(err_0) + r mean(err_1),  r = 0.6

with the hand-derived backward (the reference relies on autograd).  Pinned against the reference's own functions run
on the CPU: tests/golden/make_golden_normals.py -> tests/golden/normals_*.npz -> tests/test_normal_oracle.py.
Everything is evaluated in the dtype of the inputs (float32 or float64)."""
import math

import numpy as np

EPS = 1e-12  # torch.nn.functional.normalize default


def rays(W, H, fovx, fovy, dtype):
    """graphics_utils.py:98-108: rays_d = K^-1 [x+0.5, y+0.5, 1]; returns (3,H,W)."""
    fx = W / (2 * math.tan(fovx / 2.0))
    fy = H / (2 * math.tan(fovy / 2.0))
    k = np.array([[1 / fx, 0.0, -W / (2 * fx)], [0.0, 1 / fy, -H / (2 * fy)], [0.0, 0.0, 1.0]], dtype=np.float32).astype(dtype)
    gx, gy = np.meshgrid(np.arange(W, dtype=dtype) + dtype(0.5), np.arange(H, dtype=dtype) + dtype(0.5), indexing="xy")
    pts = np.stack([gx, gy, np.ones_like(gx)], 0).reshape(3, -1)
    return (k @ pts).reshape(3, H, W).astype(dtype)


def depths_to_points(depth1, depth2, W, H, fovx, fovy):
    r = rays(W, H, fovx, fovy, depth1.dtype.type)
    return depth1.reshape(1, H, W) * r, depth2.reshape(1, H, W) * r


def points_to_normal(points):
    """points: (2,3,H,W) -> (2,3,H,W) normals (zero on the 1-pixel border); also returns the pieces the backward needs."""
    out = np.zeros_like(points)
    a = points[..., 2:, 1:-1] - points[..., :-2, 1:-1]   # the reference calls this dx: difference ALONG H
    b = points[..., 1:-1, 2:] - points[..., 1:-1, :-2]
    v = np.cross(a, b, axis=1)
    n = np.sqrt((v * v).sum(1, keepdims=True))
    out[..., 1:-1, 1:-1] = v / np.maximum(n, points.dtype.type(EPS))
    return out, (a, b, v, n)


def points_to_normal_bwd(points, grad):
    """d<grad, normal>/d points, grad: (2,3,H,W)."""
    _, (a, b, v, n) = points_to_normal(points)
    g = grad[..., 1:-1, 1:-1]
    nn = np.maximum(n, points.dtype.type(EPS))
    N = v / nn
    gv = np.where(n > EPS, (g - N * (N * g).sum(1, keepdims=True)) / nn, g / nn)
    ga = np.cross(b, gv, axis=1)
    gb = np.cross(gv, a, axis=1)
    gp = np.zeros_like(points)
    gp[..., 2:, 1:-1] += ga
    gp[..., :-2, 1:-1] -= ga
    gp[..., 1:-1, 2:] += gb
    gp[..., 1:-1, :-2] -= gb
    return gp


def depth_double_to_normal(depth1, depth2, W, H, fovx, fovy):
    p1, p2 = depths_to_points(depth1, depth2, W, H, fovx, fovy)
    return points_to_normal(np.stack([p1, p2], 0))[0]


def depth_double_to_normal_bwd(depth1, depth2, W, H, fovx, fovy, grad):
    p1, p2 = depths_to_points(depth1, depth2, W, H, fovx, fovy)
    gp = points_to_normal_bwd(np.stack([p1, p2], 0), grad)
    r = rays(W, H, fovx, fovy, depth1.dtype.type)
    return (gp[0] * r).sum(0).reshape(depth1.shape), (gp[1] * r).sum(0).reshape(depth2.shape)


def consistency_loss(rendered_normal, normals, depth_ratio=0.6):
    """train.py:152-155.  rendered_normal (3,H,W), normals (2,3,H,W) -> scalar."""
    t = rendered_normal.dtype.type
    err = 1 - (rendered_normal[None] * normals).sum(1)
    return t(1 - depth_ratio) * err[0].mean(dtype=np.float64) + t(depth_ratio) * err[1].mean(dtype=np.float64)


def consistency_loss_bwd(rendered_normal, normals, depth_ratio=0.6, upstream=1.0):
    """returns (d loss / d rendered_normal (3,H,W), d loss / d normals (2,3,H,W))."""
    t = rendered_normal.dtype.type
    H, W = rendered_normal.shape[-2:]
    w = np.array([1 - depth_ratio, depth_ratio], dtype=np.float64) * upstream / (H * W)
    g_normals = (-w[:, None, None, None] * rendered_normal[None]).astype(t)
    g_rendered = (-(w[:, None, None, None] * normals).sum(0)).astype(t)
    return g_rendered, g_normals
