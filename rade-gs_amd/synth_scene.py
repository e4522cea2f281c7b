"""Synthetic scene + camera generator for the benchmark and the parity tests.

Restates the conventions the rasterizer's callers use in the reference (nothing here is on the
hot path):
  * camera matrices      scene/cameras.py:48-57, utils/graphics_utils.py:67-87
                         (world_view_transform and full_proj_transform are stored TRANSPOSED)
  * 3D (mip) filter      scene/gaussian_model.py:156-166 (applied outside the op, in the caller)
  * argument shapes      gaussian_renderer/__init__.py:56-79
Distributions follow SURVEY.md section 8(d).  Everything is generated on the CPU from a
seeded torch.Generator so that every rank / box sees identical inputs; move with `.to(device)`.
"""
import math
from typing import NamedTuple, Optional

import numpy as np
import torch


class _Rng:
    """Seeded float64 numpy streams (PCG64 + ziggurat: the same on every host).  All scene math below is
    done in float64 with numpy and rounded to float32 ONCE at the end, so the generated inputs are
    bit-identical on every machine; torch's CPU randn/exp/sigmoid are vectorised per ISA and differ in
    the last bit between hosts, which would make committed fixtures disagree with regenerated inputs."""

    def __init__(self, seed):
        self.g = np.random.default_rng(int(seed))

    def randn(self, *shape):
        return self.g.standard_normal(shape, dtype=np.float64)

    def rand(self, *shape):
        return self.g.random(shape, dtype=np.float64)

    def randperm(self, n):
        return self.g.permutation(n)


# BASELINE.json configs: name -> (P, W, H, sh_degree, mu_px, mode, seed)
CONFIGS = {
    "C1": dict(P=10_000, W=256, H=256, sh_degree=0, mu_px=1.5, require_coord=False, require_depth=True, seed=0),
    "C2": dict(P=1_000_000, W=1920, H=1080, sh_degree=3, mu_px=1.5, require_coord=False, require_depth=True, seed=1),
    "C3": dict(P=1_000_000, W=1600, H=1200, sh_degree=3, mu_px=1.5, require_coord=False, require_depth=True, seed=100),
    "C4": dict(P=5_000_000, W=1920, H=1080, sh_degree=3, mu_px=1.5, require_coord=True, require_depth=False, seed=4),
    "C5": dict(P=500_000, W=3840, H=2160, sh_degree=3, mu_px=12.0, require_coord=False, require_depth=True, seed=5,
               low_opacity=True),
    # not a BASELINE config: C2's size with the shape of a TRAINED scene (train.py:118-126 renders COLMAP reconstructions) -- a heavy-tailed
    # footprint distribution (log-normal sigma 1.3 instead of 0.6, plus 2 % of splats of 64 px and more) and Gaussians clustered on 200
    # blobs, so that tile-list lengths span more than an order of magnitude.  What the per-launch choice between the two blend
    # formulations has to survive (bench.py --config C2H; profiles/r05_C2H_*).
    # not a BASELINE config: VERDICT r5's "C2M" -- C2 plus 0.5 % screen-filling splats (200 .. 600 px) confined to the left third of the
    # image: two thirds of the tiles look like C2 (small splats: the entry streams' case), one third like C5 (the tile-wide kernels' case).
    # What a per-tile-band choice of blend formulation would have to beat (bench.py --config C2M).
    "C2M": dict(P=1_000_000, W=1920, H=1080, sh_degree=3, mu_px=1.5, require_coord=False, require_depth=True, seed=12,
                big_frac=0.005, big_px=200.0, big_band=(-1.0, -1.0 / 3.0)),
    "C2H": dict(P=1_000_000, W=1920, H=1080, sh_degree=3, mu_px=1.0, require_coord=False, require_depth=True, seed=11,
                sigma_ln=1.3, big_frac=0.02, big_px=64.0, clusters=200),
}


class Scene(NamedTuple):
    means3D: torch.Tensor      # (P,3)
    opacities: torch.Tensor    # (P,1) already multiplied by the 3D-filter coefficient
    scales: torch.Tensor       # (P,3) already 3D-filtered
    rotations: torch.Tensor    # (P,4) normalised (r,x,y,z)
    shs: torch.Tensor          # (P,16,3)
    viewmatrix: torch.Tensor   # (4,4) transposed world->view
    projmatrix: torch.Tensor   # (4,4) transposed full projection
    campos: torch.Tensor       # (3,)
    bg: torch.Tensor           # (3,)
    tanfovx: float
    tanfovy: float
    W: int
    H: int
    sh_degree: int
    kernel_size: float
    require_coord: bool
    require_depth: bool


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:67-87 (returned un-transposed, like the reference); float64 numpy."""
    t = math.tan(fovy / 2) * znear
    r = math.tan(fovx / 2) * znear
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * znear / (2 * r)
    P[1, 1] = 2.0 * znear / (2 * t)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def _rand_rotation(gen):
    q = gen.randn(4)
    r, x, y, z = (q / np.linalg.norm(q)).tolist()
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                     [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                     [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_scene(P, W, H, sh_degree=3, mu_px=1.5, seed=0, kernel_size=0.0, require_coord=False, require_depth=True,
               low_opacity=False, pose="identity", fovx_deg=60.0, bg=(0.0, 0.0, 0.0), near_cull_frac=0.02,
               filter3d=True, sigma_ln=0.6, big_frac=0.0, big_px=64.0, clusters=0, big_band=None) -> Scene:
    """sigma_ln: width of the log-normal footprint distribution; big_frac / big_px: that share of the splats gets a footprint of
    big_px .. 3 big_px pixels (6 sigma) instead; clusters > 0: the Gaussians sit on that many blobs (centres uniform in the frustum, blob radius
    3 .. 25 % of the blob's depth, blob populations log-normal) instead of filling the frustum uniformly."""
    gen = _Rng(seed)
    tanfovx = math.tan(math.radians(fovx_deg) * 0.5)
    tanfovy = tanfovx * H / W
    fovx, fovy = 2 * math.atan(tanfovx), 2 * math.atan(tanfovy)
    focal_x = W / (2 * tanfovx)

    # camera pose: world->view rotation Rw2c and translation T (view = Rw2c @ x + T)
    if pose == "identity":
        Rw2c, T = np.eye(3), np.zeros(3)
    else:
        Rw2c = _rand_rotation(gen)
        T = gen.randn(3) * 0.5
    w2c = np.eye(4)
    w2c[:3, :3] = Rw2c
    w2c[:3, 3] = T
    viewmatrix = w2c.T.copy()                                    # scene/cameras.py:54 (stored transposed)
    proj = projection_matrix(0.01, 100.0, fovx, fovy).T          # scene/cameras.py:55
    projmatrix = viewmatrix @ proj                               # scene/cameras.py:56
    campos = np.linalg.inv(viewmatrix)[3, :3]                    # scene/cameras.py:57

    def U(n, lo, hi):
        return gen.rand(n) * (hi - lo) + lo

    z = U(P, 2.0, 10.0)
    ncull = int(P * near_cull_frac)
    if ncull:
        z[:ncull] = U(ncull, -1.0, 0.2)
        z = z[gen.randperm(P)]
    zz = np.maximum(np.abs(z), 0.3)  # lateral extent also for culled points
    x = zz * tanfovx * U(P, -1.1, 1.1)
    y = zz * tanfovy * U(P, -1.1, 1.1)
    if clusters:
        cz = U(clusters, 2.0, 10.0)
        cx, cy = cz * tanfovx * U(clusters, -1.0, 1.0), cz * tanfovy * U(clusters, -1.0, 1.0)
        rad = cz * np.exp(U(clusters, math.log(0.03), math.log(0.25)))
        w = np.exp(1.0 * gen.randn(clusters))
        member = np.searchsorted(np.cumsum(w / w.sum()), gen.rand(P)).clip(0, clusters - 1)
        off = gen.randn(P, 3) * (rad[member] / 2.0)[:, None]
        live = z > 0.2                                  # the near-cull share keeps its place in front of the near plane
        x = np.where(live, cx[member] + off[:, 0], x)
        y = np.where(live, cy[member] + off[:, 1], y)
        z = np.where(live, np.maximum(cz[member] + off[:, 2], 0.5), z)
        zz = np.maximum(np.abs(z), 0.3)
    cam_pts = np.stack([x, y, z], 1)
    means3D = (cam_pts - T) @ Rw2c  # = Rw2c^T (p - T), row-vector form

    sigma_px = np.exp(math.log(mu_px) + sigma_ln * gen.randn(P))
    if big_frac > 0:
        big = gen.rand(P) < big_frac
        sigma_px = np.where(big, big_px / 6.0 * np.exp(U(P, 0.0, math.log(3.0))), sigma_px)   # footprint ~ 6 sigma: big_px .. 3 big_px
        if big_band is not None:   # the big splats' centres lie in this band of normalised image x (-1 .. 1); drawn AFTER everything above:
            xb = zz * tanfovx * U(P, big_band[0], big_band[1])   # configs without a band keep their streams
            cam_pts[:, 0] = np.where(big & (z > 0.2), xb, cam_pts[:, 0])
            means3D = (cam_pts - T) @ Rw2c
    aniso = np.exp(0.5 * gen.randn(P, 3))
    scales = (zz * sigma_px / focal_x)[:, None] * aniso
    if low_opacity:
        opacity = U(P, 0.02, 0.3)[:, None]
    else:
        opacity = 1.0 / (1.0 + np.exp(-2.0 * gen.randn(P, 1)))
    if filter3d:  # scene/gaussian_model.py:156-166 with filter_3D = z/focal * sqrt(0.2)
        filt = (zz / focal_x * math.sqrt(0.2))[:, None]
        s2 = scales * scales
        det1 = s2.prod(1)
        s2f = s2 + filt * filt
        det2 = s2f.prod(1)
        opacity = opacity * np.sqrt(det1 / det2)[:, None]
        scales = np.sqrt(s2f)
    q = gen.randn(P, 4)
    rotations = q / np.linalg.norm(q, axis=1, keepdims=True)
    shs = np.concatenate([gen.randn(P, 1, 3), 0.1 * gen.randn(P, 15, 3)], 1)
    return Scene(_t(means3D), _t(opacity), _t(scales), _t(rotations), _t(shs), _t(viewmatrix), _t(projmatrix), _t(campos),
                 torch.tensor(bg, dtype=torch.float32), tanfovx, tanfovy, W, H, sh_degree, float(kernel_size), bool(require_coord),
                 bool(require_depth))


def make_config(name, **over) -> Scene:
    kw = dict(CONFIGS[name])
    kw.update(over)
    return make_scene(**kw)


def jittered_view(scene: Scene, seed, angle_deg=2.0, shift=0.05, fovx_deg=60.0, dolly=0.0) -> Scene:
    """The same Gaussians seen from a slightly different camera (rotation by `angle_deg` about a random axis, translation by
    `shift`): what neighbouring training views look like.  Used for the per-rank views of the view-parallel bench so that
    every rank has the workload of the named config (an unrelated random pose would see almost none of the Gaussians).
    dolly > 0 additionally moves the camera that far INTO the scene: splats grow, num_rendered jumps -- the view a capacity
    prediction made on the others is too small for."""
    gen = _Rng(777_000 + int(seed))
    w2c = scene.viewmatrix.double().numpy().T
    axis = gen.randn(3)
    axis /= np.linalg.norm(axis)
    a = math.radians(angle_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    dR = np.eye(3) + math.sin(a) * K + (1 - math.cos(a)) * (K @ K)
    d = np.eye(4)
    d[:3, :3] = dR
    d[:3, 3] = gen.randn(3) * shift
    d[2, 3] -= dolly
    w2c = d @ w2c
    fovx, fovy = 2 * math.atan(scene.tanfovx), 2 * math.atan(scene.tanfovy)
    viewmatrix = w2c.T.copy()
    proj = projection_matrix(0.01, 100.0, fovx, fovy).T
    return scene._replace(viewmatrix=_t(viewmatrix), projmatrix=_t(viewmatrix @ proj), campos=_t(np.linalg.inv(viewmatrix)[3, :3]))


def upstream_grads(scene: Scene, seed=0):
    """Fixed random cotangents w_k for the 7 image outputs (loss = sum_k <w_k, out_k>)."""
    gen = _Rng(10_000 + int(seed))
    H, W = scene.H, scene.W

    def n(c):
        return _t(gen.randn(c, H, W))

    g = dict(color=n(3), coord=n(3), mcoord=n(3), depth=n(1), mdepth=n(1), alpha=n(1), normal=n(3))
    if not scene.require_coord:
        g["coord"].zero_()
        g["mcoord"].zero_()
    if not scene.require_depth:
        g["depth"].zero_()
        g["mdepth"].zero_()
    if not (scene.require_coord or scene.require_depth):
        g["normal"].zero_()
    return g


def to_device(scene: Scene, device) -> Scene:
    return Scene(*[v.to(device) if isinstance(v, torch.Tensor) else v for v in scene])
