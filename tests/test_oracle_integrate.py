"""Pins the ORACLE's integrate() (GaussianRasterizer.integrate, DGR forward.cu:187-235,855-900,938-1372): the
reference ships no fixture for it either, so analytic properties derived from the cited lines anchor it."""
import numpy as np
import torch

from synth_scene import make_scene
from util import oracle_for


def _scene(P=400, seed=3, **kw):
    return make_scene(P, 96, 64, sh_degree=1, mu_px=5.0, seed=seed, kernel_size=0.0, pose="random", require_coord=False,
                      require_depth=True, near_cull_frac=0.0, **kw)


def test_inverse_ray_covariance_is_the_local_affine_pullback_of_the_3d_precision():
    """For a well-conditioned Gaussian, forward.cu:187-208 builds the precision matrix in ray space (u/f, v/f, t):
    du^T icr du must equal d^T Sigma^-1 d to first order for a small world offset d, with du the exact change of
    (pixel x, pixel y, distance to the camera)."""
    s = _scene()
    o = oracle_for(s, precision=64)
    o.integrate(s.means3D.numpy()[:4])
    P = s.means3D.shape[0]
    icr, cond, radii = o.get("invraycov", (P, 6)), o.get("condition"), o.get("radii")
    view = s.viewmatrix.double().numpy()          # transposed: p_view = p @ view[:3,:3] + view[3,:3]
    fx, fy = s.W / (2 * s.tanfovx), s.H / (2 * s.tanfovy)
    sc, q = s.scales.double().numpy(), s.rotations.double().numpy()
    rng = np.random.default_rng(0)
    checked = 0
    for g in np.nonzero((radii > 0) & (cond == 1))[0][:150]:
        r, x, y, z = q[g] / np.linalg.norm(q[g])
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        prec = R @ np.diag(1.0 / sc[g] ** 2) @ R.T
        mu = s.means3D.double().numpy()[g]

        def ray(p):
            v = p @ view[:3, :3] + view[3, :3]
            return np.array([fx * v[0] / v[2], fy * v[1] / v[2], np.linalg.norm(v)])

        if abs((mu @ view[:3, :3] + view[3, :3])[0] / (mu @ view[:3, :3] + view[3, :3])[2]) > 1.25 * s.tanfovx:
            continue  # the reference clamps t.x/t.z at 1.3 tan(fov): the Jacobian is then taken elsewhere
        if abs((mu @ view[:3, :3] + view[3, :3])[1] / (mu @ view[:3, :3] + view[3, :3])[2]) > 1.25 * s.tanfovy:
            continue
        M = np.array([[icr[g, 0], icr[g, 1], icr[g, 2]], [icr[g, 1], icr[g, 3], icr[g, 4]], [icr[g, 2], icr[g, 4], icr[g, 5]]])
        for _ in range(4):
            d = rng.normal(size=3) * sc[g].min() * 1e-3
            du = ray(mu + d) - ray(mu)
            a, b = du @ M @ du, d @ prec @ d
            assert abs(a - b) <= 2e-2 * b, (g, a, b)
        checked += 1
    assert checked > 50


def test_point_bins_outputs_and_initial_values():
    s = _scene(P=300, seed=5)
    o = oracle_for(s)
    rng = np.random.default_rng(1)
    pts = np.concatenate([s.means3D.numpy()[rng.integers(0, 300, 600)] + rng.normal(size=(600, 3)).astype(np.float32) * 0.05,
                          rng.normal(size=(50, 3)).astype(np.float32) * 50.0]).astype(np.float32)  # some far outside the frustum
    color, alpha_i, color_i, coord, sdf, radii = o.integrate(pts)
    H, W = s.H, s.W
    proj = coord.any(axis=1)                       # projected points get their pixel position written
    assert 100 < proj.sum() < len(pts)
    # rasterize_points.cu:312-320: untouched points keep alpha 1, sdf -1000, colour 0
    assert (alpha_i[~proj] == 1).all() and (sdf[~proj] == -1000).all() and (color_i[~proj] == 0).all()
    assert ((alpha_i[proj] >= 0) & (alpha_i[proj] <= 1)).all() and (sdf[proj] != -1000).all()
    # channel 8 counts the points per pixel, and a point's integrated colour is its pixel's colour
    px, py = np.floor(coord[proj, 0]).astype(int), np.floor(coord[proj, 1]).astype(int)
    cnt = np.zeros((H, W))
    np.add.at(cnt, (py, px), 1)
    assert np.array_equal(cnt, color[8])
    assert np.array_equal(color_i[proj], color[:3, py, px].T)
    assert (color[5] == 0).all()
    # alpha channel = 1 - T, and expected depth <= max depth * alpha
    assert np.allclose(color[7], 1 - o.get("final_T", (H, W)), atol=2e-6)
    assert (color[3] <= color[6] * color[7] + 1e-4).all()


def test_integrated_opacity_grows_along_the_ray_and_vanishes_in_front():
    """One isotropic Gaussian on the optical axis; query points on the ray through it.  In front (many sigma) the
    integrated opacity is 0, it grows monotonically with depth and saturates at opacity * (2D footprint value) behind."""
    z0, sig, op = 4.0, 0.05, 0.8
    s = make_scene(1, 65, 65, sh_degree=0, seed=0, kernel_size=0.0, filter3d=False, near_cull_frac=0.0)
    s = s._replace(means3D=torch.tensor([[0.0, 0.0, z0]]), scales=torch.full((1, 3), sig), opacities=torch.tensor([[op]]),
                   rotations=torch.tensor([[1.0, 0, 0, 0]]))
    o = oracle_for(s, precision=64)
    zs = np.linspace(z0 - 6 * sig, z0 + 6 * sig, 25)
    # (0,0,z) projects to pixel position (W/2, H/2) = (32.5, 32.5): the centre of pixel (32,32)
    pts = np.stack([np.zeros_like(zs), np.zeros_like(zs), zs], 1)
    color, alpha_i, _, coord, sdf, _ = o.integrate(pts)
    assert np.allclose(coord, 32.5)
    assert alpha_i[0] < 1e-6 and (np.diff(alpha_i) >= -1e-12).all()
    # behind the Gaussian: du = (dx, dy, 0 along the ray's peak) -> opacity * exp(-0.5 * |0.5 px offset|^2 / sigma_px^2)
    sig_px = sig / z0 * (s.W / (2 * s.tanfovx))
    expect = op * np.exp(-0.5 * (0.5 ** 2 + 0.5 ** 2) / sig_px ** 2)   # ndc2Pix centres are 0.5 px off the point projection
    assert abs(alpha_i[-1] - expect) < 2e-3 * expect
    # half way (at the centre depth) a point sees the Gaussian only up to its own depth: du.z = 0 there as well
    assert abs(alpha_i[12] - expect) < 2e-3 * expect
    # sdf = (median-surface depth along the ray) - point depth: decreasing by exactly the spacing
    assert np.allclose(np.diff(sdf), -np.diff(zs), atol=1e-9)
