"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np
import torch

from oracle.oracle import Oracle
from synth_scene import Scene, make_scene, upstream_grads  # noqa: F401

# north_star tolerance for floating-point outputs: 1e-5 abs / 1e-4 rel (fp32)
ATOL, RTOL = 1e-5, 1e-4


def oracle_for(s: Scene, precision=32, colors=None, cov3D=None, nthreads=None, scale_modifier=1.0):
    kw = dict(bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, campos=s.campos,
              tanfovx=s.tanfovx, tanfovy=s.tanfovy, image_height=s.H, image_width=s.W, sh_degree=s.sh_degree,
              kernel_size=s.kernel_size, require_coord=s.require_coord, require_depth=s.require_depth, precision=precision,
              nthreads=nthreads, scale_modifier=scale_modifier)
    if colors is None:
        kw["shs"] = s.shs
    else:
        kw["colors_precomp"] = colors
    if cov3D is None:
        kw["scales"], kw["rotations"] = s.scales, s.rotations
    else:
        kw["cov3D_precomp"] = cov3D
    return Oracle(**kw)


def oracle_backward(o, g):
    o.backward(g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
    return o.grads()


def close(a, b, atol=ATOL, rtol=RTOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) <= atol + rtol * np.abs(b)


def frac_close(a, b, atol=ATOL, rtol=RTOL):
    c = close(a, b, atol, rtol)
    return float(c.mean()) if c.size else 1.0


def cov3d_of(s: Scene):
    """(P,6) covariance from scales/rotations, float64 math rounded to float32 (an independent input)."""
    sc = s.scales.double()
    q = s.rotations.double()
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    Sg = R @ torch.diag_embed(sc * sc) @ R.transpose(1, 2)
    return torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).float().contiguous()


def grad_noise_floor(scene, g, colors=None, cov3D=None, scale_modifier=1.0):
    """fp32 conditioning of the backward.  The algorithm itself (reference arithmetic: T recovered by
    repeated division, differences of nearly equal blended values, sums of +/- terms) is only accurate to
    ~1e-3 relative in fp32 -- measured as |oracle_fp32 - oracle_fp64| per gradient tensor.  Two correct fp32
    implementations that round differently (fma vs mul+add, hardware exp/rcp) can therefore differ by a
    fraction of this floor even when both are as close to the exact gradient as fp32 allows.
    Returns {name: max |fp32 - fp64|} or None if the fp64 run took different thresholded decisions."""
    o32 = oracle_for(scene, colors=colors, cov3D=cov3D, nthreads=1, scale_modifier=scale_modifier)
    o32.forward()
    o64 = oracle_for(scene, precision=64, colors=colors, cov3D=cov3D, nthreads=1, scale_modifier=scale_modifier)
    o64.forward()
    if not (np.array_equal(o32.get("n_contrib"), o64.get("n_contrib")) and np.array_equal(o32.get("point_list"), o64.get("point_list"))):
        return None
    g32, g64 = oracle_backward(o32, g), oracle_backward(o64, g)
    return {k: float(np.abs(g32[k].astype(np.float64) - g64[k]).max()) if g32[k].size else 0.0 for k in g32}, g64
