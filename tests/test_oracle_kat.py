"""Known-answer tests that pin the ORACLE: the reference ships no test, golden vector
or fixture for this path, so these analytic cases -- each derived from the cited reference lines --
plus the finite-difference check (test_oracle_fd.py) are what the oracle is anchored to."""
import math

import numpy as np
import torch

from oracle import oracle as orc
from synth_scene import make_scene, projection_matrix
from util import oracle_for


def test_glm_column_major_kat():
    # the reference carries this KAT in a comment: mat3(1..9)*(1,1,1) = (12,15,18)  (forward.cu:126-133)
    assert orc.kat_mat3().tolist() == [12.0, 15.0, 18.0]


def test_get_higher_msb_for_the_five_configs():
    # rasterizer_impl.cu:35-50 on the tile counts of BASELINE.json's configs (SURVEY 8c vi)
    assert [orc.higher_msb(t) for t in (16 * 16, 120 * 68, 100 * 75, 120 * 68, 240 * 135)] == [9, 13, 13, 13, 15]


def test_exp_spec_accuracy_and_monotone_neighbourhood():
    xs = np.concatenate([np.linspace(-87, 0, 5001), -np.logspace(-9, 1.9, 500)]).astype(np.float32)
    rel = [abs(orc.exp_spec(float(x)) - math.exp(float(x))) / math.exp(float(x)) for x in xs]
    assert max(rel) < 1.3e-7  # <= ~1 ulp: indistinguishable from CUDA expf (2 ulp) at the 1e-5/1e-4 tolerance
    assert orc.exp_spec(0.0) == 1.0 and orc.exp_spec(-100.0) == 0.0


def test_sym_eigen3_reconstructs():
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = rng.normal(size=(3, 3)).astype(np.float32)
        S = (A @ A.T).astype(np.float32)
        D, ev, V = orc.sym_eigen3([S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])
        assert D == 3
        np.testing.assert_allclose(V @ np.diag(ev) @ V.T, S, atol=2e-5 * np.abs(S).max())
        ref = np.linalg.eigvalsh(S.astype(np.float64))
        np.testing.assert_allclose(np.sort(ev), ref, rtol=1e-5, atol=2e-6 * ref.max())  # fp32 solver, error ~ eps*|S|


def _single(W=64, H=64, z=4.0, scale=0.05, opacity=0.8, sh0=0.7, bg=(0.2, 0.4, 0.6), require_depth=True, kernel_size=0.0,
            n=1, opac=None, zs=None):
    """n isotropic Gaussians on the optical axis of an identity camera."""
    s = make_scene(n, W, H, sh_degree=0, seed=0, kernel_size=kernel_size, require_coord=False, require_depth=require_depth,
                   near_cull_frac=0.0, filter3d=False, bg=bg)
    means = torch.zeros(n, 3)
    means[:, 2] = torch.tensor(zs if zs is not None else [z] * n)
    shs = torch.zeros(n, 16, 3)
    shs[:, 0, :] = sh0
    q = torch.zeros(n, 4)
    q[:, 0] = 1
    op = torch.tensor(opac if opac is not None else [opacity] * n).reshape(n, 1)
    return s._replace(means3D=means, shs=shs, rotations=q, scales=torch.full((n, 3), scale), opacities=op)


def test_empty_and_all_culled_scene():
    # (i) everything behind z <= 0.2 -> color = bg, every other map 0, radii 0, num_rendered 0 (forward.cu:348-353,636)
    s = _single(zs=[0.1], n=1)
    o = oracle_for(s)
    assert o.forward() == 0
    col, radii, coord, mcoord, depth, mdepth, alpha, normal = o.outputs()
    assert radii.tolist() == [0]
    for c in range(3):
        assert np.all(col[c] == np.float32(s.bg[c].item()))
    for m in (coord, mcoord, depth, mdepth, alpha, normal):
        assert np.all(m == 0)
    assert np.all(o.get("n_contrib")[: s.H * s.W] == 0)
    # P == 0: the reference does not even launch (rasterize_points.cu:90) -> all-zero outputs
    s0 = s._replace(means3D=torch.zeros(0, 3), shs=torch.zeros(0, 16, 3), rotations=torch.zeros(0, 4), scales=torch.zeros(0, 3),
                    opacities=torch.zeros(0, 1))
    o0 = oracle_for(s0)
    assert o0.forward() == 0
    assert np.all(o0.outputs()[0] == 0)


def test_single_gaussian_centre_pixel_closed_form():
    # (ii)+(iii): pixel centres are integers, the on-axis mean lands at (S-1)/2 = 31.5 (auxiliary.h:57-60)
    W = H = 64
    z, sc, op, sh0 = 4.0, 0.05, 0.8, 0.7
    s = _single(W, H, z, sc, op, sh0)
    o = oracle_for(s)
    assert o.forward() > 0
    col, radii, _, _, depth, mdepth, alpha, normal = o.outputs()
    fx = W / (2 * s.tanfovx)
    var = (fx * sc / z) ** 2  # isotropic 2D variance in px^2 (J = f/z on the axis)
    m2 = o.get("means2D")
    np.testing.assert_allclose(m2, [31.5, 31.5], atol=1e-4)
    coef = math.sqrt(var * var / (var * var + 1e-6) + 1e-6)  # kernel_size = 0 (forward.cu:119-121)
    px, py = 31, 31
    d2 = (31.5 - px) ** 2 + (31.5 - py) ** 2
    a = min(0.99, op * coef * math.exp(-0.5 * d2 / var))
    rgb = max(0.0, 0.28209479177387814 * sh0 + 0.5)
    assert abs(alpha[0, py, px] - a) < 2e-5
    for c in range(3):
        assert abs(col[c, py, px] - (a * rgb + (1 - a) * s.bg[c].item())) < 2e-5
    # single contributor with T = 1 > 0.5: median depth == expected depth == t/ln-corrected ray length
    assert abs(mdepth[0, py, px] - depth[0, py, px]) < 1e-4
    assert abs(depth[0, py, px] - z) < 0.02
    # fronto-parallel isotropic splat: normal points back at the camera
    np.testing.assert_allclose(normal[:, py, px], [0, 0, -1], atol=2e-2)
    assert radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))  # max(0.1, mid^2-det) floor (forward.cu:395-398)


def test_opaque_front_gaussian_terminates_the_pixel():
    # (iv): alpha clamps to .99; T*(1-.99f) = 0.00999999, and a second clamped splat gives
    # test_T = 9.99998e-5 < 1e-4 -> `done` before it is blended (forward.cu:565-573): only one contributor.
    s = _single(n=3, zs=[3.0, 4.0, 5.0], opac=[5.0, 5.0, 5.0], scale=0.2)  # op > 1 forces the 0.99 clamp
    o = oracle_for(s)
    o.forward()
    nc = o.get("n_contrib").reshape(2, s.H, s.W)
    assert nc[0, 31, 31] == 1 and nc[1, 31, 31] == 1
    assert o.outputs()[6][0, 31, 31] == np.float32(0.99)
    # with alpha = 0.9 the chain is T = 1, .1, .01, 1e-3: all three blend
    s = _single(n=3, zs=[3.0, 4.0, 5.0], opac=[0.9, 0.9, 0.9], scale=2.0)
    o = oracle_for(s)
    o.forward()
    nc = o.get("n_contrib").reshape(2, s.H, s.W)
    assert nc[0, 31, 31] == 3 and nc[1, 31, 31] == 1


def test_sort_is_stable_in_gaussian_index():
    # (v): identical depth bits keep ascending Gaussian index (stable radix sort of index-ordered emissions)
    s = _single(n=4, zs=[4.0, 4.0, 4.0, 4.0], opac=[0.3] * 4)
    o = oracle_for(s)
    o.forward()
    pl = o.get("point_list")
    rg = o.get("ranges").reshape(-1, 2)
    for a, b in rg:
        if b > a:
            assert pl[a:b].tolist() == sorted(pl[a:b].tolist())


def test_absgrad_column_dominates_signed_gradient():
    # (vii): dL_dmeans2D[:,2] accumulates |.| of the terms whose signed sum (G-path only) is in [:, :2]
    s = make_scene(300, 64, 64, sh_degree=1, mu_px=3.0, seed=3, require_coord=False, require_depth=False)
    o = oracle_for(s)
    o.forward()
    g = {k: torch.randn(c, 64, 64) for k, c in (("color", 3), ("alpha", 1))}
    z3, z1 = torch.zeros(3, 64, 64), torch.zeros(1, 64, 64)
    o.backward(g["color"], z3, z3, z1, z1, g["alpha"], z3)
    d = o.get("dL_dmeans2D").reshape(-1, 3)
    assert np.all(d[:, 2] >= 0)
    assert np.all(np.abs(d[:, 0]) + np.abs(d[:, 1]) <= d[:, 2] * (1 + 1e-4) + 1e-6)
