"""The product's per-Gaussian math (rade-gs_amd/csrc/rg_*.h, host+device source) compiled for the CPU
must agree BIT-FOR-BIT with the oracle: same radii / tile counts (index-determining), same fp32
planes, normals, colours, and the same per-Gaussian backward.  Two independent restatements of the
reference formulas (loop-based glm-style oracle vs. unrolled device code) agreeing to the last bit is
what lets the GPU tests demand exact indices."""
import numpy as np
import pytest
import torch

from hostcheck import hostcheck as hc
from oracle import oracle as orc
from synth_scene import make_scene, upstream_grads
from util import cov3d_of, oracle_for, oracle_backward


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _flat_scene(s, frac=0.3, seed=0):
    """squash one axis of a fraction of the Gaussians to ~0: exercises the ill-conditioned
    (lambda_min <= 1e-8) branch and the eigen-solver's early exits"""
    gen = torch.Generator().manual_seed(seed)
    sc = s.scales.clone()
    pick = torch.rand(sc.shape[0], generator=gen) < frac
    axis = torch.randint(0, 3, (sc.shape[0],), generator=gen)
    sc[pick, axis[pick]] = 1e-6
    return s._replace(scales=sc)


CASES = [dict(P=6000, W=256, H=256, sh_degree=3, mu_px=1.5, seed=0, kernel_size=0.0, pose="identity"),
         dict(P=6000, W=320, H=200, sh_degree=3, mu_px=4.0, seed=7, kernel_size=0.1, pose="random"),
         dict(P=3000, W=128, H=96, sh_degree=1, mu_px=20.0, seed=9, kernel_size=0.1, pose="random"),
         dict(P=3000, W=200, H=120, sh_degree=2, mu_px=3.0, seed=11, kernel_size=0.0, pose="random", flat=True)]


@pytest.mark.parametrize("case", CASES)
def test_preprocess_forward_and_backward_bit_exact(case):
    case = dict(case)
    flat = case.pop("flat", False)
    s = make_scene(require_coord=True, require_depth=True, **case)
    if flat:
        s = _flat_scene(s)
    P = s.means3D.shape[0]
    o = oracle_for(s)
    o.forward()
    f, i = hc.preprocess_fwd(s)
    radii = o.get("radii")
    vis = radii > 0
    assert vis.sum() > P // 2
    assert np.array_equal(radii, i[:, 0])
    assert np.array_equal(o.get("tiles_touched"), i[:, 1].astype(np.uint32))
    co = o.get("conic_opacity", (P, 4))
    pairs = {"means2D": (o.get("means2D", (P, 2)), f[:, 0:2]), "conic_opacity": (co, f[:, 2:6]), "ts": (o.get("ts"), f[:, 6]),
             "rgb": (o.get("rgb", (P, 3)), f[:, 7:10]), "ray_planes": (o.get("ray_planes", (P, 2)), f[:, 10:12]),
             "normals": (o.get("normals", (P, 3)), f[:, 12:15]), "camera_planes": (o.get("camera_planes", (P, 6)), f[:, 15:21]),
             "view_points": (o.get("view_points", (P, 3)), f[:, 21:24]), "depths": (o.get("depths"), f[:, 24])}
    for k, (a, b) in pairs.items():
        assert np.array_equal(bits(a[vis]), bits(b[vis])), k
    # the packed tile rectangle the instance emission consumes: w*h = tiles touched, origin = getRect() of the oracle's mean/radius
    rect = np.ascontiguousarray(f[:, 25]).view(np.uint32)
    x0, y0, w, h = rect & 255, (rect >> 8) & 255, (rect >> 16) & 255, rect >> 24
    assert np.array_equal((w * h)[vis], o.get("tiles_touched")[vis])
    m2 = o.get("means2D", (P, 2))
    gx, gy = (s.W + 15) // 16, (s.H + 15) // 16
    ex0 = np.clip(((m2[:, 0] - radii) / 16).astype(np.int64), 0, gx)          # auxiliary.h getRect: (int) truncation, then clamp
    ey0 = np.clip(((m2[:, 1] - radii) / 16).astype(np.int64), 0, gy)
    assert np.array_equal(x0[vis], ex0[vis]) and np.array_equal(y0[vis], ey0[vis])
    cl = o.get("clamped").reshape(P, 3)
    clb = (cl[:, 0] + 2 * cl[:, 1] + 4 * cl[:, 2]).astype(np.int32)
    assert np.array_equal(clb[vis], i[vis, 2])

    g = upstream_grads(s, case["seed"])
    for intended in (False, True):   # include/radegs.h::opacity_grad_intended: the reference's executed backward / the intended one
        orc.set_opacity_slip(0 if intended else 1)
        try:
            gr = oracle_backward(o, g)
        finally:
            orc.set_opacity_slip(1)
        _check_bwd(s, o, gr, radii, clb, co, P, intended)


def _check_bwd(s, o, gr, radii, clb, co, P, intended):
    acc = np.zeros((P, 25), np.float32)
    acc[:, 0:3] = o.get("acc_dcolors", (P, 3)); acc[:, 3] = o.get("dL_dts"); acc[:, 4:6] = o.get("dL_dray_planes", (P, 2))
    acc[:, 6:9] = o.get("dL_dnormals", (P, 3)); acc[:, 9:12] = o.get("acc_dmeans2D", (P, 3))
    dc = o.get("acc_dconic", (P, 4)); acc[:, 12] = dc[:, 0]; acc[:, 13] = dc[:, 1]; acc[:, 14] = dc[:, 3]
    acc[:, 15] = o.get("acc_dopacity"); acc[:, 16:19] = o.get("dL_dview_points", (P, 3)); acc[:, 19:25] = o.get("dL_dcamera_planes", (P, 6))
    out, dsh = hc.preprocess_bwd(s, radii, clb, co[:, 3] if intended else dc[:, 3], acc)
    for k, (a, b) in {"dL_dmeans3D": (gr["dL_dmeans3D"], out[:, 0:3]), "dL_dopacity": (gr["dL_dopacity"][:, 0], out[:, 3]),
                      "dL_dcov3D": (gr["dL_dcov3D"], out[:, 4:10]), "dL_dscales": (gr["dL_dscales"], out[:, 10:13]),
                      "dL_drotations": (gr["dL_drotations"], out[:, 13:17]), "dL_dsh": (gr["dL_dsh"], dsh)}.items():
        assert np.array_equal(bits(a), bits(b)), k


@pytest.mark.parametrize("flat", [False, True])
def test_integrate_preprocess_bit_exact(flat):
    """INTE branch of computeCov2D (forward.cu:187-235): inverse ray-space covariance + conditioning flag."""
    s = make_scene(3000, 200, 120, sh_degree=1, mu_px=3.0, seed=21, kernel_size=0.0, pose="random", require_coord=False, require_depth=True)
    if flat:
        s = _flat_scene(s, frac=0.5, seed=3)
    P = s.means3D.shape[0]
    o = oracle_for(s)
    o.integrate(s.means3D.numpy()[:10])
    f, r = hc.preprocess_inte(s)
    radii = o.get("radii")
    assert np.array_equal(radii, r)
    vis = radii > 0
    cond = o.get("condition")
    assert np.array_equal(cond[vis], f[vis, 6].astype(np.uint8))
    if flat:
        assert (cond[vis] == 0).sum() > 100
    icr = o.get("invraycov", (P, 6))
    assert np.array_equal(bits(icr[vis]), bits(f[vis, :6]))


def test_precomputed_covariance_and_colors_bit_exact():
    s = make_scene(4000, 160, 120, sh_degree=0, mu_px=2.0, seed=5, kernel_size=0.1, pose="random", require_coord=True, require_depth=True)
    P = s.means3D.shape[0]
    cov, colors = cov3d_of(s), torch.rand(P, 3)
    o = oracle_for(s, colors=colors, cov3D=cov)
    o.forward()
    f, i = hc.preprocess_fwd(s, colors=colors, cov3D=cov)
    vis = o.get("radii") > 0
    assert np.array_equal(o.get("radii"), i[:, 0])
    assert np.array_equal(bits(colors.numpy()[vis]), bits(f[vis, 7:10]))  # precomputed colours pass through unclamped
    assert np.array_equal(bits(o.get("conic_opacity", (P, 4))[vis]), bits(f[vis, 2:6]))


def test_decision_chain_pieces_bit_exact():
    rng = np.random.default_rng(0)
    L = hc.lib()
    for x in np.concatenate([np.linspace(-90, 0.5, 4001), -np.logspace(-8, 1.9, 500)]).astype(np.float32):
        assert np.float32(orc.exp_spec(float(x))).tobytes() == np.float32(L.hc_exp_spec(float(x))).tobytes()
    # the product reaches k = rint(x log2e) through a magic-number addition (no v_rndne / v_cvt): the same bits as the rintf form on EVERY
    # float in [-87, -0] and [+0, 16] (dense: every 3rd bit pattern of the negative range, ~0.36 G evaluations in the full sweep below)
    neg0, neg87, pos16 = 0x80000000, int(np.float32(-87.0).view(np.uint32)), int(np.float32(16.0).view(np.uint32))
    assert L.hc_exp_spec_sweep(neg0, neg87, 97) == 0
    assert L.hc_exp_spec_sweep(0, pos16, 97) == 0
    assert L.hc_exp_spec_sweep(int(np.float32(-88.5).view(np.uint32)) - 4096, int(np.float32(-88.5).view(np.uint32)), 1) == 0   # below -87: 0
    # power = -0.5*(cx*dx*dx + cz*dy*dy) - cy*dx*dy, rounded per operation in source order (forward.cu:555)
    for _ in range(5000):
        cx, cy, cz, dx, dy = (np.float32(v) for v in rng.normal(size=5) * [1, .5, 1, 8, 8])
        ref = np.float32(np.float32(-0.5) * (np.float32(np.float32(cx * dx) * dx) + np.float32(np.float32(cz * dy) * dy))) - \
            np.float32(np.float32(cy * dx) * dy)
        assert np.float32(ref).tobytes() == np.float32(L.hc_splat_power(cx, cy, cz, dx, dy)).tobytes()
    # the skip threshold is conservative: anything below it has alpha < 1/255 under the exact rule
    for op in (1e-3, 0.02, 0.3, 0.9, 1.0, 7.0):
        thr = L.hc_skip_threshold(op)
        for eps in (1e-6, 1e-4, 1e-2, 1.0):
            a = min(0.99, np.float32(op) * np.float32(orc.exp_spec(float(np.float32(thr - eps)))))
            assert a < 1.0 / 255.0
