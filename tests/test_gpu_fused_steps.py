"""GPU parity of the fused pre/post steps against (a) the golden vectors produced by the reference's
own Python on the CPU (tests/golden/normals_*.npz, filter3d.npz) and (b) the numpy oracles at larger sizes.
Tolerance: 1e-5 abs / 1e-4 rel on values; gradients of normalize(cross(.)) are compared at the fp32 noise level of the
reference's own autograd (the golden gradients are fp32 too)."""
import os
from collections import namedtuple

import numpy as np
import pytest
import torch

from oracle import filter3d_oracle as fo
from oracle import normal_oracle as no

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
View = namedtuple("View", "image_width image_height FoVx FoVy")


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda:0").requires_grad_(grad)


def _close(a, b, rtol=1e-4, atol=1e-5):
    return np.abs(a - b) <= atol + rtol * np.abs(b)


@pytest.mark.parametrize("name", ["depth_smooth", "depth_rough", "points"])
def test_normals_and_loss_match_reference_golden(name):
    import graphics_utils as gu
    assert torch.cuda.is_available()
    z = np.load(os.path.join(GOLD, f"normals_{name}.npz"))
    W, H = int(z["W"]), int(z["H"])
    view = View(W, H, float(z["fovx"]), float(z["fovy"]))
    pts = "points1" in z.files
    m1, m2 = (_t(z["points1"], True), _t(z["points2"], True)) if pts else (_t(z["depth1"], True), _t(z["depth2"], True))
    rn = _t(z["rendered_normal"], True)
    fn = gu.point_double_to_normal if pts else gu.depth_double_to_normal
    nm = fn(view, m1, m2)
    assert np.abs(nm.detach().cpu().numpy() - z["normals"]).max() < 2e-5
    # un-fused: the reference's loss expression on our normal maps, autograd through the HIP backward
    err = 1 - (rn.unsqueeze(0) * nm).sum(dim=1)
    loss = 0.4 * err[0].mean() + 0.6 * err[1].mean()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 2e-6
    grads_unfused = [t.grad.detach().cpu().numpy().copy() for t in (m1, m2, rn)]
    for t in (m1, m2, rn):
        t.grad = None
    # fused
    lf = gu.normal_consistency_loss(view, rn, m1, m2, 0.6, points=pts)
    lf.backward()
    assert abs(lf.item() - float(z["loss"])) < 2e-6
    grads_fused = [t.grad.detach().cpu().numpy() for t in (m1, m2, rn)]
    for got in (grads_unfused, grads_fused):
        for a, k in zip(got, ("g1", "g2", "g_rendered")):
            ref = z[k].reshape(a.shape)
            scale = np.abs(ref).max()
            assert np.abs(a - ref).max() < 2e-3 * scale, (k, np.abs(a - ref).max(), scale)
            assert np.median(np.abs(a - ref)) < 1e-5 * scale, k
    # generic cotangent
    for t in (m1, m2):
        t.grad = None
    (fn(view, m1, m2) * _t(z["cot"])).sum().backward()
    for t, k in ((m1, "c1"), (m2, "c2")):
        ref = z[k].reshape(t.shape)
        a = t.grad.cpu().numpy()
        assert np.abs(a - ref).max() < 2e-3 * np.abs(ref).max() and np.median(np.abs(a - ref)) < 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("points", [False, True])
def test_normal_loss_full_hd_against_oracle(points):
    import graphics_utils as gu
    W, H = 1920, 1080
    rng = np.random.default_rng(5)
    fovx = 1.0
    fovy = 2 * np.arctan(np.tan(fovx / 2) * H / W)
    view = View(W, H, fovx, fovy)
    y, x = np.mgrid[0:H, 0:W]
    d1 = (4 + np.sin(x / 40.0) * np.cos(y / 25.0) + 0.01 * rng.standard_normal((H, W))).astype(np.float32)
    d2 = (d1 + 0.03 * np.cos(x / 13.0)).astype(np.float32)
    rn = rng.standard_normal((3, H, W)).astype(np.float32)
    rn /= np.linalg.norm(rn, axis=0, keepdims=True)
    if points:
        p1, p2 = no.depths_to_points(d1, d2, W, H, fovx, fovy)
        maps64 = np.stack([p1, p2], 0).astype(np.float64)
        m1, m2 = _t(p1, True), _t(p2, True)
    else:
        q1, q2 = no.depths_to_points(d1.astype(np.float64), d2.astype(np.float64), W, H, fovx, fovy)
        maps64 = np.stack([q1, q2], 0)
        m1, m2 = _t(d1.reshape(1, H, W), True), _t(d2.reshape(1, H, W), True)
    r = _t(rn, True)
    loss = gu.normal_consistency_loss(view, r, m1, m2, 0.6, points=points)
    (loss * 3.0).backward()          # non-unit upstream gradient
    nm64, _ = no.points_to_normal(maps64)
    ref_loss = no.consistency_loss(rn.astype(np.float64), nm64)
    assert abs(loss.item() - ref_loss) < 1e-5
    g_rn, g_nm = no.consistency_loss_bwd(rn.astype(np.float64), nm64, upstream=3.0)
    # N is a normalised cross product of fp32 differences of depth ~4: abs error ~1e-5 of unit length
    assert _close(r.grad.cpu().numpy(), g_rn, atol=1e-4 * np.abs(g_rn).max()).all()
    gp = no.points_to_normal_bwd(maps64, g_nm)
    if points:
        refs = (gp[0], gp[1])
    else:
        ray = no.rays(W, H, fovx, fovy, np.float64)
        refs = ((gp[0] * ray).sum(0)[None], (gp[1] * ray).sum(0)[None])
    for t, ref in zip((m1, m2), refs):
        a = t.grad.cpu().numpy()
        scale = np.abs(ref).max()
        assert np.median(np.abs(a - ref)) < 1e-5 * scale
        assert (np.abs(a - ref) < 1e-2 * scale).mean() > 0.9999    # fp32 cancellation in (P(y+1)-P(y-1)) x (...) at a few pixels
    # determinism (no atomics anywhere)
    l2 = gu.normal_consistency_loss(view, r.detach(), m1.detach(), m2.detach(), 0.6, points=points)
    assert l2.item() == loss.item()


def test_filter3d_matches_reference_golden_and_oracle():
    import gaussian_model_ops as gmo
    z = np.load(os.path.join(GOLD, "filter3d.npz"))
    sc, op, f3 = _t(z["scaling_raw"], True), _t(z["opacity_raw"], True), _t(z["filter_3D"])
    s, o = gmo.scaling_n_opacity_with_3D_filter(sc, op, f3)
    assert np.allclose(s.detach().cpu().numpy(), z["scales"], rtol=1e-5, atol=1e-12)
    assert np.allclose(o.detach().cpu().numpy(), z["opacity"], rtol=1e-4, atol=1e-9)
    ((s * _t(z["cot_scales"])).sum() + (o * _t(z["cot_opacity"])).sum()).backward()
    for t, k in ((sc, "g_scaling_raw"), (op, "g_opacity_raw")):
        ref = z[k]
        assert np.allclose(t.grad.cpu().numpy(), ref, rtol=3e-4, atol=1e-6 * np.abs(ref).max()), k
    # 1M Gaussians against the float64 oracle
    rng = np.random.default_rng(1)
    P = 1_000_000
    a = (np.log(0.01) + rng.standard_normal((P, 3))).astype(np.float32)
    b = (2 * rng.standard_normal((P, 1))).astype(np.float32)
    c = (0.001 + 0.02 * rng.random((P, 1))).astype(np.float32)
    cs, co = rng.standard_normal((P, 3)).astype(np.float32), rng.standard_normal((P, 1)).astype(np.float32)
    ta, tb = _t(a, True), _t(b, True)
    s, o = gmo.scaling_n_opacity_with_3D_filter(ta, tb, _t(c))
    rs, ro = fo.forward(a.astype(np.float64), b.astype(np.float64), c.astype(np.float64))
    assert np.allclose(s.detach().cpu().numpy(), rs, rtol=1e-5) and np.allclose(o.detach().cpu().numpy(), ro, rtol=1e-4, atol=1e-9)
    ((s * _t(cs)).sum() + (o * _t(co)).sum()).backward()
    gs, go = fo.backward(a.astype(np.float64), b.astype(np.float64), c.astype(np.float64), cs.astype(np.float64), co.astype(np.float64))
    assert np.allclose(ta.grad.cpu().numpy(), gs, rtol=1e-4, atol=1e-6 * np.abs(gs).max())
    assert np.allclose(tb.grad.cpu().numpy(), go, rtol=1e-4, atol=1e-6 * np.abs(go).max())
    # only one cotangent present (the other output unused downstream)
    ta.grad = None
    s, o = gmo.scaling_n_opacity_with_3D_filter(ta, tb, _t(c))
    s.sum().backward()
    gs1, _ = fo.backward(a.astype(np.float64), b.astype(np.float64), c.astype(np.float64), np.ones_like(cs, dtype=np.float64), np.zeros((P, 1)))
    assert np.allclose(ta.grad.cpu().numpy(), gs1, rtol=1e-4, atol=1e-6 * np.abs(gs1).max())


@pytest.mark.parametrize("name", ["small", "wide"])
def test_photometric_loss_matches_reference_golden(name):
    import loss_utils as lu
    from oracle import loss_oracle as lo
    z = np.load(os.path.join(GOLD, f"losses_{name}.npz"))
    img, gt = _t(z["img"], True), _t(z["gt"])
    loss = lu.photometric_loss(img, gt.unsqueeze(0), 0.2)       # gt with the batch dimension train.py adds
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 2e-6
    ref = z["grad"]
    assert np.abs(img.grad.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
    # the un-fused names
    assert abs(lu.l1_loss(img.detach(), gt).item() - float(z["l1"])) < 1e-6
    assert abs(lu.ssim(img.detach(), gt).item() - float(z["ssim"])) < 5e-6
    # composing the two separately gives the same gradient as the fused call
    img.grad = None
    (0.8 * lu.l1_loss(img, gt) + 0.2 * (1.0 - lu.ssim(img, gt))).backward()
    assert np.abs(img.grad.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()


def test_photometric_loss_full_hd_against_oracle():
    import loss_utils as lu
    from oracle import loss_oracle as lo
    H, W = 1080, 1920
    rng = np.random.default_rng(9)
    y, x = np.mgrid[0:H, 0:W]
    gt = np.stack([0.5 + 0.4 * np.sin(x / 9.0 + c) * np.cos(y / 7.0 - c) for c in range(3)], 0).astype(np.float32)
    img = np.clip(gt + 0.1 * rng.standard_normal((3, H, W)), 0, 1).astype(np.float32)
    a = _t(img, True)
    loss = lu.photometric_loss(a, _t(gt), 0.2)
    (2.0 * loss).backward()
    assert abs(loss.item() - lo.rgb_loss(img.astype(np.float64), gt.astype(np.float64), 0.2)) < 2e-6
    ref = lo.rgb_loss_bwd(img.astype(np.float64), gt.astype(np.float64), 0.2, upstream=2.0)
    assert np.abs(a.grad.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
    assert lu.photometric_loss(a.detach(), _t(gt), 0.2).item() == loss.item()     # deterministic


def test_fused_adam_matches_torch_golden_and_survives_state_surgery():
    import fused_adam
    from oracle import adam_oracle as ao
    z = np.load(os.path.join(GOLD, "adam.npz"))
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}
    params = {k: torch.nn.Parameter(_t(z[f"p0_{k}"])) for k in lrs}
    opt = fused_adam.Adam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in lrs], lr=0.0, eps=1e-15)
    for it in range(3):
        for k, p in params.items():
            p.grad = _t(z[f"g{it}_{k}"])
        if it == 2:
            opt.param_groups[0]["lr"] = 1.0e-4
        opt.step()
        for k, p in params.items():
            ref = z[f"p{it + 1}_{k}"]
            assert np.abs(p.detach().cpu().numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()) + 2e-7, (k, it)
    for k, p in params.items():
        st = opt.state[p]
        assert float(st["step"]) == 3.0
        assert np.allclose(st["exp_avg"].cpu().numpy(), z[f"m_{k}"], rtol=1e-5, atol=1e-12)
        assert np.allclose(st["exp_avg_sq"].cpu().numpy(), z[f"v_{k}"], rtol=1e-5, atol=1e-20)
    # the reference's cat_tensors_to_optimizer surgery (scene/gaussian_model.py:615-633) on our optimizer, then one more step
    g = opt.param_groups[3]
    old = g["params"][0]
    ext = torch.zeros(7, 1, device="cuda:0")
    st = opt.state.get(old)
    st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
    st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
    del opt.state[old]
    g["params"][0] = torch.nn.Parameter(torch.cat((old.detach(), ext), dim=0).requires_grad_(True))
    opt.state[g["params"][0]] = st
    newp = g["params"][0]
    before = newp.detach().cpu().numpy().copy()
    grad = np.random.default_rng(3).standard_normal((1007, 1)).astype(np.float32)
    newp.grad = _t(grad)
    m0, v0 = st["exp_avg"].cpu().numpy().copy(), st["exp_avg_sq"].cpu().numpy().copy()
    for k, p in params.items():
        p.grad = None
    opt.step()
    ref, _, _ = ao.step(before, grad, m0, v0, 4, 0.05)
    assert np.abs(newp.detach().cpu().numpy() - ref).max() < 1e-6
    # odd sizes / unaligned tails / 1M x 59 floats against the oracle
    rng = np.random.default_rng(4)
    shapes = [(1_000_003, 3), (1_000_003, 15, 3), (5,), (1_000_003, 1)]
    ps = [torch.nn.Parameter(_t(rng.standard_normal(s))) for s in shapes]
    opt = fused_adam.Adam([{"params": [p], "lr": 1e-3 * (i + 1)} for i, p in enumerate(ps)], lr=0.0, eps=1e-15)
    gs = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    p0 = [p.detach().cpu().numpy().copy() for p in ps]
    for p, g_ in zip(ps, gs):
        p.grad = _t(g_)
    opt.step()
    for i, (p, g_, q) in enumerate(zip(ps, gs, p0)):
        ref, _, _ = ao.step(q, g_, np.zeros_like(q), np.zeros_like(q), 1, 1e-3 * (i + 1))
        assert np.abs(p.detach().cpu().numpy() - ref).max() < 1e-6


def test_compute_3D_filter_matches_reference_golden():
    import gaussian_model_ops as gmo
    from test_filter3d_oracle import cameras_from
    z = np.load(os.path.join(GOLD, "filter3d.npz"))
    cams = cameras_from(z)
    out = gmo.compute_3D_filter(_t(z["xyz"]), cams).cpu().numpy()
    ref = z["filter_out"]
    assert out.shape == ref.shape
    assert (np.abs(out - ref) <= 1e-5 * np.abs(ref)).mean() > 0.999
    # 1M Gaussians, 150 cameras against the numpy oracle
    rng = np.random.default_rng(2)
    xyz = (rng.standard_normal((1_000_000, 3)) * 3.0).astype(np.float32)
    many = [cams[i % len(cams)]._replace(T=cams[i % len(cams)].T + 0.01 * i) for i in range(150)]
    out = gmo.compute_3D_filter(_t(xyz), many).cpu().numpy()
    ref = fo.compute_3D_filter(xyz, many)
    assert (np.abs(out - ref) <= 1e-5 * np.abs(ref)).mean() > 0.999


@pytest.mark.parametrize("W,H", [(1, 1), (2, 3), (5, 7), (11, 11), (16, 64), (64, 16), (65, 17), (130, 33), (257, 5)])
def test_losses_on_odd_image_sizes(W, H):
    """Images smaller than the 11-tap SSIM window / the 3x3 normal stencil, and sizes off the 64x16 kernel tiles."""
    import graphics_utils as gu
    import loss_utils as lu
    from oracle import loss_oracle as lo
    rng = np.random.default_rng(W * 1000 + H)
    gt = rng.random((3, H, W)).astype(np.float32)
    img = np.clip(gt + 0.1 * rng.standard_normal((3, H, W)), 0, 1).astype(np.float32)
    a = _t(img, True)
    loss = lu.photometric_loss(a, _t(gt), 0.2)
    loss.backward()
    assert abs(loss.item() - lo.rgb_loss(img.astype(np.float64), gt.astype(np.float64), 0.2)) < 5e-6
    ref = lo.rgb_loss_bwd(img.astype(np.float64), gt.astype(np.float64), 0.2)
    assert np.abs(a.grad.cpu().numpy() - ref).max() < 2e-4 * np.abs(ref).max() + 1e-9
    # normals + consistency loss
    fovx = 1.0
    fovy = 2 * np.arctan(np.tan(fovx / 2) * H / W)
    view = View(W, H, fovx, float(fovy))
    d1 = (3 + rng.random((1, H, W))).astype(np.float32)
    d2 = (3 + rng.random((1, H, W))).astype(np.float32)
    rn = rng.standard_normal((3, H, W)).astype(np.float32)
    t1, t2, tr = _t(d1, True), _t(d2, True), _t(rn, True)
    nm = gu.depth_double_to_normal(view, t1, t2)
    ref_nm = no.depth_double_to_normal(d1.astype(np.float64), d2.astype(np.float64), W, H, fovx, float(fovy))
    assert np.abs(nm.detach().cpu().numpy() - ref_nm).max() < 5e-5
    l2 = gu.normal_consistency_loss(view, tr, t1, t2, 0.6)
    l2.backward()
    assert abs(l2.item() - no.consistency_loss(rn.astype(np.float64), ref_nm)) < 1e-5
    for t in (t1, t2, tr):
        assert torch.isfinite(t.grad).all()
    g_rn, g_nm = no.consistency_loss_bwd(rn.astype(np.float64), ref_nm)
    assert np.abs(tr.grad.cpu().numpy() - g_rn).max() < 1e-4 * (np.abs(g_rn).max() + 1e-12) + 1e-9
    gd1, gd2 = no.depth_double_to_normal_bwd(d1.astype(np.float64), d2.astype(np.float64), W, H, fovx, float(fovy), g_nm)
    for t, ref_g in ((t1, gd1), (t2, gd2)):
        sc = np.abs(ref_g).max() + 1e-12
        assert np.median(np.abs(t.grad.cpu().numpy() - ref_g)) < 1e-4 * sc
        assert np.abs(t.grad.cpu().numpy() - ref_g).max() < 5e-2 * sc


@pytest.mark.parametrize("P,kind", [(4, "uniform"), (100, "uniform"), (5000, "clustered"), (300_000, "uniform"), (200_000, "surface")])
def test_simple_knn_distcuda2(P, kind):
    """simple_knn._C.distCUDA2 against an exact k-d tree (oracle/knn_oracle.py)."""
    from oracle import knn_oracle
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(P)
    if kind == "uniform":
        pts = rng.random((P, 3))
    elif kind == "clustered":
        pts = rng.standard_normal((P, 3)) * 0.01 + rng.integers(0, 5, (P, 1)) * 3.0
        pts[:50] = pts[0]                                   # duplicates: zero distances
    else:
        uv = rng.random((P, 2))
        pts = np.stack([uv[:, 0] * 4, uv[:, 1] * 2, 0.1 * np.sin(6 * uv[:, 0]) * np.cos(5 * uv[:, 1])], 1)
    pts = pts.astype(np.float32)
    got = distCUDA2(_t(pts)).cpu().numpy()
    ref = knn_oracle.mean_dist2_3nn(pts)
    assert got.shape == (P,) and np.isfinite(got).all()
    assert np.allclose(got, ref, rtol=2e-4, atol=1e-12 + 1e-6 * ref.max())
