"""Drop-in proof through the REFERENCE'S OWN, UNMODIFIED caller.

`/root/reference/gaussian_renderer/__init__.py` (`render()` :19-95, `integrate()` :98-195) is imported as it is and run
against THIS repository's `diff_gaussian_rasterization` package, with a real reference `GaussianModel` (its activation /
3D-filter properties run under torch autograd) and a camera object carrying the attributes the reference reads.

The build container has no GPU and the operator has no CPU path, so the three native entry points of
`diff_gaussian_rasterization._C` are replaced -- in this test only -- by shims over the CPU oracle (test infrastructure).
Everything ABOVE the C ABI is therefore the code that ships: `GaussianRasterizationSettings`, `GaussianRasterizer.forward /
.integrate`, the argument reordering and the 8-tuple / 9-gradient plumbing of `_RasterizeGaussians`.  What is checked:
  * the unmodified `render()` / `integrate()` run to completion and return their dictionaries;
  * every entry equals what the oracle computes for the very same inputs (so each output sits in the right slot);
  * `loss.backward()` through `render()` reaches the model's raw parameters with the gradients the oracle gives for the
    rasterizer inputs, pushed through the reference's own activations (so each of the 9 gradients sits in the right slot).
The same `_C` entry points are compared kernel-against-oracle on the MI355X in tests/test_gpu_*.py (the reference tree does not
exist on the GPU box, which is why this test lives on the CPU side).
"""
import math
import os
import sys
import types
from collections import namedtuple

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "gaussian_renderer", "__init__.py")),
                                reason="the reference tree is only present in the build container")


def _import_reference():
    for name, attrs in {"plyfile": ("PlyData", "PlyElement"), "simple_knn": (), "simple_knn._C": ("distCUDA2",), "trimesh": (), "cv2": ()}.items():
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules.setdefault(name, m)
    if REF not in sys.path:
        sys.path.append(REF)          # after the repo's own paths: `diff_gaussian_rasterization` must resolve to THIS package
    if "scene" not in sys.modules:   # keep scene/__init__.py (dataset readers, PIL, ...) from running
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        sys.modules["scene"] = pkg
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gaussian_renderer", os.path.join(REF, "gaussian_renderer", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)     # the file is executed exactly as it is on disk
    from scene.gaussian_model import GaussianModel
    return mod, GaussianModel


Camera = namedtuple("Camera", "FoVx FoVy image_height image_width world_view_transform full_proj_transform camera_center")
Pipe = namedtuple("Pipe", "debug compute_cov3D_python convert_SHs_python")


def _model_and_camera(GaussianModel, P=1500, W=96, H=64, seed=7):
    from synth_scene import make_scene
    s = make_scene(P, W, H, sh_degree=3, mu_px=3.0, seed=seed, kernel_size=0.0, require_coord=True, require_depth=True, pose="random",
                   filter3d=False)
    rng = np.random.default_rng(seed)
    gm = object.__new__(GaussianModel)     # __init__ allocates CUDA tensors; only the activations are needed
    gm.setup_functions()
    gm.active_sh_degree, gm.max_sh_degree = 3, 3
    gm._xyz = s.means3D.clone().requires_grad_(True)
    gm._features_dc = s.shs[:, :1].clone().contiguous().requires_grad_(True)
    gm._features_rest = s.shs[:, 1:].clone().contiguous().requires_grad_(True)
    gm._scaling = torch.log(s.scales).clone().requires_grad_(True)
    gm._rotation = (s.rotations * torch.from_numpy(rng.uniform(0.5, 2.0, (P, 1)).astype(np.float32))).requires_grad_(True)  # un-normalised
    op = s.opacities.clamp(1e-4, 1 - 1e-4)
    gm._opacity = torch.log(op / (1 - op)).clone().requires_grad_(True)
    gm.filter_3D = torch.from_numpy((0.002 + 0.01 * rng.random((P, 1))).astype(np.float32))
    cam = Camera(2 * math.atan(s.tanfovx), 2 * math.atan(s.tanfovy), H, W, s.viewmatrix, s.projmatrix, s.campos)
    return gm, cam, s


class _OracleNative:
    """CPU stand-ins for the three `_C` entry points, over the oracle; they keep what they were called with."""

    def __init__(self):
        self.calls = []

    @staticmethod
    def _opt(t):
        return None if t is None or t.numel() == 0 else t.detach()

    def _oracle(self, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx, tanfovy,
                kernel_size, H, W, sh, degree, campos, require_coord, require_depth):
        from oracle.oracle import Oracle
        return Oracle(bg=bg, means3D=means3D.detach(), opacities=opacity.detach(), viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos,
                      tanfovx=tanfovx, tanfovy=tanfovy, image_height=H, image_width=W, shs=self._opt(sh), colors_precomp=self._opt(colors),
                      scales=self._opt(scales), rotations=self._opt(rotations), cov3D_precomp=self._opt(cov3D), sh_degree=degree,
                      scale_modifier=scale_modifier, kernel_size=kernel_size, require_coord=require_coord, require_depth=require_depth)

    def rasterize_gaussians(self, bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx,
                            tanfovy, kernel_size, H, W, sh, degree, campos, prefiltered, require_coord, require_depth, debug):
        self.calls.append("forward")
        o = self._oracle(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx, tanfovy,
                         kernel_size, H, W, sh, degree, campos, require_coord, require_depth)
        R = o.forward()
        color, radii, coord, mcoord, depth, mdepth, alpha, normal = [torch.from_numpy(np.array(x)) for x in o.outputs()]
        self.fwd_args = (bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx, tanfovy,
                         kernel_size, H, W, sh, degree, campos, require_coord, require_depth)
        e = torch.empty(0, dtype=torch.uint8)
        return R, color, coord, mcoord, alpha, normal, depth, mdepth, radii, e, e, e

    def rasterize_gaussians_backward(self, bg, means3D, radii, colors, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx,
                                     tanfovy, kernel_size, g_color, g_coord, g_mcoord, g_depth, g_mdepth, g_alpha, g_normal, normalmap, sh,
                                     degree, campos, geom, R, binning, img, alphas, require_coord, require_depth, debug):
        self.calls.append("backward")
        H, W = g_color.shape[1], g_color.shape[2]
        opacity = self.fwd_args[3]
        o = self._oracle(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx, tanfovy,
                         kernel_size, H, W, sh, degree, campos, require_coord, require_depth)
        assert o.forward() == R
        o.backward(g_color, g_coord, g_mcoord, g_depth, g_mdepth, g_alpha, g_normal)
        g = o.grads()
        self.bwd_grads = g
        t = {k: torch.from_numpy(np.array(v)) for k, v in g.items()}
        return (t["dL_dmeans2D"], t["dL_dcolors"], t["dL_dopacity"], t["dL_dmeans3D"], t["dL_dcov3D"], t["dL_dsh"], t["dL_dscales"],
                t["dL_drotations"])

    def integrate_gaussians_to_points(self, bg, points3D, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, v2g, viewmatrix,
                                      projmatrix, tanfovx, tanfovy, kernel_size, subpixel_offset, H, W, sh, degree, campos, prefiltered, debug):
        self.calls.append("integrate")
        o = self._oracle(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx, tanfovy,
                         kernel_size, H, W, sh, degree, campos, False, False)
        R = o.integrate(points3D.detach())
        PN = points3D.shape[0]
        out = [torch.from_numpy(np.array(x)) for x in (o.get("out9", (9, H, W)), o.get("out_alpha_integrated", (PN,)),
                                                       o.get("out_color_integrated", (PN, 3)), o.get("out_coordinate2d", (PN, 2)),
                                                       o.get("out_sdf", (PN,)), o.get("radii"))]
        self.integrate_out = out
        e = torch.empty(0, dtype=torch.uint8)
        return (R, *out, e, e, e)


@pytest.fixture()
def wired(monkeypatch):
    ref, GaussianModel = _import_reference()
    import diff_gaussian_rasterization as dgr
    assert os.path.dirname(dgr.__file__).endswith(os.path.join("rade-gs_amd", "diff_gaussian_rasterization")), dgr.__file__
    assert ref.GaussianRasterizer is dgr.GaussianRasterizer and ref.GaussianRasterizationSettings is dgr.GaussianRasterizationSettings
    native = _OracleNative()
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "integrate_gaussians_to_points"):
        monkeypatch.setattr(dgr._C, name, getattr(native, name))
    # the reference creates `screenspace_points` with device="cuda"; this container's torch has no CUDA
    real_zeros_like = torch.zeros_like

    def zeros_like(t, **kw):
        if kw.get("device") == "cuda":
            kw["device"] = t.device
        return real_zeros_like(t, **kw)

    monkeypatch.setattr(torch, "zeros_like", zeros_like)
    return ref, GaussianModel, native


def test_reference_render_runs_on_this_operator_and_every_slot_is_right(wired):
    ref, GaussianModel, native = wired
    gm, cam, s = _model_and_camera(GaussianModel)
    bg = torch.tensor([0.1, 0.4, 0.7])
    out = ref.render(cam, gm, Pipe(False, False, False), bg, kernel_size=0.1, require_coord=True, require_depth=True)
    assert native.calls == ["forward"]
    assert set(out) == {"render", "mask", "expected_coord", "median_coord", "expected_depth", "median_depth", "viewspace_points",
                        "visibility_filter", "radii", "normal"}
    # the rasterizer saw the reference's ACTIVATED parameters (3D filter applied by the reference's own property)
    scales_ref, opacity_ref = gm.get_scaling_n_opacity_with_3D_filter
    fa = native.fwd_args
    assert torch.equal(fa[4], scales_ref.detach()) and torch.equal(fa[3], opacity_ref.detach()) and torch.equal(fa[5], gm.get_rotation.detach())
    assert torch.equal(fa[15], gm.get_features.detach()) and fa[6] == 1.0 and fa[12] == 0.1 and fa[16] == 3
    # every dictionary entry is the oracle's output of the same name
    from oracle.oracle import Oracle
    o = Oracle(bg=bg, means3D=gm.get_xyz.detach(), opacities=opacity_ref.detach(), viewmatrix=s.viewmatrix, projmatrix=s.projmatrix,
               campos=s.campos, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), image_height=cam.image_height,
               image_width=cam.image_width, shs=gm.get_features.detach(), scales=scales_ref.detach(), rotations=gm.get_rotation.detach(),
               sh_degree=3, kernel_size=0.1, require_coord=True, require_depth=True)
    o.forward()
    color, radii, coord, mcoord, depth, mdepth, alpha, normal = o.outputs()
    for key, want in (("render", color), ("mask", alpha), ("expected_coord", coord), ("median_coord", mcoord), ("expected_depth", depth),
                      ("median_depth", mdepth), ("normal", normal), ("radii", radii)):
        assert np.array_equal(out[key].detach().numpy(), want), key
    assert np.array_equal(out["visibility_filter"].numpy(), radii > 0)
    assert (radii > 0).sum() > 100 and float(out["mask"].max()) > 0.5     # the view is not empty

    # ---- backward through the unmodified caller: gradients reach the RAW parameters through the reference's activations ----
    gen = torch.Generator().manual_seed(3)
    w = {k: torch.randn(out[k].shape, generator=gen) for k in ("render", "mask", "expected_coord", "median_coord", "expected_depth", "median_depth", "normal")}
    loss = sum((out[k] * w[k]).sum() for k in w)
    loss.backward()
    assert native.calls == ["forward", "backward"]
    g = {k: torch.from_numpy(np.array(v)) for k, v in native.bwd_grads.items()}
    # what the oracle was given as cotangents is what the caller's loss implies, slot by slot
    o.backward(w["render"], w["expected_coord"], w["median_coord"], w["expected_depth"], w["median_depth"], w["mask"], w["normal"])
    for k, v in o.grads().items():
        assert np.array_equal(np.array(v), native.bwd_grads[k]), k
    assert torch.equal(out["viewspace_points"].grad, g["dL_dmeans2D"])          # densification statistics read this tensor
    assert torch.allclose(gm._xyz.grad, g["dL_dmeans3D"])
    assert torch.allclose(gm._features_dc.grad, g["dL_dsh"][:, :1]) and torch.allclose(gm._features_rest.grad, g["dL_dsh"][:, 1:])
    # scale / opacity / rotation gradients pushed through the reference's own activations
    sc2, op2 = gm.get_scaling_n_opacity_with_3D_filter
    want_s, want_o = torch.autograd.grad([sc2, op2], [gm._scaling, gm._opacity], [g["dL_dscales"], g["dL_dopacity"]])
    assert torch.allclose(gm._scaling.grad, want_s, rtol=1e-5, atol=1e-9) and torch.allclose(gm._opacity.grad, want_o, rtol=1e-5, atol=1e-9)
    want_r, = torch.autograd.grad([gm.get_rotation], [gm._rotation], [g["dL_drotations"]])
    assert torch.allclose(gm._rotation.grad, want_r, rtol=1e-5, atol=1e-9)
    assert float(gm._scaling.grad.abs().max()) > 0 and float(gm._rotation.grad.abs().max()) > 0


def test_reference_integrate_runs_on_this_operator(wired):
    ref, GaussianModel, native = wired
    gm, cam, s = _model_and_camera(GaussianModel, P=800, W=64, H=48, seed=11)
    pts = gm.get_xyz.detach()[:300] + 0.05 * torch.randn(300, 3, generator=torch.Generator().manual_seed(1))
    out = ref.integrate(pts, cam, gm, Pipe(False, False, False), torch.zeros(3), kernel_size=0.0)
    assert native.calls == ["integrate"]
    assert set(out) == {"render", "alpha_integrated", "color_integrated", "point_coordinate", "point_sdf", "visibility_filter", "radii"}
    img, alpha_i, color_i, coord_i, sdf_i, radii = native.integrate_out
    assert out["render"].shape == (9, 48, 64) and torch.equal(out["render"], img)
    assert torch.equal(out["alpha_integrated"], alpha_i) and torch.equal(out["color_integrated"], color_i)
    assert torch.equal(out["point_coordinate"], coord_i) and torch.equal(out["point_sdf"], sdf_i)
    assert torch.equal(out["radii"], radii) and torch.equal(out["visibility_filter"], radii > 0)


def test_reference_argument_errors_are_the_reference_messages(wired):
    """`GaussianRasterizer.forward` refuses the same inconsistent argument sets with the same messages (upstream :208-212)."""
    ref, GaussianModel, native = wired
    import diff_gaussian_rasterization as dgr
    gm, cam, s = _model_and_camera(GaussianModel, P=50, W=32, H=32, seed=2)
    rs = dgr.GaussianRasterizationSettings(32, 32, 0.5, 0.5, 0.0, torch.zeros(3), 1.0, s.viewmatrix, s.projmatrix, 3, s.campos, False, True, True, False)
    r = dgr.GaussianRasterizer(rs)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(gm.get_xyz, torch.zeros(50, 3), gm.get_opacity, shs=gm.get_features, colors_precomp=torch.zeros(50, 3), scales=gm.get_scaling, rotations=gm.get_rotation)
    with pytest.raises(Exception, match="Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"):
        r(gm.get_xyz, torch.zeros(50, 3), gm.get_opacity, shs=gm.get_features)
    assert native.calls == []
