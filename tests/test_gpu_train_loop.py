"""End-to-end: one optimisation loop wired exactly like the reference's training iteration (train.py:125-160 +
gaussian_renderer/__init__.py:56-79), every step through this repository's HIP ops:
    3D-filter activations -> rasterizer (autograd Function) -> normal-consistency loss + L1/SSIM loss -> backward -> Adam
A perturbed copy of a small scene is fitted back to images rendered from the original; the loss has to fall.  This is a
wiring/gradient-sign check across all custom autograd Functions, not a parity test (those are per-op)."""
from collections import namedtuple

import math
import pytest
import torch

from synth_scene import make_scene, to_device

pytestmark = pytest.mark.gpu
View = namedtuple("View", "image_width image_height FoVx FoVy")


def test_training_iterations_reduce_the_loss():
    import fused_adam
    import gaussian_model_ops as gmo
    import graphics_utils as gu
    import loss_utils as lu
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    s = to_device(make_scene(3000, 192, 128, sh_degree=1, mu_px=4.0, seed=90, kernel_size=0.1, require_coord=False, require_depth=True,
                             filter3d=False), dev)
    view = View(s.W, s.H, 2 * math.atan(s.tanfovx), 2 * math.atan(s.tanfovy))
    rs = GaussianRasterizationSettings(image_height=s.H, image_width=s.W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, kernel_size=s.kernel_size,
                                       bg=s.bg, scale_modifier=1.0, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, sh_degree=s.sh_degree,
                                       campos=s.campos, prefiltered=False, require_depth=True, require_coord=False, debug=False)
    rast = GaussianRasterizer(rs)
    filter_3D = torch.full((s.means3D.shape[0], 1), 0.002, device=dev)

    def render(xyz, f, op_raw, sc_raw, rot):
        scales, opacity = gmo.scaling_n_opacity_with_3D_filter(sc_raw, op_raw, filter_3D)
        return rast(means3D=xyz, means2D=torch.zeros_like(xyz, requires_grad=True), shs=f, colors_precomp=None, opacities=opacity,
                    scales=scales, rotations=torch.nn.functional.normalize(rot), cov3D_precomp=None)

    gt = dict(xyz=s.means3D, f=s.shs[:, :4].contiguous(), op=torch.logit(s.opacities.clamp(1e-4, 1 - 1e-4)), sc=torch.log(s.scales), rot=s.rotations)
    with torch.no_grad():
        target = render(gt["xyz"], gt["f"], gt["op"], gt["sc"], gt["rot"])[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    P = s.means3D.shape[0]
    params = dict(xyz=(gt["xyz"] + 0.01 * torch.randn(P, 3, generator=g).to(dev)), f=(gt["f"] + 0.3 * torch.randn(P, 4, 3, generator=g).to(dev)),
                  op=(gt["op"] + 0.5 * torch.randn(P, 1, generator=g).to(dev)), sc=(gt["sc"] + 0.2 * torch.randn(P, 3, generator=g).to(dev)),
                  rot=gt["rot"].clone())
    params = {k: torch.nn.Parameter(v.contiguous()) for k, v in params.items()}
    lrs = dict(xyz=1e-4, f=5e-3, op=2e-2, sc=5e-3, rot=1e-3)
    opt = fused_adam.Adam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in params], lr=0.0, eps=1e-15)
    losses = []
    for it in range(40):
        out = render(params["xyz"], params["f"], params["op"], params["sc"], params["rot"])
        image, depth, mdepth, normal = out[0], out[4], out[5], out[7]
        loss = lu.photometric_loss(image, target, 0.2) + 0.05 * gu.normal_consistency_loss(view, normal, depth, mdepth, 0.6)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        for p in params.values():
            assert p.grad is not None and torch.isfinite(p.grad).all()
        opt.step()
        losses.append(float(loss.item()))
    assert losses[-1] < 0.7 * losses[0], losses[::8]
    assert all(torch.isfinite(p).all() for p in params.values())
