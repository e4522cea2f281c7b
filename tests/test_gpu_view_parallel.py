"""The factored gradient exchange of the view-parallel path (view_parallel.FactoredGradExchange): the SH gradient rebuilt from
per-view dL/dRGB rows must equal the sum of the per-view SH gradients the plain backward writes.  Emulated on ONE GPU with
three neighbouring views of the same Gaussians (the collectives themselves are covered by tests/test_dist_gloo.py and by
`bench.py --force-allreduce` under torch.distributed.run)."""
import numpy as np
import pytest
import torch

from synth_scene import jittered_view, make_scene, to_device, upstream_grads

pytestmark = pytest.mark.gpu


def _backward(C, s, g, e):
    fw = C.rasterize_gaussians(s.bg, s.means3D, e, s.opacities, s.scales, s.rotations, 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                               s.tanfovy, s.kernel_size, s.H, s.W, s.shs, s.sh_degree, s.campos, False, s.require_coord, s.require_depth, False)
    R, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geom, binning, img = fw
    return C.rasterize_gaussians_backward(s.bg, s.means3D, radii, e, s.scales, s.rotations, 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                          s.tanfovy, s.kernel_size, g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"],
                                          g["normal"], normal, s.shs, s.sh_degree, s.campos, geom, R, binning, img, alpha, s.require_coord,
                                          s.require_depth, False)


@pytest.mark.parametrize("deg", [0, 2, 3])
def test_factored_sh_gradient_equals_sum_over_views(deg):
    import diff_gaussian_rasterization._C as C
    from view_parallel import FactoredGradExchange
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    base = make_scene(20000, 256, 192, sh_degree=deg, mu_px=2.5, seed=50 + deg, kernel_size=0.1, require_coord=False, require_depth=True)
    views = [base, jittered_view(base, 1), jittered_view(base, 2)]
    g = {k: v.to(dev) for k, v in upstream_grads(base, 3).items()}
    e = torch.Tensor([])
    P, M = base.means3D.shape[0], base.shs.shape[1]
    plain, small_sum, drgb, campos = None, None, [], []
    try:
        for v in views:
            s = to_device(v, dev)
            C.set_grad_allocator(dev, None)
            bw = _backward(C, s, g, e)
            plain = bw[5].clone() if plain is None else plain + bw[5]
            ex = FactoredGradExchange(P, M, deg, dev)
            C.set_grad_allocator(dev, ex.allocator)
            bw2 = _backward(C, s, g, e)
            assert bw2[5] is None                                  # the (P,M,3) tensor is neither allocated nor written
            for a, b in ((bw[3], bw2[3]), (bw[2], bw2[2]), (bw[6], bw2[6]), (bw[7], bw2[7]), (bw[1], bw2[1])):
                assert a.shape == b.shape                            # the other gradients are unaffected (atomics: fp32 noise)
                assert torch.allclose(a, b, rtol=1e-3, atol=1e-6 * float(a.abs().max()))
            assert bw2[3].data_ptr() == ex.views["dL_dmeans3D"].data_ptr()   # written in place into the exchange buffer
            # single-view exchange (no process group): rebuild == this view's plain dL_dsh
            out = ex.exchange(s.means3D, s.campos, average=True)
            ref = bw[5]
            assert (out["dL_dsh"] - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-12
            drgb.append(ex.drgb.clone())
            campos.append(s.campos.clone())
    finally:
        C.set_grad_allocator(dev, None)
    total = C.sh_grad_from_views(to_device(base, dev).means3D, torch.stack(campos), torch.stack(drgb), deg, M, 1.0 / 3.0)
    ref = plain / 3.0
    assert (total - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-12
    assert (total[:, (deg + 1) ** 2:] == 0).all()
