"""The reference's compiled `_C` module, rebuilt over the C ABI (rade-gs_amd/csrc/torch_binding/radegs_torch_binding.cpp ->
diff_gaussian_rasterization/_C_torch*.so): upstream's four entry points (DGR/ext.cpp:15-19) with torch::Tensor arguments in upstream's
positional order.  CPU tier: it builds, imports and has upstream's surface.  GPU tier: bit-identical forward / identical-within-tolerance
backward against the ctypes binding (`_C.py`), and the operator package driven through it (RADEGS_BINDING=torch) end to end."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from synth_scene import make_scene, upstream_grads
from util import close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _binding():
    spec = importlib.util.spec_from_file_location("radegs_build", os.path.join(ROOT, "rade-gs_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    path = b.torch_binding_path()
    if not os.path.exists(path):   # __graft_entry__.build() makes it; a bare checkout builds on first use (g++, ~90 s)
        b.build_torch_binding(verbose=False)
    import diff_gaussian_rasterization._C_torch as T
    return T


def _arity(fn):
    sig = fn.__doc__.splitlines()[0]
    return sig[sig.index("(") + 1:sig.rindex(") ->")].count(": ")


def test_module_has_upstreams_four_entry_points_with_upstreams_arity():
    T = _binding()
    # DGR/rasterize_points.h: 22 / 32 / 3 / 23 positional arguments
    assert _arity(T.rasterize_gaussians) == 22
    assert _arity(T.rasterize_gaussians_backward) == 32
    assert _arity(T.mark_visible) == 3
    assert _arity(T.integrate_gaussians_to_points) == 23
    assert T.radegs_version().startswith("radegs-hip")
    # its first argument check is upstream's (rasterize_points.cu:60-62), raised before anything touches a device
    e = torch.Tensor([])
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        T.rasterize_gaussians(e, torch.zeros(4, 2), e, e, e, e, 1.0, e, e, e, 1.0, 1.0, 0.0, 8, 8, e, 0, e, False, False, True, False)
    # ... and, like the ctypes binding, it has no CPU path
    with pytest.raises(RuntimeError, match="must be a GPU tensor"):
        T.rasterize_gaussians(e, torch.zeros(4, 3), e, e, e, e, 1.0, e, e, e, 1.0, 1.0, 0.0, 8, 8, e, 0, e, False, False, True, False)


def _native_args(h):
    e = torch.Tensor([])
    rs = h.rs
    return (rs.bg, h.means3D.detach(), e if h.colors is None else h.colors.detach(), h.opacities.detach(),
            e if h.scales is None else h.scales.detach(), e if h.rotations is None else h.rotations.detach(), rs.scale_modifier,
            e if h.cov3D is None else h.cov3D.detach(), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size,
            rs.image_height, rs.image_width, e if h.shs is None else h.shs.detach(), rs.sh_degree, rs.campos, False, rs.require_coord,
            rs.require_depth, rs.debug)


def _same_up_to_summation_order(x, y):
    """Two runs of the SAME kernels on the same inputs differ by the order of the blend backward's atomic additions: ~1e-6 of a sum's
    terms.  The per-Gaussian chain rule multiplies that by up to 1 / lambda_min for needle-thin splats (tests/test_gpu_parity.py::
    check_backward), so a handful of rows may differ visibly: all but 0.5 % of the elements within 1e-3 relative + 1e-5 of the tensor's scale."""
    scale = max(float(np.abs(x).max()), 1e-30)
    return float(np.mean(np.isclose(x, y, rtol=1e-3, atol=1e-5 * scale))) >= 0.995


def _cov3d(s):
    from test_hostcheck import cov3d_of
    return cov3d_of(s)


@pytest.mark.gpu
@pytest.mark.parametrize("coord,depth,precomp", [(False, True, False), (True, False, False), (True, True, True), (False, False, False)])
def test_compiled_module_equals_the_ctypes_binding(coord, depth, precomp):
    from gpu_util import HipRun
    import diff_gaussian_rasterization._C as C
    T = _binding()
    dev = "cuda:0"
    # kernel_size = 0 (the reference's default): with a non-zero filter the reference's executed backward carries a term that turns the
    # summation-order noise of the conic sums into 1e-4 .. 1e-3 of the geometry gradients' scale (include/radegs.h: opacity_grad_intended),
    # and two launches could then only be compared through the noise-relative criteria of tests/arbiter.py
    s = make_scene(5000, 251, 173, sh_degree=2, mu_px=2.5, seed=311 + 2 * coord + depth, kernel_size=0.0, require_coord=coord,
                   require_depth=depth, pose="random")
    kw = dict(colors=torch.rand(5000, 3), cov3D=_cov3d(s)) if precomp else {}
    h = HipRun(s, dev, **kw)
    args = _native_args(h)
    a = C.rasterize_gaussians(*args)
    b = T.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    assert a[0] == b[0] and a[0] > 0
    names = ("color", "coord", "mcoord", "alpha", "normal", "depth", "mdepth", "radii")
    for k, n in enumerate(names, start=1):
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, n
        assert torch.equal(a[k], b[k]), n                           # same kernels on the same inputs: bit for bit
    if not coord:
        assert not bool(b[2].any()) and not bool(b[3].any())        # unproduced maps come back all-zero (rasterize_points.cu:71-77)
    if not depth:
        assert not bool(b[6].any()) and not bool(b[7].any())
    for k in (9, 10, 11):
        assert b[k].dtype == torch.uint8 and b[k].is_cuda and b[k].numel() > 0

    g = upstream_grads(s, 5)
    e = torch.Tensor([])
    rs = h.rs

    def bwd(mod, st):
        return mod.rasterize_gaussians_backward(
            rs.bg, h.means3D.detach(), st[8], args[2], args[4], args[5], rs.scale_modifier, args[7], rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, rs.kernel_size, g["color"].to(dev), g["coord"].to(dev), g["mcoord"].to(dev), g["depth"].to(dev), g["mdepth"].to(dev),
            g["alpha"].to(dev), g["normal"].to(dev), st[5], args[15], rs.sh_degree, rs.campos, st[9], st[0], st[10], st[11], st[4],
            rs.require_coord, rs.require_depth, False)
    ga, gb = bwd(C, a), bwd(T, b)
    gb2 = bwd(T, b)           # a second call: the cached, re-zeroed accumulation scratch
    gx = bwd(T, a)            # state written by the OTHER binding's forward: the buffers are interchangeable
    torch.cuda.synchronize()
    gnames = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    for n, x, y, y2, z in zip(gnames, ga, gb, gb2, gx):
        if x is None or x.numel() == 0:   # no SH input: upstream (and the module) return a (P, 0, 3) tensor
            assert y.shape == (5000, 0, 3)
            continue
        assert x.shape == y.shape, n
        for other in (y, y2, z):
            # the same kernels; only the order of the blend backward's atomic additions differs between two launches
            xs, os_ = x.cpu().numpy(), other.cpu().numpy()
            scale = max(float(np.abs(xs).max()), 1e-30)
            assert np.isfinite(os_).all(), n
            assert _same_up_to_summation_order(xs, os_), (n, float(np.abs(xs - os_).max()), scale)
    # nothing rendered: zeros and empty state, like upstream (rasterize_points.cu:90)
    z = T.rasterize_gaussians(rs.bg, torch.zeros(0, 3, device=dev), e, torch.zeros(0, 1, device=dev), torch.zeros(0, 3, device=dev),
                              torch.zeros(0, 4, device=dev), 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, 0.0, 33, 47,
                              torch.zeros(0, 9, 3, device=dev), 2, rs.campos, False, coord, depth, False)
    assert z[0] == 0 and z[1].shape == (3, 33, 47) and not bool(z[1].any()) and z[9].numel() == 0


@pytest.mark.gpu
def test_compiled_module_runs_on_the_current_stream():
    """The module queues on torch's CURRENT stream of the inputs' device (c10::hip::getCurrentHIPStreamMasqueradingAsCUDA), as upstream's
    kernels do: a forward + backward issued inside `torch.cuda.stream(side)` is ordered with the side stream's other work and equals the
    default-stream result."""
    from gpu_util import HipRun
    T = _binding()
    dev = "cuda:0"
    s = make_scene(4000, 224, 160, sh_degree=1, mu_px=2.5, seed=412, kernel_size=0.0, require_coord=False, require_depth=True, pose="random")
    h = HipRun(s, dev)
    args = _native_args(h)
    ref = T.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        scaled = h.opacities.detach() * 1.0          # produced ON the side stream: the forward must be queued behind it
        a2 = list(args); a2[3] = scaled
        out = T.rasterize_gaussians(*a2)
        doubled = out[1] * 2.0
    side.synchronize()
    assert out[0] == ref[0]
    for k in range(1, 9):
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(doubled, ref[1] * 2.0)


@pytest.mark.gpu
def test_mark_visible_and_integrate_through_the_compiled_module():
    from gpu_util import HipRun
    import diff_gaussian_rasterization._C as C
    T = _binding()
    dev = "cuda:0"
    s = make_scene(4000, 200, 144, sh_degree=1, mu_px=3.0, seed=77, kernel_size=0.1, require_coord=False, require_depth=True, pose="random")
    h = HipRun(s, dev)
    rs = h.rs
    m3 = h.means3D.detach()
    assert torch.equal(C.mark_visible(m3, rs.viewmatrix, rs.projmatrix), T.mark_visible(m3, rs.viewmatrix, rs.projmatrix))
    e = torch.Tensor([])
    gen = torch.Generator().manual_seed(3)
    pts = (m3.cpu()[torch.randint(0, 4000, (3000,), generator=gen)] + 0.05 * torch.randn(3000, 3, generator=gen)).to(dev)
    args = (rs.bg, pts, m3, e, h.opacities.detach(), h.scales.detach(), h.rotations.detach(), 1.0, e, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, rs.kernel_size, e, rs.image_height, rs.image_width, h.shs.detach(), rs.sh_degree, rs.campos, False, False)
    a = C.integrate_gaussians_to_points(*args)
    b = T.integrate_gaussians_to_points(*args)
    torch.cuda.synchronize()
    assert a[0] == b[0]
    for k in range(1, 7):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_operator_package_over_the_compiled_module_end_to_end():
    """RADEGS_BINDING=torch: GaussianRasterizer -> autograd -> the compiled `_C_torch` (what upstream's own __init__.py does with its
    `_C`), against the oracle like every other parity test -- in a fresh process, the binding being chosen at import."""
    _binding()
    code = (
        "import sys, os\n"
        "for p in %r: sys.path.insert(0, p)\n"
        "import numpy as np, torch\n"
        "import diff_gaussian_rasterization as dgr\n"
        "assert dgr._C.__name__.endswith('_C_torch'), dgr._C.__name__\n"
        "from gpu_util import HipRun\n"
        "from synth_scene import make_scene, upstream_grads\n"
        "from util import close, oracle_backward, oracle_for\n"
        "s = make_scene(4000, 256, 192, sh_degree=3, mu_px=3.0, seed=123, kernel_size=0.0, require_coord=True, require_depth=True, pose='random')\n"
        "g = upstream_grads(s, 123)\n"
        "o = oracle_for(s); o.forward(); ref_out, ref_grad = o.outputs(), oracle_backward(o, g)\n"
        "h = HipRun(s, 'cuda:0')\n"
        "out = [t.detach().cpu().numpy() for t in h.forward()]\n"
        "grads = h.backward(g)\n"
        "assert np.array_equal(out[1], ref_out[1])\n"
        "for k in (0, 2, 3, 4, 5, 6, 7): assert close(out[k], ref_out[k]).all(), k\n"
        "# gradients: against the ctypes binding's on the same inputs (itself held to the oracle by every other GPU test); the two runs differ\n"
        "# by the order of the blend backward's atomic additions only\n"
        "import diff_gaussian_rasterization._C as C\n"
        "st = h.forward_native()\n"
        "e = torch.Tensor([]); rs = h.rs; dev = 'cuda:0'\n"
        "ref = C.rasterize_gaussians_backward(rs.bg, h.means3D.detach(), st[8], e, h.scales.detach(), h.rotations.detach(), rs.scale_modifier, e,\n"
        "    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, g['color'].to(dev), g['coord'].to(dev), g['mcoord'].to(dev),\n"
        "    g['depth'].to(dev), g['mdepth'].to(dev), g['alpha'].to(dev), g['normal'].to(dev), st[5], h.shs.detach(), rs.sh_degree, rs.campos,\n"
        "    st[9], st[0], st[10], st[11], st[4], rs.require_coord, rs.require_depth, False)\n"
        "names = ('dL_dmeans2D', 'dL_dcolors', 'dL_dopacity', 'dL_dmeans3D', 'dL_dcov3D', 'dL_dsh', 'dL_dscales', 'dL_drotations')\n"
        "for n, r in zip(names, ref):\n"
        "    a = grads.get(n)\n"
        "    if a is None or r is None: continue\n"
        "    r = r.cpu().numpy().reshape(a.shape); scale = max(float(np.abs(r).max()), 1e-30)\n"
        "    assert float(np.mean(np.isclose(a, r, rtol=1e-3, atol=1e-5 * scale))) >= 0.995, (n, float(np.abs(a - r).max()), scale)\n"
        "print('END-TO-END OK')\n"
    ) % ([ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests")],)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RADEGS_BINDING="torch"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "END-TO-END OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
