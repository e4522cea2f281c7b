"""The C-ABI library loads, exports every symbol include/radegs.h declares, and the Python operator
mirrors the reference's argument checks.  No compute call is made here (no GPU in this tier)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "radegs.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(radegs_[a-z0-9_]+)\s*\(", hdr)) - {"radegs_alloc_fn"})


def test_header_symbols_are_exported():
    import diff_gaussian_rasterization._C as C
    lib_path = C._LIB_PATH
    if not os.path.exists(lib_path):
        import importlib.util
        spec = importlib.util.spec_from_file_location("radegs_build", os.path.join(ROOT, "rade-gs_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    L = ctypes.CDLL(lib_path)
    declared = _declared_symbols()
    assert set(declared) == set(C.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    L.radegs_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.radegs_version()


def test_operator_surface_matches_reference():
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "kernel_size", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "require_depth", "require_coord", "debug")
    rs = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, 0.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3),
                                           False, True, False, False)
    r = dgr.GaussianRasterizer(rs)
    m, o = torch.zeros(2, 3), torch.ones(2, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), colors_precomp=torch.zeros(2, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3), rotations=torch.ones(2, 4), cov3D_precomp=torch.zeros(2, 6))


def test_no_cpu_fallback():
    import diff_gaussian_rasterization as dgr
    rs = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, 0.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3),
                                           False, True, False, False)
    r = dgr.GaussianRasterizer(rs)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        r(torch.zeros(2, 3), torch.zeros(2, 3), torch.ones(2, 1), shs=torch.zeros(2, 1, 3), scales=torch.ones(2, 3),
          rotations=torch.ones(2, 4))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        r.markVisible(torch.zeros(2, 3))


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rade-gs_amd")):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle|liboracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_fused_step_modules_have_no_cpu_fallback():
    """graphics_utils / gaussian_model_ops refuse CPU tensors instead of falling back."""
    from collections import namedtuple
    import gaussian_model_ops as gmo
    import graphics_utils as gu
    View = namedtuple("View", "image_width image_height FoVx FoVy")
    v = View(8, 8, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        gu.depth_double_to_normal(v, torch.ones(1, 8, 8), torch.ones(1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        gu.normal_consistency_loss(v, torch.ones(3, 8, 8), torch.ones(1, 8, 8), torch.ones(1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        gmo.scaling_n_opacity_with_3D_filter(torch.zeros(4, 3), torch.zeros(4, 1), torch.zeros(4, 1))


def _c_lib():
    import diff_gaussian_rasterization._C as C
    L = ctypes.CDLL(C._LIB_PATH)
    L.radegs_last_error.restype = ctypes.c_char_p
    L.radegs_forward.restype = ctypes.c_int
    L.radegs_backward.restype = ctypes.c_int
    for f in (L.radegs_geometry_bytes, L.radegs_image_bytes, L.radegs_binning_bytes):
        f.restype = ctypes.c_size_t
    return C, L


def test_c_abi_rejects_bad_arguments_before_touching_the_device():
    """The checks of rasterize_points.cu:60-62 / __init__.py:205-210 live behind the C ABI too (a C caller has no Python layer in front):
    every rejection below returns before the first HIP call, so it runs without a GPU.  Pointers are never dereferenced here."""
    C, L = _c_lib()
    INVALID = -1        # RADEGS_ERR_INVALID_ARG (include/radegs.h)
    assert re.search(r"#define\s+RADEGS_ERR_INVALID_ARG\s+\(-1\)", open(os.path.join(ROOT, "include", "radegs.h")).read())
    cb = C._ALLOC_FN(lambda user, n: 0)
    fake = 0x1000

    def fwd(**over):
        kw = dict(P=4, D=0, M=1, width=32, height=32, background=fake, means3D=fake, shs=fake, colors_precomp=None, opacities=fake,
                  scales=fake, rotations=fake, cov3D_precomp=None, viewmatrix=fake, projmatrix=fake, cam_pos=fake, scale_modifier=1.0,
                  tan_fovx=1.0, tan_fovy=1.0, kernel_size=0.0, prefiltered=0, require_coord=0, require_depth=0, debug=0,
                  out_color=fake, out_coord=None, out_mcoord=None, out_depth=None, out_mdepth=None, out_alpha=fake, out_normal=None,
                  radii=fake)
        kw.update(over)
        a = C.RadegsFwdArgs(**kw)
        rc = L.radegs_forward(ctypes.byref(a), cb, None, cb, None, cb, None, None)
        return rc, L.radegs_last_error().decode()

    assert L.radegs_forward(None, cb, None, cb, None, cb, None, None) == INVALID
    assert fwd(P=0)[0] == 0                                   # nothing to do, nothing launched (rasterize_points.cu:90)
    for over, text in ((dict(P=-1), "bad sizes"), (dict(width=0), "bad sizes"), (dict(means3D=None), "missing required tensor"),
                       (dict(radii=None), "missing required tensor"),
                       (dict(shs=None), "excatly one of either SHs or precomputed colors"),
                       (dict(colors_precomp=fake), "excatly one of either SHs or precomputed colors"),
                       (dict(rotations=None), "exactly one of either scale/rotation pair or precomputed 3D covariance"),
                       (dict(cov3D_precomp=fake), "exactly one of either scale/rotation pair or precomputed 3D covariance"),
                       (dict(prefiltered=1), "prefiltered"), (dict(D=2, M=4), "sh_degree needs more SH rows"),
                       (dict(require_coord=1), "coord outputs missing"), (dict(require_depth=1), "depth outputs missing"),
                       (dict(require_depth=1, out_depth=fake, out_mdepth=fake), "normal output missing")):
        rc, msg = fwd(**over)
        assert rc == INVALID and text in msg, (over, rc, msg)
    # the backward: state buffers and gradient outputs are checked before anything runs
    b = C.RadegsBwdArgs()
    b.P = 3
    # a caller compiled against a different header (a structure of another size) is refused before any field beyond it is read
    assert L.radegs_backward(ctypes.byref(b), cb, None, None) == INVALID and "struct_size" in L.radegs_last_error().decode()
    b.struct_size = ctypes.sizeof(C.RadegsBwdArgs) - 8
    assert L.radegs_backward(ctypes.byref(b), cb, None, None) == INVALID and "struct_size" in L.radegs_last_error().decode()
    b.struct_size = ctypes.sizeof(C.RadegsBwdArgs)
    assert L.radegs_backward(ctypes.byref(b), cb, None, None) == INVALID and "state buffers missing" in L.radegs_last_error().decode()
    b.geom_buffer, b.image_buffer = fake, fake
    assert L.radegs_backward(ctypes.byref(b), cb, None, None) == INVALID and "gradient outputs missing" in L.radegs_last_error().decode()
    b.P = 0
    assert L.radegs_backward(ctypes.byref(b), cb, None, None) == 0
    # radegs_backward_from_sums (the per-Gaussian half over caller-supplied sums): everything its kernel dereferences is checked, and the
    # 16-byte alignment its record loads need
    L.radegs_backward_from_sums.restype = ctypes.c_int
    L.radegs_backward_from_sums.argtypes = [ctypes.POINTER(C.RadegsBwdArgs), ctypes.c_void_p, ctypes.c_void_p]
    s_ = C.RadegsBwdArgs()
    assert L.radegs_backward_from_sums(ctypes.byref(s_), fake, None) == INVALID and "struct_size" in L.radegs_last_error().decode()
    s_.struct_size = ctypes.sizeof(C.RadegsBwdArgs)
    s_.P, s_.D, s_.M, s_.width, s_.height = 3, 0, 1, 32, 32
    s_.geom_buffer = fake
    for f in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dscale", "dL_drot"):
        setattr(s_, f, fake)
    def from_sums(sums=fake):
        return L.radegs_backward_from_sums(ctypes.byref(s_), sums, None), L.radegs_last_error().decode()
    assert from_sums() == (INVALID, "missing required tensor")                       # radii, means3D, camera
    s_.radii, s_.means3D, s_.viewmatrix, s_.projmatrix, s_.cam_pos = fake, fake, fake, fake, fake
    rc, msg = from_sums()
    assert rc == INVALID and "exactly one of either scale/rotation pair or precomputed 3D covariance" in msg
    s_.scales, s_.rotations = fake, fake
    rc, msg = from_sums(fake + 4)
    assert rc == INVALID and "16-byte aligned" in msg
    assert L.radegs_backward_from_sums(ctypes.byref(s_), None, None) == INVALID
    # the switches are read once; a host may have them read again, and ask which formulation its last forward used (none yet: -1)
    L.radegs_reload_env.restype = None
    L.radegs_reload_env()
    L.radegs_last_forward_used_streams.restype = ctypes.c_int
    assert L.radegs_last_forward_used_streams() == -1
    assert re.search(r"#define\s+RADEGS_ERR_STATE\s+\(-5\)", open(os.path.join(ROOT, "include", "radegs.h")).read())


def test_state_sizes_are_host_arithmetic():
    """radegs_*_bytes (the `required<T>` of rasterizer_impl.h:84-91): pure functions of the sizes, monotone, and the image state does not
    depend on the number of Gaussians."""
    _, L = _c_lib()
    g = [L.radegs_geometry_bytes(p, 0) for p in (1, 1000, 1_000_000)]
    assert g[0] > 0 and g[0] < g[1] < g[2] and L.radegs_geometry_bytes(1000, 1) > g[1]     # the coord map adds a 48-B record per Gaussian
    assert 1_000_000 * 64 < g[2] < 1_000_000 * 400                                          # 64-B blend record + keys, indices, sort scratch
    im = L.radegs_image_bytes(1920, 1080)
    assert im >= 1920 * 1080 * 8 and L.radegs_image_bytes(960, 540) < im
    b = [L.radegs_binning_bytes(r) for r in (0, 1000, 4_000_000)]
    assert b[0] <= b[1] < b[2] and b[2] >= 4_000_000 * 16                                   # two key/value ping-pong pairs


def test_unproduced_zero_maps_can_be_edited_in_place_under_autograd():
    """the maps a call does not produce come from one zero-filled allocation; returned from an autograd Function as
    slices they would be several views of one base, and `depth[mask] = 0` / `normal *= x` on them would raise.  The reference's
    separate torch.full tensors allow it (rasterize_points.cu:71-77)."""
    import diff_gaussian_rasterization._C as C

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return (x * 2,) + tuple(C._zero_maps([3, 1, 3], 4, 5, x.device))

        @staticmethod
        def backward(ctx, *g):
            return g[0] * 2

    x = torch.ones(3, requires_grad=True)
    y, a, b, c = F.apply(x)
    assert a._base is None and b._base is None and c._base is None
    a[0, 0, 0] = 5.0
    b *= 3.0
    c.add_(1.0)
    assert float(a.sum()) == 5.0 and float(b.sum()) == 0.0 and float(c.sum()) == 60.0     # disjoint: no map sees another's edit
    y.sum().backward()
    assert torch.equal(x.grad, torch.full((3,), 2.0))
    again = C._zero_maps([3, 1, 3], 4, 5, x.device)
    assert all(float(t.abs().sum()) == 0.0 for t in again)                                 # fresh memory every call


def test_failed_allocation_request_does_not_outlive_a_successful_retry():
    """alloc_image (rg_launch.inc) retries without entry streams when the stream-sized request fails; the binding must not
    raise the first request's stale exception after the retry succeeded."""
    import diff_gaussian_rasterization._C as C
    r = C._Resizable(torch.device("cpu"))
    assert r.cb(None, 1 << 62) in (0, None)        # cannot be served: recorded, NULL returned to the native side
    assert r.error is not None
    p = r.cb(None, 4096)
    assert p and r.error is None and r.tensor.numel() == 4096
    r.release()
