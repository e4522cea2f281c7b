"""TEST INFRASTRUCTURE ONLY: ctypes front-end of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (rade-gs_amd/) never does.  The oracle restates the
reference's CUDA rasterizer (DGR/cuda_rasterizer/*.cu) on the CPU -- see the header of
radegs_oracle.cpp for the parity status ("parity unpinned": the reference ships no golden
vectors for this path).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("radegs_oracle.cpp", "oracle_linalg.h", "oracle_eigen.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.oracle_create.restype = ctypes.c_void_p
        L.oracle_create.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p] * 11 + [ctypes.c_double] * 4 + [ctypes.c_int] * 3
        L.oracle_forward.restype = ctypes.c_int
        L.oracle_forward.argtypes = [ctypes.c_void_p]
        L.oracle_backward.restype = None
        L.oracle_backward.argtypes = [ctypes.c_void_p] * 8
        L.oracle_get.restype = ctypes.c_longlong
        L.oracle_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_longlong]
        L.oracle_integrate.restype = ctypes.c_int
        L.oracle_integrate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.oracle_stat_pairs.restype = ctypes.c_longlong
        L.oracle_stat_pairs.argtypes = [ctypes.c_void_p]
        L.oracle_stat_blended.restype = ctypes.c_longlong
        L.oracle_stat_blended.argtypes = [ctypes.c_void_p]
        L.oracle_destroy.restype = None
        L.oracle_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_mark_visible.restype = None
        L.oracle_mark_visible.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
        L.oracle_exp_spec.restype = ctypes.c_float
        L.oracle_exp_spec.argtypes = [ctypes.c_float]
        L.oracle_set_exp_mode.restype = None
        L.oracle_set_exp_mode.argtypes = [ctypes.c_int]
        L.oracle_set_opacity_slip.restype = None
        L.oracle_set_opacity_slip.argtypes = [ctypes.c_int]
        L.oracle_set_ref_order.restype = None
        L.oracle_set_ref_order.argtypes = [ctypes.c_int]
        L.oracle_higher_msb.restype = ctypes.c_uint
        L.oracle_higher_msb.argtypes = [ctypes.c_uint]
        L.oracle_kat_mat3.restype = None
        L.oracle_kat_mat3.argtypes = [ctypes.c_void_p]
        L.oracle_sym_eigen3.restype = ctypes.c_int
        L.oracle_sym_eigen3.argtypes = [ctypes.c_void_p] * 3
        _LIB = L
    return _LIB


def _np(x, dtype):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)


_INT_ARRAYS = {"condition": np.uint8, "point_ranges": np.uint32, "pt_list": np.uint32, "clamped": np.uint8, "radii": np.int32, "tiles_touched": np.uint32, "point_offsets": np.uint32,
               "keys_sorted": np.uint64, "point_list": np.uint32, "ranges": np.uint32, "n_contrib": np.uint32}


class Oracle:
    """One scene + one view.  Mirrors the argument meaning of `_C.rasterize_gaussians`
    (DGR/rasterize_points.h:18-42).  precision=32 follows the reference's fp32 arithmetic;
    precision=64 runs the same formulas in double (derivative validation only)."""

    def __init__(self, *, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
                 scale_modifier=1.0, kernel_size=0.0, require_coord=False, require_depth=False, precision=32, nthreads=None):
        self.dt = np.float64 if precision == 64 else np.float32
        dt = self.dt
        self.P = int(np.asarray(_np(means3D, dt)).shape[0])
        self.H, self.W = int(image_height), int(image_width)
        self.M = 0 if shs is None else int(_np(shs, dt).shape[1])
        self.req_coord, self.req_depth = bool(require_coord), bool(require_depth)
        keep = [_np(bg, dt), _np(means3D, dt), _np(shs, dt), _np(colors_precomp, dt), _np(opacities, dt), _np(scales, dt),
                _np(rotations, dt), _np(cov3D_precomp, dt), _np(viewmatrix, dt), _np(projmatrix, dt), _np(campos, dt)]
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        self.nthreads = nthreads
        self._h = lib().oracle_create(precision, self.P, int(sh_degree), self.M, self.W, self.H, *[_ptr(a) for a in keep],
                                      float(scale_modifier), float(tanfovx), float(tanfovy), float(kernel_size),
                                      int(self.req_coord), int(self.req_depth), int(nthreads))
        self.num_rendered = None

    def forward(self):
        self.num_rendered = lib().oracle_forward(self._h)
        return self.num_rendered

    def backward(self, dL_dcolor, dL_dcoord, dL_dmcoord, dL_ddepth, dL_dmdepth, dL_dalpha, dL_dnormal):
        HW = self.H * self.W
        gs = []
        for g, c in ((dL_dcolor, 3), (dL_dcoord, 3), (dL_dmcoord, 3), (dL_ddepth, 1), (dL_dmdepth, 1), (dL_dalpha, 1), (dL_dnormal, 3)):
            a = np.zeros(c * HW, self.dt) if g is None else _np(g, self.dt).reshape(-1)
            assert a.size == c * HW
            gs.append(a)
        lib().oracle_backward(self._h, *[_ptr(a) for a in gs])

    def get(self, name, shape=None):
        dt = _INT_ARRAYS.get(name, self.dt)
        n = lib().oracle_get(self._h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n // np.dtype(dt).itemsize, dt)
        if n:
            lib().oracle_get(self._h, name.encode(), out.ctypes.data_as(ctypes.c_void_p), n)
        return out.reshape(shape) if shape is not None else out

    def integrate(self, points3D):
        """GaussianRasterizer.integrate (DGR/diff_gaussian_rasterization/__init__.py:239-306): returns
        (color[9,H,W], alpha_integrated[PN], color_integrated[PN,3], point_coordinate[PN,2], point_sdf[PN], radii)."""
        pts = _np(points3D, self.dt)
        PN = pts.shape[0]
        self.num_rendered = lib().oracle_integrate(self._h, PN, _ptr(pts))
        H, W = self.H, self.W
        return (self.get("out9", (9, H, W)), self.get("out_alpha_integrated"), self.get("out_color_integrated", (PN, 3)),
                self.get("out_coordinate2d", (PN, 2)), self.get("out_sdf"), self.get("radii"))

    def outputs(self):
        """The 8-tuple of DGR/diff_gaussian_rasterization/__init__.py:101 as numpy arrays."""
        H, W = self.H, self.W
        return (self.get("out_color", (3, H, W)), self.get("radii"), self.get("out_coord", (3, H, W)),
                self.get("out_mcoord", (3, H, W)), self.get("out_depth", (1, H, W)), self.get("out_mdepth", (1, H, W)),
                self.get("out_alpha", (1, H, W)), self.get("out_normal", (3, H, W)))

    def grads(self):
        """Same order as `_C.rasterize_gaussians_backward` (DGR/rasterize_points.h:43-76)."""
        P, M = self.P, self.M
        return dict(dL_dmeans2D=self.get("dL_dmeans2D", (P, 3)), dL_dcolors=self.get("dL_dcolors", (P, 3)),
                    dL_dopacity=self.get("dL_dopacity", (P, 1)), dL_dmeans3D=self.get("dL_dmeans3D", (P, 3)),
                    dL_dcov3D=self.get("dL_dcov3D", (P, 6)), dL_dsh=self.get("dL_dsh", (P, M, 3)),
                    dL_dscales=self.get("dL_dscales", (P, 3)), dL_drotations=self.get("dL_drotations", (P, 4)))

    def stat_pairs(self):
        return lib().oracle_stat_pairs(self._h)

    def stat_blended(self):
        return lib().oracle_stat_blended(self._h)

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _np(means3D, np.float32)
    out = np.zeros(m.shape[0], np.uint8)
    lib().oracle_mark_visible(m.shape[0], _ptr(m), _ptr(_np(viewmatrix, np.float32)), _ptr(_np(projmatrix, np.float32)), _ptr(out))
    return out.astype(bool)


def exp_spec(x):
    return float(lib().oracle_exp_spec(ctypes.c_float(x)))


def set_exp_mode(mode):
    """0 = specified exp (default); 1 = libm expf; 2 / 3 = the specification one ulp up / down.  Sensitivity experiments only:
    always restore 0."""
    lib().oracle_set_exp_mode(int(mode))


def set_opacity_slip(on):
    """1 (default): the backward the reference EXECUTES -- `combined_opacity` read from dL_dconic.w because of the argument slip at
    rasterizer_impl.cu:568 (see radegs_oracle.cpp::g_opacity_slip).  0: the derivative the formulas intend (calculus checks only;
    always restore 1)."""
    lib().oracle_set_opacity_slip(int(on))


def set_ref_order(on):
    """1: the blend backward's per-Gaussian sums are formed in fp32 in the order of the compiled reference's host schedule
    (radegs_oracle.cpp::g_ref_order), for bit-for-bit comparison with oracle/_ref.  Always restore 0."""
    lib().oracle_set_ref_order(int(on))


def higher_msb(n):
    return int(lib().oracle_higher_msb(n))


def kat_mat3():
    out = np.zeros(3, np.float32)
    lib().oracle_kat_mat3(_ptr(out))
    return out


def sym_eigen3(sym6):
    s = _np(sym6, np.float32)
    ev = np.zeros(3, np.float32)
    V = np.zeros(9, np.float32)
    D = lib().oracle_sym_eigen3(_ptr(s), _ptr(ev), _ptr(V))
    return D, ev, V.reshape(3, 3).T  # columns = eigenvectors
