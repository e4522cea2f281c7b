"""TEST INFRASTRUCTURE ONLY: ctypes front-end of oracle/_ref/libradegs_ref.so -- the reference's own rasterizer sources
(DGR/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu) compiled for the host by oracle/build_ref.py.

Same shape as oracle/oracle.py's `Oracle` (forward / backward / outputs / grads / get / integrate) so that a test can run the
hand-written oracle and the compiled reference side by side on one scene.  Only tests/ and tests/golden/make_golden_ref.py import
this module.  `available()` is False where /root/reference does not exist and no prebuilt library travelled (then tests skip).
"""
import ctypes
import os

import numpy as np

from . import build_ref

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_INT_ARRAYS = {"clamped": np.uint8, "radii": np.int32, "tiles_touched": np.uint32, "point_offsets": np.uint32, "keys_sorted": np.uint64,
               "point_list": np.uint32, "ranges": np.uint32, "point_ranges": np.uint32, "n_contrib": np.uint32, "condition": np.uint8,
               "point_tiles_touched": np.uint32}


def lib_path(fma=False):
    return os.path.join(_HERE, "_ref", "libradegs_ref_fma.so" if fma else "libradegs_ref.so")


def available():
    return os.path.exists(lib_path()) or build_ref.available()


def lib(fma=False):
    if fma not in _LIBS:
        if build_ref.available():
            build_ref.build()
        L = ctypes.CDLL(lib_path(fma))
        L.ref_create.restype = ctypes.c_void_p
        L.ref_create.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 11 + [ctypes.c_float] * 4 + [ctypes.c_int] * 3
        L.ref_forward.restype = ctypes.c_int
        L.ref_forward.argtypes = [ctypes.c_void_p]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [ctypes.c_void_p] * 8
        L.ref_integrate.restype = ctypes.c_int
        L.ref_integrate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ref_get.restype = ctypes.c_longlong
        L.ref_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_longlong]
        L.ref_destroy.restype = None
        L.ref_destroy.argtypes = [ctypes.c_void_p]
        L.ref_mark_visible.restype = None
        L.ref_mark_visible.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
        L.ref_set_exp_fn.restype = None
        L.ref_set_exp_fn.argtypes = [ctypes.c_void_p]
        L.ref_set_num_threads.restype = None
        L.ref_set_num_threads.argtypes = [ctypes.c_int]
        L.ref_higher_msb.restype = ctypes.c_uint
        L.ref_higher_msb.argtypes = [ctypes.c_uint]
        L.ref_kat_mat3.restype = None
        L.ref_kat_mat3.argtypes = [ctypes.c_void_p]
        L.ref_sym_eigen3.restype = ctypes.c_int
        L.ref_sym_eigen3.argtypes = [ctypes.c_void_p] * 3
        _LIBS[fma] = L
    return _LIBS[fma]


def _np(x):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(ctypes.c_void_p)


def set_exp(mode, fma=False):
    """'libm': glibc expf (default).  'spec': the oracle's specified exponential (csrc/rg_blend.h::exp_spec), so that the compiled
    reference and the oracle / HIP path take their thresholded decisions from the same exponential."""
    L = lib(fma)
    if mode == "libm":
        L.ref_set_exp_fn(None)
    elif mode == "spec":
        from . import oracle as orc
        fn = ctypes.cast(orc.lib().oracle_exp_spec, ctypes.c_void_p)
        L.ref_set_exp_fn(fn)
    else:
        raise ValueError(mode)


def set_num_threads(n, fma=False):
    """1 (default): blocks run in order on one thread, float atomics are applied in a fixed order (deterministic backward)."""
    lib(fma).ref_set_num_threads(int(n))


class Ref:
    """One scene + one view through the reference's CudaRasterizer::Rasterizer (argument meaning of `_C.rasterize_gaussians`,
    DGR/rasterize_points.h:18-42)."""

    def __init__(self, *, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height, image_width,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0, scale_modifier=1.0,
                 kernel_size=0.0, require_coord=False, require_depth=False, prefiltered=False, fma=False):
        self.L = lib(fma)
        m = _np(means3D)
        self.P = int(m.shape[0])
        self.H, self.W = int(image_height), int(image_width)
        s = _np(shs)
        self.M = 0 if s is None or s.size == 0 else int(s.shape[1])
        keep = [_np(bg), m, s, _np(colors_precomp), _np(opacities), _np(scales), _np(rotations), _np(cov3D_precomp), _np(viewmatrix),
                _np(projmatrix), _np(campos)]
        self._h = self.L.ref_create(self.P, int(sh_degree), self.M, self.W, self.H, *[_ptr(a) for a in keep], float(scale_modifier),
                                    float(tanfovx), float(tanfovy), float(kernel_size), int(bool(require_coord)), int(bool(require_depth)),
                                    int(bool(prefiltered)))
        self.num_rendered = None

    def forward(self):
        self.num_rendered = self.L.ref_forward(self._h)
        return self.num_rendered

    def backward(self, dL_dcolor, dL_dcoord, dL_dmcoord, dL_ddepth, dL_dmdepth, dL_dalpha, dL_dnormal):
        HW = self.H * self.W
        gs = []
        for g, c in ((dL_dcolor, 3), (dL_dcoord, 3), (dL_dmcoord, 3), (dL_ddepth, 1), (dL_dmdepth, 1), (dL_dalpha, 1), (dL_dnormal, 3)):
            a = np.zeros(c * HW, np.float32) if g is None else _np(g).reshape(-1)
            assert a.size == c * HW
            gs.append(a)
        self.L.ref_backward(self._h, *[_ptr(a) for a in gs])

    def get(self, name, shape=None):
        dt = _INT_ARRAYS.get(name, np.float32)
        n = self.L.ref_get(self._h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n // np.dtype(dt).itemsize, dt)
        if n:
            self.L.ref_get(self._h, name.encode(), out.ctypes.data_as(ctypes.c_void_p), n)
        return out.reshape(shape) if shape is not None else out

    def outputs(self):
        H, W = self.H, self.W
        return (self.get("out_color", (3, H, W)), self.get("radii"), self.get("out_coord", (3, H, W)), self.get("out_mcoord", (3, H, W)),
                self.get("out_depth", (1, H, W)), self.get("out_mdepth", (1, H, W)), self.get("out_alpha", (1, H, W)),
                self.get("out_normal", (3, H, W)))

    def grads(self):
        P, M = self.P, self.M
        return dict(dL_dmeans2D=self.get("dL_dmeans2D", (P, 3)), dL_dcolors=self.get("dL_dcolors", (P, 3)),
                    dL_dopacity=self.get("dL_dopacity", (P, 1)), dL_dmeans3D=self.get("dL_dmeans3D", (P, 3)),
                    dL_dcov3D=self.get("dL_dcov3D", (P, 6)), dL_dsh=self.get("dL_dsh", (P, M, 3)),
                    dL_dscales=self.get("dL_dscales", (P, 3)), dL_drotations=self.get("dL_drotations", (P, 4)))

    def integrate(self, points3D, subpixel_offset=None):
        """GaussianRasterizer.integrate (DGR/diff_gaussian_rasterization/__init__.py:239-306)."""
        pts = _np(points3D)
        PN = pts.shape[0]
        sub = np.zeros((self.H, self.W, 2), np.float32) if subpixel_offset is None else _np(subpixel_offset)
        self.num_rendered = self.L.ref_integrate(self._h, PN, _ptr(pts), _ptr(sub))
        H, W = self.H, self.W
        return (self.get("out9", (9, H, W)), self.get("out_alpha_integrated"), self.get("out_color_integrated", (PN, 3)),
                self.get("out_coordinate2d", (PN, 2)), self.get("out_sdf"), self.get("radii"))

    def close(self):
        if self._h:
            self.L.ref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _np(means3D)
    out = np.zeros(m.shape[0], np.uint8)
    lib().ref_mark_visible(m.shape[0], _ptr(m), _ptr(_np(viewmatrix)), _ptr(_np(projmatrix)), _ptr(out))
    return out.astype(bool)


def higher_msb(n):
    return int(lib().ref_higher_msb(n))


def kat_mat3():
    out = np.zeros(3, np.float32)
    lib().ref_kat_mat3(_ptr(out))
    return out


def sym_eigen3(sym6):
    s = _np(sym6)
    ev = np.zeros(3, np.float32)
    V = np.zeros(9, np.float32)
    D = lib().ref_sym_eigen3(_ptr(s), _ptr(ev), _ptr(V))
    return D, ev, V.reshape(3, 3).T  # columns = eigenvectors
