"""TEST HARNESS: loads tests/hostcheck/libhostcheck.so (the product's host+device headers
compiled for the CPU) -- see hostcheck.cpp."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "..", "rade-gs_amd", "csrc")
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libhostcheck.so")
    deps = [os.path.join(_HERE, "hostcheck.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                               "-Wno-unknown-pragmas", "-I", _CSRC, "-shared", "-o", so, os.path.join(_HERE, "hostcheck.cpp")])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.hc_exp_spec.restype = ctypes.c_float
        _LIB.hc_exp_spec.argtypes = [ctypes.c_float]
        _LIB.hc_exp_spec_sweep.restype = ctypes.c_longlong
        _LIB.hc_exp_spec_sweep.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
        _LIB.hc_splat_power.restype = ctypes.c_float
        _LIB.hc_splat_power.argtypes = [ctypes.c_float] * 5
        _LIB.hc_skip_threshold.restype = ctypes.c_float
        _LIB.hc_skip_threshold.argtypes = [ctypes.c_float]
    return _LIB


def _f(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def preprocess_fwd(scene, colors=None, cov3D=None, use_sh=True):
    s = scene
    P = s.means3D.shape[0]
    means, scales, rots, opac = _f(s.means3D), _f(s.scales), _f(s.rotations), _f(s.opacities)
    shs = _f(s.shs) if (use_sh and colors is None) else None
    M = 0 if shs is None else shs.shape[1]
    colors = _f(colors)
    cov3D = _f(cov3D)
    if cov3D is not None:
        scales = rots = None
    out_f = np.zeros((P, 27), np.float32)
    out_i = np.zeros((P, 3), np.int32)
    L = lib()
    L.hc_preprocess_fwd(ctypes.c_int(P), ctypes.c_int(s.sh_degree), ctypes.c_int(M), _p(means), _p(scales), _p(rots), _p(cov3D),
                        _p(opac), _p(shs), _p(colors), _p(_f(s.viewmatrix)), _p(_f(s.projmatrix)), _p(_f(s.campos)),
                        ctypes.c_int(s.W), ctypes.c_int(s.H), ctypes.c_float(s.tanfovx), ctypes.c_float(s.tanfovy),
                        ctypes.c_float(s.kernel_size), ctypes.c_float(1.0), _p(out_f), _p(out_i))
    return out_f, out_i


def preprocess_inte(scene):
    s = scene
    P = s.means3D.shape[0]
    shs = _f(s.shs)
    out_f = np.zeros((P, 7), np.float32)
    out_i = np.zeros(P, np.int32)
    lib().hc_preprocess_inte(ctypes.c_int(P), ctypes.c_int(s.sh_degree), ctypes.c_int(shs.shape[1]), _p(_f(s.means3D)), _p(_f(s.scales)),
                             _p(_f(s.rotations)), _p(_f(s.opacities)), _p(shs), _p(_f(s.viewmatrix)), _p(_f(s.projmatrix)),
                             _p(_f(s.campos)), ctypes.c_int(s.W), ctypes.c_int(s.H), ctypes.c_float(s.tanfovx),
                             ctypes.c_float(s.tanfovy), ctypes.c_float(s.kernel_size), ctypes.c_float(1.0), _p(out_f), _p(out_i))
    return out_f, out_i


def preprocess_bwd(scene, radii, clamped, op_combined, acc, use_sh=True, cov3D=None):
    s = scene
    P = s.means3D.shape[0]
    means, scales, rots = _f(s.means3D), _f(s.scales), _f(s.rotations)
    shs = _f(s.shs) if use_sh else None
    M = 0 if shs is None else shs.shape[1]
    cov3D = _f(cov3D)
    if cov3D is not None:
        scales = rots = None
    out = np.zeros((P, 17), np.float32)
    dsh = np.zeros((P, max(M, 1), 3), np.float32)
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    clamped = np.ascontiguousarray(clamped, dtype=np.int32)
    L = lib()
    L.hc_preprocess_bwd(ctypes.c_int(P), ctypes.c_int(s.sh_degree), ctypes.c_int(M), _p(means), _p(scales), _p(rots), _p(cov3D),
                        _p(shs), _p(radii), _p(clamped), _p(_f(op_combined)), _p(_f(s.viewmatrix)), _p(_f(s.projmatrix)),
                        _p(_f(s.campos)), ctypes.c_int(s.W), ctypes.c_int(s.H), ctypes.c_float(s.tanfovx),
                        ctypes.c_float(s.tanfovy), ctypes.c_float(s.kernel_size), ctypes.c_float(1.0), _p(_f(acc)), _p(out),
                        _p(dsh) if shs is not None else None)
    return out, dsh


def block_masks(rec, tx0, ty0):
    """rec [n,6] = mx,my,cx,cy,cz,op -> (mask of ellipse_block_mask, brute-force truth), uint32 [n] each."""
    rec = _f(rec)
    out = np.zeros((rec.shape[0], 2), np.uint32)
    lib().hc_block_masks(ctypes.c_int(rec.shape[0]), _p(rec), ctypes.c_float(tx0), ctypes.c_float(ty0), _p(out))
    return out[:, 0], out[:, 1]


def block_masks_rect(rec, tx0, ty0, tx1, ty1):
    """as block_masks, asked per splat for a whole tile rectangle the way the instance emission does -> [n, tiles] mask, truth."""
    rec = _f(rec)
    tiles = (tx1 - tx0) * (ty1 - ty0)
    out = np.zeros((rec.shape[0], tiles, 2), np.uint32)
    lib().hc_block_masks_rect(ctypes.c_int(rec.shape[0]), _p(rec), ctypes.c_int(tx0), ctypes.c_int(ty0), ctypes.c_int(tx1), ctypes.c_int(ty1), _p(out))
    return out[:, :, 0], out[:, :, 1]
