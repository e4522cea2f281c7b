"""HIP-backed mirror of GaussianModel.get_scaling_n_opacity_with_3D_filter (scene/gaussian_model.py:156-166), the
per-Gaussian step gaussian_renderer.render() runs right before the rasterizer (gaussian_renderer/__init__.py:63).
SURVEY 8f N3.  One streaming kernel per direction instead of ~10 / ~20 eager torch kernels.  GPU only."""
import ctypes

import torch

from diff_gaussian_rasterization import _C

_bound = False


def _lib():
    global _bound
    L = _C.library()
    if not _bound:
        vp = ctypes.c_void_p
        L.radegs_filter3d_forward.restype = ctypes.c_int
        L.radegs_filter3d_forward.argtypes = [ctypes.c_int] + [vp] * 6
        L.radegs_filter3d_backward.restype = ctypes.c_int
        L.radegs_filter3d_backward.argtypes = [ctypes.c_int] + [vp] * 8
        L.radegs_compute_filter3d.restype = ctypes.c_int
        L.radegs_compute_filter3d.argtypes = [ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_float, vp, vp, vp, vp]
        _bound = True
    return L


def _prep(t, name, cols):
    _C._require_gpu(t, name)
    if t.dtype != torch.float32 or t.dim() != 2 or t.size(1) != cols:
        raise RuntimeError(f"`{name}` must be float32 of shape (P,{cols})")
    return t.contiguous()


class _ScalingOpacity3DFilter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling_raw, opacity_raw, filter_3D):
        sc, op, f3 = _prep(scaling_raw, "_scaling", 3), _prep(opacity_raw, "_opacity", 1), _prep(filter_3D, "filter_3D", 1)
        P = sc.size(0)
        if op.size(0) != P or f3.size(0) != P:
            raise RuntimeError("_scaling, _opacity and filter_3D must have the same number of rows")
        scales, opacity = torch.empty_like(sc), torch.empty_like(op)
        with torch.cuda.device(sc.device):
            rc = _lib().radegs_filter3d_forward(P, _C._ptr(sc), _C._ptr(op), _C._ptr(f3), _C._ptr(scales), _C._ptr(opacity), _C._stream(sc.device))
        if rc != 0:
            raise RuntimeError(f"radegs_filter3d_forward failed ({rc})")
        ctx.save_for_backward(sc, op, f3)
        return scales, opacity

    @staticmethod
    def backward(ctx, g_scales, g_opacity):
        sc, op, f3 = ctx.saved_tensors
        gs = None if g_scales is None else g_scales.contiguous()
        go = None if g_opacity is None else g_opacity.contiguous()
        g_sc, g_op = torch.empty_like(sc), torch.empty_like(op)
        with torch.cuda.device(sc.device):
            rc = _lib().radegs_filter3d_backward(sc.size(0), _C._ptr(sc), _C._ptr(op), _C._ptr(f3), _C._ptr(gs), _C._ptr(go), _C._ptr(g_sc),
                                                 _C._ptr(g_op), _C._stream(sc.device))
        if rc != 0:
            raise RuntimeError(f"radegs_filter3d_backward failed ({rc})")
        return g_sc, g_op, None


def scaling_n_opacity_with_3D_filter(scaling_raw, opacity_raw, filter_3D):
    """(scales[P,3], opacity[P,1]) = GaussianModel.get_scaling_n_opacity_with_3D_filter evaluated on the raw parameters
    `_scaling` (log-space), `_opacity` (logit) and the `filter_3D` buffer.  Differentiable w.r.t. the two parameters."""
    return _ScalingOpacity3DFilter.apply(scaling_raw, opacity_raw, filter_3D)


@torch.no_grad()
def compute_3D_filter(xyz, cameras):
    """GaussianModel.compute_3D_filter (scene/gaussian_model.py:179-232): returns the (P,1) `filter_3D` buffer for the
    Gaussian centres `xyz` and an iterable of cameras (attributes R, T, image_width, image_height, FoVx, FoVy), all cameras
    in one kernel instead of ~15 torch kernels per camera."""
    import math

    import numpy as np
    _C._require_gpu(xyz, "xyz")
    x = _prep(xyz.detach(), "xyz", 3)
    rows, focal_length = [], 0.0
    for cam in cameras:
        W, H = cam.image_width, cam.image_height
        fx = W / (2 * math.tan(cam.FoVx / 2.))
        fy = H / (2 * math.tan(cam.FoVy / 2.))
        rows.append(np.concatenate([np.asarray(cam.R, dtype=np.float32).reshape(9), np.asarray(cam.T, dtype=np.float32).reshape(3),
                                    np.array([fx, fy, W, H], dtype=np.float32)]))
        focal_length = max(focal_length, fx)
    if not rows:
        raise RuntimeError("compute_3D_filter needs at least one camera")
    table = torch.from_numpy(np.stack(rows)).to(x.device)
    P = x.size(0)
    dist = torch.empty(P, dtype=torch.float32, device=x.device)
    mx = torch.empty(1, dtype=torch.int32, device=x.device)
    out = torch.empty((P, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib().radegs_compute_filter3d(P, _C._ptr(x), len(rows), _C._ptr(table), float(focal_length), _C._ptr(dist), _C._ptr(mx),
                                            _C._ptr(out), _C._stream(x.device))
    if rc != 0:
        raise RuntimeError(f"radegs_compute_filter3d failed ({rc})")
    return out
