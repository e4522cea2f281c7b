"""`simple_knn._C` -- `distCUDA2(points)`: mean squared distance of every point to its 3 nearest neighbours, as the reference
uses it to initialise the Gaussian scales (scene/gaussian_model.py:315).  The reference's own implementation lives in the
un-vendored `simple-knn` submodule (CUDA); this one calls libradegs_hip.so (radegs_knn_mean_dist2).  GPU only."""
import ctypes

import torch

from diff_gaussian_rasterization import _C as _radegs

_bound = False


def _lib():
    global _bound
    L = _radegs.library()
    if not _bound:
        L.radegs_knn_scratch_bytes.restype = ctypes.c_size_t
        L.radegs_knn_scratch_bytes.argtypes = [ctypes.c_int]
        L.radegs_knn_mean_dist2.restype = ctypes.c_int
        L.radegs_knn_mean_dist2.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _bound = True
    return L


def distCUDA2(points):
    _radegs._require_gpu(points, "points")
    if points.dim() != 2 or points.size(1) != 3 or points.dtype != torch.float32:
        raise RuntimeError("points must be a float32 tensor of shape (P,3)")
    pts = points.contiguous()
    P = int(pts.size(0))
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    L = _lib()
    scratch = torch.empty(L.radegs_knn_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = L.radegs_knn_mean_dist2(P, _radegs._ptr(pts), _radegs._ptr(scratch), _radegs._ptr(out), _radegs._stream(pts.device))
    if rc != 0:
        raise RuntimeError(f"radegs_knn_mean_dist2 failed ({rc})")
    return out
