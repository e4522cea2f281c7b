"""HIP-backed mirror of the reference's depth->normal helpers and the normal-consistency loss:

    depth_double_to_normal(view, depth1, depth2)       utils/graphics_utils.py:125-127
    point_double_to_normal(view, points1, points2)     utils/graphics_utils.py:116-123
    normal_consistency_loss(view, rendered_normal, map1, map2, depth_ratio=0.6)     train.py:146-155, fused

Same names, arguments (`view` needs image_width, image_height, FoVx, FoVy) and results as upstream; differentiable
(torch.autograd.Function over libradegs_hip.so's radegs_normals_* / radegs_normal_loss_* entry points).  One kernel per
direction instead of ~10 / ~25 eager torch kernels.  GPU only: there is no CPU path."""
import ctypes

import torch

from diff_gaussian_rasterization import _C


class RadegsNormalArgs(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("points", ctypes.c_int), ("fovx", ctypes.c_double),
                ("fovy", ctypes.c_double), ("map1", ctypes.c_void_p), ("map2", ctypes.c_void_p)]


_bound = False


def _lib():
    global _bound
    L = _C.library()
    if not _bound:
        vp, ap = ctypes.c_void_p, ctypes.POINTER(RadegsNormalArgs)
        L.radegs_normals_forward.restype = ctypes.c_int
        L.radegs_normals_forward.argtypes = [ap, vp, vp]
        L.radegs_normals_backward.restype = ctypes.c_int
        L.radegs_normals_backward.argtypes = [ap, vp, vp, vp, vp]
        L.radegs_normal_loss_scratch_bytes.restype = ctypes.c_size_t
        L.radegs_normal_loss_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.radegs_normal_loss_forward.restype = ctypes.c_int
        L.radegs_normal_loss_forward.argtypes = [ap, vp, ctypes.c_float, vp, vp, vp]
        L.radegs_normal_loss_backward.restype = ctypes.c_int
        L.radegs_normal_loss_backward.argtypes = [ap, vp, ctypes.c_float, vp, vp, vp, vp, vp]
        L.radegs_normals_last_error.restype = ctypes.c_char_p
        _bound = True
    return L


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_lib().radegs_normals_last_error().decode()}")


def _prep(t, name, channels, H, W):
    _C._require_gpu(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"`{name}` must be float32")
    if t.numel() != channels * H * W:
        raise RuntimeError(f"`{name}` must have {channels}x{H}x{W} elements, got {tuple(t.shape)}")
    return t.contiguous()


def _args(view, m1, m2, points):
    return RadegsNormalArgs(int(view.image_width), int(view.image_height), int(points), float(view.FoVx), float(view.FoVy),
                            _C._ptr(m1), _C._ptr(m2))


class _DoubleToNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, view, map1, map2, points):
        W, H = int(view.image_width), int(view.image_height)
        c = 3 if points else 1
        m1, m2 = _prep(map1, "map1", c, H, W), _prep(map2, "map2", c, H, W)
        out = torch.empty((2, 3, H, W), dtype=torch.float32, device=m1.device)
        with torch.cuda.device(m1.device):
            _check(_lib().radegs_normals_forward(ctypes.byref(_args(view, m1, m2, points)), _C._ptr(out), _C._stream(m1.device)),
                   "radegs_normals_forward")
        ctx.view, ctx.points = view, points
        ctx.shapes = (map1.shape, map2.shape)
        ctx.save_for_backward(m1, m2)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        m1, m2 = ctx.saved_tensors
        g = grad_out.contiguous()
        g1, g2 = torch.empty_like(m1), torch.empty_like(m2)
        with torch.cuda.device(m1.device):
            _check(_lib().radegs_normals_backward(ctypes.byref(_args(ctx.view, m1, m2, ctx.points)), _C._ptr(g), _C._ptr(g1), _C._ptr(g2),
                                                  _C._stream(m1.device)), "radegs_normals_backward")
        return None, g1.view(ctx.shapes[0]), g2.view(ctx.shapes[1]), None


def depth_double_to_normal(view, depth1, depth2):
    """(2,3,H,W): normals of the expected- and median-depth maps (border pixels 0)."""
    return _DoubleToNormal.apply(view, depth1, depth2, False)


def point_double_to_normal(view, points1, points2):
    """(2,3,H,W): normals of two (3,H,W) coordinate maps (border pixels 0)."""
    return _DoubleToNormal.apply(view, points1, points2, True)


class _NormalConsistencyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, view, rendered_normal, map1, map2, depth_ratio, points):
        W, H = int(view.image_width), int(view.image_height)
        c = 3 if points else 1
        m1, m2, rn = _prep(map1, "map1", c, H, W), _prep(map2, "map2", c, H, W), _prep(rendered_normal, "rendered_normal", 3, H, W)
        L = _lib()
        scratch = torch.empty(L.radegs_normal_loss_scratch_bytes(W, H), dtype=torch.uint8, device=m1.device)
        out = torch.empty(3, dtype=torch.float32, device=m1.device)
        with torch.cuda.device(m1.device):
            _check(L.radegs_normal_loss_forward(ctypes.byref(_args(view, m1, m2, points)), _C._ptr(rn), float(depth_ratio), _C._ptr(scratch),
                                                _C._ptr(out), _C._stream(m1.device)), "radegs_normal_loss_forward")
        ctx.view, ctx.points, ctx.depth_ratio = view, points, float(depth_ratio)
        ctx.shapes = (rendered_normal.shape, map1.shape, map2.shape)
        ctx.save_for_backward(m1, m2, rn)
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        m1, m2, rn = ctx.saved_tensors
        up = grad_loss.to(torch.float32).reshape(1).contiguous()
        g1, g2, grn = torch.empty_like(m1), torch.empty_like(m2), torch.empty_like(rn)
        with torch.cuda.device(m1.device):
            _check(_lib().radegs_normal_loss_backward(ctypes.byref(_args(ctx.view, m1, m2, ctx.points)), _C._ptr(rn), ctx.depth_ratio,
                                                      _C._ptr(up), _C._ptr(g1), _C._ptr(g2), _C._ptr(grn), _C._stream(m1.device)),
                   "radegs_normal_loss_backward")
        return None, grn.view(ctx.shapes[0]), g1.view(ctx.shapes[1]), g2.view(ctx.shapes[2]), None, None


def normal_consistency_loss(view, rendered_normal, map1, map2, depth_ratio=0.6, points=False):
    """train.py:146-155 in one kernel each way:
        N = depth_double_to_normal(view, map1, map2)           (point_double_to_normal when points=True)
        err = 1 - (rendered_normal[None] * N).sum(1);  loss = (1-depth_ratio) * err[0].mean() + depth_ratio * err[1].mean()
    Returns the scalar loss (differentiable w.r.t. rendered_normal, map1, map2)."""
    return _NormalConsistencyLoss.apply(view, rendered_normal, map1, map2, depth_ratio, bool(points))
