"""HIP-backed mirror of the reference's photometric loss:

    l1_loss(network_output, gt)        utils/loss_utils.py:17-18
    ssim(img1, img2)                   utils/loss_utils.py:31-63   (window 11, size_average=True)
    photometric_loss(image, gt, lambda_dssim)  =  (1-l)*l1_loss + l*(1-ssim)     train.py:159, fused

(C,H,W) float32 GPU tensors; `gt` may carry a leading batch dimension of 1 as train.py passes it.  Differentiable
w.r.t. the first argument only (the ground-truth image is data).  One tiled kernel per direction through
libradegs_hip.so (radegs_photometric_*).  GPU only: there is no CPU path."""
import ctypes

import torch

from diff_gaussian_rasterization import _C

_bound = False


def _lib():
    global _bound
    L = _C.library()
    if not _bound:
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.radegs_photometric_scratch_bytes.restype = ctypes.c_size_t
        L.radegs_photometric_scratch_bytes.argtypes = [ci, ci, ci]
        L.radegs_photometric_forward.restype = ci
        L.radegs_photometric_forward.argtypes = [ci, ci, ci, vp, vp, ctypes.c_float, vp, vp, vp, vp]
        L.radegs_photometric_backward.restype = ci
        L.radegs_photometric_backward.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp]
        _bound = True
    return L


def _prep(img, gt):
    _C._require_gpu(img, "image")
    _C._require_gpu(gt, "gt")
    if gt.dim() == img.dim() + 1 and gt.size(0) == 1:
        gt = gt[0]
    if img.dim() != 3 or gt.shape != img.shape:
        raise RuntimeError(f"image and gt must both be (C,H,W); got {tuple(img.shape)} and {tuple(gt.shape)}")
    if img.dtype != torch.float32 or gt.dtype != torch.float32:
        raise RuntimeError("image and gt must be float32")
    return img.contiguous(), gt.contiguous()


class _Photometric(torch.autograd.Function):
    """returns (loss, l1, ssim); the backward folds the three upstream gradients into two coefficients"""

    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        a, b = _prep(image, gt)
        C, H, W = a.shape
        L = _lib()
        dev = a.device
        need = ctx.needs_input_grad[0]
        scratch = torch.empty(L.radegs_photometric_scratch_bytes(W, H, C), dtype=torch.uint8, device=dev)
        dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=dev) if need else None
        out = torch.empty(3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.radegs_photometric_forward(W, H, C, _C._ptr(a), _C._ptr(b), float(lambda_dssim), _C._ptr(scratch), _C._ptr(dmaps),
                                              _C._ptr(out), _C._stream(dev))
        if rc != 0:
            raise RuntimeError(f"radegs_photometric_forward failed ({rc})")
        ctx.lam = float(lambda_dssim)
        ctx.save_for_backward(a, b, dmaps)
        ctx.mark_non_differentiable()
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loss, g_l1, g_ssim):
        a, b, dmaps = ctx.saved_tensors
        coef = torch.stack([g_loss * (1.0 - ctx.lam) + g_l1, g_ssim - g_loss * ctx.lam]).to(torch.float32).contiguous()
        grad = torch.empty_like(a)
        C, H, W = a.shape
        with torch.cuda.device(a.device):
            rc = _lib().radegs_photometric_backward(W, H, C, _C._ptr(a), _C._ptr(b), _C._ptr(dmaps), _C._ptr(coef), _C._ptr(grad),
                                                    _C._stream(a.device))
        if rc != 0:
            raise RuntimeError(f"radegs_photometric_backward failed ({rc})")
        return grad, None, None


def photometric_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt)) -- train.py:159."""
    return _Photometric.apply(image, gt, lambda_dssim)[0]


def l1_loss(network_output, gt):
    return _Photometric.apply(network_output, gt, 0.0)[1]


def ssim(img1, img2, window_size=11, size_average=True):
    if window_size != 11 or not size_average:
        raise NotImplementedError("only the configuration train.py uses (window_size=11, size_average=True) is built")
    return _Photometric.apply(img1, img2, 1.0)[2]
