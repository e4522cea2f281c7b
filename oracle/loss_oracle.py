"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of the photometric loss of train.py:159
    rgb_loss = (1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt))
with l1_loss / ssim from utils/loss_utils.py:17-63 (11x11 Gaussian window, sigma 1.5, zero padding 5, per channel) and
the hand-derived gradient w.r.t. `image`.  Pinned to golden vectors produced by the reference's own functions with torch
autograd on the CPU (tests/golden/make_golden_losses.py -> tests/golden/losses_*.npz -> tests/test_loss_oracle.py)."""
from math import exp

import numpy as np

C1, C2 = 0.01 ** 2, 0.03 ** 2


def window_1d(size=11, sigma=1.5):
    """loss_utils.py:23-25 (float32 arithmetic of torch.Tensor / sum)"""
    g = np.array([exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)], dtype=np.float32)
    return g / g.sum(dtype=np.float32)


def blur_matrix(n, dtype, size=11):
    """(n,n) matrix of the zero-padded 1D correlation with the window"""
    w = window_1d(size).astype(dtype)
    K = np.zeros((n, n), dtype=dtype)
    for i in range(n):
        for k in range(size):
            j = i + k - size // 2
            if 0 <= j < n:
                K[i, j] = w[k]
    return K


def _pieces(img1, img2):
    dt = img1.dtype.type
    H, W = img1.shape[-2:]
    Kh, Kw = blur_matrix(H, dt), blur_matrix(W, dt)
    # the reference's 2D window is the outer product of the 1D one rounded to float32; the separable form differs by ~1e-8
    blur = lambda x: np.matmul(np.matmul(Kh, x), Kw.T)   # (H,H) @ (C,H,W) @ (W,W): BLAS, fine at 1080p
    mu1, mu2 = blur(img1), blur(img2)
    s11, s22, s12 = blur(img1 * img1) - mu1 * mu1, blur(img2 * img2) - mu2 * mu2, blur(img1 * img2) - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + dt(C1), 2 * s12 + dt(C2)
    B1, B2 = mu1 * mu1 + mu2 * mu2 + dt(C1), s11 + s22 + dt(C2)
    return blur, mu1, mu2, A1, A2, B1, B2


def ssim(img1, img2):
    _, _, _, A1, A2, B1, B2 = _pieces(img1, img2)
    return ((A1 * A2) / (B1 * B2)).mean(dtype=np.float64)


def l1_loss(img1, img2):
    return np.abs(img1 - img2).mean(dtype=np.float64)


def rgb_loss(img, gt, lambda_dssim=0.2):
    return (1.0 - lambda_dssim) * l1_loss(img, gt) + lambda_dssim * (1.0 - ssim(img, gt))


def rgb_loss_bwd(img, gt, lambda_dssim=0.2, upstream=1.0):
    """d rgb_loss / d img"""
    blur, mu1, mu2, A1, A2, B1, B2 = _pieces(img, gt)
    n = img.size
    d_mu1 = ((2 * mu2 * A2 - 2 * mu2 * A1) * (B1 * B2) - (A1 * A2) * (2 * mu1 * B2 - 2 * mu1 * B1)) / np.square(B1 * B2)
    d_e11 = -(A1 * A2) / (B1 * B2 * B2)
    d_e12 = 2 * A1 / (B1 * B2)
    g_ssim = (blur(d_mu1) + 2 * img * blur(d_e11) + gt * blur(d_e12)) / n
    g_l1 = np.sign(img - gt) / n
    return upstream * ((1.0 - lambda_dssim) * g_l1 - lambda_dssim * g_ssim)
