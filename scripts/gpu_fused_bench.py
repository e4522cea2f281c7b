"""Times the fused pre/post steps against the same math written as eager torch ops the way the
reference does (utils/graphics_utils.py:97-127 + train.py:152-155; scene/gaussian_model.py:156-166).  GPU box only."""
import math, os, sys, time
from collections import namedtuple
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "rade-gs_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import graphics_utils as gu
import gaussian_model_ops as gmo

dev = torch.device("cuda:0")
View = namedtuple("View", "image_width image_height FoVx FoVy")
W, H = 1920, 1080
view = View(W, H, 1.0, 2 * math.atan(math.tan(0.5) * H / W))


def timeit(fn, n=50):
    """ms per call: the larger of wall-clock and GPU-event time over n back-to-back calls (host- or device-bound)"""
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return max((time.perf_counter() - t0) * 1e3, e0.elapsed_time(e1)) / n


# ---- eager restatement of the reference step (timing baseline only) ----
def eager_depth_double_to_normal(view, d1, d2):
    fx = W / (2 * math.tan(view.FoVx / 2.)); fy = H / (2 * math.tan(view.FoVy / 2.))
    k = torch.tensor([[1 / fx, 0., -W / (2 * fx)], [0., 1 / fy, -H / (2 * fy)], [0., 0., 1.0]]).float().to(dev)
    gx, gy = torch.meshgrid(torch.arange(W) + 0.5, torch.arange(H) + 0.5, indexing='xy')
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=0).reshape(3, -1).float().to(dev)
    rays = k @ pts
    p = torch.stack([(d1.reshape(1, -1) * rays).reshape(3, H, W), (d2.reshape(1, -1) * rays).reshape(3, H, W)], dim=0)
    out = torch.zeros_like(p)
    dx = p[..., 2:, 1:-1] - p[..., :-2, 1:-1]
    dy = p[..., 1:-1, 2:] - p[..., 1:-1, :-2]
    out[..., 1:-1, 1:-1] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=1), dim=1)
    return out


d1 = (4 + torch.rand(1, H, W, device=dev)).requires_grad_(True)
d2 = (4 + torch.rand(1, H, W, device=dev)).requires_grad_(True)
rn = torch.nn.functional.normalize(torch.randn(3, H, W, device=dev), dim=0).requires_grad_(True)


def eager_step():
    nm = eager_depth_double_to_normal(view, d1, d2)
    err = 1 - (rn.unsqueeze(0) * nm).sum(dim=1)
    loss = 0.4 * err[0].mean() + 0.6 * err[1].mean()
    loss.backward()


def fused_step():
    gu.normal_consistency_loss(view, rn, d1, d2, 0.6).backward()


te, tf = timeit(eager_step), timeit(fused_step)
# algorithmic bytes: fwd reads 2 depth + 3 normal maps (20 B/px); bwd reads them again and writes 2 + 3 maps (40 B/px)
gb = 60.0 * W * H / 1e9
print(f"normal-consistency loss fwd+bwd @1080p: eager torch {te:.3f} ms, fused HIP {tf:.3f} ms ({te/tf:.1f}x); "
      f"fused = {gb / (tf * 1e-3):.0f} GB/s algorithmic incl. autograd/launch overhead")

P = 1_000_000
sc = (torch.randn(P, 3, device=dev) - 4.6).requires_grad_(True)
op = torch.randn(P, 1, device=dev).requires_grad_(True)
f3 = 0.001 + 0.02 * torch.rand(P, 1, device=dev)


def eager_filter():
    opacity = torch.sigmoid(op)
    scales = torch.exp(sc)
    s2 = torch.square(scales)
    det1 = s2.prod(dim=1)
    a2 = s2 + torch.square(f3)
    det2 = a2.prod(dim=1)
    coef = torch.sqrt(det1 / det2)
    s, o = torch.sqrt(a2), opacity * coef[..., None]
    (s.sum() + o.sum()).backward()


def fused_filter():
    s, o = gmo.scaling_n_opacity_with_3D_filter(sc, op, f3)
    (s.sum() + o.sum()).backward()


te, tf = timeit(eager_filter), timeit(fused_filter)
print(f"3D filter + activations fwd+bwd, P=1M: eager torch {te:.3f} ms, fused HIP {tf:.3f} ms ({te/tf:.1f}x)")

# ---- photometric loss ----
import torch.nn.functional as F
import loss_utils as lu
from math import exp
img = torch.rand(3, H, W, device=dev).requires_grad_(True)
gt = torch.rand(3, H, W, device=dev)


def eager_ssim(img1, img2):
    g = torch.Tensor([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(3, 1, 11, 11).contiguous().to(dev)
    mu1 = F.conv2d(img1, window, padding=5, groups=3); mu2 = F.conv2d(img2, window, padding=5, groups=3)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=5, groups=3) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=5, groups=3) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=5, groups=3) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


def eager_photo():
    (0.8 * torch.abs(img - gt).mean() + 0.2 * (1.0 - eager_ssim(img, gt.unsqueeze(0)))).backward()


def fused_photo():
    lu.photometric_loss(img, gt, 0.2).backward()


te, tf = timeit(eager_photo), timeit(fused_photo)
print(f"L1 + SSIM loss fwd+bwd @1080p: eager torch {te:.3f} ms, fused HIP {tf:.3f} ms ({te/tf:.1f}x)")

# ---- Adam over the six per-Gaussian groups, P = 1M ----
import fused_adam
shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]


def make(opt_cls):
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p)
    return opt_cls([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15)


o_ref, o_fused = make(torch.optim.Adam), make(fused_adam.Adam)
te, tf = timeit(o_ref.step), timeit(o_fused.step)
gb = 59 * P * 28 / 1e9
print(f"Adam step, 59 floats x 1M Gaussians: torch.optim.Adam {te:.3f} ms, fused HIP {tf:.3f} ms ({te/tf:.1f}x) = {gb / (tf * 1e-3):.0f} GB/s")
