"""Generates tests/golden/losses_*.npz by running the REFERENCE's l1_loss / ssim (/root/reference/utils/loss_utils.py)
on the CPU with torch autograd, combined as in train.py:159.  Build container only."""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_loss_utils", "/root/reference/utils/loss_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def case(name, H, W, seed, noise):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    gt = np.stack([0.5 + 0.4 * np.sin(x / 5.0 + c) * np.cos(y / 4.0 - c) for c in range(3)], 0).astype(np.float32)
    img = np.clip(gt + noise * rng.standard_normal((3, H, W)), 0, 1).astype(np.float32)
    image = torch.from_numpy(img).requires_grad_(True)
    gt_image = torch.from_numpy(gt)
    lambda_dssim = 0.2
    Ll1 = ref.l1_loss(image, gt_image)
    s = ref.ssim(image, gt_image.unsqueeze(0))          # train.py:159 passes gt with a batch dimension
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - s)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, f"losses_{name}.npz"), img=img, gt=gt, l1=np.float32(Ll1.item()), ssim=np.float32(s.item()),
                        loss=np.float32(loss.item()), grad=image.grad.numpy())
    print(name, float(Ll1), float(s), float(loss))


if __name__ == "__main__":
    case("small", 37, 50, 1, 0.05)      # narrower than two windows in places: padding everywhere
    case("wide", 48, 150, 2, 0.15)
