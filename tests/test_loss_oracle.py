"""oracle/loss_oracle.py against golden vectors from the REFERENCE's l1_loss / ssim + torch autograd
(tests/golden/make_golden_losses.py): this oracle row is PINNED to the reference."""
import os

import numpy as np
import pytest

from oracle import loss_oracle as lo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["small", "wide"])
def test_photometric_loss_and_gradient_match_reference(name):
    z = np.load(os.path.join(GOLD, f"losses_{name}.npz"))
    for dt in (np.float32, np.float64):
        img, gt = z["img"].astype(dt), z["gt"].astype(dt)
        assert abs(lo.l1_loss(img, gt) - float(z["l1"])) < 1e-6
        assert abs(lo.ssim(img, gt) - float(z["ssim"])) < 5e-6
        assert abs(lo.rgb_loss(img, gt, 0.2) - float(z["loss"])) < 2e-6
    g = lo.rgb_loss_bwd(z["img"].astype(np.float64), z["gt"].astype(np.float64), 0.2)
    ref = z["grad"]
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() < 1e-4 * scale, np.abs(g - ref).max() / scale
