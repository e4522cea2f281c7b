"""TEST INFRASTRUCTURE: numpy restatement of one torch.optim.Adam step as the reference configures it
(scene/gaussian_model.py:349: default betas, eps=1e-15, no weight decay / amsgrad; per-group lr).  Pinned to golden
vectors produced by torch.optim.Adam itself on the CPU (tests/golden/make_golden_adam.py)."""
import numpy as np


def step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """t: step count after this update.  Returns new (p, m, v) in the dtype of p."""
    dt = p.dtype.type
    m = m + dt(1 - beta1) * (g - m)
    v = v * dt(beta2) + dt(1 - beta2) * (g * g)
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
    return p - dt(lr / bc1) * (m / denom), m, v
