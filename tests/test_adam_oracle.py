"""oracle/adam_oracle.py against torch.optim.Adam's own CPU results (tests/golden/adam.npz)."""
import os

import numpy as np

from oracle import adam_oracle as ao

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adam.npz"))
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}


def test_three_steps_match_torch():
    for k, lr in LRS.items():
        p = Z[f"p0_{k}"].copy()
        m, v = np.zeros_like(p), np.zeros_like(p)
        for it in range(3):
            cur = 1.0e-4 if (it == 2 and k == "xyz") else lr
            p, m, v = ao.step(p, Z[f"g{it}_{k}"], m, v, it + 1, cur)
            ref = Z[f"p{it + 1}_{k}"]
            assert np.abs(p - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()) + 2e-7, (k, it, np.abs(p - ref).max())
        assert np.allclose(m, Z[f"m_{k}"], rtol=1e-5, atol=1e-12) and np.allclose(v, Z[f"v_{k}"], rtol=1e-5, atol=1e-20)
