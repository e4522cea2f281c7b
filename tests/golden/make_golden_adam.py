"""tests/golden/adam.npz: three steps of torch.optim.Adam(l, lr=0.0, eps=1e-15) -- the optimizer the reference builds at
scene/gaussian_model.py:338-349 -- on the CPU, with per-group learning rates like the reference's and one lr change
between steps (update_learning_rate).  Build container or anywhere torch runs."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(0)
shapes = {"xyz": (1000, 3), "f_dc": (1000, 1, 3), "f_rest": (1000, 15, 3), "opacity": (1000, 1), "scaling": (1000, 3), "rotation": (1000, 4)}
lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001}
params = {k: torch.nn.Parameter(torch.from_numpy(rng.standard_normal(s).astype(np.float32))) for k, s in shapes.items()}
opt = torch.optim.Adam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in shapes], lr=0.0, eps=1e-15)
out = {f"p0_{k}": v.detach().numpy().copy() for k, v in params.items()}
for it in range(3):
    for k, v in params.items():
        g = (rng.standard_normal(shapes[k]) * 10.0 ** rng.uniform(-6, 0)).astype(np.float32)
        if it == 1 and k == "opacity":
            g[:] = 0           # zero gradient rows: the eps=1e-15 regime
        v.grad = torch.from_numpy(g)
        out[f"g{it}_{k}"] = g
    if it == 2:
        opt.param_groups[0]["lr"] = 1.0e-4   # update_learning_rate
    opt.step()
    for k, v in params.items():
        out[f"p{it + 1}_{k}"] = v.detach().numpy().copy()
for k, v in params.items():
    out[f"m_{k}"] = opt.state[v]["exp_avg"].numpy().copy()
    out[f"v_{k}"] = opt.state[v]["exp_avg_sq"].numpy().copy()
np.savez_compressed(os.path.join(HERE, "adam.npz"), **out)
print("ok")
