"""Drop-in for `torch.optim.Adam(l, lr=0.0, eps=1e-15)` as scene/gaussian_model.py:338-349 uses it: the
same param_groups / state layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so the reference's optimizer-state
surgery (replace_tensor_to_optimizer, _prune_optimizer, cat_tensors_to_optimizer, state_dict round trips) works
unchanged -- but `step()` is ONE HIP launch over all parameter tensors (libradegs_hip.so: radegs_adam_step).
Not supported (raises): weight_decay, amsgrad, maximize, sparse or non-float32 / non-GPU parameters."""
import ctypes

import torch

from diff_gaussian_rasterization import _C

MAX_TENSORS = 16


class RadegsAdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("numel", ctypes.c_ulonglong), ("lr", ctypes.c_float), ("step", ctypes.c_double)]


_bound = False


def _lib():
    global _bound
    L = _C.library()
    if not _bound:
        L.radegs_adam_step.restype = ctypes.c_int
        L.radegs_adam_step.argtypes = [ctypes.c_int, ctypes.POINTER(RadegsAdamTensor), ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_void_p]
        _bound = True
    return L


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise NotImplementedError("fused Adam: weight_decay / amsgrad / maximize are not used by the reference and not built")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib()
        batches = {}  # (device, betas, eps) -> list of RadegsAdamTensor (+ keep-alive refs)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("fused Adam does not support sparse gradients")
                _C._require_gpu(p, "parameter")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("fused Adam needs contiguous float32 parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad.contiguous()
                if not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous():
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
                key = (p.device, tuple(group["betas"]), float(group["eps"]))
                batches.setdefault(key, []).append((RadegsAdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                                     st["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]),
                                                                     float(st["step"])), g))
        for (dev, betas, eps), items in batches.items():
            for i in range(0, len(items), MAX_TENSORS):
                chunk = items[i:i + MAX_TENSORS]
                arr = (RadegsAdamTensor * len(chunk))(*[c[0] for c in chunk])
                with torch.cuda.device(dev):
                    rc = L.radegs_adam_step(len(chunk), arr, float(betas[0]), float(betas[1]), eps, _C._stream(dev))
                if rc != 0:
                    raise RuntimeError(f"radegs_adam_step failed ({rc})")
        return loss
