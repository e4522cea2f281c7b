"""The accumulation scratch of the backward (RadegsBwdArgs.acc_reuse, include/radegs.h): ONE buffer per (device, stream) that the
binding zeroes once and every backward hands back zeroed (the per-Gaussian kernel clears each record it consumes), instead of a
10-us fill per call.  What must hold: (1) after any backward the cached buffer is all zeros -- for both blend formulations, with and
without the coord map, after a scene grew or shrank; (2) gradients taken through the reused buffer equal those taken through a fresh,
explicitly zeroed one; (3) a call whose contents are wanted afterwards (KEEP_ACC) does not touch the cached buffer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(s, seed, monkeypatch, streams=None):
    from gpu_util import HipRun
    from synth_scene import upstream_grads
    if streams is not None:
        monkeypatch.setenv("RADEGS_STREAMS", str(streams))
    h = HipRun(s, "cuda:0")
    h.forward()
    return h.backward(upstream_grads(s, seed))


def _scratch_is_zero(C):
    assert C._ACC_SCRATCH, "no cached accumulation scratch after a backward"
    return all(int(torch.count_nonzero(t)) == 0 for t in C._ACC_SCRATCH.values())


@pytest.mark.parametrize("coord", [False, True])
@pytest.mark.parametrize("streams", [0, 1])
def test_scratch_comes_back_zero_and_reuse_changes_nothing(coord, streams, monkeypatch):
    import diff_gaussian_rasterization._C as C
    from synth_scene import make_scene
    from util import close
    C._ACC_SCRATCH.clear()
    scenes = [make_scene(P, 192, 128, sh_degree=1, mu_px=px, seed=seed, kernel_size=0.1, require_coord=coord, require_depth=True, pose="random")
              for P, px, seed in ((3000, 3.0, 1), (5000, 1.5, 2), (800, 8.0, 3))]   # grows, then shrinks: the buffer is reused or regrown
    through_cache = []
    for k, s in enumerate(scenes):
        through_cache.append(_grads(s, k, monkeypatch, streams))
        assert _scratch_is_zero(C), f"scene {k}: the cached scratch is not all zeros after the backward"
    assert len(C._ACC_SCRATCH) == 1
    cached = next(iter(C._ACC_SCRATCH.values()))
    ptr = cached.data_ptr()
    for k, s in enumerate(scenes):   # the same backwards through a scratch of their own (filled with zeros by the call: acc_reuse = 0)
        C.KEEP_ACC = True
        try:
            fresh = _grads(s, k, monkeypatch, streams)
        finally:
            C.KEEP_ACC = False
            C.LAST_ACC = None
        assert next(iter(C._ACC_SCRATCH.values())).data_ptr() == ptr and _scratch_is_zero(C)   # untouched by a KEEP_ACC call
        for name, g in through_cache[k].items():
            if g is None:
                continue
            scale = float(np.abs(fresh[name]).max()) + 1e-30
            # two runs of the same kernels: only the order of the float atomics differs
            assert close(g, fresh[name], atol=2e-5 * scale, rtol=1e-3).all(), (k, name)
