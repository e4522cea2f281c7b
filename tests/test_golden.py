"""Frozen oracle outputs (tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU tier: the
oracle must still reproduce them bit-for-bit (fp32, single-threaded accumulation order).  GPU tier:
the HIP path must match them to the parity bar.  See make_golden.py for why these are oracle
fixtures and not reference outputs (parity unpinned upstream)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("case", list(make_golden.CASES))
def test_oracle_reproduces_golden(case):
    want = np.load(os.path.join(HERE, "golden", case + ".npz"))
    got = make_golden.run(case)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        if k.startswith("floor_"):
            assert abs(float(a) - float(b)) <= 1e-3 * float(b) + 1e-12, k
        elif a.dtype.kind == "f":
            # -ffp-contract=off + IEEE ops: identical across x86-64 hosts; double accumulators of the
            # blend backward are order-independent to well below one fp32 ulp
            assert np.allclose(a, b, rtol=1e-6, atol=1e-7), k
        else:
            assert np.array_equal(a, b), k


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(make_golden.CASES))
def test_hip_matches_golden(case):
    import torch
    from gpu_util import HipRun
    from synth_scene import make_scene, upstream_grads
    from util import ATOL, close
    want = np.load(os.path.join(HERE, "golden", case + ".npz"))
    s = make_scene(**make_golden.CASES[case])
    h = HipRun(s, "cuda:0")
    st = h.forward_native()
    torch.cuda.synchronize()
    R = int(want["num_rendered"])
    assert st[0] == R
    assert np.array_equal(st[8].cpu().numpy(), want["radii"])
    assert np.array_equal(h.export("point_list", torch.int32, R).view(np.uint32), want["point_list"])
    assert np.array_equal(h.export("n_contrib", torch.int32, 2 * s.H * s.W).view(np.uint32)[: s.H * s.W], want["n_contrib"][: s.H * s.W])
    for k, t in (("color", st[1]), ("coord", st[2]), ("mcoord", st[3]), ("alpha", st[4]), ("normal", st[5]), ("depth", st[6]), ("mdepth", st[7])):
        assert close(t.cpu().numpy(), want[k]).all(), k
    h2 = HipRun(s, "cuda:0")
    h2.forward()
    got = h2.backward(upstream_grads(s, make_golden.CASES[case]["seed"]))
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        b = want[k].reshape(got[k].shape)
        scale = float(np.abs(b).max()) + 1e-30
        band = ATOL + max(2e-6 * scale, 0.25 * float(want["floor_" + k]))  # see util.grad_noise_floor
        assert close(got[k], b, atol=band, rtol=1e-3).all(), k
