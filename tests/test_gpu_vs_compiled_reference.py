"""The HIP path against the COMPILED REFERENCE directly (oracle/_ref: the reference's own forward.cu / backward.cu / rasterizer_impl.cu
built for the host, tests/test_ref_parity.py), with no hand-written oracle in between, at a size beyond the committed golden vectors:
a C2-shaped scene of 100 000 Gaussians at 608 x 342 and every BASELINE config with a GPU (C2, C3, C4, C5) at its named size.  The prebuilt library travels to the GPU box with the snapshot (it is built where
/root/reference exists); without it the test skips and tests/test_golden.py's reference-produced vectors carry the statement."""
import numpy as np
import pytest
import torch

from oracle import ref
from synth_scene import make_scene, upstream_grads
from util import ATOL, close, frac_close

pytestmark = [pytest.mark.gpu, pytest.mark.executed_grad,
              pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libradegs_ref.so was not built (no /root/reference on the build host)")]


@pytest.mark.parametrize("coord,depth,ks,full", [(False, True, 0.0, None), (True, True, 0.1, None), (False, True, 0.0, "C2"),
                                                 (False, True, 0.0, "C3"), (True, False, 0.0, "C4"), (False, True, 0.0, "C5")],
                         ids=["100k_depth_ks0", "100k_both_ks0.1", "C2_full_size", "C3_full_size", "C4_full_size", "C5_full_size"])
def test_hip_equals_the_references_own_code(coord, depth, ks, full):
    from gpu_util import HipRun
    from synth_scene import make_config
    from test_ref_parity import ref_for
    if full:    # a BASELINE.json config at its named size through the reference's own code on the host cores (C2: ~7 s, C5: minutes)
        if full in ("C4", "C5") and __import__("os").environ.get("RADEGS_SKIP_FULL_ORACLE", "0") == "1":
            pytest.skip("RADEGS_SKIP_FULL_ORACLE=1")
        s = make_config(full)
    else:
        s = make_scene(100_000, 608, 342, sh_degree=3, mu_px=1.5, seed=7, kernel_size=ks, require_coord=coord, require_depth=depth)
    ref.set_exp("spec")
    ref.set_num_threads(__import__("os").cpu_count() or 8)
    try:
        r = ref_for(s)
        R = r.forward()
        h = HipRun(s, "cuda:0")
        st = h.forward_native()
        torch.cuda.synchronize()
        # ---- indices: exact ----
        assert st[0] == R
        assert np.array_equal(st[8].cpu().numpy(), r.get("radii"))
        assert np.array_equal(h.export("tiles_touched", torch.int32, s.means3D.shape[0]).view(np.uint32), r.get("tiles_touched"))
        assert np.array_equal(h.export("point_list", torch.int32, R).view(np.uint32), r.get("point_list"))
        ntiles = ((s.W + 15) // 16) * ((s.H + 15) // 16)
        assert np.array_equal(h.export("ranges", torch.int32, 2 * ntiles).view(np.uint32), r.get("ranges"))
        nc = h.export("n_contrib", torch.int32, 2 * s.H * s.W).view(np.uint32)
        assert np.array_equal(nc, r.get("n_contrib"))
        # ---- maps: 1e-5 abs / 1e-4 rel ----
        want = r.outputs()
        for k, t in (("color", (st[1], want[0])), ("coord", (st[2], want[2])), ("mcoord", (st[3], want[3])), ("alpha", (st[4], want[6])),
                     ("normal", (st[5], want[7])), ("depth", (st[6], want[4])), ("mdepth", (st[7], want[5]))):
            a, b = t[0].cpu().numpy(), t[1]
            bad = ~close(a, b)
            assert not bad.any(), f"{k}: {int(bad.sum())} elements outside 1e-5/1e-4, max |diff| {float(np.abs(a - b).max()):.3e}"
        # ---- gradients of the backward the reference executes (its float atomics land in another order than ours: the band is the
        # fp32 conditioning of the sums, as everywhere else; geometry gradients carry the slip term's order noise, conftest) ----
        g = upstream_grads(s, 7)
        del want
        r.backward(g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
        want_g = r.grads()
        h2 = HipRun(s, "cuda:0")
        h2.forward()
        got = h2.backward(g)
        for k in ("dL_dmeans2D", "dL_dopacity", "dL_dsh"):
            b = want_g[k].reshape(got[k].shape)
            scale = float(np.abs(b).max()) + 1e-30
            assert frac_close(got[k], b) > 0.99, (k, frac_close(got[k], b))
            assert close(got[k], b, atol=ATOL + 1e-4 * scale, rtol=1e-3).all(), (k, float(np.abs(got[k] - b).max()), scale)
        for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations"):
            b = want_g[k].reshape(got[k].shape)
            scale = float(np.abs(b).max()) + 1e-30
            assert frac_close(got[k], b) > 0.97, (k, frac_close(got[k], b))
            # at kernel_size 0 the slip term is a cancellation residue scaled by an accumulated sum: the reference's own value moves by
            # up to ~1e-3 of the scale with the order of its atomics, with a heavy tail over a million Gaussians and C5's long sums
            # (the same build, same inputs: largest single difference 3e-2 of the scale in one run, 6e-2 in the next -- our atomics land
            # in a different order every run).  A bound on the single worst element is therefore not a property of either
            # implementation; the distribution is: all but 1e-4 of the elements inside 3e-3 of the scale, and the rms of the
            # difference (which a handful of outliers of the size of the scale itself would already break) inside 3e-3 as well.
            # The strict element-wise check of these three tensors at full size runs on the intended derivative (test_gpu_full.py).
            assert np.isfinite(got[k]).all(), k
            d = np.abs(got[k].astype(np.float64) - b)
            inside = d <= ATOL + 3e-3 * scale + 1e-3 * np.abs(b)
            rms = float(np.sqrt(np.mean(d * d)))
            print(f"{k}: executed-mode difference / scale: rms {rms / scale:.2e}, max {float(d.max()) / scale:.2e}, outside the band {1.0 - float(inside.mean()):.2e}")
            assert inside.mean() >= 1.0 - 1e-4, (k, float(inside.mean()), float(d.max()), scale)
            assert rms <= 3e-3 * scale, (k, rms, float(d.max()), scale)
    finally:
        ref.set_exp("libm")
        ref.set_num_threads(1)
