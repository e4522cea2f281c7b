"""Full-size checks on the MI355X (-m gpu): the BASELINE.json workloads themselves.

* C2 (the benchmark workload, 1M Gaussians @1080p) and C3 (1M @1600x1200, the per-GPU view of the 8-view
  data-parallel config): complete parity against the oracle -- exact indices, images within tolerance,
  gradients by the strict-fraction criterion AND the fp64 arbiter (the HIP gradients must be as close to
  the float64 oracle as the float32 oracle is) -- plus the size-independent properties below.
* C4 / C5 (coord-map mode, 5M Gaussians at 1080p; 4K heavy overdraw, 500k) at a reduced Gaussian count the
  oracle finishes in seconds, same checks -- and at their NAMED size (the oracle takes 16 s / 43 s on the box's
  256 host cores; log of the round's run: profiles/r02_full_size_parity.log).
Properties that need no oracle (also run at full C4/C5 size):
  - point_list is a permutation-with-repetition consistent with `ranges` (every tile range is sorted by
    depth key, ties by index; ranges partition [0, R));
  - forward is deterministic (bitwise identical twice);
  - alpha in [0, 1], color - T*bg >= 0, n_contrib <= tile list length;
  - linearity of the backward in the cotangents: grad(2*w) == 2*grad(w) within fp32 noise.
"""
import math
import os

import numpy as np
import pytest
import torch

from synth_scene import CONFIGS, make_config, upstream_grads
from util import ATOL, close, frac_close, oracle_backward, oracle_for

pytestmark = pytest.mark.gpu


def _forward_checks(s, o=None):
    from gpu_util import HipRun
    h = HipRun(s, "cuda:0")
    st = h.forward_native()
    torch.cuda.synchronize()
    R, P, H, W = st[0], h.P, s.H, s.W
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    pl = h.export("point_list", torch.int32, R).view(np.uint32)
    rg = h.export("ranges", torch.int32, 2 * ntiles).view(np.uint32).reshape(-1, 2)
    dk = h.export("depth_key", torch.int32, P).view(np.uint32)
    radii = st[8].cpu().numpy()
    # ranges partition [0, R) in tile order
    nz = rg[rg[:, 1] > rg[:, 0]]
    assert nz[0, 0] == 0 and nz[-1, 1] == R and np.array_equal(nz[1:, 0], nz[:-1, 1])
    # inside every tile: ascending (depth key, index); only visible Gaussians appear
    assert (radii[pl] > 0).all()
    keys = (dk[pl].astype(np.uint64) << np.uint64(32)) | pl.astype(np.uint64)
    tile_of = np.repeat(np.arange(ntiles), (rg[:, 1] - rg[:, 0]).astype(np.int64))
    assert tile_of.shape[0] == R
    same_tile = tile_of[1:] == tile_of[:-1]
    assert (keys[1:][same_tile] > keys[:-1][same_tile]).all()
    alpha = st[4].cpu().numpy()
    assert alpha.min() >= 0 and alpha.max() <= 1.0 + 1e-5
    nc = h.export("n_contrib", torch.int32, 2 * H * W).view(np.uint32)[: H * W].reshape(H, W)
    tl = (rg[:, 1] - rg[:, 0]).reshape((H + 15) // 16, (W + 15) // 16)
    assert (nc <= np.kron(tl, np.ones((16, 16), np.uint32))[:H, :W]).all()
    st2 = HipRun(s, "cuda:0").forward_native()
    for x, y in zip(st[1:9], st2[1:9]):
        assert torch.equal(x, y)
    if o is not None:
        ref = o.outputs()
        assert R == o.num_rendered
        assert np.array_equal(radii, ref[1])
        assert np.array_equal(pl, o.get("point_list"))
        assert np.array_equal(rg.reshape(-1), o.get("ranges"))
        assert np.array_equal(h.export("n_contrib", torch.int32, 2 * H * W).view(np.uint32), o.get("n_contrib"))
        got = [st[1], None, st[2], st[3], st[6], st[7], st[4], st[5]]
        for k in (0, 2, 3, 4, 5, 6, 7):
            a, b = got[k].cpu().numpy(), ref[k]
            assert close(a, b).all(), (k, float(np.abs(a - b).max()))
        # the headline metric's quality term: depth L1 vs the reference restatement
        return float(np.abs(st[6].cpu().numpy() - ref[4]).mean())
    return None


def _backward_checks(s, o, seed, arbiter=False):
    from gpu_util import HipRun
    g = upstream_grads(s, seed)
    h = HipRun(s, "cuda:0")
    h.forward()
    got = h.backward(g)
    if o is not None:
        ref = oracle_backward(o, g)
        for k, b in ref.items():
            if got[k] is None:
                continue
            a, b = got[k], b.reshape(got[k].shape)
            assert not np.isnan(a).any()
            # >= 99 % inside 1e-5 / 1e-4, all inside the scale band
            assert frac_close(a, b) > 0.99, (k, frac_close(a, b))
            scale = float(np.abs(b).max())
            assert close(a, b, atol=ATOL + 1e-4 * scale, rtol=1e-3).all(), (k, float(np.abs(a - b).max()), scale)
        if arbiter:
            # fp64 arbiter at full size: the same formulas in float64 on all host cores.  The fp32 oracle's distance to it is the
            # accuracy fp32 arithmetic allows on this scene; the HIP gradients must be as close (rms within 1.1x).  A float64 run
            # takes a different thresholded decision at ~0.1 % of the pixels (alpha at 1/255, T at 1e-4 or 0.5 -- C2: 0.12 %); those
            # move single elements of BOTH fp32 results alike, so the max criterion (1.25x) is only applied when there are none.
            o64 = oracle_for(s, precision=64)
            o64.forward()
            nc32, nc64 = o.get("n_contrib"), o64.get("n_contrib")
            differing = float((nc32 != nc64).mean())
            assert differing < 1e-2, differing
            g64 = oracle_backward(o64, g)
            for k, b in ref.items():
                if got[k] is None:
                    continue
                a, b, c = got[k].astype(np.float64), b.reshape(got[k].shape).astype(np.float64), g64[k].reshape(got[k].shape)
                e_hip, e_ref = np.abs(a - c), np.abs(b - c)
                rms_hip, rms_ref = np.sqrt((e_hip ** 2).mean()), np.sqrt((e_ref ** 2).mean())
                assert rms_hip <= 1.1 * rms_ref + 1e-7, (k, rms_hip, rms_ref)
                if differing == 0.0:
                    assert e_hip.max() <= 1.25 * e_ref.max() + ATOL, (k, e_hip.max(), e_ref.max())
    # linearity in the cotangents
    g2 = {k: 2 * v for k, v in g.items()}
    h2 = HipRun(s, "cuda:0")
    h2.forward()
    got2 = h2.backward(g2)
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
        a, b = got2[k], 2 * got[k]
        scale = float(np.abs(b).max())
        assert close(a, b, atol=ATOL + 1e-4 * scale, rtol=1e-3).all(), k


def test_C2_full_parity_and_depth_L1():
    s = make_config("C2")
    o = oracle_for(s)
    o.forward()
    l1 = _forward_checks(s, o)
    assert l1 < 1e-5, l1  # depth L1 vs ref (BASELINE.json metric's quality term)
    _backward_checks(s, o, CONFIGS["C2"]["seed"], arbiter=True)


def test_C3_full_parity():
    """BASELINE.json configs[2]: 1M Gaussians at 1600x1200 (DTU shape; 100x75 tiles) -- what every GPU of the 8-view
    data-parallel configuration renders per step."""
    s = make_config("C3")
    o = oracle_for(s)
    o.forward()
    l1 = _forward_checks(s, o)
    assert l1 < 1e-5, l1
    _backward_checks(s, o, CONFIGS["C3"]["seed"], arbiter=True)


# Opt-in (RADEGS_FULL_ORACLE=1): 140 s of hand-written oracle (fp32 + fp64) on the box's host cores.  The default suite already checks C4
# and C5 at their named size against the REFERENCE'S OWN code compiled for the host (tests/test_gpu_vs_compiled_reference.py), to which
# the oracle is pinned bit for bit (tests/test_ref_parity.py); this adds the fp64 arbiter at those sizes.  The round's run:
# profiles/r05_full_size_oracle.log.
@pytest.mark.skipif(os.environ.get("RADEGS_FULL_ORACLE", "0") != "1", reason="opt-in: RADEGS_FULL_ORACLE=1 (140 s of CPU oracle; log of the round's run in profiles/)")
@pytest.mark.parametrize("name", ["C4", "C5"])
def test_full_size_oracle_parity(name):
    """C4 (5M Gaussians, 1080p, coord-map mode) and C5 (500k, 3840x2160, heavy overdraw) against the oracle at the NAMED size:
    exact indices, all maps, all gradients, fp64 arbiter."""
    s = make_config(name)
    o = oracle_for(s)
    o.forward()
    _forward_checks(s, o)
    _backward_checks(s, o, CONFIGS[name]["seed"], arbiter=True)   # the fp64 arbiter wherever a gradient is checked


def _arbiter_checks(s, seed):
    """The gradients with the float64 oracle as arbiter and the reference's own arithmetic -- its compiled backward at two thread counts,
    the fp32 oracle -- as the yardstick (tests/arbiter.py, criteria A-D): as accurate as the reference's code, at every size."""
    import arbiter
    # (the fp32 oracle alone as the yardstick: it is bit-identical to the compiled reference in the one-thread schedule, and the compiled
    # reference's fiber emulation needs minutes on a 4K scene -- the randomised sweep and scripts/gpu_arbiter_table.py run both)
    info, rows = arbiter.evaluate(s, upstream_grads(s, seed), use_compiled_reference=False)
    bad = {k: arbiter.failed_criteria(v, worst_element=info["same_decisions"]) for k, v in rows.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, {k: [(c, float(m), float(a)) for c, m, a in v] for k, v in bad.items()}
    return info


def test_C4_shape_coord_map_reduced():
    """C4's shape (coord-map mode, 1080p) at 300 k Gaussians: complete parity plus the arbiter (entry streams with the 32-float record)."""
    s = make_config("C4", P=300_000)
    o = oracle_for(s)
    o.forward()
    _forward_checks(s, o)
    _backward_checks(s, o, 4)
    assert _arbiter_checks(s, 4)["streams"] is True


def test_C5_shape_4k_heavy_overdraw_reduced():
    """C5's shape (4K, 100-tile splats) at 60 k Gaussians: complete parity plus the arbiter on the TILE-WIDE kernels, whose association
    inside a tile is the one thing the reference's own order noise does not contain (the K x self-noise band of
    tests/test_gpu_vs_compiled_reference.py needs its absolute backstop there; this is the statement that replaces a chosen K)."""
    s = make_config("C5", P=60_000)
    o = oracle_for(s)
    o.forward()
    _forward_checks(s, o)
    _backward_checks(s, o, 5)
    assert _arbiter_checks(s, 5)["streams"] is False


@pytest.mark.parametrize("name", ["C4", "C5"])
def test_full_size_properties(name):
    s = make_config(name)
    _forward_checks(s, None)
    _backward_checks(s, None, CONFIGS[name]["seed"])


def test_scene_of_more_than_2p24_gaussians_runs_through_the_entry_streams():
    """17 M Gaussians (R ~ 45 M instances of small splats): more than the 24 bits a Gaussian index has next to the block mask in an
    instance value, and with the worst-case reservation of rounds 2-4 more stream storage than RADEGS_STREAMS_MAX_MB allows -- both
    used to switch such a scene silently to the slower tile-wide kernels.  Now the mask rides in the 32-bit tile
    key and the lists are stored at their real size: the automatic choice is the entry streams, the first call (no usage history) may
    still go tile-wide, the second runs the streams inside the default budget -- and both formulations return the same image (exact
    contributor counts, maps within 1e-5 / 1e-4).  No oracle at this size: the two formulations check each other."""
    import diff_gaussian_rasterization._C as C
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    P, W, H = 17_000_000, 1920, 1080
    assert P > (1 << 24)
    g = torch.Generator(device=dev).manual_seed(17)
    tanfovx = math.tan(math.radians(60.0) * 0.5)
    tanfovy = tanfovx * H / W
    focal = W / (2 * tanfovx)
    z = torch.rand(P, device=dev, generator=g) * 8 + 2
    x = z * tanfovx * (torch.rand(P, device=dev, generator=g) * 2.2 - 1.1)
    y = z * tanfovy * (torch.rand(P, device=dev, generator=g) * 2.2 - 1.1)
    means = torch.stack([x, y, z], 1).contiguous()
    sigma_px = torch.exp(math.log(0.8) + 0.5 * torch.randn(P, device=dev, generator=g))
    scales = ((z * sigma_px / focal)[:, None] * torch.exp(0.3 * torch.randn(P, 3, device=dev, generator=g))).contiguous()
    rot = torch.nn.functional.normalize(torch.randn(P, 4, device=dev, generator=g)).contiguous()
    opac = torch.sigmoid(torch.randn(P, 1, device=dev, generator=g)).contiguous()
    shs = torch.randn(P, 1, 3, device=dev, generator=g).contiguous()
    eye = torch.eye(4, device=dev)
    from synth_scene import projection_matrix
    proj = torch.from_numpy(projection_matrix(0.01, 100.0, 2 * math.atan(tanfovx), 2 * math.atan(tanfovy)).T.astype(np.float32)).to(dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, kernel_size=0.0, bg=torch.zeros(3, device=dev),
                                       scale_modifier=1.0, viewmatrix=eye, projmatrix=proj.contiguous(), sh_degree=0, campos=torch.zeros(3, device=dev),
                                       prefiltered=False, require_depth=True, require_coord=False, debug=False)

    def render():
        with torch.no_grad():
            out = GaussianRasterizer(rs)(means, torch.zeros_like(means), opac, shs=shs, scales=scales, rotations=rot)
        torch.cuda.synchronize()
        return out, C.last_forward_used_streams()

    try:
        os.environ["RADEGS_STREAMS"] = "0"
        C.reload_env()
        ref, used = render()
        assert used is False
        os.environ.pop("RADEGS_STREAMS")
        C.reload_env()
        render()                       # first automatic call: may run without a usage history
        out, used = render()           # from the second call on the lists are sized from the previous view
        assert used is True, "a small-splat scene of 17 M Gaussians must run the entry streams by itself"
    finally:
        os.environ.pop("RADEGS_STREAMS", None)
        C.reload_env()
    assert torch.equal(out[1], ref[1])                                   # radii
    assert int((ref[1] > 0).sum()) > 10_000_000
    for k in (0, 4, 5, 6, 7):
        a, b = out[k].cpu().numpy(), ref[k].cpu().numpy()
        assert close(a, b).all(), (k, float(np.abs(a - b).max()))
