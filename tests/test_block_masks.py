"""The block lists of the sub-tile entry streams may only drop a (tile entry, 8x4 block) pair when NO pixel of the block can pass
the forward blend loop's test.  `ellipse_block_mask` (csrc/rg_blend.h, host+device source) is compiled for the CPU and compared
with the brute-force truth -- the kernels' own per-pixel rule, evaluated at all 32 pixel centres of every block -- over random
conics: round, elongated, needle-thin, nearly degenerate, far from / straddling the tile, opacities from 1/255 to 1."""
import numpy as np
import pytest

from hostcheck import hostcheck as hc


def _conics(n, rng, sig_lo, sig_hi, aspect_hi, filt):
    s1 = np.exp(rng.uniform(np.log(sig_lo), np.log(sig_hi), n))
    s2 = s1 / np.exp(rng.uniform(0.0, np.log(aspect_hi), n))
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    a = c * c * s1 * s1 + s * s * s2 * s2 + filt       # covariance (+ the reference's low-pass term on the diagonal)
    b = c * s * (s1 * s1 - s2 * s2)
    d = s * s * s1 * s1 + c * c * s2 * s2 + filt
    det = a * d - b * b
    return (d / det).astype(np.float32), (-b / det).astype(np.float32), (a / det).astype(np.float32)   # conic = inverse


@pytest.mark.parametrize("name,sig_lo,sig_hi,aspect_hi,filt", [("small", 0.3, 6.0, 4.0, 0.3), ("large", 4.0, 300.0, 8.0, 0.3),
                                                                ("needles", 0.5, 200.0, 3000.0, 0.0), ("mixed", 0.2, 60.0, 50.0, 0.1)])
def test_block_mask_never_drops_a_reachable_block(name, sig_lo, sig_hi, aspect_hi, filt):
    rng = np.random.default_rng({"small": 1, "large": 2, "needles": 3, "mixed": 4}[name])
    n = 120000
    cx, cy, cz = _conics(n, rng, sig_lo, sig_hi, aspect_hi, filt)
    tx0, ty0 = 16.0 * rng.integers(0, 240), 16.0 * rng.integers(0, 135)
    reach = 3.0 * np.sqrt(np.maximum(1.0 / np.minimum(cx, cz), 1.0))     # means up to ~3 sigma-ish outside the tile as well as inside
    mx = (tx0 + rng.uniform(-1.0, 1.0, n) * (8 + reach) + 7.5).astype(np.float32)
    my = (ty0 + rng.uniform(-1.0, 1.0, n) * (8 + reach) + 7.5).astype(np.float32)
    op = np.exp(rng.uniform(np.log(1.0 / 300.0), 0.0, n)).astype(np.float32)
    mask, truth = hc.block_masks(np.stack([mx, my, cx, cy, cz, op], 1), tx0, ty0)
    missed = truth & ~mask
    assert not missed.any(), (name, int((missed != 0).sum()), np.stack([mx, my, cx, cy, cz, op], 1)[missed != 0][:5])
    kept, needed = int(sum(bin(int(m)).count("1") for m in mask)), int(sum(bin(int(t)).count("1") for t in truth))
    print(f"{name}: blocks kept {kept}, reachable {needed}, ratio {kept / max(needed, 1):.3f}, all-8 fallbacks {(mask == 255).mean():.3f}")
    if name in ("small", "large"):
        assert kept <= 1.25 * needed + 100    # and it still culls: at most 25 % above the exact answer on ordinary splats


def test_block_mask_edge_cases():
    rec = np.array([[8, 8, 1, 0, 1, 1.0 / 400.0],            # opacity below 1/255: nothing can blend
                    [8, 8, np.nan, 0, 1, 0.5],                # NaN conic: keep everything (the exact rule decides)
                    [8, 8, 1, 2, 1, 0.5],                     # indefinite conic: keep everything
                    [8, 8, 0.0, 0, 1, 0.5],                   # degenerate: keep everything
                    [-500, -500, 1, 0, 1, 0.9],               # far away small splat: nothing
                    [3.2, 1.7, 4, 0, 4, 0.9]], np.float32)    # tiny splat inside block 0 only
    mask, truth = hc.block_masks(rec, 0.0, 0.0)
    assert list(mask) == [0, 255, 255, 255, 0, 1] and int(truth[5]) == 1 and int(truth[0]) == 0 and int(truth[4]) == 0


@pytest.mark.parametrize("sig_lo,sig_hi,aspect_hi", [(0.5, 8.0, 6.0), (3.0, 40.0, 30.0)])
def test_block_masks_as_the_emission_asks(sig_lo, sig_hi, aspect_hi):
    """One set-up per splat for its whole 3x4-tile rectangle (slack from the rectangle's extent), slab extents shared by the tiles
    of a tile row -- the call pattern of emit_instances_kernel<true>."""
    rng = np.random.default_rng(7)
    n = 15000
    cx, cy, cz = _conics(n, rng, sig_lo, sig_hi, aspect_hi, 0.3)
    tx0, ty0, tx1, ty1 = 50, 20, 53, 24
    mx = rng.uniform(tx0 * 16 - 20, tx1 * 16 + 20, n).astype(np.float32)
    my = rng.uniform(ty0 * 16 - 20, ty1 * 16 + 20, n).astype(np.float32)
    op = np.exp(rng.uniform(np.log(1.0 / 300.0), 0.0, n)).astype(np.float32)
    mask, truth = hc.block_masks_rect(np.stack([mx, my, cx, cy, cz, op], 1), tx0, ty0, tx1, ty1)
    assert not (truth & ~mask).any()
    kept = int(np.unpackbits(mask.astype(np.uint8)).sum()), int(np.unpackbits(truth.astype(np.uint8)).sum())
    print("rect pattern: blocks kept %d, reachable %d" % kept)
    assert kept[0] <= 1.3 * kept[1] + 100
