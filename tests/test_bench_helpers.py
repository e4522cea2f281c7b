"""bench.py's host-side bookkeeping (CPU tier): the algorithmic-byte model, the committed PMC summaries it reads, the stage mapping of
`stage_bytes_moved`.  No GPU, nothing is launched."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_the_survey_formula_for_C2():
    b = _bench()
    P, Pv, R, N = 1_000_000, 818_447, 3_914_644, 1920 * 1080
    ab = b.algorithmic_bytes(P, Pv, R, N, 3, 0, 1)
    # the dominant kernel's 447 MB and the path's 2.42 GB per view
    assert abs(ab["blend_bwd"] - 447.2e6) < 0.5e6
    assert abs(ab["total"] - 2.4167e9) < 2e6
    assert ab["total"] == sum(v for k, v in ab.items() if k != "total")


def test_committed_pmc_summaries_feed_the_bench_line():
    b = _bench()
    grouped = {"preprocess_fwd": 0.094, "binning": 0.243, "blend_fwd": 0.37, "blend_bwd": 0.575, "preprocess_bwd": 0.182}
    moved = b.stage_bytes_moved("C2", "C2", 1_000_000, 1920, 1080, grouped)
    import glob
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_per_kernel.json")))[-1]
    assert moved is not None and moved["source"] == os.path.basename(newest)          # the newest round's pass of this workload
    assert set(moved["bytes"]) == set(grouped)                       # every kernel of the pass found its stage
    assert 0.5e9 < moved["bytes"]["blend_bwd"] < 1.5e9 and 0.05 < moved["frac_of_hbm_peak"]["preprocess_bwd"] < 1.0
    traffic, note = b.pmc_traffic("blend_bwd_", "C2", "C2", 1_000_000, 1920, 1080)
    assert traffic and "TCC_EA0" in note
    # a workload without a PMC pass reports None instead of another workload's counters
    assert b.stage_bytes_moved("C2", "C2", 123, 1920, 1080, grouped) is None
    d = json.load(open(newest))
    assert all("void rg::" != k.strip() for k in d)                  # the sort's template kernels keep their names apart
    assert any("block_counts_kernel" in k for k in d) and any("blend_bwd_streams_kernel" in k for k in d)   # this round's kernels


def test_binning_bytes_moved_counts_the_key_width():
    b = _bench()
    small = b.binning_bytes_moved(1_000_000, 3_900_000, 8160, True)       # 1080p: 16-bit tile keys
    wide = b.binning_bytes_moved(1_000_000, 3_900_000, 70000, True) - 8 * (70000 - 8160)   # a grid above 65 536 tiles keeps 32-bit keys
    assert wide - small == (2 + 2 * 6 + 2) * 3_900_000               # emission 2 B, two sort passes x 6 B, ranges 2 B per instance
