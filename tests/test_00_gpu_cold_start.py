"""The FIRST GPU test of the session (the file name sorts first): a fresh subprocess whose very first launch is the forward of the
golden scene g_C1, poisoned buffers, no warm-up anywhere -- tests/cold_first_launch.py has the story.  Nothing in
this pytest process has touched the GPU before this test runs (there is no warm-up fixture any more), so on a fresh box the
subprocess is also the box's first GPU process: the one condition under which round 3 saw a mismatch."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cold(case, *extra):
    r = subprocess.run([sys.executable, os.path.join(HERE, "cold_first_launch.py"), case, "--tag", "pytest", *extra],
                       cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert r.returncode == 0 and rec["ok"], json.dumps(rec, indent=1)[:6000]
    assert rec["first_launch_of_process"] and rec["poison"]
    return rec


@pytest.mark.gpu
def test_first_launch_of_a_fresh_process_matches_golden():
    _cold("g_C1")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["g_depth", "g_C2s"])
def test_first_launch_other_modes(case):
    """the same for a tiny depth-mode scene and the render.py-mode slice (coord + depth maps)"""
    _cold(case)
