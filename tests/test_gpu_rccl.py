"""The RCCL leg of the view-parallel path on ONE GPU ("ship the launcher + a world_size=1 RCCL smoke test").

gpurun boxes have a single GPU and RCCL refuses two ranks on one device, so the scaling curve is the driver's to measure; what
can be exercised here is every collective of the exchange inside a real `nccl` (= RCCL) process group of one rank, launched exactly
as the driver launches N ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ...`):
  * tests/rccl_worker.py: factored exchange, plain bucket, packed all-reduce and densification statistics -- results equal to the
    non-distributed backward;
  * bench.py itself under the launcher with --force-allreduce, both --exchange modes: rc 0 and a well-formed JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(script_args, timeout=600, nproc=1):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_exchange_collectives_run_through_rccl_and_match_the_plain_backward(tmp_path):
    out = tmp_path / "rccl.json"
    p = _launch([os.path.join(ROOT, "tests", "rccl_worker.py"), str(out)])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.load(open(out))
    assert res["ok"], res
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["early_started"]          # the backward called drgb_ready() and the all-gather was queued under its last kernel
    for mode in ("factored", "factored_early", "bucket", "packed"):
        for k, v in res[mode].items():
            # two backward runs differ by the order of their float atomics, the factored SH rebuild by summation order (the worker
            # selects the intended derivative, so no slip-term noise rides on the geometry gradients)
            assert v <= (2e-5 if k in ("dL_dsh", "dL_dopacity") else 1e-3), (mode, k, v)
    assert res["stats_ok"]
    assert res["chunks_started"]         # ... and grads_ready(): the first half of the 44-B rows left under the second per-Gaussian launch
    assert res["folded_stats_ok"]        # densification statistics ride in the same bucket; radii take the maximum


@pytest.mark.parametrize("exchange", ["factored", "allreduce"])
def test_bench_under_the_launcher_with_the_exchange_forced(exchange):
    p = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--force-allreduce",
                 "--exchange", exchange, "--points", "200000"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["roofline"]["frac"] > 0
    # what the exchange costs and how much of it the step waits for is on the line whenever an exchange runs (the driver's scaling runs)
    assert d["exchange_ms"] > 0 and 0 < d["exchange_exposed_ms"] <= d["exchange_ms"] + 1e-3, (d.get("exchange_ms"), d.get("exchange_exposed_ms"))


def test_two_ranks_share_one_gpu_through_gloo(tmp_path):
    """Two REAL ranks on the one GPU of the box (process group gloo: RCCL refuses two ranks per device): each renders its own view with
    the HIP kernels, the factored exchange and the plain bucket run across the two processes, and the HIP rebuild kernel sums both
    ranks' dL/dRGB rows -- against the batch mean every rank computes locally from both views (tests/gloo_gpu_worker.py)."""
    out = tmp_path / "gloo"
    p = _launch([os.path.join(ROOT, "tests", "gloo_gpu_worker.py"), str(out)], nproc=2)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    for rank in (0, 1):
        res = json.load(open(f"{out}.{rank}"))
        if not res["ok"] and "gloo" in res.get("error", "").lower() and "cuda" in res.get("error", "").lower():
            pytest.skip("this torch build's gloo cannot move device tensors: " + res["error"][:200])
        assert res["ok"], res
        assert res["backend"] == "gloo" and res["world"] == 2
        for mode in ("factored", "factored_early", "bucket"):
            for k, v in res[mode].items():
                assert v <= (2e-5 if k in ("dL_dsh", "dL_dopacity") else 1e-3), (rank, mode, k, v)
        assert res["chunks_started"] and res["folded_stats_ok"], res    # two-launch per-Gaussian backward + statistics in the bucket, across two real ranks
