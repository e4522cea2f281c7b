"""Worker of tests/test_gpu_rccl.py::test_two_ranks_share_one_gpu_through_gloo: TWO ranks under `python -m torch.distributed.run
--nproc-per-node 2`, both on cuda:0, process group `gloo` (RCCL refuses two ranks on one device; gloo moves device tensors through
the host).  It is the only way a one-GPU box can run the view-parallel step with more than one REAL rank end to end: each rank renders
its own view with the HIP kernels, the factored exchange all-gathers the dL/dRGB rows and camera positions and all-reduces the small
bucket across the two processes, and the HIP rebuild kernel (sh_grad_from_views) sums two ranks' rows.  Every rank also computes both
views' plain backward locally; the exchanged mean must equal that.  Writes a JSON verdict per rank to argv[1] + ".<rank>"."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    import diff_gaussian_rasterization._C as C
    from synth_scene import jittered_view, make_scene, to_device, upstream_grads
    from test_gpu_view_parallel import _backward
    from view_parallel import FactoredGradExchange, GradBucket
    res = {"backend": dist.get_backend(), "world": world, "rank": rank}
    base = make_scene(20000, 256, 192, sh_degree=3, mu_px=2.5, seed=91, kernel_size=0.0, require_coord=False, require_depth=True)
    views = [to_device(base if r == 0 else jittered_view(base, r), dev) for r in range(world)]
    g = {k: v.to(dev) for k, v in upstream_grads(base, 91).items()}
    e = torch.Tensor([])
    P, M = base.means3D.shape[0], base.shs.shape[1]

    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))

    try:
        C.OPACITY_GRAD_INTENDED = True          # the exchange is what is tested: no slip-term order noise on the geometry gradients
        C.set_grad_allocator(dev, None)
        want = None
        for v in views:                          # the batch mean, computed locally from both views' plain backward
            bw = _backward(C, v, g, e)
            cur = dict(dL_dmeans3D=bw[3], dL_dsh=bw[5], dL_dopacity=bw[2], dL_dscales=bw[6], dL_drotations=bw[7])
            want = {k: t.clone() for k, t in cur.items()} if want is None else {k: want[k] + cur[k] for k in want}
        want = {k: t / world for k, t in want.items()}
        mine = views[rank]
        for early in (False, True):
            ex = FactoredGradExchange(P, M, base.sh_degree, dev)
            C.set_grad_allocator(dev, ex.allocator)
            if early:
                ex.set_view(mine.campos)
            _backward(C, mine, g, e)
            if early:      # the two-launch per-Gaussian backward: the first half's all-reduce was started from its hook
                res["chunks_started"] = len(ex._chunk_handles) == 4 and 0 < ex._rows_done < P
            got = ex.exchange(mine.means3D, mine.campos, average=True, stats=(torch.full((P,), float(rank + 1), device=dev),) * 3,
                              radii=torch.full((P,), rank + 5, device=dev, dtype=torch.int32))
            torch.cuda.synchronize(dev)
            res["factored_early" if early else "factored"] = {k: rel(got[k], want[k]) for k in want}
            res["folded_stats_ok"] = bool((got["densify_stats"] == float(sum(range(1, world + 1)))).all() and (got["radii_max"] == world + 4).all())
        bk = GradBucket(P, M, dev)
        C.set_grad_allocator(dev, bk.allocator)
        _backward(C, mine, g, e)
        got = bk.allreduce(average=True)
        torch.cuda.synchronize(dev)
        res["bucket"] = {k: rel(got[k], want[k]) for k in want}
        res["ok"] = True
    except Exception as ex_:  # noqa: BLE001
        res["ok"] = False
        res["error"] = repr(ex_)
    finally:
        C.set_grad_allocator(dev, None)
    json.dump(res, open(sys.argv[1] + f".{rank}", "w"))
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()
