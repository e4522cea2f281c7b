"""Worker of tests/test_gpu_rccl.py: runs under `python -m torch.distributed.run --nproc-per-node 1`, i.e. inside a REAL RCCL
process group of one rank (RCCL refuses two ranks on one device, and gpurun boxes have one GPU).  Every collective of the
view-parallel exchange -- all_gather_into_tensor of the dL/dRGB rows and camera positions, the all-reduce of the small bucket,
the all-reduce of the plain 236-B bucket, the densification statistics -- executes through RCCL, and the exchanged gradients must
equal the plain, non-distributed backward of the same view.  Writes a JSON verdict to argv[1]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rade-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path = sys.argv[1]
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
    import diff_gaussian_rasterization._C as C
    from synth_scene import make_scene, to_device, upstream_grads
    from test_gpu_view_parallel import _backward
    from view_parallel import FactoredGradExchange, GradBucket, allreduce_densification_stats, allreduce_gradients
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    base = make_scene(30000, 320, 240, sh_degree=3, mu_px=2.5, seed=77, kernel_size=0.0, require_coord=False, require_depth=True)
    s = to_device(base, dev)
    g = {k: v.to(dev) for k, v in upstream_grads(base, 77).items()}
    e = torch.Tensor([])
    P, M = base.means3D.shape[0], base.shs.shape[1]

    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))

    try:
        # the intended derivative: run-to-run differences are then plain summation order (the executed default adds the slip term's
        # heavy-tailed order noise to the geometry gradients, conftest._gradient_mode -- this test is about the exchange)
        C.OPACITY_GRAD_INTENDED = True
        C.set_grad_allocator(dev, None)
        plain = _backward(C, s, g, e)           # (means2D, colors, opacity, means3D, cov3D, sh, scales, rotations)
        want = dict(dL_dmeans3D=plain[3], dL_dsh=plain[5], dL_dopacity=plain[2], dL_dscales=plain[6], dL_drotations=plain[7])
        # 1. factored exchange: all-gather (dL/dRGB, campos) + all-reduce (44 B/Gaussian) through RCCL
        ex = FactoredGradExchange(P, M, base.sh_degree, dev)
        C.set_grad_allocator(dev, ex.allocator)
        _backward(C, s, g, e)
        got = ex.exchange(s.means3D, s.campos, average=True)
        torch.cuda.synchronize(dev)
        res["factored"] = {k: rel(got[k], want[k]) for k in want}
        # 1b. the same with the all-gather started early, from the backward's drgb_ready hook, on the side stream
        ex.set_view(s.campos)
        assert ex.early_drgb
        _backward(C, s, g, e)
        res["early_started"] = ex._early is not None
        # ... and the per-Gaussian backward queued in two launches: the first half's 44-B rows are on their way under the second launch
        res["chunks_started"] = len(ex._chunk_handles) == 4 and 0 < ex._rows_done < P
        st = (plain[0][:, :2].norm(dim=1), plain[0][:, 2].abs(), (plain[0][:, 2] != 0).float())
        rr = torch.arange(P, device=dev, dtype=torch.int32)
        got = ex.exchange(s.means3D, s.campos, average=True, stats=st, radii=rr)
        torch.cuda.synchronize(dev)
        res["factored_early"] = {k: rel(got[k], want[k]) for k in want}
        res["folded_stats_ok"] = bool(torch.equal(got["densify_stats"][:, 0], st[0]) and torch.equal(got["densify_stats"][:, 2], st[2])
                                      and torch.equal(got["radii_max"], rr))
        # 2. plain bucket: one all-reduce of 236 B/Gaussian written in place by the backward
        bk = GradBucket(P, M, dev)
        C.set_grad_allocator(dev, bk.allocator)
        _backward(C, s, g, e)
        got = bk.allreduce(average=True)
        torch.cuda.synchronize(dev)
        res["bucket"] = {k: rel(got[k], want[k]) for k in want}
        # 3. the packing all-reduce and the densification statistics with the collective forced at world size 1
        C.set_grad_allocator(dev, None)
        got = allreduce_gradients(dict(want), average=True, force=True)
        res["packed"] = {k: rel(got[k], want[k]) for k in want}
        a, b, c, r = allreduce_densification_stats(plain[0][:, :2].norm(dim=1), plain[0][:, 2], (plain[0][:, 2] != 0), torch.arange(P, device=dev, dtype=torch.int32))
        res["stats_ok"] = bool(torch.equal(r, torch.arange(P, device=dev, dtype=torch.int32)))
        res["ok"] = True
    except Exception as ex_:  # noqa: BLE001
        res["ok"] = False
        res["error"] = repr(ex_)
    finally:
        C.set_grad_allocator(dev, None)
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main()
