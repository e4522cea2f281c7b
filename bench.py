#!/usr/bin/env python3
"""Benchmark of the hot path: fwd+bwd Msplats/s of the differentiable splat rasterizer.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic view: `_C.rasterize_gaussians` followed by
`_C.rasterize_gaussians_backward` with fixed random cotangents (allocation of outputs/grads included,
loss arithmetic excluded -- SURVEY.md 8d).  Workload at N=1: BASELINE.json configs[1] ("C2": 1M
Gaussians, 1920x1080, SH degree 3, RGB + depth + normal); `--config C3|C4|C5` measures the other
BASELINE configs the same way.  Like training (train.py:118 picks another camera every iteration) the
steps walk a STREAM OF DIFFERENT VIEWS of the same Gaussians: the config's own view, `--views`-2
neighbouring ones (2 deg / 0.05 units apart) and one dolly-in view whose num_rendered is well above the
others' (what the speculative binning's capacity prediction has to survive; `speculation.miss_rate` says how
often it did not).  For N>1 every rank walks its OWN views of the replicated Gaussians (weak scaling) and the
step ends with the RCCL exchange of the parameter gradients.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel: algorithmic bytes per launch / its average
                launch duration, measured with HIP events recorded by the library on the launch stream
                during the timed region
  cpu_baseline  the CPU oracle (oracle/, a port of the reference algorithm) timed on this box's host
                cores on the same workload -- a reported baseline, not the target
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(P, Pv, R, N, D, c, d):
    """Per-stage ALGORITHMIC bytes of one view (SURVEY.md 8d split by stage)."""
    n = 1 if (c or d) else 0
    K = (D + 1) ** 2
    G = 36 + 12 * d + 36 * c + 12 * n
    k = 6  # ceil((32 + tile bits) / 8) for every BASELINE config
    A = 44 + 36 * c + 12 * d + 12 * n
    st = {
        "preprocess_fwd": 52 * P + Pv * (12 * K + 127),
        "binning": 16 * P + R * (12 + 24 * k + 8),
        "blend_fwd": R * (4 + G) + N * (24 + 36 * c + 12 * d + 16 * n),
        "blend_bwd": R * (4 + G) + N * (28 + 36 * c + 12 * d + 28 * n) + Pv * A,
        "preprocess_bwd": Pv * (12 + 160 + (123 + 12 * K) + (40 + 12 * K)) + 284 * (P - Pv),
    }
    st["total"] = sum(st.values())
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", help="BASELINE.json config (C1..C5); the headline metric is quoted on C2")
    ap.add_argument("--points", type=int, default=0, help="override the number of Gaussians (debug only)")
    ap.add_argument("--mu-px", type=float, default=0.0, help="override the median splat size in pixels (debug only)")
    ap.add_argument("--views", type=int, default=9, help="distinct camera views the steps rotate through (1 = the same view every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-allreduce", action="store_true", help="run the RCCL gradient exchange even at world size 1 (path check)")
    ap.add_argument("--exchange", choices=("factored", "allreduce"), default="factored",
                    help="N>1 gradient exchange: 'factored' all-gathers the 12-B dL/dRGB rows and rebuilds the SH gradient locally "
                         "(161 B/Gaussian over xGMI at N=8), 'allreduce' sums the whole 236-B bucket (413 B/Gaussian)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run (any N)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import diff_gaussian_rasterization._C as C
    from synth_scene import CONFIGS, jittered_view, make_config, to_device, upstream_grads
    from view_parallel import FactoredGradExchange, GradBucket

    force_allreduce = args.force_allreduce and launched
    over = {}
    if args.points:
        over["P"] = args.points
    if args.mu_px:
        over["mu_px"] = args.mu_px
    cfg = dict(CONFIGS[args.config])
    scene_cpu = make_config(args.config, **over)
    # the view stream of this rank: view 0 of rank 0 is the config's own view; the last one is the dolly-in view
    nviews = max(1, args.views)
    views_cpu = []
    for k in range(nviews):
        vid = rank * 64 + k
        if vid == 0:
            views_cpu.append(scene_cpu)
        else:
            views_cpu.append(jittered_view(scene_cpu, vid, dolly=0.6 if (nviews > 2 and k == nviews - 1) else 0.0))
    s = to_device(scene_cpu, dev)
    cams = [tuple(t.to(dev) for t in (v.viewmatrix, v.projmatrix, v.campos)) for v in views_cpu]
    g = {k: v.to(dev) for k, v in upstream_grads(scene_cpu, cfg["seed"]).items()}
    P, W, H = s.means3D.shape[0], s.W, s.H
    e = torch.Tensor([])
    bucket = None
    if world > 1 or force_allreduce:  # the backward writes its gradients straight into the exchange buffers
        if args.exchange == "factored":
            bucket = FactoredGradExchange(P, s.shs.shape[1], s.sh_degree, dev)
        else:
            bucket = GradBucket(P, s.shs.shape[1], dev)
        C.set_grad_allocator(dev, bucket.allocator)
    counter = [0]

    def step():
        vm, pm, cp = cams[counter[0] % nviews]
        counter[0] += 1
        fw = C.rasterize_gaussians(s.bg, s.means3D, e, s.opacities, s.scales, s.rotations, 1.0, e, vm, pm, s.tanfovx,
                                   s.tanfovy, s.kernel_size, H, W, s.shs, s.sh_degree, cp, False, s.require_coord,
                                   s.require_depth, False)
        R, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geom, binning, img = fw
        bw = C.rasterize_gaussians_backward(s.bg, s.means3D, radii, e, s.scales, s.rotations, 1.0, e, vm, pm,
                                            s.tanfovx, s.tanfovy, s.kernel_size, g["color"], g["coord"], g["mcoord"], g["depth"],
                                            g["mdepth"], g["alpha"], g["normal"], normal, s.shs, s.sh_degree, cp, geom, R,
                                            binning, img, alpha, s.require_coord, s.require_depth, False)
        grads = dict(dL_dmeans3D=bw[3], dL_dsh=bw[5], dL_dopacity=bw[2], dL_dscales=bw[6], dL_drotations=bw[7])
        if bucket is not None:  # the one exchange step of the path (RCCL over xGMI)
            grads = bucket.exchange(s.means3D, cp, average=True) if args.exchange == "factored" else bucket.allreduce(average=True)
        return R, radii, grads

    def fence():
        if launched:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up: the per-stage breakdown (`stages_ms`) is measured HERE, with an event pair around every stage.  Each recorded
    # stage boundary costs ~10 us of stream bubble (10 stages = ~0.1 ms per step), so the timed steps below record events
    # around the dominant kernel only -- the one the roofline object reports, live, inside the timed region.
    if args.warmup > 1:  # first call: allocator growth, code-object load -- keep it out of the per-stage averages
        R, radii, _ = step()
        fence()
    C.profile_collect()
    C.profile_enable(True)
    for _ in range(args.warmup - 1 if args.warmup > 1 else args.warmup):
        R, radii, _ = step()
    fence()
    C.profile_enable(False)
    stages_warm = C.profile_collect()
    dom = None
    if args.warmup > 0:
        dom = max(("preprocess_fwd", "blend_fwd", "blend_bwd", "preprocess_bwd"), key=lambda k: stages_warm[k][0] / max(stages_warm[k][1], 1))
    # the dominant kernel is timed live on every 2nd step of the timed region (the event pair around it is a ~12 us stream bubble;
    # every 2nd step of a 9-view rotation still visits every view)
    C.profile_enable(True, only=dom, every=2 if (dom is not None and args.steps >= 8) else 1)   # no warm-up steps: every stage, in the timed region
    C.binning_stats(reset=True)
    Rs, vis = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        R, radii, _ = step()
        Rs.append(R)
        vis.append(radii)          # kept alive, counted after the timed region
    fence()
    elapsed = time.perf_counter() - t0
    C.profile_enable(False)
    spec_calls, spec_misses = C.binning_stats()
    timed = C.profile_collect()
    if dom is None:
        stages = timed
        dom = max(("preprocess_fwd", "blend_fwd", "blend_bwd", "preprocess_bwd"), key=lambda k: timed[k][0] / max(timed[k][1], 1))
    else:
        stages = dict(stages_warm)
        stages[dom] = timed[dom]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * P / 1e6 / (elapsed / args.steps)

    if rank == 0:
        # algorithmic bytes: per-step averages over the views the timed steps rendered
        Pv = int(round(sum(int((r > 0).sum().item()) for r in vis) / max(len(vis), 1)))
        R = int(round(sum(Rs) / max(len(Rs), 1)))
        c, d = int(s.require_coord), int(s.require_depth)
        ab = algorithmic_bytes(P, Pv, R, W * H, s.sh_degree, c, d)
        # per STEP, not per recorded interval: a stage may be timed in several pieces (block_lists: one kernel before the tile sort,
        # one after it)
        nrec = {k: v[1] for k, v in stages.items()}
        per_step = max(nrec.get("preprocess_bwd", 0) if dom != "preprocess_bwd" else nrec.get("blend_bwd", 0), 1)
        ms = {k: (v[0] / (v[1] if k == dom else per_step) if v[1] else 0.0) for k, v in stages.items()}
        grouped = {"preprocess_fwd": ms["preprocess_fwd"],
                   "binning": ms["sort_depth"] + ms["scan"] + ms["emit_instances"] + ms["sort_tile"] + ms["tile_ranges"],
                   "blend_fwd": ms["blend_fwd"] + ms.get("block_lists", 0.0), "blend_bwd": ms["blend_bwd"] + ms["acc_zero"], "preprocess_bwd": ms["preprocess_bwd"]}
        dom_ms = ms[dom]
        achieved = ab[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        gpu_ms = sum(grouped.values())
        kernel_name = {"blend_bwd": "blend_bwd_"}.get(dom, dom + "_")   # prefix of the kernel's name in the rocprof summaries
        traffic, traffic_note = pmc_traffic(kernel_name, args.config, P, W, H)
        out = {
            "metric": "fwd+bwd Msplats/s @1080p, 1M Gaussians; depth L1 vs ref" if args.config == "C2" else f"fwd+bwd Msplats/s, {args.config}", "value": round(value, 2), "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, {W}x{H}, SH degree {s.sh_degree}, fwd+bwd one view per step per GPU, "
                                   f"RGB{'+coord' if c else ''}{'+depth' if d else ''}{'+normal' if (c or d) else ''}",
                       "parallelism": f"view-parallel x{world}" + ((", RCCL all-reduce of 44 B + all-gather of 12 B/Gaussian/view (SH gradient factored)"
                                                                     if args.exchange == "factored" else ", RCCL all-reduce of 236 B/Gaussian grads")
                                                                    if world > 1 else ""),
                       "views": nviews, "num_rendered": int(R), "num_rendered_min_max": [int(min(Rs)), int(max(Rs))], "visible": Pv},
            "speculation": {"speculative_forwards": spec_calls, "redone": spec_misses,
                            "miss_rate": round(spec_misses / spec_calls, 4) if spec_calls else None},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": int(ab[dom]), "avg_launch_ms": round(dom_ms, 4),
                         "launches_timed": int(stages[dom][1]), "timed_with": "HIP events on the launch stream, inside the timed region"},
            "path_roofline": {"algorithmic_bytes_per_view": int(ab["total"]), "gpu_ms_per_view": round(gpu_ms, 4),
                              "achieved_GBs_gpu_time": round(ab["total"] / (gpu_ms * 1e-3) / 1e9, 1) if gpu_ms > 0 else 0.0,
                              "achieved_GBs_wall": round(ab["total"] / (ms_per_step * 1e-3) / 1e9, 1),
                              "frac_wall": round(ab["total"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "stages_ms": {k: round(v, 4) for k, v in ms.items()},
            "stages_ms_source": "warm-up steps with an event pair per stage; the roofline kernel is re-timed alone inside the timed steps",
            "stage_GBs": {k: round(ab[k] / (grouped[k] * 1e-3) / 1e9, 1) if grouped[k] > 0 else 0.0 for k in grouped},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_cpu, cfg["seed"])
        print(json.dumps(out), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel_name, config, P, W, H):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/*_pmc_per_kernel.json:
    separate `rocprofv3 --pmc` runs of this same command, TCC_EA0_RDREQ/WRREQ x 64 B as MI355X_MICROARCH.md's HBM section
    prescribes; its gfx950 note applies: 16-B/lane streaming reads may be under-counted up to 2x).  Counters cannot be
    collected inside a timed run, so the value is only reported for the workload the pass was made on; else null."""
    import glob
    from synth_scene import CONFIGS
    c = CONFIGS.get(config)
    if c is None or (P, W, H) != (c["P"], c["W"], c["H"]):
        return None, "no PMC pass for this workload"
    tag = "" if config == "C2" else "_" + config
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r[0-9][0-9]{tag}_pmc_per_kernel.json")))
    if not files:
        return None, f"profiles/rNN{tag}_pmc_per_kernel.json not found"
    try:
        d = json.load(open(files[-1]))
        for name, v in d.items():
            if kernel_name in name:
                return int((v["TCC_EA0_RDREQ_sum"] + v["TCC_EA0_WRREQ_sum"]) * 64), os.path.basename(files[-1]) + ": (RDREQ+WRREQ)*64 B per launch"
    except Exception as ex:  # the bench line must not die on a malformed side file
        return None, f"unreadable PMC summary: {ex}"
    return None, "kernel not in PMC summary"


def cpu_baseline(scene_cpu, seed):
    """The oracle (a CPU port of the reference algorithm; the reference itself is CUDA-only and cannot
    run here) timed on this box's host cores: one fwd+bwd pass over the SAME view."""
    from oracle.oracle import Oracle
    from synth_scene import upstream_grads
    s = scene_cpu
    cores = os.cpu_count() or 1
    o = Oracle(bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, campos=s.campos,
               tanfovx=s.tanfovx, tanfovy=s.tanfovy, image_height=s.H, image_width=s.W, shs=s.shs, scales=s.scales,
               rotations=s.rotations, sh_degree=s.sh_degree, kernel_size=s.kernel_size, require_coord=s.require_coord,
               require_depth=s.require_depth, nthreads=cores)
    g = upstream_grads(s, seed)
    t0 = time.perf_counter()
    o.forward()
    t1 = time.perf_counter()
    o.backward(g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
    t2 = time.perf_counter()
    P = s.means3D.shape[0]
    return {"value": round(P / 1e6 / (t2 - t0), 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
            "sample": f"1 fwd+bwd pass over the full workload view ({P} Gaussians, {s.W}x{s.H}); fwd {t1 - t0:.2f} s, bwd {t2 - t1:.2f} s, "
                      f"OpenMP over Gaussians/tiles"}


if __name__ == "__main__":
    main()
