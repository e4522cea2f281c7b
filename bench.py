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

Order of a run: W warm-up steps (per-stage event timing -> `stages_ms`), fence, untimed repetitions of the step for `--settle-ms`
(default 40 ms; reported as `settle`: the GPU's clock settles only after ~20 ms of continuous work, 0 turns it off), fence
(barrier + synchronize), EXACTLY K timed steps, fence.  `step_gpu_span_ms` carries every timed step's GPU span in order.

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
# VALU issue peak in wave64 instructions: 256 CUs x 4 SIMDs x 2.4 GHz, one instruction every 2 cycles per SIMD (same guide: "issues each
# VALU instruction over 2 cycles"; 157.3 TFLOP/s fp32 = this x 64 lanes x 2 flop)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.3   # 1 024 SIMDs, 2.4 GHz, 2.3 cycles per full-rate wave64 instruction (profiles/r06_valu_calibration.txt)


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


def binning_bytes_moved(P, R, tiles, streams):
    """What THIS implementation's binning moves, as opposed to the reference algorithm's 16 P + 164 R that
    `algorithmic_bytes` charges: depth sort of P 32-bit keys, 4 passes x (histogram reads the key, scatter reads key (+ value after
    the first pass) and writes both) = 76 P; gather-scan 20 P; emission reads 12 P (index, offset, packed rectangle) + the 64-B record
    of every visible splat when block masks are computed, writes 6 R (16-bit tile key + value; 8 R above 65 536 tiles); tile sort
    2 passes x 14 R (20 R with 32-bit keys); ranges 2 R + 8 tiles."""
    k = 2 if tiles <= 65535 else 4
    return 76 * P + 20 * P + 12 * P + (4 + k) * R + 2 * (8 + 3 * k) * R + k * R + 8 * tiles + (64 * P if streams else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=float, default=40.0,
                    help="untimed repetitions of the step between the warm-up and the timed region until this much time has gone by, so that the GPU "
                         "has reached its sustained clock when timing starts (0 = none; reported on the line as `settle`)")
    ap.add_argument("--config", default="C2", help="BASELINE.json config (C1..C5); the headline metric is quoted on C2")
    ap.add_argument("--points", type=int, default=0, help="override the number of Gaussians (debug only)")
    ap.add_argument("--mu-px", type=float, default=0.0, help="override the median splat size in pixels (debug only)")
    ap.add_argument("--views", type=int, default=9, help="distinct camera views the steps rotate through (1 = the same view every step)")
    ap.add_argument("--flags", choices=("config", "both"), default="config",
                    help="'both': require_coord = require_depth = True whatever the config says -- the render.py mode "
                         "(gaussian_renderer/__init__.py:19,71-79: render() defaults), SURVEY.md 8(d)")
    ap.add_argument("--mode", choices=("train", "forward"), default="train",
                    help="'forward': the step is the forward alone under no_grad -- what the reference's inference callers execute "
                         "(render.py:32, mesh_extract.py:56: render() with both maps on; use with --flags both); metric Mimages/s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the legs that run AFTER (outside) the timed region at N=1 on the headline config: one short bench line each for "
                         "the other BASELINE configs with a GPU (C3, C4, C5 -> `other_configs`) and one full training iteration (`train_iter_ms`)")
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
    if args.flags == "both":
        over["require_coord"] = over["require_depth"] = True
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
            bucket = FactoredGradExchange(P, s.shs.shape[1], s.sh_degree, dev, timing=True)
        else:
            bucket = GradBucket(P, s.shs.shape[1], dev, timing=True)
        C.set_grad_allocator(dev, bucket.allocator)
    counter = [0]
    last_state = []
    STAGES4 = ("preprocess_fwd", "blend_fwd") if args.mode == "forward" else ("preprocess_fwd", "blend_fwd", "blend_bwd", "preprocess_bwd")

    def step():
        vm, pm, cp = cams[counter[0] % nviews]
        counter[0] += 1
        if bucket is not None and args.exchange == "factored":
            bucket.set_view(cp)      # lets the all-gather of the dL/dRGB rows start under the per-Gaussian backward kernel
        fw = C.rasterize_gaussians(s.bg, s.means3D, e, s.opacities, s.scales, s.rotations, 1.0, e, vm, pm, s.tanfovx,
                                   s.tanfovy, s.kernel_size, H, W, s.shs, s.sh_degree, cp, False, s.require_coord,
                                   s.require_depth, False)
        R, color, coord, mcoord, alpha, normal, depth, mdepth, radii, geom, binning, img = fw
        last_state[:] = [R, geom, binning, img]
        if args.mode == "forward":
            return R, radii, None
        bw = C.rasterize_gaussians_backward(s.bg, s.means3D, radii, e, s.scales, s.rotations, 1.0, e, vm, pm,
                                            s.tanfovx, s.tanfovy, s.kernel_size, g["color"], g["coord"], g["mcoord"], g["depth"],
                                            g["mdepth"], g["alpha"], g["normal"], normal, s.shs, s.sh_degree, cp, geom, R,
                                            binning, img, alpha, s.require_coord, s.require_depth, False)
        grads = dict(dL_dmeans3D=bw[3], dL_dsh=bw[5], dL_dopacity=bw[2], dL_dscales=bw[6], dL_drotations=bw[7])
        last_state[:] = [R, geom, binning, img]
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
    # Python's cyclic collector is emptied HERE -- before the warm-up, not between the warm-up and the timed region -- and held off until
    # the timed region ends: a generation-2 pass over a torch process's ~1 M objects is a 5-30 ms host stall (seen as ONE step of 6 / 30 /
    # 31 ms on C2 / C4 / C5 lines of round 6 whose median span was 1.41 / 4.09 / 7.71).  Emptying it right before the timed region (the
    # round's first version) idled the GPU for that long after its warm-up: the clocks fell back and the first six timed steps ran 10-20 %
    # slow (`step_gpu_span_ms.in_step_order`: 1.50 1.52 1.59 1.49 1.46 1.43 1.37 ... 1.30).  Nothing between the warm-up's closing fence and
    # the first timed step takes longer than a few hundred microseconds now.  No GPU work is skipped or moved by any of this.
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    # one event per step boundary on the launch stream (~2 us each, no bubble: nothing waits on them): the spans between them say what a
    # step costs on the GPU, so that a host stalled by a neighbour (the boxes' hosts are shared) shows up as wall >> median span
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
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
        dom = max(STAGES4, key=lambda k: stages_warm[k][0] / max(stages_warm[k][1], 1))
    # Clock settling (disclosed on the line as `settle`): after an idle period the GPU reaches its sustained clock only after ~20 ms of
    # continuous work -- with `--views 1` as with rotating views the first ~15 steps after the warm-up's fence run 10 % slow and decay
    # smoothly (`step_gpu_span_ms.in_step_order` of a run with --settle-ms 0: 1.42 1.37 1.41 1.38 1.36 ... 1.29), and W = 5 warm-up steps
    # are 7 ms.  Throughput is a steady-state quantity (a training run is thousands of iterations), so the same step is repeated, untimed,
    # until `--settle-ms` of GPU work have gone by; the timed K steps follow the usual barrier + synchronize.  Nothing inside the timed
    # region changes.
    n_settle, settle_ms_per_step = 0, None
    if args.settle_ms > 0 and args.warmup > 0:
        # how many: from the duration of the first four, the SAME count on every rank (under the launcher a step ends with a collective:
        # ranks that counted by their own clocks would issue different numbers of them and hang)
        t_s = time.perf_counter()
        for _ in range(4):
            R, radii, _ = step()
        torch.cuda.synchronize(dev)
        est_ms = (time.perf_counter() - t_s) * 1e3 / 4
        if launched:
            t_est = torch.tensor([est_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t_est, op=dist.ReduceOp.MAX)
            est_ms = float(t_est.item())
        n_settle = min(200, max(4, -(-int(args.settle_ms * 1000) // max(int(est_ms * 1000), 1))))
        n_settle = (n_settle + 3) // 4 * 4
        for _ in range(n_settle - 4):
            R, radii, _ = step()
        fence()
        settle_ms_per_step = (time.perf_counter() - t_s) * 1e3 / n_settle   # what these steps -- the first after the warm-up -- took
    # the dominant kernel is timed live on every 2nd step of the timed region (the event pair around it is a ~12 us stream bubble;
    # every 2nd step of a 9-view rotation still visits every view)
    C.profile_enable(True, only=dom, every=2 if (dom is not None and args.steps >= 8) else 1)   # no warm-up steps: every stage, in the timed region
    if bucket is not None and hasattr(bucket, "collect_timing"):
        bucket.collect_timing()      # drop the warm-up steps' exchange timings
    C.binning_stats(reset=True)
    Rs, vis = [], []
    torch.cuda.reset_peak_memory_stats(dev)
    mem_before = torch.cuda.memory_allocated(dev)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        R, radii, _ = step()
        marks[i + 1].record()
        Rs.append(R)
        vis.append(radii)          # kept alive, counted after the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if gc_was_enabled:
        gc.enable()
    spans = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    xch = bucket.collect_timing() if (bucket is not None and hasattr(bucket, "collect_timing")) else []
    peak_step_bytes = torch.cuda.max_memory_allocated(dev) - mem_before - sum(r.numel() * 4 for r in vis[:-1])
    C.profile_enable(False)
    spec_calls, spec_misses = C.binning_stats()
    timed = C.profile_collect()
    if dom is None:
        stages = timed
        dom = max(STAGES4, key=lambda k: timed[k][0] / max(timed[k][1], 1))
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
        if args.mode == "forward":
            per_step = max(nrec.get("preprocess_fwd", 0) if dom != "preprocess_fwd" else nrec.get("blend_fwd", 0), 1)
        else:
            per_step = max(nrec.get("preprocess_bwd", 0) if dom != "preprocess_bwd" else nrec.get("blend_bwd", 0), 1)
        ms = {k: (v[0] / (v[1] if k == dom else per_step) if v[1] else 0.0) for k, v in stages.items()}
        grouped = {"preprocess_fwd": ms["preprocess_fwd"],
                   "binning": ms["sort_depth"] + ms["scan"] + ms["emit_instances"] + ms["sort_tile"] + ms["tile_ranges"],
                   "blend_fwd": ms["blend_fwd"] + ms.get("block_lists", 0.0), "blend_bwd": ms["blend_bwd"] + ms["acc_zero"], "preprocess_bwd": ms["preprocess_bwd"]}
        dom_ms = ms[dom]
        achieved = ab[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        gpu_ms = sum(grouped.values())
        kernel_name = {"blend_bwd": "blend_bwd_"}.get(dom, dom + "_")   # prefix of the kernel's name in the rocprof summaries
        tag = args.config if args.flags == "config" else args.config + "_both"
        if args.mode == "forward":
            tag += "_fwd"
            for k in ("blend_bwd", "preprocess_bwd"):
                ab["total"] -= ab[k]
                ab[k] = 0
        traffic, traffic_note = pmc_traffic(kernel_name, tag, args.config, P, W, H)
        pairs = pair_evaluations(C, s, last_state, c)
        streams = pairs["formulation"].startswith("entry streams")
        bmoved = binning_bytes_moved(P, R, ((W + 15) // 16) * ((H + 15) // 16), streams)
        valu = valu_roofline(kernel_name, tag, args.config, P, W, H, dom_ms)
        if args.mode == "forward":
            metric, unit, value = f"forward-only Mimages/s, {args.config} (render.py / mesh_extract.py: render() under no_grad)", "Mimages/s", world / 1e6 / (elapsed / args.steps)
        else:
            metric, unit = ("fwd+bwd Msplats/s @1080p, 1M Gaussians; depth L1 vs ref" if args.config == "C2" else f"fwd+bwd Msplats/s, {args.config}"), "Msplats/s"
        out = {
            "metric": metric + (" [render.py mode: coord + depth maps]" if args.flags == "both" else ""), "value": round(value, 6 if args.mode == "forward" else 2), "unit": unit,
            "images_per_s": round(world / (elapsed / args.steps), 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {P} Gaussians, {W}x{H}, SH degree {s.sh_degree}, {'forward only (no_grad)' if args.mode == 'forward' else 'fwd+bwd'} one view per step per GPU, "
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
            "settle": {"untimed_steps": n_settle, "ms": args.settle_ms, "ms_per_step_during_settle": None if settle_ms_per_step is None else round(settle_ms_per_step, 4),
                       "note": "untimed repetitions of the step between the W warm-up steps and the timed region, until the GPU has worked for `ms`: "
                               "its clock settles only after ~20 ms of continuous work (first steps after an idle period run ~10 % slow); "
                               "ms_per_step_during_settle = wall clock of these very steps, i.e. what a line without them would read"},
            "step_gpu_span_ms": {"median": round(spans[len(spans) // 2], 4), "min": round(spans[0], 4), "max": round(spans[-1], 4), "mean": round(sum(spans) / len(spans), 4), "in_step_order": [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(args.steps)],
                                 "note": "HIP-event spans between consecutive steps on the launch stream; ms_per_step is the wall clock of the whole region / steps; "
                                         "Python's cyclic GC is collected before and disabled during the timed region (a gen-2 pass is a 5-30 ms host stall)"},
            "device_memory": {"peak_bytes_of_one_step": int(peak_step_bytes),
                              "reference_formula_bytes": int(139 * P + 44 * W * H + 24 * R),
                              "reference_with_outputs_and_gradients_bytes": int(139 * P + 44 * W * H + 24 * R + 60 * W * H + 360 * P),
                              "note": "peak torch allocation above the resident inputs during the timed steps (state buffers incl. entry-stream capacity, "
                                      "outputs, accumulators, gradients); reference: 139 B/Gaussian + 44 B/pixel + 24 B/instance of state "
                                      "(rasterizer_impl.cu:190-250); second figure: plus its 15 output planes (60 B/pixel) and its gradient / per-Gaussian "
                                      "sum tensors (360 B/Gaussian at SH degree 3, rasterize_points.cu:175-193), still without its sort temp"},
            "stages_ms": {k: round(v, 4) for k, v in ms.items()},
            "stages_ms_source": "warm-up steps with an event pair per stage; the roofline kernel is re-timed alone inside the timed steps",
            "stage_GBs": {k: round(ab[k] / (grouped[k] * 1e-3) / 1e9, 1) if grouped[k] > 0 else 0.0 for k in grouped},
            # the binning line above divides the REFERENCE algorithm's bytes (6-pass 64-bit sort) by this implementation's time; what the
            # implementation itself moves is about half of that (binning_bytes_moved)
            "stage_bytes_moved": stage_bytes_moved(tag, args.config, P, W, H, grouped),
            "binning_bytes_moved": {"bytes": int(bmoved), "GBs": round(bmoved / (grouped["binning"] * 1e-3) / 1e9, 1) if grouped["binning"] > 0 else 0.0},
            # the render kernels are gather + ALU bound -- (pixel, list entry) evaluations per second next to their GB/s
            "pairs": {"formulation": pairs["formulation"], "evaluated_per_pass": pairs["evaluated"], "reference_definition_per_pass": pairs["tilewide"],
                      "blend_fwd_Gpairs_s": round(pairs["evaluated"] / (ms["blend_fwd"] * 1e-3) / 1e9, 1) if ms["blend_fwd"] > 0 else 0.0,
                      "blend_bwd_Gpairs_s": round(pairs["evaluated"] / (ms["blend_bwd"] * 1e-3) / 1e9, 1) if ms["blend_bwd"] > 0 else 0.0,
                      "note": "evaluated = pixel slots the kernels walk in the last view of the run (entry streams: 32 x entries each 8x4 block "
                              "consumed); reference_definition = 256 x entries each tile's walk reaches (SURVEY 8d)"},
            "valu_roofline": valu,
        }
        if xch:   # N > 1 (or --force-allreduce): what the gradient exchange costs and how much of it the step waits for (view_parallel.py)
            out["exchange_ms"] = round(sum(t[0] for t in xch) / len(xch), 4)
            out["exchange_exposed_ms"] = round(sum(t[1] for t in xch) / len(xch), 4)
            out["exchange_note"] = ("rank 0, mean over the timed steps, GPU clock: exchange_ms = first collective of the step issued (under the "
                                    "backward) -> gradients ready; exchange_exposed_ms = the part after the backward's last kernel was queued")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene_cpu, cfg["seed"])
        if (world == 1 and not launched and not args.no_other_configs and args.config == "C2" and args.flags == "config" and args.mode == "train"
                and not args.points and not args.mu_px):
            del s, g, cams                      # the other configs run in child processes of their own: give the memory back first
            torch.cuda.empty_cache()
            out["other_configs"], out["train_iter_ms"] = other_configs_and_train_iter()
        print(json.dumps(out), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()


def other_configs_and_train_iter():
    """Outside the timed region, after it: the other BASELINE configs that run on one GPU, each as a child `bench.py --config Cx`
    (10 steps), reduced to {ms_per_step, value, roofline kernel and fraction}; and scripts/gpu_train_iter.py's full training iteration
    (3D filter -> rasterizer -> L1/SSIM + normal loss -> backward -> Adam at C2 scale).  A leg that fails or runs out of time reports
    that instead of a number -- the headline line is printed either way, and all legs together get at most 240 s (a leg 120 s)."""
    import re
    import subprocess
    res = {}
    t_start = time.time()
    BUDGET, LEG = 240.0, 120.0

    def left():
        return BUDGET - (time.time() - t_start)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-other-configs"]
    legs = [("C4", ["--config", "C4"]), ("C5", ["--config", "C5"]), ("C3", ["--config", "C3"]), ("C2_both_maps", ["--flags", "both"]),
            ("C2_both_maps_forward_only", ["--flags", "both", "--mode", "forward"])]
    for name, extra in legs:
        if left() < 20.0:
            res[name] = {"error": "skipped: the 240 s the legs share were used up"}
            continue
        try:
            p = subprocess.run(base + extra, capture_output=True, text=True, timeout=min(LEG, left()))
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[name] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "roofline_kernel": d["roofline"]["kernel"],
                         "roofline_frac": d["roofline"]["frac"], "blend": d["pairs"]["formulation"], "num_rendered": d["config"]["num_rendered"]}
        except Exception as ex:   # noqa: BLE001 -- a reporting leg must not take the headline line with it
            res[name] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    train = None
    try:
        if left() < 20.0:
            raise TimeoutError("skipped")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_train_iter.py")], capture_output=True, text=True, timeout=min(LEG, left()))
        m = re.search(r"full training iteration.*?:\s*([0-9.]+) ms", p.stdout)
        train = float(m.group(1)) if m else None
    except Exception:   # noqa: BLE001
        train = None
    res["_seconds"] = round(time.time() - t_start, 1)
    return res, train


def pair_evaluations(C, s, last_state, coord):
    """(pixel, list entry) evaluations of one blend pass over the last view rendered, from the state the forward left."""
    import numpy as np
    R, geom, binning, img = last_state
    W, H, P = s.W, s.H, s.means3D.shape[0]
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    if R == 0:
        return {"formulation": "empty", "evaluated": 0, "tilewide": 0}
    ncon = C.debug_export("n_contrib", torch.int32, 2 * W * H, P, R, W, H, coord, geom, binning, img)[: W * H].view(H, W)
    pad = torch.zeros(((H + 15) // 16) * 16, ((W + 15) // 16) * 16, dtype=torch.int32, device=ncon.device)
    pad[:H, :W] = ncon
    reach = pad.view(pad.shape[0] // 16, 16, pad.shape[1] // 16, 16).amax(dim=(1, 3)).to(torch.int64)   # per tile: last entry any pixel blended
    tilewide = int(256 * reach.sum().item())
    streams = bool(C.last_forward_used_streams())   # the library's own decision for the forward that left this state (include/radegs.h)
    if streams:
        cons = C.debug_export("blk_consumed", torch.int32, 8 * tiles, P, R, W, H, coord, geom, binning, img).to(torch.int64)
        return {"formulation": "entry streams (8x4-pixel blocks)", "evaluated": int(32 * cons.sum().item()), "tilewide": tilewide}
    return {"formulation": "tile-wide lists", "evaluated": tilewide, "tilewide": tilewide}


# which stage of the bench line a kernel of the committed PMC pass belongs to (name fragments, profiles/*_pmc_per_kernel.json)
_STAGE_OF = (("preprocess_fwd_kernel", "preprocess_fwd"), ("preprocess_bwd_kernel", "preprocess_bwd"), ("drgb_clamped", "preprocess_bwd"),
             ("blend_fwd", "blend_fwd"), ("block_lists", "blend_fwd"), ("block_counts", "blend_fwd"), ("balance_blocks", "blend_fwd"), ("blend_bwd", "blend_bwd"),
             ("digit_histogram", "binning"), ("scan_rows", "binning"), ("scatter_kernel", "binning"), ("gather_block_sums", "binning"),
             ("gather_scan", "binning"), ("emit_instances", "binning"), ("tile_ranges", "binning"))


def stage_bytes_moved(tag, config, P, W, H, grouped_ms):
    """Per stage: HBM bytes the kernels really moved per step (committed PMC pass: (2 x TCC_EA0_RDREQ + TCC_EA0_WRREQ) x 64 B per dispatch,
    gfx950 read correction as in pmc_traffic(), x dispatches per step) and the GB/s that makes of this run's stage times -- next to
    `stage_GBs`, which divides SURVEY 8(d)'s MODEL bytes.  None without a PMC pass for this workload (or one from before round 4, whose
    summaries merged the sort kernels' names)."""
    f = _pmc_file(tag, config, P, W, H)
    if f is None:
        return None
    try:
        d = json.load(open(f))
        if not any("dispatches_per_step" in v for v in d.values()):
            return None
        tot = {}
        for name, v in d.items():
            st = next((stg for frag, stg in _STAGE_OF if frag in name), None)
            if st is None or "TCC_EA0_RDREQ_sum" not in v:
                continue
            tot[st] = tot.get(st, 0.0) + v.get("dispatches_per_step", 1.0) * (2 * v["TCC_EA0_RDREQ_sum"] + v["TCC_EA0_WRREQ_sum"]) * 64
        return {"source": os.path.basename(f),
                "bytes": {k: int(b) for k, b in tot.items()},
                "GBs": {k: round(b / (grouped_ms[k] * 1e-3) / 1e9, 1) for k, b in tot.items() if grouped_ms.get(k, 0) > 0},
                "frac_of_hbm_peak": {k: round(b / (grouped_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, b in tot.items() if grouped_ms.get(k, 0) > 0}}
    except Exception:
        return None


def _pmc_file(tag, config, P, W, H):
    import glob
    from synth_scene import CONFIGS
    c = CONFIGS.get(config)
    if c is None or (P, W, H) != (c["P"], c["W"], c["H"]):
        return None
    suffix = "" if tag == "C2" else "_" + tag
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r[0-9][0-9]{suffix}_pmc_per_kernel.json")))
    return files[-1] if files else None


def valu_roofline(kernel_name, tag, config, P, W, H, dom_ms):
    """The bound the blend kernels actually run into: wave64 VALU instructions per launch from the committed
    PMC pass (SQ_INSTS_VALU) / the live launch duration, against the issue peak of the chip."""
    f = _pmc_file(tag, config, P, W, H)
    if f is None or dom_ms <= 0:
        return None
    try:
        for name, v in json.load(open(f)).items():
            if kernel_name in name and "SQ_INSTS_VALU" in v:
                ach = v["SQ_INSTS_VALU"] / (dom_ms * 1e-3) / 1e9
                return {"bound": "valu", "kernel": kernel_name, "achieved": round(ach, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s",
                        "frac": round(ach / VALU_PEAK_GINST, 4), "wave_instructions_per_launch": int(v["SQ_INSTS_VALU"]),
                        "source": os.path.basename(f) + ": SQ_INSTS_VALU per launch (separate --pmc pass); peak = 1024 SIMDs x 2.4 GHz / 2.3 cycles per "
                                  "full-rate instruction, measured (profiles/r06_valu_calibration.txt); half-rate classes (DPP, select, compare: "
                                  "4.3 cycles), transcendentals (8.2) and the 2.0-GHz clock under fp32 load make 1.0 unreachable: the kernel's "
                                  "own busy fraction is in profiles/r06_valu_busy.txt"}
    except Exception:
        return None
    return None


def pmc_traffic(kernel_name, tag, config, P, W, H):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/*_pmc_per_kernel.json:
    separate `rocprofv3 --pmc` runs of this same command, TCC_EA0_RDREQ/WRREQ x 64 B as MI355X_MICROARCH.md's HBM section
    prescribes, with its gfx950 correction applied: 16-B/lane reads are tallied at half their bytes, so the read side is doubled).  Counters cannot be
    collected inside a timed run, so the value is only reported for the workload the pass was made on; else null."""
    f = _pmc_file(tag, config, P, W, H)
    if f is None:
        return None, "no PMC pass for this workload"
    files = [f]
    try:
        d = json.load(open(files[-1]))
        for name, v in d.items():
            if kernel_name in name:
                # the guide's gfx950 correction: wide (16 B per lane) reads are tallied at half their bytes -- every global read of the
                # blend / per-Gaussian kernels is a dwordx4 -- so the read side is doubled; the write side is taken as counted
                rd, wr = v["TCC_EA0_RDREQ_sum"] * 64, v["TCC_EA0_WRREQ_sum"] * 64
                return int(2 * rd + wr), (os.path.basename(files[-1]) + f": 2 x TCC_EA0_RDREQ x 64 B (gfx950: 16-B/lane reads counted at half) + "
                                          f"TCC_EA0_WRREQ x 64 B per launch; raw counters: read {int(rd)} B, write {int(wr)} B")
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
    port = {"value": round(P / 1e6 / (t2 - t0), 4), "unit": "Msplats/s", "cores": cores, "kind": "port",
            "sample": f"1 fwd+bwd pass over the full workload view ({P} Gaussians, {s.W}x{s.H}); fwd {t1 - t0:.2f} s, bwd {t2 - t1:.2f} s, "
                      f"OpenMP over Gaussians/tiles"}
    o.close()
    ref = cpu_baseline_reference(s, g, cores)
    if ref is None:
        return port
    ref["port"] = {k: port[k] for k in ("value", "sample")}
    return ref


def cpu_baseline_reference(s, g, cores):
    """kind "reference": the reference's OWN rasterizer sources (forward.cu / backward.cu / rasterizer_impl.cu) compiled for the host
    (oracle/_ref, built by oracle/build_ref.py where /root/reference exists; the .so travels to the GPU box) and driven through
    CudaRasterizer::Rasterizer::forward / backward: thread blocks as fibers, blocks spread over the host cores with OpenMP.  None when
    the library is not there."""
    try:
        from oracle import ref
        if not os.path.exists(ref.lib_path()):
            return None
        ref.set_exp("libm")
        ref.set_num_threads(cores)
        r = ref.Ref(bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, campos=s.campos,
                    tanfovx=s.tanfovx, tanfovy=s.tanfovy, image_height=s.H, image_width=s.W, shs=s.shs, scales=s.scales,
                    rotations=s.rotations, sh_degree=s.sh_degree, kernel_size=s.kernel_size, require_coord=s.require_coord,
                    require_depth=s.require_depth)
        t0 = time.perf_counter()
        r.forward()
        t1 = time.perf_counter()
        r.backward(g["color"], g["coord"], g["mcoord"], g["depth"], g["mdepth"], g["alpha"], g["normal"])
        t2 = time.perf_counter()
        ref.set_num_threads(1)
        P = s.means3D.shape[0]
        return {"value": round(P / 1e6 / (t2 - t0), 4), "unit": "Msplats/s", "cores": cores, "kind": "reference",
                "sample": f"1 fwd+bwd pass over the full workload view ({P} Gaussians, {s.W}x{s.H}) through the reference's own sources compiled "
                          f"for the host (CUDA blocks as fibers, OpenMP over blocks); fwd {t1 - t0:.2f} s, bwd {t2 - t1:.2f} s"}
    except Exception as ex:   # the bench line must not die on the optional leg
        sys.stderr.write(f"[bench] reference cpu_baseline skipped: {ex}\n")
        return None


if __name__ == "__main__":
    main()
