#!/usr/bin/env python3
"""gpurun_out/profiles_<tag>/ (scratch, written on the GPU box by scripts/gpu_profile.sh) ->
profiles/<tag>_* (tracked): bench line, rocprofv3 kernel stats, per-kernel PMC averages."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"      # "r02" (C2) or "r02_C4" etc.
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
# kernel stats: keep our kernels + the device primitives, drop torch fill/copy noise below 0.1 %
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_rocprofv3_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        w.writerow(r)
out = {}
for name in ("pmc_sq", "pmc_tcc", "pmc_clk"):
    p = os.path.join(src, name + "_counters.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    rows_c = list(csv.DictReader(open(p)))
    # the counters of the LAST dispatches only: the first ones of a process run on cold caches / exact-size buffers
    per_kernel = collections.defaultdict(list)
    for r in rows_c:
        if "rg::" in r["Kernel_Name"]:
            # "void rg::(anonymous namespace)::scatter_kernel<8>(args...)" -> "rg::scatter_kernel<8>": the name up to its argument list
            # (round 3's summaries cut at the first "(", which merged every kernel of the sort's anonymous namespace into "void rg::")
            nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
            per_kernel[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
    # steps of the profiled run = dispatches of a kernel that runs exactly once per step (the bench's warm-up, clock-settling and timed
    # steps all count: 9 + 9 + the `settle` repetitions since round 6); PMC_RUN_STEPS overrides
    once = [len(v) for (k, c), v in per_kernel.items() if "preprocess_fwd_kernel" in k or "points_preprocess_kernel" in k]
    steps_in_run = int(os.environ.get("PMC_RUN_STEPS", "0")) or (max(once) if once else 18)
    for (k, c), vals in per_kernel.items():
        agg[k][c] = vals[len(vals) // 2:]
        out.setdefault(k, {})["dispatches_per_step"] = round(len(vals) / steps_in_run, 3)
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: round(sum(x) / len(x)) for c, x in v.items()})
for k, v in out.items():
    if "TCC_EA0_RDREQ_sum" in v:
        # FETCH/WRITE bytes as the microarch guide derives them: requests x 64 B (read side under-counts
        # wide streaming reads by up to 2x on gfx950 -- MI355X_MICROARCH.md "HBM")
        v["hbm_read_MB_64B_requests"] = round(v["TCC_EA0_RDREQ_sum"] * 64 / 1e6, 1)
        v["hbm_write_MB_64B_requests"] = round(v["TCC_EA0_WRREQ_sum"] * 64 / 1e6, 1)
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_bench.json")).read()[:400])
for r in rows[:14]:
    print(r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"))
