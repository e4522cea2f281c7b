#!/usr/bin/env python3
"""profiles/<tag>_step_timeline.txt from gpurun_out/trace/kernel_trace.csv (scripts/gpu_trace.sh): one bench step, kernel by kernel."""
import csv
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(os.path.join(root, "gpurun_out/trace/kernel_trace.csv")))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r[2]]
seg = rows[idx[-2]:idx[-1]]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
out = ["# One bench step (C2, rotating views) under `rocprofv3 --kernel-trace` (scripts/gpu_trace.sh): kernel start gaps and durations, microseconds.",
       "# step span %.1f us, kernel time %.1f us, %d kernels (the host looks at num_rendered only after the blend is queued: no stall)" % (span / 1e3, busy / 1e3, len(seg)),
       "# gap_before  duration  kernel"]
prev = None
for s, e, n in seg:
    gap = (s - prev) / 1e3 if prev else 0.0
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("rg::", "")
    n = re.sub(r"\(.*", "", n)
    out.append("%10.1f %9.1f  %s" % (max(gap, 0.0), (e - s) / 1e3, n[:90]))
    prev = e
open(os.path.join(root, "profiles", tag + "_step_timeline.txt"), "w").write("\n".join(out) + "\n")
print(out[1])
