#!/usr/bin/env python3
"""Which kernels are bound by VALU issue?  From the committed PMC pass and rocprofv3 kernel stats of one workload:
busy = SQ_ACTIVE_INST_VALU (quad-cycles a SIMD spends issuing VALU work, summed over the chip) x 4 / 1024 SIMDs, against the kernel's
average duration x clock.  DESIGN.md section 4.3.      python scripts/valu_busy.py [r05] [clock GHz = 2.1]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_per_kernel.json")))
dur = {}
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_stats.csv"))):
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    dur[name] = float(r["AverageNs"]) * 1e-3
print(f"# {tag}: per launch; VALU busy cycles per SIMD = SQ_ACTIVE_INST_VALU x 4 / 1024; duration in cycles at {ghz} GHz (under VALU load the clock sits at 1.9-2.1 GHz)")
print(f"# {'kernel':52s} {'us':>8s} {'Minstr':>8s} {'busy kcyc':>10s} {'dur kcyc':>9s} {'busy':>6s}")
rows = []
for k, v in pmc.items():
    if k not in dur or "SQ_ACTIVE_INST_VALU" not in v:
        continue
    busy = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0
    cyc = dur[k] * 1e-6 * ghz * 1e9
    rows.append((dur[k] * v.get("dispatches_per_step", 1.0), k, dur[k], v["SQ_INSTS_VALU"] / 1e6, busy / 1e3, cyc / 1e3, busy / cyc))
for _, k, us, mi, b, c, f in sorted(rows, reverse=True):
    print(f"{k:54s} {us:8.1f} {mi:8.2f} {b:10.1f} {c:9.1f} {f:6.2f}")
