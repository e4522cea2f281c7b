#!/usr/bin/env python3
"""What the SQ counters count and what a VALU instruction costs on gfx950, from the CSVs scripts/gpu_valu_calibration.sh
left in gpurun_out/<tag>_valu_cal/: every kernel of scripts/ubench/valu_rates issues iters x 64 instructions of ONE class per wave,
4 waves per SIMD on all 1 024 SIMDs -- a known count -- so per class:

    SQ_INSTS_VALU        per issued wave64 instruction                (is it 1?)
    SQ_ACTIVE_INST_VALU  per issued wave64 instruction                (a busy-cycle counter would differ between the classes)
    clock                GRBM_GUI_ACTIVE / 8 XCDs / duration          (the kernel's own clock under that load)
    cycles               GRBM_GUI_ACTIVE / 8 / instructions per SIMD  (SIMD cycles one instruction of the class occupies the pipe)

    python scripts/valu_calibration.py r6 > profiles/r06_valu_calibration.txt"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
src = os.path.join(ROOT, "gpurun_out", f"{tag}_valu_cal")


def load(p):
    tr, cnt = {}, collections.defaultdict(dict)
    for r in csv.DictReader(open(os.path.join(src, f"{p}_trace.csv"))):
        tr[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) // 64)
    for r in csv.DictReader(open(os.path.join(src, f"{p}_counters.csv"))):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    out, seen = {}, collections.Counter()
    for d, (name, ns, waves) in sorted(tr.items(), key=lambda kv: int(kv[0])):
        seen[name] += 1
        if seen[name] == 2:   # the timed launch (the first one of every kernel is the 10-iteration warm-up)
            out[name] = (ns, waves, cnt[d])
    return out


ITERS, PER_ITER, SIMDS, XCDS = 2000, 64, 1024, 8
p1, p4 = load("p1"), load("p4")
print(f"# {tag}: scripts/ubench/valu_rates under rocprofv3 (scripts/gpu_valu_calibration.sh); {ITERS} x {PER_ITER} instructions of one class per wave, 4 waves per SIMD")
print(f"# {'kernel (instruction class)':28s} {'us':>8s} {'INSTS_VALU/instr':>17s} {'ACTIVE_INST_VALU/instr':>23s} {'clock GHz':>10s} {'cycles/instr':>13s}")
for name, (ns, waves, c) in p4.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_INSTS_VALU" not in c:
        continue
    issued = waves * ITERS * PER_ITER
    if name == "k_cnd32_init":
        issued = waves * ITERS * 72
    cyc = c["GRBM_GUI_ACTIVE"] / XCDS
    print(f"{name:30s} {ns * 1e-3:8.1f} {c['SQ_INSTS_VALU'] / issued:17.3f} {c['SQ_ACTIVE_INST_VALU'] / issued:23.3f} {cyc / ns:10.3f} {cyc / (issued / SIMDS):13.2f}")
print("# SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (first pass / fourth pass, same kernels):",
      ", ".join(f"{k} {p1[k][2]['SQ_BUSY_CYCLES'] / p4[k][2]['GRBM_GUI_ACTIVE']:.2f}" for k in ("k_fma3", "k_adddpp", "k_exp") if k in p1 and k in p4))
print("# Reading: SQ_INSTS_VALU counts wave64 instructions; SQ_ACTIVE_INST_VALU counts the SAME (1 per instruction of every class, 2 per transcendental) --")
print("# it is an issue count, not a busy-cycle count, so 'ACTIVE x 4 / SIMDs / duration' (rounds 3-5) is not a utilisation.  GRBM_GUI_ACTIVE is")
print("# summed over the 8 XCDs: / 8 = the kernel's cycles, / duration = its clock: 1.98 GHz under full-rate fp32 load, 2.3-2.4 GHz for the")
print("# half- and quarter-rate classes.  SQ_BUSY_CYCLES is summed over 32 shader engines (4 x GRBM_GUI_ACTIVE).")
