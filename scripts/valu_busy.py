#!/usr/bin/env python3
"""How busy is the VALU pipe in every kernel of a step?  A MEASURED statement, built from

  * the committed PMC summary of one workload (profiles/<tag>_pmc_per_kernel.json): SQ_INSTS_VALU (wave64 instructions issued),
    SQ_ACTIVE_INST_VALU (the same count with transcendentals counted twice -- NOT a busy-cycle counter: profiles/r06_valu_calibration.txt)
    and GRBM_GUI_ACTIVE (summed over the 8 XCDs: / 8 = the kernel's own cycles, so no clock has to be assumed);
  * the calibration of scripts/ubench/valu_rates on the same part (profiles/r06_valu_calibration.txt): SIMD cycles one wave64 instruction
    occupies the pipe -- 2.3 full-rate (v_fma/mul/add/and/mov ...), 4.3 half-rate (every DPP form, v_cmp, v_cndmask_e64, v_min/max,
    v_bfe, v_lshl_or/add, v_cvt, v_rndne, packed fp32), 8.2 quarter-rate (v_exp, v_rcp, v_sqrt, permlane swaps);
  * the instruction-class shares of the kernel's code (scripts/isa_mix.py on the in-tree library): the share of half-rate
    instructions in its hottest loop (whole kernel where it has no loop); the quarter-rate count is dynamic: ACTIVE - INSTS.

    busy = [ (I - Q - h I) x 2.3 + h I x 4.3 + Q x 8.2 ] / 1 024 SIMDs / (GRBM_GUI_ACTIVE / 8)          I = SQ_INSTS_VALU, Q = ACTIVE - I

    python scripts/valu_busy.py [r06] > profiles/r06_valu_busy.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_mix  # noqa: E402

C_FULL, C_HALF, C_QUARTER, SIMDS, XCDS = 2.3, 4.3, 8.2, 1024, 8   # profiles/r06_valu_calibration.txt: 2.27-2.33 / 4.18-4.53 / 8.18-8.22
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_per_kernel.json")))
KERNELS = {   # PMC name -> substring of the mangled name
    "rg::blend_bwd_streams_kernel<false, true>": "blend_bwd_streams_kernelILb0ELb1EE", "rg::blend_fwd_streams_kernel<false, true>": "blend_fwd_streams_kernelILb0ELb1EE",
    "rg::preprocess_bwd_kernel": "preprocess_bwd_kernel", "rg::preprocess_fwd_kernel<false>": "preprocess_fwd_kernelILb0E",
    "rg::emit_instances_kernel<true>": "emit_instances_kernelILb1E", "rg::block_lists_kernel<false>": "block_lists_kernelILb0E",
    "rg::block_counts_kernel<false>": "block_counts_kernelILb0E", "rg::scatter_kernel<8, unsigned int, 512>": "scatter_kernelILi8EjLi512E",
    "rg::scatter_kernel<16, unsigned short, 256>": "scatter_kernelILi16EtLi256E"}


def half_share(parts):
    """share of half-rate instructions among the VALU instructions of the kernel's hottest loop (whole kernel without one)"""
    try:
        name, asm = isa_mix.disassemble([parts])
    except SystemExit:
        return None
    ins = isa_mix.parse(asm)
    spans = isa_mix.hottest_loop(ins, asm)
    lo, hi = spans[0] if spans else (0, len(ins) - 1)
    cls = [isa_mix.classify(op, text) for _, op, text in ins[lo:hi + 1]]
    valu = [c for c in cls if c.startswith("valu")]
    return sum("half" in c or "packed" in c for c in valu) / max(len(valu), 1)


print(f"# {tag}: per launch.  busy = VALU pipe cycles of the issued instructions / the kernel's own cycles (GRBM_GUI_ACTIVE / 8); see the docstring of scripts/valu_busy.py")
print("# The class costs carry the calibration's spread (+-3 %) and the half-rate share is a static one: read 0.95-1.05 as 'issue-bound'.")
print(f"# {'kernel':48s} {'Minstr':>8s} {'quarter M':>9s} {'half share':>10s} {'kcycles':>9s} {'clock GHz':>9s} {'busy':>6s}")
rows = []
for k, v in pmc.items():
    if "GRBM_GUI_ACTIVE" not in v or "SQ_INSTS_VALU" not in v:
        continue
    inst, act = v["SQ_INSTS_VALU"], v.get("SQ_ACTIVE_INST_VALU", v["SQ_INSTS_VALU"])
    q = max(act - inst, 0)
    h = half_share(KERNELS[k]) if k in KERNELS else None
    hh = 0.2 if h is None else h
    cyc = v["GRBM_GUI_ACTIVE"] / XCDS
    pipe = ((inst - q - hh * inst) * C_FULL + hh * inst * C_HALF + q * C_QUARTER) / SIMDS
    rows.append((cyc, k, inst / 1e6, q / 1e6, ("%.2f" % h) if h is not None else "(0.20)", cyc / 1e3, pipe / cyc))
dur = {}
try:
    import csv
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_stats.csv"))):
        dur[r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()] = float(r["AverageNs"])
except OSError:
    pass
for cyc, k, mi, mq, h, kc, busy in sorted(rows, reverse=True):
    ghz = ("%.2f" % (cyc / dur[k])) if (k in dur and dur[k] > 50e3) else "-"   # (GRBM_GUI_ACTIVE includes a launch's ramp: meaningless for 5-40 us kernels)
    print(f"{k:50s} {mi:8.2f} {mq:9.2f} {h:>10s} {kc:9.1f} {ghz:>9s} {busy:6.2f}")
