#!/usr/bin/env python3
"""Builds a second copy of libradegs_hip.so with extra compiler flags / defines for radegs_kernels.hip (same-box A/B runs:
`RADEGS_LIB=gpurun_ab/libradegs_<name>.so python bench.py ...`).  Usage: scripts/build_alt.py <name> [flags...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rade-gs_amd"))
import build as b  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
b.build(verbose=False)
out_dir = os.path.join(ROOT, "gpurun_ab")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, name + ".o")
subprocess.check_call(["hipcc"] + b.FLAGS + extra + ["-c", os.path.join(b.CSRC, "radegs_kernels.hip"), "-o", obj])
objs = [obj] + [os.path.join(b.OBJ_DIR, u + ".o") for u in b.UNITS if u not in ("radegs_prims", "radegs_kernels")]
lib = os.path.join(out_dir, "libradegs_%s.so" % name)
subprocess.check_call(["hipcc", "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
