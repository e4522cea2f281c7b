"""Builds libradegs_hip.so (the C-ABI HIP library, include/radegs.h) for gfx950, in-tree.

    python rade-gs_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Flags that matter:
  -ffp-contract=off        one rounding per fp32 op; fma only where the source spells fmaf().  This is
                           what makes radii / tile rects / depth keys / blend decisions bit-identical to
                           the CPU oracle (see csrc/rg_math.h, csrc/rg_blend.h).
  -munsafe-fp-atomics      global_atomic_add_f32 in hardware (no CAS loop) for the gradient accumulators.
  -fno-slp-vectorize       keeps clang from fusing the two pixels' (or any two independent) fp32 chains into v_pk_* instructions:
                           packed fp32 issues as two passes on gfx950 (no throughput gain, scripts/ubench/valu_rates.hip) and the
                           fused chains lose the instruction-level parallelism of two independent ones -- measured on C2: forward
                           blend 0.36 -> 0.32 ms, backward 0.60 -> 0.57, preprocess_bwd 0.21 -> 0.18.
  (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt stays on: IEEE divide/sqrt.)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "diff_gaussian_rasterization")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libradegs_hip.so")
CHECK_LIB = os.path.join(OUT_DIR, "libradegs_prims_check.so")   # test-only: rocPRIM cross-check of the hand-written sorts
ARCH = "gfx950"   # the only target: kernels use gfx950's 160 KB LDS (radegs_sort.hip's 32-item scatter needs 70 KB per workgroup), its DPP /
                  # bank-mask forms and wave64 tilings -- changing this does not give a working gfx90a / gfx942 library
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fno-slp-vectorize", "-Wno-unused-value", "-I", CSRC]
UNITS = {
    "radegs_prims": ["radegs_prims.hip"],
    "radegs_sort": ["radegs_sort.hip", "rg_prims.h"],
    "radegs_kernels": ["radegs_kernels.hip", "rg_launch.inc", "rg_streams.inc", "rg_math.h", "rg_blend.h", "rg_preprocess.h", "rg_preprocess_bwd.h",
                       "rg_layout.h", "rg_prims.h", os.path.join("..", "..", "include", "radegs.h")],
    "radegs_normals": ["radegs_normals.hip", os.path.join("..", "..", "include", "radegs.h")],
    "radegs_filter3d": ["radegs_filter3d.hip", os.path.join("..", "..", "include", "radegs.h")],
    "radegs_photometric": ["radegs_photometric.hip", os.path.join("..", "..", "include", "radegs.h")],
    "radegs_adam": ["radegs_adam.hip", os.path.join("..", "..", "include", "radegs.h")],
    "radegs_knn": ["radegs_knn.hip", "rg_prims.h", os.path.join("..", "..", "include", "radegs.h")],
}
# Units outside the rasterizer's decision chain have no bit-exactness contract with the oracle: let them contract to fma.
UNIT_FLAGS = {"radegs_normals": ["-ffp-contract=fast"], "radegs_filter3d": ["-ffp-contract=fast"],
              "radegs_photometric": ["-ffp-contract=fast"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    for name, files in UNITS.items():
        deps = [os.path.join(CSRC, f) for f in files] + [os.path.abspath(__file__)]
        obj = os.path.join(OBJ_DIR, name + ".o")
        if force or _stale(obj, deps):
            cmd = [hipcc] + FLAGS + UNIT_FLAGS.get(name, []) + ["-c", os.path.join(CSRC, files[0]), "-o", obj]
            if verbose:
                print("[radegs build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        if name == "radegs_prims":   # never linked into the product: its own shared object, loaded on demand (RADEGS_PRIMS=rocprim)
            if force or _stale(CHECK_LIB, [obj]):
                cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", CHECK_LIB, obj]
                if verbose:
                    print("[radegs build]", " ".join(cmd), flush=True)
                subprocess.check_call(cmd)
            continue
        objs.append(obj)
    if force or _stale(LIB, objs + [os.path.abspath(__file__)]):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[radegs build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


TORCH_BINDING_SRC = os.path.join(CSRC, "torch_binding", "radegs_torch_binding.cpp")
TORCH_BINDING_NAME = "_C_torch"


def torch_binding_path():
    import sysconfig
    return os.path.join(OUT_DIR, TORCH_BINDING_NAME + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_torch_binding(force=False, verbose=True):
    """diff_gaussian_rasterization/_C_torch*.so: the reference's pybind `_C` module (DGR/ext.cpp:15-19 -- rasterize_gaussians,
    rasterize_gaussians_backward, mark_visible, integrate_gaussians_to_points with torch::Tensor arguments) as HOST C++ over the C ABI
    of libradegs_hip.so.  No device code in it: g++ against the torch / pybind11 / HIP runtime headers, linked to the library next to it
    ($ORIGIN rpath).  `RADEGS_BINDING=torch` makes the operator package use it instead of the ctypes binding."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    lib = build(force=False, verbose=verbose)
    out = torch_binding_path()
    header = os.path.join(HERE, "..", "include", "radegs.h")
    if not (force or _stale(out, [TORCH_BINDING_SRC, header, lib, os.path.abspath(__file__)])):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ce.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"], os.path.join(HERE, "..", "include")]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=" + TORCH_BINDING_NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for i in inc:
        cmd += ["-isystem", i]
    cmd += [TORCH_BINDING_SRC, "-o", out, "-L" + tlib, "-L" + OUT_DIR, "-L" + os.path.join(rocm, "lib"), "-lradegs_hip", "-lc10", "-lc10_hip", "-ltorch_cpu",
            "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib, "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print("[radegs build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--torch-binding" in sys.argv:
        print(build_torch_binding(force="--force" in sys.argv))
