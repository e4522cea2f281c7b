"""TEST INFRASTRUCTURE: builds oracle/_ref/ -- the reference's OWN rasterizer sources compiled for the host.

    python oracle/build_ref.py [--force]

What is compiled: forward.cu, backward.cu, rasterizer_impl.cu (with auxiliary.h, config.h, forward.h, backward.h, rasterizer.h,
rasterizer_impl.h) read from /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer WHERE THEY LIE.  Nothing is
copied into the repository: each .cu file is piped through ONE textual substitution into g++'s stdin -- CUDA's launch tokens
`k <<<g, b>>> (args)` become `k % cuda_on_host::cfg(g, b) (args)`, because no macro can make `<<<` parse -- and only object code
lands in oracle/_ref/ (git-ignored, not gpurun-ignored: the .so travels to the GPU box like the product's own).  What stands in
for nvcc, the CUDA runtime, CUB and the un-vendored glm submodule is oracle/ref_shim/ (this repository's code; see the headers
there).  The reference's own build system (setup.py / CMakeLists.txt, both of which need nvcc and torch's C++ headers) is not run.

Two libraries:
  libradegs_ref.so       -ffp-contract=off : one rounding per operation, comparable bit for bit with the oracle
  libradegs_ref_fma.so   -ffp-contract=fast -mfma : mul+add pairs contracted like nvcc's default (-fmad=true) would -- the pairs gcc
                         picks are not necessarily nvcc's, so this is a sensitivity probe, not a CUDA emulation.

If /root/reference is absent (the GPU box) this does nothing: tests that need the library skip, and the golden vectors it
produced (tests/golden/ref_*.npz, written by tests/golden/make_golden_ref.py) are what is compared there.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer"
SHIM = os.path.join(HERE, "ref_shim")
OUT = os.path.join(HERE, "_ref")
UNITS = ("forward.cu", "backward.cu", "rasterizer_impl.cu")
COMMON = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-fopenmp", "-fno-fast-math", "-w", "-Wno-narrowing", "-I", SHIM, "-I", REF]
VARIANTS = {"libradegs_ref.so": ["-ffp-contract=off"], "libradegs_ref_fma.so": ["-ffp-contract=fast", "-mfma"]}

_OPEN = re.compile(r"<<\s*<")
_CLOSE = re.compile(r">>\s*>")


def launch_syntax_to_cxx(text):
    """`k <<<g, b>>> (args)`  ->  `k % cuda_on_host::cfg(g, b) (args)`.  The reference uses `<<<`/`>>>` (also spelled `<< <`, `>> >`) for
    kernel launches only (checked: no shift-then-compare or nested template closers match in the three files)."""
    return _CLOSE.sub(")", _OPEN.sub("% cuda_on_host::cfg(", text))


def available():
    return all(os.path.exists(os.path.join(REF, u)) for u in UNITS)


def _deps():
    d = [os.path.abspath(__file__)]
    for root, _, files in os.walk(SHIM):
        d += [os.path.join(root, f) for f in files]
    d += [os.path.join(REF, f) for f in os.listdir(REF)]
    return d


def build(force=False, verbose=False):
    """Returns the path of libradegs_ref.so, or None when the reference sources are not on this machine."""
    if not available():
        return None
    os.makedirs(OUT, exist_ok=True)
    cxx = os.environ.get("CXX", "g++")
    newest = max(os.path.getmtime(p) for p in _deps())
    for lib, flags in VARIANTS.items():
        so = os.path.join(OUT, lib)
        if not force and os.path.exists(so) and os.path.getmtime(so) >= newest:
            continue
        tag = lib.split(".")[0]
        objs = []
        for u in UNITS:
            with open(os.path.join(REF, u)) as f:
                src = '#line 1 "%s"\n' % os.path.join(REF, u) + launch_syntax_to_cxx(f.read())
            obj = os.path.join(OUT, "%s_%s.o" % (tag, u.replace(".cu", "")))
            cmd = [cxx] + COMMON + flags + ["-c", "-o", obj, "-"]
            if verbose:
                print("[ref build]", " ".join(cmd), "<", u, flush=True)
            subprocess.run(cmd, input=src.encode(), check=True)
            objs.append(obj)
        obj = os.path.join(OUT, "%s_api.o" % tag)
        subprocess.check_call([cxx] + COMMON + flags + ["-c", "-o", obj, os.path.join(SHIM, "ref_api.cpp")])
        objs.append(obj)
        subprocess.check_call([cxx, "-shared", "-fopenmp", "-o", so] + objs)
        for o in objs:
            os.remove(o)
    return os.path.join(OUT, "libradegs_ref.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
