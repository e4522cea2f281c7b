"""Static resource budget of the hot kernels, read from the code objects inside the in-tree libradegs_hip.so (no GPU needed).

Every kernel on the path is bound by VALU issue or by memory latency at a given number of resident waves, so the
numbers the compiler ends up with ARE performance properties: a few VGPRs more drop a blend kernel from 6 to 5 waves per SIMD, a
spilled register turns into scratch traffic, and preprocess_bwd_kernel is only fast while all of a block's reads are issued before
its first wait (round 4).  A compiler or source change that moves one of them should fail here, on the CPU tier, not show up as an
unexplained 5 % on the next GPU run.  Budgets are the measured configuration's occupancy steps, not the exact counts.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("clang-offload-bundler", "llvm-readelf", "llvm-objdump")]
pytestmark = pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS) or shutil.which("objcopy") is None,
                                reason="ROCm LLVM binary tools not found")


@pytest.fixture(scope="module")
def code_objects():
    """{kernel symbol: (resources dict, code object path)} for every gfx950 kernel in the library"""
    import diff_gaussian_rasterization._C as C
    lib = C.library_path() if hasattr(C, "library_path") else os.path.join(os.path.dirname(C.__file__), "libradegs_hip.so")
    tmp = tempfile.mkdtemp(prefix="radegs_co_")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]   # one bundle per translation unit
    assert offs, "no offload bundle in " + lib
    kernels = {}
    for n, o in enumerate(offs):
        end = offs[n + 1] if n + 1 < len(offs) else len(data)
        b, co = os.path.join(tmp, f"b{n}.bin"), os.path.join(tmp, f"b{n}.co")
        open(b, "wb").write(data[o:end])
        subprocess.check_call([TOOLS[0], "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co, "--unbundle"])
        notes = subprocess.check_output([TOOLS[1], "--notes", co]).decode()
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            kernels[name] = (dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                                  lds=g("group_segment_fixed_size")), co)
    yield kernels
    shutil.rmtree(tmp, ignore_errors=True)


def waves_per_simd(vgpr):
    """gfx950: 512 VGPRs per SIMD lane, allocated in granules of 8, at most 8 waves"""
    return min(8, 512 // (8 * ((vgpr + 7) // 8)))


def find(kernels, *parts):
    hits = [k for k in kernels if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, hits)
    return kernels[hits[0]][0]


# (substrings of the mangled name) -> (minimum waves per SIMD, maximum scratch bytes, maximum static LDS bytes)
BUDGET = [
    (("blend_fwd_streams_kernelILb0ELb1EE",), 7, 0, 4608),      # C2 / C3 forward blend (depth mode)
    (("blend_fwd_streams_kernelILb0ELb0EE",), 7, 0, 4608),
    (("blend_fwd_streams_kernelILb1ELb1EE",), 4, 0, 8192),      # coord-map modes: no scratch since the camera-plane record is loaded unconditionally
    (("blend_fwd_streams_kernelILb1ELb0EE",), 5, 0, 8192),      # and zeroed by selects (round 6; rounds 3-5: 4 waves + 32 B of scratch in both)
    # the stream backward: staged records + ids + positions + 2.3 KB of row-reduction scratch; 20 waves per CU (5 per SIMD) x 6 976 B = 139 KB
    (("blend_bwd_streams_kernelILb0ELb1EE",), 5, 0, 6976),      # the dominant kernel (5 waves: measured faster than 6 with a spill)
    (("blend_bwd_streams_kernelILb0ELb0EE",), 5, 0, 6976),
    (("blend_bwd_streams_kernelILb1ELb1EE",), 4, 0, 10240),     # coord-map modes: two 64-byte lines per (block, entry), one atomic instruction each
    (("blend_bwd_streams_kernelILb1ELb0EE",), 4, 0, 10240),
    (("preprocess_fwd_kernelILb0E",), 5, 0, 0),                      # 5 since round 6: all 48 SH coefficients in flight at once (one round trip
                                                                     # instead of four; same-box A/B against two batches at 6 waves: 0.094 | 0.096 ms)
    (("preprocess_bwd_kernel",), 3, 16, 0),                          # dynamic LDS: the SH slab
    (("block_lists_kernelILb0E",), 8, 0, 4352),                      # 4 KB of it: the block-balancing sort its first workgroups run (round 6)
    (("block_counts_kernelILb0E",), 8, 0, 512),
    (("emit_instances_kernelILb1E",), 6, 0, 0),
    (("tile_ranges_kernelItE",), 8, 0, 0),
    (("integrate_kernel",), 5, 0, 16384),
]


@pytest.mark.parametrize("parts,min_waves,max_scratch,max_lds", BUDGET, ids=[b[0][0] for b in BUDGET])
def test_hot_kernel_keeps_its_occupancy_step(code_objects, parts, min_waves, max_scratch, max_lds):
    r = find(code_objects, *parts)
    assert waves_per_simd(r["vgpr"]) >= min_waves, r
    assert r["scratch"] <= max_scratch, r
    assert r["lds"] <= max_lds, r


def test_no_default_blend_or_binning_kernel_spills(code_objects):
    """every blend and binning kernel of the library but the coord-map stream forward (a few bytes once per round, see BUDGET)"""
    default = ("blend_fwd_kernel", "blend_bwd_packed_kernel", "blend_fwd_streams_kernelILb0", "blend_bwd_streams_kernel",
               "scatter_kernel", "digit_histogram", "scan_rows", "gather_", "block_lists", "block_counts", "balance_blocks",
               "emit_instances", "tile_ranges", "preprocess_fwd")
    spilling = {k: v[0]["scratch"] for k, v in code_objects.items() if v[0]["scratch"] and any(s in k for s in default)}
    assert not spilling, spilling


def test_preprocess_bwd_issues_all_reads_before_it_waits(code_objects):
    """the slab's twelve 16-byte loads and the Gaussian's own records are requested before the first barrier; the old
    form (4-byte loads in a loop, eight in flight) ran at 3.7 TB/s for that reason alone."""
    name = [k for k in code_objects if "preprocess_bwd_kernel" in k][0]
    asm = subprocess.check_output([TOOLS[2], "-d", "--disassemble-symbols=" + name, code_objects[name][1]]).decode()
    ins = [l.split("//")[0].strip() for l in asm.splitlines() if re.match(r"^\s+(s_|v_|ds_|global_|scratch_|buffer_|flat_)", l)]
    first_barrier = next(i for i, l in enumerate(ins) if l.startswith("s_barrier"))
    head = ins[:first_barrier]
    wide = sum(l.startswith("global_load_dwordx4") for l in head)
    assert wide >= 12 + 4, (wide, "16-byte loads before the first barrier: 12 slab pieces + the 4 of the accumulator record")
    # ... and they are not inside a loop: no backward branch before the barrier in the vector path (the word-loop fallback has one,
    # so count the loads that precede the first backward branch instead of forbidding it)
    assert sum(l.startswith("global_load_dwordx4") for l in ins[first_barrier:]) <= 4   # cov3D_precomp / intended-mode reads only
