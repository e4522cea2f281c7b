"""Static ISA mix of a kernel of the in-tree library (no GPU needed).

    python scripts/isa_mix.py blend_bwd_streams_kernelILb0ELb1EE [--dump out.s] [--loop]

Unbundles the gfx950 code objects of libradegs_hip.so, disassembles the kernel whose mangled name contains the given substring(s) and
prints its instruction histogram by issue class (full-rate VALU, half-rate VALU, quarter-rate / transcendental, LDS, VMEM, SALU, ...)
-- the classes of scripts/ubench/valu_rates.hip.  With --loop only the hottest loop is counted: the innermost backward
branch target .. branch span that holds the most VALU instructions.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RADEGS_LIB") or os.path.join(ROOT, "rade-gs_amd", "diff_gaussian_rasterization", "libradegs_hip.so")

QUARTER = ("v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_permlane", "v_rcp_iflag", "v_div_fmas",
           "v_div_scale", "v_div_fixup", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32")
FULL = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_fmaak_f32", "v_fmamk_f32",
        "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_accvgpr")


def classify(op, text):
    if op.startswith("v_"):
        if "dpp" in text or "row_" in text or "quad_perm" in text:
            return "valu_dpp(half)"
        if any(op.startswith(q) for q in QUARTER):
            return "valu_quarter"
        if op.startswith("v_cndmask_b32"):
            return "valu_cndmask_e64(half)" if ("_e64" in op or text.count("s[") or "vcc" not in text) else "valu_full"
        if op.startswith("v_pk_"):
            return "valu_packed(2 passes)"
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            return "mfma"
        if any(op == f or op.startswith(f + "_e32") or op.startswith(f + "_e64") for f in FULL):
            return "valu_full"
        return "valu_other(half)"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
        return "vmem_atomic"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait/nop/barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def code_objects():
    tmp = tempfile.mkdtemp(prefix="radegs_isa_")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    out = []
    for n, o in enumerate(offs):
        end = offs[n + 1] if n + 1 < len(offs) else len(data)
        b, co = os.path.join(tmp, f"b{n}.bin"), os.path.join(tmp, f"b{n}.co")
        open(b, "wb").write(data[o:end])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + b, "--output=" + co, "--unbundle"])
        out.append(co)
    return out


def disassemble(parts):
    for co in code_objects():
        syms = subprocess.check_output(["nm", co]).decode().split("\n")
        names = [l.split()[-1] for l in syms if l.strip() and l.split()[-2] in "Tt" and all(p in l for p in parts)]
        names = [n for n in names if not n.endswith(".kd")]
        if names:
            assert len(names) == 1, names
            asm = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + names[0], co]).decode()
            return names[0], asm
    raise SystemExit("no kernel matches " + repr(parts))


def parse(asm):
    """-> [(address, op, text)]"""
    ins = []
    for l in asm.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(1) + " " + m.group(2)))
    return ins


def hottest_loop(ins, asm):
    """innermost span [target, branch] of a backward branch holding the most VALU instructions and no other backward branch inside"""
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    labels = {}
    for l in asm.splitlines():
        m = re.match(r"^([0-9a-fA-F]+) <(\S+)>:", l)
        if m:
            labels[m.group(2)] = int(m.group(1), 16)
    spans = []
    for i, (a, op, text) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"<(\S+?)>", text) or re.search(r"(L\S+)", text)
            t = None
            m2 = re.search(r"//.*", text)
            # llvm-objdump prints "s_cbranch_scc1 65314" (a signed 16-bit word offset) -- compute the target from it
            mo = re.search(r"\s(-?\d+)\s*$", text)
            if mo:
                off = int(mo.group(1))
                if off >= 32768:
                    off -= 65536
                t = a + 4 + 4 * off
            if t is not None and t <= a and t in addr_idx:
                spans.append((addr_idx[t], i))
    inner = [s for s in spans if not any((o != s and o[0] >= s[0] and o[1] <= s[1]) for o in spans)]
    # the blend / sort loops read their operands from LDS: prefer spans that do, then the most VALU work
    def score(s):
        ops = [ins[k][1] for k in range(s[0], s[1] + 1)]
        return (any(o.startswith("ds_read") for o in ops), sum(o.startswith("v_") for o in ops))
    ranked = sorted(inner, key=score, reverse=True)
    return ranked


def main():
    argv = sys.argv[1:]
    dump = None
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    name, asm = disassemble(args)
    if dump:
        open(dump, "w").write(asm)
    ins = parse(asm)
    spans = [(0, len(ins) - 1)]
    if "--loop" in sys.argv:   # every innermost loop that reads LDS and holds at least 40 VALU instructions, largest first
        ranked = hottest_loop(ins, asm)
        spans = [sp for sp in ranked if sum(ins[k][1].startswith("v_") for k in range(sp[0], sp[1] + 1)) >= 40
                 and any(ins[k][1].startswith("ds_read") for k in range(sp[0], sp[1] + 1))] or ranked[:1]
    for span in spans:
        report(name, ins, span, "--loop" in sys.argv, "--ops" in sys.argv)


def report(name, ins, span, is_loop, show_ops):
    sel = ins[span[0]:span[1] + 1]
    cls = Counter(classify(op, text) for _, op, text in sel)
    ops = Counter(op for _, op, _ in sel)
    print(f"{name}: {len(sel)} instructions in {'loop at instruction %d' % span[0] if is_loop else 'kernel'} (of {len(ins)})")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v:5d}")
    # issue cycles per wave64 instruction measured on the part (scripts/ubench/valu_rates.hip)
    cyc = {"valu_full": 2.5, "valu_dpp(half)": 4.5, "valu_other(half)": 4.5, "valu_cndmask_e64(half)": 4.5, "valu_quarter": 8.0,
           "valu_packed(2 passes)": 5.0}
    print(f"  VALU issue-cycle estimate: {sum(cyc.get(k, 0.0) * v for k, v in cls.items()):.0f}")
    if show_ops:
        print("  -- by opcode --")
        for k, v in ops.most_common(40):
            print(f"  {k:28s} {v:5d}")


if __name__ == "__main__":
    main()
