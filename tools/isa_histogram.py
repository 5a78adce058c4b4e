#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop, from the compiler's own assembly (no GPU needed).

    python tools/isa_histogram.py --part 5 --kernel 'align_fill_affine_tag_kernelILi16ELi10ELi1ELb1'

compiles versalignlib_amd/csrc/kernel_part.hip (-DVALIGN_PART=n), engine_long.hip (--part main: band / long-read kernels) or
engine_align.hip (--part align: strip / fused / traceback kernels) for gfx950 with
--cuda-device-only -S, finds the largest innermost loop of the first kernel whose mangled name contains --kernel, and
counts its instructions by issue class.  Classes follow the measured rates of profiles/r02_valu_rate_microbench.txt and
r02_valu_rate_bitops.txt: "full" (2.3-2.7 cycles per wave64 instruction per SIMD from two waves per SIMD up) and
"half" (4.1-4.5: every VOP3P packed instruction, maxima wider than 16 bits, v_perm, v_mad_u24, shifts-with-or, DPP)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "versalignlib_amd", "csrc")

FULL = ("v_add_f32", "v_fma_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_bitop3_b32",
        "v_max_u16", "v_max_i16", "v_add_u16", "v_sub_u16", "v_mov_b32", "v_cndmask_b32", "v_ashrrev_i32", "v_lshrrev_b32",
        "v_add_co_u32", "v_addc_co_u32", "v_cmp", "v_not_b32", "v_add_nc_u32", "v_sub_nc_u32")


def classify(op):
    if op.startswith("v_pk_") or op.startswith("v_max3") or op.startswith("v_min3") or op.startswith("v_med3"):
        return "half"
    if op in ("v_perm_b32", "v_mad_u32_u24", "v_bfe_u32", "v_bfe_i32", "v_lshl_add_u32", "v_add3_u32", "v_and_or_b32", "v_bfi_b32",
              "v_lshl_or_b32", "v_or3_b32", "v_lshlrev_b32", "v_max_i32", "v_max_u32", "v_min_i32", "v_min_u32", "v_max_f32",
              "v_mad_i32_i16", "v_mad_u64_u32", "v_mul_lo_u32", "v_alignbit_b32"):
        return "half"
    if "_dpp" in op or "_sdwa" in op:
        return "half"
    if any(op.startswith(f) for f in FULL):
        return "full"
    return "other valu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", default="5")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--asm", default="", help="an existing .s file instead of compiling")
    a = ap.parse_args()
    asm = a.asm
    if not asm:
        asm = os.path.join(tempfile.gettempdir(), "valign_part_%s.s" % a.part)
        src = "engine_long.hip" if a.part == "main" else ("engine_align.hip" if a.part == "align" else "kernel_part.hip")
        cmd = ["hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC, os.path.join(CSRC, src), "-o", asm]
        if a.part not in ("main", "align"):
            cmd.insert(5, "-DVALIGN_PART=%s" % a.part)
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(a.kernel), ln))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    print(body[0].rstrip(":"))
    labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
    loops = []
    for i, ln in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    lo, hi = max(inner, key=lambda lp: lp[1] - lp[0])
    ops = [ln.split()[0] for ln in body[lo:hi + 1] if ln.startswith("\t") and not ln.startswith("\t.") and not ln.strip().startswith(";")]
    counts = collections.Counter()
    detail = collections.Counter()
    for op in ops:
        if op.startswith("v_"):
            c = classify(op)
            counts[c] += 1
            detail[(c, re.sub(r"_e(32|64)$", "", op))] += 1
        elif op.startswith("ds_"):
            counts["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            counts["vmem"] += 1
        elif op.startswith("s_"):
            counts["salu / control"] += 1
        else:
            counts[op] += 1
    print("hot loop: %d instructions (lines %d-%d of the kernel)" % (len(ops), lo, hi))
    for k in ("half", "full", "other valu", "lds", "vmem", "salu / control"):
        print("  %-16s %4d" % (k, counts.get(k, 0)))
    valu = counts["half"] + counts["full"] + counts["other valu"]
    print("  VALU total       %4d   issue cycles per wave at 4.3 (half) / 2.6 (full, other): %.0f" %
          (valu, 4.3 * counts["half"] + 2.6 * (counts["full"] + counts["other valu"])))
    for (c, op), k in sorted(detail.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print("    %-10s %-28s %3d" % (c, op, k))


if __name__ == "__main__":
    main()
