#!/usr/bin/env python3
"""Static instruction statistics of orr_step_kernel<0> from the compiler's assembly (development aid).

usage: python tools/isa_stats.py [extra hipcc flags...]
Compiles csrc/orr_kernels.hip to gfx950 assembly with the product's flags, finds the step kernel, and prints instruction
counts by class for the whole kernel and for its hottest region (the sub-step loop = the largest backward-branch body)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402


def classify(m):
    if m.startswith("v_pk_"):
        return "valu_pk"
    if m.startswith("v_mfma"):
        return "mfma"
    if m.startswith("v_accvgpr"):
        return "agpr_move"
    if m.endswith("_dpp") or "dpp" in m:
        return "valu_dpp"
    if m.startswith("v_mov") or m.startswith("v_cndmask") or m.startswith("v_readlane") or m.startswith("v_writelane") or m.startswith("v_readfirstlane"):
        return "valu_move/select"
    if m.startswith("v_"):
        return "valu_math"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_") or m.startswith("scratch_"):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "s_waitcnt"
    if m.startswith("s_nop"):
        return "s_nop"
    if m.startswith("s_cbranch") or m.startswith("s_branch"):
        return "branch"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + sys.argv[1:]
    out = os.path.join(tempfile.mkdtemp(), "step.s")
    src = os.environ.get("ORR_ISA_SRC", _lib.SRC)      # another tree's orr_kernels.hip (A/B of code generation)
    subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    src_w2 = os.path.join(os.path.dirname(src), "orr_kernels_w2.hip")
    if os.path.exists(src_w2):       # the two-waves-per-SIMD variant is its own translation unit with its own flags
        flags_w2 = [f for f in _lib.HIPCC_FLAGS_W2 if f not in ("-shared", "-fPIC")] + sys.argv[1:]
        subprocess.check_call([_lib.HIPCC] + flags_w2 + ["-S", "--cuda-device-only", "-o", out + "2", src_w2], stderr=subprocess.DEVNULL)
        lines += open(out + "2").read().split("\n")
    src_an = os.path.join(os.path.dirname(src), "orr_kernels_anchor.hip")
    if os.path.exists(src_an):       # the friction-anchor variants: third translation unit, the main unit's flags
        subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out + "3", src_an], stderr=subprocess.DEVNULL)
        lines += open(out + "3").read().split("\n")
    meta = "\n".join(lines)
    # one report per variant of the step kernel (WPE 1: one wave per SIMD, WPE 2: two; see orr_kernels.hip)
    for sym, title in (("_Z15orr_step_kernelILi0ELi1ELb0E", "step kernel, one wave per SIMD"), ("_Z15orr_step_kernelILi0ELi2ELb0E", "step kernel, two waves per SIMD"),
                       ("_Z15orr_step_kernelILi0ELi1ELb1E", "step kernel with friction anchors (one wave per SIMD)")):
        try:
            start = next(i for i, l in enumerate(lines) if re.match(r"^%s.*:" % sym, l))
        except StopIteration:
            continue
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        body = lines[start:end + 1]
        labels = {}
        insts = []
        for l in body:
            t = l.split(";")[0].strip()
            if not t:
                continue
            m = re.match(r"^(\.?[A-Za-z_0-9$]+):$", t)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            if t.startswith("."):
                continue
            insts.append(t)
        # the sub-step loop = the backward branch whose body holds the most DPP instructions (the Gauss-Seidel sweeps live there); the
        # largest backward branch alone can be some other loop of the step-end / reset code
        best = (0, 0, 0)
        dpp_prefix = [0]
        for t in insts:
            dpp_prefix.append(dpp_prefix[-1] + ("dpp" in t.split()[0]))
        best_key = (-1, -1)
        for i, t in enumerate(insts):
            m = re.match(r"^s_c?branch\S*\s+(\S+)$", t)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo = labels[m.group(1)]
                key = (dpp_prefix[i + 1] - dpp_prefix[lo], i - lo)
                if key > best_key:
                    best_key, best = key, (i - lo, lo, i)
        print("==== %s" % title)
        for name, seg in (("whole kernel", insts), ("largest loop (sub-steps)", insts[best[1]:best[2] + 1])):
            c = collections.Counter(classify(t.split()[0]) for t in seg)
            tot = sum(c.values())
            scratch = sum(1 for t in seg if t.startswith("scratch_") or t.startswith("buffer_") and "offen" in t)
            print("%s: %d instructions (scratch accesses: %d)" % (name, tot, scratch))
            for k, v in c.most_common():
                print("   %-18s %6d  %5.1f%%" % (k, v, 100.0 * v / tot))
        m = re.search(r"\.group_segment_fixed_size:\s+(\d+)(?:(?!\.group_segment_fixed_size).)*?\.name:\s+%s.*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)" % sym, meta, re.S)
        if m:
            print(" LDS %s B, scratch %s B per lane, sgpr %s (spilled %s), vgpr %s (spilled %s)" % m.groups())


if __name__ == "__main__":
    main()
