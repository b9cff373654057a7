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
    subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, _lib.SRC], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z15orr_step_kernelILi0E.*:", l))
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
    # largest backward branch
    best = (0, 0, 0)
    for i, t in enumerate(insts):
        m = re.match(r"^s_c?branch\S*\s+(\S+)$", t)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    for name, seg in (("whole kernel", insts), ("largest loop (sub-steps)", insts[best[1]:best[2] + 1])):
        c = collections.Counter(classify(t.split()[0]) for t in seg)
        tot = sum(c.values())
        print("%s: %d instructions" % (name, tot))
        for k, v in c.most_common():
            print("   %-18s %6d  %5.1f%%" % (k, v, 100.0 * v / tot))
    for l in lines:
        if "orr_step_kernelILi0E" in l and ("NumVgprs" in l or "spill" in l):
            print(l)
    meta = "\n".join(lines)
    m = re.search(r"\.name:\s+_Z15orr_step_kernelILi0E.*?\.vgpr_spill_count:\s+\d+", meta, re.S)
    if m:
        print(re.sub(r"\s+", " ", " ".join(x for x in m.group(0).split("\n") if "count" in x or "lds" in x or "group_segment" in x)))


if __name__ == "__main__":
    main()
