#!/usr/bin/env python3
"""Branches inside the sub-step loop of the step kernel (development aid): a lone wave per SIMD pays 13-30 ticks per branch
(profiles/r02_issue_costs.txt), so every divergent `if` and every uniform skip in the loop is listed with the number of instructions
it jumps over and the first instructions of the guarded block.

usage: tools/isa_branches.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402

flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + sys.argv[1:]
out = os.path.join(tempfile.mkdtemp(), "step.s")
subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, _lib.SRC], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z15orr_step_kernelILi0E.*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
labels = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m:
        labels[m.group(1)] = i
best = (0, 0)
for i in range(start, end):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i and best[1] - best[0] < i - labels[m.group(1)] < 4500:
        best = (labels[m.group(1)], i)


def code(k):
    t = lines[k].split(";")[0].strip()
    return t if t and not t.startswith(".") and not t.endswith(":") else ""


n = 0
for i in range(best[0], best[1] + 1):
    t = code(i)
    m = re.match(r"(s_c?branch\w*)\s+(\.LBB\d+_\d+)", t)
    if not m:
        continue
    tgt = labels.get(m.group(2), 0)
    skipped = sum(1 for k in range(i + 1, tgt) if code(k)) if tgt > i else -1
    nxt = [code(k) for k in range(i + 1, i + 12) if code(k)][:3]
    prev = [code(k) for k in range(i - 3, i) if code(k)][-1:]
    n += 1
    print("%5d %-18s over %4d | after: %s | then: %s" % (i - best[0], m.group(1), skipped, "; ".join(prev), " ; ".join(nxt)))
print("%d branches in the loop (%d instructions)" % (n, sum(1 for k in range(best[0], best[1] + 1) if code(k))))
