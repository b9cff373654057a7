#!/usr/bin/env python3
"""Static instruction counts of the step kernel per phase (development aid): compiles the -DORR_PHASE_TIMERS build to
assembly and counts the instructions between consecutive phase-timer reads (s_memtime)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402

flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + ["-DORR_PHASE_TIMERS"] + sys.argv[1:]
out = os.path.join(tempfile.mkdtemp(), "step.s")
subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, _lib.SRC], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z15orr_step_kernelILi0E.*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.split(";")[0].strip() for l in lines[start:end]]
body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
segs, cur = [], []
for l in body:
    if l.startswith("s_memtime") or l.startswith("s_memrealtime"):
        segs.append(cur)
        cur = []
    else:
        cur.append(l)
segs.append(cur)
for i, s in enumerate(segs):
    if len(s) < 40:
        continue
    c = collections.Counter()
    for l in s:
        m = l.split()[0]
        if m.startswith("v_cndmask") or m.startswith("v_mov_b32_e"):
            c["mov/sel"] += 1
        elif "dpp" in m:
            c["dpp"] += 1
        elif m.startswith("v_accvgpr"):
            c["agpr"] += 1
        elif m.startswith("v_readlane") or m.startswith("v_writelane"):
            c["lane"] += 1
        elif m.startswith("v_"):
            c["valu"] += 1
        elif m.startswith("ds_"):
            c["lds"] += 1
        elif m.startswith("s_waitcnt"):
            c["wait"] += 1
        elif m.startswith("s_nop"):
            c["nop"] += 1
        elif m.startswith("s_"):
            c["salu"] += 1
        else:
            c["other"] += 1
    print("segment %2d: %5d instructions  %s" % (i, len(s), dict(c)))
