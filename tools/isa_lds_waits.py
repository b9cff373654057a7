#!/usr/bin/env python3
"""Static estimate of exposed LDS latency in the step kernel (development aid): for every s_waitcnt lgkmcnt(n) between the phase-timer
reads of the -DORR_PHASE_TIMERS build, the number of instructions issued since the LDS / scalar-memory operation it waits for.  A lone
wave per SIMD cannot hide that latency (~100+ cycles for a ds_read at ~4 cycles per issued instruction), so short distances are stalls.

usage: tools/isa_lds_waits.py [latency_cycles]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402

LAT = float(sys.argv[1]) if len(sys.argv) > 1 else 110.0
flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + ["-DORR_PHASE_TIMERS"]
out = os.path.join(tempfile.mkdtemp(), "step.s")
subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", out, _lib.SRC], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z15orr_step_kernelILi0E.*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.split(";")[0].strip() for l in lines[start:end]]
body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
# timeline: every instruction takes 4 cycles to issue; an LDS / scalar-memory operation completes LAT cycles after its issue (and
# not before the one in front of it: in-order return); a wait moves the clock to the completion of the operation it needs
seg, t, queue, segs, ninst = 0, 0.0, [], {}, {}
for l in body:
    m = l.split()[0]
    if m.startswith("s_memtime") or m.startswith("s_memrealtime"):
        seg += 1
        queue.append(t + LAT)
        continue
    t += 4.0
    ninst[seg] = ninst.get(seg, 0) + 1
    if m.startswith("ds_") or m.startswith("s_load") or m.startswith("flat_"):
        queue.append(max(t + LAT, queue[-1] + 4.0 if queue else 0.0))
    elif m.startswith("s_waitcnt") and "lgkmcnt" in l:
        n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
        if len(queue) > n:
            done = queue[len(queue) - n - 1]
            stall = max(0.0, done - t)
            t += stall
            s = segs.setdefault(seg, [0, 0.0, []])
            s[0] += 1; s[1] += stall; s[2].append(int(stall))
            queue = queue[len(queue) - n:]
for k in sorted(segs):
    n, st, d = segs[k]
    if ninst.get(k, 0) >= 40:
        print("segment %2d (%4d instructions): %3d waits, estimated exposed latency %6.0f cycles, stalls %s" % (k, ninst[k], n, st, [x for x in d if x > 0][:24]))
