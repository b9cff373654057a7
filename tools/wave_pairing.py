"""Do two waves that share a SIMD overlap?  (development aid, runs on the GPU box.)

usage:  python tools/wave_pairing.py [robots=8192] [launches=30] [waves_per_eu=2]
Builds the step kernel with amdgpu_waves_per_eu(2, 2) (<= 256 VGPRs) + the per-wave timeline (-DORR_PHASE_TIMERS records realtime start / end,
shader cycles and the HW_ID / XCC_ID registers of every wave), runs `robots` Laikago robots and classifies each wave of a launch by what
else ran on ITS SIMD (same XCC, SE, CU, SIMD) while it ran: alone the whole time, or overlapped by another wave for a fraction of
its life.  If co-resident waves overlapped perfectly a paired wave would take as long as a lone one; if VALU issue were the only
resource and already saturated by one wave, twice as long.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
robots = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
wpe = int(sys.argv[3]) if len(sys.argv) > 3 else 2
LIB = os.path.join(ROOT, "openroborl_amd", "libopenroborl_pairing_w%d.so" % wpe)
from openroborl_amd import _lib as _build  # noqa: E402
_build.build(out_path=LIB, extra_flags=["-DORR_PHASE_TIMERS", "-DORR_WAVES_PER_EU=%d" % wpe])
os.environ["ORR_LIB_PATH"] = LIB
os.environ["ORR_STEP_WAVES_PER_EU"] = "1"     # = the main translation unit's kernel, which this build compiled for `wpe` waves per SIMD + timers

import torch  # noqa: E402
from openroborl_amd import _lib  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=robots, seed=0)
env.reset()
g = torch.Generator().manual_seed(0)
act = (torch.randn(robots, 12, generator=g) * 0.1).to(env.device)
L = _lib.load()
L.orr_debug_wave_timeline.argtypes = [C.POINTER(C.c_longlong), C.c_int]
for _ in range(200):
    env.step(act)
W = min(robots // 4, 2048)
buf = (C.c_longlong * (4 * W))()
dur_alone, dur_paired, frac_all, launch_len, cyc_alone, cyc_paired, simd_load = [], [], [], [], [], [], []
for _ in range(launches):
    env.step(act)
    L.orr_debug_wave_timeline(buf, W)
    a = np.frombuffer(buf, dtype=np.int64).reshape(W, 4).copy()
    start, end, cyc = a[:, 0].astype(np.float64) / 100.0, a[:, 1].astype(np.float64) / 100.0, a[:, 2].astype(np.float64)
    hw = (a[:, 3] >> 8) & 0xFFFFFFFF
    xcc = (a[:, 3] >> 40) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    launch_len.append(end.max() - start.min())
    order = np.argsort(key, kind="stable")
    frac = np.zeros(W)
    bounds = np.flatnonzero(np.diff(key[order])) + 1
    for grp in np.split(order, bounds):
        simd_load.append(len(grp))
        for i in grp:
            ov = 0.0
            for j in grp:
                if j != i:
                    ov += max(0.0, min(end[i], end[j]) - max(start[i], start[j]))
            frac[i] = ov / (end[i] - start[i])
    d = end - start
    alone, paired = frac < 0.05, frac > 0.8
    dur_alone += list(d[alone]); dur_paired += list(d[paired]); frac_all += list(frac)
    cyc_alone += list(cyc[alone]); cyc_paired += list(cyc[paired])
frac_all = np.array(frac_all)
print("build: amdgpu_waves_per_eu(%d), robots %d = %d waves per launch, %d launches" % (wpe, robots, robots // 4, launches))
print("launch length (us)                                   %10.2f" % np.mean(launch_len))
print("waves per SIMD slot actually used (mean / max)       %10.2f %6d" % (np.mean(simd_load), np.max(simd_load)))
print("waves alone on their SIMD (overlap < 5 %% of life)    %10d   duration %8.2f us   shader cycles %9.0f" % (
    len(dur_alone), np.mean(dur_alone) if dur_alone else np.nan, np.mean(cyc_alone) if cyc_alone else np.nan))
print("waves sharing their SIMD (overlap > 80 %% of life)    %10d   duration %8.2f us   shader cycles %9.0f" % (
    len(dur_paired), np.mean(dur_paired) if dur_paired else np.nan, np.mean(cyc_paired) if cyc_paired else np.nan))
print("waves in between                                     %10d" % int(((frac_all >= 0.05) & (frac_all <= 0.8)).sum()))
if dur_alone and dur_paired:
    r = np.mean(dur_paired) / np.mean(dur_alone)
    print("paired / alone duration                              %10.3f   (1.0 = perfect overlap, 2.0 = no overlap at all)" % r)
    print("throughput of a shared SIMD vs a SIMD with one wave  %10.3f x" % (2.0 / r))
