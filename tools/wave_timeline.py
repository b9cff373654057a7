"""Per-wave timeline of step launches (development aid): start skew, durations and the tail of the 1024 waves.

usage (GPU box):  python tools/wave_timeline.py [launches]
Uses the -DORR_PHASE_TIMERS build (see tools/phase_cycles.py).  The kernel ends when its slowest wave ends, so what matters is the
latest end, not the mean: the report splits the waves of each launch by whether one of their robots finished an episode (reset inside
the launch) and shows which group sets the end of the launch.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openroborl_amd", "libopenroborl_phase_timers.so")
from openroborl_amd import _lib as _build  # noqa: E402
_build.build(out_path=LIB, extra_flags=["-DORR_PHASE_TIMERS"])
os.environ["ORR_LIB_PATH"] = LIB

import torch  # noqa: E402
from openroborl_amd import _lib  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=4096, seed=0)
env.reset()
g = torch.Generator().manual_seed(0)
act = (torch.randn(4096, 12, generator=g) * 0.1).to(env.device)
L = _lib.load()
L.orr_debug_wave_timeline.argtypes = [C.POINTER(C.c_longlong), C.c_int]
for _ in range(300):
    env.step(act)
W = 1024
buf = (C.c_longlong * (4 * W))()
rows = []
for _ in range(n):
    env.step(act)
    L.orr_debug_wave_timeline(buf, W)
    a = np.frombuffer(buf, dtype=np.int64).reshape(W, 4).copy()
    t0 = a[:, 0].min()
    start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0      # microseconds (100 MHz realtime counter)
    reset = (a[:, 3] & 0xFF) != 0
    last = int(np.argmax(end))
    rows.append((start.max(), end.max(), (end - start).mean(), (end - start)[reset].mean() if reset.any() else np.nan,
                 (end - start)[~reset].mean(), end[~reset].max(), end[reset].max() if reset.any() else np.nan, reset.sum(), bool(reset[last]),
                 a[:, 2].mean(), a[reset, 2].mean() if reset.any() else np.nan, a[~reset, 2].mean(), a[~reset, 2].max()))
r = np.array(rows, dtype=np.float64)
names = ["latest wave start (us)", "latest wave end = launch length (us)", "mean wave duration (us)", "  waves with a reset (us)",
         "  waves without (us)", "latest end among waves without a reset (us)", "latest end among waves with a reset (us)", "waves with a reset",
         "launches whose last wave had a reset (fraction)", "mean shader cycles per wave", "  waves with a reset", "  waves without",
         "  max over waves without"]
for k, nm in enumerate(names):
    print("%-55s %10.2f" % (nm, np.nanmean(r[:, k])))
