"""Which phases make the slowest waves of a launch slow?  (development aid, GPU box; -DORR_PHASE_TIMERS build)

usage:  python tools/wave_phases.py [launches=40]
Every wave of the instrumented build keeps its own per-phase shader-clock totals; this tool contrasts, per launch, the slowest 2 % of the
1024 waves with the median waves, split by whether the wave had a reset, and prints the phases by their contribution to the gap
(the launch ends with its slowest wave: tools/wave_timeline.py).
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

NAMES = ["load+leg consts", "set_act/filter", "substep control", "leg dynamics", "fall proxies", "row setup", "row response",
         "Delassus columns", "PGS sweeps", "du+integrate", "receive_obs (ring)", "ctrl_obs+sensors", "reward+ref update",
         "termination+obs", "episode end/reset", "store"]
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W = 1024
env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=4 * W, seed=0)
env.reset()
g = torch.Generator().manual_seed(0)
act = (torch.randn(4 * W, 12, generator=g) * 0.1).to(env.device)
L = _lib.load()
L.orr_debug_wave_phases.argtypes = [C.POINTER(C.c_longlong), C.c_int]
L.orr_debug_wave_timeline.argtypes = [C.POINTER(C.c_longlong), C.c_int]
for _ in range(300):
    env.step(act)
pb, tb = (C.c_longlong * (40 * W))(), (C.c_longlong * (4 * W))()
gap_all, gap_nr, med_all, tot_rows, lim_rows = [], [], [], [], []
for _ in range(launches):
    env.step(act)
    L.orr_debug_wave_phases(pb, W)
    L.orr_debug_wave_timeline(tb, W)
    ph = np.frombuffer(pb, dtype=np.int64).reshape(W, 40).astype(np.float64).copy()
    tl = np.frombuffer(tb, dtype=np.int64).reshape(W, 4).copy()
    reset = (tl[:, 3] & 0xFF) != 0
    tot = ph[:, :16].sum(axis=1)
    order = np.argsort(tot)
    slow = order[-W // 50:]
    mid = order[W // 2 - W // 20: W // 2 + W // 20]
    gap_all.append(ph[slow, :16].mean(axis=0) - ph[mid, :16].mean(axis=0))
    nr = np.flatnonzero(~reset)
    o2 = nr[np.argsort(tot[nr])]
    gap_nr.append(ph[o2[-len(o2) // 50:], :16].mean(axis=0) - ph[o2[len(o2) // 2 - len(o2) // 20: len(o2) // 2 + len(o2) // 20], :16].mean(axis=0))
    med_all.append(ph[mid, :16].mean(axis=0))
    tot_rows.append((tot[mid].mean(), tot[slow].mean(), tot.max(), reset[slow].mean(), tot[o2[-1]], np.percentile(tot, 99)))
    lim_rows.append((ph[slow, 34].mean(), ph[mid, 34].mean()))
ga, gn, md, tr, lr = (np.mean(np.array(x), axis=0) for x in (gap_all, gap_nr, med_all, tot_rows, lim_rows))
print("shader cycles per wave and launch: median waves %.0f, slowest 2 %% %.0f, slowest %.0f (p99 %.0f); share of the slowest 2 %% that had a reset %.2f; slowest wave WITHOUT a reset %.0f"
      % (tr[0], tr[1], tr[2], tr[5], tr[3], tr[4]))
print("PGS sweeps with the joint-limit bank (inside 'PGS sweeps'): slowest 2 %% %.0f, median waves %.0f cycles" % (lr[0], lr[1]))
print("%-24s %12s %22s %28s" % ("phase", "median wave", "slowest 2 % - median", "same, waves without reset"))
for k in np.argsort(-ga):
    print("%-24s %12.0f %22.0f %28.0f" % (NAMES[k], md[k], ga[k], gn[k]))
print("%-24s %12.0f %22.0f %28.0f" % ("sum", md.sum(), ga.sum(), gn.sum()))
