"""Per-phase shader-clock cycles of the step kernel, measured by one instrumented wave (development aid).

usage (GPU box):  python tools/phase_cycles.py [steps]
Builds a second library with -DORR_PHASE_TIMERS next to the shipped one and runs the bench workload through it.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openroborl_amd", "libopenroborl_phase_timers.so")
from openroborl_amd import _lib as _build  # noqa: E402  (build only; the library is loaded below)
_build.build(out_path=LIB, extra_flags=["-DORR_PHASE_TIMERS"])
os.environ["ORR_LIB_PATH"] = LIB

import torch  # noqa: E402
from openroborl_amd import _lib  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

NAMES = ["load+leg consts", "set_act/filter", "substep control", "leg dynamics", "fall proxies", "row setup", "row response",
         "Delassus columns", "PGS sweeps", "du+integrate", "receive_obs (ring)", "ctrl_obs+sensors", "reward+ref update",
         "termination+obs", "episode end/reset", "store"]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=4096, seed=0)
env.reset()
g = torch.Generator().manual_seed(0)
act = (torch.randn(4096, 12, generator=g) * 0.1).to(env.device)
L = _lib.load()
L.orr_debug_phase_cycles.argtypes = [C.POINTER(C.c_longlong), C.c_int]
buf = (C.c_longlong * 40)()
for _ in range(50):
    env.step(act)
L.orr_debug_phase_cycles(buf, 1)
events = 0       # env steps in which the instrumented wave (block gridDim / 2 = robots 2048..2051) reset at least one robot
dones = torch.zeros(4, dtype=torch.int64, device=env.device)
for _ in range(steps):
    _, _, d, _ = env.step(act)
    dones += d[2048:2052].long()
    events += int(d[2048:2052].any())
L.orr_debug_phase_cycles(buf, 1)
tot = float(sum(buf[:16]))
print("cycles per env step (one wave, %d steps): %.0f" % (steps, tot / steps))
for n, v in zip(NAMES, buf[:16]):
    print("  %-22s %9.0f  %5.1f%%" % (n, v / steps, 100.0 * v / tot))
print("steps with a reset in the instrumented wave: %d of %d (robot resets: %s); cycles per such step in 'episode end/reset': %.0f"
      % (events, steps, dones.tolist(), buf[14] / max(events, 1)))
print("PGS sweeps in sub-steps with the joint-limit bank (not in the table above): %.0f cycles per env step" % (buf[34] / steps))
print("stages of reset_robot in program order, cycles per reset of robot 0 of the wave (a mark closes the interval since the previous one):")
RESET = [(16, "state defaults"), (24, "Philox blocks (28 draws)"), (20, "task draws: start time"), (26, "frame indices (clip_index)"),
         (19, "frame + mass-table loads issued"), (17, "ring entry #1"), (29, "control observation copy"), (18, "sensor histories"),
         (25, "randomiser scatter + mass refresh"), (27, "frames staged to LDS"), (21, "pose blend"), (22, "origin + teleport"),
         (23, "ring entry #2 + time limit"), (30, "target observation"), (31, "episode log + entry (before reset_robot)")]
tot_r = 0.0
for k, n in RESET:
    v = buf[k] / max(int(dones[0]), 1)
    tot_r += v
    print("  %-40s %9.0f" % (n, v))
print("  %-40s %9.0f" % ("sum", tot_r))
