"""What makes some waves of a step launch slower than others: where they run (XCD / CU) or what they simulate (robots)?  (development aid)
usage (GPU box): python tools/diag/wave_spread.py [launches=200]          needs the -DORR_PHASE_TIMERS build (tools/wave_timeline.py)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openroborl_amd", "libopenroborl_phase_timers.so")
from openroborl_amd import _lib as _build  # noqa: E402
_build.build(out_path=LIB, extra_flags=["-DORR_PHASE_TIMERS"])
os.environ["ORR_LIB_PATH"] = LIB

import torch  # noqa: E402
from openroborl_amd import _lib  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, W = 4096, 1024
env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=N, seed=0, mode="train", auto_reset=True)
dev = env.device
m = env.models[int(env.robot_type[0])]
mdir = torch.tensor(m["motor_dir"], dtype=torch.float32, device=dev)
const = torch.tensor(m["motor_offset"], dtype=torch.float32, device=dev) * mdir + torch.tensor(m["init_motor_angles"], dtype=torch.float32, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
pool = torch.randn(64, N, 12, generator=gen, device=dev) * 0.125 - const
obs = env.reset()
L = _lib.load()
L.orr_debug_wave_timeline.argtypes = [C.POINTER(C.c_longlong), C.c_int]
L.orr_debug_wave_phases.argtypes = [C.POINTER(C.c_longlong), C.c_int]
NAMES = ["load+leg consts", "set_act/filter", "substep control", "leg dynamics", "fall proxies", "row setup", "row response",
         "Delassus columns", "PGS sweeps", "du+integrate", "receive_obs (ring)", "ctrl_obs+sensors", "reward+ref update",
         "termination+obs", "episode end/reset", "store"]


def step(k):
    global obs
    obs, _, _, _ = env.step(torch.addcmul(pool[k & 63], obs[:, 91:103], mdir))     # bench.py's stress actions


for k in range(3000):
    step(k)
buf = (C.c_longlong * (4 * W))()
pbuf = (C.c_longlong * (40 * W))()
phases = []
dur, cyc, start, xcc, cu, se, simd, reset = [], [], [], [], [], [], [], []
for k in range(n):
    step(3000 + k)
    L.orr_debug_wave_timeline(buf, W)
    L.orr_debug_wave_phases(pbuf, W)
    phases.append(np.frombuffer(pbuf, dtype=np.int64).reshape(W, 40)[:, :16].astype(np.float64).copy())
    a = np.frombuffer(buf, dtype=np.int64).reshape(W, 4).copy()
    t0 = a[:, 0].min()
    start.append((a[:, 0] - t0) / 100.0)
    dur.append((a[:, 1] - a[:, 0]) / 100.0)
    cyc.append(a[:, 2].astype(np.float64))
    hw = (a[:, 3] >> 8) & 0xFFFFFFFF            # HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9 layout)
    simd.append((hw >> 4) & 3); cu.append((hw >> 8) & 15); se.append((hw >> 13) & 7)
    xcc.append((a[:, 3] >> 40) & 15)
    reset.append((a[:, 3] & 0xFF) != 0)
dur, cyc, start, xcc, cu, se, simd, reset = (np.array(x) for x in (dur, cyc, start, xcc, cu, se, simd, reset))
ok = ~reset
print("launches %d; wave duration (us): mean %.2f, p50 %.2f, p99 %.2f, max %.2f; launch length mean %.2f; without-reset waves only below"
      % (n, dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max(axis=1).mean(), (start + dur).max(axis=1).mean()))
d = np.where(ok, dur, np.nan)
rel = d / np.nanmedian(d, axis=1, keepdims=True) - 1.0          # relative to the launch's median wave
print("spread within a launch (no-reset waves): std %.2f %%, p99 %.2f %%, max %.2f %% above the median" %
      (100 * np.nanstd(rel), 100 * np.nanpercentile(rel, 99), 100 * np.nanmean(np.nanmax(rel, axis=1))))
# (a) by wave index = by robots: does the same wave stay slow from launch to launch?
by_wave = np.nanmean(rel, axis=0)
print("by wave index (the same four robots over %d launches): std of the per-wave mean %.2f %%, slowest wave %.2f %%, lag-1 autocorrelation of a wave's excess %.2f"
      % (n, 100 * np.nanstd(by_wave), 100 * np.nanmax(by_wave),
         np.nanmean([np.corrcoef(rel[:-1, w][~np.isnan(rel[:-1, w]) & ~np.isnan(rel[1:, w])], rel[1:, w][~np.isnan(rel[:-1, w]) & ~np.isnan(rel[1:, w])])[0, 1] for w in range(0, W, 8)])))
# (b) by location
for name, key, card in (("XCD", xcc, 8), ("shader engine", se, 8), ("CU within the engine", cu, 16), ("SIMD", simd, 4)):
    means = [100 * np.nanmean(rel[key == v]) for v in range(card) if (key == v).any()]
    print("by %-22s mean excess %% per value: %s" % (name, " ".join("%+.2f" % x for x in means)))
loc = xcc * 1000 + se * 100 + cu                      # one CU
ids = np.unique(loc)
per_cu = np.array([np.nanmean(rel[loc == i]) for i in ids])
print("by CU (%d distinct): std of the per-CU mean %.2f %%, slowest CU %+.2f %%, fastest %+.2f %%" % (len(ids), 100 * per_cu.std(), 100 * per_cu.max(), 100 * per_cu.min()))
# is the placement of a wave index stable from launch to launch?
print("placement: a wave index runs on the same CU as in the previous launch in %.1f %% of the cases" % (100 * (loc[1:] == loc[:-1]).mean()))
# (c) start skew
print("start skew: correlation(start time, duration) %.2f; latest start %.2f us; duration of the last-starting 5 %% of waves %+.2f %% vs median"
      % (np.corrcoef(start.ravel(), dur.ravel())[0, 1], start.max(axis=1).mean(),
         100 * np.nanmean(rel[start >= np.percentile(start, 95, axis=1, keepdims=True)])))
# shader-cycle count vs wall time: a clock effect (same cycles, more time) or more cycles?
c = np.where(ok, cyc, np.nan)
relc = c / np.nanmedian(c, axis=1, keepdims=True) - 1.0
m_ = ~np.isnan(rel) & ~np.isnan(relc)
print("excess in shader cycles vs excess in wall time: correlation %.2f; slowest 2 %% of waves: wall %+.2f %%, cycles %+.2f %%"
      % (np.corrcoef(rel[m_], relc[m_])[0, 1], 100 * np.nanmean(rel[rel >= np.nanpercentile(rel, 98)]), 100 * np.nanmean(relc[rel >= np.nanpercentile(rel, 98)])))
print("percentiles of a no-reset wave's duration relative to its launch's median (%%): " +
      "  ".join("p%d %+.2f" % (q, 100 * np.nanpercentile(rel, q)) for q in (1, 5, 10, 25, 50, 75, 90, 95, 99)))
ph = np.array(phases)                                    # [launch, wave, phase]
fast = rel <= np.nanpercentile(rel, 5)
mid = (rel >= np.nanpercentile(rel, 45)) & (rel <= np.nanpercentile(rel, 55))
slow = rel >= np.nanpercentile(rel, 95)
print("%-22s %12s %16s %16s" % ("phase (shader cycles)", "median waves", "fastest 5 % - median", "slowest 5 % - median"))
for k, nm in enumerate(NAMES):
    print("%-22s %12.0f %16.0f %16.0f" % (nm, ph[mid][:, k].mean(), ph[fast][:, k].mean() - ph[mid][:, k].mean(), ph[slow][:, k].mean() - ph[mid][:, k].mean()))
env.close()
