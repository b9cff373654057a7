#!/usr/bin/env python3
"""When do the waves of a step launch start and end, and what does the launch wait for?  (development aid, GPU box)

usage:  python tools/wave_times.py [config=mixed8192|laikago4096|minicheetah4096] [launches=60] [robots (overrides the config's count)]
Builds the PRODUCT kernels (both variants, the shipped flags) with -DORR_WAVE_TIMELINE: two clock reads and one 32-byte store per wave,
nothing else instrumented.  Drives the env like bench.py (stress actions) and reports per launch, averaged over the launches:
  * start skew, wave durations (mean / quantiles), launch length = last end - first start
  * by SIMD (XCC, SE, CU, SIMD from HW_ID): how many waves it ran, when its LAST wave ended, how long its waves ran alone at the end
    (the partner already gone) and at the start (the partner not yet there)
  * which SIMDs end last: their waves' durations, reset flags, and how the two waves of a pair compare (older / younger)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
config = sys.argv[1] if len(sys.argv) > 1 else "mixed8192"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 60
LIB = os.path.join(ROOT, "openroborl_amd", "libopenroborl_wave_times.so")
from openroborl_amd import _lib as _build  # noqa: E402
_build.build(out_path=LIB, extra_flags=["-DORR_WAVE_TIMELINE"])
os.environ["ORR_LIB_PATH"] = LIB

import torch  # noqa: E402
import bench  # noqa: E402
from openroborl_amd import _lib  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

kw, n, _ = bench.CONFIGS[config]
if len(sys.argv) > 3:
    n = int(sys.argv[3])
env = VecQuadrupedEnv(num_robot=n, seed=0, **kw)
obs = env.reset()
L = _lib.load()
L.orr_debug_wave_times.argtypes = [C.POINTER(C.c_longlong), C.c_int]
W = (n + 3) // 4
L.orr_debug_wave_times(None, W)                       # allocate: the launches from here on are recorded
gen = torch.Generator(device=env.device).manual_seed(0)
noise = 0.125 * torch.randn(16, n, 12, device=env.device, generator=gen)
act = torch.empty(n, 12, device=env.device)
obs_b, rew_b, done_b = torch.empty_like(obs), torch.empty(n, device=env.device), torch.empty(n, dtype=torch.uint8, device=env.device)


def step(k):
    env.stress_actions(obs_b if k else obs, noise[k % 16], act)
    env.step_into(act, obs_b, rew_b, done_b)


for k in range(400):
    step(k)
buf = (C.c_longlong * (4 * W))()
acc = {}


def add(name, v):
    acc.setdefault(name, []).append(float(v))


last_simd_rows = []
for k in range(launches):
    step(400 + k)
    L.orr_debug_wave_times(buf, W)
    a = np.frombuffer(buf, dtype=np.int64).reshape(W, 4).copy()
    start, end, cyc = a[:, 0] / 100.0, a[:, 1] / 100.0, a[:, 2].astype(np.float64)      # us
    t0 = start.min()
    start, end = start - t0, end - t0
    reset = (a[:, 3] & 0xFF) != 0
    hw = (a[:, 3] >> 8) & 0xFFFFFFFF
    xcc = (a[:, 3] >> 40) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    dur = end - start
    add("launch length (last end - first start), us", end.max())
    add("start of the last wave, us", start.max())
    add("wave duration mean, us", dur.mean())
    for q in (5, 50, 95, 100):
        add("wave duration p%d, us" % q, np.percentile(dur, q))
    add("wave end p5, us", np.percentile(end, 5)); add("wave end p50, us", np.percentile(end, 50)); add("wave end p95, us", np.percentile(end, 95))
    add("shader clock, GHz (cycles / duration)", (cyc / dur).mean() / 1e3)
    add("waves with a reset", reset.sum())
    order = np.argsort(key, kind="stable")
    bounds = np.flatnonzero(np.diff(key[order])) + 1
    groups = np.split(order, bounds)
    add("SIMDs used", len(groups))
    nper = np.array([len(g) for g in groups])
    add("waves per SIMD: max", nper.max()); add("SIMDs with a number of waves other than the mean", (nper != round(W / len(groups))).sum())
    simd_end = np.array([end[g].max() for g in groups])
    add("SIMD end (its last wave) mean, us", simd_end.mean()); add("SIMD end p95, us", np.percentile(simd_end, 95))
    # pairs: SIMDs that ran exactly two waves, overlapping
    alone_end, alone_start, older_first, d_old, d_young, gap_end = [], [], [], [], [], []
    for g in groups:
        if len(g) != 2:
            continue
        i, j = (g[0], g[1]) if start[g[0]] <= start[g[1]] else (g[1], g[0])      # i = older
        alone_start.append(start[j] - start[i])
        alone_end.append(abs(end[i] - end[j]))
        older_first.append(end[i] <= end[j])
        d_old.append(dur[i]); d_young.append(dur[j]); gap_end.append(end[j] - end[i])
    if alone_end:
        add("pairs: younger starts after the older by, us", np.mean(alone_start))
        add("pairs: one wave alone at the END for, us (mean)", np.mean(alone_end)); add("pairs: ... p95", np.percentile(alone_end, 95))
        add("pairs: the older wave ends first (fraction)", np.mean(older_first))
        add("pairs: duration of the older / the younger wave, us", np.mean(d_old)); add("pairs: (younger)", np.mean(d_young))
        add("pairs: end(younger) - end(older), us", np.mean(gap_end))
    # the SIMDs that end last
    late = np.argsort(simd_end)[-max(1, len(groups) // 50):]                              # the last 2 %
    lw = np.concatenate([groups[i] for i in late])
    add("last 2 % of the SIMDs: their waves' mean duration, us", dur[lw].mean())
    add("last 2 % of the SIMDs: share of their waves with a reset", reset[lw].mean())
    add("last 2 % of the SIMDs: start of their first wave, us", np.mean([start[groups[i]].min() for i in late]))
    add("all SIMDs: share of waves with a reset", reset.mean())
print("%s, %d robots = %d waves per launch, %d launches, product kernels + -DORR_WAVE_TIMELINE" % (config, n, W, launches))
for name, v in acc.items():
    print("%-66s %10.2f" % (name, np.mean(v)))
