#!/usr/bin/env python3
"""GPU-side dump for the clip-sampler parity test (tests/test_gpu_clips.py): states before / after each env step, both sides,
into gpurun_out/clip_dump_<clip>.npz for analysis without a GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openroborl_amd import state as statemod   # noqa: E402
from tests import oracle_lib as ol             # noqa: E402
from tests.test_gpu_clips import _pair, _g64, engineered_times   # noqa: E402

for clip in sys.argv[1:]:
    n = 96
    env, orc = _pair(clip, n, seed=22)
    env.reset(); orc.reset()
    lay = env.layout
    st = engineered_times(env, _g64(env))
    env.state.copy_(torch.from_numpy(statemod.from_float64(lay, st)).to(env.device))
    rng = np.random.RandomState(5)
    out = {}
    head = lay.sl("RING").start
    for step in range(3):
        orc.state[:] = _g64(env)
        out["before%d" % step] = orc.state[:, :head].copy()
        a = rng.uniform(-0.1, 0.1, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        out["gpu%d" % step] = _g64(env)[:, :head]
        out["orc%d" % step] = orc.state[:, :head].copy()
        out["obs_gpu%d" % step] = og.cpu().numpy()
        out["obs_orc%d" % step] = oo
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "clip_dump_%s.npz" % clip), **out)
    env.close(); orc.close()
    print("dumped", clip)
