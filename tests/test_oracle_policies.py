"""The shipped policies on the float64 CPU oracle (no GPU): the behavioural evidence of DESIGN.md section 7c is a property of the restated
algorithm + the hand-authored tables, not of the HIP implementation - the pace policy walks on the oracle too, and the out-of-sample
trot policy falls there as well (tests/test_gpu_policies.py pins the levels on the HIP path with 256-1024 robots)."""
import os

import numpy as np
import pytest

from openroborl_amd import _abi, config, motion, robots
from tests import oracle_lib as ol


def _run(policy, clip_name, n=32, steps=300, seed=1):
    W = np.load(os.path.join(ol.GOLDEN, "policy_%s.npz" % policy))
    w = {k: W[k].astype(np.float64) for k in W.files}
    clip = motion.MotionClip(clip_name)
    cfg = config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False, seed=seed, num_procs=1,
                             auto_reset=False, legacy_grid=False)
    models = [robots.laikago(), None, None, None]
    orc = ol.OracleEnv(cfg, models, [clip], n, robot_type=np.zeros(n, dtype=np.int32), clip_id=np.zeros(n, dtype=np.int32), threads=8)
    obs = orc.reset()
    alive = np.ones(n, dtype=bool)
    length = np.zeros(n)
    lay = orc.lay
    for _ in range(steps):
        h = np.maximum(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"], 0.0)
        h = np.maximum(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"], 0.0)
        a = np.clip(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
        obs, rew, done = orc.step(a)
        length += alive
        reason = orc.state[:, lay.sl("DONE_REASON")][:, 0].astype(int)
        alive &= ~(done & ((reason & ~_abi.DONE_TIME_LIMIT) != 0))
    orc.close()
    return alive.mean(), length.mean()


def test_pace_policy_walks_and_trot_policy_falls_on_the_oracle():
    up, ln = _run("laikago_pace", "laikago_pace")
    assert up == 1.0 and ln == 300                                   # HIP path: 1.000 of 1024 over 600 steps
    up, ln = _run("laikago_trot", "laikago_trot")
    assert up <= 0.4 and 60 <= ln <= 260, (up, ln)                   # HIP path: 0.19 still up after 200 steps, mean survival 137-144 steps
