"""The shipped policies on the float64 CPU oracle (no GPU): the behavioural evidence of DESIGN.md section 7 is a property of the restated
algorithm + the robot tables, not of the HIP implementation.  Round 5: on the Laikago table identified against laikago_trot + laikago_spin
(tools/laikago_identify.py) the trot policy walks on the oracle as it does on the HIP path, and on round 4's table (robots.LAIKAGO_R04) it
falls on both (tests/test_gpu_policies.py pins the levels on the HIP path with 256-1024 robots)."""
import os

import numpy as np
import pytest

from openroborl_amd import _abi, config, motion, robots
from tests import oracle_lib as ol


def _run(policy, clip_name, n=32, steps=300, seed=1, table=None):
    W = np.load(os.path.join(ol.GOLDEN, "policy_%s.npz" % policy))
    w = {k: W[k].astype(np.float64) for k in W.files}
    clip = motion.MotionClip(clip_name)
    cfg = config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False, seed=seed, num_procs=1,
                             auto_reset=False, legacy_grid=False)
    models = [robots.laikago(**(table or {})), None, None, None]
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


def test_pace_and_trot_policies_walk_on_the_oracle_and_the_trot_policy_falls_on_the_round_4_table():
    up, ln = _run("laikago_pace", "laikago_pace")
    assert up == 1.0 and ln == 300                                   # held-out policy; HIP path: 1.000 over 600 steps
    up, ln = _run("laikago_trot", "laikago_trot")
    assert up >= 0.85 and ln >= 270, (up, ln)                        # fit policy; HIP path: 0.92 finish the 600 steps
    up, ln = _run("laikago_trot", "laikago_trot", table=robots.LAIKAGO_R04)
    assert up <= 0.4 and 60 <= ln <= 260, (up, ln)                   # round 4's table: HIP path 0.19 still up after 200 steps, mean survival 137-144


def _minicheetah_run(n, steps, margin=None, seed=1, table=None):
    W = np.load(os.path.join(ol.GOLDEN, "policy_minicheetah_trot.npz"))
    w = {k: W[k].astype(np.float64) for k in W.files}
    clip = motion.MotionClip("minicheetah_trot")
    cfg = config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False, seed=seed, num_procs=1,
                             auto_reset=False, legacy_grid=False)
    cfg.ref_state_init_prob = 1.0
    if margin is not None:
        cfg.contact_margin = margin
    orc = ol.OracleEnv(cfg, [None, robots.mini_cheetah(**(table or {})), None, None], [clip], n, robot_type=np.ones(n, dtype=np.int32),
                       clip_id=np.zeros(n, dtype=np.int32), threads=8)
    obs = orc.reset()
    phase = (orc.field("TIME_OFFSET")[:, 0] / (clip.frame_duration * (clip.num_frames - 1))) % 1.0
    lay = orc.lay
    alive = np.ones(n, dtype=bool)
    length = np.zeros(n)
    first_reason = np.zeros(n, dtype=int)
    z_at_end = np.zeros(n)
    for _ in range(steps):
        h = np.maximum(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"], 0.0)
        h = np.maximum(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"], 0.0)
        a = np.clip(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
        obs, rew, done = orc.step(a)
        length += alive
        reason = orc.state[:, lay.sl("DONE_REASON")][:, 0].astype(int)
        failed = alive & done & ((reason & ~_abi.DONE_TIME_LIMIT) != 0)
        first_reason[failed] = reason[failed]
        z_at_end[failed] = orc.field("POS")[failed, 2]
        alive &= ~failed
    orc.close()
    return phase, alive, length, first_reason, z_at_end


def test_minicheetah_phase_020_fallers_are_ended_by_the_contact_margin_not_by_a_fall():
    """HISTORY.md round 5, item 5 / VERDICT r4 item 2, on the table of ROUNDS 3-5 (robots.MINI_CHEETAH_R03): the shipped mini-cheetah policy
    lost ~10 % of its episodes there, every one of them started in one of two windows of the trot cycle.  The window around phase 0.20, on the
    oracle: the teleported reference state is the flight phase just before the FL / RR touchdown (one toe < 1 mm above the ground, the others
    2-8 cm up; NO toe penetrates, so the erp push-out of a teleport into the ground - the round-4 reviewer's hypothesis - never happens:
    the first normal impulse comes ~40 sub-steps later).  The robot lands on ONE leg of the pair, its partner stays 2-6 mm above the ground
    for four env steps, the trunk sinks ~5 cm and rolls 0.27 rad - and the episode ends at step 6-10 because a KNEE (a termination-only
    proxy of radius 0) comes within the 2 cm contact margin of the plane: imitation_task.py:536-546 ends an episode on ANY getContactPoints
    entry of a non-foot link, and this engine lists a proxy as soon as it is inside cfg.contact_margin (0.02 until round 5: Bullet's absolute
    gContactBreakingThreshold; with its relative threshold the margin of a link-sized shape is a few millimetres - recollection,
    unverifiable here).  The robot has not fallen: its trunk is still 24-26 cm up, and with a 4 mm margin - the default since round 6's
    cross-robot rule - the very same starts walk on.  On the table that SHIPS since round 6 both windows walk."""
    n = 384
    r3 = robots.MINI_CHEETAH_R03
    phase, alive, length, reason, z_end = _minicheetah_run(n, 60, margin=0.02, table=r3)      # the Bullet library's margin: what rounds 1-5 shipped
    win = (phase >= 0.195) & (phase < 0.215)
    assert win.sum() >= 5
    assert not alive[win].any() and length[win].max() <= 14                 # every start in the window ends within 0.5 s ...
    assert np.all((reason[win] & _abi.DONE_CONTACT_FALL) != 0)              # ... by a "contact" of a non-foot link ...
    assert np.all(z_end[win] > 0.22)                                        # ... with the trunk still up (it stands at 0.28)
    phase2, alive2, length2, _, _ = _minicheetah_run(n, 60, table=r3)       # round 6's default margin: the 4 mm of config.PYBULLET_REMEMBERED (rule P5)
    np.testing.assert_array_equal(phase, phase2)
    assert alive2[win].all()                                                # the same starts with a 4 mm margin: nobody is stopped
    # the second window (0.93-0.98) is a different story - the robot lands on the wrong pair and really falls - and stays, on that table
    win2 = (phase >= 0.94) & (phase < 0.965)
    assert win2.sum() >= 5 and not alive[win2].any() and not alive2[win2].any()
    # the table that ships (round 6): both windows walk through the first 60 steps
    phase3, alive3, _, _, _ = _minicheetah_run(n, 60)
    np.testing.assert_array_equal(phase, phase3)
    assert alive3[win].all() and alive3[win2].mean() >= 0.8
