#!/usr/bin/env python3
"""Golden fixtures from the reference's OWN task / robot / env / wrapper / randomiser code, driven with a scripted client.

Run ONLY in the build container (needs /root/reference; the GPU box never has it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_task.py

What is imported from the reference (read-only, never copied) and exercised END TO END through WrapperEnv.reset()/step():
  envs/quadruped_robot/wrapper_env.py           WrapperEnv            (A1, A2: time limit, curriculum, target-obs append)
  envs/quadruped_robot/quadruped_gym_env.py     LocomotionGymEnv      (A3-A5: 33-sub-step loop, reward -> update -> done)
  envs/quadruped_robot/robots/minitaur.py       Minitaur              (B1-B6, D1-D3: filter, lerp, clip, PD, latency ring, sensors)
  envs/quadruped_robot/task/imitation_task.py   ImitationTask         (F1-F5: reset, update, reward terms, termination, target obs)
  envs/utilities/randomizer/controllable_env_randomizer_from_config.py  (H: parameter mapping)
plus everything make_golden.py already covers (motion_data, pose3d, action_filter, minitaur_motor, sensors).

The pybullet client those classes talk to is tests/golden/fake_bullet.py: a scripted stand-in (NOT a physics engine) whose
per-sub-step states, link positions and contact lists are recorded here and injected into the oracle's replay mode by
tests/test_oracle_golden_task.py.  So these fixtures pin every row of SURVEY.md section 8a except C (the physics engine
itself) and the link-COM forward kinematics (URDF geometry; also third-party data).

Environment shims (tests/golden/_shims): gym, tensorflow (logging only), pybullet constants, pybullet_data,
pybullet_utils (transformations restatement + bullet_client -> the scripted client), absl.logging.  One interpreter
compatibility patch: `collections.Sequence` (removed in Python 3.10, used at minitaur.py:169,174) is aliased to
collections.abc.Sequence.  Reproducibility patches on INSTANCES (no reference source is modified): np.random is seeded,
each randomiser's unseeded RandomState (controllable_env_randomizer_from_config.py:60) is replaced by a seeded one, each
task's _rand_uniform / cal_reward are wrapped to log the draws and the five reward terms.

Note: run.py:57 builds Minitaur(name_robot="minicheetah"), which minitaur.py:93-97 rejects ("wrong robot select"); the
mini-cheetah robot is constructed here with the name minitaur.py accepts ("mini_cheetah").

Outputs (committed): task_laikago.npz, task_mini_cheetah.npz, task_laikago_testmode.npz, task_laikago_spin.npz
"""
import collections
import collections.abc
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
REF = os.path.join(REF_ROOT, "OpenRoboRL")
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, REF)
if not hasattr(collections, "Sequence"):
    collections.Sequence = collections.abc.Sequence

os.chdir(REF_ROOT)   # quadruped_gym_env.py:159 opens 'OpenRoboRL/config/pybullet_sim_param.yaml' relative to the CWD

from envs.quadruped_robot import quadruped_gym_env, wrapper_env  # noqa: E402
from envs.quadruped_robot.robots import minitaur  # noqa: E402
from envs.quadruped_robot.task import imitation_task  # noqa: E402

MOTIONS = os.path.join(REF, "envs/quadruped_robot/task/motions")
CLIP = {"laikago": "laikago_pace.txt", "mini_cheetah": "minicheetah_trot.txt"}
TOES = [3, 7, 11, 15]


def build(robot_name, n, randomizer, ep_start, ep_end, curriculum_steps, seed, clip=None):
    np.random.seed(seed)
    robots = [minitaur.Minitaur(name_robot=robot_name, robot_index=i, enable_randomizer=randomizer) for i in range(n)]
    for i, r in enumerate(robots):
        r._randomizers[0]._np_random = np.random.RandomState(1000 * seed + i)
    tasks = [imitation_task.ImitationTask(ref_motion_filenames=[os.path.join(MOTIONS, clip or CLIP[robot_name])],
                                          enable_cycle_sync=True, tar_frame_steps=[1, 2, 10, 30],
                                          ref_state_init_prob=0.9, warmup_time=0.25) for _ in range(n)]     # run.py:58-64
    for t in tasks:
        t._draws = []
        t._terms = None

        def logged_uniform(lo, hi, size=None, _t=t, _orig=t._rand_uniform):
            v = _orig(lo, hi, size=size)
            _t._draws.append((lo, hi, v))
            return v
        t._rand_uniform = logged_uniform

        def logged_reward(_t=t):
            _t._terms = [_t._calc_reward_pose(), _t._calc_reward_velocity(), _t._calc_reward_end_effector(),
                         _t._calc_reward_root_pose(), _t._calc_reward_root_velocity()]
            del _t._pybullet_client.link_state_log[-16:]     # keep exactly one set of getLinkState calls per reward
            return _t.reward()
        t.cal_reward = logged_reward
    env = quadruped_gym_env.LocomotionGymEnv(robots, tasks)
    env = wrapper_env.WrapperEnv(env, episode_length_start=ep_start, episode_length_end=ep_end,
                                 curriculum_steps=curriculum_steps, num_parallel_envs=1)
    return env, robots, tasks


def reset_record(env, robots, tasks, fake):
    """One WrapperEnv.reset(); returns per-robot dict of everything the oracle's reset needs / must reproduce."""
    n = len(robots)
    for t in tasks:
        t._draws = []
    obs = env.reset()
    out = []
    for i in range(n):
        r, t = robots[i], tasks[i]
        uni = np.zeros(28)
        uni[14:26] = 0.5      # neutral values when the randomiser is off (never read then)
        if r._enable_randomizer:
            d = r._randomizers[0]._randomization_param_value_dict
            u = {k: (np.asarray(v, dtype=np.float64) + 1.0) / 2.0 for k, v in d.items()}   # sample in [-1,1] -> [0,1)
            # oracle draw order = sorted parameter names with the two no-op parameters (battery, motor friction) left out:
            # inertia 2, joint friction 8, latency 1, lateral friction 1, mass 2, motor strength 12
            uni[0:2] = u["inertia"]
            uni[2:10] = u["joint friction"]
            uni[10] = u["latency"]
            uni[11] = u["lateral friction"]
            uni[12:14] = u["mass"]
            uni[14:26] = u["motor strength"]
        assert len(t._draws) == 2, t._draws
        uni[26] = t._draws[0][2]
        uni[27] = t._draws[1][2] / t._draws[1][1]           # U(0, hi) / hi
        body = r.quadruped
        b = fake.bodies[body]
        out.append(dict(
            uniforms=uni, obs=np.asarray(obs[i], dtype=np.float64), state37=fake.state37(body),
            time_offset=t._motion_time_offset, warmup=float(t._curr_episode_warmup),
            origin_pos=np.array(t._origin_offset_pos, dtype=np.float64), origin_rot=np.array(t._origin_offset_rot, dtype=np.float64),
            prev_phase=float(t._prev_motion_phase), ref_pose=np.array(t._ref_pose), ref_vel=np.array(t._ref_vel),
            latency=float(r._control_latency), strength=np.array(r._motor_model._strength_ratios, dtype=np.float64) * np.ones(12),
            ring_len=float(len(r._observation_history)), max_episode_steps=float(env._max_episode_steps),
            total_step_count=float(env._total_step_count),
            # what the randomiser did to the (made-up) URDF inertial data, per link -1..15: ratios new / urdf
            mass_ratio=np.array([_last(b.dyn_calls, l, "mass", b.mass[l]) / b.mass[l] for l in range(-1, 16)]),
            inertia_ratio=np.array([np.asarray(_last(b.dyn_calls, l, "localInertiaDiagonal", b.inertia[l]))[0] / b.inertia[l][0]
                                    for l in range(-1, 16)]),
            lateral_friction=np.array([_last(b.dyn_calls, l, "lateralFriction", -1.0) for l in range(-1, 16)]),
            joint_friction_force=b.vel_motor_force.copy()))
    return out


def _last(calls, link, key, default):
    v = default
    for l, kw in calls:
        if l == link and key in kw:
            v = kw[key]
    return v


def step_record(env, robots, tasks, fake, actions):
    n = len(robots)
    fake.substep_log = []
    fake.link_state_log = []
    a_in = [a.copy() for a in actions]
    obs, rew, done, info = env.step(actions)
    assert len(fake.substep_log) == 33 and len(fake.link_state_log) == 16 * n
    out = []
    for i in range(n):
        r, t = robots[i], tasks[i]
        body = r.quadruped
        ghost = t._ref_model
        mine = [e for e in fake.link_state_log if e[0] in (body, ghost)]
        eff_ref = np.stack([e[2] for e in mine if e[0] == ghost])
        eff_sim = np.stack([e[2] for e in mine if e[0] == body])
        assert eff_ref.shape == (8, 3) and eff_sim.shape == (8, 3)
        fall = float(any(l not in r._foot_link_ids for l in fake.contact_links.get(body, [])))
        out.append(dict(
            action=a_in[i], action_mutated=np.asarray(actions[i], dtype=np.float64),
            traj=np.stack([fake.substep_log[s][body][0] for s in range(33)]),
            tau_urdf=np.stack([fake.substep_log[s][body][1] for s in range(33)]),
            eff_ref=eff_ref, eff_sim=eff_sim, fall=fall,
            obs=np.asarray(obs[i], dtype=np.float64), reward=float(rew[i]), terms=np.array(t._terms, dtype=np.float64),
            done=float(done[i]), env_step_counter=float(env.env_step_counter), max_episode_steps=float(env._max_episode_steps),
            total_step_count=float(env._total_step_count),
            filtered_action=np.array(r._action, dtype=np.float64), ctrl_obs=np.array(r._control_observation, dtype=np.float64),
            origin_pos=np.array(t._origin_offset_pos, dtype=np.float64), prev_phase=float(t._prev_motion_phase),
            ref_pose=np.array(t._ref_pose), ref_vel=np.array(t._ref_vel)))
    return out, done


def kick(body, delta):
    def ev(world):
        world.bodies[body].pos = world.bodies[body].pos + np.asarray(delta, dtype=np.float64)
    return ev


def flip(body, angle):
    def ev(world):
        import fake_bullet as fb
        b = world.bodies[body]
        b.orn = fb.qmul(np.array([np.sin(angle / 2), 0.0, 0.0, np.cos(angle / 2)]), b.orn)
    return ev


def run(robot_name, n, randomizer, ep_start, ep_end, curriculum_steps, seed, total_steps, events, clip=None):
    env, robots, tasks = build(robot_name, n, randomizer, ep_start, ep_end, curriculum_steps, seed, clip)
    fake = env.pybullet_client
    rng = np.random.RandomState(seed + 77)
    init = robots[0]._init_motor_angle
    jdir, joff = robots[0]._motor_direction, robots[0]._motor_offset
    resets, steps, marks = [], [], []      # marks: ("reset", index into resets) / ("step", index into steps)
    for i in range(n):
        fake.contact_links[robots[i].quadruped] = list(TOES)
    need_reset = True
    k = 0
    while k < total_steps:
        if need_reset:           # imitation_runners.py:185-205: the caller resets the whole env when any robot is done
            resets.append(reset_record(env, robots, tasks, fake))
            marks.append(("reset", len(resets) - 1))
            need_reset = False
        # actions: follow the reference pose one control step ahead (motor space, minus the offset the env adds) + noise
        acts = []
        for i in range(n):
            tar = tasks[i]._calc_ref_pose(tasks[i]._get_motion_time() + 0.033)[7:]
            jid = [robots[i]._joint_name_to_id[nm] for nm in robots[i].name_motor]          # URDF joint id per motor
            rev = [j - j // 4 for j in jid]                                                   # index among the 12 revolute joints
            tar_motor = (tar[rev] - joff) * jdir
            acts.append(np.clip(tar_motor - init + rng.randn(12) * 0.125, -2 * np.pi, 2 * np.pi))
        ev = events.get(k)
        for i in range(n):
            fake.contact_links[robots[i].quadruped] = list(TOES)
        if ev is not None:
            kind, who, arg = ev
            body = robots[who].quadruped
            if kind == "kick":
                fake.events[fake.sim_steps + 20] = kick(body, arg)
            elif kind == "flip":
                fake.events[fake.sim_steps + 25] = flip(body, arg)
            elif kind == "contact":
                fake.contact_links[body] = list(arg)
        rec, done = step_record(env, robots, tasks, fake, acts)
        steps.append(rec)
        marks.append(("step", len(steps) - 1))
        k += 1
        if any(done):
            need_reset = True
    out = {"robot": np.array(robot_name), "clip": np.array((clip or CLIP[robot_name])[:-4]), "num_robot": np.float64(n), "randomizer": np.float64(randomizer),
           "ep_start": np.float64(ep_start), "ep_end": np.float64(ep_end), "curriculum_steps": np.float64(curriculum_steps),
           "marks": np.array([(0.0 if m[0] == "reset" else 1.0, float(m[1])) for m in marks]),
           "joint_of_motor": np.array([(lambda j: j - j // 4)(robots[0]._joint_name_to_id[nm]) for nm in robots[0].name_motor], dtype=np.float64),
           "foot_link_ids": np.array(robots[0]._foot_link_ids, dtype=np.float64),
           "chassis_link_ids": np.array(robots[0]._chassis_link_ids, dtype=np.float64),
           "leg_link_ids": np.array(robots[0]._leg_link_ids, dtype=np.float64),
           "motor_link_ids": np.array(robots[0]._motor_link_ids, dtype=np.float64),
           "foreign_dyn_calls": np.array([[c[0], c[1]] for c in fake.foreign_dyn_calls], dtype=np.float64),
           "engine_numSolverIterations": np.float64(fake.engine.get("numSolverIterations", -1)),
           "engine_enableConeFriction": np.float64(fake.engine.get("enableConeFriction", -1)),
           "gravity": np.array(fake.gravity, dtype=np.float64), "time_step": np.float64(fake.time_step)}
    f32 = ("traj",)     # float32-exact by construction except the quaternion columns, which are stored separately in f64
    for name in resets[0][0]:
        out["reset/" + name] = np.stack([np.stack([np.asarray(r[i][name], dtype=np.float64) for i in range(n)]) for r in resets])
    for name in steps[0][0]:
        arr = np.stack([np.stack([np.asarray(s[i][name], dtype=np.float64) for i in range(n)]) for s in steps])
        if name in f32:
            out["step/traj_quat"] = arr[..., 3:7].copy()
            a32 = arr.astype(np.float32)
            chk = a32.astype(np.float64)
            chk[..., 3:7] = arr[..., 3:7]
            assert np.array_equal(chk, arr)
            out["step/traj_f32"] = a32
        else:
            out["step/" + name] = arr
    return out


def main():
    # events: env step index -> (kind, robot, argument)
    ev_l = {9: ("contact", 1, [3, 7, 2]),           # lower-leg contact: allowed (foot link set includes lower legs)
            15: ("contact", 0, [3, 1]),             # upper-leg contact: contact_fall
            33: ("kick", 1, [0.9, 0.7, 0.0]),       # root position drift > 1 m
            52: ("flip", 0, 1.9),                   # root rotation error > pi/2
            70: ("contact", 1, [-1])}               # chassis contact
    out = run("laikago", 2, True, 8, 24, 60, seed=1, total_steps=110, events=ev_l)
    np.savez_compressed(os.path.join(HERE, "task_laikago.npz"), **out)
    ev_m = {11: ("contact", 0, [0, 15]), 30: ("flip", 1, -1.8), 47: ("kick", 0, [-1.1, 0.2, 0.1])}
    out = run("mini_cheetah", 2, True, 8, 24, 60, seed=2, total_steps=90, events=ev_m)
    np.savez_compressed(os.path.join(HERE, "task_mini_cheetah.npz"), **out)
    # test mode (run.py:66-67,205-206): no randomiser, fixed 2 ms latency, full-length episodes, no curriculum effect
    out = run("laikago", 1, False, 600, 600, 30000000, seed=3, total_steps=45, events={})
    np.savez_compressed(os.path.join(HERE, "task_laikago_testmode.npz"), **out)
    # a clip with EnableCycleOffsetRotation (motion_data.py:591-633: the cycle offset rotates from cycle to cycle): 55 steps = 2.4
    # cycles of the 0.75 s spin clip, so target frames and the reference pose cross several cycle boundaries
    out = run("laikago", 1, True, 600, 600, 30000000, seed=4, total_steps=55, events={}, clip="laikago_spin.txt")
    np.savez_compressed(os.path.join(HERE, "task_laikago_spin.npz"), **out)
    for f in ("task_laikago.npz", "task_mini_cheetah.npz", "task_laikago_testmode.npz", "task_laikago_spin.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
