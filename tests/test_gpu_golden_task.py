"""The HIP path against the reference's OWN Python, directly (run with -m gpu on an MI355X).

Same fixtures as tests/test_oracle_golden_task.py (tests/golden/task_*.npz: the reference's WrapperEnv / LocomotionGymEnv /
Minitaur / ImitationTask / randomiser driven end to end with a scripted pybullet client), replayed through the device kernels
(`orr_debug_replay_reset` / `orr_debug_replay_step`, kernel MODE 2): the recorded per-sub-step rigid states replace the physics
sub-step, the recorded link positions feed the end-effector reward, the recorded draws replace the Philox stream; everything
else -- action offset, Butterworth filter, lerp, clip, PD torques, latency ring and delay blend, sensor histories, reward,
reference-motion update / cycle sync, termination, time limit, target observation, reset -- is the product's kernel code.
float32 on the device vs the reference's float64: observations 1e-5 (rates 1e-3), torques 2e-5 relative + 2e-3, reward 5e-6
(measured worst cases: 4e-4 Nm, 1.8e-6, 9e-7).
"""
import os

import numpy as np
import pytest

from openroborl_amd import _abi
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
CLIP = {"laikago": "laikago_pace", "mini_cheetah": "minicheetah_trot"}


def _replay(name):
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    g = np.load(os.path.join(ol.GOLDEN, name))
    robot, n = str(g["robot"]), int(g["num_robot"])
    rnd = bool(g["randomizer"])
    env = VecQuadrupedEnv(num_robot=n, robot=robot, motion_file=str(g["clip"]), mode="train", enable_randomizer=rnd, auto_reset=False,
                          legacy_grid=True, seed=0,
                          config_overrides=dict(ep_len_start=int(g["ep_start"]), ep_len_end=int(g["ep_end"]), curriculum_steps=int(g["curriculum_steps"])))
    dev = env.device
    m = env.models[int(env.robot_type[0])]
    jom = np.asarray(m["joint_of_motor"])
    mdir = np.asarray(m["motor_dir"])
    traj = g["step/traj_f32"].astype(np.float64)
    traj[..., 3:7] = g["step/traj_quat"]
    f32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    count, worst = 0, {"obs": 0.0, "tau": 0.0, "rew": 0.0}
    tau_out = torch.zeros((n, 33, 12), dtype=torch.float32, device=dev)
    for kind, idx in g["marks"]:
        idx = int(idx)
        if kind == 0.0:
            env.counters[_abi.CNT_TOTAL_STEP_COUNT] = count
            obs = env.replay_reset(f32(g["reset/uniforms"][idx])).cpu().numpy()
            R = lambda key: g["reset/" + key][idx]
            what = "reset %d " % idx
            np.testing.assert_allclose(obs, R("obs"), atol=2e-5, err_msg=what + "observation")
            F = lambda name: env.field(name).cpu().numpy()
            np.testing.assert_array_equal(env.field_int("MAX_EP_STEPS")[:, 0].cpu().numpy(), R("max_episode_steps").astype(int), err_msg=what + "time limit")
            np.testing.assert_array_equal(env.field_int("WARMUP")[:, 0].cpu().numpy(), R("warmup").astype(int), err_msg=what + "warm-up flag")
            np.testing.assert_array_equal(env.field_int("RING_LEN")[:, 0].cpu().numpy(), R("ring_len").astype(int))
            st = env.state[:, 0:37].cpu().numpy()
            np.testing.assert_allclose(st[:, 0:7], R("state37")[:, 0:7], atol=2e-6, err_msg=what + "teleported base pose")
            np.testing.assert_allclose(st[:, 13:25], R("state37")[:, 13:25], atol=2e-6, err_msg=what + "teleported joints")
            np.testing.assert_allclose(st[:, 7:13], R("state37")[:, 7:13], atol=2e-4, err_msg=what + "teleported base velocity")
            np.testing.assert_allclose(st[:, 25:37], R("state37")[:, 25:37], atol=2e-3, rtol=1e-5, err_msg=what + "teleported joint rates")
            np.testing.assert_allclose(F("TIME_OFFSET")[:, 0], R("time_offset"), atol=1e-6)
            np.testing.assert_allclose(F("ORIGIN_POS"), R("origin_pos"), atol=2e-6)
            np.testing.assert_allclose(F("ORIGIN_ROT"), R("origin_rot"), atol=2e-6)
            np.testing.assert_allclose(F("REF_POSE"), R("ref_pose"), atol=5e-6)
            np.testing.assert_allclose(F("LATENCY")[:, 0], R("latency"), atol=1e-8)
            np.testing.assert_allclose(F("STRENGTH"), R("strength"), atol=1e-6)
            if rnd:
                np.testing.assert_allclose(F("MASS_RATIO"), R("mass_ratio")[:, [0, 2]], atol=1e-6)        # link -1 (base group), link 1 (leg group)
                np.testing.assert_allclose(F("INERTIA_RATIO"), R("inertia_ratio")[:, [0, 2]], atol=1e-6)
                np.testing.assert_allclose(F("FOOT_MU")[:, 0], R("lateral_friction")[:, 3], atol=1e-6)    # link 2 = a lower leg
                np.testing.assert_allclose(F("KNEE_FRICTION"), R("joint_friction_force")[:, [2, 6, 10, 14]], atol=1e-7)
        else:
            S = lambda key: g["step/" + key][idx]
            eff = np.stack([S("eff_sim"), S("eff_ref")], axis=1)           # [n, 2, 8, 3]
            fall = torch.tensor(S("fall").astype(np.uint8), device=dev)
            obs, rew, done = env.replay_step(f32(S("action")), f32(traj[idx]), f32(eff), fall, tau_out)
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)
            what = "step %d " % idx
            tau = tau_out.cpu().numpy().astype(np.float64) * mdir[None, None, :]
            ref_tau = S("tau_urdf")[:, :, jom]
            np.testing.assert_allclose(tau, ref_tau, atol=2e-3, rtol=2e-5, err_msg=what + "motor torques")
            ro = S("obs")
            np.testing.assert_allclose(obs[:, 0:12].reshape(n, 3, 4)[:, :, 0:2], ro[:, 0:12].reshape(n, 3, 4)[:, :, 0:2], atol=1e-5, err_msg=what + "IMU roll / pitch")
            np.testing.assert_allclose(obs[:, 0:12].reshape(n, 3, 4)[:, :, 2:4], ro[:, 0:12].reshape(n, 3, 4)[:, :, 2:4], atol=1e-3, rtol=1e-5, err_msg=what + "IMU rates")
            np.testing.assert_allclose(obs[:, 12:], ro[:, 12:], atol=1e-5, err_msg=what + "last actions / motor angles / target frames")
            np.testing.assert_allclose(rew, S("reward"), atol=5e-6, err_msg=what + "reward")
            np.testing.assert_array_equal(done, S("done").astype(bool), err_msg=what + "done")
            np.testing.assert_allclose(env.field("ORIGIN_POS").cpu().numpy(), S("origin_pos"), atol=5e-6, err_msg=what + "origin (cycle sync)")
            np.testing.assert_allclose(env.field("REF_POSE").cpu().numpy(), S("ref_pose"), atol=1e-5, err_msg=what + "reference pose")
            worst["tau"] = max(worst["tau"], float(np.abs(tau - ref_tau).max()))
            worst["obs"] = max(worst["obs"], float(np.abs(obs[:, 12:] - ro[:, 12:]).max()))
            worst["rew"] = max(worst["rew"], float(np.abs(rew - S("reward")).max()))
            if done.any():
                count += n     # wrapper_env.py:82-83
    print("GOLDEN_REPLAY %s worst |d tau| %.2e  |d obs| %.2e  |d reward| %.2e" % (name, worst["tau"], worst["obs"], worst["rew"]))
    env.close()


@pytest.mark.parametrize("name", ["task_laikago.npz", "task_mini_cheetah.npz", "task_laikago_testmode.npz", "task_laikago_spin.npz"])
def test_hip_path_reproduces_the_reference_python(name):
    _replay(name)
