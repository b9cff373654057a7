#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's importable pieces.

Run ONLY in the build container (needs /root/reference; the GPU box never has it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference (read-only, never copied):
  * envs/quadruped_robot/task/motion_data.py      (MotionData: E1-E4 of SURVEY.md section 8a)
  * envs/utilities/pose3d.py                      (G)
  * envs/utilities/action_filter.py               (B2)
  * envs/quadruped_robot/robots/minitaur_motor.py (B5)
  * envs/utilities/sensors/*.py                   (D3 / A5 history + ordering)
and what is only *read as data*: task/motions/*.txt (JSON) and task/policies/*.zip.

Two import shims live in tests/golden/_shims (absl.logging -> stdlib logging, and a restatement
of pybullet_utils.transformations; see the caveat in that file and in DESIGN.md).  Everything
that needs a pybullet client object (quadruped_gym_env, wrapper_env, minitaur, imitation_task, the
randomiser) is driven by make_golden_task.py with a scripted client instead.

Outputs (committed):
  clips.npz        per clip: raw frames, processed frames, frame velocities, scalars, and
                   calc_frame / calc_frame_vel / calc_blend_idx samples at fixed times
  pose3d.npz       quaternion helper vectors
  filter.npz       Butterworth coefficients + a filtered sequence incl. init_history
  motor.npz        PD torque vectors
  sensors.npz      3-deep history layout of the flattened 84-d proprioceptive observation
  spaces.npz       160-d observation-space bounds + action bounds pickled inside the policy zips
  policy_*.npz     shipped MLP weights (data) for the behavioural probe
  policy_parameter_list.json   variable names + shapes of a shipped policy zip, in saved order
"""
import base64
import io
import json
import os
import pickle
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/OpenRoboRL"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, REF)

from envs.quadruped_robot.task import motion_data  # noqa: E402
from envs.utilities import pose3d, action_filter  # noqa: E402
from envs.quadruped_robot.robots import minitaur_motor  # noqa: E402
from envs.utilities.sensors import sensor_wrappers, robot_sensors, environment_sensors  # noqa: E402

MOTIONS = os.path.join(REF, "envs/quadruped_robot/task/motions")
POLICIES = os.path.join(REF, "envs/quadruped_robot/task/policies")


def gen_clips():
    out = {}
    names = sorted(f[:-4] for f in os.listdir(MOTIONS) if f.endswith(".txt"))
    rng = np.random.RandomState(1234)
    for name in names:
        path = os.path.join(MOTIONS, name + ".txt")
        with open(path) as f:
            raw = json.load(f)
        m = motion_data.MotionData(path)
        dur = m.get_duration()
        times = np.concatenate([
            np.array([0.0, 1e-4, 0.033, 0.1, 0.5 * dur, dur - 1e-6, dur, dur + 1e-6, 1.0, 2.5 * dur,
                      7.3, -0.01, -0.25, -1.3 * dur]),
            rng.uniform(-2 * dur, 6 * dur, size=18)])
        frames_t = np.stack([m.calc_frame(t) for t in times])
        vels_t = np.stack([m.calc_frame_vel(t) for t in times])
        idx = np.array([m.calc_blend_idx(t) for t in times], dtype=np.float64)
        phase = np.array([m.calc_phase(t) for t in times])
        count = np.array([m.calc_cycle_count(t) for t in times], dtype=np.float64)
        p = name + "/"
        out[p + "raw_frames"] = np.array(raw["Frames"], dtype=np.float64)
        out[p + "frame_duration"] = np.float64(raw["FrameDuration"])
        out[p + "loop_wrap"] = np.float64(raw["LoopMode"] == "Wrap")
        out[p + "cycle_pos"] = np.float64(bool(raw.get("EnableCycleOffsetPosition", False)))
        out[p + "cycle_rot"] = np.float64(bool(raw.get("EnableCycleOffsetRotation", False)))
        out[p + "frames"] = m.get_frames().copy()
        out[p + "frame_vels"] = m._frame_vels.copy()
        out[p + "duration"] = np.float64(dur)
        out[p + "cycle_delta_pos"] = m._cycle_delta_pos.copy()
        out[p + "cycle_delta_heading"] = np.float64(m._cycle_delta_heading)
        out[p + "times"] = times
        out[p + "calc_frame"] = frames_t
        out[p + "calc_frame_vel"] = vels_t
        out[p + "blend_idx"] = idx
        out[p + "phase"] = phase
        out[p + "cycle_count"] = count
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "clips.npz"), **out)
    return names


def gen_pose3d():
    rng = np.random.RandomState(7)
    q = rng.randn(64, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # a few special cases: identity, laikago init, pure yaw, w<0, tiny rotation
    q[0] = [0, 0, 0, 1]
    q[1] = [0.5, 0.5, 0.5, 0.5]
    q[2] = [0, 0, np.sin(0.4), np.cos(0.4)]
    q[3] = -q[2]
    q[4] = [1e-10, 0, 0, 1.0]
    q[4] /= np.linalg.norm(q[4])
    p = rng.randn(64, 3)
    rot = np.stack([pose3d.QuaternionRotatePoint(p[i], q[i]) for i in range(64)])
    axis = np.zeros((64, 3))
    angle = np.zeros(64)
    for i in range(64):
        a, th = pose3d.QuaternionToAxisAngle(q[i])
        axis[i] = a
        angle[i] = th
    heading = np.array([pose3d.calc_heading(q[i]) for i in range(64)])
    heading_rot = np.stack([pose3d.calc_heading_rot(q[i]) for i in range(64)])
    std = np.stack([pose3d.standardize_quaternion(q[i].copy()) for i in range(64)])
    theta = np.concatenate([np.linspace(-9.5, 9.5, 39), [np.pi, -np.pi, 2 * np.pi, -2 * np.pi, 0.0]])
    norm_theta = np.array([pose3d.normalize_rotation_angle(t) for t in theta])
    ang = np.concatenate([np.linspace(-13.0, 13.0, 53), [np.pi, -np.pi, 3 * np.pi, -3 * np.pi]])
    mapped = np.array(pose3d.MapToMinusPiToPi(list(ang)))
    np.savez_compressed(os.path.join(HERE, "pose3d.npz"), q=q, p=p, rotate_point=rot, axis=axis,
                        angle=angle, heading=heading, heading_rot=heading_rot, standardize=std,
                        theta=theta, normalize_rotation_angle=norm_theta, map_in=ang,
                        map_to_minus_pi_to_pi=mapped)


def gen_filter():
    rng = np.random.RandomState(3)
    f = action_filter.ActionFilterButter(sampling_rate=1.0 / (0.001 * 33), num_joints=12)
    b, a = f.b[0].copy(), f.a[0].copy()
    f.reset()
    x = rng.uniform(-1.5, 1.5, size=(40, 12))
    init = rng.uniform(-1.0, 1.0, size=12)
    f.init_history(init)
    y = np.stack([f.filter(x[i]) for i in range(40)])
    # second sequence: zero history (reset() semantics)
    f.reset()
    y0 = np.stack([f.filter(x[i]) for i in range(10)])
    np.savez_compressed(os.path.join(HERE, "filter.npz"), b=b, a=a, x=x, init=init, y=y, y_zero_hist=y0)


def gen_motor():
    rng = np.random.RandomState(5)
    out = {}
    for name, kp, kd in (("laikago", [220.0] * 12, [0.3, 2.0, 2.0] * 4),
                         ("mini_cheetah", [80.0] * 12, [0.1, 1.0, 1.0] * 4)):
        mm = minitaur_motor.MotorModel(kp=np.array(kp), kd=np.array(kd), torque_limits=None)
        cmd = rng.uniform(-2, 2, size=(16, 12))
        q = rng.uniform(-2, 2, size=(16, 12))
        qd = rng.uniform(-20, 20, size=(16, 12))
        strength = rng.uniform(0.8, 1.2, size=(16, 12))
        tau = np.zeros((16, 12))
        for i in range(16):
            mm.set_strength_ratios(strength[i])
            tau[i], _ = mm.convert_to_torque(cmd[i], q[i], qd[i], qd[i])
        out[name + "/cmd"] = cmd
        out[name + "/q"] = q
        out[name + "/qd"] = qd
        out[name + "/strength"] = strength
        out[name + "/tau"] = tau
    # SURVEY appendix-B known answer
    mm = minitaur_motor.MotorModel(kp=np.array([220.0] * 12), kd=np.array([0.3, 2.0, 2.0] * 4),
                                   torque_limits=None)
    out["kat/tau"], _ = mm.convert_to_torque(np.zeros(12), np.linspace(-0.5, 0.5, 12),
                                              np.linspace(1, -1, 12), None)
    np.savez_compressed(os.path.join(HERE, "motor.npz"), **out)


class _StubRobot(object):
    """Feeds the sensor objects with scripted readings (robot_sensors.py:74-83,153-190)."""

    def __init__(self, rng):
        self.rng = rng
        self.advance()

    def advance(self):
        self.angles = self.rng.uniform(-3, 3, 12)
        self.rpy = self.rng.uniform(-1, 1, 3)
        self.drpy = self.rng.uniform(-5, 5, 3)
        self.last_action = self.rng.uniform(-2, 2, 12)

    def get_motor_angles(self):
        return self.angles

    def get_base_rpy(self):
        return self.rpy

    def get_base_rpy_rate(self):
        return self.drpy


def gen_sensors():
    rng = np.random.RandomState(11)
    robot = _StubRobot(rng)
    sensors = [
        sensor_wrappers.HistoricSensorWrapper(
            wrapped_sensor=robot_sensors.MotorAngleSensor(num_motors=12), num_history=3),
        sensor_wrappers.HistoricSensorWrapper(wrapped_sensor=robot_sensors.IMUSensor(), num_history=3),
        sensor_wrappers.HistoricSensorWrapper(
            wrapped_sensor=environment_sensors.LastActionSensor(num_actions=12), num_history=3)]
    for s in sensors:
        s.set_robot(robot)

    def flat():
        # same rule as Minitaur._get_observation (minitaur.py:529-541) +
        # LocomotionGymEnv._flatten_observation (quadruped_gym_env.py:289-320):
        # dict sorted by sensor name, values concatenated.
        d = {s.get_name(): s.get_observation() for s in sensors}
        return np.concatenate([np.asarray(d[k]).flatten() for k in sorted(d)])

    feeds = []
    obs = []
    feeds.append(np.concatenate([robot.angles, robot.rpy, robot.drpy, robot.last_action]))
    for s in sensors:
        s.on_reset(robot)
    obs.append(flat())
    for _ in range(6):
        robot.advance()
        feeds.append(np.concatenate([robot.angles, robot.rpy, robot.drpy, robot.last_action]))
        for s in sensors:
            s.on_step()
        obs.append(flat())
    names = sorted(s.get_name() for s in sensors)
    low = np.concatenate([np.asarray(s.get_lower_bound()).flatten()
                          for s in sorted(sensors, key=lambda s: s.get_name())])
    high = np.concatenate([np.asarray(s.get_upper_bound()).flatten()
                           for s in sorted(sensors, key=lambda s: s.get_name())])
    np.savez_compressed(os.path.join(HERE, "sensors.npz"), feeds=np.stack(feeds), obs=np.stack(obs),
                        names=np.array(names), low=low, high=high)


class _Stub(object):
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("numpy"):
            return super().find_class(module, name)
        return type(name, (_Stub,), {})


def _clip_target_bounds(name):
    """ImitationTask.get_target_obs_bounds (imitation_task.py:303-335) over the reference's own MotionData of one clip: root position
    +-2, root rotation +-1, joint min / max over the frames, tiled over the four target frames."""
    m = motion_data.MotionData(os.path.join(MOTIONS, name + ".txt"))
    fr = np.asarray(m.get_frames())
    jl, jh = fr[:, 7:].min(axis=0), fr[:, 7:].max(axis=0)
    low = np.concatenate([-2 * np.ones(3), -np.ones(4), jl])
    high = np.concatenate([2 * np.ones(3), np.ones(4), jh])
    return np.tile(low, 4), np.tile(high, 4)


def gen_spaces_and_policies():
    out = {}
    clips = sorted(f[:-4] for f in os.listdir(MOTIONS) if f.endswith(".txt"))
    tb = {c: _clip_target_bounds(c) for c in clips}
    match = {}
    for pol in sorted(f[:-4] for f in os.listdir(POLICIES) if f.endswith(".zip")):
        with zipfile.ZipFile(os.path.join(POLICIES, pol + ".zip")) as z:
            data = json.loads(z.read("data"))
            for key in ("observation_space", "action_space"):
                ser = data[key][":serialized:"]
                obj = _Unpickler(io.BytesIO(base64.b64decode(ser))).load()
                out["%s/%s/low" % (pol, key)] = np.asarray(obj.__dict__["low"])
                out["%s/%s/high" % (pol, key)] = np.asarray(obj.__dict__["high"])
            # which clip was this policy trained on?  The zips carry no clip name; the 76 target-observation bounds pickled in them
            # are those of the training clip (imitation_task.py:303-335, wrapper_env.py:127-145).  Every clip whose bounds agree to
            # float32 is recorded (a clip and its time reversal have the same bounds); the zip's own name picks among them.
            lo, hi = out["%s/observation_space/low" % pol][84:], out["%s/observation_space/high" % pol][84:]
            err = {c: float(max(np.abs(lo - tb[c][0]).max(), np.abs(hi - tb[c][1]).max())) for c in clips}
            same = sorted(c for c in clips if err[c] < 1e-6)
            named = [c for c in same if pol.rstrip("0123456789") == c]
            match[pol] = {"clip": (named or same)[0], "clips_with_equal_bounds": same,
                          "max_abs_bound_difference": err[(named or same)[0]],
                          "next_best": sorted((e, c) for c, e in err.items() if c not in same)[0][::-1]}
            params = np.load(io.BytesIO(z.read("parameters")))
            if pol == "laikago_pace":
                # variable names, in the order stable-baselines saved them, with shapes (data): what a zip must contain for
                # BaseRLModel.load_parameters(exact_match=True) (stable_baselines/common/base_class.py:437-500) to accept it
                plist = json.loads(z.read("parameter_list"))
                with open(os.path.join(HERE, "policy_parameter_list.json"), "w") as f:
                    json.dump([[k, list(params[k].shape)] for k in plist], f, indent=0)
            w = {k.replace("/", "__").replace(":", "_"): params[k].astype(np.float32) for k in params.files
                 if k.startswith("model/pi") }
            np.savez_compressed(os.path.join(HERE, "policy_%s.npz" % pol), **w)
    with open(os.path.join(HERE, "policy_clips.json"), "w") as f:
        json.dump(match, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "spaces.npz"), **out)


if __name__ == "__main__":
    names = gen_clips()
    gen_pose3d()
    gen_filter()
    gen_motor()
    gen_sensors()
    gen_spaces_and_policies()
    print("golden fixtures written for clips:", names)
