"""All 11 shipped motion clips through the KERNEL's frame sampler.

The host loader and the oracle are golden-pinned on every clip (tests/golden/clips.npz, from the reference's motion_data.py);
the kernel's float32 sampling arithmetic (task/motion_data.py:682-718 calc_blend_idx, :417-449 calc_frame, :591-633 cycle offsets,
:451-476 calc_frame_vel; device code: csrc/orr_task.h clip_index / sample_poses) had only seen pace / trot / spin.  The other
clips have different frame durations (hopturn 1/24 s, inplace_steps 0.03 s), lengths (91-208 frames) and cycle times (up to 7 s).
Per clip: (1) reset parity, (2) three env steps from a state copied across before every step, with time offsets that put
t + 30 * 0.033 s and t itself across the cycle wrap, times ON frame boundaries, and motion times of 3 s and 19 s.
"""
import os

import numpy as np
import pytest

from openroborl_amd import _abi, motion, state as statemod
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

CLIPS = sorted(f[:-4] for f in os.listdir(os.path.join(os.path.dirname(motion.__file__), "data", "motions")) if f.endswith(".txt"))


def _pair(clip, n, seed):
    from openroborl_amd.env import VecQuadrupedEnv
    robot = "mini_cheetah" if clip.startswith("minicheetah") else "laikago"
    env = VecQuadrupedEnv(num_robot=n, seed=seed, robot=robot, motion_file=clip, mode="test", enable_randomizer=False, auto_reset=False)
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=8)
    return env, orc


def _g64(env):
    return statemod.to_float64(env.layout, env.state.detach().cpu().numpy())


def test_all_shipped_clips_are_covered():
    assert len(CLIPS) == 11 and "laikago_hopturn" in CLIPS and "laikago_inplace_steps" in CLIPS and "minicheetah_trot" in CLIPS


@pytest.mark.parametrize("clip", CLIPS)
def test_clip_reset_parity(clip):
    """Reference-state initialisation at U(0, duration) time offsets: observation (incl. the four target frames), reference pose /
    velocity and origin, at the tolerances of test_reset_parity."""
    env, orc = _pair(clip, 128, seed=21)
    og = env.reset().cpu().numpy()
    oo = orc.reset()
    # The start time U(0, dur) is STATE: float32 on the device, float64 in the oracle, i.e. up to ulp32(dur) apart; a time error dt moves
    # the blend factor by dt / FrameDuration, and the result by that times the frame-to-frame jump of the quantity (the finite-difference
    # velocities of the clips jump by up to 150 rad/s where a frame pair straddles a quaternion sign flip).  Everything else: 2e-6 / 5e-5.
    c = env.clips[0]
    s = 2.0 * (c.duration * 6e-8) / c.frame_duration
    jp = np.abs(np.diff(c.frames, axis=0)).max()
    jv = np.abs(np.diff(c.frame_vels, axis=0)).max(axis=0)
    jl, ja, jj = jv[0:3].max(), jv[3:6].max(), jv[6:].max()
    np.testing.assert_allclose(og, oo, atol=5e-6 + s * jp, err_msg=clip)
    g = _g64(env)
    vel_tol = 5e-5 + s * np.concatenate([[jl] * 3, [ja] * 3, [jj] * 12])
    for name, tol in (("TIME_OFFSET", 1e-6), ("ORIGIN_POS", 2e-6 + s * jp), ("ORIGIN_ROT", 2e-6 + s * jp), ("PREV_PHASE", 2e-6),
                      ("REF_POSE", 2e-6 + s * jp), ("POS", 2e-6 + s * jp), ("QUAT", 2e-6 + s * jp), ("Q", 2e-6 + s * jp),
                      ("REF_VEL", vel_tol), ("QD", 5e-5 + s * jj), ("LINVEL", 5e-5 + s * jl), ("ANGVEL", 5e-5 + s * ja)):
        sl = env.layout.sl(name)
        err = np.abs(g[:, sl] - orc.state[:, sl])
        assert (err <= tol).all(), "%s %s: max err %.3g (tol %s) at %s" % (clip, name, err.max(), np.max(tol), np.argwhere(err > tol)[:4].tolist())
    env.close(); orc.close()


def engineered_times(env, st):
    """Motion times that exercise the sampler.  rows 0..31 keep the reset's U(0, dur) offsets; 32..47: the furthest target frame
    (30 * 0.033 s) and then the nearer ones cross the cycle wrap during the three steps; 48..63: the update time itself wraps (cycle
    sync fires; kept >= 2.5 ms away from the wrap itself: WHEN the sync fires is a knife edge between float32 and float64, and the
    re-anchored origin is a different one a step later); 64..79: times exactly ON frame boundaries (int(phase * (F - 1)) may land on
    either side in float32); 80..87: 3.3 s into an episode; 88..95: 19.5 s into an episode (step 590 of 600), large float32 t / dur."""
    c, lay, n = env.clips[0], env.layout, st.shape[0]
    dur, fdt = c.frame_duration * (c.num_frames - 1), c.frame_duration
    warm = st[:, lay.sl("WARMUP")][:, 0] > 0
    off = st[:, lay.sl("TIME_OFFSET")][:, 0].copy()
    cnt = np.zeros(n, dtype=np.int64)
    k = np.arange(16)
    off[32:48] = np.mod(dur - 0.99 - 0.004 * k, dur)
    off[48:64] = dur - 0.033 * (1 + k % 3) + 0.002 * (1 + k // 3) + 0.0005   # wraps in step 1 + k % 3, 2.5-12.5 ms past the end
    off[64:80] = np.float32(fdt) * ((3 + 5 * k) % (c.num_frames - 1))
    cnt[80:88] = 33 * 100
    cnt[88:96] = 33 * 590
    off[warm] = st[warm, lay.sl("TIME_OFFSET").start]          # warm-up episodes keep their small offset (imitation_task.py:1103-1110)
    st[:, lay.sl("TIME_OFFSET").start] = np.float32(off)
    st[:, lay.sl("STATE_ACTION_COUNTER").start] = cnt
    return st


@pytest.mark.parametrize("clip", CLIPS)
def test_clip_sampler_steps_across_wraps_and_frame_boundaries(clip):
    import torch
    n = 96
    env, orc = _pair(clip, n, seed=22)
    env.reset(); orc.reset()
    c = env.clips[0]
    dur, fdt = c.frame_duration * (c.num_frames - 1), c.frame_duration
    lay = env.layout
    st = engineered_times(env, _g64(env))
    warm = st[:, lay.sl("WARMUP")][:, 0] > 0
    cnt = st[:, lay.sl("STATE_ACTION_COUNTER")][:, 0].astype(np.int64)
    env.state.copy_(torch.from_numpy(statemod.from_float64(lay, st)).to(env.device))
    rng = np.random.RandomState(5)
    worst = {}
    t_grp = np.where(cnt > 10000, 2, np.where(cnt > 0, 1, 0))
    wrapped_any = np.zeros(n, dtype=bool)
    for step in range(3):
        orc.state[:] = _g64(env)            # the physics between the two sides is not what this test is about
        before = orc.state.copy()
        a = rng.uniform(-0.1, 0.1, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        og = og.cpu().numpy()
        g = _g64(env)
        # a phase wrap re-anchors the origin to the SIMULATED robot (imitation_task.py:1047-1053): there the reference pose inherits the
        # one-env-step float32 error of the physics (2e-4, test_step_parity); elsewhere it is pure sampling arithmetic
        wrapped = np.abs(orc.state[:, lay.sl("ORIGIN_POS")] - before[:, lay.sl("ORIGIN_POS")]).max(axis=1) > 1e-9
        wrapped_any |= wrapped
        # the kernel keeps the motion time in float64 (csrc/orr_device.h DevClip): the same tolerance 19 s into an episode as at its start
        # (with a float32 time the blend factor is off by ulp(t) / frame_dt: 4e-4 on the pose and 1e-2 on the frame velocities at 20 s)
        pose_tol = np.array([2e-5, 2e-5, 2e-5])[t_grp] + 3e-4 * wrapped
        vel_tol = np.array([2e-4, 2e-4, 2e-4])[t_grp] + 2e-2 * wrapped
        for name, tol in (("REF_POSE", pose_tol), ("ORIGIN_POS", 3e-4 * wrapped + 1e-9), ("ORIGIN_ROT", 1e-6 + 0 * pose_tol),
                          ("PREV_PHASE", np.array([2e-6, 2e-6, 2e-6])[t_grp]), ("REF_VEL", vel_tol)):
            sl = lay.sl(name)
            err = np.abs(g[:, sl] - orc.state[:, sl]).max(axis=1)
            if name == "PREV_PHASE":        # a phase within float32 rounding of the wrap may read 0.99999 on one side and 0.00001 on the other
                err = np.minimum(err, 1.0 - err)
            worst[name] = max(worst.get(name, 0.0), float((err / tol).max()))
            assert (err <= tol).all(), "%s step %d %s: robots %s err %s tol %s" % (clip, step, name, np.nonzero(err > tol)[0][:8],
                                                                                   err[err > tol][:8], np.broadcast_to(tol, err.shape)[err > tol][:8])
        # target observation: four future frames relative to the current reference root, in the robot's heading frame
        # (imitation_task.py:254-301); heading comes from the delayed IMU reading = physics: the 1-step tolerance of test_step_parity
        err = np.abs(og[:, 84:] - oo[:, 84:]).max(axis=1)
        tol = 5e-4 + pose_tol
        worst["target_obs"] = max(worst.get("target_obs", 0.0), float((err / tol).max()))
        assert (err <= tol).all(), "%s step %d target obs: robots %s err %s" % (clip, step, np.nonzero(err > tol)[0][:8], err[err > tol][:8])
        np.testing.assert_allclose(rg.cpu().numpy(), ro, atol=3e-3, err_msg=clip)
    if c.flags & _abi.CLIP_WRAP:
        assert wrapped_any[48:64][~warm[48:64]].all(), "the engineered update-time wraps did not happen"
    print("CLIP_SAMPLER %s F=%d frame_dt=%.5f dur=%.4f worst err/tol: %s" % (
        clip, c.num_frames, fdt, dur, " ".join("%s=%.2f" % kv for kv in sorted(worst.items()))))
    env.close(); orc.close()
