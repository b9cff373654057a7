"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the
CPU oracle on identical seeded inputs.  Tolerances: the device computes in float32, the oracle in float64;
one physics sub-step agrees to 2e-6 on positions / 1.5e-4 on velocities, a full env step (33 sub-steps with contacts) to ~1e-3 on states and
2e-3 on rewards; longer rollouts are compared statistically (contact dynamics amplify rounding)."""
import os

import numpy as np
import pytest

from openroborl_amd import _abi, config, motion, robots, state as statemod
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

CLIP = {"laikago": "laikago_pace", "mini_cheetah": "minicheetah_trot"}


def make_pair(robot="laikago", n=32, randomizer=False, auto_reset=False, seed=3, mode="test", mixed=None, legacy_grid=False,
              model_overrides=None):
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    if mixed:
        env = VecQuadrupedEnv(num_robot=n, seed=seed, mode=mode, enable_randomizer=randomizer, auto_reset=auto_reset,
                              mixed_robots=mixed, motion_file=[CLIP[m] for m in mixed], legacy_grid=legacy_grid,
                              model_overrides=model_overrides)
    else:
        env = VecQuadrupedEnv(num_robot=n, seed=seed, robot=robot, motion_file=CLIP[robot], mode=mode,
                              enable_randomizer=randomizer, auto_reset=auto_reset, legacy_grid=legacy_grid,
                              model_overrides=model_overrides)
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=8)
    return env, orc


def f32_twin(env, orc):
    """The float32 build of the oracle (make -C oracle f32) in the state of `orc`: its distance from `orc` after the same actions is the
    float32 NOISE FLOOR the HIP path is measured against (tests/drift.py, tests/test_gpu_drift.py) instead of hand-set tolerances."""
    o32 = ol.OracleEnv(env.cfg, env.models, env.clips, orc.n, robot_type=env.robot_type, clip_id=env.clip_id, threads=8, f32="parity")
    o32.reset()
    o32.state[:] = orc.state.astype(np.float32)
    return o32


def compare_to_floor(env, orc, o32, names, what, mask=None):
    from tests import drift
    g = gpu_state64(env)
    m = slice(None) if mask is None else mask
    for name in names:
        sl = env.layout.sl(name)
        drift.assert_within_float32_floor(np.abs(g[m][:, sl] - orc.state[m][:, sl]).max(axis=1),
                                          np.abs(o32.state[m][:, sl].astype(np.float64) - orc.state[m][:, sl]).max(axis=1), "%s %s" % (what, name))


def gpu_state64(env):
    return statemod.to_float64(env.layout, env.state.detach().cpu().numpy())


def push_state(env, st64):
    import torch
    env.state.copy_(torch.from_numpy(statemod.from_float64(env.layout, st64)).to(env.device))


def compare_fields(env, orc, names, atol, rtol=0.0, what="", outlier_robots=0, floor=None):
    """floor: the float32 build of the ORACLE (f32_twin) stepped like `orc` - a robot whose largest error stays within 1.5 x what the same
    algorithm in float32 does to that robot passes even beyond the hand-set tolerance (ill-conditioned inputs: a robot lying on its shanks
    with a 0.25 kg toe; measured round 6: float32 oracle 4.0e-4 on one joint rate of 384, HIP 3.0e-4, tolerance 2.2e-4).
    outlier_robots: that many robots may miss both - by at most 10 x the tolerance - at a multi-sub-step horizon, where a contact row
    that one side creates a sub-step earlier than the other (float32 distance against the contact margin) is a discrete event, not an
    error of the arithmetic; 0 (every one-sub-step comparison) = nobody."""
    g = gpu_state64(env)
    for name in names:
        sl = env.layout.sl(name)
        if env.layout.is_int(name):
            np.testing.assert_array_equal(g[:, sl], orc.state[:, sl], err_msg="%s %s" % (what, name))
        elif outlier_robots or floor is not None:
            e = np.abs(g[:, sl] - orc.state[:, sl])
            bad = (e > atol + rtol * np.abs(orc.state[:, sl])).any(axis=1)
            if floor is not None:
                e32 = np.abs(floor.state[:, sl].astype(np.float64) - orc.state[:, sl])
                bad &= e.max(axis=1) > 1.5 * e32.max(axis=1)
            assert bad.sum() <= outlier_robots, "%s %s: %d robots beyond tolerance and float32 floor (largest error %.3g)" % (what, name, bad.sum(), e[bad].max())
            if bad.any():
                np.testing.assert_allclose(g[bad][:, sl], orc.state[bad][:, sl], atol=10 * atol, rtol=10 * rtol, err_msg="%s %s (outlier bound)" % (what, name))
        else:
            np.testing.assert_allclose(g[:, sl], orc.state[:, sl], atol=atol, rtol=rtol, err_msg="%s %s" % (what, name))


RIGID = ["POS", "QUAT", "LINVEL", "ANGVEL", "Q", "QD"]
SOFT_TOES = {"contact_stiffness": 30000.0, "contact_damping": 1000.0, "foot_friction": 3.0}


ANCHOR_TOES = {"friction_anchor": 1}


@pytest.mark.parametrize("robot,soft", [("laikago", False), ("mini_cheetah", False), ("laikago", True), ("mini_cheetah", True),
                                        ("laikago", "anchor"), ("mini_cheetah", "anchor"), ("laikago", "anchor+soft")])
def test_physics_substep_parity(robot, soft):
    """Row C in isolation: ABA + contact/limit/friction rows + PGS + integration, fixed torques.  soft: the toe's normal rows with
    Bullet's contact stiffness / damping (cfm on the diagonal, own erp) and a friction coefficient of 3.  anchor (ABI v5): Bullet's
    friction anchors - the ANCHOR variant of the kernels against the oracle's cached contact points, over sub-step sequences long
    enough for points to be kept, replaced (friction impulse outside the cone) and dropped (lift-off, tangential drift)."""
    import torch
    from tests.parity_inputs import substep_parity_inputs
    n = 64
    anchor = isinstance(soft, str) and "anchor" in soft
    over = dict(SOFT_TOES if soft is True or (isinstance(soft, str) and "soft" in soft) else {})
    if anchor:
        over.update(ANCHOR_TOES)
    env, orc = make_pair(robot, n=n, model_overrides={robot: over} if over else None)
    env.reset(); orc.reset()
    _, _, _, st, tau = substep_parity_inputs(robot, n)       # the same seeded inputs tools/pybullet_ref.py feeds to PyBullet
    push_state(env, st); orc.state[:] = st
    tg = torch.tensor(tau, dtype=torch.float32, device=env.device)
    # one sub-step: positions agree to 2e-6 (they move by dt * velocity), velocities (up to ~20 rad/s, through the
    # articulated-body solve in float32) to 1.5e-4; after 8 sub-steps with contacts 1e-3
    for nsub, ptol, vtol in ((1, 2e-6, 1.5e-4), (8, 1e-4, 1e-3)) + (((16, 4e-4, 6e-3),) if anchor else ()):
        fall_g = env.debug_physics(tg, nsub).cpu().numpy()
        fall_o = np.zeros(n, dtype=int)
        for i in range(n):
            for _ in range(nsub):
                fall_o[i] = orc.L.orc_physics_substep(orc.h, ol.P(orc.state[i]), ol.P(np.ascontiguousarray(tau[i])))
        compare_fields(env, orc, ["POS", "QUAT", "Q"], atol=ptol, rtol=ptol, what="nsub=%d" % nsub)
        compare_fields(env, orc, ["LINVEL", "ANGVEL", "QD"], atol=vtol, rtol=vtol, what="nsub=%d" % nsub)
        compare_fields(env, orc, ["LAMBDA"], atol=5e-4 if nsub == 1 else (5e-3 if nsub == 8 else 2e-2), what="nsub=%d" % nsub)
        assert (fall_g.astype(int) == fall_o).mean() > 0.95
        if anchor:
            g = gpu_state64(env)
            lay = env.layout
            vg, vo = g[:, lay.sl("ANCHOR_VALID")], orc.state[:, lay.sl("ANCHOR_VALID")]
            np.testing.assert_array_equal(vg, vo, err_msg="cached contact points, nsub=%d" % nsub)
            ag, ao = g[:, lay.sl("ANCHOR")].reshape(n, 4, 6), orc.state[:, lay.sl("ANCHOR")].reshape(n, 4, 6)
            m = vo.astype(bool)
            np.testing.assert_allclose(ag[m], ao[m], atol=max(ptol, 1e-6) * 4, err_msg="anchor points, nsub=%d" % nsub)
            if nsub == 1:
                assert m.mean() > 0.05                     # a share of the toes holds a cached point (measured: Laikago 0.2+, mini-cheetah 0.14)
    if anchor:
        # the anchors did something: some toes kept a point that is NOT the fresh sphere-plane point any more (the link turned over it)
        lay = env.layout
        an = orc.state[:, lay.sl("ANCHOR")].reshape(n, 4, 6)
        vo = orc.state[:, lay.sl("ANCHOR_VALID")].astype(bool)
        assert vo.any()
    env.close(); orc.close()


@pytest.mark.parametrize("robot,randomizer", [("laikago", False), ("laikago", True), ("mini_cheetah", True)])
def test_reset_parity(robot, randomizer):
    env, orc = make_pair(robot, n=128, randomizer=randomizer, mode="train" if randomizer else "test")
    og = env.reset().cpu().numpy()
    oo = orc.reset()
    np.testing.assert_allclose(og, oo, atol=2e-6)
    # frame velocities jump by O(10) between clip frames, so the float32 blend factor shows up at ~1e-5 there
    vel_like = ("LINVEL", "ANGVEL", "QD", "REF_VEL")
    names = [f for f in env.layout.order if f not in ("RING", "RESERVED_I") + vel_like]
    compare_fields(env, orc, names, atol=2e-6, what="reset")
    compare_fields(env, orc, vel_like, atol=5e-5, what="reset")
    # ring: entries 0 and 1 were written
    g = gpu_state64(env)
    sl = env.layout.sl("RING")
    np.testing.assert_allclose(g[:, sl][:, :2 * _abi.RING_ENTRY], orc.state[:, sl][:, :2 * _abi.RING_ENTRY], atol=2e-6)
    env.close(); orc.close()


@pytest.mark.parametrize("robot,randomizer", [("laikago", False), ("laikago", True), ("mini_cheetah", False)])
def test_step_parity(robot, randomizer):
    import torch
    n = 256
    env, orc = make_pair(robot, n=n, randomizer=randomizer, mode="train" if randomizer else "test")
    env.reset(); orc.reset()
    # identical starting point on both sides (float32-representable)
    st = gpu_state64(env)
    orc.state[:] = st
    o32 = f32_twin(env, orc)
    rng = np.random.RandomState(5)
    act = torch.tensor(rng.uniform(-0.3, 0.3, (n, 12)), dtype=torch.float32, device=env.device)
    og, rg, dg, _ = env.step(act)
    oo, ro, do = orc.step(act.cpu().numpy().astype(np.float64))
    o3, r3, d3 = o32.step(act.cpu().numpy())
    og, rg, dg = og.cpu().numpy(), rg.cpu().numpy(), dg.cpu().numpy().astype(bool)
    # 33 sub-steps of contact dynamics amplify float32 rounding (the float32 ORACLE is up to 1e-2 rad/s off on the base rates and
    # 0.1-1 rad/s on the joint rates in the worst robot after ONE env step, tests/test_gpu_drift.py): the bound is that floor, measured
    # on the same robots, not a hand-set tolerance (round 2: atol 2e-4 / 2e-2)
    compare_to_floor(env, orc, o32, RIGID, "step")
    compare_fields(env, orc, ["ACTION", "FILTER_ACTION", "LAST_ACTION", "XHIST", "YHIST", "TIME_OFFSET", "ORIGIN_ROT",
                              "PREV_PHASE", "REF_POSE"], atol=5e-6, what="step")
    compare_fields(env, orc, ["STATE_ACTION_COUNTER", "STEP_COUNTER", "FILTER_VALID", "RING_LEN", "RING_HEAD", "EP_STEP",
                              "WARMUP", "MAX_EP_STEPS", "EPISODE_IDX"], atol=0, what="step")
    from tests import drift
    drift.assert_within_float32_floor(np.abs(rg - ro), np.abs(r3.astype(np.float64) - ro), "step reward")
    for what, sl in drift.OBS_GROUPS:                                              # IMU | last action + motor angles | target frames
        drift.assert_within_float32_floor(np.abs(og[:, sl] - oo[:, sl]).max(axis=1), np.abs(o3[:, sl].astype(np.float64) - oo[:, sl]).max(axis=1),
                                          "step " + what)
    assert (dg == do).mean() > 0.97
    env.close(); orc.close(); o32.close()


def test_env_step_with_friction_anchors_matches_the_oracle_and_resets_clear_them():
    """The ANCHOR variant of the ENV STEP kernel (orr_step_kernel<0, 1, true>: its own translation unit; load of the cached contact
    points at the start of a launch, the carry over 33 sub-steps in registers, the store by the normal-row lanes, the clearing by an inline
    reset) against the oracle: a full env step from an identical start (float32-floor bounds like test_step_parity), the same toes holding a
    cached point on both sides, the same points; then train mode until the 20-step time limit: a robot reset inside the launch has no
    cached point left, as on the oracle; and the standalone reset kernel clears them too."""
    import torch
    n = 256
    env, orc = make_pair("laikago", n=n, model_overrides={"laikago": ANCHOR_TOES})
    env.reset(); orc.reset()
    st = gpu_state64(env)
    orc.state[:] = st
    o32 = f32_twin(env, orc)
    rng = np.random.RandomState(5)
    act = torch.tensor(rng.uniform(-0.3, 0.3, (n, 12)), dtype=torch.float32, device=env.device)
    og, rg, dg, _ = env.step(act)
    oo, ro, do = orc.step(act.cpu().numpy().astype(np.float64))
    o32.step(act.cpu().numpy())
    compare_to_floor(env, orc, o32, RIGID, "anchor step")
    g = gpu_state64(env)
    lay = env.layout
    vg, vo = g[:, lay.sl("ANCHOR_VALID")], orc.state[:, lay.sl("ANCHOR_VALID")]
    assert (vg == vo).mean() > 0.97 and vo.mean() > 0.2            # 33 un-synced sub-steps of contact chaos: a toe or two may differ
    same = (vg == vo) & (vo > 0)
    ag, ao = g[:, lay.sl("ANCHOR")].reshape(n, 4, 6), orc.state[:, lay.sl("ANCHOR")].reshape(n, 4, 6)
    assert np.median(np.abs(ag[same] - ao[same])) < 1e-5
    env.close(); orc.close(); o32.close()
    # inline reset and reset kernel clear the cached points
    env, orc = make_pair("laikago", n=64, randomizer=True, auto_reset=True, mode="train", seed=17, model_overrides={"laikago": ANCHOR_TOES})
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    limit = int(env.field_int("MAX_EP_STEPS").max())
    rng = np.random.RandomState(1)
    for k in range(limit):
        a = rng.uniform(-0.05, 0.05, (64, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        if k == limit - 2:
            assert env.field_int("ANCHOR_VALID").any()              # standing robots hold cached points right up to the reset
    done = dg.cpu().numpy().astype(bool)
    assert done.sum() > 32
    g = gpu_state64(env)
    assert not g[done][:, env.layout.sl("ANCHOR_VALID")].any() and not g[done][:, env.layout.sl("ANCHOR")].any()
    assert not orc.state[done & do][:, env.layout.sl("ANCHOR_VALID")].any()
    env.step(torch.zeros(64, 12, device=env.device))
    env.reset()
    assert not env.field_int("ANCHOR_VALID").any() and not env.field("ANCHOR").any()
    env.close(); orc.close()


def test_short_rollout_tracks_oracle():
    """10 env steps from an identical start: trajectories stay close (loose: contact dynamics amplify rounding)."""
    import torch
    n = 256       # quantiles of a heavy-tailed error distribution: 32 robots are too few for a stable median
    env, orc = make_pair("laikago", n=n)
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    o32 = f32_twin(env, orc)
    rng = np.random.RandomState(9)
    alive = np.ones(n, dtype=bool)
    for k in range(10):
        a = rng.uniform(-0.2, 0.2, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        o3, r3, d3 = o32.step(a)
        if k < 9:
            alive &= ~(dg.cpu().numpy().astype(bool) | do | d3)
    assert alive.mean() > 0.8
    from tests import drift
    drift.assert_within_float32_floor(np.abs(rg.cpu().numpy() - ro)[alive], np.abs(r3.astype(np.float64) - ro)[alive], "10-step reward")
    compare_to_floor(env, orc, o32, RIGID + ["REF_POSE"], "10 steps", mask=alive)
    env.close(); orc.close(); o32.close()


@pytest.mark.parametrize("soft_type", [None, "mini_cheetah", "anchor:mini_cheetah"])
def test_mixed_batch_parity(soft_type):
    """BASELINE config 5: interleaved Laikago / mini-cheetah robots in one launch (divergent wavefronts).  soft_type: that robot type's
    toes with Bullet's contact stiffness / damping (the Laikago table's are soft since round 5), the other type as shipped - two contact
    models side by side in every wavefront; "anchor:<type>": that type's toes with friction anchors, i.e. the ANCHOR variant of the step
    kernel serving robots with and without cached contact points in the same wavefront."""
    import torch
    n = 64
    over = None
    if soft_type and soft_type.startswith("anchor:"):
        over = {soft_type.split(":")[1]: ANCHOR_TOES}
    elif soft_type:
        over = {soft_type: SOFT_TOES}
    env, orc = make_pair(n=n, mixed=["laikago", "mini_cheetah"], model_overrides=over)
    og = env.reset().cpu().numpy()
    oo = orc.reset()
    np.testing.assert_allclose(og, oo, atol=2e-6)
    orc.state[:] = gpu_state64(env)
    a = np.random.RandomState(1).uniform(-0.2, 0.2, (n, 12)).astype(np.float32)
    og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
    oo, ro, do = orc.step(a.astype(np.float64))
    np.testing.assert_allclose(rg.cpu().numpy(), ro, atol=3e-3)
    np.testing.assert_allclose(og.cpu().numpy()[:, 84:], oo[:, 84:], atol=5e-4)
    if soft_type and soft_type.startswith("anchor:"):
        g = gpu_state64(env)
        v = g[:, env.layout.sl("ANCHOR_VALID")]
        t = env.robot_type
        anchored = t == robots.ROBOT_TYPE_ID[soft_type.split(":")[1]]
        assert v[anchored].any() and not v[~anchored].any()                       # only the type with anchors ever caches a point
        assert (v == orc.state[:, env.layout.sl("ANCHOR_VALID")]).mean() > 0.97
    env.close(); orc.close()


def test_shipped_policy_on_gpu():
    """Behavioural probe on the HIP path: the reference's laikago_pace policy keeps 64 robots on the clip for the full 600-step episode.
    STATUS OF THIS ANCHOR: on round 4's Laikago table this policy had been looked at while the table was written (an in-sample anchor);
    on the table shipped since round 5 it is one of the two policies HELD OUT by the identification's protocol (tools/laikago_identify.py
    fitted on laikago_trot + laikago_spin only).  Not a proof of physics parity either way: see tests/test_gpu_policies.py."""
    import torch
    W = np.load(os.path.join(ol.GOLDEN, "policy_laikago_pace.npz"))
    n = 64
    env, orc = make_pair("laikago", n=n, seed=1)
    orc.close()
    dev = env.device
    w = {k: torch.tensor(W[k], device=dev) for k in W.files}
    obs = env.reset()
    ret = torch.zeros(n, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    for step in range(600):
        h = torch.relu(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"])
        h = torch.relu(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"])
        a = torch.clamp(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
        obs, rew, done, _ = env.step(a.contiguous())
        ret += rew * alive
        if step < 599:
            alive &= ~done.bool()
    assert alive.float().mean().item() > 0.95, alive.float().mean().item()
    assert ret[alive].mean().item() > 350.0
    reasons = env.field_int("DONE_REASON")[:, 0].cpu().numpy()
    assert np.all(reasons[alive.cpu().numpy()] == _abi.DONE_TIME_LIMIT)
    env.close()


def test_masked_reset_and_legacy_protocol():
    import torch
    from openroborl_amd.env import VecQuadrupedEnv, LegacyListEnv
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=6, mode="test", auto_reset=False, seed=2)
    leg = LegacyListEnv(env)
    o = leg.reset()
    assert len(o) == 6 and o[0].shape == (160,) and o[0].dtype == np.float64
    acts = [np.zeros(12, dtype=np.float32) for _ in range(6)]
    o, r, d, info = leg.step(acts)
    assert isinstance(r[0], float) and isinstance(d[0], bool)
    assert info[0]["terminated"] is d          # the reference's info dicts alias the done LIST (wrapper_env.py:76-77)
    np.testing.assert_allclose(acts[0], env.models[0]["init_motor_angles"], atol=1e-6)   # in-place += INIT (minitaur.py:281)
    assert leg.env_step_counter == 1 and leg.num_robot == 6
    # curriculum counter: += num_robot in a step where any robot is done (wrapper_env.py:82-83)
    env.field_int("MAX_EP_STEPS")[:] = 2
    o, r, d, info = leg.step([np.zeros(12, dtype=np.float32) for _ in range(6)])
    assert all(d) and info[0]["terminated"] is d
    torch.cuda.synchronize()
    assert int(env.counters[_abi.CNT_TOTAL_STEP_COUNT].item()) == 6
    o = leg.reset()                                   # the runner resets the whole env (imitation_runners.py:185-205)
    assert leg.env_step_counter == 0
    assert leg.observation_space.shape == (160,) and leg.action_space.shape == (12,)
    # masked reset leaves the other robots untouched
    before = env.state.clone()
    mask = torch.tensor([1, 0, 0, 1, 0, 0], dtype=torch.uint8, device=env.device)
    env.reset(mask)
    after = env.state
    assert torch.equal(before[1], after[1]) and torch.equal(before[4], after[4])
    assert not torch.equal(before[0], after[0])
    env.close()


def test_odd_robot_count_padding_lane_group():
    """Two robots share a wavefront; with an odd N the last wave carries a padding lane group that must not
    write anything (it shadows robot 0)."""
    import torch
    for n in (1, 5):
        env, orc = make_pair("laikago", n=n, seed=4)
        og = env.reset().cpu().numpy()
        oo = orc.reset()
        np.testing.assert_allclose(og, oo, atol=2e-6)
        orc.state[:] = gpu_state64(env)
        a = np.random.RandomState(n).uniform(-0.2, 0.2, (n, 12)).astype(np.float32)
        for _ in range(3):
            og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
            oo, ro, do = orc.step(a.astype(np.float64))
        np.testing.assert_allclose(rg.cpu().numpy(), ro, atol=5e-3)
        np.testing.assert_allclose(og.cpu().numpy()[:, 84:], oo[:, 84:], atol=1e-3)
        torch.cuda.synchronize()
        assert env.counters.cpu().numpy()[_abi.CNT_TOTAL_TIMESTEPS] == 3 * n
        env.close(); orc.close()


def test_rollout_across_cycle_wrap_tracks_oracle():
    """25 env steps (> one pace cycle of 19.2 steps) so every robot passes the cycle-sync re-anchoring of
    ImitationTask._sync_ref_origin (imitation_task.py:751-754,1047-1053) at least once."""
    import torch
    n = 128
    env, orc = make_pair("laikago", n=n, seed=13)
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    W = np.load(os.path.join(ol.GOLDEN, "policy_laikago_pace.npz"))

    def policy(o):
        h = np.maximum(o @ W["model__pi_fc0__w_0"] + W["model__pi_fc0__b_0"], 0)
        h = np.maximum(h @ W["model__pi_fc1__w_0"] + W["model__pi_fc1__b_0"], 0)
        return np.clip(h @ W["model__pi__w_0"] + W["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
    og = env.obs.cpu().numpy().astype(np.float64)
    origin0 = orc.field("ORIGIN_POS").copy()
    o32 = f32_twin(env, orc)
    for k in range(25):
        a = policy(og).astype(np.float32)         # same actions on all sides (driven by the GPU observation)
        o_t, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        o3, r3, d3 = o32.step(a)
        og = o_t.cpu().numpy().astype(np.float64)
    moved = np.abs(orc.field("ORIGIN_POS") - origin0).max(axis=1) > 1e-6
    warm = orc.field("WARMUP")[:, 0] > 0
    assert moved[~warm].all()                      # the wrap happened for every non-warm-up robot
    # after 25 env steps of contacts the bound is the float32 oracle's own drift on the same robots (round 2: hand-set 2e-3 .. 0.1)
    from tests import drift
    compare_to_floor(env, orc, o32, ["ORIGIN_POS", "POS", "REF_POSE", "QUAT", "Q"], "25 steps")
    drift.assert_within_float32_floor(np.abs(rg.cpu().numpy() - ro), np.abs(r3.astype(np.float64) - ro), "25-step reward")
    assert (dg.cpu().numpy().astype(bool) == do).all()
    env.close(); orc.close(); o32.close()


def test_device_policy_rollout_and_gae():
    """SURVEY 8f items 1-2 on the device: batched MLP + rollout buffers + per-robot GAE without leaving the GPU."""
    import torch
    from openroborl_amd import policy as pol, rollout
    from openroborl_amd.env import VecQuadrupedEnv
    env = VecQuadrupedEnv(num_robot=256, seed=3, robot="laikago", motion_file="laikago_pace", mode="test", auto_reset=True)
    p = pol.MLPPolicy.from_file(os.path.join(ol.GOLDEN, "policy_laikago_pace.npz"), env.device)
    buf = rollout.collect_rollout(env, p, horizon=64, deterministic=True)
    assert buf["obs"].shape == (64, 256, 160) and buf["rewards"].is_cuda
    assert torch.isfinite(buf["rewards"]).all() and buf["rewards"].mean().item() > 0.5     # the policy tracks the clip
    assert buf["dones"].float().mean().item() < 0.01
    adv, ret = rollout.gae(buf["rewards"], buf["vpred"], buf["dones"])
    nrm = rollout.normalize_per_robot(adv)
    assert torch.isfinite(nrm).all() and abs(nrm.mean().item()) < 1e-3
    env.close()


def test_legacy_host_buffer_rate_is_reported():
    """Not a parity test: measures the PCIe-inclusive rate of the list-of-numpy protocol for DESIGN.md."""
    import time
    import torch
    from openroborl_amd.env import VecQuadrupedEnv, LegacyListEnv
    n = 1024
    env = VecQuadrupedEnv(num_robot=n, seed=1, robot="laikago", motion_file="laikago_pace", mode="test", auto_reset=False)
    leg = LegacyListEnv(env, mutate_actions=False)
    leg.reset()
    acts = [np.zeros(12, dtype=np.float32) for _ in range(n)]
    for _ in range(3):
        leg.step(acts)
    t0 = time.perf_counter()
    k = 20
    for _ in range(k):
        leg.step(acts)
    dt = time.perf_counter() - t0
    print("LEGACY_HOST_RATE robots=%d steps_per_s=%.0f ms_per_step=%.3f" % (n, n * k / dt, 1e3 * dt / k))
    assert n * k / dt > 1e4
    env.close()


def test_checkpoint_resume_is_bitwise():
    """state_dict / load_state_dict (the env state is a handful of device tensors)."""
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    env = VecQuadrupedEnv(num_robot=64, seed=9, robot="laikago", motion_file="laikago_pace", mode="train", auto_reset=True)
    env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [(torch.randn(64, 12, generator=g) * 0.2).to(env.device) for _ in range(30)]
    for a in acts[:10]:
        env.step(a)
    ck = env.state_dict()
    outs = []
    for a in acts[10:]:
        o, r, d, _ = env.step(a)
        outs.append((o.clone(), r.clone(), d.clone()))
    env.load_state_dict(ck)
    for a, (o0, r0, d0) in zip(acts[10:], outs):
        o, r, d, _ = env.step(a)
        assert torch.equal(o, o0) and torch.equal(r, r0) and torch.equal(d, d0)
    st = env.stats()
    assert st["total_timesteps"] > 0
    assert set(st["last_done_reason"]) == {"contact_fall", "root_pos", "root_rot", "time_limit", "non_finite", "motion_over"}
    env.close()


def test_latency_ring_wraps_with_random_latency():
    """Three env steps (99 pushes into the 44-deep ring: it wraps twice) with the randomiser on, i.e. per-robot latencies
    of 0-40 ms: ring cursor, ring contents and the delayed observations still follow the oracle."""
    import torch
    n = 64
    env, orc = make_pair("laikago", n=n, randomizer=True, mode="train", seed=11)
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    rng = np.random.RandomState(3)
    for k in range(3):
        # both sides start every env step from the device's state (ring included): what is compared is the ring / latency logic
        # over the wrap-arounds, not how far three steps of contact dynamics amplify float32 rounding
        orc.state[:] = gpu_state64(env)
        a = rng.uniform(-0.15, 0.15, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
    lat = env.field("LATENCY")[:, 0].cpu().numpy()
    assert lat.max() > 0.03 and lat.min() < 0.01          # the sample really spans short and long latencies
    compare_fields(env, orc, ["RING_LEN", "RING_HEAD"], atol=0, what="ring cursor")
    g = gpu_state64(env)
    sl = env.layout.sl("RING")
    alive = ~(dg.cpu().numpy().astype(bool) | do)
    assert alive.mean() > 0.9
    rg_, ro_ = g[alive][:, sl].reshape(-1, _abi.RING_DEPTH, _abi.RING_ENTRY), orc.state[alive][:, sl].reshape(-1, _abi.RING_DEPTH, _abi.RING_ENTRY)
    # motor angles + relative quaternion of every ring entry (one env step apart at most): 33 sub-steps of contact dynamics amplify
    # float32 rounding in the odd robot, so 99.5 % of the entries within 5e-3 and none beyond 0.05
    da = np.abs(rg_[:, :, :16] - ro_[:, :, :16])
    assert (da < 5e-3).mean() > 0.995 and da.max() < 0.05, ((da < 5e-3).mean(), da.max())
    dr = np.abs(rg_[:, :, 16:19] - ro_[:, :, 16:19])                                  # base rates: noisier (contacts)
    assert np.median(dr) < 1e-3 and dr.max() < 0.3, (np.median(dr), dr.max())
    do_ = np.abs(og.cpu().numpy()[alive][:, 12:84] - oo[alive][:, 12:84])                           # last actions + delayed motor angles
    assert (do_ < 5e-3).mean() > 0.995 and do_.max() < 0.05, ((do_ < 5e-3).mean(), do_.max())


def test_auto_reset_inside_step_matches_oracle():
    """Train mode: an episode that survives hits the 20-step time limit, and the auto-reset inside the step launch then
    happens on both sides at the same env step.  What a reset produces depends only on the RNG stream and the clip (not on
    the chaotic physics before it), so the post-reset state of those robots must match tightly."""
    import torch
    n = 64
    env, orc = make_pair("laikago", n=n, randomizer=True, auto_reset=True, mode="train", seed=17)
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    limit = int(env.field_int("MAX_EP_STEPS").max())
    assert limit == 20
    rng = np.random.RandomState(1)
    clean = np.ones(n, dtype=bool)                   # no reset so far on either side
    checked = 0
    for k in range(limit + 1):
        a = rng.uniform(-0.05, 0.05, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        dgn = dg.cpu().numpy().astype(bool)
        if k == limit - 1:
            sel = clean & dgn & do
            assert sel.sum() > n // 2 and (dgn[clean] == do[clean]).all()
            g = gpu_state64(env)
            for name, tol in [(f, 0) for f in ("EPISODE_IDX", "EP_STEP", "RING_LEN", "RING_HEAD", "STEP_COUNTER", "STATE_ACTION_COUNTER",
                                                "FILTER_VALID", "WARMUP", "MAX_EP_STEPS")] + \
                             [(f, 2e-5) for f in ("TIME_OFFSET", "LATENCY", "FOOT_MU", "KNEE_FRICTION", "MASS_RATIO", "INERTIA_RATIO",
                                                   "STRENGTH", "ORIGIN_POS", "ORIGIN_ROT", "REF_POSE", "REF_VEL", "POS", "QUAT", "Q", "QD",
                                                   "LINVEL", "ANGVEL")]:
                sl = env.layout.sl(name)
                np.testing.assert_allclose(g[sel][:, sl], orc.state[sel][:, sl], atol=tol, rtol=0, err_msg=name)
            np.testing.assert_allclose(og.cpu().numpy()[sel], oo[sel], atol=2e-4)   # the observation returned is the reset observation
            checked = int(sel.sum())
            kept = sel
        clean &= ~(dgn | do)
    assert checked > 0
    # one step into the next episode the robots that reset together still track each other
    g = gpu_state64(env)
    for name in ("POS", "QUAT", "Q"):
        sl = env.layout.sl(name)
        assert np.median(np.abs(g[kept][:, sl] - orc.state[kept][:, sl])) < 1e-4, name
    env.close(); orc.close()


def test_motion_over_on_a_non_looping_clip(tmp_path):
    """ORR_DONE_MOTION_OVER (imitation_task.py:224-233,532,567) on the device, against the oracle, for a Clamp clip."""
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    from tests.test_oracle_env import clamp_clip
    clip = clamp_clip(tmp_path)
    n = 32
    env = VecQuadrupedEnv(num_robot=n, seed=3, robot="laikago", motion_file=clip.path, mode="test", enable_randomizer=False, auto_reset=False)
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=4)
    env.reset(); orc.reset()
    orc.state[:] = gpu_state64(env)
    seen = np.zeros(n, dtype=bool)
    for k in range(30):          # duration 0.63 s = 19.2 steps, + 0.25 s = 7.6 steps for a warm-up episode
        a = np.zeros((n, 12), dtype=np.float32)
        env.step(torch.from_numpy(a).to(env.device)); orc.step(a.astype(np.float64))
        rg = env.field_int("DONE_REASON")[:, 0].cpu().numpy() & _abi.DONE_MOTION_OVER
        ro = orc.field("DONE_REASON")[:, 0].astype(int) & _abi.DONE_MOTION_OVER
        np.testing.assert_array_equal(rg, ro)
        seen |= rg != 0
    assert seen.all()
    env.close(); orc.close()


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_shank_contact_parity(robot):
    """Row C with robots laid on their shanks (lower legs are feet, minitaur.py:842-844): the per-leg contact is the shank
    sphere, not the toe; HIP vs oracle like test_physics_substep_parity."""
    import torch
    from tests.parity_inputs import shank_contact_inputs
    n = 32
    env, orc = make_pair(robot, n=n)
    env.reset(); orc.reset()
    _, _, _, st, tau = shank_contact_inputs(robot, n)
    push_state(env, st); orc.state[:] = st
    tg = torch.tensor(tau, dtype=torch.float32, device=env.device)
    o32 = f32_twin(env, orc)
    tau32 = tau.astype(np.float32)
    for nsub, ptol, vtol in ((1, 2e-6, 2e-4), (8, 1e-4, 2e-3)):
        env.debug_physics(tg, nsub)
        for i in range(n):
            for _ in range(nsub):
                orc.L.orc_physics_substep(orc.h, ol.P(orc.state[i]), ol.P(np.ascontiguousarray(tau[i])))
                o32.L.orc_physics_substep(o32.h, o32.P(o32.state[i]), o32.P(np.ascontiguousarray(tau32[i])))
        out = 0 if nsub == 1 else 1     # 8 sub-steps: one robot of 32 may see a contact row switch a sub-step apart (round 6: 4 mm contact margin)
        compare_fields(env, orc, ["POS", "QUAT", "Q"], atol=ptol, rtol=ptol, what="shank nsub=%d" % nsub, outlier_robots=out, floor=o32)
        compare_fields(env, orc, ["LINVEL", "ANGVEL", "QD"], atol=vtol, rtol=vtol, what="shank nsub=%d" % nsub, outlier_robots=out, floor=o32)
        compare_fields(env, orc, ["LAMBDA"], atol=5e-4 if nsub == 1 else 5e-3, what="shank nsub=%d" % nsub, outlier_robots=out, floor=o32)
        if nsub == 1:   # the robots start with a penetrating shank sphere: its contact is live in (nearly) every robot
            lam = orc.field("LAMBDA").reshape(n, 4, 3)
            assert (lam[:, :, 0].sum(axis=1) > 0).mean() > 0.9
    env.close(); orc.close(); o32.close()


def test_shipped_minicheetah_policy_probe():
    """IN-SAMPLE anchor (the table was identified against this very policy; there is no second mini-cheetah policy to hold out).
    Behavioural anchor for config 3's robot: the reference's minicheetah_trot policy (trained in PyBullet on the real URDF) walks the
    600-step episode on the mini-cheetah table of robots.py.  Round 2: 0 % of the robots finished (mean survival 158 steps) on the
    hand-authored table; round 3 identified the distal masses / COMs / hip height against this very policy by survival (0.90 finish);
    round 6 re-identified it on the policy's own reward under the adopted solver constants (tools/identify_r6.py P8 + P7; DESIGN.md
    section 7.3): 0.98 finish at 0.70 reward per step.  A failure is counted only for a termination other than the time limit."""
    import torch
    W = np.load(os.path.join(ol.GOLDEN, "policy_minicheetah_trot.npz"))
    n = 1024
    env, orc = make_pair("mini_cheetah", n=n, seed=1)
    orc.close()
    dev = env.device
    w = {k: torch.tensor(W[k], device=dev) for k in W.files}
    obs = env.reset()
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    length = torch.zeros(n, device=dev)
    ret = torch.zeros(n, device=dev)
    reason = env.field_int("DONE_REASON")[:, 0]
    for step in range(600):
        h = torch.relu(obs @ w["model__pi_fc0__w_0"] + w["model__pi_fc0__b_0"])
        h = torch.relu(h @ w["model__pi_fc1__w_0"] + w["model__pi_fc1__b_0"])
        a = torch.clamp(h @ w["model__pi__w_0"] + w["model__pi__b_0"], -2 * np.pi, 2 * np.pi)
        obs, rew, done, _ = env.step(a.contiguous())
        ret += rew * alive
        length += alive.float()
        alive &= ~(done.bool() & ((reason & ~_abi.DONE_TIME_LIMIT) != 0))
    finished = alive.float().mean().item()
    mean_len = length.mean().item()
    rps = (ret / length).mean().item()
    print("MINICHEETAH_PROBE finished=%.3f mean_survival_steps=%.1f reward_per_step=%.3f" % (finished, mean_len, rps))
    assert finished >= 0.95            # measured 0.983 / 0.982 (seeds 1 / 2; rounds 3-5: 0.90)
    assert mean_len > 570.0 and rps > 0.66                 # 590, 0.697
    env.close()


@pytest.mark.parametrize("n", [1, 5, 7])
def test_robot_counts_that_do_not_fill_a_wavefront(n):
    """Four robots share a wavefront; when N is not a multiple of four the spare lane groups shadow robot 0 and must neither store
    nor disturb their neighbours (every lane stores into LDS unconditionally since round 2): reset + two env steps with auto-reset
    against the oracle, and the buffers behind the last robot stay untouched."""
    import torch
    env, orc = make_pair("laikago", n=n, randomizer=True, auto_reset=True, mode="train", seed=13)
    og = env.reset().cpu().numpy(); oo = orc.reset()
    np.testing.assert_allclose(og, oo, atol=5e-4)
    orc.state[:] = gpu_state64(env)
    rng = np.random.RandomState(n)
    for _ in range(2):
        a = rng.uniform(-0.2, 0.2, (n, 12)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).to(env.device))
        oo, ro, do = orc.step(a.astype(np.float64))
        assert og.shape == (n, 160) and rg.shape == (n,) and np.isfinite(og.cpu().numpy()).all()
        np.testing.assert_allclose(rg.cpu().numpy(), ro, atol=3e-3)
        np.testing.assert_allclose(og.cpu().numpy()[:, 12:84], oo[:, 12:84], atol=5e-4)
        orc.state[:] = gpu_state64(env)
    assert int(env.counters[_abi.CNT_TOTAL_TIMESTEPS].item()) == 2 * n      # the padding groups are not counted
    env.close(); orc.close()


@pytest.mark.parametrize("iters", [1, 2, 4, 7])
def test_physics_substep_parity_other_solver_iteration_counts(iters):
    """The Gauss-Seidel loop of the kernel runs three sweeps per iteration and the remaining ones singly: iteration counts that are not
    the reference's 9 (numSolverIterations, quadruped_gym_env.py:165) take the other entry and exit paths of that loop."""
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    from tests.parity_inputs import substep_parity_inputs
    n = 64
    env = VecQuadrupedEnv(num_robot=n, seed=3, robot="laikago", motion_file=CLIP["laikago"], mode="test", enable_randomizer=False,
                          auto_reset=False, config_overrides=dict(solver_iters=iters))
    assert env.cfg.solver_iters == iters
    orc = ol.OracleEnv(env.cfg, env.models, env.clips, n, robot_type=env.robot_type, clip_id=env.clip_id, threads=8)
    env.reset(); orc.reset()
    _, _, _, st, tau = substep_parity_inputs("laikago", n)
    push_state(env, st); orc.state[:] = st
    tg = torch.tensor(tau, dtype=torch.float32, device=env.device)
    env.debug_physics(tg, 4)
    for i in range(n):
        for _ in range(4):
            orc.L.orc_physics_substep(orc.h, ol.P(orc.state[i]), ol.P(np.ascontiguousarray(tau[i])))
    compare_fields(env, orc, ["POS", "QUAT", "Q"], atol=5e-5, rtol=5e-5, what="iters=%d" % iters)
    compare_fields(env, orc, ["LINVEL", "ANGVEL", "QD"], atol=1e-3, rtol=1e-3, what="iters=%d" % iters)
    compare_fields(env, orc, ["LAMBDA"], atol=5e-3, what="iters=%d" % iters)
    env.close(); orc.close()


@pytest.mark.parametrize("auto_reset", [True, False])
def test_non_finite_state_is_caught_by_the_state_guard(auto_reset):
    """ORR_DONE_NAN: the ONE detection point for non-finite numbers is the |x| < 1e30 sweep over POS..QD (+ the reward check) at the end of
    the step.  The branch-free atan2 / asin / acos (csrc/orr_device.h) drop NaNs through fmax / fmin / selects, so a NaN orientation does
    NOT propagate through the IMU / heading paths - it cannot be detected downstream, only in the state itself.  NaN / inf injected into
    QUAT, ANGVEL, Q: those robots finish with DONE_NAN and a finite reward; with auto-reset the returned observation is the (finite) reset
    observation and the next step is clean; without it a masked reset recovers them.  Neighbours in the same wavefront are untouched."""
    import torch
    n = 8
    env, orc = make_pair("laikago", n=n, auto_reset=auto_reset, seed=6)
    orc.close()
    env.reset()
    ref = env.state.clone()
    lay = env.layout
    env.state[0, lay.sl("QUAT").start + 1] = float("nan")
    env.state[1, lay.sl("ANGVEL").start] = float("nan")
    env.state[2, lay.sl("Q").start + 5] = float("inf")
    a = torch.zeros(n, 12, device=env.device)
    obs, rew, done, _ = env.step(a)
    torch.cuda.synchronize()
    bad = torch.tensor([1, 1, 1, 0, 0, 0, 0, 0], dtype=torch.bool, device=env.device)
    assert done.bool()[bad].all() and not done.bool()[~bad].any()
    reason = env.field_int("DONE_REASON")[:, 0]
    print("NAN_GUARD auto_reset=%s reasons %s" % (auto_reset, reason.cpu().numpy().tolist()))
    assert ((reason[bad] & _abi.DONE_NAN) != 0).all() and (reason[~bad] == 0).all()
    assert torch.isfinite(rew).all() and (rew[bad] == 0).all() and (rew[~bad] > 0).all()
    assert torch.isfinite(obs[~bad]).all()
    if auto_reset:
        assert torch.isfinite(obs).all()                       # the reset observation
        assert torch.isfinite(env.state[:, :lay.sl("RING").start]).all()
    else:
        env.reset(bad.to(torch.uint8))
    obs, rew, done, _ = env.step(a)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and not done.any()
    env.close()


@pytest.mark.parametrize("auto_reset", [True, False])
def test_non_finite_action_is_caught_by_the_action_guard(auto_reset):
    """The third entry of the non-finite guard: the ACTION.  The +-0.2 rad clip of the motor command is fmin / fmax, which silently drops
    a NaN, while the last-action sensor and the Butterworth history would keep it for the rest of the episode: orr_step records a NaN /
    inf action like a non-finite incoming state (ORR_DONE_NAN, reward 0) and computes with 0 in its place, so nothing non-finite
    enters the state record or the observation.  Neighbours in the same wavefront (robots 3..7 share waves with 0..2) are untouched:
    their step equals the step of the same batch with clean actions bit for bit."""
    import torch
    n = 8
    env, orc = make_pair("laikago", n=n, auto_reset=auto_reset, seed=6)
    orc.close()
    env.reset()
    start = env.state_dict()
    a = torch.zeros(n, 12, device=env.device)
    obs_c, rew_c, done_c = (x.clone() for x in env.step(a)[:3])
    clean = env.state.clone()
    env.load_state_dict(start)
    a_bad = a.clone()
    a_bad[0, 3] = float("nan"); a_bad[1, 0] = float("inf"); a_bad[2, 11] = float("-inf")
    obs, rew, done, _ = env.step(a_bad)
    torch.cuda.synchronize()
    bad = torch.tensor([1, 1, 1, 0, 0, 0, 0, 0], dtype=torch.bool, device=env.device)
    reason = env.field_int("DONE_REASON")[:, 0]
    print("NAN_ACTION_GUARD auto_reset=%s reasons %s" % (auto_reset, reason.cpu().numpy().tolist()))
    assert done.bool()[bad].all() and ((reason[bad] & _abi.DONE_NAN) != 0).all() and (rew[bad] == 0).all()
    assert torch.equal(obs[~bad], obs_c[~bad]) and torch.equal(rew[~bad], rew_c[~bad]) and torch.equal(done[~bad], done_c[~bad])
    assert torch.equal(env.state[~bad], clean[~bad])
    lay = env.layout
    assert torch.isfinite(obs).all() and torch.isfinite(env.state[:, :lay.sl("RING").start]).all()    # nothing non-finite was stored
    if not auto_reset:
        env.reset(bad.to(torch.uint8))
    obs, rew, done, _ = env.step(a)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and not done.any()
    env.close()


@pytest.mark.parametrize("robot,randomizer", [("laikago", True), ("mini_cheetah", False)])
def test_two_waves_per_simd_variant_parity(robot, randomizer, monkeypatch):
    """orr_step runs a second build of the step kernel (two waves per SIMD: <= 256 registers, its own translation unit and compiler
    flags, csrc/orr_kernels_w2.hip) for batches of more than 4 x #SIMDs robots.  ORR_STEP_WAVES_PER_EU=2 forces it onto small batches:
    the one-step parity test, the auto-reset test and a reset + 3-step run with a padding lane group, against the oracle.  (The whole
    -m gpu suite also passes with the variable set; tests/test_gpu_scale.py runs the 8192-robot config through it by default.)"""
    monkeypatch.setenv("ORR_STEP_WAVES_PER_EU", "2")
    test_step_parity(robot, randomizer)
    if robot == "laikago":
        test_auto_reset_inside_step_matches_oracle()
        test_robot_counts_that_do_not_fill_a_wavefront(5)
        test_non_finite_state_is_caught_by_the_state_guard(True)


def test_variant_selection_by_batch_size():
    """4096 robots = 1024 waves = one per SIMD of an MI355X: the one-wave kernel; one robot more: the two-wave kernel.  Both give the same
    physics: the first 4096 robots of a 4100-robot env (two-wave kernel) track a 4096-robot env (one-wave kernel) from the same seed to
    float32 rounding over three steps (compiler flags differ between the two builds, so not bitwise)."""
    import torch
    from openroborl_amd.env import VecQuadrupedEnv
    kw = dict(seed=4, robot="laikago", motion_file=CLIP["laikago"], mode="test", enable_randomizer=False, auto_reset=False)
    a = VecQuadrupedEnv(num_robot=4096, **kw)
    b = VecQuadrupedEnv(num_robot=4100, **kw)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob[:4096])                       # the reset kernel is the same build for both
    act = (torch.randn(4100, 12, generator=torch.Generator().manual_seed(0)) * 0.1).to(a.device)
    for _ in range(3):
        oa, ra, da, _ = a.step(act[:4096].contiguous())
        ob, rb, db, _ = b.step(act)
    # two different builds of the same source ran (different scheduler, -O2 / -O3): with this compiler they come out BITWISE equal
    # (floating-point contraction is decided before scheduling); the contract is only float32 closeness
    print("VARIANTS bitwise_equal=%s" % bool(torch.equal(oa, ob[:4096]) and torch.equal(ra, rb[:4096])))
    assert (ra - rb[:4096]).abs().median().item() < 1e-5
    assert (oa[:, 84:] - ob[:4096, 84:]).abs().median().item() < 1e-5
    assert (da == db[:4096]).float().mean().item() > 0.999
    a.close(); b.close()


def test_joint_limit_rows_appear_on_time_when_the_setup_is_skipped():
    """The joint-limit bank's row setup is skipped for as many sub-steps as no joint of the wave can reach the activation distance of a bound
    (a coordinate moves at most max_coord_velocity * dt per sub-step; csrc/orr_physics.h: limit_idle).  Airborne robots whose knees /
    thighs / hips run towards a limit at up to 95 rad/s from 0.12 .. 1.5 rad away, 14 sub-steps in ONE launch (the skip counter lives
    across the sub-steps of a launch) against the oracle: a row that came one sub-step late would let the joint run 0.1 rad through it."""
    import torch
    from tests.parity_inputs import substep_parity_inputs
    n = 48
    env, orc = make_pair("laikago", n=n)
    env.reset(); orc.reset()
    _, _, _, st, tau = substep_parity_inputs("laikago", n)
    lay = env.layout
    m = env.models[0]
    rng = np.random.RandomState(12)
    st[:, lay.sl("POS").start + 2] = 2.0                          # no ground contact
    st[:, lay.sl("LINVEL")] = 0.0
    st[:, lay.sl("ANGVEL")] = 0.0
    st[:, lay.sl("KNEE_FRICTION")] = 0.0
    q = np.tile((m["init_motor_angles"] + m["motor_offset"]) * m["motor_dir"], (n, 1))
    qd = np.zeros((n, 12))
    lo, hi = m["joint_lo"], m["joint_hi"]                        # limits of the kinematic angle = motor angle (joint axes +x / +y here)
    for i in range(n):
        j = rng.randint(12)
        mot = list(m["joint_of_motor"]).index(j)
        away = rng.choice([0.12, 0.2, 0.35, 0.5, 0.8, 1.2, 1.5])
        side = rng.rand() < 0.5
        room = hi[j] - lo[j]
        away = min(away, 0.45 * room)
        ang = (hi[j] - away) if side else (lo[j] + away)          # motor angle
        speed = rng.uniform(40.0, 95.0) * (1.0 if side else -1.0)
        q[i, j] = ang * m["motor_dir"][mot] + m["motor_offset"][mot]
        qd[i, j] = speed * m["motor_dir"][mot]
    st[:, lay.sl("Q")] = q
    st[:, lay.sl("QD")] = qd
    st = statemod.to_float64(lay, statemod.from_float64(lay, st))
    push_state(env, st); orc.state[:] = st
    tau0 = np.zeros((n, 12))
    nsub = 14
    env.debug_physics(torch.zeros(n, 12, device=env.device), nsub)
    hit = np.zeros(n, dtype=bool)
    for i in range(n):
        for _ in range(nsub):
            orc.L.orc_physics_substep(orc.h, ol.P(orc.state[i]), ol.P(np.ascontiguousarray(tau0[i])))
    # the oracle's joints were stopped by their limits (they would have travelled 0.56 .. 1.3 rad otherwise)
    moved = np.abs(orc.state[:, lay.sl("Q")] - q).max(axis=1)
    assert (moved < 1.45).all() and (np.abs(orc.state[:, lay.sl("QD")]).max(axis=1) < 96.0).all()
    g = gpu_state64(env)
    dq = np.abs(g[:, lay.sl("Q")] - orc.state[:, lay.sl("Q")]).max(axis=1)
    dv = np.abs(g[:, lay.sl("QD")] - orc.state[:, lay.sl("QD")]).max(axis=1)
    assert dq.max() < 2e-3 and dv.max() < 0.2, (dq.max(), dv.max(), np.argmax(dq))      # a late row: 0.04 .. 0.1 rad
    env.close(); orc.close()


def test_misaligned_buffers_are_refused():
    """Records and observations move in 16-byte pieces on the device: a caller-owned state / observation buffer that is not 16-byte
    aligned is refused by the C-ABI before anything is launched."""
    import ctypes as C
    import torch
    env, orc = make_pair("laikago", n=8)
    orc.close()
    L = env.L
    stream = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
    act = torch.zeros(8, 12, device=env.device)
    big = torch.zeros(8 * 160 + 4, dtype=torch.float32, device=env.device)
    rc = L.orr_step(env.h, act.data_ptr(), big.data_ptr() + 4, env.reward.data_ptr(), env.done.data_ptr(), stream)
    assert rc < 0 and b"16-byte aligned" in L.orr_last_error()
    assert L.orr_step(env.h, act.data_ptr(), big.data_ptr(), env.reward.data_ptr(), env.done.data_ptr(), stream) == 0
    st2 = torch.zeros(env.state.numel() + 4, dtype=torch.float32, device=env.device)
    rc = L.orr_bind(env.h, st2.data_ptr() + 4, env.counters.data_ptr(), None, 0)
    assert rc < 0 and b"16-byte aligned" in L.orr_last_error()
    torch.cuda.synchronize()
    env.close()
