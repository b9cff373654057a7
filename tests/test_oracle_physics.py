"""Physics invariants of the oracle's dynamics (the part whose parity vs PyBullet is UNPINNED):
checked against an independent numpy Lagrangian reference (tests/phys_ref.py).  CPU only."""
import ctypes as C

import numpy as np
import pytest

from openroborl_amd import _abi, config, motion, robots
from tests import oracle_lib as ol
from tests import phys_ref as pr
from tests.oracle_lib import P


def make_env(robot="laikago", n=1, **kw):
    cfg = config.make_config(n, mode="test", enable_randomizer=False, auto_reset=False, **kw)
    model = robots.ROBOTS[robot]()
    # the C-ABI model table is float32 and the oracle recovers decimal constants from it: same for the independent reference's copy
    for k, v in list(model.items()):
        if isinstance(v, np.ndarray) and v.dtype == np.float64:
            model[k] = ol.dec32(v)
        elif isinstance(v, float):
            model[k] = ol.dec32(v)
    clip = motion.MotionClip("laikago_pace" if robot == "laikago" else "minicheetah_trot")
    models = [None, None]
    t = robots.ROBOT_TYPE_ID[robot]
    models[t] = model
    env = ol.OracleEnv(cfg, models, [clip], n, robot_type=t)
    return env, model


def random_state(env, model, rng, vel=True, height=1.0):
    s = env.state[0]
    lay = env.lay
    q = rng.randn(4); q /= np.linalg.norm(q)
    s[lay.sl("POS")] = [rng.randn() * 0.1, rng.randn() * 0.1, height]
    s[lay.sl("QUAT")] = q
    s[lay.sl("Q")] = rng.uniform(-0.8, 0.8, 12) + np.tile([0, 0, -0.5], 4)
    if vel:
        s[lay.sl("LINVEL")] = rng.randn(3)
        s[lay.sl("ANGVEL")] = rng.randn(3) * 2
        s[lay.sl("QD")] = rng.randn(12) * 3
    else:
        s[lay.sl("LINVEL")] = 0; s[lay.sl("ANGVEL")] = 0; s[lay.sl("QD")] = 0


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_inverse_mass_matrix_matches_lagrangian(robot):
    env, model = make_env(robot)
    rng = np.random.RandomState(0)
    lay = env.lay
    for trial in range(5):
        random_state(env, model, rng)
        s = env.state[0]
        acc = np.zeros(18); Minv = np.zeros((18, 18)); tau = np.zeros(12)
        env.L.orc_dynamics_probe(env.h, P(s), P(tau), P(acc), P(Minv))
        bodies, axes = pr.kinematics(model, s[lay.sl("POS")], s[lay.sl("QUAT")], s[lay.sl("Q")])
        Js = pr.body_jacobians(bodies, axes)
        M = pr.mass_matrix(bodies, Js)
        np.testing.assert_allclose(Minv, Minv.T, atol=1e-9)
        np.testing.assert_allclose(Minv @ M, np.eye(18), atol=1e-8)
    env.close()


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_forward_dynamics_at_rest(robot):
    """zero velocity: acc = M^-1 (tau + Q_gravity)."""
    env, model = make_env(robot)
    rng = np.random.RandomState(1)
    lay = env.lay
    dirj, offj, motor_of_joint = pr.joint_maps(model)
    for trial in range(5):
        random_state(env, model, rng, vel=False)
        s = env.state[0]
        tau_m = rng.randn(12) * 5
        acc = np.zeros(18); Minv = np.zeros((18, 18))
        env.L.orc_dynamics_probe(env.h, P(s), P(tau_m), P(acc), P(Minv))
        bodies, axes = pr.kinematics(model, s[lay.sl("POS")], s[lay.sl("QUAT")], s[lay.sl("Q")])
        Js = pr.body_jacobians(bodies, axes)
        M = pr.mass_matrix(bodies, Js)
        Q = pr.gravity_force(bodies, Js, -10.0)
        Q[6:] += tau_m[motor_of_joint]
        np.testing.assert_allclose(acc, np.linalg.solve(M, Q), atol=1e-7)
    env.close()


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_free_flight_conserves_energy_and_momentum(robot):
    """No contacts, zero torque: energy drift small (semi-implicit Euler, dt = 1 ms), linear momentum
    changes by m g t, angular momentum about the COM conserved -> checks the velocity-product terms."""
    env, model = make_env(robot)
    rng = np.random.RandomState(2)
    lay = env.lay
    random_state(env, model, rng, height=50.0)
    s = env.state[0]
    # keep joints away from the (Laikago) limits for the duration
    dirj, offj, moj = pr.joint_maps(model)
    akin = np.tile([0.1, 0.5, -1.3], 4) if robot == "laikago" else np.tile([0.1, -0.6, 1.3], 4)
    s[lay.sl("Q")] = akin * dirj + offj
    s[lay.sl("QD")] = rng.randn(12) * 1.0

    def em():
        return pr.energy_momentum(model, s[lay.sl("POS")], s[lay.sl("QUAT")], s[lay.sl("Q")], s[lay.sl("ANGVEL")],
                                  s[lay.sl("LINVEL")], s[lay.sl("QD")], -10.0)
    ke0, pe0, P0, L0, mtot = em()
    tau = np.zeros(12)
    nsteps = 300
    for _ in range(nsteps):
        env.L.orc_physics_substep(env.h, P(s), P(tau))
    ke1, pe1, P1, L1, _ = em()
    e0, e1 = ke0 + pe0, ke1 + pe1
    # semi-implicit Euler in a uniform field loses exactly m g^2 dt^2 / 2 per step; the rest is rotational drift
    drift = (e1 - e0) + 0.5 * mtot * 100.0 * 1e-6 * nsteps
    assert abs(drift) < 1e-2 * max(ke0, 1.0), (e0, e1, ke0, drift)
    np.testing.assert_allclose(P1 - P0, [0, 0, -10.0 * mtot * nsteps * 1e-3], atol=5e-3)  # O(dt^2) per-step integrator error in generalised coordinates
    np.testing.assert_allclose(L1, L0, atol=2e-3 * max(1.0, np.linalg.norm(L0)))
    env.close()


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_static_stance_holds(robot):
    """PD at the default pose on the ground: settles, stays up, contact impulses carry the weight."""
    env, model = make_env(robot)
    lay = env.lay
    s = env.state[0]
    dirj, offj, moj = pr.joint_maps(model)
    akin = np.array(model["init_motor_angles"])[moj]  # kinematic angle per joint = its motor's angle
    s[lay.sl("Q")] = akin * dirj + offj
    s[lay.sl("QUAT")] = model["init_quat"]
    s[lay.sl("POS")] = model["init_pos"]
    s[lay.sl("QD")] = 0; s[lay.sl("LINVEL")] = 0; s[lay.sl("ANGVEL")] = 0
    kp, kd = np.array(model["kp"]), np.array(model["kd"])
    for step in range(1500):
        qm = (s[lay.sl("Q")][model["joint_of_motor"]] - model["motor_offset"]) * model["motor_dir"]
        qdm = s[lay.sl("QD")][model["joint_of_motor"]] * model["motor_dir"]
        tau = -kp * (qm - model["init_motor_angles"]) - kd * qdm
        fall = env.L.orc_physics_substep(env.h, P(s), P(np.ascontiguousarray(tau)))
        assert fall == 0
    bodies, _ = pr.kinematics(model, s[lay.sl("POS")], s[lay.sl("QUAT")], s[lay.sl("Q")])
    mtot = sum(b["m"] for b in bodies)
    lam = s[lay.sl("LAMBDA")].reshape(4, 3)
    assert np.all(np.isfinite(s[:lay.fields["RING"][0]]))
    assert abs(np.linalg.norm(s[lay.sl("LINVEL")])) < 0.1  # lightly damped fore-aft rocking on the PD legs
    assert abs(np.linalg.norm(s[lay.sl("ANGVEL")])) < 0.5
    # sum of normal impulses / dt = weight
    np.testing.assert_allclose(lam[:, 0].sum() / 1e-3, mtot * 10.0, rtol=2e-2)
    assert s[lay.sl("POS")][2] > 0.6 * model["init_pos"][2]
    env.close()


def _make_env_model(robot, **over):
    cfg = config.make_config(1, mode="test", enable_randomizer=False, auto_reset=False)
    model = robots.ROBOTS[robot]()
    model.update(over)
    for k, v in list(model.items()):
        if isinstance(v, np.ndarray) and v.dtype == np.float64:
            model[k] = ol.dec32(v)
        elif isinstance(v, float):
            model[k] = ol.dec32(v)
    clip = motion.MotionClip("laikago_pace" if robot == "laikago" else "minicheetah_trot")
    models = [None, None]
    models[robots.ROBOT_TYPE_ID[robot]] = model
    return ol.OracleEnv(cfg, models, [clip], 1, robot_type=robots.ROBOT_TYPE_ID[robot]), model


@pytest.mark.parametrize("stiffness,damping", [(30000.0, 1000.0), (10000.0, 300.0)])
def test_soft_toe_contact_settles_at_the_spring_law(stiffness, damping):
    """Bullet's contact stiffness / damping on the toe links (orr_model::contact_stiffness / contact_damping): a
    normal row with cfm = 1 / (dt k + d) and erp = dt k / (dt k + d) is an implicit spring-damper, so a robot standing still must sink
    into the ground until every toe carries F = k x depth; with rigid toes the same stance rests at (almost) zero depth.  The toe
    depth is read off the forward kinematics; the normal impulses still add up to the weight."""
    depth = {}
    for name, over in (("rigid", {"contact_stiffness": 0.0, "contact_damping": 0.0}), ("soft", {"contact_stiffness": stiffness, "contact_damping": damping})):
        env, model = _make_env_model("laikago", **over)
        lay = env.lay
        s = env.state[0]
        dirj, offj, moj = pr.joint_maps(model)
        s[lay.sl("Q")] = np.array(model["init_motor_angles"])[moj] * dirj + offj
        s[lay.sl("QUAT")] = model["init_quat"]
        s[lay.sl("POS")] = model["init_pos"]
        s[lay.sl("QD")] = 0; s[lay.sl("LINVEL")] = 0; s[lay.sl("ANGVEL")] = 0
        kp, kd = np.array(model["kp"]), np.array(model["kd"])
        for _ in range(3000):
            qm = (s[lay.sl("Q")][model["joint_of_motor"]] - model["motor_offset"]) * model["motor_dir"]
            qdm = s[lay.sl("QD")][model["joint_of_motor"]] * model["motor_dir"]
            tau = -kp * (qm - model["init_motor_angles"]) - kd * qdm
            assert env.L.orc_physics_substep(env.h, P(s), P(np.ascontiguousarray(tau))) == 0
        out, masses = np.zeros(34 * 3), np.zeros(13)
        env.L.orc_fk_probe(env.h, P(s), P(out), P(masses))
        toe_z = out[26 * 3:].reshape(8, 3)[1::2, 2] - model["toe_radius"]          # clearance of the four toe spheres (negative = depth)
        lam_n = s[lay.sl("LAMBDA")].reshape(4, 3)[:, 0]
        np.testing.assert_allclose(lam_n.sum() / 1e-3, masses.sum() * 10.0, rtol=2e-2)     # the weight is carried either way
        depth[name] = (-toe_z, lam_n / 1e-3)
        env.close()
    d_rigid, _ = depth["rigid"]
    d_soft, force = depth["soft"]
    assert np.all(np.abs(d_rigid) < 3e-4)                                          # rigid: rests on the surface (erp pushes depth out)
    np.testing.assert_allclose(d_soft, force / stiffness, rtol=0.15, atol=1e-4)    # soft: Hooke's law per toe
    assert np.all(d_soft > 3.0 * np.abs(d_rigid).max())


def test_friction_pyramid_bounds_tangential_impulse():
    """Robot sliding sideways on its feet: |lambda_t| <= mu * lambda_n per axis, and it decelerates."""
    env, model = make_env("laikago")
    lay = env.lay
    s = env.state[0]
    dirj, offj, moj = pr.joint_maps(model)
    s[lay.sl("Q")] = np.array(model["init_motor_angles"])[moj] * dirj + offj
    s[lay.sl("QUAT")] = model["init_quat"]
    s[lay.sl("POS")] = [0, 0, 0.434]
    s[lay.sl("FOOT_MU")] = 0.5
    kp, kd = np.array(model["kp"]), np.array(model["kd"])
    def pd():
        qm = (s[lay.sl("Q")][model["joint_of_motor"]] - model["motor_offset"]) * model["motor_dir"]
        qdm = s[lay.sl("QD")][model["joint_of_motor"]] * model["motor_dir"]
        return np.ascontiguousarray(-kp * (qm - model["init_motor_angles"]) - kd * qdm)
    for _ in range(300):
        env.L.orc_physics_substep(env.h, P(s), P(pd()))
    s[lay.sl("LINVEL")] = [0.0, 1.5, 0.0]
    v0 = 1.5
    for _ in range(50):
        env.L.orc_physics_substep(env.h, P(s), P(pd()))
        lam = s[lay.sl("LAMBDA")].reshape(4, 3)
        assert np.all(np.abs(lam[:, 1]) <= 0.5 * lam[:, 0] + 1e-9)
        assert np.all(np.abs(lam[:, 2]) <= 0.5 * lam[:, 0] + 1e-9)
    assert s[lay.sl("LINVEL")][1] < v0 - 0.02        # (the abduction joints give first: most of the 50 ms the toes stick and the legs swing)
    env.close()


@pytest.mark.parametrize("robot", ["laikago", "mini_cheetah"])
def test_shank_contact_carries_the_robot(robot):
    """Lower legs are feet (minitaur.py:842-844): a robot resting on the knee ends of its shanks is held up by contact forces
    there (normal impulses on the legs, no free fall through the plane) while its toes are in the air."""
    from tests.parity_inputs import shank_contact_inputs
    cfg, models, clips, st, tau = shank_contact_inputs(robot, n=8)
    t = robots.ROBOT_TYPE_ID[robot]
    env = ol.OracleEnv(cfg, models, clips, 8, robot_type=t)
    env.state[:] = st
    lay = env.lay
    z0 = st[:, lay.sl("POS")][:, 2].copy()
    zero = np.zeros(12)
    imp = np.zeros(8)
    for _ in range(60):
        for i in range(8):
            env.L.orc_physics_substep(env.h, P(env.state[i]), P(zero))
        imp += env.state[:, lay.sl("LAMBDA")].reshape(8, 4, 3)[:, :, 0].sum(axis=1)
    s = env.state
    bodies, _ = pr.kinematics(models[t], s[0, lay.sl("POS")], s[0, lay.sl("QUAT")], s[0, lay.sl("Q")])
    weight_impulse = sum(b["m"] for b in bodies) * 10.0 * 1e-3 * 60
    assert (imp > 0.35 * weight_impulse).all(), imp / weight_impulse   # limp legs: the robot sags while the shank contacts take 0.5-0.9 of the weight
    # 60 ms of free fall would be 18 mm and 0.6 m/s; the shank contact holds the robot (legs are limp, so it may sag a little)
    assert (z0 - s[:, lay.sl("POS")][:, 2] < 0.016).all(), z0 - s[:, lay.sl("POS")][:, 2]
    assert (np.abs(s[:, lay.sl("LINVEL")][:, 2]) < 0.45).all()      # free fall: 0.6 m/s (the identified Laikago table's limp legs fold at 0.31-0.41)
    env.close()


def _stand(env, model, s, nsub):
    lay = env.lay
    kp, kd = np.array(model["kp"]), np.array(model["kd"])
    for _ in range(nsub):
        qm = (s[lay.sl("Q")][model["joint_of_motor"]] - model["motor_offset"]) * model["motor_dir"]
        qdm = s[lay.sl("QD")][model["joint_of_motor"]] * model["motor_dir"]
        tau = -kp * (qm - model["init_motor_angles"]) - kd * qdm
        env.L.orc_physics_substep(env.h, P(s), P(np.ascontiguousarray(tau)))


def test_friction_anchor_caches_the_toe_contact_points():
    """Bullet's friction anchor (orr_model::friction_anchor, ABI v5): a toe's contact point is CACHED while its
    friction impulse stays inside the cone.  A robot standing still: every toe holds a cached point, the point on the plane does not
    move any more although the stance keeps rocking on its PD legs, the toes' tangential offsets from their anchors stay far below the
    threshold; lifting the robot by 3 cm drops all four points at the next sub-step; a reset clears them; without the flag nothing is cached."""
    for anchor in (1, 0):
        env, model = _make_env_model("laikago", friction_anchor=anchor)
        lay = env.lay
        s = env.state[0]
        dirj, offj, moj = pr.joint_maps(model)
        s[lay.sl("Q")] = np.array(model["init_motor_angles"])[moj] * dirj + offj
        s[lay.sl("QUAT")] = model["init_quat"]
        s[lay.sl("POS")] = model["init_pos"]
        s[lay.sl("QD")] = 0; s[lay.sl("LINVEL")] = 0; s[lay.sl("ANGVEL")] = 0
        _stand(env, model, s, 600)
        valid = s[lay.sl("ANCHOR_VALID")].copy()
        if not anchor:
            assert not valid.any() and not s[lay.sl("ANCHOR")].any()
            env.close()
            continue
        assert valid.all()
        a0 = s[lay.sl("ANCHOR")].reshape(4, 6).copy()
        _stand(env, model, s, 400)
        a1 = s[lay.sl("ANCHOR")].reshape(4, 6)
        np.testing.assert_array_equal(a0, a1)                       # kept, not replaced: the friction impulses of a quiet stance are inside the cone
        assert np.all(a1[:, 5] == 0.0)                              # the points on the plane lie ON the plane
        # the cached toe points sit where the toes are: world position of the local point = the plane point to well under a millimetre
        out, masses = np.zeros(34 * 3), np.zeros(13)
        env.L.orc_fk_probe(env.h, P(s), P(out), P(masses))
        toes = out[26 * 3:].reshape(8, 3)[1::2]
        assert np.all(np.abs(toes[:, :2] - a1[:, 3:5]) < 2e-3)
        s[lay.sl("POS")][2] += 0.03                                 # lift: distance > contact breaking threshold
        _stand(env, model, s, 1)
        assert not s[lay.sl("ANCHOR_VALID")].any()
        _stand(env, model, s, 300)
        assert s[lay.sl("ANCHOR_VALID")].all()                      # landed again: fresh points
        env.reset()
        assert not env.state[0][lay.sl("ANCHOR_VALID")].any()       # a new episode starts without cached points
        env.close()


def test_clip_stance_toes_touch_the_ground_on_the_shipped_tables():
    """Geometry pinned by in-tree DATA (no physics, no policy): the clips were made by inverse kinematics on the real URDFs, so in every
    frame the lowest toe of a correct table touches the ground.  The shipped tables do (median clearance of the lowest toe within a few
    millimetres); the Laikago hip height the policy search of round 5 preferred (-0.068 m) does not - it puts the stance toes 2 cm under
    the ground - which is why that ONE entry of the identified candidate was put back to this calibration (robots.py)."""
    def lowest_toe(robot, clip_name, **over):
        clip = motion.MotionClip(clip_name)
        cfg = config.make_config(1, mode="test", enable_randomizer=False, auto_reset=False)
        m = robots.ROBOTS[robot](**over)
        models = [None, None]
        t = robots.ROBOT_TYPE_ID[robot]
        models[t] = m
        env = ol.OracleEnv(cfg, models, [clip], 1, robot_type=t)
        lay = env.lay
        lows = []
        for f in clip.frames:
            s = env.state[0].copy()
            s[lay.sl("POS")] = f[:3]; s[lay.sl("QUAT")] = f[3:7]; s[lay.sl("Q")] = f[7:]
            out, masses = np.zeros(34 * 3), np.zeros(13)
            env.L.orc_fk_probe(env.h, P(s), P(out), P(masses))
            lows.append((out[26 * 3:].reshape(8, 3)[1::2, 2] - m["toe_radius"]).min())
        env.close()
        return float(np.median(lows))
    for clip_name in ("laikago_pace", "laikago_trot", "laikago_spin", "laikago_inplace_steps", "laikago_turn"):
        assert -0.003 < lowest_toe("laikago", clip_name) < 0.008, clip_name
    assert lowest_toe("laikago", "laikago_trot", hip_z=-0.068136) < -0.015          # the search's winner: toes under the ground
    assert -0.003 < lowest_toe("mini_cheetah", "minicheetah_trot") < 0.006
    assert lowest_toe("mini_cheetah", "minicheetah_trot", hip_z=0.0) < -0.006        # round 2's table, before the policy-based identification
