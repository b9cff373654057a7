"""A scripted stand-in for the pybullet client object -- NOT a physics engine.

Purpose: the reference's task / robot / wrapper code (imitation_task.py, minitaur.py,
quadruped_gym_env.py, wrapper_env.py, controllable_env_randomizer_from_config.py) takes its pybullet client as an
object and only ever talks to it through ~30 C-API style calls.  This class answers those calls from plain numpy state
so the reference's OWN Python can be driven end to end in the build container (where pybullet is not installed), and
every value it reads or writes can be recorded as a golden fixture for the oracle (tests/golden/make_golden_task.py).

What is scripted instead of simulated:
  * stepSimulation(): joints follow a damped response to the applied torques, the floating base is pulled towards the
    kinematic "ghost" reference model with noise.  The resulting per-sub-step states are recorded and INJECTED into the
    oracle's replay mode, so nothing here has to be (or is) physically right;
  * getLinkState(): a smooth made-up function of the body state (recorded, injected into the oracle's reward);
  * getContactPoints(): a scripted list of link indices.
What is restated from published pybullet semantics (the same caveat as tests/golden/_shims/pybullet_utils):
  getJointInfo's qIndex/uIndex numbering (7 + dof / 6 + dof, -1 for fixed joints), getEulerFromQuaternion /
  getQuaternionFromEuler (Bullet ZYX), invertTransform / multiplyTransforms, the (position, velocity, reaction, torque)
  tuples of getJointState(s) / getJointStateMultiDof.
The two URDFs are not available either: joint names / order below follow robots/laikago.py:31-47 and
robots/mini_cheetah.py:31-47 (MOTOR_NAMES, PATTERN) and the joint order of the motion-clip frames.

Test infrastructure only (my own code, not the reference's); never imported by the product package.
"""
import numpy as np

LAIKAGO_LEGS = ("FR", "FL", "RR", "RL")
MINICHEETAH_LEGS = ("fr", "fl", "hr", "hl")


def urdf_joint_names(urdf):
    """16 joints in URDF order: per leg (abduction, upper, lower, fixed toe)."""
    names = []
    if "laikago" in urdf:
        for leg in LAIKAGO_LEGS:
            names += ["%s_hip_motor_2_chassis_joint" % leg, "%s_upper_leg_2_hip_motor_joint" % leg,
                      "%s_lower_leg_2_upper_leg_joint" % leg, "jtoe%s" % leg]
    elif "mini_cheetah" in urdf:
        for leg in MINICHEETAH_LEGS:
            names += ["torso_to_abduct_%s_j" % leg, "abduct_%s_to_thigh_%s_j" % (leg, leg),
                      "thigh_%s_to_knee_%s_j" % (leg, leg), "toe_%s_joint" % leg]
    else:
        raise ValueError("unknown urdf %r" % (urdf,))
    return names


def qmul(a, b):
    x1, y1, z1, w1 = a
    x0, y0, z0, w0 = b
    return np.array([x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0, -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
                     x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0, -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def qrot(q, v):
    return qmul(qmul(q, np.array([v[0], v[1], v[2], 0.0])), qconj(q))[:3] / np.dot(q, q)


def qexp(w, dt):
    th = np.linalg.norm(w) * dt
    if th < 1e-12:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = np.asarray(w) / np.linalg.norm(w)
    return np.concatenate([ax * np.sin(th / 2), [np.cos(th / 2)]])


class _Body(object):
    def __init__(self, urdf, pos, orn, fixed):
        self.urdf = urdf
        self.fixed = fixed
        self.pos = np.array(pos, dtype=np.float64)
        self.orn = np.array(orn, dtype=np.float64)
        self.lin = np.zeros(3)
        self.ang = np.zeros(3)
        self.is_plane = "plane" in urdf
        self.names = [] if self.is_plane else urdf_joint_names(urdf)
        n = len(self.names)
        self.q = np.zeros(n)
        self.qd = np.zeros(n)
        self.tau = np.zeros(n)         # TORQUE_CONTROL forces of the current sub-step
        self.vel_motor_force = np.zeros(n)   # VELOCITY_CONTROL max forces (joint friction)
        self.revolute = np.array([(i % 4) != 3 for i in range(n)], dtype=bool)
        # made-up, pairwise distinct inertial data so that every changeDynamics call can be attributed to its link
        self.mass = {-1: 10.0}
        self.inertia = {-1: (0.11, 0.22, 0.33)}
        for i in range(n):
            self.mass[i] = 0.5 + 0.07 * i
            self.inertia[i] = (0.001 * (i + 1), 0.002 * (i + 1), 0.003 * (i + 1))
        self.dyn_calls = []            # (link, kwargs) of every changeDynamics on this body


class FakeBulletClient(object):
    # constants read by the reference (values are arbitrary tags)
    VELOCITY_CONTROL, TORQUE_CONTROL, POSITION_CONTROL = 0, 1, 2
    JOINT_FIXED = 4
    URDF_USE_SELF_COLLISION = 8
    ACTIVATION_STATE_SLEEP, ACTIVATION_STATE_ENABLE_SLEEPING, ACTIVATION_STATE_DISABLE_WAKEUP = 2, 1, 32
    COV_ENABLE_RENDERING, COV_ENABLE_GUI, COV_ENABLE_SINGLE_STEP_RENDERING = 7, 1, 13

    def __init__(self, connection_mode=None, seed=0):
        self.bodies = []
        self.engine = {}
        self.gravity = None
        self.time_step = None
        self.sim_steps = 0
        self.rng = np.random.RandomState(seed)
        self.link_state_log = []       # (body, link, position) of every getLinkState call
        self.substep_log = []          # per stepSimulation: {body: (state37 as in the oracle layout, applied torques[12])}
        self.contact_links = {}        # body id -> list of link indices reported by getContactPoints
        self.events = {}               # sim step index -> callable(world), applied at the end of that stepSimulation
        self.ghost_of = {}             # robot body id -> ghost (fixed-base reference model) body id
        self.foreign_dyn_calls = []    # changeDynamics on ids that are not bodies (minitaur.py:853-858 quirk)

    # ---- world setup ----------------------------------------------------------------------
    def setAdditionalSearchPath(self, path):
        pass

    def resetSimulation(self):
        pass

    def setPhysicsEngineParameter(self, **kw):
        self.engine.update(kw)

    def setTimeStep(self, dt):
        self.time_step = dt

    def setGravity(self, x, y, z):
        self.gravity = (x, y, z)

    def configureDebugVisualizer(self, *a, **k):
        pass

    def loadURDF(self, urdf, basePosition=(0, 0, 0), baseOrientation=(0, 0, 0, 1), useFixedBase=False, flags=0):
        b = _Body(urdf, basePosition, baseOrientation, bool(useFixedBase))
        self.bodies.append(b)
        bid = len(self.bodies) - 1
        if b.fixed and not b.is_plane:
            # ghosts are created in task order (imitation_task.py:176-178): the k-th ghost belongs to the k-th robot
            robots = [i for i, x in enumerate(self.bodies) if not x.fixed and not x.is_plane]
            ghosts = [i for i, x in enumerate(self.bodies) if x.fixed and not x.is_plane]
            self.ghost_of[robots[len(ghosts) - 1]] = bid
        return bid

    def getNumJoints(self, body):
        return len(self.bodies[body].names)

    def getJointInfo(self, body, j):
        b = self.bodies[body]
        dof_before = int(np.sum(b.revolute[:j]))
        rev = bool(b.revolute[j])
        return (j, b.names[j].encode("UTF-8"), 0 if rev else self.JOINT_FIXED, 7 + dof_before if rev else -1,
                6 + dof_before if rev else -1)

    def getDynamicsInfo(self, body, link):
        b = self.bodies[body]
        return (b.mass[link], 1.0, b.inertia[link])

    def changeDynamics(self, body, link, **kw):
        if not (0 <= body < len(self.bodies)) or self.bodies[body].is_plane:
            self.foreign_dyn_calls.append((body, link, dict(kw)))
            return
        self.bodies[body].dyn_calls.append((link, dict(kw)))

    def setCollisionFilterGroupMask(self, *a, **k):
        pass

    def changeVisualShape(self, *a, **k):
        pass

    # ---- state writes ---------------------------------------------------------------------
    def resetBasePositionAndOrientation(self, body, pos, orn):
        b = self.bodies[body]
        b.pos = np.array(pos, dtype=np.float64)
        b.orn = np.array(orn, dtype=np.float64)

    def resetBaseVelocity(self, body, lin, ang):
        b = self.bodies[body]
        b.lin = np.array(lin, dtype=np.float64)
        b.ang = np.array(ang, dtype=np.float64)

    def resetJointState(self, body, joint, targetValue, targetVelocity=0):
        b = self.bodies[body]
        b.q[joint] = float(targetValue)
        b.qd[joint] = float(targetVelocity)

    def resetJointStateMultiDof(self, body, joint, pose, vel):
        b = self.bodies[body]
        b.q[joint] = float(pose[0])
        b.qd[joint] = float(vel[0])

    def setJointMotorControl2(self, bodyIndex, jointIndex, controlMode, targetVelocity=0, force=0):
        b = self.bodies[bodyIndex]
        if controlMode == self.VELOCITY_CONTROL:
            b.vel_motor_force[jointIndex] = force
        elif controlMode == self.TORQUE_CONTROL:
            b.tau[jointIndex] = force

    def setJointMotorControlArray(self, bodyIndex, jointIndices, controlMode, forces=None):
        assert controlMode == self.TORQUE_CONTROL
        b = self.bodies[bodyIndex]
        for j, f in zip(jointIndices, forces):
            b.tau[j] = f

    # ---- state reads ----------------------------------------------------------------------
    def getBasePositionAndOrientation(self, body):
        b = self.bodies[body]
        return tuple(b.pos), tuple(b.orn)

    def getBaseVelocity(self, body):
        b = self.bodies[body]
        return tuple(b.lin), tuple(b.ang)

    def getJointStates(self, body, ids):
        b = self.bodies[body]
        return [(b.q[j], b.qd[j], (0.0,) * 6, b.tau[j]) for j in ids]

    def getJointStateMultiDof(self, body, j):
        b = self.bodies[body]
        if b.revolute[j]:
            return ([b.q[j]], [b.qd[j]], [0.0] * 6, [b.tau[j]])
        return ([], [], [0.0] * 6, [])

    def getLinkState(self, body, link):
        """Made-up smooth 'forward kinematics' (recorded and injected into the oracle; see the module docstring)."""
        b = self.bodies[body]
        leg = link // 4
        qa, qb, qc = b.q[4 * leg], b.q[4 * leg + 1], b.q[4 * leg + 2]
        sx = 1.0 if leg < 2 else -1.0
        sy = -1.0 if leg % 2 == 0 else 1.0
        local = np.array([0.2 * sx + 0.2 * np.sin(qb) + 0.15 * np.sin(qb + qc) * (1.0 if link % 4 == 3 else 0.5),
                          0.1 * sy + 0.1 * np.sin(qa),
                          -0.2 * np.cos(qb) - 0.15 * np.cos(qb + qc) * (1.0 if link % 4 == 3 else 0.5)])
        p = b.pos + qrot(b.orn, local)
        self.link_state_log.append((body, link, p.copy()))
        return (tuple(p), tuple(b.orn))

    def getContactPoints(self, bodyA=None, bodyB=None):
        return [(0, bodyA, bodyB, link, -1) for link in self.contact_links.get(bodyA, [])]

    # ---- transform helpers (Bullet conventions) ---------------------------------------------
    def invertTransform(self, position, orientation):
        qi = qconj(np.asarray(orientation, dtype=np.float64))
        return tuple(-qrot(qi, np.asarray(position, dtype=np.float64))), tuple(qi)

    def multiplyTransforms(self, positionA, orientationA, positionB, orientationB):
        qa = np.asarray(orientationA, dtype=np.float64)
        p = np.asarray(positionA, dtype=np.float64) + qrot(qa, np.asarray(positionB, dtype=np.float64))
        return tuple(p), tuple(qmul(qa, np.asarray(orientationB, dtype=np.float64)))

    def getEulerFromQuaternion(self, q):
        x, y, z, w = q
        sarg = -2.0 * (x * z - w * y)
        roll = np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z)
        pitch = -0.5 * np.pi if sarg <= -1.0 else (0.5 * np.pi if sarg >= 1.0 else np.arcsin(sarg))
        yaw = np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)
        return (roll, pitch, yaw)

    def getQuaternionFromEuler(self, rpy):
        hr, hp, hy = rpy[0] * 0.5, rpy[1] * 0.5, rpy[2] * 0.5
        cr, sr, cp, sp, cy, sy = np.cos(hr), np.sin(hr), np.cos(hp), np.sin(hp), np.cos(hy), np.sin(hy)
        return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                cr * cp * cy + sr * sp * sy)

    # ---- the scripted "physics" -------------------------------------------------------------
    def state37(self, body):
        """(pos3, quat4, linvel3, angvel3, q12, qd12) in the oracle's rigid-state layout (revolute joints, URDF order)."""
        b = self.bodies[body]
        return np.concatenate([b.pos, b.orn, b.lin, b.ang, b.q[b.revolute], b.qd[b.revolute]])

    def stepSimulation(self):
        dt = self.time_step
        rec = {}
        for bid, b in enumerate(self.bodies):
            if b.fixed or b.is_plane:
                continue
            tau = b.tau.copy()
            # joints: damped response to the applied torque (revolute joints only)
            b.qd = np.where(b.revolute, b.qd + dt * (8.0 * tau - 5.0 * b.qd), 0.0)
            b.q = b.q + dt * b.qd
            # base: pulled towards the ghost with noise
            g = self.bodies[self.ghost_of[bid]] if bid in self.ghost_of else None
            acc = self.rng.randn(3) * 2.0
            wacc = self.rng.randn(3) * 3.0
            if g is not None:
                acc += 60.0 * (g.pos - b.pos) + 12.0 * (g.lin - b.lin)
                dq = qmul(g.orn, qconj(b.orn))
                if dq[3] < 0:
                    dq = -dq
                wacc += 80.0 * 2.0 * dq[:3] + 15.0 * (g.ang - b.ang)
            b.lin = b.lin + dt * acc
            b.ang = b.ang + dt * wacc
            b.pos = b.pos + dt * b.lin
            o = qmul(qexp(b.ang, dt), b.orn)
            b.orn = o / np.linalg.norm(o)
            rec[bid] = tau[b.revolute]
        ev = self.events.pop(self.sim_steps, None)
        if ev is not None:
            ev(self)
        for bid in rec:
            # everything except the quaternion is rounded to float32 so that the fixtures can store it compactly
            b = self.bodies[bid]
            b.pos = b.pos.astype(np.float32).astype(np.float64)
            b.lin = b.lin.astype(np.float32).astype(np.float64)
            b.ang = b.ang.astype(np.float32).astype(np.float64)
            b.q = b.q.astype(np.float32).astype(np.float64)
            b.qd = b.qd.astype(np.float32).astype(np.float64)
            rec[bid] = (self.state37(bid), rec[bid])
        self.substep_log.append(rec)
        self.sim_steps += 1
