"""Robot model tables (data) for the Laikago and mini-cheetah quadrupeds.

The reference loads `laikago/laikago_toes_limits.urdf` and `mini_cheetah/mini_cheetah.urdf` from
the third-party `pybullet_data` package (robots/laikago.py:23, robots/mini_cheetah.py:23); those
files are neither under /root/reference nor in this image.  What IS in the reference tree and is
reproduced exactly here:

  * control constants: INIT_POSITION / INIT_QUAT / INIT_MOTOR_ANGLES / JOINT_DIRECTIONS /
    JOINT_OFFSETS / motor_kp / motor_kd / MOTOR_NAMES order (robots/laikago.py:29-66,
    robots/mini_cheetah.py:29-67);
  * leg kinematics (link lengths and the motor-angle sign conventions) used by the authors' own
    retargeting scripts: coxa/femur/tibia = 0.032875/0.25223/0.251 (Laikago) and 0.062/0.209/0.18
    (mini-cheetah), FK p = [t*s23 + f*s2, c*side*c1 + (t*c23 + f*c2)*s1, c*side*s1 - (t*c23 + f*c2)*c1]
    (task/motions/trans2minicheetah.m:3-5,28-30; trans_data.py:55-69), and the URDF<->kinematic
    angle maps (trans2minicheetah.m:8-9,32);
  * hip positions +-0.21 / +-0.1157 (robots/laikago.py:54-59);
  * URDF joint order = motion-frame joint order: FR, FL, RR, RL (Laikago; laikago.py:31-44) and
    fr, fl, hr, hl (mini-cheetah; sign pattern of the abduction columns of minicheetah_trot.txt),
    while the mini-cheetah MOTOR order is fl, hl, fr, hr (mini_cheetah.py:31-44).

Inertial parameters, collision proxies, joint limits and the toe radius are HAND-AUTHORED
(Unitree / MIT-published figures from memory) and are **parity-unpinned**; they live only in this
file so they can be swapped without touching a kernel.

Frames: "kinematic" body frame x forward, y left, z up; all link frames are parallel to it at zero
motor angles; the kinematic joint angle equals the motor angle `dir * (q_urdf - offset)`.
"""
import ctypes as C

import numpy as np

from . import _abi

LEG_SX = np.array([1.0, 1.0, -1.0, -1.0])   # front / rear
LEG_SY = np.array([-1.0, 1.0, -1.0, 1.0])   # right / left  (URDF leg order R, L, R, L)


def _sphere_inertia(m, r):
    return np.array([0.4 * m * r * r] * 3 + [0.0] * 3)


def _pa(m, d):
    """Parallel-axis inertia (xx yy zz xy xz yz) of a point mass m at offset d."""
    x, y, z = d
    return m * np.array([y * y + z * z, x * x + z * z, x * x + y * y, -x * y, -x * z, -y * z])


def _build(name, init_pos, init_quat, init_motor_angles, motor_dir, motor_offset, joint_of_motor, kp, kd,
           base_mass, base_inertia, hip_xy, hip_z, coxa, femur, tibia, pitch_axis,
           hip_m, hip_com, hip_I, up_m, up_com, up_I, lo_m, lo_com, lo_I, toe_m, toe_r,
           limits, chassis_half, hip_r, knee_r, foot_friction, shank_r=0.0, shank_at=0.0, contact_stiffness=0.0, contact_damping=0.0,
           friction_anchor=0, com_x=0.0):
    """com_x: base COM in front of the geometric centre of the four hips [m] (the hips and the chassis box sit that far BEHIND the COM frame)."""
    m = {
        "name": name,
        "init_pos": np.array(init_pos, dtype=np.float64),
        "init_quat": np.array(init_quat, dtype=np.float64),
        "init_motor_angles": np.array(init_motor_angles, dtype=np.float64),
        "motor_dir": np.array(motor_dir, dtype=np.float64),
        "motor_offset": np.array(motor_offset, dtype=np.float64),
        "joint_of_motor": np.array(joint_of_motor, dtype=np.int32),
        "kp": np.array(kp, dtype=np.float64),
        "kd": np.array(kd, dtype=np.float64),
        "base_mass": float(base_mass),
        "base_inertia": np.array(list(base_inertia) + [0.0, 0.0, 0.0]),
        "toe_radius": float(toe_r),
        "shank_radius": float(shank_r),
        "foot_friction": float(foot_friction),
        "contact_stiffness": float(contact_stiffness),
        "contact_damping": float(contact_damping),
        "friction_anchor": int(friction_anchor),
    }
    link_mass = np.zeros(12)
    link_com = np.zeros((12, 3))
    link_I = np.zeros((12, 6))
    link_Ipa = np.zeros((12, 6))
    link_group = np.zeros(12, dtype=np.int32)
    jpos = np.zeros((12, 3))
    jaxis = np.zeros((12, 3))
    jlo = np.zeros(12)
    jhi = np.zeros(12)
    toe_pos = np.zeros((4, 3))
    shank_pos = np.zeros((4, 3))
    lower_com = np.zeros((4, 3))
    fall_body, fall_pos, fall_radius = [], [], []
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (1, -1):
                fall_body.append(0)
                fall_pos.append([sx * chassis_half[0] - com_x, sy * chassis_half[1], sz * chassis_half[2]])
                fall_radius.append(0.0)
    for leg in range(4):
        sx, sy = LEG_SX[leg], LEG_SY[leg]
        j0 = 3 * leg
        # hip (abduction) link: "base" randomisation group (minitaur.py:828-829 chassis_link_ids)
        jpos[j0] = [sx * hip_xy[0] - com_x, sy * hip_xy[1], hip_z]
        jaxis[j0] = [1.0, 0.0, 0.0]
        link_mass[j0] = hip_m
        link_com[j0] = [hip_com[0] * sx, hip_com[1] * sy, hip_com[2]]
        link_I[j0] = list(hip_I) + [0, 0, 0]
        link_group[j0] = 0
        # upper leg: "leg" group (motor_link_ids)
        jpos[j0 + 1] = [0.0, sy * coxa, 0.0]
        jaxis[j0 + 1] = pitch_axis
        link_mass[j0 + 1] = up_m
        link_com[j0 + 1] = [up_com[0], up_com[1] * sy, up_com[2]]
        link_I[j0 + 1] = list(up_I) + [0, 0, 0]
        link_group[j0 + 1] = 1
        # lower leg + fixed toe merged for the dynamics: "leg" group (knee + foot link ids)
        jpos[j0 + 2] = [0.0, 0.0, -femur]
        jaxis[j0 + 2] = pitch_axis
        c_l = np.array(lo_com, dtype=np.float64)
        c_t = np.array([0.0, 0.0, -tibia])
        mm = lo_m + toe_m
        c = (lo_m * c_l + toe_m * c_t) / mm
        link_mass[j0 + 2] = mm
        link_com[j0 + 2] = c
        link_I[j0 + 2] = np.array(list(lo_I) + [0, 0, 0]) + _sphere_inertia(toe_m, toe_r)
        link_Ipa[j0 + 2] = _pa(lo_m, c_l - c) + _pa(toe_m, c_t - c)
        link_group[j0 + 2] = 1
        toe_pos[leg] = c_t
        shank_pos[leg] = [0.0, 0.0, -shank_at]     # on the shank axis, `shank_at` below the knee
        lower_com[leg] = c_l
        for k in range(3):
            jlo[j0 + k], jhi[j0 + k] = limits[k]
        fall_body.append(1 + j0)
        fall_pos.append([0.0, 0.0, 0.0])
        fall_radius.append(hip_r)
    for leg in range(4):
        fall_body.append(1 + 3 * leg + 1)
        fall_pos.append([0.0, 0.0, -femur])
        fall_radius.append(knee_r)
    m.update(link_mass=link_mass, link_com=link_com, link_inertia=link_I, link_inertia_pa=link_Ipa,
             link_group=link_group, joint_pos=jpos, joint_axis=jaxis, joint_lo=jlo, joint_hi=jhi,
             toe_pos=toe_pos, shank_pos=shank_pos, lower_com=lower_com, num_fall_proxies=len(fall_body),
             fall_body=np.array(fall_body, dtype=np.int32), fall_pos=np.array(fall_pos),
             fall_radius=np.array(fall_radius))
    assert m["num_fall_proxies"] <= _abi.MAX_FALL_PROXIES
    return m


# Round 4's table: Unitree / URDF figures from memory, hip height calibrated on the clips; kept for the record and for the identification
# tool, whose intervals and distances are stated relative to it (tools/laikago_identify.py: PARAMS).
LAIKAGO_R04 = dict(
    base_mass=13.715, base_inertia=[0.073348887, 0.250684593, 0.254469458], hip_xy=[0.21, 0.1157 - 0.032875], hip_z=-0.044, com_x=0.0,
    hip_m=1.095, hip_com=[0.0, 0.0, 0.0], hip_I=[0.00100, 0.00120, 0.00100],
    up_m=1.527, up_com=[0.0, 0.0, -0.04], up_I=[0.0078, 0.0081, 0.0012],
    lo_m=0.241, lo_com=[0.0, 0.0, -0.11], lo_I=[0.0013, 0.0013, 0.00005], toe_m=0.06, toe_r=0.0265,
    chassis_half=[0.27, 0.09, 0.055], hip_r=0.045, knee_r=0.035, foot_friction=1.0, shank_r=0.02, shank_at=0.03,
    contact_stiffness=0.0, contact_damping=0.0)


def laikago(**over):
    """robots/laikago.py constants + inertial / collision data.  over: replaces keyword arguments of _build (experiments on the
    hand-authored entries: tools/policy_probe.py --sensitivity, tools/laikago_identify.py).

    Round 5: the hand-authored entries are the candidate IDENTIFIED against the reference's PyBullet-trained policies with a held-out
    protocol fixed before the run (tools/laikago_identify.py; profiles/r05_laikago_identify.json): all of them varied
    at once inside stated plausible intervals (8112 candidates), fitted on laikago_trot + laikago_spin ONLY, the accepted candidate closest
    to round 4's table chosen, and only then run - once - on the two held-out policies: laikago_trot0 0.55 and laikago_pace 1.00 of the
    robots finish the 600-step episode (round-4 table: 0.00 / 1.00; fit policies: 0.00 / 0.00 -> 0.92 / 0.86).  The table below is that
    candidate with ONE entry corrected afterwards (hip_z, see there; decided on in-tree clip data and the fit policies, before its
    hold-out level was known): fit 0.90 / 0.88, held out 0.93 / 1.00 (profiles/r05_policy_probe.txt).  What the acceptance hangs
    on (fit-set ablation, profiles/r05_laikago_identify_ablation.txt): the toes' contact softness (k 25.3 kN/m, d 2.1 kN s/m: near the
    (30000, 1000) that pybullet_data's quadruped URDFs are remembered to carry), a toe friction of 0.5, the base COM 2.1 cm in front of
    the hips' centre, hips 1.7 cm further out and 1.5 cm further apart lengthwise, heavier distal links; NOT the fall proxies and not the
    solver constants (erp, warm start, contact margin: the table is accepted under the shipped orr_config as well, which is what ships).
    Round-4 values: LAIKAGO_R04 above."""
    return _build(**dict(dict(
        name="laikago",
        init_pos=[0, 0, 0.48], init_quat=[0.5, 0.5, 0.5, 0.5],               # laikago.py:48-49
        init_motor_angles=[0, 0.67, -1.25] * 4,                              # laikago.py:62
        motor_dir=[-1, 1, 1, 1, 1, 1, -1, 1, 1, 1, 1, 1],                    # laikago.py:50
        motor_offset=[0.0, -0.6, 0.66] * 4,                                  # laikago.py:52
        joint_of_motor=list(range(12)),                                      # laikago.py:31-44 = URDF order
        kp=[220.0] * 12, kd=[0.3, 2.0, 2.0] * 4,                             # laikago.py:65-66
        coxa=0.032875, femur=0.25223, tibia=0.251,                           # trans2minicheetah.m:3-5
        pitch_axis=[0.0, 1.0, 0.0],                                          # FK sign: trans_data.py:55-69
        # ---- identified entries (round-4 values in LAIKAGO_R04) ----
        base_mass=13.841, base_inertia=[1.2126 * x for x in (0.073348887, 0.250684593, 0.254469458)],
        # hip_z: the search's winner had -0.068136, but the hip plane's height is pinned by in-tree DATA, not by a policy: with -0.044 the stance
        # toes of every Laikago clip touch the ground (lowest toe clearance per frame: median +0.2 .. +5 mm over five clips) and the default pose
        # stands at INIT_POSITION's height; with the winner's value they sit 2 cm UNDER the ground (tools/diag/clip_toe_clearance.py).  The search
        # box should never have contained this entry; the fit-set ablation shows the fit does not care (0.90 with it put back), so the
        # calibrated value ships.  This is the ONE entry in which the shipped table differs from the recorded candidate.
        hip_xy=[0.22686, 0.097958], hip_z=-0.044, com_x=0.021374,
        hip_m=0.97061, hip_com=[0.0, 0.00082832, 0.0], hip_I=[1.5115 * x for x in (0.00100, 0.00120, 0.00100)],
        up_m=1.7255, up_com=[0.0085448, 0.03206, -0.044706], up_I=[1.5115 * x for x in (0.0078, 0.0081, 0.0012)],
        lo_m=0.36971, lo_com=[0.0082597, 0.0, -0.12418], lo_I=[1.5115 * x for x in (0.0013, 0.0013, 0.00005)],
        toe_m=0.13175, toe_r=0.026656,
        # toe contact (test mode keeps the table's friction; train mode draws U[0.5, 1.25] per episode like the reference)
        foot_friction=0.5, contact_stiffness=25335.0, contact_damping=2110.9,
        # termination-only fall proxies (chassis box corners, hip / knee spheres) and the second contact sphere of the lower leg (lower
        # legs are feet: minitaur.py:842-844); the ablation shows none of these matters to the fit
        chassis_half=[0.64568 * x for x in (0.27, 0.09, 0.055)], hip_r=0.0066596, knee_r=0.02321, shank_r=0.016522, shank_at=0.073903,
        # Unitree Laikago spec in motor convention: hip +-60 deg, thigh -30..225 deg, calf -159..-35 deg (not varied by the search's winner)
        limits=[(-1.0471975512, 1.0471975512), (-0.5235987756, 3.9269908170), (-2.7750735107, -0.6108652382)]), **over))


# Round 2's mini-cheetah entries (published MIT figures from memory) that round 3's identification moved: the reference point of the
# identification tools' intervals and distances (tools/mc_identify.py, tools/identify_r6.py).
MINI_CHEETAH_R02 = dict(toe_m=0.15, lo_m=0.064, lo_com=[0.0, 0.0, -0.061], lo_I=[0.000245, 0.000248, 0.000006], hip_z=0.0,
                        up_com=[0.0, 0.016, -0.02], shank_r=0.012, shank_at=0.02)


def mini_cheetah(**over):
    """robots/mini_cheetah.py constants + MIT mini-cheetah published inertial figures.  over: as in laikago()."""
    return _build(**dict(dict(
        name="mini_cheetah",
        init_pos=[0, 0, 0.28], init_quat=[0.0, 0.0, 0.0, 1.0],               # mini_cheetah.py:49-50
        init_motor_angles=[0, -0.78, 1.74] * 4,                              # mini_cheetah.py:63
        motor_dir=[1] * 12, motor_offset=[0.0] * 12,                         # mini_cheetah.py:51,53
        # MOTOR_NAMES order fl, hl, fr, hr (mini_cheetah.py:31-44) -> URDF legs fr, fl, hr, hl
        joint_of_motor=[3, 4, 5, 9, 10, 11, 0, 1, 2, 6, 7, 8],
        kp=[80.0] * 12, kd=[0.1, 1.0, 1.0] * 4,                              # mini_cheetah.py:66-67
        base_mass=3.3, base_inertia=[0.011253, 0.036203, 0.042673],
        # Round 3: the physically uncertain, hand-authored entries (distal masses and COMs, hip axis height, shank sphere) were IDENTIFIED
        # against the one PyBullet-derived artefact for this robot, the reference's shipped minicheetah_trot policy
        # (tools/mc_identify.py, criterion fixed beforehand: the accepted candidate CLOSEST to the round-2 table inside stated plausible
        # intervals; profiles/r03_mc_identify.json).  Round-2 values in brackets.  With them 90 % of 1024 robots
        # walk the full 600-step episode under that policy (round-2 table: 0 %, mean survival 158 steps).  Never touched: the control
        # constants above (mini_cheetah.py:49-67), link lengths and hip positions (trans2minicheetah.m:28-30), base / hip / thigh masses.
        hip_xy=[0.19, 0.049], hip_z=0.011, coxa=0.062, femur=0.209, tibia=0.18,         # trans2minicheetah.m:28-30; hip_z [0.0]
        pitch_axis=[0.0, -1.0, 0.0],                                         # trans2minicheetah.m:32 (q_urdf = -kin)
        hip_m=0.54, hip_com=[0.0, 0.036, 0.0], hip_I=[0.000381, 0.000560, 0.000444],
        up_m=0.634, up_com=[0.0, 0.016, -0.023], up_I=[0.001983, 0.002103, 0.000408],   # thigh COM z [-0.02]
        # shank: mass [0.064], COM [-0.061], transverse inertia of a slender 0.18 m rod of that mass + 7e-5 [0.000245, 0.000248]
        lo_m=0.091, lo_com=[0.0, 0.0, -0.073], lo_I=[0.000316, 0.000316, 0.000006],
        # toe link mass [0.15]: the decisive entry - the policy needs ~0.3 kg below the knee (one-at-a-time sweep: 0 % of the robots
        # finish with a 0.135 kg toe, 77 % with 0.19 kg, 84 % with 0.22 kg)
        toe_m=0.214, toe_r=0.0175,
        limits=[(-1e9, 1e9)] * 3,                                            # continuous joints
        # knee proxy radius 0: with a finite knee sphere the shipped minicheetah_trot policy is stopped by knee
        # "contacts" within ~10 steps while still upright; the thigh/shank of this robot are thin plates
        chassis_half=[0.19, 0.049, 0.05], hip_r=0.04, knee_r=0.0, foot_friction=1.0, shank_r=0.0094, shank_at=0.0196), **over))   # shank sphere [0.012 @ 0.02]


ROBOTS = {"laikago": laikago, "mini_cheetah": mini_cheetah}
ROBOT_TYPE_ID = {"laikago": 0, "mini_cheetah": 1}


def to_struct(model):
    """dict of arrays -> ctypes OrrModel (float32 table handed to orr_set_model)."""
    s = _abi.OrrModel()
    for fname, _ in _abi.OrrModel._fields_:
        val = model[fname]
        field = getattr(s, fname)
        if isinstance(field, C.Array):
            arr = np.ctypeslib.as_array(field)
            v = np.asarray(val)
            if v.shape != arr.shape:  # padded tables (fall proxies)
                pad = np.zeros(arr.shape, dtype=arr.dtype)
                pad[tuple(slice(0, n) for n in v.shape)] = v
                v = pad
            arr[...] = v.astype(arr.dtype)
        else:
            setattr(s, fname, val)
    return s
