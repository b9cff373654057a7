"""ctypes mirror of include/openroborl_hip.h (struct layouts and constants only)."""
import ctypes as C

ABI_VERSION = 5
NUM_MOTORS = 12
POSE_DIM = 19
VEL_DIM = 18
PROPRIO_DIM = 84
TARGET_DIM = 76
OBS_DIM = 160
MAX_ROBOT_TYPES = 32
MAX_CLIPS = 16
MAX_FALL_PROXIES = 16
RING_DEPTH = 44
RING_ENTRY = 20
STATE_STRIDE = 1216

DONE_CONTACT_FALL, DONE_ROOT_POS, DONE_ROOT_ROT, DONE_TIME_LIMIT, DONE_NAN, DONE_MOTION_OVER = 1, 2, 4, 8, 16, 32

FLAG_AUTO_RESET = 1
FLAG_RANDOMIZER = 2
FLAG_CYCLE_SYNC = 4
FLAG_LEGACY_GRID = 8
FLAG_CURRICULUM = 16

CLIP_WRAP, CLIP_CYCLE_POS, CLIP_CYCLE_ROT = 1, 2, 4

CNT_TOTAL_STEP_COUNT = 0
CNT_DONE_ACCUM = 1
CNT_TICKET = 2
CNT_TOTAL_TIMESTEPS = 3
CNT_EPISODES = 4
CNT_EPLOG_DROPPED = 5
NUM_COUNTERS = 8


class OrrConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("num_robots", C.c_int32),
        ("action_repeat", C.c_int32),
        ("solver_iters", C.c_int32),
        ("sim_dt", C.c_float),
        ("gravity_z", C.c_float),
        ("reward_w", C.c_float * 5),
        ("reward_scale", C.c_float * 6),
        ("tar_frame_steps", C.c_int32 * 4),
        ("ref_state_init_prob", C.c_float),
        ("warmup_time", C.c_float),
        ("ep_len_start", C.c_int32),
        ("ep_len_end", C.c_int32),
        ("curriculum_steps", C.c_int64),
        ("seed", C.c_uint64),
        ("flags", C.c_int32),
        ("contact_erp", C.c_float),
        ("contact_margin", C.c_float),
        ("warmstart_factor", C.c_float),
        ("max_coord_velocity", C.c_float),
        ("plane_friction", C.c_float),
        ("limit_activation", C.c_float),
        ("max_angle_change", C.c_float),
        ("dist_fail_threshold", C.c_float),
        ("rot_fail_threshold", C.c_float),
        ("friction_erp", C.c_float),
    ]


class OrrModel(C.Structure):
    _fields_ = [
        ("init_pos", C.c_float * 3),
        ("init_quat", C.c_float * 4),
        ("init_motor_angles", C.c_float * 12),
        ("motor_dir", C.c_float * 12),
        ("motor_offset", C.c_float * 12),
        ("joint_of_motor", C.c_int32 * 12),
        ("kp", C.c_float * 12),
        ("kd", C.c_float * 12),
        ("base_mass", C.c_float),
        ("base_inertia", C.c_float * 6),
        ("link_mass", C.c_float * 12),
        ("link_com", (C.c_float * 3) * 12),
        ("link_inertia", (C.c_float * 6) * 12),
        ("link_inertia_pa", (C.c_float * 6) * 12),
        ("link_group", C.c_int32 * 12),
        ("joint_pos", (C.c_float * 3) * 12),
        ("joint_axis", (C.c_float * 3) * 12),
        ("joint_lo", C.c_float * 12),
        ("joint_hi", C.c_float * 12),
        ("toe_pos", (C.c_float * 3) * 4),
        ("lower_com", (C.c_float * 3) * 4),
        ("toe_radius", C.c_float),
        ("shank_pos", (C.c_float * 3) * 4),
        ("shank_radius", C.c_float),
        ("foot_friction", C.c_float),
        ("contact_stiffness", C.c_float),
        ("contact_damping", C.c_float),
        ("friction_anchor", C.c_int32),
        ("num_fall_proxies", C.c_int32),
        ("fall_body", C.c_int32 * MAX_FALL_PROXIES),
        ("fall_pos", (C.c_float * 3) * MAX_FALL_PROXIES),
        ("fall_radius", C.c_float * MAX_FALL_PROXIES),
    ]


def fill(struct, name, values):
    """Assign a (nested) python sequence / scalar to a ctypes struct field."""
    import numpy as np
    field = getattr(struct, name)
    if isinstance(field, C.Array):
        arr = np.ctypeslib.as_array(field)
        arr[...] = np.asarray(values, dtype=arr.dtype).reshape(arr.shape)
    else:
        setattr(struct, name, values)


class OrrPolicyNet(C.Structure):
    """include/openroborl_policy.h: orr_policy_net (packed weights + biases of the actor and the critic)."""
    _fields_ = [(n, C.c_void_p) for n in ("w0_pi", "b0_pi", "w1_pi", "b1_pi", "w2_pi", "b2_pi",
                                            "w0_vf", "b0_vf", "w1_vf", "b1_vf", "w2_vf", "b2_vf")]


class OrrColsumJob(C.Structure):
    """include/openroborl_learner.h: orr_colsum_job."""
    _fields_ = [("partials", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]
