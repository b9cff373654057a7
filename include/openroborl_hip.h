/*
 * openroborl_hip.h -- C-ABI of the MI355X-native vectorised quadruped imitation environment.
 *
 * Drop-in boundary.  The reference has no FFI: its boundary is the
 * duck-typed Gym env consumed by PPOImitation / traj_segment_generator
 *   (OpenRoboRL/envs/quadruped_robot/wrapper_env.py:58-107  WrapperEnv.step / reset,
 *    OpenRoboRL/envs/quadruped_robot/quadruped_gym_env.py:63-104,213-239  reset / _step).
 * Every entry point below names the reference call(s) it replaces.  The Python host side
 * (openroborl_amd/env.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C, no torch types; all device buffers are CALLER-OWNED (torch tensors' data_ptr()),
 *     the library allocates only its small model/clip tables inside orr_create();
 *   - every launch is asynchronous on the hipStream_t passed as void* (NULL = default stream);
 *   - return value 0 = OK, negative = error (text via orr_last_error()); nothing throws;
 *   - quaternions are [x, y, z, w] (reference: envs/utilities/pose3d.py:31,132-136);
 *   - arithmetic type on device: float32.
 */
#ifndef OPENROBORL_HIP_H_
#define OPENROBORL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORR_ABI_VERSION 5

#define ORR_NUM_MOTORS 12 /* laikago.py:29, mini_cheetah.py:29 */
#define ORR_NUM_LEGS 4
#define ORR_POSE_DIM 19     /* root pos 3 + root quat 4 + 12 joints  (motion_data.py:46-49)   */
#define ORR_VEL_DIM 18      /* root vel 3 + root ang vel 3 + 12 joint rates                   */
#define ORR_PROPRIO_DIM 84  /* IMU 12 | LastAction 36 | MotorAngle 36 (quadruped_gym_env.py:289-320) */
#define ORR_NUM_TAR_FRAMES 4
#define ORR_TARGET_DIM (ORR_NUM_TAR_FRAMES * ORR_POSE_DIM) /* imitation_task.py:254-301 */
#define ORR_OBS_DIM (ORR_PROPRIO_DIM + ORR_TARGET_DIM)     /* 160, wrapper_env.py:109-125 */
#define ORR_MAX_ROBOT_TYPES 32 /* slots of the device model table (a batch may mix that many tables: heterogeneous batches, table sweeps) */
#define ORR_MAX_CLIPS 16
#define ORR_MAX_FALL_PROXIES 16
#define ORR_RING_DEPTH 44   /* latency <= 0.04 s -> int(0.04/0.001)+1 = 41 entries needed      */
#define ORR_RING_ENTRY 20   /* 12 motor angles + 4 rel. quat + 3 rpy rate (+1 pad)             */

/* ---------------------------------------------------------------------------------------
 * Per-robot state record: ORR_STATE_STRIDE 32-bit words, one record per robot, records
 * contiguous ([N, ORR_STATE_STRIDE] tensor).  A 16-lane group (a quarter of a wavefront: four
 * robots per wave) owns one record: lane k moves words k, k+16, ... in 16-byte pieces
 * (coalesced).  X(name, words, kind) with kind F = float32, I = int32.
 * Reference provenance of every group: SURVEY.md Appendix A.1.
 * ------------------------------------------------------------------------------------- */
#define ORR_STATE_FIELDS(X)                                                                   \
  /* rigid state, Bullet conventions: base COM frame, world-frame velocities */               \
  X(POS, 3, F) X(QUAT, 4, F) X(LINVEL, 3, F) X(ANGVEL, 3, F)                                  \
  X(Q, 12, F)  /* URDF joint coordinates, URDF joint order */                                 \
  X(QD, 12, F)                                                                                \
  /* actuator (minitaur.py:160-164,280-293; action_filter.py:99-127), motor order */          \
  X(ACTION, 12, F) X(FILTER_ACTION, 12, F) X(LAST_ACTION, 12, F)                              \
  X(XHIST, 24, F) /* x[n-1] (12), x[n-2] (12) */                                              \
  X(YHIST, 24, F)                                                                             \
  /* 3-deep sensor histories, newest first (sensor_wrappers.py:122-142) */                    \
  X(IMU_HIST, 12, F) X(LASTACT_HIST, 36, F) X(MOTORANG_HIST, 36, F)                           \
  /* task (imitation_task.py:105-139) */                                                      \
  X(TIME_OFFSET, 1, F) X(ORIGIN_POS, 3, F) X(ORIGIN_ROT, 4, F) X(PREV_PHASE, 1, F)            \
  X(REF_POSE, 19, F) X(REF_VEL, 18, F)                                                        \
  /* per-episode randomised parameters (controllable_env_randomizer_from_config.py) */        \
  X(STRENGTH, 12, F) X(LATENCY, 1, F) X(FOOT_MU, 1, F) X(KNEE_FRICTION, 4, F)                 \
  X(MASS_RATIO, 2, F) X(INERTIA_RATIO, 2, F) X(BASE_DAMPING, 2, F)                            \
  /* contact warm start: per leg (normal, t1, t2) impulses */                                 \
  X(LAMBDA, 12, F)                                                                            \
  X(EP_RETURN, 1, F) X(LAST_EP_RETURN, 1, F) X(GRID_OFFSET, 2, F)                             \
  /* integers */                                                                              \
  X(STATE_ACTION_COUNTER, 1, I) /* minitaur.py:186-189 */                                     \
  X(STEP_COUNTER, 1, I)                                                                       \
  X(FILTER_VALID, 1, I)  /* _filter_action is not None (minitaur.py:450-453) */               \
  X(RING_LEN, 1, I) X(RING_HEAD, 1, I)                                                        \
  X(EPISODE_IDX, 1, I)   /* RNG stream selector */                                            \
  X(EP_STEP, 1, I)       /* env_step_counter, per robot (quadruped_gym_env.py:88,237) */      \
  X(WARMUP, 1, I)        /* _curr_episode_warmup */                                           \
  X(MAX_EP_STEPS, 1, I)  /* wrapper_env.py:151-159 */                                         \
  X(ROBOT_TYPE, 1, I) X(CLIP_ID, 1, I)                                                        \
  X(ROBOT_INDEX, 1, I)   /* global robot index (RNG key, grid slot) */                        \
  X(LAST_EP_LEN, 1, I)                                                                        \
  X(DONE_REASON, 1, I)   /* of the last step; survives the (auto-)reset that follows it */    \
  X(RESERVED_I, 2, I)                                                                         \
  /* latency ring (minitaur.py:127,313-357): ORR_RING_DEPTH entries of ORR_RING_ENTRY */      \
  X(RING, ORR_RING_DEPTH * ORR_RING_ENTRY, F)                                                 \
  /* ABI v5, behind the ring (never staged with the head): Bullet's friction anchors, one cached contact point per toe             \
   * (btPersistentManifold::replaceContactPoint): the point on the toe in the lower-leg link frame (3) and on the plane in world (3) */ \
  X(ANCHOR, 4 * 6, F)                                                                         \
  X(ANCHOR_VALID, 4, I)  /* per leg: 1 = that toe holds a cached point */

enum orr_state_offset_e {
#define ORR_X_OFF(name, words, kind) ORR_OFF_##name, ORR_OFFEND_##name = ORR_OFF_##name + (words)-1,
  ORR_STATE_FIELDS(ORR_X_OFF)
#undef ORR_X_OFF
      ORR_STATE_WORDS
};
#define ORR_STATE_STRIDE 1216 /* ORR_STATE_WORDS rounded up to a multiple of 64 (19 x 64) */
typedef char orr_state_words_fit_the_stride[(ORR_STATE_WORDS <= ORR_STATE_STRIDE) ? 1 : -1]; /* C99 static assertion */

/* done reasons (bit mask) */
#define ORR_DONE_CONTACT_FALL 1 /* imitation_task.py:536-546 */
#define ORR_DONE_ROOT_POS 2     /* imitation_task.py:553-556 */
#define ORR_DONE_ROOT_ROT 4     /* imitation_task.py:558-565 */
#define ORR_DONE_TIME_LIMIT 8   /* wrapper_env.py:79 */
#define ORR_DONE_NAN 16         /* new: non-finite state guard */
#define ORR_DONE_MOTION_OVER 32 /* imitation_task.py:532,567 + motion_data.py:265-276: a non-looping clip played to its end */

/* orr_config.flags */
#define ORR_FLAG_AUTO_RESET 1        /* per-robot masked auto-reset inside orr_step (native mode) */
#define ORR_FLAG_RANDOMIZER 2        /* run.py:205-206: enabled when mode == "train" */
#define ORR_FLAG_CYCLE_SYNC 4        /* run.py:61 enable_cycle_sync=True */
#define ORR_FLAG_LEGACY_GRID 8       /* minitaur.py:246-248: 2 m grid slots (float32 loses precision at N>>100) */
#define ORR_FLAG_CURRICULUM 16       /* wrapper_env.py:147-159 */

/* clip flags (motion_data.py:83-97) */
#define ORR_CLIP_WRAP 1
#define ORR_CLIP_CYCLE_POS 2
#define ORR_CLIP_CYCLE_ROT 4

typedef struct orr_config {
  int32_t abi_version;     /* ORR_ABI_VERSION */
  int32_t num_robots;      /* robots on THIS device (shard size) */
  int32_t action_repeat;   /* 33: laikago.py:26 */
  int32_t solver_iters;    /* int(300/33) = 9: quadruped_gym_env.py:177-178 */
  float sim_dt;            /* 0.001: pybullet_sim_param.yaml:3 */
  float gravity_z;         /* -10: quadruped_gym_env.py:200 */
  float reward_w[5];       /* pose, velocity, end-effector, root pose, root velocity: imitation_task.py:45-49 */
  float reward_scale[6];   /* pose, vel, end-eff, end-eff height, root pose, root vel: imitation_task.py:50-55 */
  int32_t tar_frame_steps[ORR_NUM_TAR_FRAMES]; /* run.py:62 [1,2,10,30] */
  float ref_state_init_prob; /* run.py:63 0.9 */
  float warmup_time;         /* run.py:64 0.25 */
  int32_t ep_len_start;      /* run.py:54 20 */
  int32_t ep_len_end;        /* run.py:55 600 */
  int64_t curriculum_steps;  /* ceil(3e7 / num_procs): wrapper_env.py:45-46 */
  uint64_t seed;
  int32_t flags;
  /* physics-engine constants (parity-unpinned).  First value: what the Python host passes since round 6 = what PyBullet's
     createEmptyDynamicsWorld is remembered to set; in brackets the Bullet library's default (rounds 1-5).  DESIGN.md section 4. */
  float contact_erp;         /* 0.08 [0.2] */
  float contact_margin;      /* 0.004 [0.02] contact breaking threshold: a normal row exists / a termination proxy counts inside it */
  float warmstart_factor;    /* 0.1 [0.85] */
  float max_coord_velocity;  /* 100; orr_create refuses sqrt(3) * max_coord_velocity * sim_dt / 2 >= 0.2 (the base may turn at most 0.4 rad per sub-step) */
  float plane_friction;      /* 1.0 plane_implicit.urdf */
  float limit_activation;    /* 0.1 rad: a joint-limit row exists iff the joint is this close */
  float max_angle_change;    /* 0.2: laikago.py:71 MAX_MOTOR_ANGLE_CHANGE_PER_STEP */
  float dist_fail_threshold; /* 1.0: imitation_task.py:518 */
  float rot_fail_threshold;  /* pi/2 */
  float friction_erp;        /* ABI v5: 0.2 = Bullet's m_frictionERP: share of a friction anchor's tangential drift removed per sub-step
                                (friction rows get -drift * friction_erp / dt; only toes with orr_model::friction_anchor) */
} orr_config;

/* Robot model table (data, swappable without touching kernels).  All geometry is given in the
 * "kinematic" body frame (x forward, y left, z up) at zero motor angles, where every link frame
 * is parallel to the base frame.  Base orientation used by the dynamics is
 * QUAT (x) INIT_QUAT^-1 (the reference's "relative orientation", minitaur.py:325-331), so the
 * stored QUAT stays in the URDF convention of the motion clips.
 * Bodies: 0 = base; 1 + 3*leg + k, k = 0 hip(abduction) link, 1 upper leg, 2 lower leg (+ fixed toe
 * merged for the dynamics).  Joint j = 3*leg + k (URDF joint order) connects body j+1 to its parent. */
typedef struct orr_model {
  float init_pos[3];                    /* laikago.py:48 / mini_cheetah.py:49 */
  float init_quat[4];                   /* laikago.py:49 / mini_cheetah.py:50 */
  float init_motor_angles[ORR_NUM_MOTORS]; /* motor order; laikago.py:62 */
  float motor_dir[ORR_NUM_MOTORS];      /* JOINT_DIRECTIONS, motor order; +1 or -1 (orr_set_model refuses anything else) */
  float motor_offset[ORR_NUM_MOTORS];   /* JOINT_OFFSETS, motor order */
  int32_t joint_of_motor[ORR_NUM_MOTORS]; /* URDF joint index driven by motor m (MOTOR_NAMES order) */
  float kp[ORR_NUM_MOTORS];             /* motor order; laikago.py:65 */
  float kd[ORR_NUM_MOTORS];
  float base_mass;
  float base_inertia[6];                /* xx yy zz xy xz yz about the base COM */
  float link_mass[12];
  float link_com[12][3];                /* in link frame (origin = joint origin) */
  float link_inertia[12][6];            /* about link COM; scales with the inertia ratio */
  float link_inertia_pa[12][6];         /* parallel-axis part of a merged (lower leg + toe) link; scales with the mass ratio */
  int32_t link_group[12];               /* 0 = "base" randomisation group (hip links), 1 = "leg" group (minitaur.py:812-851) */
  float joint_pos[12][3];               /* joint origin in the parent link frame */
  float joint_axis[12][3];              /* unit axis; kinematic angle = dir * (q_urdf - offset) */
  float joint_lo[12];                   /* limits in kinematic (motor-convention) angle */
  float joint_hi[12];
  float toe_pos[4][3];                  /* toe sphere centre in the lower-leg link frame (= toe link COM) */
  float lower_com[4][3];                /* lower leg's own COM (end-effector reward, imitation_task.py:441-446) */
  float toe_radius;
  float shank_pos[4][3];                /* second contact sphere of the lower leg ("shank"; lower legs are feet: minitaur.py:842-844), */
  float shank_radius;                   /* in the lower-leg link frame; radius 0 = none.  A leg touches the ground with whichever of its
                                           two spheres (toe, shank) is lower: one contact point per leg and sub-step */
  float foot_friction;                  /* default lateral friction of toe / lower leg */
  float contact_stiffness;              /* URDF <contact><stiffness/><damping/> of the TOE link (Bullet's BT_CONTACT_FLAG_CONTACT_STIFFNESS_DAMPING): */
  float contact_damping;                /* the toe's normal row gets cfm = 1 / (dt k + d), erp = dt k / (dt k + d) instead of the global
                                           contact_erp and cfm 0 (btMultiBodyConstraintSolver::setupMultiBodyContactConstraint).
                                           stiffness <= 0 = rigid contact (the global pair); ABI v4.  These are the values of the PAIR: Bullet combines
                                           the two bodies' entries (btManifoldResult: k = 1 / (1 / k_toe + 1 / k_plane), d = d_toe + d_plane; the
                                           plane's defaults - a huge stiffness and a damping of 0.1 - change (30000, 1000) by 0.01 %), the caller folds
                                           them in if it has them.  The configuration's sim_dt and contact_erp must be final before the model is set: the
                                           row's cfm / erp are folded when the model is set */
  int32_t friction_anchor;              /* URDF <contact><friction_anchor/> of the TOE link (ABI v5): the toe's contact point is CACHED while its
                                           friction impulse stays inside the cone (btPersistentManifold::replaceContactPoint), and the friction
                                           rows pull the cached pair of points together (orr_config::friction_erp).  0 = off */
  int32_t num_fall_proxies;             /* termination-only collision spheres on non-foot links */
  int32_t fall_body[ORR_MAX_FALL_PROXIES];
  float fall_pos[ORR_MAX_FALL_PROXIES][3];
  float fall_radius[ORR_MAX_FALL_PROXIES];
} orr_model;

/* device-side global counters, caller-owned int64[ORR_NUM_COUNTERS], zero-initialised */
enum orr_counter_e {
  ORR_CNT_TOTAL_STEP_COUNT = 0, /* wrapper_env.py:47,82-83 curriculum counter */
  ORR_CNT_DONE_ACCUM = 1,       /* per-launch scratch */
  ORR_CNT_TICKET = 2,           /* per-launch scratch */
  ORR_CNT_TOTAL_TIMESTEPS = 3,  /* robot-steps executed (ppo_imitation.py:421) */
  ORR_CNT_EPISODES = 4,         /* finished episodes appended to the episode log */
  ORR_CNT_EPLOG_DROPPED = 5,
  ORR_NUM_COUNTERS = 8
};

typedef struct orr_handle orr_handle;

const char* orr_last_error(void);
int32_t orr_abi_version(void);
const char* orr_source_hash(void);              /* hash of the sources this library was built from (host loader: stale-build check) */
int32_t orr_state_stride(void);                 /* ORR_STATE_STRIDE */
int32_t orr_layout_count(void);                 /* number of fields in ORR_STATE_FIELDS */
const char* orr_layout_name(int32_t i);
int32_t orr_layout_offset(int32_t i);
int32_t orr_layout_size(int32_t i);
int32_t orr_layout_is_int(int32_t i);
int32_t orr_sizeof_config(void);
int32_t orr_sizeof_model(void);

/* replaces LocomotionGymEnv._init world setup (quadruped_gym_env.py:158-211) */
int32_t orr_create(const orr_config* cfg, orr_handle** out);
int32_t orr_destroy(orr_handle* h);

/* replaces LocomotionGymEnv.seed (quadruped_gym_env.py:59-61): new key of the counter-based RNG, used from the next reset on */
int32_t orr_set_seed(orr_handle* h, uint64_t seed);

/* replaces loadURDF + _build_urdf_ids + _record_*_from_urdf (minitaur.py:201-230,812-851,897-903) */
int32_t orr_set_model(orr_handle* h, int32_t robot_type, const orr_model* model_host);

/* replaces MotionData.load results (motion_data.py:72-112): frames [F,19] and frame velocities
 * [F,18] are DEVICE pointers to post-processed data; cycle_delta = {dx, dy, dz(=0), dheading}.
 * frame_dt = the clip's "FrameDuration" as a DOUBLE (ABI v3): the sampler keeps the motion time in float64 like the reference
 * (shipped clips use 1/24 s and 0.03 s; 20 s into an episode a float32 frame time misplaces the blend factor by 1e-5, which
 * the O(100) jumps of the finite-difference frame velocities turn into 1e-3). */
int32_t orr_set_motion(orr_handle* h, int32_t clip_id, const float* frames_dev, const float* frame_vels_dev,
                       int32_t num_frames, double frame_dt, int32_t clip_flags, const float cycle_delta[4]);

/* bind caller-owned device buffers: state [N, ORR_STATE_STRIDE] words (16-byte aligned), counters int64[8],
 * episode log float[ep_log_capacity][2] = (return, length) (may be NULL / 0). */
int32_t orr_bind(orr_handle* h, void* state_dev, int64_t* counters_dev, float* ep_log_dev, int32_t ep_log_capacity);

/* replaces WrapperEnv.reset (wrapper_env.py:87-107): mask_dev NULL = all robots; obs_dev [N,160]
 * (rows of robots that are not reset are left untouched). */
int32_t orr_reset(orr_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream);

/* replaces WrapperEnv.step (wrapper_env.py:58-85): actions [N,12] policy outputs (already clipped to
 * +-2pi by the caller, imitation_runners.py:140-143; NOT modified, unlike minitaur.py:281);
 * obs [N,160] (16-byte aligned), reward [N], done [N] (uint8).
 * One launch.  The library holds two builds of the same kernel source and picks by batch size: up to 4 robots x #SIMDs of the device
 * (4096 on an MI355X) one wave per SIMD, above that two waves per SIMD (identical results; the environment variable
 * ORR_STEP_WAVES_PER_EU = 1 | 2, read by orr_create, forces one of them - measurements and tests only). */
int32_t orr_step(orr_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                 void* stream);

/* replaces the rank-local side of MPI allgather((ep_lens, ep_rets)) + allreduce(total_timestep) at a rollout boundary
 * (agents/ppo_imitation.py:405-423): packs the episode log into the fixed-size float64 payload of ONE all-gather,
 *   out_dev[6 + 2 * capacity] = [n_listed, total_timesteps, n_dropped, n_episodes, sum_ret, sum_len, ret[capacity], len[capacity]],
 * and clears the log (one launch, no host sync; n_episodes / sums cover every logged episode, the lists the first `capacity`). */
int32_t orr_episode_stats(orr_handle* h, double total_timesteps, int32_t capacity, double* out_dev, void* stream);

/* parity / debug entry (not part of the drop-in surface): nsub physics sub-steps with fixed motor torques
 * [N,12] applied as tau_urdf = tau * JOINT_DIRECTIONS; fall_dev [N] receives the fall-proxy flag (may be NULL). */
int32_t orr_debug_physics(orr_handle* h, const float* torques_dev, uint8_t* fall_dev, int32_t nsub, void* stream);

/* parity / debug entries (not part of the drop-in surface): WrapperEnv.step / reset with the physics engine REPLAYED.
 * traj [N][action_repeat][37] = rigid state (POS QUAT LINVEL ANGVEL Q QD) after each sub-step, eff [N][2][8][3] = link positions
 * for the end-effector reward ([0] robot, [1] reference model; lower leg, toe per leg), fall [N] = non-foot ground contact,
 * tau_out [N][action_repeat][12] receives the motor torques; uniforms [N][28] = the draws of the reset in [0, 1).  They exist so
 * that the HIP path can be compared with fixtures recorded from the reference's own Python (tests/golden/make_golden_task.py). */
int32_t orr_debug_replay_step(orr_handle* h, const float* actions_dev, const float* traj_dev, const float* eff_dev,
                              const uint8_t* fall_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                              float* tau_out_dev, void* stream);
int32_t orr_debug_replay_reset(orr_handle* h, const float* uniforms_dev, float* obs_dev, void* stream);

/* last launch durations in ms measured with hipEvents on the launch stream (bench only; syncs) */
int32_t orr_time_steps(orr_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                       void* stream, int32_t num_steps, float* total_ms_out);

/* measurement input (bench only; no reference counterpart): the policy-free stress actions of SURVEY.md section 8d (i) in ONE launch for
 * any mix of robot types: actions[i][m] = clip((obs[i][84 + 7 + joint_of_motor[m]] - motor_offset[m]) * motor_dir[m]
 * - init_motor_angles[m] + noise[i][m], +-2 pi) with the table of robot i's type (obs [N,160], noise and actions [N,12], device). */
int32_t orr_stress_actions(orr_handle* h, const float* obs_dev, const float* noise_dev, float* actions_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENROBORL_HIP_H_ */
