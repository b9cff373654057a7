// orr_device.h -- device-side building blocks of the quadruped env kernels (gfx950, wave64).
//
// A wavefront serves kRPW = 4 robots, 16 lanes each (ORR_LANES_PER_ROBOT; 32 / 64 are kept for experiments).
// Lane roles inside a robot's lane group change phase by phase:
//   motors      lanes 0..11   action filter / interpolation / clip / PD torque (minitaur.py:280-293,438-460,706-769)
//   legs        leg = lane & 3, link = lane >> 2: forward dynamics of the 3-link leg chains (pybullet stepSimulation)
//   rows        lanes 0..15   constraint rows in two banks: impulse response, Delassus column, PGS state
//   dofs        18 generalised velocities u = [omega_w, v_w, joint rates]: lane l owns DOF l (lanes 0, 1 also 16, 17)
// Cross-lane data goes through DPP (a robot's 16 lanes are one DPP row) or LDS; register arrays are never indexed
// dynamically.
//
// Arithmetic: float32.  Reference citations are relative to /root/reference/OpenRoboRL/.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/openroborl_hip.h"

#define ORR_PI_F 3.14159265358979323846f
// A workgroup is exactly one wavefront.  A wave's LDS instructions execute in issue order, so a ds_write followed
// by another lane's ds_read of the same word needs no s_waitcnt and no s_barrier -- only the COMPILER must not move
// memory operations across the hand-off.  (A workgroup-scope release/acquire fence also works but makes the wave
// drain vmcnt, i.e. wait for the latency-ring store to reach memory, ten times per sub-step.)
#define WSYNC()                             \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_wave_barrier();        \
    asm volatile("" ::: "memory");          \
  } while (0)
#define SCHED_FENCE()  // measured: fencing the scheduler RAISES spills (104 vs 0 at 256 VGPRs); kept for experiments

namespace orr {

#ifndef ORR_LANES_PER_ROBOT
#define ORR_LANES_PER_ROBOT 16
#endif
constexpr int kLanes = ORR_LANES_PER_ROBOT;  // lanes of a wavefront that serve one robot (16 -> four robots per wave)
constexpr int kRPW = 64 / kLanes;            // robots per wavefront
static_assert(kLanes == 16 || kLanes == 32 || kLanes == 64, "a robot needs 16 row lanes");
constexpr int kMaxRows = 28;  // 4 knee-friction + <=12 joint-limit + 12 contact rows
constexpr int kHead = 308;    // words of the state record staged in LDS: everything before the ring (307) + one spare word (the non-finite guard's flag)

// Clip header (16 words: one per lane when it is staged into LDS).  Times are DOUBLES: the motion time reaches 20 s by the end of a
// 600-step episode, where float32 resolves 2e-6 s = 1e-4 of a frame, and the frame velocities jump by O(10) between frames
// (tests/test_gpu_clips.py).  The sampler runs twice per env step, so its few dozen f64 instructions are free.  dt_d / sim_dt_d
// are the DECIMAL constants of the clip file / yaml (0.01667, 0.001), recovered from the float32 ABI values on the host (dec7).
struct DevClip {
  const float* frames;
  const float* vels;
  int F, flags;
  double dt_d, dur_d;   // FrameDuration, FrameDuration * (F - 1): motion_data.py:198-208
  double sim_dt_d;      // cfg.sim_dt (the same for every clip; kept here because the clip header is what the sampler has in LDS)
  float cdp[3];
  float cdh;
};
static_assert(sizeof(DevClip) == 64, "clip header = 16 words");

// Compact, kernel-facing robot model; built on the host by orr_set_model from orr_model.
// All joints are required to turn about coordinate axes of the kinematic frame: the hip (k = 0) about
// +-x, upper and lower leg (k = 1, 2) about +-y.  The sign is folded into the internal joint angle
//   a = jdir * (q_urdf - joff),  jdir = JOINT_DIRECTION * axis_sign,
// so that inside the kernels every joint turns about +x or +y.
// HOT part: staged in LDS per robot (512 B) - what the 33 sub-steps and the reward's forward kinematics read again and again.
struct ModelHot {
  float jdir[12], joff[12];   // per JOINT (URDF order)
  float joint_lo[12], joint_hi[12];  // limits of the internal angle
  float joint_pos[12][3];
  float toe_pos[4][3];
  float shank_pos[4][3];      // second contact sphere of the lower leg (see orr_model)
  float lower_com[4][3];
  float init_quat[4];
  float toe_radius, shank_radius;
  // toe contact normal row (orr_model::contact_stiffness / contact_damping, folded on the host): constraint-force mixing added to the
  // row's diagonal (0 = rigid) and erp / dt of a penetrating toe contact (rigid: cfg.contact_erp / dt)
  float contact_cfm, contact_erp_dt;
};
static_assert(sizeof(ModelHot) == 512, "LDS budget: 4 robots per wave, two waves per SIMD = 20 KB per wave (DESIGN.md section 3)");
// COLD part: read straight from the device table (global memory, L2-resident) where it is needed - per-motor and per-link constants
// that go into registers once per launch, reset-only data, and the termination-only fall proxies (fetched at the top of the last
// sub-step of an env step).  Round 2 staged all of it in LDS (1420 B per robot), which capped a CU at 5-6 resident waves.
struct ModelCold {
  float init_pos[3];
  float foot_friction;
  float init_motor_angles[12], motor_dir[12], motor_offset[12];
  int joint_of_motor[12];
  float kp[12], kd[12];
  float tau_sign[12];         // per joint: internal torque = tau_sign * motor torque
  float tau_sign_motor[12];   // the same per MOTOR (tau_sign[joint_of_motor[m]]): no dependent second load
  float link_com[12][3];
  float default_joints[12];   // (INIT_MOTOR_ANGLES + OFFSET) * DIR, motor order (imitation_task.py:1245-1252)
  int num_fall;
  int fall_body[ORR_MAX_FALL_PROXIES];
  float fall_pos[ORR_MAX_FALL_PROXIES][3];
  float fall_radius[ORR_MAX_FALL_PROXIES];
  // mass properties before randomisation: body 0 = base, 1 + j = link j
  float mass[13];
  float inertia[13][6];
  float inertia_pa[13][6];
  int group[13];
  int friction_anchor;        // orr_model::friction_anchor (read by the anchor variant of the step kernel only; LAST: the fields above keep their offsets)
};
struct alignas(16) DevModel {   // 16-byte aligned in the device table: the hot part is copied to LDS in 16-byte pieces
  ModelHot hot;
  ModelCold cold;
};
static_assert(sizeof(DevModel) % 16 == 0 && offsetof(DevModel, hot) == 0, "16-byte pieces");
constexpr int kModelLdsWords = (int)(sizeof(ModelHot) / 4);

struct DevTables {
  DevModel model[ORR_MAX_ROBOT_TYPES];
  DevClip clip[ORR_MAX_CLIPS];
};

// Replay inputs of the parity entry points orr_debug_replay_reset / orr_debug_replay_step (kernel MODE 2): the scripted states,
// link positions, contact flags and random draws recorded while the reference's own Python was driven with a scripted pybullet
// client (tests/golden/make_golden_task.py).  All NULL in the product launches.
struct ReplayArgs {
  const float* traj;       // [N][action_repeat][37] rigid state after each sub-step: replaces physics_substep
  const float* eff;        // [N][2][8][3] link positions for the end-effector reward: [0] sim robot, [1] reference model
  const uint8_t* fall;     // [N] a ground contact on a non-foot link was reported
  float* tau_out;          // [N][action_repeat][12] motor torques of the actuator chain (motor order), recorded
  const float* uniforms;   // [N][28] random draws in [0, 1): replace the Philox stream at reset
};

struct KParams {
  orr_config cfg;
  float fb[3], fa[3];  // Butterworth coefficients (action_filter.py:196-217), computed on the host in double
  const DevTables* tab;
  float* state;
  long long* counters;
  float* ep_log;
  int ep_log_cap;
  int simds;           // SIMDs of the device: workgroup b belongs to dispatch round b / simds (two-waves-per-SIMD variant: priority alternation)
  int anchor_on;       // some robot type has orr_model::friction_anchor: the launches run the anchor variant of the step kernel, resets clear the anchors
#ifdef ORR_WAVE_TIMELINE
  long long* wave_times;   // development aid (tools/wave_times.py): 4 words per wave of the step kernel, either variant
#endif
};
// The cold part of a robot type's model, as a GLOBAL-address-space pointer: loads through it are global_load instructions (a generic
// pointer would make them FLAT loads, which also count on the LDS counter and serialise with every LDS access in between).
typedef const ModelCold __attribute__((address_space(1)))* ColdPtr;
__device__ __forceinline__ ColdPtr model_cold(const KParams& P, int robot_type) { return (ColdPtr)&P.tab->model[robot_type].cold; }

// ------------------------------------------------------------------------------------------------
// LDS image of one robot: ~6.6 KB, four per wave
// ------------------------------------------------------------------------------------------------
// The LDS structures are 16-byte aligned and sized in multiples of 16 bytes so that runs of consecutive words are moved
// with ds_read_b128 / ds_write_b128 (a quarter of the LDS instructions of the 4-byte-aligned layout).
struct alignas(16) LinkCache {  // per movable link, written by the lane that owns the link (leg, part), read by the row lanes
  float Rw[9];      // link -> world
  float ow[3];      // link origin, world
  float s[3], sv[3];  // motion axis of the joint in front of the link, about the base COM: (axis; (origin - base) x axis)
};
// Index 3 of the per-part arrays below (and lc[12..15], W[kMaxRows], ustar[18..21]) are DUMP slots for the lanes that own no link /
// joint / row: every lane then stores unconditionally -- a divergent `if` costs a lone wave 20-35 ticks (profiles/r02_issue_costs.txt).
struct alignas(16) LegExchange {  // per leg, hand-over between the lanes (parts) of a leg inside leg_dynamics
  float F[4][6];      // F_k = Ic_k S_k of joint k, written by part k
  float Hc[4][4];     // Hc[k][i] = S_i . F_k (valid for i <= k)
  float b[4];         // tau_k - C_k
};
struct alignas(16) LegSolve {   // per leg: column k of T written by the leg's part-k lane, Hi by all of them; read by the row lanes
  float T[3][6];    // F H^-1: column k = base wrench (angular; linear, world axes, about the base COM) per unit of joint k
  float Hi[6];      // H^-1 of the leg's 3x3 joint-space inertia (00 11 22 01 02 12)
};
static_assert(sizeof(LegSolve) == 96, "LDS budget");

// Code alignment of the hand-written loops.  A lone wave pays for instruction fetch: an 8-byte instruction (VOP3, DPP) that starts on an
// odd dword costs ~0.7 ticks more than an aligned one (shifting the whole kernel by one dword changes its run time by 2.5 %; DESIGN.md
// section 6), and the Gauss-Seidel loops are almost entirely 8-byte instructions with a few 4-byte ones in between -- with the wrong
// start parity 45 of the 57 eight-byte instructions of a sweep are misaligned, with the right one 11.  Left alone, that parity is decided
// by whatever code happens to precede the loop.  The loops therefore align themselves to 8 bytes and add one 4-byte s_nop if their bit of
// ORR_PARITY is set (bit 12: sweeps without the joint-limit bank, bit 13: with it); chosen on the GPU (tools/parity_search.py:
// bit 12 = 0.244 ms, clear = 0.249 ms; bit 13 and the parity of the compiler-generated phases make no measurable difference).
#ifndef ORR_PARITY
#define ORR_PARITY 0x1000
#endif
#define ORR_STR2(x) #x
#define ORR_STR(x) ORR_STR2(x)
constexpr int kPhaseSlots = 40;   // phase timers of the -DORR_PHASE_TIMERS build (tools/phase_cycles.py)
constexpr int kWStride = 20;   // 18 DOFs, padded to a multiple of 16 bytes
struct alignas(16) DynamicsBuf {           // leg dynamics -> row setup hand-over (dead once the impulse responses are written)
  LinkCache lc[16];           // 12 links + the part-3 lanes' dump slots
  LegExchange legx[4];
};
union alignas(16) SubstepBuf {            // live only inside a physics sub-step
  float W[kMaxRows + 1][kWStride];  // M^-1 J^T per row slot (18 used): written by the impulse responses, read until the velocity update
  DynamicsBuf dyn;            // shares its space: written by the next sub-step's dynamics, last read by the row setup
};
struct alignas(16) StepEndBuf {           // live only at reset / end of step
  union {
    struct {
      float frames[11][19];   // staged clip frames: 5 sample times x (f0, f1) + frame 0
      float fvel[2][18];
    };
    // the observation being assembled shares the space of the staged frames: it is written (step end, end of reset_robot) only after
    // the frames were blended into pose[] / vel[] (sample_poses_finish), and a reset that follows a step inside the same launch
    // rewrites all 160 values at its end
    float obs[ORR_OBS_DIM];
  };
  float pose[5][19];          // sampled reference poses (update time + 4 target times)
  float vel[18];
  float ee[2][8][3];          // end-effector world positions, [0] sim [1] ref
  float red[80];              // small cross-lane reductions of the step-end code; reset_robot: ring entries #1 / #2 and the 28 draws
};
static_assert(sizeof(StepEndBuf) <= sizeof(SubstepBuf), "LDS budget: the step-end buffers fit into the sub-step buffers' space");
union PhaseBuf {
  SubstepBuf sub;
  StepEndBuf end;
};

struct alignas(16) Shared {
  alignas(16) float s[kHead];  // state head (float / int bit patterns)
  alignas(16) ModelHot m;      // robot model (hot part)
  alignas(16) DevClip clip;    // header of the robot's motion clip (copied once per launch: its fields are read many times at the end of a step)
  LegSolve leg[4];
  alignas(16) float tdump[8];  // where the part-3 lanes (which own no joint) put their "column of T" (leg_dynamics)
  alignas(16) float Rb[9];     // kinematic base frame -> world
  alignas(16) float tau[16];   // joint torques (internal sign convention), joint order; 12..15: dump slots of the lanes that own no motor
  alignas(16) float ustar[24];
  alignas(16) float co[20];    // control (latency-delayed) observation
  alignas(16) PhaseBuf ph;
#ifdef ORR_PHASE_TIMERS
  long long pt_acc[kPhaseSlots], pt_last, pt_t0, pt_r0;  // development aid, see PT() in orr_kernels.hip
#endif
};

// ------------------------------------------------------------------------------------------------
// small math (xyzw quaternions; pose3d.py / pybullet_utils.transformations semantics)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void mv3(const float M[9], const float v[3], float o[3]) {
  float a = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  float b = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  float c = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
__device__ __forceinline__ void mtv3(const float M[9], const float v[3], float o[3]) {
  float a = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  float b = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  float c = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
__device__ __forceinline__ void mm3(const float A[9], const float B[9], float C[9]) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = t[i];
}
// C = A * B^T
__device__ __forceinline__ void mmt3(const float A[9], const float B[9], float C[9]) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = t[i];
}
// skew(r) * M
__device__ __forceinline__ void skewmul(const float r[3], const float M[9], float C[9]) {
  float t[9];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    t[0 + j] = -r[2] * M[3 + j] + r[1] * M[6 + j];
    t[3 + j] = r[2] * M[0 + j] - r[0] * M[6 + j];
    t[6 + j] = -r[1] * M[0 + j] + r[0] * M[3 + j];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = t[i];
}
// M * skew(r)
__device__ __forceinline__ void mulskew(const float M[9], const float r[3], float C[9]) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i * 3 + 0] = M[i * 3 + 1] * r[2] - M[i * 3 + 2] * r[1];
    t[i * 3 + 1] = -M[i * 3 + 0] * r[2] + M[i * 3 + 2] * r[0];
    t[i * 3 + 2] = M[i * 3 + 0] * r[1] - M[i * 3 + 1] * r[0];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = t[i];
}
__device__ __forceinline__ void sym_to_m3(const float s[6], float M[9]) {
  M[0] = s[0]; M[4] = s[1]; M[8] = s[2];
  M[1] = M[3] = s[3]; M[2] = M[6] = s[4]; M[5] = M[7] = s[5];
}
// Rodrigues rotation about unit axis: maps child-frame coordinates to parent-frame coordinates
__device__ __forceinline__ void rodrigues(const float ax[3], float ang, float R[9]) {
  float s, c;
  sincosf(ang, &s, &c);
  float t = 1.0f - c, x = ax[0], y = ax[1], z = ax[2];
  R[0] = t * x * x + c; R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
  R[3] = t * x * y + s * z; R[4] = t * y * y + c; R[5] = t * y * z - s * x;
  R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}

// ------------------------------------------------------------------------------------------------
// rotations about a coordinate axis AX (0 = x, 1 = y) by the angle with cosine c, sine s.
// R maps child-frame coordinates to parent-frame coordinates.
// ------------------------------------------------------------------------------------------------
template <int AX>
__device__ __forceinline__ void rot_fwd(float c, float s, const float v[3], float o[3]) {  // o = R v
  const float v0 = v[0], v1 = v[1], v2 = v[2];
  if (AX == 0) { o[0] = v0; o[1] = c * v1 - s * v2; o[2] = s * v1 + c * v2; }
  else { o[0] = c * v0 + s * v2; o[1] = v1; o[2] = -s * v0 + c * v2; }
}
template <int AX>
__device__ __forceinline__ void rot_inv(float c, float s, const float v[3], float o[3]) {  // o = R^T v
  const float v0 = v[0], v1 = v[1], v2 = v[2];
  if (AX == 0) { o[0] = v0; o[1] = c * v1 + s * v2; o[2] = -s * v1 + c * v2; }
  else { o[0] = c * v0 - s * v2; o[1] = v1; o[2] = s * v0 + c * v2; }
}
// O = R S R^T for symmetric S = (xx yy zz xy xz yz)
template <int AX>
__device__ __forceinline__ void rot_sym(float c, float s, const float S[6], float O[6]) {
  const float cc = c * c, ss = s * s, cs = c * s;
  const float xx = S[0], yy = S[1], zz = S[2], xy = S[3], xz = S[4], yz = S[5];
  if (AX == 0) {
    O[0] = xx;
    O[1] = cc * yy - 2.0f * cs * yz + ss * zz;
    O[2] = ss * yy + 2.0f * cs * yz + cc * zz;
    O[3] = c * xy - s * xz;
    O[4] = s * xy + c * xz;
    O[5] = cs * (yy - zz) + (cc - ss) * yz;
  } else {
    O[0] = cc * xx + 2.0f * cs * xz + ss * zz;
    O[1] = yy;
    O[2] = ss * xx - 2.0f * cs * xz + cc * zz;
    O[3] = c * xy + s * yz;
    O[4] = cs * (zz - xx) + (cc - ss) * xz;
    O[5] = -s * xy + c * yz;
  }
}
// O = R H R^T for a general 3x3 H (row-major)
template <int AX>
__device__ __forceinline__ void rot_gen(float c, float s, const float H[9], float O[9]) {
  float T[9];
#pragma unroll
  for (int j = 0; j < 3; j++) {  // T = R H : rotate every column
    const float col[3] = {H[j], H[3 + j], H[6 + j]};
    float o[3];
    rot_fwd<AX>(c, s, col, o);
    T[j] = o[0]; T[3 + j] = o[1]; T[6 + j] = o[2];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {  // O = T R^T : rotate every row
    float o[3];
    rot_fwd<AX>(c, s, &T[3 * i], o);
    O[3 * i] = o[0]; O[3 * i + 1] = o[1]; O[3 * i + 2] = o[2];
  }
}
__device__ __forceinline__ void symv(const float S[6], const float v[3], float o[3]) {
  const float a = S[0] * v[0] + S[3] * v[1] + S[4] * v[2];
  const float b = S[3] * v[0] + S[1] * v[1] + S[5] * v[2];
  const float c = S[4] * v[0] + S[5] * v[1] + S[2] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}

// sum of x over the 4 lanes of a quad (lanes 4q..4q+3), in every lane of the quad
__device__ __forceinline__ float quad_sum(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));  // quad_perm:[1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));  // quad_perm:[2,3,0,1]
  return x;
}

// sum of x over the 16 lanes of this robot (one DPP row), in every lane: quad butterflies, then the two mirror permutes
__device__ __forceinline__ float row_sum16(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));  // row_half_mirror
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));  // row_mirror
  return x;
}

// value of x in lane R (< 16) of this robot's lane group.  With 16 lanes per robot a robot is one DPP row and the
// broadcast is a single v_mov_b32_dpp row_newbcast:R (gfx90a+; every lane of the wave is active at the call sites).
template <int R>
__device__ __forceinline__ float bcast_lane(float x, int sub) {
  const int v = __float_as_int(x);
  if (kRPW == 1) return __int_as_float(__builtin_amdgcn_readlane(v, R));
  if (kRPW == 2) {
    const int a = __builtin_amdgcn_readlane(v, R), b = __builtin_amdgcn_readlane(v, R + 32);
    return __int_as_float(sub ? b : a);
  }
  // old = 0 with bound_ctrl (every source lane of a row_newbcast is valid, so neither matters): the form the compiler's DPP
  // combiner folds into the consuming VOP2 (v_mul_f32_dpp / v_fmac_f32_dpp / v_max_f32_dpp ...) instead of a separate v_mov_b32_dpp
  return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x150 + R, 0xF, 0xF, true));
}
// sum_k x_k(lane R of this robot) * w_k(own lane), k < 9: the broadcast rides as the DPP operand of the multiply-adds
// (v_fmac_f32_dpp ... row_newbcast:R; the compiler's DPP combiner only folds v_mov_b32_dpp into v_mul, not into v_fmac).
// Two accumulators halve the dependent chain.  The leading s_nop covers the "VALU write -> DPP read" hazard (2 wait states)
// for whatever the compiler placed in front of the block: it does not look inside inline assembly.
template <int R>
__device__ __forceinline__ float dpp_dot9(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7, float x8,
                                          float w0, float w1, float w2, float w3, float w4, float w5, float w6, float w7, float w8) {
  static_assert(kRPW == 4, "row_newbcast needs 16 lanes per robot");
  float a, t;
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, %2, %11 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %3, %12 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %4, %13 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %5, %14 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %6, %15 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %7, %16 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %8, %17 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %9, %18 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %10, %19 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32 %0, %0, %1"
      : "=&v"(a), "=&v"(t)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(x8),
        "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7), "v"(w8), "n"(R));
  return a;
}
// The three Delassus entries of one leg's contact rows (normal z, friction x, friction y) for the impulse response (wa, wq) held
// by this lane: t = wa_lin + wa_ang x rr + sum_k ck[k] wq[k], where rr (contact point relative to the base COM) and ck[k]
// (velocity of the contact point per unit rate of joint k of that leg) come from lane R (the leg's normal-row lane) as DPP
// operands.  18 instructions for three columns (three separate 9-term dot products: 30).
template <int R>
__device__ __forceinline__ void dpp_contact_triplet(float rr0, float rr1, float rr2, float c00, float c01, float c02, float c10, float c11,
                                                    float c12, float c20, float c21, float c22, float wa0, float wa1, float wa2, float wa3,
                                                    float wa4, float wa5, float q0, float q1, float q2, float& tx, float& ty, float& tz) {
  static_assert(kRPW == 4, "row_newbcast needs 16 lanes per robot");
#define ORR_NB " row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %0, %5, %16" ORR_NB     // tx  = rr2 * wa1
      "v_mul_f32_dpp %1, %3, %17" ORR_NB     // ty  = rr0 * wa2
      "v_mul_f32_dpp %2, %4, %15" ORR_NB     // tz  = rr1 * wa0
      "v_fmac_f32_dpp %0, -%4, %17" ORR_NB   // tx -= rr1 * wa2
      "v_fmac_f32_dpp %1, -%5, %15" ORR_NB   // ty -= rr2 * wa0
      "v_fmac_f32_dpp %2, -%3, %16" ORR_NB   // tz -= rr0 * wa1
      "v_add_f32 %0, %0, %18\n\t"
      "v_add_f32 %1, %1, %19\n\t"
      "v_add_f32 %2, %2, %20\n\t"
      "v_fmac_f32_dpp %0, %6, %22" ORR_NB
      "v_fmac_f32_dpp %1, %7, %22" ORR_NB
      "v_fmac_f32_dpp %2, %8, %22" ORR_NB
      "v_fmac_f32_dpp %0, %9, %23" ORR_NB
      "v_fmac_f32_dpp %1, %10, %23" ORR_NB
      "v_fmac_f32_dpp %2, %11, %23" ORR_NB
      "v_fmac_f32_dpp %0, %12, %24" ORR_NB
      "v_fmac_f32_dpp %1, %13, %24" ORR_NB
      "v_fmac_f32_dpp %2, %14, %24 row_newbcast:%21 row_mask:0xf bank_mask:0xf"
      : "=&v"(tx), "=&v"(ty), "=&v"(tz)
      : "v"(rr0), "v"(rr1), "v"(rr2),                                             // 3 4 5
        "v"(c00), "v"(c01), "v"(c02), "v"(c10), "v"(c11), "v"(c12),               // 6..11
        "v"(c20), "v"(c21), "v"(c22),                                             // 12 13 14
        "v"(wa0), "v"(wa1), "v"(wa2), "v"(wa3), "v"(wa4), "v"(wa5),               // 15..20
        "n"(R), "v"(q0), "v"(q1), "v"(q2));                                       // 21, 22 23 24
#undef ORR_NB
}
// max(x in lane R of this robot, 0) in one instruction (v_max_f32_dpp): the clamp of a unilateral row (contact normal, joint
// limit: bounds [0, inf)) fused with its broadcast.  x has usually just been written, hence the s_nop (DPP read hazard).
template <int R>
__device__ __forceinline__ float dpp_bcast_max0(float x, float zero) {
  static_assert(kRPW == 4, "row_newbcast needs 16 lanes per robot");
  float o;
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(x), "v"(zero), "n"(R));
  return o;
}
// same with r a loop counter of an unrolled loop (the switch folds to one case)
__device__ __forceinline__ float bcast_row(float x, int r, int sub) {
  switch (r) {
    case 0: return bcast_lane<0>(x, sub);   case 1: return bcast_lane<1>(x, sub);   case 2: return bcast_lane<2>(x, sub);
    case 3: return bcast_lane<3>(x, sub);   case 4: return bcast_lane<4>(x, sub);   case 5: return bcast_lane<5>(x, sub);
    case 6: return bcast_lane<6>(x, sub);   case 7: return bcast_lane<7>(x, sub);   case 8: return bcast_lane<8>(x, sub);
    case 9: return bcast_lane<9>(x, sub);   case 10: return bcast_lane<10>(x, sub); case 11: return bcast_lane<11>(x, sub);
    case 12: return bcast_lane<12>(x, sub); case 13: return bcast_lane<13>(x, sub); case 14: return bcast_lane<14>(x, sub);
    default: return bcast_lane<15>(x, sub);
  }
}

// sine / cosine of a joint angle (|a| is a few radians at most).  Cody-Waite reduction to [-pi/4, pi/4] with a
// two-part pi/2 and minimax polynomials (~25 instructions, error < 1e-7, no large-argument branch): +2.3 % env steps/s
// over libm's sincosf at unchanged parity tolerances.
__device__ __forceinline__ void joint_sincos(float a, float* sn, float* cs) {
  const float k = rintf(a * 0.63661977236758134f);       // a * 2/pi
  float r = fmaf(-k, 1.57079625129699707031f, a);         // pi/2 high part (exact in float)
  r = fmaf(-k, 7.54978941586159635335e-08f, r);           // pi/2 low part
  const float r2 = r * r;
  // sin(r), |r| <= pi/4
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  const float s = fmaf(ps * r2, r, r);
  // cos(r)
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  const float c = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

// Inverse trigonometric functions without branches (libm's atan2f / asinf / acosf cost a lone wave several taken or skipped
// branches each, 13-30 ticks apiece: profiles/r02_issue_costs.txt).  atan2: octant reduction to t = min / max in [0, 1], then the
// Cephes atanf kernel on [0, tan(pi/8)] (t -> (t - 1) / (t + 1) above it); absolute error < 3e-7 (checked against float64 on 2 M
// random arguments).
// NaN arguments come out FINITE (fmax / fmin and the selects drop NaNs): non-finite numbers
// are detected in ONE place, the |state| < 1e30 sweep + reward check at the end of the step (ORR_DONE_NAN, orr_kernels.hip), never through
// these functions (tests/test_gpu_parity.py::test_non_finite_state_is_caught_by_the_state_guard).
__device__ __forceinline__ float atan2_bf(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
  const bool big = t > 0.41421356237f;
  const float u = big ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
  const float z = u * u;
  float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  float r = fmaf(p * z, u, u);
  r = big ? r + 0.78539816339744831f : r;
  r = ay > ax ? 1.57079632679489662f - r : r;
  // quadrant by the SIGN BITS, like libm: atan2(0, -0.0) = pi and atan2(-0.0, -1) = -pi (a comparison with 0.0f treats -0.0 as positive)
  r = __float_as_int(x) < 0 ? 3.14159265358979324f - r : r;
  return __float_as_int(y) < 0 ? -r : r;
}
__device__ __forceinline__ float asin_bf(float x) {   // |x| <= 1
  return atan2_bf(x, __builtin_amdgcn_sqrtf(fmaxf((1.0f - x) * (1.0f + x), 0.0f)));
}
__device__ __forceinline__ float acos_bf(float x) {   // |x| <= 1
  return atan2_bf(__builtin_amdgcn_sqrtf(fmaxf((1.0f - x) * (1.0f + x), 0.0f)), x);
}

// transformations.quaternion_multiply(a, b): Hamilton product (pose3d.py:228-230)
__device__ __forceinline__ void qmul(const float a[4], const float b[4], float o[4]) {
  float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x0 = b[0], y0 = b[1], z0 = b[2], w0 = b[3];
  o[0] = x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0;
  o[1] = -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0;
  o[2] = x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0;
  o[3] = -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0;
}
__device__ __forceinline__ void qconj(const float q[4], float o[4]) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
__device__ __forceinline__ void qinv(const float q[4], float o[4]) {
  float n = __builtin_amdgcn_rcpf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  o[0] = -q[0] * n; o[1] = -q[1] * n; o[2] = -q[2] * n; o[3] = q[3] * n;
}
// pose3d.QuaternionRotatePoint (pose3d.py:213-231): q [p,0] q^-1
__device__ __forceinline__ void qrot(const float p[3], const float q[4], float o[3]) {
  float qp[4] = {p[0], p[1], p[2], 0.0f}, qi[4], t[4], r[4];
  qinv(q, qi);
  qmul(q, qp, t);
  qmul(t, qi, r);
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
__device__ __forceinline__ void qstd(float q[4]) {  // pose3d.py:289-301
  const float sg = q[3] < 0.0f ? -1.0f : 1.0f;
  q[0] *= sg; q[1] *= sg; q[2] *= sg; q[3] *= sg;
}
__device__ __forceinline__ float qheading(const float q[4]) {  // pose3d.py:325-341
  float x[3] = {1.0f, 0.0f, 0.0f}, r[3];
  qrot(x, q, r);
  return atan2_bf(r[1], r[0]);
}
__device__ __forceinline__ void q_about_z(float ang, float o[4]) {
  float s, c;
  joint_sincos(0.5f * ang, &s, &c);
  o[0] = 0.0f; o[1] = 0.0f; o[2] = s; o[3] = c;
}
// |angle| of pose3d.QuaternionToAxisAngle + normalize_rotation_angle (pose3d.py:139-187,304-322)
__device__ __forceinline__ float q_norm_angle(const float q[4]) {
  const float n = __builtin_amdgcn_sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const float ang = 2.0f * atan2_bf(n, q[3]);   // in [0, 2 pi] (n >= 0)
  // beyond pi: fmod(ang, 2 pi) (= ang here, or 0 at exactly 2 pi) minus a full turn
  const float m = ang >= 2.0f * ORR_PI_F ? ang - 2.0f * ORR_PI_F : ang;
  return ang > ORR_PI_F ? m - 2.0f * ORR_PI_F : ang;
}
// 1 / sqrt(x) for a normal positive x (sums of squares of unit-ish quaternions, pivots of the base inertia): the bare hardware
// instruction (1 ulp); libm's rsqrtf wraps it in a denormal-range rescaling, five more instructions per call
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// x[lane & 3] as a two-level select on the bits of the lane (a chain of `lane == k ? ... :` on one variable is turned into a switch by
// the optimiser, which the back end lowers to nested divergent branches: 20-35 ticks each for a lone wave)
__device__ __forceinline__ float pick4(int lane, float x0, float x1, float x2, float x3) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  const float lo = b0 ? x1 : x0, hi = b0 ? x3 : x2;
  return b1 ? hi : lo;
}
__device__ __forceinline__ float map_pi(float a) {  // pose3d.MapToMinusPiToPi (pose3d.py:358-374)
  // fmod(a, 2 pi) without libm's loop and branches (a taken or skipped branch costs a lone wave 3-6 multiply-adds): whole turns
  // k = trunc(a / 2 pi), removed with a two-part 2 pi; exact (k = 0) for |a| < 2 pi, i.e. for every joint angle
  const float k = truncf(a * 0.15915494309189535f);
  float m = fmaf(-k, 6.2831854820251465f, a);       // 2 pi rounded to float ...
  m = fmaf(-k, -1.7484555e-07f, m);                 // ... and the rest of it
  m = m >= ORR_PI_F ? m - 2.0f * ORR_PI_F : (m < -ORR_PI_F ? m + 2.0f * ORR_PI_F : m);
  return m;
}
__device__ __forceinline__ void q_to_mat(const float qin[4], float R[9]) {
  float n = rsq(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
  float x = qin[0] * n, y = qin[1] * n, z = qin[2] * n, w = qin[3] * n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
// pybullet.getEulerFromQuaternion: roll, pitch, yaw (Bullet ZYX)
__device__ __forceinline__ void euler_from_quat(const float q[4], float rpy[3]) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float sqx = x * x, sqy = y * y, sqz = z * z, sqw = w * w;
  float sarg = -2.0f * (x * z - w * y);
  rpy[0] = atan2_bf(2.0f * (y * z + w * x), -sqx - sqy + sqz + sqw);
  rpy[1] = asin_bf(fminf(fmaxf(sarg, -1.0f), 1.0f));     // = -+ pi / 2 at the clamped ends
  rpy[2] = atan2_bf(2.0f * (x * y + w * z), sqx - sqy - sqz + sqw);
}
// transformations.quaternion_slerp (shortest path)
__device__ __forceinline__ void qslerp(const float a[4], const float b[4], float f, float o[4]) {
  // the reference's "the ends coincide" threshold is 4 x the float64 epsilon on |d| - 1 (and on the angle), i.e. an angle below
  // 4e-8 rad: consecutive frames of the slow clips (inplace_steps, sidesteps: 1e-3 rad apart) ARE interpolated.  In float32 the dot
  // product cannot resolve 1 - |d| = angle^2 / 2 there, so 1 - |d| is taken from the difference quaternion (exact subtraction of
  // nearby values), and the angle and its sine come from the same number: s0 + s1 = 1 + O(angle^2) whatever its rounding
  // (round 2 used float32's epsilon and acos(|d|): up to 5e-4 off on those clips, tests/test_gpu_clips.py)
  const float EPS = 2.220446049250313e-16f * 4.0f;
  float n0 = rsq(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]);
  float n1 = rsq(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3]);
  float q0[4] = {a[0] * n0, a[1] * n0, a[2] * n0, a[3] * n0};
  float q1[4] = {b[0] * n1, b[1] * n1, b[2] * n1, b[3] * n1};
  float d = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
  // no branches: the general formula is evaluated always and the special cases of the reference are selected afterwards
  const float sgn = d < 0.0f ? -1.0f : 1.0f;
  const float e0 = q1[0] - sgn * q0[0], e1 = q1[1] - sgn * q0[1], e2 = q1[2] - sgn * q0[2], e3 = q1[3] - sgn * q0[3];
  const float omd = fminf(0.5f * (e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3), 1.0f);        // 1 - |d|
  const float sn = __builtin_amdgcn_sqrtf(fmaxf(omd * (2.0f - omd), 1e-36f));            // sin(angle), angle = acos |d| in [0, pi/2]
  const float ang = atan2_bf(sn, 1.0f - omd);
  float sa, sb, unused;
  joint_sincos((1.0f - f) * ang, &sa, &unused);
  joint_sincos(f * ang, &sb, &unused);
  const float isin = __builtin_amdgcn_rcpf(sn);
  const bool first = f == 0.0f, second = f == 1.0f;
  const bool same = omd < EPS || ang < EPS;   // the ends coincide
  const float s0 = first ? 1.0f : (second ? 0.0f : (same ? 1.0f : sa * isin));
  const float s1 = first ? 0.0f : (second ? 1.0f : (same ? 0.0f : sgn * sb * isin));
#pragma unroll
  for (int i = 0; i < 4; i++) o[i] = q0[i] * s0 + q1[i] * s1;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (same stream definition as oracle/orr_oracle.c: key = seed, ctr = (robot, episode, idx>>2, "ORRL"))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float philox_uniform(unsigned long long seed, uint32_t robot, uint32_t episode, uint32_t idx) {
  uint32_t c0 = robot, c1 = episode, c2 = idx >> 2, c3 = 0x4F52524Cu;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  uint32_t sel = idx & 3u;
  uint32_t x = sel == 0 ? c0 : (sel == 1 ? c1 : (sel == 2 ? c2 : c3));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// one whole block: draws 4 * block .. 4 * block + 3 of stream (robot, episode)
__device__ __forceinline__ void philox_block(unsigned long long seed, uint32_t robot, uint32_t episode, uint32_t block, float u[4]) {
  uint32_t c0 = robot, c1 = episode, c2 = block, c3 = 0x4F52524Cu;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  u[0] = (float)(c0 >> 8) * (1.0f / 16777216.0f); u[1] = (float)(c1 >> 8) * (1.0f / 16777216.0f);
  u[2] = (float)(c2 >> 8) * (1.0f / 16777216.0f); u[3] = (float)(c3 >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------
// state helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int geti(const Shared& S, int off) { return __float_as_int(S.s[off]); }
__device__ __forceinline__ void seti(Shared& S, int off, int v) { S.s[off] = __int_as_float(v); }

}  // namespace orr
