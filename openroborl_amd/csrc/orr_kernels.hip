// orr_kernels.hip -- HIP kernels (gfx950) + C-ABI of the vectorised quadruped imitation env.
//
// Hot path replaced: WrapperEnv.step / reset (wrapper_env.py:58-107) -> LocomotionGymEnv._step /
// reset (quadruped_gym_env.py:63-104,213-239) -> Minitaur (minitaur.py) + ImitationTask
// (imitation_task.py) + pybullet.stepSimulation.  One launch = one env step for all robots of
// this device: 33 physics sub-steps, observation, reward, termination, optional auto-reset.
// Specification of every stage: DESIGN.md section 4; CPU restatement: oracle/orr_oracle.c.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "orr_device.h"

// Development aid (tools/phase_cycles.py): -DORR_PHASE_TIMERS makes lane 0 of one wave accumulate shader-clock cycles
// per phase (Shared::pt_acc) and add them to g_phase_cycles at the end of the launch.
#ifdef ORR_PHASE_TIMERS
__device__ long long g_phase_cycles[16];
#define PT_INIT() do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 16; i_++) S.pt_acc[i_] = 0; S.pt_last = clock64(); } } while (0)
#define PT(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); S.pt_acc[k] += t_ - S.pt_last; S.pt_last = clock64(); } } while (0)
#define PT_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) for (int i_ = 0; i_ < 16; i_++) atomicAdd((unsigned long long*)&g_phase_cycles[i_], (unsigned long long)S.pt_acc[i_]); } while (0)
#else
#define PT_INIT()
#define PT(k)
#define PT_FLUSH()
#endif

using namespace orr;

#define O(name) ORR_OFF_##name

// ================================================================================================
// load / store of the per-robot record
// ================================================================================================
__device__ static void refresh_mass(const DevModel& gm, Shared& S, int lane) {
  // randomised mass properties (controllable_env_randomizer_from_config.py:193-222,309-335)
  if (lane < 13) {
    const int g = gm.group[lane];
    const float mr = S.s[O(MASS_RATIO) + g], ir = S.s[O(INERTIA_RATIO) + g];
    S.mass[lane] = gm.mass[lane] * mr;
#pragma unroll
    for (int k = 0; k < 6; k++) S.Ic[lane][k] = gm.inertia[lane][k] * ir + gm.inertia_pa[lane][k] * mr;
  }
}

__device__ static void load_robot(const KParams& P, const float* rec, Shared& S, int lane) {
  for (int i = lane; i < kHead; i += kLanes) S.s[i] = rec[i];
  WSYNC();
  const DevModel& gm = P.tab->model[geti(S, O(ROBOT_TYPE))];
  const float* mp = reinterpret_cast<const float*>(&gm.hot);
  float* dst = reinterpret_cast<float*>(&S.m);
  for (int i = lane; i < kModelLdsWords; i += kLanes) dst[i] = mp[i];
  refresh_mass(gm, S, lane);
  WSYNC();
}

__device__ static void store_robot(float* rec, const Shared& S, int lane, bool valid) {
  if (valid)
    for (int i = lane; i < O(RING); i += kLanes) rec[i] = S.s[i];
}

// ================================================================================================
// latency ring (minitaur.py:127,313-357) -- lives in global memory, lane k owns word k of an entry
// ================================================================================================
__device__ static void ctrl_obs(const KParams& P, const float* rec, Shared& S, int lane) {
  const float lat = S.s[O(LATENCY)], dt = P.cfg.sim_dt;
  const int len = geti(S, O(RING_LEN)), head = geti(S, O(RING_HEAD));
  int k0 = 0, k1 = 0;
  float al = 0.0f;
  if (!(lat <= 0.0f || len == 1)) {  // Minitaur._get_delay_obs (minitaur.py:336-357)
    int n = (int)(lat / dt);
    if (n + 1 >= len) { k0 = k1 = len - 1; }
    else { k0 = n; k1 = n + 1; al = (lat - n * dt) / dt; }
  }
  const int i0 = (head - k0 + 2 * ORR_RING_DEPTH) % ORR_RING_DEPTH, i1 = (head - k1 + 2 * ORR_RING_DEPTH) % ORR_RING_DEPTH;
  for (int i = lane; i < 19; i += kLanes) {
    float e0 = rec[O(RING) + i0 * ORR_RING_ENTRY + i], e1 = rec[O(RING) + i1 * ORR_RING_ENTRY + i];
    S.co[i] = (k0 == k1) ? e0 : (1.0f - al) * e0 + al * e1;
  }
  WSYNC();
}

// Orientation relative to the initial one (minitaur.py:325-331) and its rotation matrix (kinematic base frame -> world)
// -> Shared::Rb.  Called after every change of the base quaternion; the caller syncs.
__device__ __forceinline__ void base_rotation(Shared& S, int lane, float rel[4], float Rb[9]) {
  float qi[4];
  qinv(S.m.init_quat, qi);
  qmul(&S.s[O(QUAT)], qi, rel);
  q_to_mat(rel, Rb);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) S.Rb[i] = Rb[i];
  }
}

// Minitaur.receive_obs + get_true_obs (minitaur.py:304-334): push the true observation
__device__ static void receive_obs(float* rec, Shared& S, int lane, bool valid) {
  const int head = (geti(S, O(RING_HEAD)) + 1) % ORR_RING_DEPTH, len = geti(S, O(RING_LEN));
  float rel[4], Rb[9], rate[3];
  base_rotation(S, lane, rel, Rb);
  mtv3(Rb, &S.s[O(ANGVEL)], rate);  // get_true_base_rpy_rate (minitaur.py:640-672): angular velocity in the base frame
  for (int i = lane; i < ORR_RING_ENTRY; i += kLanes) {
    float val = 0.0f;
    if (i < 12) {
      int j = S.m.joint_of_motor[i];
      val = (S.s[O(Q) + j] - S.m.motor_offset[i]) * S.m.motor_dir[i];  // get_true_motor_angles (:543-553)
    } else if (i < 16) {
      val = i == 12 ? rel[0] : (i == 13 ? rel[1] : (i == 14 ? rel[2] : rel[3]));
    } else if (i < 19) {
      val = i == 16 ? rate[0] : (i == 17 ? rate[1] : rate[2]);
    }
    if (valid) rec[O(RING) + head * ORR_RING_ENTRY + i] = val;
  }
  WSYNC();
  if (lane == 0) {
    seti(S, O(RING_HEAD), head);
    seti(S, O(RING_LEN), len + 1 > ORR_RING_DEPTH ? ORR_RING_DEPTH : len + 1);
  }
  WSYNC();
}

// Sub-step fast path of (receive_obs; ctrl_obs): the ring entries that the control observation after the next push will
// need are already in the ring before the physics sub-step (all but the pushed one), so their loads are issued early
// (ring_prefetch) and consumed after the sub-step (ring_push_and_ctrl_obs); the global-memory latency is hidden.
struct RingFetch {
  float e0[2], e1[2];  // words lane and 16 + lane of the two entries being blended
  float al;
  bool new0, new1, same;  // entry k is the one about to be pushed
};
struct RingLatency {  // per-episode constants of Minitaur._get_delay_obs (minitaur.py:336-357)
  int n;       // whole sub-steps of latency
  float al;    // fraction towards entry n + 1
  bool none;   // latency <= 0: newest entry
};
__device__ __forceinline__ RingLatency ring_latency(const KParams& P, const Shared& S) {
  const float lat = S.s[O(LATENCY)], dt = P.cfg.sim_dt;
  RingLatency L;
  L.none = lat <= 0.0f;
  L.n = (int)(lat / dt);
  L.al = (lat - L.n * dt) / dt;
  return L;
}
struct RingCursor { int head, len; };  // RING_HEAD / RING_LEN carried in registers over the sub-steps
__device__ __forceinline__ int ring_wrap_up(int i) { return i >= ORR_RING_DEPTH ? i - ORR_RING_DEPTH : i; }   // i < 2 depth
__device__ __forceinline__ int ring_wrap_down(int i) { return i < 0 ? i + ORR_RING_DEPTH : i; }                 // i >= -depth
__device__ __forceinline__ void ring_prefetch(const RingLatency& L, const float* rec, const RingCursor& C, int lane, RingFetch& F) {
  const int head = ring_wrap_up(C.head + 1);  // after the push
  const int len = C.len + 1 > ORR_RING_DEPTH ? ORR_RING_DEPTH : C.len + 1;
  int k0 = 0, k1 = 0;
  F.al = 0.0f;
  if (!(L.none || len == 1)) {
    if (L.n + 1 >= len) { k0 = k1 = len - 1; }
    else { k0 = L.n; k1 = L.n + 1; F.al = L.al; }
  }
  F.same = k0 == k1; F.new0 = k0 == 0; F.new1 = k1 == 0;
  const int i0 = ring_wrap_down(head - k0), i1 = ring_wrap_down(head - k1);   // k < len <= depth
  const float* p0 = rec + O(RING) + i0 * ORR_RING_ENTRY;
  const float* p1 = rec + O(RING) + i1 * ORR_RING_ENTRY;
  const int hi = lane < 3 ? 16 + lane : lane;  // lanes >= 3: harmless duplicate of word `lane`
  F.e0[0] = p0[lane]; F.e0[1] = p0[hi];
  F.e1[0] = p1[lane]; F.e1[1] = p1[hi];
}
// mang: this lane's true motor angle (lane < 12), computed by the caller from its register copy of the motor constants
__device__ __forceinline__ void ring_push_and_ctrl_obs(float* rec, Shared& S, int lane, bool valid, const RingFetch& F, RingCursor& C,
                                                       float mang) {
  static_assert(ORR_RING_ENTRY == 20, "lane mapping below assumes 20-word entries");
  C.head = ring_wrap_up(C.head + 1);
  C.len = C.len + 1 > ORR_RING_DEPTH ? ORR_RING_DEPTH : C.len + 1;
  float rel[4], Rb[9], rate[3];
  base_rotation(S, lane, rel, Rb);
  mtv3(Rb, &S.s[O(ANGVEL)], rate);  // get_true_base_rpy_rate (minitaur.py:640-672): angular velocity in the base frame
  // word `lane`: motor angles 0..11 (get_true_motor_angles, :543-553), relative quaternion 12..15;
  // word 16 + lane (lanes 0..3): rate 16..18, pad 19
  const float va = lane < 12 ? mang : (lane == 12 ? rel[0] : (lane == 13 ? rel[1] : (lane == 14 ? rel[2] : rel[3])));
  const float vb = lane == 0 ? rate[0] : (lane == 1 ? rate[1] : (lane == 2 ? rate[2] : 0.0f));
  float* dst = rec + O(RING) + C.head * ORR_RING_ENTRY;
  if (valid) {
    dst[lane] = va;
    if (lane < 4) dst[16 + lane] = vb;
  }
  const float a0 = F.new0 ? va : F.e0[0], a1 = F.new1 ? va : F.e1[0];
  const float b0 = F.new0 ? vb : F.e0[1], b1 = F.new1 ? vb : F.e1[1];
  S.co[lane] = F.same ? a0 : (1.0f - F.al) * a0 + F.al * a1;
  if (lane < 3) S.co[16 + lane] = F.same ? b0 : (1.0f - F.al) * b0 + F.al * b1;
  WSYNC();
}

// ================================================================================================
// physics sub-step (pybullet stepSimulation, quadruped_gym_env.py:223)
// ================================================================================================

// Cholesky factor of a 6x6 SPD matrix (row-major full storage); L packed row-wise, (i,j) -> i(i+1)/2 + j
__device__ __forceinline__ void chol6(const float A[36], float L[21], float invdiag[6]) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      if (i == j) {
        const float rs = rsqrtf(s);
        L[i * (i + 1) / 2 + j] = s * rs;
        invdiag[i] = rs;
      } else {
        L[i * (i + 1) / 2 + j] = s * invdiag[j];
      }
    }
  }
}
__device__ __forceinline__ void chol6_solve(const float L[21], const float invdiag[6], const float b[6], float x[6]) {
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * (i + 1) / 2 + k] * y[k];
    y[i] = s * invdiag[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= L[k * (k + 1) / 2 + i] * x[k];
    x[i] = s * invdiag[i];
  }
}

// Per-lane constants, loaded once per launch and kept in registers over the 33 sub-steps (a lone wave per SIMD cannot
// hide the LDS round trips of re-reading them every sub-step).  Lane (leg = lane & 3, part = (lane >> 2) & 3) walks the
// joints 0..min(part, 2) of its leg and owns link `part` (part 3: an idle copy with zero inertia): the chain constants of
// the joints beyond its own are zeroed, so that walking "through" them is the identity.
struct LegConst {
  float r[3][3], jdir[3], joff[3];  // chain: joint origin in the parent frame, internal angle = jdir * (q - joff)
  float com[3], m, Ic[6];           // own link: COM in the link frame, mass, inertia about the COM (xx yy zz xy xz yz)
};
__device__ static void load_leg_const(const Shared& S, int lane, LegConst& K) {
  const int leg = lane & 3, part = (lane >> 2) & 3, own = 3 * leg + (part < 3 ? part : 2);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int j = 3 * leg + k;
    const bool on = k <= part;
#pragma unroll
    for (int i = 0; i < 3; i++) K.r[k][i] = on ? S.m.joint_pos[j][i] : 0.0f;
    K.jdir[k] = on ? S.m.jdir[j] : 0.0f;
    K.joff[k] = S.m.joff[j];
  }
  const bool real = part < 3;
#pragma unroll
  for (int i = 0; i < 3; i++) K.com[i] = real ? S.m.link_com[own][i] : 0.0f;
#pragma unroll
  for (int i = 0; i < 6; i++) K.Ic[i] = real ? S.Ic[own + 1][i] : 0.0f;
  K.m = real ? S.mass[own + 1] : 0.0f;
}

// ================================================================================================
// Forward dynamics of the floating base + 4 x 3-link legs.
//
// Formulation (mathematically the articulated-body result; tools/crba_proto.py checks it against the oracle's ABA):
// everything in world-aligned axes with the origin O at the base COM, spatial vectors (angular; linear),
//   M = [[ Ic_tot , F ],     F_j = Ic_j S_j (composite inertia of the subtree of joint j times its motion axis),
//        [ F^T    , H ]]     H block-diagonal: one symmetric 3x3 per leg
//   bias forces by recursive Newton-Euler with zero accelerations (C per joint, p for the base)
//   T_L = F_L H_L^-1;  A0 = Ic_tot - sum_L T_L F_L^T;  a0 = -A0^-1 (p + sum_L T_L (tau_L - C_L))
//   qdd_L = H_L^-1 (tau_L - C_L - F_L^T a0)
// A leg is a chain of three joints about coordinate axes of the link frames (hip x, upper / lower leg y), so its
// world joint axis is a column of the link rotation.  What the constraint rows need afterwards is small and goes to
// LDS: T_L (6x3), H_L^-1 per leg and A0^-1 (LegSolve / Shared::IA0inv) - the impulse response of a row is then
//   da0 = A0^-1 (Jb - T_L jl);  dqdd_L = H_L^-1 jl - T_L^T da0;  dqdd_K = -T_K^T da0  (K != L).
// Lane (leg = lane & 3, part = (lane >> 2) & 3): all lanes of a leg walk down its joints, but each computes the costly
// per-link terms (inertia about O, bias force) only for link `part`; subtree sums run over the parts with DPP row
// shifts, F / H / bias torques of the three joints are exchanged through LDS (LegExchange), and from there on every
// lane of the leg holds the whole leg again (the base system is solved redundantly in all lanes).
// ================================================================================================

// R S R^T for a symmetric S (xx yy zz xy xz yz) and a general rotation R (row-major)
__device__ __forceinline__ void rot_sym_full(const float R[9], const float S[6], float O[6]) {
  float T[9];  // T = R S
#pragma unroll
  for (int i = 0; i < 3; i++) {
    T[3 * i] = R[3 * i] * S[0] + R[3 * i + 1] * S[3] + R[3 * i + 2] * S[4];
    T[3 * i + 1] = R[3 * i] * S[3] + R[3 * i + 1] * S[1] + R[3 * i + 2] * S[5];
    T[3 * i + 2] = R[3 * i] * S[4] + R[3 * i + 1] * S[5] + R[3 * i + 2] * S[2];
  }
  O[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
  O[1] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
  O[2] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
  O[3] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
  O[4] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
  O[5] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
}
// spatial inertia about O (I symmetric, first moment h, mass m) times a spatial motion vector (w; v)
__device__ __forceinline__ void spatial_inertia_mul(const float I[6], const float h[3], float m, const float w[3], const float v[3],
                                                    float oa[3], float ol[3]) {
  float t[3], u[3];
  symv(I, w, t);
  cross3(h, v, u);
  oa[0] = t[0] + u[0]; oa[1] = t[1] + u[1]; oa[2] = t[2] + u[2];
  cross3(h, w, u);
  ol[0] = m * v[0] - u[0]; ol[1] = m * v[1] - u[1]; ol[2] = m * v[2] - u[2];
}

// one joint on the way down the leg: pose of the link behind it, its motion axis S = (s; d x s) about O, spatial
// velocity and velocity-product acceleration.  For a joint beyond the lane's own link the constants are zero and the
// step is the identity (angle 0, rate 0, offset 0).
template <int AX>
__device__ __forceinline__ void joint_down(const Shared& S, const LegConst& K, int k, int j, float Rw[9], float d[3], float Vw[3],
                                           float Vv[3], float Aa[3], float Al[3], float s[3], float sv[3], float& ad_out) {
  const float a = K.jdir[k] * (S.s[O(Q) + j] - K.joff[k]);
  const float ad = K.jdir[k] * S.s[O(QD) + j];
  ad_out = ad;
  float sn, cs;
  joint_sincos(a, &sn, &cs);
  // pose: d += Rw_parent r;  Rw = Rw_parent R(a)
  {
    float t[3];
    mv3(Rw, K.r[k], t);
    d[0] += t[0]; d[1] += t[1]; d[2] += t[2];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float p0 = Rw[3 * i], p1 = Rw[3 * i + 1], p2 = Rw[3 * i + 2];
    if (AX == 0) { Rw[3 * i + 1] = cs * p1 + sn * p2; Rw[3 * i + 2] = -sn * p1 + cs * p2; }
    else { Rw[3 * i] = cs * p0 - sn * p2; Rw[3 * i + 2] = sn * p0 + cs * p2; }
  }
  s[0] = Rw[AX]; s[1] = Rw[3 + AX]; s[2] = Rw[6 + AX];
  cross3(d, s, sv);
  // V += S ad;  A += V x (S ad)
  const float ga[3] = {s[0] * ad, s[1] * ad, s[2] * ad}, gl[3] = {sv[0] * ad, sv[1] * ad, sv[2] * ad};
#pragma unroll
  for (int i = 0; i < 3; i++) { Vw[i] += ga[i]; Vv[i] += gl[i]; }
  float t0[3], t1[3], t2[3];
  cross3(Vw, ga, t0);
  cross3(Vw, gl, t1);
  cross3(Vv, ga, t2);
#pragma unroll
  for (int i = 0; i < 3; i++) { Aa[i] += t0[i]; Al[i] += t1[i] + t2[i]; }
}

// x_q + x_{q+1} + x_{q+2} over the lanes of one leg (lane = leg + 4 q): two DPP row shifts, zero beyond the row
__device__ __forceinline__ float part_suffix_sum(float x) {
  const int v = __float_as_int(x);
  const float a = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xF, true));  // row_shl:4
  const float b = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x108, 0xF, 0xF, true));  // row_shl:8
  return x + a + b;
}

__device__ static void leg_dynamics(const KParams& P, Shared& S, const LegConst& K, int lane) {
  const int leg = lane & 3, part = (lane >> 2) & 3;
  const bool first = lane < 4;            // the leg's results are written by its part-0 lane
  float Rb[9];  // kinematic base frame -> world: kept current by base_rotation() (after every change of the quaternion)
#pragma unroll
  for (int i = 0; i < 9; i++) Rb[i] = S.Rb[i];
  const float wb[3] = {S.s[O(ANGVEL)], S.s[O(ANGVEL) + 1], S.s[O(ANGVEL) + 2]};
  const float vb[3] = {S.s[O(LINVEL)], S.s[O(LINVEL) + 1], S.s[O(LINVEL) + 2]};
  // ---- way down: this lane stops at its own link (joints beyond it are identity steps) ----
  float Rw[9], d[3] = {0, 0, 0}, Vw[3] = {wb[0], wb[1], wb[2]}, Vv[3] = {vb[0], vb[1], vb[2]};
  float Aa[3] = {0, 0, 0}, Al[3] = {0, 0, 0};
  float s0[3], sv0[3], s1[3], sv1[3], s2[3], sv2[3], ad0, ad1, ad2;
#pragma unroll
  for (int i = 0; i < 9; i++) Rw[i] = Rb[i];
  joint_down<0>(S, K, 0, 3 * leg, Rw, d, Vw, Vv, Aa, Al, s0, sv0, ad0);
  joint_down<1>(S, K, 1, 3 * leg + 1, Rw, d, Vw, Vv, Aa, Al, s1, sv1, ad1);
  joint_down<1>(S, K, 2, 3 * leg + 2, Rw, d, Vw, Vv, Aa, Al, s2, sv2, ad2);
  // own joint: axis and rate
  float so[3], svo[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    so[i] = part == 0 ? s0[i] : (part == 1 ? s1[i] : s2[i]);
    svo[i] = part == 0 ? sv0[i] : (part == 1 ? sv1[i] : sv2[i]);
  }
  const float ado = part == 0 ? ad0 : (part == 1 ? ad1 : ad2);
  if (lane < 12) {  // pose and joint axis of the own link for the constraint rows
    LinkCache& L = S.lc[3 * leg + part];
#pragma unroll
    for (int i = 0; i < 9; i++) L.Rw[i] = Rw[i];
#pragma unroll
    for (int i = 0; i < 3; i++) { L.ow[i] = S.s[O(POS) + i] + d[i]; L.s[i] = so[i]; L.sv[i] = svo[i]; }
  }
  // ---- own link: spatial inertia about O and bias force f = I A + V x* (I V) ----
  float I[6], h[3], m = K.m, f[6];
  {
    float c[3];
    mv3(Rw, K.com, c);
    c[0] += d[0]; c[1] += d[1]; c[2] += d[2];
    h[0] = m * c[0]; h[1] = m * c[1]; h[2] = m * c[2];
    rot_sym_full(Rw, K.Ic, I);
    const float hc = h[0] * c[0] + h[1] * c[1] + h[2] * c[2];
    I[0] += hc - h[0] * c[0]; I[1] += hc - h[1] * c[1]; I[2] += hc - h[2] * c[2];
    I[3] -= h[0] * c[1]; I[4] -= h[0] * c[2]; I[5] -= h[1] * c[2];
    float Pa[3], Pl[3], Fa[3], Fl[3], t0[3], t1[3], t2[3];
    spatial_inertia_mul(I, h, m, Vw, Vv, Pa, Pl);
    spatial_inertia_mul(I, h, m, Aa, Al, Fa, Fl);
    cross3(Vw, Pa, t0);
    cross3(Vv, Pl, t1);
    cross3(Vw, Pl, t2);
#pragma unroll
    for (int i = 0; i < 3; i++) { f[i] = Fa[i] + t0[i] + t1[i]; f[3 + i] = Fl[i] + t2[i]; }
  }
  // ---- way up: composite inertia and force sum of the subtree behind the own joint (sum over the leg's later parts) ----
#pragma unroll
  for (int i = 0; i < 6; i++) { I[i] = part_suffix_sum(I[i]); f[i] = part_suffix_sum(f[i]); }
#pragma unroll
  for (int i = 0; i < 3; i++) h[i] = part_suffix_sum(h[i]);
  m = part_suffix_sum(m);
  // own column of F and of the leg's joint-space inertia H (entries H[i][part], i <= part), own bias torque
  {
    float Fo[6];
    spatial_inertia_mul(I, h, m, so, svo, &Fo[0], &Fo[3]);
    const float bo = S.tau[3 * leg + (part < 3 ? part : 2)] - (dot3(so, &f[0]) + dot3(svo, &f[3]));
    const float hc0 = dot3(s0, &Fo[0]) + dot3(sv0, &Fo[3]), hc1 = dot3(s1, &Fo[0]) + dot3(sv1, &Fo[3]);
    const float hc2 = dot3(s2, &Fo[0]) + dot3(sv2, &Fo[3]);
    if (lane < 12) {
      LegExchange& X = S.legx[leg];
#pragma unroll
      for (int i = 0; i < 6; i++) X.F[part][i] = Fo[i];
      X.b[part] = bo;
      X.Hc[part][0] = hc0; X.Hc[part][1] = hc1; X.Hc[part][2] = hc2;
      if (part == 0) {  // composite of the whole leg
#pragma unroll
        for (int i = 0; i < 6; i++) { X.I[i] = I[i]; X.f[i] = f[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) X.h[i] = h[i];
        X.m = m;
      }
    }
  }
  WSYNC();
  // ---- every lane of the leg: all three F columns, H, the leg composite ----
  float F[3][6], b[3], GI[6], Gh[3], Gm, Gf[6];
  float H00, H01, H02, H11, H12, H22;
  {
    const LegExchange& X = S.legx[leg];
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int i = 0; i < 6; i++) F[k][i] = X.F[k][i];
      b[k] = X.b[k];
    }
    H00 = X.Hc[0][0]; H01 = X.Hc[1][0]; H11 = X.Hc[1][1]; H02 = X.Hc[2][0]; H12 = X.Hc[2][1]; H22 = X.Hc[2][2];
#pragma unroll
    for (int i = 0; i < 6; i++) { GI[i] = X.I[i]; Gf[i] = X.f[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) Gh[i] = X.h[i];
    Gm = X.m;
  }
  float Hi[6];  // H^-1, symmetric (00 11 22 01 02 12), by cofactors
  {
    const float c00 = H11 * H22 - H12 * H12, c01 = H02 * H12 - H01 * H22, c02 = H01 * H12 - H02 * H11;
    const float c11 = H00 * H22 - H02 * H02, c12 = H01 * H02 - H00 * H12, c22 = H00 * H11 - H01 * H01;
    const float idet = __builtin_amdgcn_rcpf(H00 * c00 + H01 * c01 + H02 * c02);
    Hi[0] = c00 * idet; Hi[1] = c11 * idet; Hi[2] = c22 * idet; Hi[3] = c01 * idet; Hi[4] = c02 * idet; Hi[5] = c12 * idet;
  }
  float T[3][6];  // T = F H^-1 (column k of T = sum_m F_m Hi[m][k])
#pragma unroll
  for (int i = 0; i < 6; i++) {
    T[0][i] = F[0][i] * Hi[0] + F[1][i] * Hi[3] + F[2][i] * Hi[4];
    T[1][i] = F[0][i] * Hi[3] + F[1][i] * Hi[1] + F[2][i] * Hi[5];
    T[2][i] = F[0][i] * Hi[4] + F[1][i] * Hi[5] + F[2][i] * Hi[2];
  }
  if (first) {
    LegSolve& Q = S.leg[leg];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int i = 0; i < 6; i++) Q.T[k][i] = T[k][i];
#pragma unroll
    for (int i = 0; i < 6; i++) Q.Hi[i] = Hi[i];
  }
  // this leg's part of the base equation: composite inertia minus T F^T, force p + T b
  float Iacc[6], Hacc[9], Macc[6], pacc[6];
#define TFT(i, j) (T[0][i] * F[0][j] + T[1][i] * F[1][j] + T[2][i] * F[2][j])
  Iacc[0] = GI[0] - TFT(0, 0); Iacc[1] = GI[1] - TFT(1, 1); Iacc[2] = GI[2] - TFT(2, 2);
  Iacc[3] = GI[3] - TFT(0, 1); Iacc[4] = GI[4] - TFT(0, 2); Iacc[5] = GI[5] - TFT(1, 2);
  Macc[0] = Gm - TFT(3, 3); Macc[1] = Gm - TFT(4, 4); Macc[2] = Gm - TFT(5, 5);
  Macc[3] = -TFT(3, 4); Macc[4] = -TFT(3, 5); Macc[5] = -TFT(4, 5);
  // top-right block: skew(h) - (T F^T)[a][3 + b]
  Hacc[0] = -TFT(0, 3);          Hacc[1] = -Gh[2] - TFT(0, 4); Hacc[2] = Gh[1] - TFT(0, 5);
  Hacc[3] = Gh[2] - TFT(1, 3);   Hacc[4] = -TFT(1, 4);         Hacc[5] = -Gh[0] - TFT(1, 5);
  Hacc[6] = -Gh[1] - TFT(2, 3);  Hacc[7] = Gh[0] - TFT(2, 4);  Hacc[8] = -TFT(2, 5);
#undef TFT
#pragma unroll
  for (int i = 0; i < 6; i++) pacc[i] = Gf[i] + T[0][i] * b[0] + T[1][i] * b[1] + T[2][i] * b[2];
  // base: sum the four leg contributions (butterfly over lane bits 0, 1 = DPP quad permutes, fused into the adds)
#pragma unroll
  for (int i = 0; i < 6; i++) { Iacc[i] = quad_sum(Iacc[i]); Macc[i] = quad_sum(Macc[i]); pacc[i] = quad_sum(pacc[i]); }
#pragma unroll
  for (int i = 0; i < 9; i++) Hacc[i] = quad_sum(Hacc[i]);
  float a0[6];
  {
    float A6[36], pA0[6], Ibw[6];
    const float m0 = S.mass[0];
    rot_sym_full(Rb, S.Ic[0], Ibw);
    float n[3], fb[3] = {m0 * vb[0], m0 * vb[1], m0 * vb[2]}, t1[3], t2[3];
    symv(Ibw, wb, n);
    cross3(wb, n, t1);
    cross3(wb, fb, t2);
    // Bullet base damping (btMultiBody): torque k_a I w, force k_l m v on the bias side
    const float kl = S.s[O(BASE_DAMPING)], ka = S.s[O(BASE_DAMPING) + 1];
#pragma unroll
    for (int i = 0; i < 3; i++) { pA0[i] = t1[i] + ka * n[i] + pacc[i]; pA0[3 + i] = t2[i] + kl * fb[i] + pacc[3 + i]; }
    float Ib[9], Im[9], Mm[9];
    sym_to_m3(Ibw, Ib);
    sym_to_m3(Iacc, Im);
    sym_to_m3(Macc, Mm);
#pragma unroll
    for (int a_ = 0; a_ < 3; a_++)
#pragma unroll
      for (int b_ = 0; b_ < 3; b_++) {
        A6[a_ * 6 + b_] = Ib[a_ * 3 + b_] + Im[a_ * 3 + b_];
        A6[a_ * 6 + 3 + b_] = Hacc[a_ * 3 + b_];
        A6[(3 + a_) * 6 + b_] = Hacc[b_ * 3 + a_];
        A6[(3 + a_) * 6 + 3 + b_] = Mm[a_ * 3 + b_] + (a_ == b_ ? m0 : 0.0f);
      }
    float Lc[21], idg[6], nb[6];
    chol6(A6, Lc, idg);
#pragma unroll
    for (int i = 0; i < 6; i++) nb[i] = -pA0[i];
    chol6_solve(Lc, idg, nb, a0);
    // explicit inverse for the impulse responses: lane c (< 6) solves for unit column c
    float e[6], x[6];
    const int col = lane % 6;
#pragma unroll
    for (int i = 0; i < 6; i++) e[i] = (i == col) ? 1.0f : 0.0f;
    chol6_solve(Lc, idg, e, x);
    if (lane < 6) {
#pragma unroll
      for (int i = 0; i < 6; i++) S.IA0inv[i * 6 + col] = x[i];
    }
  }
  // joint accelerations qdd = H^-1 (b - F^T a0); written as the unconstrained velocities u* = u + dt udot by the lane
  // that owns the joint (it has the joint rate)
  const float dt = P.cfg.sim_dt;
  {
    float g[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
      g[k] = b[k] - (F[k][0] * a0[0] + F[k][1] * a0[1] + F[k][2] * a0[2] + F[k][3] * a0[3] + F[k][4] * a0[4] + F[k][5] * a0[5]);
    const float q0 = Hi[0] * g[0] + Hi[3] * g[1] + Hi[4] * g[2], q1 = Hi[3] * g[0] + Hi[1] * g[1] + Hi[5] * g[2];
    const float q2 = Hi[4] * g[0] + Hi[5] * g[1] + Hi[2] * g[2];
    if (lane < 12) S.ustar[6 + 3 * leg + part] = ado + dt * (part == 0 ? q0 : (part == 1 ? q1 : q2));
  }
  if (lane == 0) {
    // classical base acceleration (Bullet: vdot = a_lin + w x v); gravity = uniform-field offset
    float wxv[3];
    cross3(wb, vb, wxv);
    S.ustar[0] = wb[0] + dt * a0[0]; S.ustar[1] = wb[1] + dt * a0[1]; S.ustar[2] = wb[2] + dt * a0[2];
    S.ustar[3] = vb[0] + dt * (a0[3] + wxv[0]);
    S.ustar[4] = vb[1] + dt * (a0[4] + wxv[1]);
    S.ustar[5] = vb[2] + dt * (a0[5] + wxv[2] + P.cfg.gravity_z);
  }
}

// One constraint row (state of a row lane for one of its two banks)
struct Row {
  bool active;
  int leg, nrm_slot, warm;
  float Jb[6], jl[3];          // Jacobian: base part (world angular, linear) and the 3 joints of `leg`
  float rhs, jdi, lam, w, lam_n;
  float lo_c, hi_c, mu_e;      // bounds = constant part -/+ mu_e * lambda_normal
};

__device__ __forceinline__ float row_dot(const Row& R, const float* Wr) {
  float a = R.Jb[0] * Wr[0] + R.Jb[1] * Wr[1] + R.Jb[2] * Wr[2] + R.Jb[3] * Wr[3] + R.Jb[4] * Wr[4] + R.Jb[5] * Wr[5];
  a += R.jl[0] * Wr[6 + 3 * R.leg] + R.jl[1] * Wr[6 + 3 * R.leg + 1] + R.jl[2] * Wr[6 + 3 * R.leg + 2];
  return a;
}

// Jacobian, right-hand side (not yet scaled by 1/diag) and bounds of row slot `slot`
__device__ __forceinline__ void row_setup(const Shared& S, const orr_config& cfg, int slot, bool enable, float dt, float inv_dt,
                                          float erp_dt, Row& R) {
  R.active = false; R.leg = 0; R.nrm_slot = -1; R.warm = -1;
#pragma unroll
  for (int i = 0; i < 6; i++) R.Jb[i] = 0.0f;
  R.jl[0] = R.jl[1] = R.jl[2] = 0.0f;
  R.rhs = 0.0f; R.jdi = 0.0f; R.lam = 0.0f; R.w = 0.0f; R.lam_n = 0.0f;
  float lo = 0.0f, hi = 0.0f, mu = 0.0f;
  if (slot < 4) {
    R.leg = slot;
    const float fr = S.s[O(KNEE_FRICTION) + R.leg];
    R.active = fr > 0.0f;
    R.jl[2] = 1.0f;
    lo = -fr * dt; hi = fr * dt;
    R.rhs = -S.ustar[6 + 3 * R.leg + 2];
  } else if (slot < 16) {
    const int j = slot - 4;
    R.leg = j / 3;
    const int kk = j - 3 * R.leg;
    const float a = S.m.jdir[j] * (S.s[O(Q) + j] - S.m.joff[j]);
    const float pen_lo = a - S.m.joint_lo[j], pen_hi = S.m.joint_hi[j] - a;
    const bool use_lo = pen_lo < cfg.limit_activation;
    const bool use_hi = (!use_lo) && pen_hi < cfg.limit_activation;
    R.active = use_lo || use_hi;
    const float sgn = use_lo ? 1.0f : -1.0f, pen = use_lo ? pen_lo : pen_hi;
    R.jl[0] = kk == 0 ? sgn : 0.0f; R.jl[1] = kk == 1 ? sgn : 0.0f; R.jl[2] = kk == 2 ? sgn : 0.0f;
    const float rel = sgn * S.ustar[6 + j];
    lo = 0.0f; hi = 1e30f;
    R.rhs = pen > 0.0f ? -rel - pen * inv_dt : -rel - pen * erp_dt;
  } else {
    int d;
    if (slot < 20) { R.leg = slot - 16; d = 0; }
    else { R.leg = (slot - 20) >> 1; d = 1 + ((slot - 20) & 1); }
    const int leg = R.leg;
    const LinkCache& Lb = S.lc[3 * leg + 2];
    float cw[3];
    mv3(Lb.Rw, S.m.toe_pos[leg], cw);
    cw[0] += Lb.ow[0]; cw[1] += Lb.ow[1]; cw[2] += Lb.ow[2];
    const float dist = cw[2] - S.m.toe_radius;
    R.active = dist < cfg.contact_margin;
    const float Pw[3] = {cw[0], cw[1], cw[2] - S.m.toe_radius};
    const float dir[3] = {d == 1 ? 1.0f : 0.0f, d == 2 ? 1.0f : 0.0f, d == 0 ? 1.0f : 0.0f};
    float rr[3] = {Pw[0] - S.s[O(POS)], Pw[1] - S.s[O(POS) + 1], Pw[2] - S.s[O(POS) + 2]};
    cross3(rr, dir, &R.Jb[0]);
    R.Jb[3] = dir[0]; R.Jb[4] = dir[1]; R.Jb[5] = dir[2];
    float rel = R.Jb[0] * S.ustar[0] + R.Jb[1] * S.ustar[1] + R.Jb[2] * S.ustar[2] + R.Jb[3] * S.ustar[3] + R.Jb[4] * S.ustar[4] + R.Jb[5] * S.ustar[5];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      // velocity of the contact point per unit joint rate: s x (P - o) = s x rr + (d x s), rr and d relative to the base COM
      const LinkCache& L = S.lc[3 * leg + k];
      float cr[3];
      cross3(L.s, rr, cr);
      R.jl[k] = dir[0] * (cr[0] + L.sv[0]) + dir[1] * (cr[1] + L.sv[1]) + dir[2] * (cr[2] + L.sv[2]);
      rel += R.jl[k] * S.ustar[6 + 3 * leg + k];
    }
    R.warm = 3 * leg + d;
    if (d == 0) {
      lo = 0.0f; hi = 1e30f;
      R.rhs = dist > 0.0f ? -rel - dist * inv_dt : -rel - dist * erp_dt;
    } else {
      R.nrm_slot = 16 + leg;
      mu = S.s[O(FOOT_MU)] * cfg.plane_friction;  // combined friction = product of the two coefficients
      R.rhs = -rel;
    }
  }
  if (!enable) R.active = false;
  // bounds as (constant part) + mu * lambda_normal: friction rows have a zero constant part, the others mu = 0;
  // an inactive row is pinned to zero
  R.mu_e = (R.active && R.nrm_slot >= 0) ? mu : 0.0f;
  R.hi_c = (R.active && R.nrm_slot < 0) ? hi : 0.0f;
  R.lo_c = (R.active && R.nrm_slot < 0) ? lo : 0.0f;
  if (!R.active) R.rhs = 0.0f;
}

// impulse response M^-1 J^T of the row (what btMultiBody::calcAccelerationDeltasMultiDof returns) -> W[slot]; 1/diag;
// warm start.  Block form, see leg_dynamics: da0 = A0^-1 (Jb - T_L jl); dqdd_L = H_L^-1 jl - T_L^T da0; dqdd_K = -T_K^T da0.
__device__ __forceinline__ void row_response(Shared& S, const orr_config& cfg, Row& R, int slot) {
  const int leg = R.leg;
  const LegSolve& QL = S.leg[leg];
  float fb[6], a0[6], mq[12];
#pragma unroll
  for (int i = 0; i < 6; i++) fb[i] = R.Jb[i] - (QL.T[0][i] * R.jl[0] + QL.T[1][i] * R.jl[1] + QL.T[2][i] * R.jl[2]);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float sacc = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; k++) sacc += S.IA0inv[i * 6 + k] * fb[k];
    a0[i] = sacc;
  }
  const float h0 = QL.Hi[0] * R.jl[0] + QL.Hi[3] * R.jl[1] + QL.Hi[4] * R.jl[2];
  const float h1 = QL.Hi[3] * R.jl[0] + QL.Hi[1] * R.jl[1] + QL.Hi[5] * R.jl[2];
  const float h2 = QL.Hi[4] * R.jl[0] + QL.Hi[5] * R.jl[1] + QL.Hi[2] * R.jl[2];
  float diag = 0.0f;
#pragma unroll
  for (int L4 = 0; L4 < 4; L4++) {
    const LegSolve& Q = S.leg[L4];
    const bool mine = (L4 == leg);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float t = 0.0f;
#pragma unroll
      for (int i = 0; i < 6; i++) t += Q.T[k][i] * a0[i];
      mq[3 * L4 + k] = (mine ? (k == 0 ? h0 : (k == 1 ? h1 : h2)) : 0.0f) - t;
    }
    diag += mine ? (R.jl[0] * mq[3 * L4] + R.jl[1] * mq[3 * L4 + 1] + R.jl[2] * mq[3 * L4 + 2]) : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) diag += R.Jb[i] * a0[i];
  if (R.active) {
#pragma unroll
    for (int i = 0; i < 6; i++) S.ph.sub.W[slot][i] = a0[i];
#pragma unroll
    for (int i = 0; i < 12; i++) S.ph.sub.W[slot][6 + i] = mq[i];
  }
  R.jdi = R.active ? __builtin_amdgcn_rcpf(diag) : 0.0f;
  R.rhs *= R.jdi;
  R.lam = (R.active && R.warm >= 0) ? cfg.warmstart_factor * S.s[O(LAMBDA) + R.warm] : 0.0f;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// The Gauss-Seidel sweeps over the row slots in solve order (btMultiBodyConstraintSolver::solveSingleIteration),
// Delassus form.  A row lane keeps y = lambda + (rhs - (A lambda)) / diag of its row (the unclamped Gauss-Seidel value),
// and EVERY lane keeps the impulses of all rows (lam[r], equal in all lanes of the robot).  One row update is
//   y' = y - Ac[r] lam[r];   lam[r] = broadcast_from_lane_of_r(clamp(y, lo, hi));   y = y' + Ac[r] lam[r]
// with Ac[r] = -A[row][r] / diag(row) off the diagonal and 0 on it (y of the updated row does not move): four vector
// instructions (fma, v_med3, v_mov_dpp row_newbcast, fma), three of them on the dependent chain.
// HAS_B: some robot of the wave has an active joint-limit row (bank B is swept too).
// Knee and contact rows are swept unconditionally (a row visited for a robot where it is inactive is a no-op: its
// bounds, 1/diag, lambda and Delassus column are zero); measured 9 % faster than one scalar branch per leg, which also
// stopped the scheduler from overlapping consecutive row updates.
template <bool HAS_B>
__device__ __forceinline__ void pgs_sweeps(int iters, unsigned int mask, int lane, int sub, Row& A, Row& B,
                                           const float (&AcA)[kMaxRows], const float (&AcB)[kMaxRows], float (&lam)[kMaxRows]) {
  float yA = A.lam + fmaf(-A.w, A.jdi, A.rhs), yB = B.lam + fmaf(-B.w, B.jdi, B.rhs);
  const float hiB = B.hi_c, loB = B.lo_c;
  float mun[4];  // friction rows: d(bound) / d(normal impulse of their toe)
#pragma unroll
  for (int g = 0; g < 4; g++) mun[g] = A.nrm_slot == 16 + g ? A.mu_e : 0.0f;
  float hiE = fmaf(A.mu_e, A.lam_n, A.hi_c), loE = fmaf(-A.mu_e, A.lam_n, A.lo_c);
  for (int it = 0; it < iters; it++) {
    auto rowA = [&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value, src = r < 4 ? r : r - 12;
      const float old = lam[r];
      const float yp = fmaf(-AcA[r], old, yA);
      const float sb = bcast_lane<src>(__builtin_amdgcn_fmed3f(yA, loE, hiE), sub);
      yA = fmaf(AcA[r], sb, yp);
      lam[r] = sb;
      if (HAS_B) yB = fmaf(AcB[r], sb - old, yB);
      if (r >= 16 && r < 20) {  // a normal impulse moved: friction bounds of the rows of that toe follow
        constexpr int g = r >= 16 && r < 20 ? r - 16 : 0;
        const float d = sb - old;
        hiE = fmaf(mun[g], d, hiE); loE = fmaf(-mun[g], d, loE);
      }
    };
    static_for<0, 4>(rowA);
    if (HAS_B) {
      static_for<4, 16>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        if ((mask >> r) & 1u) {
          const float old = lam[r];
          const float yp = fmaf(-AcB[r], old, yB);
          const float sb = bcast_lane<r>(__builtin_amdgcn_fmed3f(yB, loB, hiB), sub);
          yB = fmaf(AcB[r], sb, yp);
          lam[r] = sb;
          yA = fmaf(AcA[r], sb - old, yA);
        }
      });
    }
    static_for<16, 28>(rowA);
  }
}

// Delassus columns A[row][r] = J . W[r], kept in registers already scaled for the sweeps: Ac[r] = -A[row][r] / diag(row),
// 0 on the diagonal; w = (A lambda) of the warm start; lam[r] = warm-start impulse of row r (in every lane).  Knee rows
// always (an inactive one has a zero impulse response), joint-limit rows one by one, contact rows per leg.
template <bool HAS_B>
__device__ __forceinline__ void delassus_columns(const Shared& S, unsigned int mask, int lane, int sub, Row& A, Row& B,
                                                 float (&AcA)[kMaxRows], float (&AcB)[kMaxRows], float (&lam)[kMaxRows]) {
#pragma unroll
  for (int r = 0; r < kMaxRows; r++) { AcA[r] = 0.0f; AcB[r] = 0.0f; lam[r] = 0.0f; }
  auto column = [&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    constexpr bool inB = r >= 4 && r < 16;
    constexpr int src = inB ? r : (r < 4 ? r : r - 12);
    const float* Wr = S.ph.sub.W[r];
    const float l0 = bcast_lane<src>(inB ? B.lam : A.lam, sub);
    lam[r] = l0;
    const float a = row_dot(A, Wr);
    A.w += a * l0;
    AcA[r] = (!inB && lane == src) ? 0.0f : -a * A.jdi;
    if (r >= 16 && r < 20 && A.nrm_slot == r) A.lam_n = l0;
    if (HAS_B) {
      const float b = row_dot(B, Wr);
      B.w += b * l0;
      AcB[r] = (inB && lane == src) ? 0.0f : -b * B.jdi;
    }
    asm("" : "+v"(AcA[r]), "+v"(AcB[r]));  // keep the scaled value (do not re-derive it inside the sweeps); not volatile:
                                           // a volatile asm would end the scheduling region and expose every LDS read
  };
  static_for<0, 4>(column);
  if (HAS_B) {
    static_for<4, 16>([&](auto rc) __attribute__((always_inline)) {
      if ((mask >> decltype(rc)::value) & 1u) column(rc);
    });
  }
  const unsigned int cm = (mask >> 16) & 0xFu;
  static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    if ((cm >> g) & 1u) {
      column(std::integral_constant<int, 16 + g>{});
      column(std::integral_constant<int, 20 + 2 * g>{});
      column(std::integral_constant<int, 21 + 2 * g>{});
    }
  });
}

// One physics sub-step.  Returns the fall-proxy flag (wave-uniform) when want_fall.
__device__ static int physics_substep(const KParams& P, Shared& S, const LegConst& K, int lane, int sub, bool want_fall) {
  const orr_config& cfg = P.cfg;
  const float dt = cfg.sim_dt, inv_dt = 1.0f / cfg.sim_dt, erp_dt = cfg.contact_erp / cfg.sim_dt;
  leg_dynamics(P, S, K, lane);  // -> link poses, leg solves, unconstrained velocities u*
  WSYNC();
  PT(3);
  int fall = 0;
  if (want_fall) {  // termination-only collision proxies (imitation_task.py:536-546)
    bool hit = false;
    if (lane < S.m.num_fall) {
      const int b = S.m.fall_body[lane];
      const float* Rw = b == 0 ? S.Rb : S.lc[b - 1].Rw;
      const float oz = b == 0 ? S.s[O(POS) + 2] : S.lc[b - 1].ow[2];
      const float wz = Rw[6] * S.m.fall_pos[lane][0] + Rw[7] * S.m.fall_pos[lane][1] + Rw[8] * S.m.fall_pos[lane][2];
      hit = (oz + wz - S.m.fall_radius[lane]) < cfg.contact_margin;
    }
    fall = ((__ballot(hit) >> (sub * kLanes)) & ((1ull << (kLanes - 1)) * 2ull - 1ull)) != 0ull;
  }
  PT(4);

  // ---------------- constraint rows ----------------
  // 28 row slots, slot index = solve order:
  //    0..3   knee joint-friction motors (minitaur.py:1063-1070)
  //    4..15  joint limits (joint j = slot-4; at most one side can be within limit_activation)
  //   16..19  toe contact normals, 20..27 pyramid friction (leg = (slot-20)/2, t1 = +x, t2 = +y)
  // A robot has 16 row lanes holding two banks: bank A = slots 0..3 and 16..27 (lane l -> slot l < 4 ? l : l+12),
  // bank B = the joint-limit slots 4..15 (lane l -> slot l).  Bank B is skipped unless some robot of the wave has
  // a joint near its limit.
  Row A, B;
  const bool rowlane = lane < 16;
  row_setup(S, cfg, rowlane ? (lane < 4 ? lane : lane + 12) : 0, rowlane, dt, inv_dt, erp_dt, A);
  row_setup(S, cfg, (rowlane && lane >= 4) ? lane : 4, rowlane && lane >= 4, dt, inv_dt, erp_dt, B);
  const unsigned long long balA = __ballot(A.active), balB = __ballot(B.active);
  const bool anyB = balB != 0ull;  // wave-uniform
  // union over the robots of this wave of the active slots (a slot visited for a robot where it is inactive is a no-op)
  unsigned int mask = 0;
#pragma unroll
  for (int g = 0; g < kRPW; g++) {
    const unsigned int a = (unsigned int)(balA >> (g * kLanes)) & 0xFFFFu, b2 = (unsigned int)(balB >> (g * kLanes)) & 0xFFF0u;
    mask |= (a & 0xFu) | ((a >> 4) << 16) | b2;
  }
  PT(5);
  // ---------------- impulse responses M^-1 J^T, diagonal, warm start ----------------
  row_response(S, cfg, A, rowlane ? (lane < 4 ? lane : lane + 12) : 0);
  if (anyB) row_response(S, cfg, B, (rowlane && lane >= 4) ? lane : 4);
  WSYNC();
  PT(6);
  // Delassus columns, then the Gauss-Seidel sweeps; two instantiations: with and without the joint-limit bank
  float AcA[kMaxRows], AcB[kMaxRows], lam[kMaxRows];
  if (anyB) {
    delassus_columns<true>(S, mask, lane, sub, A, B, AcA, AcB, lam);
    PT(7);
    pgs_sweeps<true>(cfg.solver_iters, mask, lane, sub, A, B, AcA, AcB, lam);
  } else {
    delassus_columns<false>(S, mask, lane, sub, A, B, AcA, AcB, lam);
    PT(7);
    pgs_sweeps<false>(cfg.solver_iters, mask, lane, sub, A, B, AcA, AcB, lam);
  }
  PT(8);
  // contact impulses are remembered for the next sub-step's warm start (0 for open contacts)
  if (lane == 0) {  // warm-start slot 3 leg + d: normal (slot 16 + leg), then the two friction rows (20 + 2 leg, 21 + 2 leg)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      S.s[O(LAMBDA) + 3 * g] = lam[16 + g];
      S.s[O(LAMBDA) + 3 * g + 1] = lam[20 + 2 * g];
      S.s[O(LAMBDA) + 3 * g + 2] = lam[21 + 2 * g];
    }
  }
  // ---------------- velocity update, Bullet coordinate-velocity clamp, semi-implicit Euler ----------------
  // lane l owns DOF l (v0) and, for l < 2, DOF 16 + l (v1); the new coordinates are written by the owning lane
  float v0, v1;
  {
    float du0 = 0.0f, du1 = 0.0f;
    const int k1 = lane + 16 < 18 ? lane + 16 : 0;
    const int k0 = lane < 18 ? lane : 0;
    auto add_row = [&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      du0 += S.ph.sub.W[r][k0] * lam[r];
      if (kLanes < 18) du1 += S.ph.sub.W[r][k1] * lam[r];
    };
    static_for<0, 4>(add_row);
    if (anyB) {
      static_for<4, 16>([&](auto rc) __attribute__((always_inline)) {
        if ((mask >> decltype(rc)::value) & 1u) add_row(rc);
      });
    }
    static_for<16, 28>(add_row);
    const float vmax = cfg.max_coord_velocity;
    v0 = __builtin_amdgcn_fmed3f(S.ustar[k0] + du0, -vmax, vmax);
    v1 = __builtin_amdgcn_fmed3f(S.ustar[k1] + du1, -vmax, vmax);
  }
  if (kLanes == 16) {
    // quaternion: exponential map of the world angular velocity (DOFs 0..2, broadcast from their lanes), then normalise
    const float w0 = bcast_lane<0>(v0, sub), w1 = bcast_lane<1>(v0, sub), w2 = bcast_lane<2>(v0, sub);
    const float ww = w0 * w0 + w1 * w1 + w2 * w2, h2 = 0.25f * dt * dt * ww;  // h = |w| dt / 2
    float sc, ch;  // sin(h) / |w| and cos(h)
    if (h2 < 0.04f) {  // always, unless max_coord_velocity is raised a lot: Taylor series exact to float precision
      sc = 0.5f * dt * fmaf(h2, fmaf(h2, fmaf(h2, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
      ch = fmaf(h2, fmaf(h2, fmaf(h2, fmaf(h2, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
    } else {
      const float wn = sqrtf(ww);
      float sh;
      sincosf(0.5f * wn * dt, &sh, &ch);
      sc = sh / wn;
    }
    const float dq[4] = {w0 * sc, w1 * sc, w2 * sc, ch};
    float qn[4];
    qmul(dq, &S.s[O(QUAT)], qn);
    const float nn = rsqrtf(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    WSYNC();
    if (lane < 3) {
      S.s[O(ANGVEL) + lane] = v0;
    } else if (lane < 6) {
      S.s[O(LINVEL) + lane - 3] = v0;
      S.s[O(POS) + lane - 3] += dt * v0;
    } else {
      const int j = lane - 6;
      const float a = S.m.jdir[j] * (S.s[O(Q) + j] - S.m.joff[j]) + dt * v0;
      S.s[O(Q) + j] = a * S.m.jdir[j] + S.m.joff[j];
      S.s[O(QD) + j] = v0 * S.m.jdir[j];
    }
    if (lane < 2) {
      const int j = 10 + lane;
      const float a = S.m.jdir[j] * (S.s[O(Q) + j] - S.m.joff[j]) + dt * v1;
      S.s[O(Q) + j] = a * S.m.jdir[j] + S.m.joff[j];
      S.s[O(QD) + j] = v1 * S.m.jdir[j];
    }
    if (lane < 4) S.s[O(QUAT) + lane] = (lane == 0 ? qn[0] : (lane == 1 ? qn[1] : (lane == 2 ? qn[2] : qn[3]))) * nn;
  } else {
    // wider lane groups (tuning builds): through LDS
    if (lane < 18) S.ustar[lane] = v0;
    WSYNC();
    const float w0 = S.ustar[0], w1 = S.ustar[1], w2 = S.ustar[2];
    const float wn = sqrtf(w0 * w0 + w1 * w1 + w2 * w2), half = 0.5f * wn * dt;
    float sc, ch;
    if (wn < 1e-12f) { sc = 0.5f * dt; ch = 1.0f; }
    else { float sh; sincosf(half, &sh, &ch); sc = sh / wn; }
    float dq[4] = {w0 * sc, w1 * sc, w2 * sc, ch}, qn[4];
    qmul(dq, &S.s[O(QUAT)], qn);
    const float nn = rsqrtf(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    WSYNC();
    for (int i = lane; i < 22; i += kLanes) {
      if (i < 3) S.s[O(ANGVEL) + i] = S.ustar[i];
      else if (i < 6) { S.s[O(LINVEL) + i - 3] = S.ustar[i]; S.s[O(POS) + i - 3] += dt * S.ustar[i]; }
      else if (i < 18) {
        const int j = i - 6;
        const float a = S.m.jdir[j] * (S.s[O(Q) + j] - S.m.joff[j]) + dt * S.ustar[i];
        S.s[O(Q) + j] = a * S.m.jdir[j] + S.m.joff[j];
        S.s[O(QD) + j] = S.ustar[i] * S.m.jdir[j];
      } else {
        const int q = i - 18;
        S.s[O(QUAT) + q] = (q == 0 ? qn[0] : (q == 1 ? qn[1] : (q == 2 ? qn[2] : qn[3]))) * nn;
      }
    }
  }
  WSYNC();
  PT(9);
  return fall;
}

// ================================================================================================
// reference-motion sampling (task/motion_data.py:417-509,591-633,682-718)
// ================================================================================================
struct Sample {
  int f0, f1, count;
  float blend, phase;
};
__device__ __forceinline__ float clip_phase(const DevClip& c, float t) {  // motion_data.py:210-232
  float ph = t / c.dur;
  if (c.flags & ORR_CLIP_WRAP) ph -= floorf(ph);
  else ph = fminf(fmaxf(ph, 0.0f), 1.0f);
  return ph;
}
__device__ __forceinline__ Sample clip_index(const DevClip& c, float t) {  // motion_data.py:234-253,682-718
  Sample s;
  const bool wrap = c.flags & ORR_CLIP_WRAP;
  s.count = (int)floorf(t / c.dur);
  if (!wrap) s.count = s.count < 0 ? 0 : (s.count > 1 ? 1 : s.count);
  s.phase = clip_phase(c, t);
  if (!wrap && t <= 0.0f) { s.f0 = 0; s.f1 = 0; s.blend = 0.0f; }
  else if (!wrap && t >= c.dur) { s.f0 = c.F - 1; s.f1 = c.F - 1; s.blend = 0.0f; }
  else {
    s.f0 = (int)(s.phase * (c.F - 1));
    s.f0 = s.f0 > c.F - 1 ? c.F - 1 : s.f0;
    s.f1 = s.f0 + 1 < c.F - 1 ? s.f0 + 1 : c.F - 1;
    const float nt = s.phase * c.dur, t0 = s.f0 * c.dt, t1 = s.f1 * c.dt;
    s.blend = s.f1 == s.f0 ? 0.0f : (nt - t0) / (t1 - t0);
  }
  return s;
}
__device__ static void cycle_offset(const DevClip& c, int count, float pos[3], float rot[4]) {  // motion_data.py:591-633
  pos[0] = pos[1] = pos[2] = 0.0f;
  if (c.flags & ORR_CLIP_CYCLE_POS) {
    if (!(c.flags & ORR_CLIP_CYCLE_ROT)) {
      pos[0] = count * c.cdp[0]; pos[1] = count * c.cdp[1]; pos[2] = count * c.cdp[2];
    } else {
      for (int i = 0; i < count; i++) {
        float r[4], o[3];
        q_about_z(i * c.cdh, r);
        qrot(c.cdp, r, o);
        pos[0] += o[0]; pos[1] += o[1]; pos[2] += o[2];
      }
    }
  }
  if (!(c.flags & ORR_CLIP_CYCLE_ROT)) { rot[0] = rot[1] = rot[2] = 0.0f; rot[3] = 1.0f; }
  else q_about_z(count * c.cdh, rot);
}

// Sample the active clip at up to 5 times (lane l < nt samples time t_l): frames are staged into LDS by
// coalesced row loads (lanes 0..18 read one 19-float frame row), then lanes 0..nt-1 blend serially.
// Result: S.ph.end.pose[l] = raw (no origin offset) pose; if with_vel, S.vel = raw frame velocity at time of lane 0.
// Warm-up poses (imitation_task.py:985-1009) are substituted where `warm` and -warmup <= t < 0.
__device__ static void sample_poses(const KParams& P, Shared& S, int lane, int nt, float t_lane, bool with_vel) {
  const DevClip& c = P.tab->clip[geti(S, O(CLIP_ID))];
  const bool warm_ep = geti(S, O(WARMUP)) != 0;
  Sample sm = clip_index(c, lane < nt ? t_lane : 0.0f);
  if (lane < nt) { S.red[2 * lane] = __int_as_float(sm.f0); S.red[2 * lane + 1] = __int_as_float(sm.f1); }
  WSYNC();
  for (int e = 0; e < 2 * nt; e++) {
    const int f = __float_as_int(S.red[e]);
    for (int i = lane; i < 19; i += kLanes) S.ph.end.frames[e][i] = c.frames[f * 19 + i];
  }
  if (with_vel) {
    const int f0 = __float_as_int(S.red[0]), f1 = __float_as_int(S.red[1]);
    for (int i = lane; i < 18; i += kLanes) { S.ph.end.fvel[0][i] = c.vels[f0 * 18 + i]; S.ph.end.fvel[1][i] = c.vels[f1 * 18 + i]; }
  }
  for (int i = lane; i < 19; i += kLanes) S.ph.end.frames[10][i] = c.frames[i];  // frame 0 (warm-up heading)
  WSYNC();
  if (lane < nt) {
    const bool warm_pose = warm_ep && t_lane >= -P.cfg.warmup_time && t_lane < 0.0f;
    float out[19];
    if (warm_pose) {
      // default pose rotated to the heading of frame(0) (imitation_task.py:985-1009, 1245-1252)
      const float* fr0 = S.ph.end.frames[10];
      float dr[4], pp[3], qq[4], q0[4] = {fr0[3], fr0[4], fr0[5], fr0[6]};
      const float dh = qheading(q0) - qheading(S.m.init_quat);
      q_about_z(dh, dr);
      qrot(S.m.init_pos, dr, pp);
      qmul(dr, S.m.init_quat, qq);
      out[0] = pp[0]; out[1] = pp[1]; out[2] = pp[2];
      out[3] = qq[0]; out[4] = qq[1]; out[5] = qq[2]; out[6] = qq[3];
#pragma unroll
      for (int i = 0; i < 12; i++) out[7 + i] = S.m.default_joints[i];
    } else {
      const float* a = S.ph.end.frames[2 * lane];
      const float* b = S.ph.end.frames[2 * lane + 1];
      const float bl = sm.blend;
#pragma unroll
      for (int k = 0; k < 3; k++) out[k] = (1.0f - bl) * a[k] + bl * b[k];
      float q[4];
      qslerp(a + 3, b + 3, bl, q);
      qstd(q);
#pragma unroll
      for (int k = 7; k < 19; k++) out[k] = (1.0f - bl) * a[k] + bl * b[k];
      float cp[3], cr[4], p[3], q2[4];
      cycle_offset(c, sm.count, cp, cr);
      qrot(out, cr, p);
      out[0] = p[0] + cp[0]; out[1] = p[1] + cp[1]; out[2] = p[2] + cp[2];
      qmul(cr, q, q2);
      qstd(q2);
      out[3] = q2[0]; out[4] = q2[1]; out[5] = q2[2]; out[6] = q2[3];
    }
#pragma unroll
    for (int k = 0; k < 19; k++) S.ph.end.pose[lane][k] = out[k];
    if (with_vel && lane == 0) {
      if (warm_pose) {
#pragma unroll
        for (int k = 0; k < 18; k++) S.ph.end.vel[k] = 0.0f;
      } else {
        float v[18], cp[3], cr[4], t3[3];
#pragma unroll
        for (int k = 0; k < 18; k++) v[k] = (1.0f - sm.blend) * S.ph.end.fvel[0][k] + sm.blend * S.ph.end.fvel[1][k];
        cycle_offset(c, sm.count, cp, cr);
        qrot(&v[0], cr, t3); v[0] = t3[0]; v[1] = t3[1]; v[2] = t3[2];
        qrot(&v[3], cr, t3); v[3] = t3[0]; v[4] = t3[1]; v[5] = t3[2];
#pragma unroll
        for (int k = 0; k < 18; k++) S.ph.end.vel[k] = v[k];
      }
    }
  }
  WSYNC();
}

// apply the origin offset (imitation_task.py:938-951) to S.ph.end.pose[l] in place (lane l < nt)
__device__ static void apply_origin(Shared& S, int lane, int nt) {
  if (lane < nt) {
    float qq[4], pp[3];
    qmul(&S.s[O(ORIGIN_ROT)], &S.ph.end.pose[lane][3], qq);
    qrot(&S.ph.end.pose[lane][0], &S.s[O(ORIGIN_ROT)], pp);
    S.ph.end.pose[lane][0] = pp[0] + S.s[O(ORIGIN_POS)]; S.ph.end.pose[lane][1] = pp[1] + S.s[O(ORIGIN_POS) + 1]; S.ph.end.pose[lane][2] = pp[2] + S.s[O(ORIGIN_POS) + 2];
    S.ph.end.pose[lane][3] = qq[0]; S.ph.end.pose[lane][4] = qq[1]; S.ph.end.pose[lane][5] = qq[2]; S.ph.end.pose[lane][6] = qq[3];
  }
  WSYNC();
}

__device__ __forceinline__ float motion_time(const KParams& P, const Shared& S) {  // imitation_task.py:831-848
  float t = geti(S, O(STATE_ACTION_COUNTER)) * P.cfg.sim_dt + S.s[O(TIME_OFFSET)];
  if (geti(S, O(WARMUP))) t -= P.cfg.warmup_time;
  return t;
}

// build the 76-d target observation into obs76 (LDS) from S.ph.end.pose[1..4] (already origin-offset) -- imitation_task.py:254-301
__device__ static void target_obs(const KParams& P, const float* rec, Shared& S, int lane, float* obs76) {
  ctrl_obs(P, rec, S, lane);
  if (lane >= 1 && lane <= 4) {
    float rpy[3];
    euler_from_quat(&S.co[12], rpy);
    // robot.get_base_orientation (minitaur.py:630-638) = quaternion of the delayed rpy; its heading is the
    // direction of the rotated x axis = atan2(sin(yaw) cos(pitch), cos(yaw) cos(pitch))
    float sy, cy, cpch = cosf(rpy[1]);
    sincosf(rpy[2], &sy, &cy);
    const float heading = atan2f(sy * cpch, cy * cpch);
    float ih[4], p[3], pr[3], q[4];
    q_about_z(-heading, ih);
    const float* pose = S.ph.end.pose[lane];
    p[0] = pose[0] - S.s[O(REF_POSE)]; p[1] = pose[1] - S.s[O(REF_POSE) + 1]; p[2] = pose[2] - S.s[O(REF_POSE) + 2];
    qrot(p, ih, pr);
    qmul(ih, pose + 3, q);
    qstd(q);
    float* o = obs76 + (lane - 1) * 19;
    o[0] = pr[0]; o[1] = pr[1]; o[2] = pr[2]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
#pragma unroll
    for (int k = 7; k < 19; k++) o[k] = pose[k];
  }
  WSYNC();
}

// forward kinematics of one leg's two end-effector link COMs (lower leg, toe) -- getLinkState in
// imitation_task.py:441-446; link set minitaur.py:842-844
__device__ static void leg_end_effectors(const Shared& S, const float pos[3], const float quat[4], const float* qj, int leg,
                                         float lower[3], float toe[3]) {
  float qi[4], qrel[4], R[9], o[3] = {pos[0], pos[1], pos[2]};
  qinv(S.m.init_quat, qi);
  qmul(quat, qi, qrel);
  q_to_mat(qrel, R);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int j = 3 * leg + k;
    float t[3];
    mv3(R, S.m.joint_pos[j], t);
    o[0] += t[0]; o[1] += t[1]; o[2] += t[2];
    const float a = S.m.jdir[j] * (qj[j] - S.m.joff[j]);
    float sn, cs;
    joint_sincos(a, &sn, &cs);
#pragma unroll
    for (int i = 0; i < 3; i++) {  // R <- R Rj (joint k = 0 turns about x, k = 1, 2 about y)
      const float p0 = R[3 * i], p1 = R[3 * i + 1], p2 = R[3 * i + 2];
      if (k == 0) { R[3 * i + 1] = cs * p1 + sn * p2; R[3 * i + 2] = -sn * p1 + cs * p2; }
      else { R[3 * i] = cs * p0 - sn * p2; R[3 * i + 2] = sn * p0 + cs * p2; }
    }
  }
  float t[3];
  mv3(R, S.m.lower_com[leg], t); lower[0] = o[0] + t[0]; lower[1] = o[1] + t[1]; lower[2] = o[2] + t[2];
  mv3(R, S.m.toe_pos[leg], t); toe[0] = o[0] + t[0]; toe[1] = o[1] + t[1]; toe[2] = o[2] + t[2];
}

__device__ __forceinline__ void task_heading_rot(const Shared& S, const float q[4], float out[4]) {  // imitation_task.py:1168-1189
  float dc[4], rel[4];
  qconj(S.m.init_quat, dc);
  qmul(q, dc, rel);
  q_about_z(qheading(rel), out);
}

// ImitationTask.reward (imitation_task.py:341-516); every lane returns the same value
__device__ static float calc_reward(const KParams& P, Shared& S, int lane) {
  const float* rp = &S.s[O(REF_POSE)];
  const float* rv = &S.s[O(REF_VEL)];
  if (lane < 8) {
    const int leg = lane & 3, which = lane >> 2;  // 0 sim, 1 ref
    float lower[3], toe[3];
    if (which == 0) leg_end_effectors(S, &S.s[O(POS)], &S.s[O(QUAT)], &S.s[O(Q)], leg, lower, toe);
    else leg_end_effectors(S, rp, rp + 3, rp + 7, leg, lower, toe);
#pragma unroll
    for (int i = 0; i < 3; i++) { S.ph.end.ee[which][2 * leg][i] = lower[i]; S.ph.end.ee[which][2 * leg + 1][i] = toe[i]; }
  }
  WSYNC();
  const orr_config& c = P.cfg;
  float pose_err = 0.0f, vel_err = 0.0f, ee_err = 0.0f;
#pragma unroll
  for (int j = 0; j < 12; j++) {
    float d = rp[7 + j] - S.s[O(Q) + j];
    pose_err += d * d;
    d = rv[6 + j] - S.s[O(QD) + j];
    vel_err += d * d;
  }
  {
    float hr[4], hs[4], ihr[4], ihs[4];
    task_heading_rot(S, rp + 3, hr);
    task_heading_rot(S, &S.s[O(QUAT)], hs);
    qconj(hr, ihr);
    qconj(hs, ihs);
    // each of lanes 0..7 handles one end effector, then an 8-lane sum
    float e = 0.0f;
    if (lane < 8) {
      float a[3], b[3], ar[3], br[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { a[k] = S.ph.end.ee[1][lane][k] - rp[k]; b[k] = S.ph.end.ee[0][lane][k] - S.s[O(POS) + k]; }
      qrot(a, ihr, ar);
      qrot(b, ihs, br);
      const float dh = S.ph.end.ee[1][lane][2] - S.ph.end.ee[0][lane][2];
      e = (ar[0] - br[0]) * (ar[0] - br[0]) + (ar[1] - br[1]) * (ar[1] - br[1]) + c.reward_scale[3] * dh * dh;
    }
    S.red[lane] = e;
    WSYNC();
#pragma unroll
    for (int k = 0; k < 8; k++) ee_err += S.red[k];
  }
  float root_pose_err, root_vel_err;
  {
    float pe = 0.0f, qc[4], dq[4];
#pragma unroll
    for (int k = 0; k < 3; k++) { float d = rp[k] - S.s[O(POS) + k]; pe += d * d; }
    qconj(&S.s[O(QUAT)], qc);
    qmul(rp + 3, qc, dq);
    const float ang = q_norm_angle(dq);
    root_pose_err = pe + 0.5f * ang * ang;
    float ve = 0.0f, we = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float d = rv[k] - S.s[O(LINVEL) + k]; ve += d * d;
      d = rv[3 + k] - S.s[O(ANGVEL) + k]; we += d * d;
    }
    root_vel_err = ve + 0.1f * we;
  }
  const float r = c.reward_w[0] * expf(-c.reward_scale[0] * pose_err) + c.reward_w[1] * expf(-c.reward_scale[1] * vel_err) +
                  c.reward_w[2] * expf(-c.reward_scale[2] * ee_err) + c.reward_w[3] * expf(-c.reward_scale[4] * root_pose_err) +
                  c.reward_w[4] * expf(-c.reward_scale[5] * root_vel_err);
  WSYNC();
  return r;
}

__device__ __forceinline__ int time_limit(const orr_config& c, long long total) {  // wrapper_env.py:151-159
  if (!(c.flags & ORR_FLAG_CURRICULUM) || c.curriculum_steps <= 0) return c.ep_len_end;
  double t = (double)total / (double)c.curriculum_steps;
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  t = t * t * t;
  return (int)((1.0 - t) * c.ep_len_start + t * c.ep_len_end);
}

// current sensor readings (robot_sensors.py:74-83,153-190) from S.co -> push into the 3-deep histories
__device__ static void sensors_push(Shared& S, int lane, bool fill_all) {
  float rpy[3];
  euler_from_quat(&S.co[12], rpy);
  // 28 history columns: 0..11 motor angle k, 12..15 IMU channel, 16..27 last action; a lane owns columns lane, lane+kLanes
  constexpr int kCols = (28 + kLanes - 1) / kLanes;
  float newest[kCols], h0[kCols], h1[kCols];
  int base[kCols], w[kCols], kk[kCols];
#pragma unroll
  for (int c = 0; c < kCols; c++) {
    const int col = lane + c * kLanes;
    base[c] = 0; w[c] = 0; kk[c] = 0; newest[c] = 0.0f; h0[c] = 0.0f; h1[c] = 0.0f;
    if (col < 12) { base[c] = O(MOTORANG_HIST); w[c] = 12; kk[c] = col; newest[c] = map_pi(S.co[col]); }
    else if (col < 16) { base[c] = O(IMU_HIST); w[c] = 4; kk[c] = col - 12; newest[c] = kk[c] == 0 ? rpy[0] : (kk[c] == 1 ? rpy[1] : (kk[c] == 2 ? S.co[16] : S.co[17])); }
    else if (col < 28) { base[c] = O(LASTACT_HIST); w[c] = 12; kk[c] = col - 16; newest[c] = S.s[O(LAST_ACTION) + kk[c]]; }
    if (col < 28) { h0[c] = S.s[base[c] + kk[c]]; h1[c] = S.s[base[c] + w[c] + kk[c]]; }
  }
  WSYNC();
#pragma unroll
  for (int c = 0; c < kCols; c++) {
    if (lane + c * kLanes < 28) {
      S.s[base[c] + kk[c]] = newest[c];
      S.s[base[c] + w[c] + kk[c]] = fill_all ? newest[c] : h0[c];
      S.s[base[c] + 2 * w[c] + kk[c]] = fill_all ? newest[c] : h1[c];
    }
  }
  WSYNC();
}

// ================================================================================================
// reset of one robot (wrapper_env.py:87-107 -> quadruped_gym_env.py:63-104 -> minitaur.py:232-278 ->
// imitation_task.py:166-199); SURVEY.md Appendix A.2.  Writes the 160-d observation into obs (LDS).
// ================================================================================================
__device__ static void reset_robot(const KParams& P, float* rec, Shared& S, int lane, bool valid, long long total_step_count, float* obs) {
  const orr_config& c = P.cfg;
  // every reset starts a new episode = a new RNG stream (robot, episode)
  const uint32_t robot = (uint32_t)geti(S, O(ROBOT_INDEX)), ep = (uint32_t)geti(S, O(EPISODE_IDX)) + 1u;
  WSYNC();
  if (lane == 0) seti(S, O(EPISODE_IDX), (int)ep);
  // 1-2. default pose at the grid slot, counters, ring, filter (minitaur.py:246-268, 465-483)
  if (lane < 3) {
    S.s[O(POS) + lane] = S.m.init_pos[lane] + (lane < 2 ? S.s[O(GRID_OFFSET) + lane] : 0.0f);
    S.s[O(LINVEL) + lane] = 0.0f; S.s[O(ANGVEL) + lane] = 0.0f;
  }
  if (lane < 4) S.s[O(QUAT) + lane] = S.m.init_quat[lane];
  if (lane < 12) {
    const int j = S.m.joint_of_motor[lane];
    S.s[O(Q) + j] = S.m.init_motor_angles[lane] + S.m.motor_offset[lane];  // no direction factor (minitaur.py:481)
    S.s[O(QD) + j] = 0.0f;
    S.s[O(LAST_ACTION) + lane] = 0.0f; S.s[O(ACTION) + lane] = 0.0f; S.s[O(FILTER_ACTION) + lane] = 0.0f; S.s[O(LAMBDA) + lane] = 0.0f;
    S.s[O(XHIST) + lane] = 0.0f; S.s[O(XHIST) + 12 + lane] = 0.0f; S.s[O(YHIST) + lane] = 0.0f; S.s[O(YHIST) + 12 + lane] = 0.0f;
  }
  if (lane == 0) {
    seti(S, O(RING_LEN), 0); seti(S, O(RING_HEAD), ORR_RING_DEPTH - 1);
    seti(S, O(STATE_ACTION_COUNTER), 0); seti(S, O(STEP_COUNTER), 0); seti(S, O(FILTER_VALID), 0);
    seti(S, O(EP_STEP), 0); seti(S, O(DONE_REASON), 0);
    S.s[O(EP_RETURN)] = 0.0f;
  }
  WSYNC();
  receive_obs(rec, S, lane, valid);  // ring entry #1
  // 3. sensor histories <- 3 copies of the current readings (minitaur.py:270-271; sensor_wrappers.py:122-129)
  ctrl_obs(P, rec, S, lane);
  sensors_push(S, lane, true);
  // 4. randomiser (controllable_env_randomizer_from_config.py:92-122), sorted-name draw order:
  //    inertia 2 | joint friction 8 | latency 1 | lateral friction 1 | mass 2 | motor strength 12
  if (c.flags & ORR_FLAG_RANDOMIZER) {
    for (int i = lane; i < 26; i += kLanes) {
      const float u = philox_uniform(c.seed, robot, ep, (uint32_t)i);
      if (i < 2) S.s[O(INERTIA_RATIO) + i] = 0.5f + u * 1.0f;
      else if (i < 10) { if (((i - 2) & 1) == 0) S.s[O(KNEE_FRICTION) + ((i - 2) >> 1)] = u * 0.05f; }
      else if (i == 10) S.s[O(LATENCY)] = u * 0.04f;
      else if (i == 11) S.s[O(FOOT_MU)] = 0.5f + u * 0.75f;
      else if (i < 14) S.s[O(MASS_RATIO) + i - 12] = 0.8f + u * 0.4f;
      else S.s[O(STRENGTH) + i - 14] = 0.8f + u * 0.4f;
    }
    WSYNC();
    refresh_mass(P.tab->model[geti(S, O(ROBOT_TYPE))], S, lane);
    WSYNC();
  }
  // 5. task reset (imitation_task.py:183-199, 694-732, 1103-1110)
  const DevClip& clip = P.tab->clip[geti(S, O(CLIP_ID))];
  {
    const float u1 = philox_uniform(c.seed, robot, ep, 26u), u2 = philox_uniform(c.seed, robot, ep, 27u);
    const bool ref_init = u1 < c.ref_state_init_prob;
    const bool warm = (!ref_init) && c.warmup_time > 0.0f;
    if (lane == 0) {
      seti(S, O(WARMUP), warm ? 1 : 0);
      S.s[O(TIME_OFFSET)] = warm ? u2 * c.warmup_time : u2 * clip.dur;
      S.s[O(ORIGIN_POS)] = 0.0f; S.s[O(ORIGIN_POS) + 1] = 0.0f; S.s[O(ORIGIN_POS) + 2] = 0.0f;
      S.s[O(ORIGIN_ROT)] = 0.0f; S.s[O(ORIGIN_ROT) + 1] = 0.0f; S.s[O(ORIGIN_ROT) + 2] = 0.0f; S.s[O(ORIGIN_ROT) + 3] = 1.0f;
    }
    WSYNC();
  }
  const float t = motion_time(P, S);
  const float step_dt = c.sim_dt * c.action_repeat;
  float tl = t;
  if (lane >= 1 && lane <= 4) tl = t + c.tar_frame_steps[lane - 1] * step_dt;
  sample_poses(P, S, lane, 5, tl, true);
  if (lane == 0) {
    // origin offset: position first (with identity rotation), then rotation; position is NOT recomputed
    // afterwards (imitation_task.py:712-723)
    S.s[O(ORIGIN_POS)] = S.s[O(POS)] - S.ph.end.pose[0][0];
    S.s[O(ORIGIN_POS) + 1] = S.s[O(POS) + 1] - S.ph.end.pose[0][1];
    S.s[O(ORIGIN_POS) + 2] = 0.0f;
    const float dh = qheading(&S.s[O(QUAT)]) - qheading(&S.ph.end.pose[0][3]);
    q_about_z(dh, &S.s[O(ORIGIN_ROT)]);
    S.s[O(PREV_PHASE)] = clip_phase(clip, t);
  }
  WSYNC();
  apply_origin(S, lane, 5);
  for (int i = lane; i < 19; i += kLanes) S.s[O(REF_POSE) + i] = S.ph.end.pose[0][i];
  if (lane == 0) {
    float v[3];
    qrot(&S.ph.end.vel[0], &S.s[O(ORIGIN_ROT)], v); S.ph.end.vel[0] = v[0]; S.ph.end.vel[1] = v[1]; S.ph.end.vel[2] = v[2];
    qrot(&S.ph.end.vel[3], &S.s[O(ORIGIN_ROT)], v); S.ph.end.vel[3] = v[0]; S.ph.end.vel[4] = v[1]; S.ph.end.vel[5] = v[2];
  }
  WSYNC();
  for (int i = lane; i < 18; i += kLanes) S.s[O(REF_VEL) + i] = S.ph.end.vel[i];
  // 6. _sync_sim_model / _set_state (:778-829): teleport the sim robot onto the reference
  if (lane < 3) { S.s[O(POS) + lane] = S.ph.end.pose[0][lane]; S.s[O(LINVEL) + lane] = S.ph.end.vel[lane]; S.s[O(ANGVEL) + lane] = S.ph.end.vel[3 + lane]; }
  if (lane < 4) S.s[O(QUAT) + lane] = S.ph.end.pose[0][3 + lane];
  if (lane < 12) { S.s[O(Q) + lane] = S.ph.end.pose[0][7 + lane]; S.s[O(QD) + lane] = S.ph.end.vel[6 + lane]; }
  WSYNC();
  receive_obs(rec, S, lane, valid);  // ring entry #2 (imitation_task.py:792)
  // 7. observation = histories from step 3 + target observation (quadruped_gym_env.py:100-102; wrapper_env.py:101-105)
  if (lane == 0) seti(S, O(MAX_EP_STEPS), time_limit(c, total_step_count));
  if (lane < 12) obs[lane] = S.s[O(IMU_HIST) + lane];
  for (int i = lane; i < 36; i += kLanes) { obs[12 + i] = S.s[O(LASTACT_HIST) + i]; obs[48 + i] = S.s[O(MOTORANG_HIST) + i]; }
  target_obs(P, rec, S, lane, obs + ORR_PROPRIO_DIM);
}

// ================================================================================================
// kernels
// ================================================================================================
// lane group bookkeeping shared by the kernels: `sub` = which robot of this wave, `lane` = lane within the robot
#define ORR_PROLOGUE()                                                                   \
  __shared__ Shared Sarr[kRPW];                                                          \
  const int sub = threadIdx.x / kLanes, lane = threadIdx.x % kLanes;                     \
  Shared& S = Sarr[sub];                                                                 \
  float* obs = S.ph.end.obs;                                                             \
  const int robot_raw = blockIdx.x * kRPW + sub;                                         \
  const bool in_range = robot_raw < P.cfg.num_robots;                                    \
  const int robot = in_range ? robot_raw : 0; /* a padding lane group shadows robot 0 and never stores */ \
  float* rec = P.state + (size_t)robot * ORR_STATE_STRIDE

__global__ __launch_bounds__(64) void orr_reset_kernel(KParams P, const uint8_t* mask, float* obs_out) {
  ORR_PROLOGUE();
  const bool valid = in_range && !(mask && !mask[robot]);
  load_robot(P, rec, S, lane);
  const long long total = P.counters ? P.counters[ORR_CNT_TOTAL_STEP_COUNT] : 0;
  reset_robot(P, rec, S, lane, valid, total, obs);
  WSYNC();
  store_robot(rec, S, lane, valid);
  if (obs_out && valid)
    for (int i = lane; i < ORR_OBS_DIM; i += kLanes) obs_out[(size_t)robot * ORR_OBS_DIM + i] = obs[i];
}

// mode 0: full env step.  mode 1 (debug / parity of row C): nsub physics sub-steps with the given
// motor torques (actions = torques), no robot or task logic.
#ifndef ORR_WAVES_PER_EU
#define ORR_WAVES_PER_EU 1  // 4096 robots, four per wave = one wave on each of the 1024 SIMDs: the whole batch is resident at once
#endif
// (min, max) waves per SIMD are pinned to the same value: with a higher maximum this LLVM's iterative-ilp scheduler tries
// occupancy-improving reschedules once the kernel fits 256 VGPRs and then crashes in the register allocator.
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ORR_WAVES_PER_EU, ORR_WAVES_PER_EU))) void orr_step_kernel(KParams P, const float* actions, float* obs_out, float* reward_out,
                                                      uint8_t* done_out, int nsub) {
  ORR_PROLOGUE();
  const bool valid = in_range;
  const orr_config& c = P.cfg;
  PT_INIT();
  load_robot(P, rec, S, lane);
  // impulse-response table: stale rows are multiplied by zero impulses, so they only have to be finite
  for (int i = lane; i < kMaxRows * 18; i += kLanes) (&S.ph.sub.W[0][0])[i] = 0.0f;
  WSYNC();
  LegConst K;
  load_leg_const(S, lane, K);
  {
    float rel[4], Rb[9];
    base_rotation(S, lane, rel, Rb);  // Shared::Rb for the first sub-step; the ring push keeps it current afterwards
  }
  PT(0);

  if (MODE == 1) {
    if (lane < 12) { const int j = S.m.joint_of_motor[lane]; S.tau[j] = S.m.tau_sign[j] * actions[(size_t)robot * 12 + lane]; }
    WSYNC();
    int fall = 0;
    for (int s = 0; s < nsub; s++) {
      fall = physics_substep(P, S, K, lane, sub, true);
      float rel[4], Rb[9];
      base_rotation(S, lane, rel, Rb);
      WSYNC();
    }
    if (valid && lane == 0 && done_out) done_out[robot] = (uint8_t)fall;
    store_robot(rec, S, lane, valid);
    return;
  }

  // ---- set_act (minitaur.py:280-285): offset, last action, Butterworth filter ----
  ctrl_obs(P, rec, S, lane);
  if (lane < 12) {
    const float act = actions[(size_t)robot * 12 + lane] + S.m.init_motor_angles[lane];
    S.s[O(LAST_ACTION) + lane] = act;
    float x1 = S.s[O(XHIST) + lane], x2 = S.s[O(XHIST) + 12 + lane], y1 = S.s[O(YHIST) + lane], y2 = S.s[O(YHIST) + 12 + lane];
    if (geti(S, O(STATE_ACTION_COUNTER)) == 0) {  // _filter (minitaur.py:1169-1178): init_history(current delayed angles)
      const float d = map_pi(S.co[lane]);
      x1 = x2 = y1 = y2 = d;
    }
    const float y = act * P.fb[0] + (x1 * P.fb[1] + x2 * P.fb[2]) - (y1 * P.fa[1] + y2 * P.fa[2]);  // action_filter.py:111-120
    S.s[O(XHIST) + 12 + lane] = x1; S.s[O(XHIST) + lane] = act;
    S.s[O(YHIST) + 12 + lane] = y1; S.s[O(YHIST) + lane] = y;
    S.s[O(ACTION) + lane] = y;
  }
  WSYNC();
  PT(1);
  int fall = 0;
  const float inv_repeat = 1.0f / (float)c.action_repeat;
  const RingLatency rlat = ring_latency(P, S);
  // per-motor constants of the PD loop and the step's filtered target, in registers over the sub-steps (lane = motor)
  const int ml = lane < 12 ? lane : 0;
  const int mj = S.m.joint_of_motor[ml];
  const float m_off = S.m.motor_offset[ml], m_dir = S.m.motor_dir[ml], m_kp = S.m.kp[ml], m_kd = S.m.kd[ml];
  const float m_gain = S.m.tau_sign[mj] * S.s[O(STRENGTH) + ml];
  const float m_target = S.s[O(ACTION) + ml], m_prev = S.s[O(FILTER_ACTION) + ml];
  const bool m_has_prev = geti(S, O(FILTER_VALID)) != 0;
  int action_counter = geti(S, O(STATE_ACTION_COUNTER));
  RingCursor ring = {geti(S, O(RING_HEAD)), geti(S, O(RING_LEN))};
  for (int sstep = 0; sstep < c.action_repeat; sstep++) {
    if (kLanes != 16 && sstep > 0) ctrl_obs(P, rec, S, lane);
    if (lane < 12) {
      const float lerp = (float)(sstep + 1) * inv_repeat;  // process_action (minitaur.py:438-460)
      const float cur = map_pi(S.co[lane]);
      const float prev = m_has_prev ? m_prev : cur;
      float cmd = prev + lerp * (m_target - prev);
      cmd = fminf(fmaxf(cmd, cur - c.max_angle_change), cur + c.max_angle_change);  // _clip_motor_commands (:706-723)
      const float qm = (S.s[O(Q) + mj] - m_off) * m_dir;  // pd latency 0 (:359-363)
      const float qdm = S.s[O(QD) + mj] * m_dir;
      // MotorModel.convert_to_torque, POSITION mode (minitaur_motor.py:163-171)
      S.tau[mj] = m_gain * (-1.0f * (m_kp * (qm - cmd)) - m_kd * qdm);
    }
    WSYNC();
    action_counter++;  // robot_step bookkeeping (minitaur.py:287-293); written back after the loop
    PT(2);
    if (kLanes == 16) {  // receive_obs, then the control observation of the next sub-step / of get_obs
      RingFetch F;
      ring_prefetch(rlat, rec, ring, lane, F);
      fall = physics_substep(P, S, K, lane, sub, sstep == c.action_repeat - 1);
      ring_push_and_ctrl_obs(rec, S, lane, valid, F, ring, (S.s[O(Q) + mj] - m_off) * m_dir);
    } else {
      fall = physics_substep(P, S, K, lane, sub, sstep == c.action_repeat - 1);
      receive_obs(rec, S, lane, valid);
    }
    PT(10);
  }
  if (lane == 0) {  // end of robot_step (minitaur.py:287-293)
    if (kLanes == 16) { seti(S, O(RING_HEAD), ring.head); seti(S, O(RING_LEN), ring.len); }
    seti(S, O(STATE_ACTION_COUNTER), action_counter);
    seti(S, O(FILTER_VALID), 1);
    seti(S, O(STEP_COUNTER), geti(S, O(STEP_COUNTER)) + 1);
  }
  if (lane < 12) S.s[O(FILTER_ACTION) + lane] = m_target;
  WSYNC();
  // ---- get_obs: sensors on_step (minitaur.py:295-299) ----
  if (kLanes != 16) ctrl_obs(P, rec, S, lane);
  sensors_push(S, lane, false);
  PT(11);
  // ---- reward -> update -> done (quadruped_gym_env.py:230-233) ----
  float rew = calc_reward(P, S, lane);
  const DevClip& clip = P.tab->clip[geti(S, O(CLIP_ID))];
  const float t = motion_time(P, S);
  const float step_dt = c.sim_dt * c.action_repeat;
  float tl = t;
  if (lane >= 1 && lane <= 4) tl = t + c.tar_frame_steps[lane - 1] * step_dt;
  sample_poses(P, S, lane, 5, tl, true);
  {
    // _update_ref_motion (imitation_task.py:734-761) with _sync_ref_origin (:1020-1055)
    const float ph = clip_phase(clip, t);
    if (lane == 0) {
      if ((c.flags & ORR_FLAG_CYCLE_SYNC) && ph < S.s[O(PREV_PHASE)]) {
        float pr[3];
        qrot(&S.ph.end.pose[0][0], &S.s[O(ORIGIN_ROT)], pr);
        S.s[O(ORIGIN_POS)] = S.s[O(POS)] - pr[0];
        S.s[O(ORIGIN_POS) + 1] = S.s[O(POS) + 1] - pr[1];
        S.s[O(ORIGIN_POS) + 2] = 0.0f;
      }
      S.s[O(PREV_PHASE)] = ph;
      float v[3];
      qrot(&S.ph.end.vel[0], &S.s[O(ORIGIN_ROT)], v); S.ph.end.vel[0] = v[0]; S.ph.end.vel[1] = v[1]; S.ph.end.vel[2] = v[2];
      qrot(&S.ph.end.vel[3], &S.s[O(ORIGIN_ROT)], v); S.ph.end.vel[3] = v[0]; S.ph.end.vel[4] = v[1]; S.ph.end.vel[5] = v[2];
    }
    WSYNC();
    apply_origin(S, lane, 5);
    for (int i = lane; i < 19; i += kLanes) S.s[O(REF_POSE) + i] = S.ph.end.pose[0][i];
    for (int i = lane; i < 18; i += kLanes) S.s[O(REF_VEL) + i] = S.ph.end.vel[i];
    WSYNC();
  }
  PT(12);
  // _terminal_condition (imitation_task.py:518-572) + time limit (wrapper_env.py:79) + non-finite guard
  int reason = 0;
  {
    const float* rp = &S.s[O(REF_POSE)];
    float pe = 0.0f, qc[4], dq[4];
#pragma unroll
    for (int k = 0; k < 3; k++) { float d = rp[k] - S.s[O(POS) + k]; pe += d * d; }
    qconj(&S.s[O(QUAT)], qc);
    qmul(rp + 3, qc, dq);
    const float ang = q_norm_angle(dq);
    if (geti(S, O(STEP_COUNTER)) > 0 && fall) reason |= ORR_DONE_CONTACT_FALL;
    if (pe > c.dist_fail_threshold * c.dist_fail_threshold) reason |= ORR_DONE_ROOT_POS;
    if (fabsf(ang) > c.rot_fail_threshold) reason |= ORR_DONE_ROOT_ROT;
    bool bad = false;
    for (int i = lane; i < 37; i += kLanes) bad = bad || !(fabsf(S.s[O(POS) + i]) < 1e30f);
    if (((__ballot(bad) >> (sub * kLanes)) & ((1ull << (kLanes - 1)) * 2ull - 1ull)) != 0ull) reason |= ORR_DONE_NAN;
    if (!(fabsf(rew) < 1e30f)) { reason |= ORR_DONE_NAN; rew = 0.0f; }
    const int ep_step = geti(S, O(EP_STEP)) + 1;  // quadruped_gym_env.py:237
    if (ep_step >= geti(S, O(MAX_EP_STEPS))) reason |= ORR_DONE_TIME_LIMIT;
    WSYNC();
    if (lane == 0) {
      seti(S, O(EP_STEP), ep_step);
      seti(S, O(DONE_REASON), reason);
      S.s[O(EP_RETURN)] += rew;
    }
  }
  // observation (wrapper_env.py:109-125)
  if (lane < 12) obs[lane] = S.s[O(IMU_HIST) + lane];
  for (int i = lane; i < 36; i += kLanes) { obs[12 + i] = S.s[O(LASTACT_HIST) + i]; obs[48 + i] = S.s[O(MOTORANG_HIST) + i]; }
  target_obs(P, rec, S, lane, obs + ORR_PROPRIO_DIM);
  if (valid && lane == 0) {
    reward_out[robot] = rew;
    done_out[robot] = reason != 0;
  }
  PT(13);
  long long total_snapshot = P.counters ? P.counters[ORR_CNT_TOTAL_STEP_COUNT] : 0;
  if (reason != 0) {
    if (lane == 0) {
      S.s[O(LAST_EP_RETURN)] = S.s[O(EP_RETURN)];
      seti(S, O(LAST_EP_LEN), geti(S, O(EP_STEP)));
      if (P.counters && valid) {
        atomicAdd((unsigned long long*)&P.counters[ORR_CNT_DONE_ACCUM], 1ull);
        const unsigned long long slot = atomicAdd((unsigned long long*)&P.counters[ORR_CNT_EPISODES], 1ull);
        if (P.ep_log && slot < (unsigned long long)P.ep_log_cap) {
          P.ep_log[2 * slot] = S.s[O(EP_RETURN)];
          P.ep_log[2 * slot + 1] = (float)geti(S, O(EP_STEP));
        } else if (P.ep_log) {
          atomicAdd((unsigned long long*)&P.counters[ORR_CNT_EPLOG_DROPPED], 1ull);
        }
      }
    }
    WSYNC();
    if (c.flags & ORR_FLAG_AUTO_RESET) {
      reset_robot(P, rec, S, lane, valid, total_snapshot, obs);
    }
  }
  WSYNC();
  PT(14);
  store_robot(rec, S, lane, valid);
  if (valid)
    for (int i = lane; i < ORR_OBS_DIM; i += kLanes) obs_out[(size_t)robot * ORR_OBS_DIM + i] = obs[i];
  PT(15);
  PT_FLUSH();
  // the last wave to finish folds this launch's done count into the curriculum counter (wrapper_env.py:82-83)
  if (P.counters && valid && lane == 0) {
    const unsigned long long ticket = atomicAdd((unsigned long long*)&P.counters[ORR_CNT_TICKET], 1ull);
    if (ticket == (unsigned long long)P.cfg.num_robots - 1ull) {
      const unsigned long long nd = atomicExch((unsigned long long*)&P.counters[ORR_CNT_DONE_ACCUM], 0ull);
      atomicAdd((unsigned long long*)&P.counters[ORR_CNT_TOTAL_STEP_COUNT], nd);
      atomicAdd((unsigned long long*)&P.counters[ORR_CNT_TOTAL_TIMESTEPS], (unsigned long long)P.cfg.num_robots);
      atomicExch((unsigned long long*)&P.counters[ORR_CNT_TICKET], 0ull);
    }
  }
}

// ================================================================================================
// C-ABI (include/openroborl_hip.h)
// ================================================================================================
struct orr_handle {
  orr_config cfg;
  DevTables* tab_dev;
  DevTables tab_host;
  float fb[3], fa[3];
  float* state;
  long long* counters;
  float* ep_log;
  int ep_log_cap;
  hipEvent_t ev0, ev1;
};

static thread_local char g_err[512] = "";
// records the message returned by orr_last_error(); shared with orr_policy.hip (same library)
__attribute__((visibility("hidden"))) int orr_fail(int code, const char* msg, hipError_t e) {
  if (e != hipSuccess) snprintf(g_err, sizeof(g_err), "%s: %s", msg, hipGetErrorString(e));
  else snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int fail(int code, const char* msg, hipError_t e = hipSuccess) { return orr_fail(code, msg, e); }
#define HIPCHK(call, msg)                         \
  do {                                            \
    hipError_t e_ = (call);                       \
    if (e_ != hipSuccess) return fail(-2, msg, e_); \
  } while (0)

struct field_t { const char* name; int off, size, is_int; };
static const field_t g_fields[] = {
#define ORR_X_F(name, words, kind) {#name, ORR_OFF_##name, words, (#kind)[0] == 'I'},
    ORR_STATE_FIELDS(ORR_X_F)
#undef ORR_X_F
};

extern "C" {

const char* orr_last_error(void) { return g_err; }
int32_t orr_abi_version(void) { return ORR_ABI_VERSION; }
int32_t orr_state_stride(void) { return ORR_STATE_STRIDE; }
int32_t orr_layout_count(void) { return (int32_t)(sizeof(g_fields) / sizeof(g_fields[0])); }
const char* orr_layout_name(int32_t i) { return g_fields[i].name; }
int32_t orr_layout_offset(int32_t i) { return g_fields[i].off; }
int32_t orr_layout_size(int32_t i) { return g_fields[i].size; }
int32_t orr_layout_is_int(int32_t i) { return g_fields[i].is_int; }
int32_t orr_sizeof_config(void) { return (int32_t)sizeof(orr_config); }
int32_t orr_sizeof_model(void) { return (int32_t)sizeof(orr_model); }

int32_t orr_create(const orr_config* cfg, orr_handle** out) {
  if (!cfg || !out) return fail(-1, "orr_create: null argument");
  if (cfg->abi_version != ORR_ABI_VERSION) return fail(-1, "orr_create: ABI version mismatch");
  if (cfg->num_robots < 1) return fail(-1, "orr_create: num_robots must be >= 1");
  if (cfg->action_repeat < 1 || cfg->solver_iters < 1) return fail(-1, "orr_create: action_repeat / solver_iters must be >= 1");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1) return fail(-3, "orr_create: no HIP device available (this library has no CPU fallback)", e);
  orr_handle* h = new orr_handle();
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg;
  // ActionFilterButter.butter_filter (action_filter.py:196-217): scipy.signal.butter(2, [4 / (fs / 2)], 'low')
  {
    const double fs = 1.0 / ((double)cfg->sim_dt * cfg->action_repeat), wn = 4.0 / (0.5 * fs);
    const double K = tan(M_PI * wn / 2.0), K2 = K * K, den = 1.0 + sqrt(2.0) * K + K2;
    h->fb[0] = (float)(K2 / den); h->fb[1] = (float)(2.0 * K2 / den); h->fb[2] = (float)(K2 / den);
    h->fa[0] = 1.0f; h->fa[1] = (float)(2.0 * (K2 - 1.0) / den); h->fa[2] = (float)((1.0 - sqrt(2.0) * K + K2) / den);
  }
  e = hipMalloc((void**)&h->tab_dev, sizeof(DevTables));
  if (e != hipSuccess) { delete h; return fail(-2, "orr_create: hipMalloc", e); }
  e = hipMemset(h->tab_dev, 0, sizeof(DevTables));
  if (e != hipSuccess) { hipFree(h->tab_dev); delete h; return fail(-2, "orr_create: hipMemset", e); }
  hipEventCreate(&h->ev0);
  hipEventCreate(&h->ev1);
  *out = h;
  return 0;
}

int32_t orr_destroy(orr_handle* h) {
  if (!h) return 0;
  hipFree(h->tab_dev);
  hipEventDestroy(h->ev0);
  hipEventDestroy(h->ev1);
  delete h;
  return 0;
}

int32_t orr_set_model(orr_handle* h, int32_t robot_type, const orr_model* m) {
  if (!h || !m) return fail(-1, "orr_set_model: null argument");
  if (robot_type < 0 || robot_type >= ORR_MAX_ROBOT_TYPES) return fail(-1, "orr_set_model: robot_type out of range");
  if (m->num_fall_proxies < 0 || m->num_fall_proxies > ORR_MAX_FALL_PROXIES) return fail(-1, "orr_set_model: bad num_fall_proxies");
  for (int i = 0; i < 12; i++) {
    if (m->joint_of_motor[i] < 0 || m->joint_of_motor[i] > 11) return fail(-1, "orr_set_model: joint_of_motor out of range");
    if (m->link_group[i] < 0 || m->link_group[i] > 1) return fail(-1, "orr_set_model: link_group must be 0 or 1");
  }
  for (int i = 0; i < m->num_fall_proxies; i++)
    if (m->fall_body[i] < 0 || m->fall_body[i] > 12) return fail(-1, "orr_set_model: fall_body out of range");
  // every joint must turn about a coordinate axis of the kinematic frame: hip about +-x, upper / lower leg about +-y
  DevModel& d = h->tab_host.model[robot_type];
  memset(&d, 0, sizeof(d));
  float axsgn[12];
  for (int j = 0; j < 12; j++) {
    const int ax = (j % 3 == 0) ? 0 : 1;
    const float* a = m->joint_axis[j];
    for (int k = 0; k < 3; k++)
      if (k != ax && fabsf(a[k]) > 1e-6f) return fail(-1, "orr_set_model: joint axes must be +-x (hip) / +-y (upper, lower leg)");
    if (fabsf(fabsf(a[ax]) - 1.0f) > 1e-5f) return fail(-1, "orr_set_model: joint axis is not a unit coordinate axis");
    axsgn[j] = a[ax] > 0 ? 1.0f : -1.0f;
  }
  ModelHot& H = d.hot;
  for (int i = 0; i < 3; i++) H.init_pos[i] = m->init_pos[i];
  for (int i = 0; i < 4; i++) H.init_quat[i] = m->init_quat[i];
  for (int i = 0; i < 12; i++) {
    const int j = m->joint_of_motor[i];
    H.init_motor_angles[i] = m->init_motor_angles[i];
    H.motor_dir[i] = m->motor_dir[i];
    H.motor_offset[i] = m->motor_offset[i];
    H.joint_of_motor[i] = j;
    H.motor_of_joint[j] = i;
    H.kp[i] = m->kp[i];
    H.kd[i] = m->kd[i];
    H.jdir[j] = m->motor_dir[i] * axsgn[j];
    H.joff[j] = m->motor_offset[i];
    H.tau_sign[j] = axsgn[j];
    H.default_joints[i] = (m->init_motor_angles[i] + m->motor_offset[i]) * m->motor_dir[i];
  }
  for (int j = 0; j < 12; j++) {
    for (int k = 0; k < 3; k++) { H.link_com[j][k] = m->link_com[j][k]; H.joint_pos[j][k] = m->joint_pos[j][k]; }
    // limits are given for the kinematic angle; the internal angle is axis_sign times it
    H.joint_lo[j] = axsgn[j] > 0 ? m->joint_lo[j] : -m->joint_hi[j];
    H.joint_hi[j] = axsgn[j] > 0 ? m->joint_hi[j] : -m->joint_lo[j];
    if (!(H.joint_hi[j] - H.joint_lo[j] >= 2.0f * h->cfg.limit_activation))
      return fail(-1, "orr_set_model: joint range must be at least 2 * limit_activation");
  }
  for (int l = 0; l < 4; l++)
    for (int k = 0; k < 3; k++) { H.toe_pos[l][k] = m->toe_pos[l][k]; H.lower_com[l][k] = m->lower_com[l][k]; }
  H.toe_radius = m->toe_radius;
  H.foot_friction = m->foot_friction;
  H.num_fall = m->num_fall_proxies;
  for (int i = 0; i < ORR_MAX_FALL_PROXIES; i++) {
    H.fall_body[i] = m->fall_body[i];
    H.fall_radius[i] = m->fall_radius[i];
    for (int k = 0; k < 3; k++) H.fall_pos[i][k] = m->fall_pos[i][k];
  }
  d.mass[0] = m->base_mass;
  d.group[0] = 0;
  for (int k = 0; k < 6; k++) { d.inertia[0][k] = m->base_inertia[k]; d.inertia_pa[0][k] = 0.0f; }
  for (int j = 0; j < 12; j++) {
    d.mass[j + 1] = m->link_mass[j];
    d.group[j + 1] = m->link_group[j];
    for (int k = 0; k < 6; k++) { d.inertia[j + 1][k] = m->link_inertia[j][k]; d.inertia_pa[j + 1][k] = m->link_inertia_pa[j][k]; }
  }
  HIPCHK(hipMemcpy(&h->tab_dev->model[robot_type], &d, sizeof(DevModel), hipMemcpyHostToDevice), "orr_set_model: hipMemcpy");
  return 0;
}

int32_t orr_set_motion(orr_handle* h, int32_t clip_id, const float* frames_dev, const float* frame_vels_dev, int32_t num_frames,
                       float frame_dt, int32_t clip_flags, const float cycle_delta[4]) {
  if (!h || !frames_dev || !frame_vels_dev || !cycle_delta) return fail(-1, "orr_set_motion: null argument");
  if (clip_id < 0 || clip_id >= ORR_MAX_CLIPS) return fail(-1, "orr_set_motion: clip_id out of range");
  if (num_frames < 2) return fail(-1, "orr_set_motion: need at least 2 frames");
  if (!(frame_dt > 0.0f)) return fail(-1, "orr_set_motion: Frame duration must be positive.");
  DevClip c;
  c.frames = frames_dev; c.vels = frame_vels_dev; c.F = num_frames; c.flags = clip_flags;
  c.dt = frame_dt; c.dur = frame_dt * (num_frames - 1);
  c.cdp[0] = cycle_delta[0]; c.cdp[1] = cycle_delta[1]; c.cdp[2] = cycle_delta[2]; c.cdh = cycle_delta[3];
  h->tab_host.clip[clip_id] = c;
  HIPCHK(hipMemcpy(&h->tab_dev->clip[clip_id], &c, sizeof(DevClip), hipMemcpyHostToDevice), "orr_set_motion: hipMemcpy");
  return 0;
}

int32_t orr_bind(orr_handle* h, void* state_dev, int64_t* counters_dev, float* ep_log_dev, int32_t ep_log_capacity) {
  if (!h || !state_dev) return fail(-1, "orr_bind: null argument");
  h->state = (float*)state_dev;
  h->counters = (long long*)counters_dev;
  h->ep_log = ep_log_dev;
  h->ep_log_cap = ep_log_dev ? ep_log_capacity : 0;
  return 0;
}

static KParams make_params(const orr_handle* h) {
  KParams P;
  P.cfg = h->cfg;
  for (int i = 0; i < 3; i++) { P.fb[i] = h->fb[i]; P.fa[i] = h->fa[i]; }
  P.tab = h->tab_dev;
  P.state = h->state;
  P.counters = h->counters;
  P.ep_log = h->ep_log;
  P.ep_log_cap = h->ep_log_cap;
  return P;
}

int32_t orr_reset(orr_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!h || !h->state) return fail(-1, "orr_reset: handle not bound");
  hipLaunchKernelGGL(orr_reset_kernel, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h), mask_dev, obs_dev);
  HIPCHK(hipGetLastError(), "orr_reset: launch");
  return 0;
}

int32_t orr_step(orr_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream) {
  if (!h || !h->state) return fail(-1, "orr_step: handle not bound");
  if (!actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(-1, "orr_step: null buffer");
  hipLaunchKernelGGL(orr_step_kernel<0>, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h), actions_dev,
                     obs_dev, reward_dev, done_dev, 0);
  HIPCHK(hipGetLastError(), "orr_step: launch");
  return 0;
}

// parity / debug entry point (not part of the drop-in surface): nsub physics sub-steps with fixed motor torques
int32_t orr_debug_physics(orr_handle* h, const float* torques_dev, uint8_t* fall_dev, int32_t nsub, void* stream) {
  if (!h || !h->state || !torques_dev) return fail(-1, "orr_debug_physics: bad argument");
  hipLaunchKernelGGL(orr_step_kernel<1>, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h), torques_dev,
                     nullptr, nullptr, fall_dev, nsub);
  HIPCHK(hipGetLastError(), "orr_debug_physics: launch");
  return 0;
}

int32_t orr_time_steps(orr_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream,
                       int32_t num_steps, float* total_ms_out) {
  if (!h || !h->state || !total_ms_out) return fail(-1, "orr_time_steps: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipEventRecord(h->ev0, st), "orr_time_steps: event");
  for (int i = 0; i < num_steps; i++) {
    int rc = orr_step(h, actions_dev, obs_dev, reward_dev, done_dev, stream);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(h->ev1, st), "orr_time_steps: event");
  HIPCHK(hipEventSynchronize(h->ev1), "orr_time_steps: sync");
  HIPCHK(hipEventElapsedTime(total_ms_out, h->ev0, h->ev1), "orr_time_steps: elapsed");
  return 0;
}

#ifdef ORR_PHASE_TIMERS
// development aid: read (and optionally clear) the per-phase cycle totals of the instrumented wave
int orr_debug_phase_cycles(long long* out16, int reset) {
  HIPCHK(hipDeviceSynchronize(), "orr_debug_phase_cycles: sync");
  HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(long long)), "orr_debug_phase_cycles: read");
  if (reset) {
    long long z[16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)), "orr_debug_phase_cycles: clear");
  }
  return 0;
}
#endif

}  // extern "C"
