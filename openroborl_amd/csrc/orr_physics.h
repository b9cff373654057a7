// orr_physics.h -- one physics sub-step: leg dynamics, constraint rows, Gauss-Seidel sweeps, integration (pybullet stepSimulation)
// (device code of libopenroborl_hip.so, included by orr_kernels.hip after orr_device.h; see DESIGN.md sections 3-5)
#pragma once

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
        const float rs = rsq(s);
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

// The same factorisation on PACKED float pairs (v_pk_fma_f32 / v_pk_mul_f32: two multiply-adds per issued instruction; round 4, v38).
// The factor is kept by COLUMNS with the rows below the diagonal paired (2,3) and (4,5): a rank-1 update of the trailing columns and a
// forward substitution step then work on whole pairs with the scalar factor as an op_sel broadcast.  Every entry sees the operations of
// chol6 / chol6_solve in the same order (the sums run over k ascending either way; the backward substitution stays scalar: by columns it
// would subtract in descending order), each step ONE fused multiply-add.  Packed: the factorisation
// and the solves, and also the column of T, the elimination terms -T_k F_k^T / T_k b_k and the 16-lane sums in that layout (leg_dynamics).
// 4096 robots 0.2187 -> 0.2164 ms, 8192 robots 0.3033 -> 0.3005 ms (interleaved A/B, profiles/r04_ab17_*.log).
typedef float pk2 __attribute__((ext_vector_type(2)));
// c - a * s and c - a * b as ONE fused operation per component (v_pk_fma_f32; the scalar rides as an op_sel broadcast).  Spelled out, not
// left to the compiler's contraction of `c - a * s`: in the row solves it kept some of those as v_pk_mul + v_pk_add, i.e. other bits
// than the scalar code's v_fma.
__device__ __forceinline__ pk2 pk_nfma(pk2 a, float s, pk2 c) { return __builtin_elementwise_fma(-a, pk2{s, s}, c); }
struct Chol6Pk {
  float l10, l32, l54;            // the sub-diagonal entries outside the pairs
  pk2 c0a, c0b;                   // column 0: rows (2,3), (4,5)
  pk2 c1a, c1b;                   // column 1: rows (2,3), (4,5)
  pk2 c2b, c3b;                   // columns 2, 3: rows (4,5)
  float idg[6];                   // 1 / diagonal
};
// A by columns of its lower triangle: d0, d1, d3, d5 diagonal entries, a10 = A[1][0], d2a = (A[2][2], A[3][2]), d4a = (A[4][4], A[5][4]),
// aKa = (A[2][K], A[3][K]), aKb = (A[4][K], A[5][K])
__device__ __forceinline__ void chol6_pk(float d0, float d1, float d3, float d5, float a10, pk2 d2a, pk2 d4a,
                                         pk2 a0a, pk2 a0b, pk2 a1a, pk2 a1b, pk2 a2b, pk2 a3b, Chol6Pk& F) {
  // column 0
  const float r0 = rsq(d0);
  const float l10 = a10 * r0;
  const pk2 c0a = a0a * r0, c0b = a0b * r0;
  d1 = fmaf(-l10, l10, d1);
  a1a = pk_nfma(c0a, l10, a1a); a1b = pk_nfma(c0b, l10, a1b);
  d2a = pk_nfma(c0a, c0a.x, d2a);                            // (2,2), (3,2)
  d3 = fmaf(-c0a.y, c0a.y, d3);
  a2b = pk_nfma(c0b, c0a.x, a2b); a3b = pk_nfma(c0b, c0a.y, a3b);
  d4a = pk_nfma(c0b, c0b.x, d4a);                            // (4,4), (5,4)
  d5 = fmaf(-c0b.y, c0b.y, d5);
  // column 1
  const float r1 = rsq(d1);
  const pk2 c1a = a1a * r1, c1b = a1b * r1;
  d2a = pk_nfma(c1a, c1a.x, d2a);
  d3 = fmaf(-c1a.y, c1a.y, d3);
  a2b = pk_nfma(c1b, c1a.x, a2b); a3b = pk_nfma(c1b, c1a.y, a3b);
  d4a = pk_nfma(c1b, c1b.x, d4a);
  d5 = fmaf(-c1b.y, c1b.y, d5);
  // column 2
  const float r2 = rsq(d2a.x);
  const float l32 = d2a.y * r2;
  const pk2 c2b = a2b * r2;
  d3 = fmaf(-l32, l32, d3);
  a3b = pk_nfma(c2b, l32, a3b);
  d4a = pk_nfma(c2b, c2b.x, d4a);
  d5 = fmaf(-c2b.y, c2b.y, d5);
  // column 3
  const float r3 = rsq(d3);
  const pk2 c3b = a3b * r3;
  d4a = pk_nfma(c3b, c3b.x, d4a);
  d5 = fmaf(-c3b.y, c3b.y, d5);
  // columns 4, 5
  const float r4 = rsq(d4a.x);
  const float l54 = d4a.y * r4;
  d5 = fmaf(-l54, l54, d5);
  F.idg[0] = r0; F.idg[1] = r1; F.idg[2] = r2; F.idg[3] = r3; F.idg[4] = r4; F.idg[5] = rsq(d5);
  F.l10 = l10; F.l32 = l32; F.l54 = l54;
  F.c0a = c0a; F.c0b = c0b; F.c1a = c1a; F.c1b = c1b; F.c2b = c2b; F.c3b = c3b;
}
// right-hand side (b0, b1, (b2,b3), (b4,b5))
__device__ __forceinline__ void chol6_solve_pk(const Chol6Pk& F, float b0, float b1, pk2 b23, pk2 b45, float x[6]) {
  float y[6];
  y[0] = b0 * F.idg[0];
  b1 = fmaf(-F.l10, y[0], b1); b23 = pk_nfma(F.c0a, y[0], b23); b45 = pk_nfma(F.c0b, y[0], b45);
  y[1] = b1 * F.idg[1];
  b23 = pk_nfma(F.c1a, y[1], b23); b45 = pk_nfma(F.c1b, y[1], b45);
  y[2] = b23.x * F.idg[2];
  b23.y = fmaf(-F.l32, y[2], b23.y); b45 = pk_nfma(F.c2b, y[2], b45);
  y[3] = b23.y * F.idg[3];
  b45 = pk_nfma(F.c3b, y[3], b45);
  y[4] = b45.x * F.idg[4];
  b45.y = fmaf(-F.l54, y[4], b45.y);
  y[5] = b45.y * F.idg[5];
  // backward substitution: x[i] = (y[i] - sum_{k > i} L[k][i] x[k]) / L[i][i], k ascending (the order of chol6_solve)
  x[5] = y[5] * F.idg[5];
  x[4] = fmaf(-F.l54, x[5], y[4]) * F.idg[4];
  x[3] = fmaf(-F.c3b.y, x[5], fmaf(-F.c3b.x, x[4], y[3])) * F.idg[3];
  x[2] = fmaf(-F.c2b.y, x[5], fmaf(-F.c2b.x, x[4], fmaf(-F.l32, x[3], y[2]))) * F.idg[2];
  x[1] = fmaf(-F.c1b.y, x[5], fmaf(-F.c1b.x, x[4], fmaf(-F.c1a.y, x[3], fmaf(-F.c1a.x, x[2], y[1])))) * F.idg[1];
  x[0] = fmaf(-F.c0b.y, x[5], fmaf(-F.c0b.x, x[4], fmaf(-F.c0a.y, x[3], fmaf(-F.c0a.x, x[2], fmaf(-F.l10, x[1], y[0]))))) * F.idg[0];
}

// Per-lane constants, loaded once per launch and kept in registers over the 33 sub-steps (a lone wave per SIMD cannot
// hide the LDS round trips of re-reading them every sub-step).  Lane (leg = lane & 3, part = (lane >> 2) & 3) walks the
// joints 0..min(part, 2) of its leg and owns link `part` (part 3: an idle copy with zero inertia): the chain constants of
// the joints beyond its own are zeroed, so that walking "through" them is the identity.
// Two choices differ between the translation units (orr_kernels.hip: why there are several); they are properties of the unit, not -D knobs:
//   kCarrySubtreeMass  the subtree mass lives in a register across the launch in the one-wave build only: the 256-register build pays more
//                      for the live register than for the two adds that rebuild it every sub-step
//   kOwnLegFactor      row_response's variant (b), see there: four more live registers cost the two-wave unit 30 spilled ones
#ifdef ORR_TU_STEP_W2
constexpr bool kCarrySubtreeMass = false, kOwnLegFactor = false;
#else
constexpr bool kCarrySubtreeMass = true, kOwnLegFactor = true;
#endif
struct LegConst {
  float r[3][3], jdir[3], joff[3];  // chain: joint origin in the parent frame, internal angle = jdir * (q - joff)
  float com[3], m, Ic[6];           // own link: COM in the link frame, mass, inertia about the COM (xx yy zz xy xz yz)
  float msub;                       // mass of the subtree behind the own joint (own + later links of the leg): constant over a launch
  float damp_l, damp_a;             // Bullet base damping coefficients in the lane that owns the base body, 0 elsewhere
  float jdir_own, joff_own;         // the lane's own joint (part < 3; jdir 0 for part 3): scalars, NOT selects over jdir[] / joff[] --
                                    // the compiler turns such a select into a dynamically indexed load, which pushes the whole
                                    // struct out of registers (into LDS via promote-alloca)
  float int_c0, int_c1;             // integration (lane l owns DOF l and, l < 2, DOF 16 + l): d(coordinate) / d(velocity) -- 0 for the
                                    // angular base DOFs (the quaternion is integrated separately), 1 for the base position, jdir
                                    // (= +-1, orr_set_model checks) for a joint
};
// The coordinates a lane integrates, carried in registers over the sub-steps of a launch (the LDS copy is written every sub-step for the
// other lanes, never read back by the owner): x0 = base position component (lanes 3..5) or joint angle (lanes 6..15: joints 0..9),
// x1 = joint 10 + lane (lanes 0, 1).
struct OwnCoord { float x0, x1; };
__device__ __forceinline__ void load_own_coord(const Shared& S, int lane, OwnCoord& X) {
  X.x0 = lane < 3 ? 0.0f : (lane < 6 ? S.s[O(POS) + lane - 3] : S.s[O(Q) + lane - 6]);
  X.x1 = lane < 2 ? S.s[O(Q) + 10 + lane] : 0.0f;
}
__device__ __forceinline__ float part_suffix_sum(float x);
__device__ static void load_leg_const(const KParams& P, const Shared& S, int lane, LegConst& K) {
  const int leg = lane & 3, part = (lane >> 2) & 3, own = 3 * leg + (part < 3 ? part : 2);
  const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int j = 3 * leg + k;
    const bool on = k <= part && part < 3;   // part-3 lanes walk no joint at all: their link frame is the base frame
#pragma unroll
    for (int i = 0; i < 3; i++) K.r[k][i] = on ? S.m.joint_pos[j][i] : 0.0f;
    K.jdir[k] = on ? S.m.jdir[j] : 0.0f;
    K.joff[k] = S.m.joff[j];
  }
  // the part-3 lane of leg 0 (lane 12) owns the BASE body (its joint walk is the identity, so its link frame is the base frame and
  // its bias force the base's gyroscopic terms); the part-3 lanes of the other legs stay idle copies with zero inertia
  const bool real = part < 3, base = lane == 12;
  const int body = real ? own + 1 : 0;
#pragma unroll
  for (int i = 0; i < 3; i++) K.com[i] = real ? mc->link_com[own][i] : 0.0f;
  {
    // mass properties of the own body with this episode's randomisation ratios (controllable_env_randomizer_from_config.py:193-222,
    // 309-335), straight from the device table into registers: nothing else reads them, so they are not staged in LDS (the ratios of a
    // robot that resets inside this launch take effect with the next launch's load_leg_const, like its new state)
    const int g = mc->group[body];
    const float mr = S.s[O(MASS_RATIO) + g], ir = S.s[O(INERTIA_RATIO) + g];
    const float mass = mc->mass[body] * mr;
#pragma unroll
    for (int i = 0; i < 6; i++) K.Ic[i] = (real || base) ? mc->inertia[body][i] * ir + mc->inertia_pa[body][i] * mr : 0.0f;
    K.m = (real || base) ? mass : 0.0f;
    if (kCarrySubtreeMass) K.msub = part_suffix_sum(K.m);
  }
  K.jdir_own = real ? S.m.jdir[own] : 0.0f;
  K.joff_own = S.m.joff[own];
  K.damp_l = base ? S.s[O(BASE_DAMPING)] : 0.0f;
  K.damp_a = base ? S.s[O(BASE_DAMPING) + 1] : 0.0f;
  K.int_c0 = lane < 3 ? 0.0f : (lane < 6 ? 1.0f : S.m.jdir[lane < 6 ? 0 : lane - 6]);
  K.int_c1 = lane < 2 ? S.m.jdir[10 + lane] : 0.0f;
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
// LDS (T_L (6x3), H_L^-1 per leg: LegSolve) or stays in registers (the Cholesky factor of A0: BaseFactor) - the impulse response of a row is then
//   da0 = A0^-1 (Jb - T_L jl);  dqdd_L = H_L^-1 jl - T_L^T da0;  dqdd_K = -T_K^T da0  (K != L).
// Lane (leg = lane & 3, part = (lane >> 2) & 3): all lanes of a leg walk down its joints, but each computes the costly
// per-link terms (inertia about O, bias force) only for link `part` (lane 12 = (leg 0, part 3) for the base body itself);
// subtree sums run over the parts with DPP row shifts, F / H / bias torques of the three joints are exchanged through LDS
// (LegExchange).  The base system is then ASSEMBLED across the 16 lanes: lane (leg, k) contributes its own link's inertia /
// bias force and the k-th term of the leg's elimination, -T_k F_k^T and T_k b_k (T_k = column k of F H^-1), and one 16-lane
// DPP butterfly per entry sums bodies, legs and joints at once; the 6x6 solve itself is redundant in all lanes.  Elimination terms, sums,
// factorisation and solves share ONE layout on packed float pairs (Chol6Pk above).
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
  // multiply-add chains (the separate cross products + adds were six instructions more)
  oa[0] = fmaf(I[0], w[0], fmaf(I[3], w[1], fmaf(I[4], w[2], fmaf(h[1], v[2], -h[2] * v[1]))));
  oa[1] = fmaf(I[3], w[0], fmaf(I[1], w[1], fmaf(I[5], w[2], fmaf(h[2], v[0], -h[0] * v[2]))));
  oa[2] = fmaf(I[4], w[0], fmaf(I[5], w[1], fmaf(I[2], w[2], fmaf(h[0], v[1], -h[1] * v[0]))));
  ol[0] = fmaf(m, v[0], fmaf(h[2], w[1], -h[1] * w[2]));
  ol[1] = fmaf(m, v[1], fmaf(h[0], w[2], -h[2] * w[0]));
  ol[2] = fmaf(m, v[2], fmaf(h[1], w[0], -h[0] * w[1]));
}

// one joint on the way down the leg: pose of the link behind it, its motion axis S = (s; d x s) about O, spatial
// velocity and velocity-product acceleration.  For a joint beyond the lane's own link the constants are zero and the
// step is the identity (angle 0, rate 0, offset 0).
template <int AX, bool SAME_AXIS = false>
__device__ __forceinline__ void joint_down(const Shared& S, const LegConst& K, int k, int j, float sn, float cs, float Rw[9], float d[3],
                                           float Vw[3], float Vv[3], float Aa[3], float Al[3], float s[3], float sv[3], float& ad_out,
                                           float* c0 = nullptr) {
  const float ad = K.jdir[k] * S.s[O(QD) + j];
  ad_out = ad;
  // pose: d += Rw_parent r;  Rw = Rw_parent R(a)
#pragma unroll
  for (int i = 0; i < 3; i++)     // accumulated into d by multiply-adds (a separate sum + add is one instruction more per component)
    d[i] = fmaf(Rw[3 * i], K.r[k][0], fmaf(Rw[3 * i + 1], K.r[k][1], fmaf(Rw[3 * i + 2], K.r[k][2], d[i])));
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float p0 = Rw[3 * i], p1 = Rw[3 * i + 1], p2 = Rw[3 * i + 2];
    if (AX == 0) { Rw[3 * i + 1] = cs * p1 + sn * p2; Rw[3 * i + 2] = -sn * p1 + cs * p2; }
    else { Rw[3 * i] = cs * p0 - sn * p2; Rw[3 * i + 2] = sn * p0 + cs * p2; }
  }
  s[0] = Rw[AX]; s[1] = Rw[3 + AX]; s[2] = Rw[6 + AX];
  cross3(d, s, sv);
  // A += V x (S ad) = ad (Vw x s ; Vw x sv + Vv x s) with the velocity of the PARENT (the joint's own S ad drops out of the cross
  // products: s x s = 0 and (s ad) x (sv ad) + (sv ad) x (s ad) = 0), then V += S ad: 30 instructions instead of 39 per joint step.
  // 4096 robots 0.2256 -> 0.2236 ms, 8192 robots 0.3158 -> 0.3136 ms (round 4, interleaved A/B).
  {
    // SAME_AXIS: the joint's axis is the previous joint's (hip pitch -> knee: both about the link's y, a pure translation between them),
    // so Vw x s = (Vw_prev + qd_prev s) x s = Vw_prev x s: the previous joint's product is taken over (-6 instructions)
    float c0x, c0y, c0z;
    if (SAME_AXIS) { c0x = c0[0]; c0y = c0[1]; c0z = c0[2]; }
    else { c0x = Vw[1] * s[2] - Vw[2] * s[1]; c0y = Vw[2] * s[0] - Vw[0] * s[2]; c0z = Vw[0] * s[1] - Vw[1] * s[0]; }
    if (c0) { c0[0] = c0x; c0[1] = c0y; c0[2] = c0z; }
    const float c1x = fmaf(Vv[1], s[2], fmaf(-Vv[2], s[1], Vw[1] * sv[2] - Vw[2] * sv[1]));
    const float c1y = fmaf(Vv[2], s[0], fmaf(-Vv[0], s[2], Vw[2] * sv[0] - Vw[0] * sv[2]));
    const float c1z = fmaf(Vv[0], s[1], fmaf(-Vv[1], s[0], Vw[0] * sv[1] - Vw[1] * sv[0]));
    Aa[0] = fmaf(ad, c0x, Aa[0]); Aa[1] = fmaf(ad, c0y, Aa[1]); Aa[2] = fmaf(ad, c0z, Aa[2]);
    Al[0] = fmaf(ad, c1x, Al[0]); Al[1] = fmaf(ad, c1y, Al[1]); Al[2] = fmaf(ad, c1z, Al[2]);
#pragma unroll
    for (int i = 0; i < 3; i++) { Vw[i] = fmaf(ad, s[i], Vw[i]); Vv[i] = fmaf(ad, sv[i], Vv[i]); }
  }
}

// sum of x over the own and the later LINK parts (q..2) of one leg (lane = leg + 4 q): two DPP row shifts, zero beyond the
// row.  The part-3 lanes are not links of the chain (lane 12 carries the base body), so they are never a source: the bank masks
// switch off the destination quads that would read them (lanes 8..11 for row_shl:4, lanes 4..7 for row_shl:8).
__device__ __forceinline__ float part_suffix_sum(float x) {
  const int v = __float_as_int(x);
  const float a = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xB, true));  // row_shl:4, not into bank 2
  const float b = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x108, 0xF, 0xD, true));  // row_shl:8, not into bank 1
  return x + a + b;
}

struct BaseFactor { Chol6Pk F; };
__device__ static void leg_dynamics(const KParams& P, Shared& S, const LegConst& K, int lane, BaseFactor& BF) {
  const int leg = lane & 3, part = (lane >> 2) & 3;
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
  // sine / cosine of the joint angles: every lane evaluates ONE polynomial pair, for its own joint (part 3: angle 0), and gets
  // the joints in front of it from the lanes of the earlier parts of its leg (DPP row_shr:4 / :8); joints behind its own link
  // are identity steps (sin 0, cos 1)
  float sn0, cs0, sn1, cs1, sn2, cs2;
  {
    const int jo = 3 * leg + (part < 3 ? part : 2);
    float sno, cso;
    joint_sincos(K.jdir_own * (S.s[O(Q) + jo] - K.joff_own), &sno, &cso);
    const float sA = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sno), 0x114, 0xF, 0xF, true));  // row_shr:4
    const float cA = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cso), 0x114, 0xF, 0xF, true));
    const float sB = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sno), 0x118, 0xF, 0xF, true));  // row_shr:8
    const float cB = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cso), 0x118, 0xF, 0xF, true));
    // the selects over `part` as bank-masked DPP moves (a bank = the four lanes of one part): a lane keeps its own value where the mask
    // is off, and a shift from beyond the row writes 0 (bound_ctrl) -- the own values of the part-3 lanes are sin 0 = 0, cos 0 = 1
    //   sn0 / cs0 = [own, shr4, shr8, own]   sn1 = [0 (from beyond the row), own, shr4, own = 0]   cs1 = [1, own, shr4, own = 1]
    sn0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(sno), __float_as_int(sno), 0x114, 0xF, 0x2, false));
    sn0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(sn0), __float_as_int(sno), 0x118, 0xF, 0x4, false));
    cs0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cso), __float_as_int(cso), 0x114, 0xF, 0x2, false));
    cs0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cs0), __float_as_int(cso), 0x118, 0xF, 0x4, false));
    sn1 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(sno), __float_as_int(sno), 0x114, 0xF, 0x5, true));
    cs1 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cso), __float_as_int(cso), 0x114, 0xF, 0x4, false));
    cs1 = part == 0 ? 1.0f : cs1;
    (void)sA; (void)cA; (void)sB; (void)cB;
    sn2 = part == 2 ? sno : 0.0f;
    cs2 = part == 2 ? cso : 1.0f;
  }
  float c0k[3];
  joint_down<0>(S, K, 0, 3 * leg, sn0, cs0, Rw, d, Vw, Vv, Aa, Al, s0, sv0, ad0);
  joint_down<1>(S, K, 1, 3 * leg + 1, sn1, cs1, Rw, d, Vw, Vv, Aa, Al, s1, sv1, ad1, c0k);
  joint_down<1, true>(S, K, 2, 3 * leg + 2, sn2, cs2, Rw, d, Vw, Vv, Aa, Al, s2, sv2, ad2, c0k);
  // own joint: axis and rate
  float so[3], svo[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    // the joints behind a lane's own link are identity steps with zero offsets: for a part-1 lane (s2, sv2) ARE (s1, sv1), bit for bit
    so[i] = part == 0 ? s0[i] : (part == 3 ? 0.0f : s2[i]);        // part 3 owns no joint: F, T columns = 0
    svo[i] = part == 0 ? sv0[i] : (part == 3 ? 0.0f : sv2[i]);
  }
  const float ado = part == 0 ? ad0 : (part == 1 ? ad1 : ad2);
  {  // pose and joint axis of the own link for the constraint rows (part-3 lanes: dump slot)
    LinkCache& L = S.ph.sub.dyn.lc[part < 3 ? 3 * leg + part : 12 + leg];
#pragma unroll
    for (int i = 0; i < 9; i++) L.Rw[i] = Rw[i];
#pragma unroll
    for (int i = 0; i < 3; i++) { L.ow[i] = S.s[O(POS) + i] + d[i]; L.s[i] = so[i]; L.sv[i] = svo[i]; }
  }
  // ---- own link: spatial inertia about O and bias force f = I A + V x* (I V) ----
  float I[6], h[3], m = K.m, f[6], pacc[6];
  {
    float c[3];
#pragma unroll
    for (int i = 0; i < 3; i++) c[i] = fmaf(Rw[3 * i], K.com[0], fmaf(Rw[3 * i + 1], K.com[1], fmaf(Rw[3 * i + 2], K.com[2], d[i])));
    h[0] = m * c[0]; h[1] = m * c[1]; h[2] = m * c[2];
    rot_sym_full(Rw, K.Ic, I);
    // Bias force f = I A + V x* (I V) evaluated at the link's COM and shifted to O (round 4: 54 instead of ~80 instructions; the same
    // vector - with L = (Ic w + c x p; p), p = m (v_O + w x c): the cross terms of the spatial form collapse to c x (m a_c + w x p)):
    //   f_lin = m (Al + Aa x c) + w x p,    f_ang = Ic Aa + w x (Ic w) + c x f_lin      (Ic = inertia about the COM, world axes)
    float Pa[3], Pl[3];
    {
      const float ux = fmaf(Vw[1], c[2], fmaf(-Vw[2], c[1], Vv[0])), uy = fmaf(Vw[2], c[0], fmaf(-Vw[0], c[2], Vv[1]));
      const float uz = fmaf(Vw[0], c[1], fmaf(-Vw[1], c[0], Vv[2]));
      Pl[0] = m * ux; Pl[1] = m * uy; Pl[2] = m * uz;                                            // linear momentum p
      const float ax = fmaf(Aa[1], c[2], fmaf(-Aa[2], c[1], Al[0])), ay = fmaf(Aa[2], c[0], fmaf(-Aa[0], c[2], Al[1]));
      const float az = fmaf(Aa[0], c[1], fmaf(-Aa[1], c[0], Al[2]));
      f[3] = fmaf(m, ax, Vw[1] * Pl[2] - Vw[2] * Pl[1]);
      f[4] = fmaf(m, ay, Vw[2] * Pl[0] - Vw[0] * Pl[2]);
      f[5] = fmaf(m, az, Vw[0] * Pl[1] - Vw[1] * Pl[0]);
      symv(I, Vw, Pa);                                                                           // Ic w (the base body has c = 0: its angular momentum about O)
      float tx = Vw[1] * Pa[2] - Vw[2] * Pa[1], ty = Vw[2] * Pa[0] - Vw[0] * Pa[2], tz = Vw[0] * Pa[1] - Vw[1] * Pa[0];
      tx = fmaf(I[0], Aa[0], fmaf(I[3], Aa[1], fmaf(I[4], Aa[2], tx)));
      ty = fmaf(I[3], Aa[0], fmaf(I[1], Aa[1], fmaf(I[5], Aa[2], ty)));
      tz = fmaf(I[4], Aa[0], fmaf(I[5], Aa[1], fmaf(I[2], Aa[2], tz)));
      f[0] = fmaf(c[1], f[5], fmaf(-c[2], f[4], tx));
      f[1] = fmaf(c[2], f[3], fmaf(-c[0], f[5], ty));
      f[2] = fmaf(c[0], f[4], fmaf(-c[1], f[3], tz));
    }
    // inertia about O for the composite sums and the base matrix
    const float hc = h[0] * c[0] + h[1] * c[1] + h[2] * c[2];
    I[0] += hc - h[0] * c[0]; I[1] += hc - h[1] * c[1]; I[2] += hc - h[2] * c[2];
    I[3] -= h[0] * c[1]; I[4] -= h[0] * c[2]; I[5] -= h[1] * c[2];
    // force side of the base system: the link's bias force (+ Bullet's base damping on the base body, whose momentum is
    // (Pa, Pl) = (I w, m v): torque k_a I w, force k_l m v; btMultiBody; damp_* are zero in every lane but the base body's, whose COM is O)
#pragma unroll
    for (int i = 0; i < 3; i++) { pacc[i] = f[i] + K.damp_a * Pa[i]; pacc[3 + i] = f[3 + i] + K.damp_l * Pl[i]; }
  }
  // own link's share of the matrix of the base system, before the subtree sums overwrite I / h / m: inertia about O, first
  // moment, mass
  float Iacc[6], Hacc[9], Macc[6];
  {
#pragma unroll
    for (int i = 0; i < 6; i++) Iacc[i] = I[i];
    Hacc[0] = 0.0f;   Hacc[1] = -h[2]; Hacc[2] = h[1];      // skew(h): top-right block of the composite spatial inertia
    Hacc[3] = h[2];   Hacc[4] = 0.0f;  Hacc[5] = -h[0];
    Hacc[6] = -h[1];  Hacc[7] = h[0];  Hacc[8] = 0.0f;
    Macc[0] = Macc[1] = Macc[2] = m; Macc[3] = Macc[4] = Macc[5] = 0.0f;
  }
  // ---- way up: composite inertia and force sum of the subtree behind the own joint (sum over the leg's later parts) ----
#pragma unroll
  for (int i = 0; i < 6; i++) { I[i] = part_suffix_sum(I[i]); f[i] = part_suffix_sum(f[i]); }
#pragma unroll
  for (int i = 0; i < 3; i++) h[i] = part_suffix_sum(h[i]);
  m = kCarrySubtreeMass ? K.msub : part_suffix_sum(m);
  // own column of F and of the leg's joint-space inertia H (entries H[i][part], i <= part), own bias torque
  float Fo[6], bo;
  {
    spatial_inertia_mul(I, h, m, so, svo, &Fo[0], &Fo[3]);
    auto dot6 = [](const float a[3], const float b[3], const float x[3], const float y[3]) __attribute__((always_inline)) {
      return fmaf(a[0], x[0], fmaf(a[1], x[1], fmaf(a[2], x[2], fmaf(b[0], y[0], fmaf(b[1], y[1], b[2] * y[2])))));   // one chain: 6, not 7
    };
    bo = S.tau[3 * leg + (part < 3 ? part : 2)] - dot6(so, svo, &f[0], &f[3]);
    const float hc0 = dot6(s0, sv0, &Fo[0], &Fo[3]), hc1 = dot6(s1, sv1, &Fo[0], &Fo[3]), hc2 = dot6(s2, sv2, &Fo[0], &Fo[3]);
    {  // part-3 lanes: index 3 = dump slot
      LegExchange& X = S.ph.sub.dyn.legx[leg];
#pragma unroll
      for (int i = 0; i < 6; i++) X.F[part][i] = Fo[i];
      X.b[part] = bo;
      X.Hc[part][0] = hc0; X.Hc[part][1] = hc1; X.Hc[part][2] = hc2;
    }
  }
  WSYNC();
  // ---- every lane of the leg: all three F columns and H ----
  float b[3];
  pk2 F2[3][3];   // F2[k][p] = (F_k[2p], F_k[2p+1]): the columns of F come out of LDS as aligned pairs
  float H00, H01, H02, H11, H12, H22;
  {
    const LegExchange& X = S.ph.sub.dyn.legx[leg];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      static_assert(sizeof(X.F[0]) == 24 && alignof(LegExchange) >= 8, "pairs");
      const pk2* XF = reinterpret_cast<const pk2*>(&X.F[k][0]);
#pragma unroll
      for (int q = 0; q < 3; q++) F2[k][q] = XF[q];
      b[k] = X.b[k];
    }
    H00 = X.Hc[0][0]; H01 = X.Hc[1][0]; H11 = X.Hc[1][1]; H02 = X.Hc[2][0]; H12 = X.Hc[2][1]; H22 = X.Hc[2][2];
  }
  float Hi[6];  // H^-1, symmetric (00 11 22 01 02 12), by cofactors
  {
    const float c00 = H11 * H22 - H12 * H12, c01 = H02 * H12 - H01 * H22, c02 = H01 * H12 - H02 * H11;
    const float c11 = H00 * H22 - H02 * H02, c12 = H01 * H02 - H00 * H12, c22 = H00 * H11 - H01 * H01;
    const float idet = __builtin_amdgcn_rcpf(H00 * c00 + H01 * c01 + H02 * c02);
    Hi[0] = c00 * idet; Hi[1] = c11 * idet; Hi[2] = c22 * idet; Hi[3] = c01 * idet; Hi[4] = c02 * idet; Hi[5] = c12 * idet;
  }
  // row / column `part` of H^-1 (zero in the part-3 lanes: they own no joint)
  const float hq0 = part == 0 ? Hi[0] : (part == 1 ? Hi[3] : (part == 2 ? Hi[4] : 0.0f));
  const float hq1 = part == 0 ? Hi[3] : (part == 1 ? Hi[1] : (part == 2 ? Hi[5] : 0.0f));
  const float hq2 = part == 0 ? Hi[4] : (part == 1 ? Hi[5] : (part == 2 ? Hi[2] : 0.0f));
  float Tq[6];  // column `part` of T = F H^-1
  pk2 Tq2[3];
#pragma unroll
  for (int q = 0; q < 3; q++) { Tq2[q] = F2[0][q] * hq0 + F2[1][q] * hq1 + F2[2][q] * hq2; Tq[2 * q] = Tq2[q].x; Tq[2 * q + 1] = Tq2[q].y; }
  {  // part-3 lanes: one shared dump slot; H^-1 is the same in all lanes of the leg, every one of them stores it
    LegSolve& Q = S.leg[leg];
    float* const tdst = part < 3 ? &Q.T[part][0] : &S.tdump[0];
#pragma unroll
    for (int q = 0; q < 3; q++) reinterpret_cast<pk2*>(tdst)[q] = Tq2[q];
#pragma unroll
    for (int i = 0; i < 6; i++) Q.Hi[i] = Hi[i];
  }
  // The base matrix is accumulated, summed over the 16 lanes and factorised in ONE layout: the lower triangle by columns with the rows
  // (2,3) and (4,5) paired (Chol6Pk).  Entry (j, i), j >= i, is the (i, j) entry of the old upper-triangle form: own inertia - Tq[i] Fo[j],
  // i.e. column i = (own column) - Tq[i] * (Fo[i..5]) with Fo in pairs -- the same products and sums, 13 instead of 21 instructions.
  float a0[6];
  {
    const pk2 Fo23 = {Fo[2], Fo[3]}, Fo45 = {Fo[4], Fo[5]};
    float d0 = fmaf(-Tq[0], Fo[0], Iacc[0]), a10 = fmaf(-Tq[0], Fo[1], Iacc[3]);
    pk2 a0a = pk_nfma(Fo23, Tq[0], pk2{Iacc[4], Hacc[0]}), a0b = pk_nfma(Fo45, Tq[0], pk2{Hacc[1], Hacc[2]});
    float d1 = fmaf(-Tq[1], Fo[1], Iacc[1]);
    pk2 a1a = pk_nfma(Fo23, Tq[1], pk2{Iacc[5], Hacc[3]}), a1b = pk_nfma(Fo45, Tq[1], pk2{Hacc[4], Hacc[5]});
    pk2 d2a = pk_nfma(Fo23, Tq[2], pk2{Iacc[2], Hacc[6]}), a2b = pk_nfma(Fo45, Tq[2], pk2{Hacc[7], Hacc[8]});
    float d3 = fmaf(-Tq[3], Fo[3], Macc[0]);
    pk2 a3b = pk_nfma(Fo45, Tq[3], pk2{Macc[3], Macc[4]});
    pk2 d4a = pk_nfma(Fo45, Tq[4], pk2{Macc[1], Macc[5]});
    float d5 = fmaf(-Tq[5], Fo[5], Macc[2]);
    pk2 p01 = __builtin_elementwise_fma(Tq2[0], pk2{bo, bo}, pk2{pacc[0], pacc[1]}), p23 = __builtin_elementwise_fma(Tq2[1], pk2{bo, bo}, pk2{pacc[2], pacc[3]});
    pk2 p45 = __builtin_elementwise_fma(Tq2[2], pk2{bo, bo}, pk2{pacc[4], pacc[5]});
    // base system: sum over all bodies / legs / joints = over the robot's 16 lanes (DPP butterfly fused into the adds)
    auto sum2 = [](pk2& v) __attribute__((always_inline)) { v.x = row_sum16(v.x); v.y = row_sum16(v.y); };
    d0 = row_sum16(d0); a10 = row_sum16(a10); d1 = row_sum16(d1); d3 = row_sum16(d3); d5 = row_sum16(d5);
    sum2(a0a); sum2(a0b); sum2(a1a); sum2(a1b); sum2(d2a); sum2(a2b); sum2(a3b); sum2(d4a); sum2(p01); sum2(p23); sum2(p45);
    chol6_pk(d0, d1, d3, d5, a10, d2a, d4a, a0a, a0b, a1a, a1b, a2b, a3b, BF.F);
    chol6_solve_pk(BF.F, -p01.x, -p01.y, -p23, -p45, a0);
  }
  // joint accelerations qdd = H^-1 (b - F^T a0): row `part` of H^-1 for the lane's own joint; written as the unconstrained
  // velocity u* = u + dt udot by the lane that owns the joint (it has the joint rate)
  const float dt = P.cfg.sim_dt;
  {
    // row `part` of H^-1 (b - F^T a0) = hq . b - (sum_k hq_k F_k) . a0 = hq . b - Tq . a0: the lane's own column of T is already there
    const float hb = hq0 * b[0] + hq1 * b[1] + hq2 * b[2];
    const float ta = Tq[0] * a0[0] + Tq[1] * a0[1] + Tq[2] * a0[2] + Tq[3] * a0[3] + Tq[4] * a0[4] + Tq[5] * a0[5];
    S.ustar[part < 3 ? 6 + 3 * leg + part : 18 + leg] = ado + dt * (hb - ta);   // part 3: dump slot
  }
  {
    // classical base acceleration (Bullet: vdot = a_lin + w x v); gravity = uniform-field offset.  The same in every lane: all of
    // them store it (no divergent `if`)
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
  float cfm;                   // constraint-force mixing of a soft toe normal row (0 = rigid)
  float lo_c, hi_c, mu_e;      // bounds = constant part -/+ mu_e * lambda_normal
  float wa[6], wq[12];         // own impulse response M^-1 J^T (base part, joint part): kept in registers for the Delassus columns
};
// contact rows only (bank A): contact point relative to the base COM and its velocity per unit rate of the leg's joints; the
// Jacobian of a contact row is ((rr x dir, dir), dir . ck[k]), which the Delassus columns exploit per leg (dpp_contact_triplet)
struct ContactGeom {
  float rr0, rr1, rr2, c00, c01, c02, c10, c11, c12, c20, c21, c22;   // scalars (not arrays): stays in registers for the inline-asm operands
};

// Jacobian, right-hand side (not yet scaled by 1/diag) and bounds of row slot `slot`.  BANK 0: knee-friction and contact slots
// (0..3, 16..27), BANK 1: joint-limit slots (4..15) -- two instantiations, so that the joint-limit bank carries no contact code
// and its base Jacobian is the compile-time constant zero.
template <int BANK>
__device__ __forceinline__ void row_setup(const Shared& S, const orr_config& cfg, int slot, bool enable, float dt, float inv_dt,
                                          float erp_dt, Row& R, ContactGeom& G, float* limit_margin = nullptr) {
  if (BANK == 0) G = ContactGeom{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  R.active = false; R.leg = 0; R.nrm_slot = -1; R.warm = -1;
#pragma unroll
  for (int i = 0; i < 6; i++) R.Jb[i] = 0.0f;
  R.jl[0] = R.jl[1] = R.jl[2] = 0.0f;
  R.rhs = 0.0f; R.jdi = 0.0f; R.lam = 0.0f; R.w = 0.0f; R.lam_n = 0.0f; R.cfm = 0.0f;
  float lo = 0.0f, hi = 0.0f, mu = 0.0f;
  if (BANK == 0 && slot < 4) {
    R.leg = slot;
    const float fr = S.s[O(KNEE_FRICTION) + R.leg];
    R.active = fr > 0.0f;
    R.jl[2] = 1.0f;
    lo = -fr * dt; hi = fr * dt;
    R.rhs = -S.ustar[6 + 3 * R.leg + 2];
  } else if (BANK == 1) {
    const int j = slot - 4;
    R.leg = j / 3;
    const int kk = j - 3 * R.leg;
    const float a = S.m.jdir[j] * (S.s[O(Q) + j] - S.m.joff[j]);
    const float pen_lo = a - S.m.joint_lo[j], pen_hi = S.m.joint_hi[j] - a;
    if (limit_margin) *limit_margin = fminf(pen_lo, pen_hi) - cfg.limit_activation;   // how far the joint is from getting a limit row
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
    const LinkCache& Lb = S.ph.sub.dyn.lc[3 * leg + 2];
    // lower legs are feet (minitaur.py:842-844): the leg touches the ground with its toe sphere or its shank sphere, whichever is
    // lower (one contact point per leg and sub-step; shank_radius 0 = toe only)
    float cw[3], cs[3];
    mv3(Lb.Rw, S.m.toe_pos[leg], cw);
    mv3(Lb.Rw, S.m.shank_pos[leg], cs);
    const float dist_t = cw[2] + Lb.ow[2] - S.m.toe_radius, dist_s = cs[2] + Lb.ow[2] - S.m.shank_radius;
    const bool shank = S.m.shank_radius > 0.0f && dist_s < dist_t;
    const float dist = shank ? dist_s : dist_t;
    R.active = dist < cfg.contact_margin;
    {   // instrumented build only (tools/dual_contact.py): normal-row lanes count the leg-sub-steps by which spheres touch
      const bool cnt = enable && d == 0, has_s = S.m.shank_radius > 0.0f;
      const bool t_in = dist_t < cfg.contact_margin, s_in = has_s && dist_s < cfg.contact_margin;
      ORR_DUAL_COUNT(0, cnt);                                        // leg-sub-steps
      ORR_DUAL_COUNT(1, cnt && (t_in || s_in));                      // ... with a contact row
      ORR_DUAL_COUNT(2, cnt && t_in && s_in);                        // both spheres within the contact margin (Bullet: two contact points)
      ORR_DUAL_COUNT(3, cnt && dist_t < 0.0f && has_s && dist_s < 0.0f);   // both spheres penetrating
      ORR_DUAL_COUNT(4, cnt && s_in && !t_in);                       // shank only
      ORR_DUAL_COUNT(5, cnt && shank && R.active);                   // the row was made at the shank sphere
      (void)cnt; (void)has_s; (void)t_in; (void)s_in;
    }
    const float Pw[3] = {(shank ? cs[0] : cw[0]) + Lb.ow[0], (shank ? cs[1] : cw[1]) + Lb.ow[1], dist};
    const float dir[3] = {d == 1 ? 1.0f : 0.0f, d == 2 ? 1.0f : 0.0f, d == 0 ? 1.0f : 0.0f};
    float rr[3] = {Pw[0] - S.s[O(POS)], Pw[1] - S.s[O(POS) + 1], Pw[2] - S.s[O(POS) + 2]};
    cross3(rr, dir, &R.Jb[0]);
    R.Jb[3] = dir[0]; R.Jb[4] = dir[1]; R.Jb[5] = dir[2];
    float rel = R.Jb[0] * S.ustar[0] + R.Jb[1] * S.ustar[1] + R.Jb[2] * S.ustar[2] + R.Jb[3] * S.ustar[3] + R.Jb[4] * S.ustar[4] + R.Jb[5] * S.ustar[5];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      // velocity of the contact point per unit joint rate: s x (P - o) = s x rr + (d x s), rr and d relative to the base COM
      const LinkCache& L = S.ph.sub.dyn.lc[3 * leg + k];
      float cr[3];
      cross3(L.s, rr, cr);
      cr[0] += L.sv[0]; cr[1] += L.sv[1]; cr[2] += L.sv[2];
      if (k == 0) { G.c00 = cr[0]; G.c01 = cr[1]; G.c02 = cr[2]; }
      else if (k == 1) { G.c10 = cr[0]; G.c11 = cr[1]; G.c12 = cr[2]; }
      else { G.c20 = cr[0]; G.c21 = cr[1]; G.c22 = cr[2]; }
      R.jl[k] = dir[0] * cr[0] + dir[1] * cr[1] + dir[2] * cr[2];
      rel += R.jl[k] * S.ustar[6 + 3 * leg + k];
    }
    G.rr0 = rr[0]; G.rr1 = rr[1]; G.rr2 = rr[2];
    R.warm = 3 * leg + d;
    if (d == 0) {
      lo = 0.0f; hi = 1e30f;
      // a TOE contact may be soft (URDF <contact><stiffness/><damping/>, see ModelHot): its own erp, and cfm on the row's diagonal
      R.rhs = dist > 0.0f ? -rel - dist * inv_dt : -rel - dist * (shank ? erp_dt : S.m.contact_erp_dt);
      R.cfm = shank ? 0.0f : S.m.contact_cfm;
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

// Bank A (knee-friction slots 0..3, contact slots 16..27) WITHOUT divergent control flow.  The generic form above sends the knee lanes
// and the contact lanes (and inside those the normal and the friction lanes) through separate exec-masked regions; each region costs a
// lone wave its saveexec / branch overhead, ~30 moves that set the other branch's values at the merge, and - the expensive part - an LDS
// round trip that cannot overlap with anything (the region's own loads: friction coefficient, knee friction, contact softness).  Here
// every lane loads all of these up front, computes the contact geometry of "its" leg (knee lane l: leg l; the result is discarded) and
// picks its row's values with selects on lane-constant masks.  Same arithmetic per row, bit for bit.
__device__ __forceinline__ void row_setup_bank_a(const Shared& S, const orr_config& cfg, int slot, bool enable, float dt, float inv_dt,
                                                 float erp_dt, Row& R, ContactGeom& G) {
  const bool knee = slot < 4;
  const int leg = knee ? slot : (slot < 20 ? slot - 16 : (slot - 20) >> 1);
  const int d = (knee || slot < 20) ? 0 : 1 + ((slot - 20) & 1);
  const bool normal = !knee && d == 0, fric = d != 0;
  // everything that comes out of LDS, for every lane, before any use (opaque copies keep the loads here)
  float fr = S.s[O(KNEE_FRICTION) + leg], uknee = S.ustar[6 + 3 * leg + 2], mu_s = S.s[O(FOOT_MU)];
  float cfm_m = S.m.contact_cfm, erp_m = S.m.contact_erp_dt;
  asm volatile("" : "+v"(fr), "+v"(uknee), "+v"(mu_s), "+v"(cfm_m), "+v"(erp_m));
  R.leg = leg;
  const LinkCache& Lb = S.ph.sub.dyn.lc[3 * leg + 2];
  float cw[3], cs[3];
  mv3(Lb.Rw, S.m.toe_pos[leg], cw);
  mv3(Lb.Rw, S.m.shank_pos[leg], cs);
  const float dist_t = cw[2] + Lb.ow[2] - S.m.toe_radius, dist_s = cs[2] + Lb.ow[2] - S.m.shank_radius;
  const bool shank = S.m.shank_radius > 0.0f && dist_s < dist_t;
  const float dist = shank ? dist_s : dist_t;
  {   // instrumented build only (tools/dual_contact.py): normal-row lanes count the leg-sub-steps by which spheres touch
    const bool cnt = enable && normal, has_s = S.m.shank_radius > 0.0f;
    const bool t_in = dist_t < cfg.contact_margin, s_in = has_s && dist_s < cfg.contact_margin;
    ORR_DUAL_COUNT(0, cnt);
    ORR_DUAL_COUNT(1, cnt && (t_in || s_in));
    ORR_DUAL_COUNT(2, cnt && t_in && s_in);
    ORR_DUAL_COUNT(3, cnt && dist_t < 0.0f && has_s && dist_s < 0.0f);
    ORR_DUAL_COUNT(4, cnt && s_in && !t_in);
    ORR_DUAL_COUNT(5, cnt && shank && dist < cfg.contact_margin);
    (void)cnt; (void)has_s; (void)t_in; (void)s_in;
  }
  const float Pw[3] = {(shank ? cs[0] : cw[0]) + Lb.ow[0], (shank ? cs[1] : cw[1]) + Lb.ow[1], dist};
  const float dir[3] = {d == 1 ? 1.0f : 0.0f, d == 2 ? 1.0f : 0.0f, d == 0 ? 1.0f : 0.0f};
  float rr[3] = {Pw[0] - S.s[O(POS)], Pw[1] - S.s[O(POS) + 1], Pw[2] - S.s[O(POS) + 2]};
  float Jb[6];
  cross3(rr, dir, &Jb[0]);
  Jb[3] = dir[0]; Jb[4] = dir[1]; Jb[5] = dir[2];
  float rel = Jb[0] * S.ustar[0] + Jb[1] * S.ustar[1] + Jb[2] * S.ustar[2] + Jb[3] * S.ustar[3] + Jb[4] * S.ustar[4] + Jb[5] * S.ustar[5];
  float jl[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // velocity of the contact point per unit joint rate: s x (P - o) = s x rr + (d x s), rr and d relative to the base COM
    const LinkCache& L = S.ph.sub.dyn.lc[3 * leg + k];
    const float cr[3] = {fmaf(L.s[1], rr[2], fmaf(-L.s[2], rr[1], L.sv[0])), fmaf(L.s[2], rr[0], fmaf(-L.s[0], rr[2], L.sv[1])),
                         fmaf(L.s[0], rr[1], fmaf(-L.s[1], rr[0], L.sv[2]))};
    if (k == 0) { G.c00 = cr[0]; G.c01 = cr[1]; G.c02 = cr[2]; }
    else if (k == 1) { G.c10 = cr[0]; G.c11 = cr[1]; G.c12 = cr[2]; }
    else { G.c20 = cr[0]; G.c21 = cr[1]; G.c22 = cr[2]; }
    jl[k] = dir[0] * cr[0] + dir[1] * cr[1] + dir[2] * cr[2];
    rel += jl[k] * S.ustar[6 + 3 * leg + k];
  }
  G.rr0 = rr[0]; G.rr1 = rr[1]; G.rr2 = rr[2];   // only the contact lanes' geometry is ever read (dpp_contact_triplet<4 + g>)
  // the row of this lane
  const bool active = enable && (knee ? fr > 0.0f : dist < cfg.contact_margin);
#pragma unroll
  for (int i = 0; i < 6; i++) R.Jb[i] = knee ? 0.0f : Jb[i];
  R.jl[0] = knee ? 0.0f : jl[0]; R.jl[1] = knee ? 0.0f : jl[1]; R.jl[2] = knee ? 1.0f : jl[2];
  // right-hand side: knee motor -u*; friction -rel; normal -rel - dist / dt (open) or -rel - dist erp / dt (penetrating; a toe may be soft)
  const float kpen = normal ? (dist > 0.0f ? inv_dt : (shank ? erp_dt : erp_m)) : 0.0f;     // one multiply-add, as in the generic form
  const float rhs = knee ? -uknee : -rel - dist * kpen;
  R.rhs = active ? rhs : 0.0f;
  R.cfm = (normal && !shank) ? cfm_m : 0.0f;
  R.nrm_slot = fric ? 16 + leg : -1;
  R.warm = knee ? -1 : 3 * leg + d;
  R.active = active;
  R.jdi = 0.0f; R.lam = 0.0f; R.w = 0.0f; R.lam_n = 0.0f;
  // bounds as (constant part) + mu * lambda_normal: friction rows have a zero constant part, the others mu = 0; an inactive row is
  // pinned to zero
  const float hi = knee ? fr * dt : 1e30f, lo = knee ? -fr * dt : 0.0f;
  R.mu_e = (active && fric) ? mu_s * cfg.plane_friction : 0.0f;
  R.hi_c = (active && !fric) ? hi : 0.0f;
  R.lo_c = (active && !fric) ? lo : 0.0f;
}

// The same row setup with Bullet's friction anchors: a COPY of the function above with the cached contact point worked in (the default
// kernels keep the function above token for token: a semantically neutral rewrite of it moved the one-wave kernel by six instructions
// per sub-step and 0.7 % of run time, round 5).
// Bullet's friction anchor (orr_model::friction_anchor, ABI v5; the ANCHOR variant of the kernels): the cached contact point of the lane's
// leg - the point on the toe in the lower-leg link frame, the point on the plane in world - carried in REGISTERS over the sub-steps of a
// launch by each of the leg's three contact lanes (normal + two friction rows).  They make the same decisions from the same inputs
// (link pose and last sub-step's impulses from LDS), so the copies stay equal; the normal-row lane loads / stores the record's words.
struct AnchorState { float la[3], wb[3]; int valid; };
__device__ __forceinline__ void row_setup_bank_a_anchor(const Shared& S, const orr_config& cfg, int slot, bool enable, float dt, float inv_dt,
                                                 float erp_dt, Row& R, ContactGeom& G, AnchorState* AS, bool anchor_robot) {
  constexpr bool ANCHOR = true;
  const bool knee = slot < 4;
  const int leg = knee ? slot : (slot < 20 ? slot - 16 : (slot - 20) >> 1);
  const int d = (knee || slot < 20) ? 0 : 1 + ((slot - 20) & 1);
  const bool normal = !knee && d == 0, fric = d != 0;
  // everything that comes out of LDS, for every lane, before any use (opaque copies keep the loads here)
  float fr = S.s[O(KNEE_FRICTION) + leg], uknee = S.ustar[6 + 3 * leg + 2], mu_s = S.s[O(FOOT_MU)];
  float cfm_m = S.m.contact_cfm, erp_m = S.m.contact_erp_dt;
  asm volatile("" : "+v"(fr), "+v"(uknee), "+v"(mu_s), "+v"(cfm_m), "+v"(erp_m));
  R.leg = leg;
  const LinkCache& Lb = S.ph.sub.dyn.lc[3 * leg + 2];
  float cw[3], cs[3];
  mv3(Lb.Rw, S.m.toe_pos[leg], cw);
  mv3(Lb.Rw, S.m.shank_pos[leg], cs);
  const float dist_t = cw[2] + Lb.ow[2] - S.m.toe_radius, dist_s = cs[2] + Lb.ow[2] - S.m.shank_radius;
  const bool shank = S.m.shank_radius > 0.0f && dist_s < dist_t;
  float dist = shank ? dist_s : dist_t;
  float Pw0 = (shank ? cs[0] : cw[0]) + Lb.ow[0], Pw1 = (shank ? cs[1] : cw[1]) + Lb.ow[1];
  float drift_x = 0.0f, drift_y = 0.0f;
  bool have = dist < cfg.contact_margin;
  if constexpr (ANCHOR) {
    // btManifoldResult::addContactPoint -> btPersistentManifold::replaceContactPoint -> refreshContactPoints, one cached point per toe
    // (the oracle's physics_substep, same order of operations): the fresh sphere-plane point replaces the cached one only when the cached
    // point's friction impulse of the last solve left the cone (or the point moved further than the breaking threshold over the toe's
    // surface); the cached points are carried along by their bodies, distance and tangential offset are re-measured, and the point is
    // dropped when either exceeds the threshold.  A leg on its shank sphere, or a robot whose model has no anchors, keeps the plain point.
    AnchorState& A = *AS;
    if (anchor_robot) {
      int valid = A.valid;
      if (shank) valid = 0;
      else {
        const float m2 = cfg.contact_margin * cfg.contact_margin;
        if (have) {
          // the fresh point on the toe, in the link frame: toe centre - r * (world z in link coordinates)
          const float l0 = S.m.toe_pos[leg][0] - S.m.toe_radius * Lb.Rw[6], l1 = S.m.toe_pos[leg][1] - S.m.toe_radius * Lb.Rw[7];
          const float l2 = S.m.toe_pos[leg][2] - S.m.toe_radius * Lb.Rw[8];
          const float ln = S.s[O(LAMBDA) + 3 * leg], t1 = S.s[O(LAMBDA) + 3 * leg + 1], t2 = S.s[O(LAMBDA) + 3 * leg + 2];
          const float b = mu_s * cfg.plane_friction * ln;
          const float e0 = l0 - A.la[0], e1 = l1 - A.la[1], e2 = l2 - A.la[2];
          const bool replace = !valid || (t1 * t1 + t2 * t2 > b * b) || (e0 * e0 + e1 * e1 + e2 * e2 >= m2);
          if (replace) { A.la[0] = l0; A.la[1] = l1; A.la[2] = l2; A.wb[0] = Pw0; A.wb[1] = Pw1; A.wb[2] = 0.0f; valid = 1; }
        }
        if (valid) {
          float pa[3];
          mv3(Lb.Rw, A.la, pa);
          pa[0] += Lb.ow[0]; pa[1] += Lb.ow[1]; pa[2] += Lb.ow[2];
          const float dn = pa[2] - A.wb[2], dx = pa[0] - A.wb[0], dy = pa[1] - A.wb[1];
          if (dn > cfg.contact_margin || dx * dx + dy * dy > m2) { valid = 0; have = false; }
          else { have = true; dist = dn; Pw0 = pa[0]; Pw1 = pa[1]; drift_x = dx; drift_y = dy; }
        } else have = false;
      }
      A.valid = valid;
    }
  }
  {   // instrumented build only (tools/dual_contact.py): normal-row lanes count the leg-sub-steps by which spheres touch
    const bool cnt = enable && normal, has_s = S.m.shank_radius > 0.0f;
    const bool t_in = dist_t < cfg.contact_margin, s_in = has_s && dist_s < cfg.contact_margin;
    ORR_DUAL_COUNT(0, cnt);
    ORR_DUAL_COUNT(1, cnt && (t_in || s_in));
    ORR_DUAL_COUNT(2, cnt && t_in && s_in);
    ORR_DUAL_COUNT(3, cnt && dist_t < 0.0f && has_s && dist_s < 0.0f);
    ORR_DUAL_COUNT(4, cnt && s_in && !t_in);
    ORR_DUAL_COUNT(5, cnt && shank && dist < cfg.contact_margin);
    (void)cnt; (void)has_s; (void)t_in; (void)s_in;
  }
  const float Pw[3] = {Pw0, Pw1, dist};
  const float dir[3] = {d == 1 ? 1.0f : 0.0f, d == 2 ? 1.0f : 0.0f, d == 0 ? 1.0f : 0.0f};
  float rr[3] = {Pw[0] - S.s[O(POS)], Pw[1] - S.s[O(POS) + 1], Pw[2] - S.s[O(POS) + 2]};
  float Jb[6];
  cross3(rr, dir, &Jb[0]);
  Jb[3] = dir[0]; Jb[4] = dir[1]; Jb[5] = dir[2];
  float rel = Jb[0] * S.ustar[0] + Jb[1] * S.ustar[1] + Jb[2] * S.ustar[2] + Jb[3] * S.ustar[3] + Jb[4] * S.ustar[4] + Jb[5] * S.ustar[5];
  float jl[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // velocity of the contact point per unit joint rate: s x (P - o) = s x rr + (d x s), rr and d relative to the base COM
    const LinkCache& L = S.ph.sub.dyn.lc[3 * leg + k];
    const float cr[3] = {fmaf(L.s[1], rr[2], fmaf(-L.s[2], rr[1], L.sv[0])), fmaf(L.s[2], rr[0], fmaf(-L.s[0], rr[2], L.sv[1])),
                         fmaf(L.s[0], rr[1], fmaf(-L.s[1], rr[0], L.sv[2]))};
    if (k == 0) { G.c00 = cr[0]; G.c01 = cr[1]; G.c02 = cr[2]; }
    else if (k == 1) { G.c10 = cr[0]; G.c11 = cr[1]; G.c12 = cr[2]; }
    else { G.c20 = cr[0]; G.c21 = cr[1]; G.c22 = cr[2]; }
    jl[k] = dir[0] * cr[0] + dir[1] * cr[1] + dir[2] * cr[2];
    rel += jl[k] * S.ustar[6 + 3 * leg + k];
  }
  G.rr0 = rr[0]; G.rr1 = rr[1]; G.rr2 = rr[2];   // only the contact lanes' geometry is ever read (dpp_contact_triplet<4 + g>)
  // the row of this lane
  const bool active = enable && (knee ? fr > 0.0f : have);
#pragma unroll
  for (int i = 0; i < 6; i++) R.Jb[i] = knee ? 0.0f : Jb[i];
  R.jl[0] = knee ? 0.0f : jl[0]; R.jl[1] = knee ? 0.0f : jl[1]; R.jl[2] = knee ? 1.0f : jl[2];
  // right-hand side: knee motor -u*; friction -rel; normal -rel - dist / dt (open) or -rel - dist erp / dt (penetrating; a toe may be soft)
  const float kpen = normal ? (dist > 0.0f ? inv_dt : (shank ? erp_dt : erp_m)) : 0.0f;     // one multiply-add, as in the generic form
  float rhs = knee ? -uknee : -rel - dist * kpen;
  // friction anchor: positionalError = -distance * frictionERP / dt along the friction direction (setupMultiBodyContactConstraint)
  if constexpr (ANCHOR) rhs -= (d == 1 ? drift_x : (d == 2 ? drift_y : 0.0f)) * (cfg.friction_erp * inv_dt);
  R.rhs = active ? rhs : 0.0f;
  R.cfm = (normal && !shank) ? cfm_m : 0.0f;
  R.nrm_slot = fric ? 16 + leg : -1;
  R.warm = knee ? -1 : 3 * leg + d;
  R.active = active;
  R.jdi = 0.0f; R.lam = 0.0f; R.w = 0.0f; R.lam_n = 0.0f;
  // bounds as (constant part) + mu * lambda_normal: friction rows have a zero constant part, the others mu = 0; an inactive row is
  // pinned to zero
  const float hi = knee ? fr * dt : 1e30f, lo = knee ? -fr * dt : 0.0f;
  R.mu_e = (active && fric) ? mu_s * cfg.plane_friction : 0.0f;
  R.hi_c = (active && !fric) ? hi : 0.0f;
  R.lo_c = (active && !fric) ? lo : 0.0f;
}

// impulse response M^-1 J^T of the row (what btMultiBody::calcAccelerationDeltasMultiDof returns) -> W[slot]; 1/diag;
// warm start.  Block form, see leg_dynamics: da0 = A0^-1 (Jb - T_L jl); dqdd_L = H_L^-1 jl - T_L^T da0; dqdd_K = -T_K^T da0.
// The 6-vector algebra runs on PACKED float pairs (v_pk_fma_f32 / v_pk_mul_f32: two multiply-adds per issued instruction): rows
// of T and of A0^-1 come out of LDS as aligned pairs, scalar factors ride as op_sel broadcasts.  A0^-1 is symmetric, so its
// row k doubles as column k and a pair of outputs needs no horizontal add.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void row_response(Shared& S, const orr_config& cfg, Row& R, int slot, const BaseFactor& BF) {
  const int leg = R.leg;
  const LegSolve& QL = S.leg[leg];
  float a0[6], mq[12];
  v2f fb2[3], a02[3];
  // warm-start impulse of the row: loaded HERE, pinned behind the base solve (left to the compiler the load sits
  // right in front of its use at the end of the function, behind the W stores, and a lone wave waits out the LDS round trip there:
  // 4096 robots 0.2256 -> 0.2252 ms, 8192 robots neutral)
  float prev = S.s[O(LAMBDA) + (R.warm >= 0 ? R.warm : 0)];   // every lane loads (opaque, so that the load is not put behind a branch)
  // the row's Jacobian through opaque copies: otherwise the compiler fuses neighbouring fields of the Row struct into vector loads
  // for the packed operations, which stops it from keeping the struct in registers (the fields went to LDS via promote-alloca)
  float jl0 = R.jl[0], jl1 = R.jl[1], jl2 = R.jl[2];
  asm("" : "+v"(jl0), "+v"(jl1), "+v"(jl2));
  {
    const v2f* TL = reinterpret_cast<const v2f*>(&QL.T[0][0]);   // TL[3 k + p] = (T[k][2p], T[k][2p+1])
#pragma unroll
    for (int p = 0; p < 3; p++) {
      float jb0 = R.Jb[2 * p], jb1 = R.Jb[2 * p + 1];
      asm("" : "+v"(jb0), "+v"(jb1));
      const v2f jb = {jb0, jb1};
      fb2[p] = jb - (TL[p] * jl0 + TL[3 + p] * jl1 + TL[6 + p] * jl2);
    }
    chol6_solve_pk(BF.F, fb2[0].x, fb2[0].y, fb2[1], fb2[2], a0);
#pragma unroll
    for (int p = 0; p < 3; p++) a02[p] = v2f{a0[2 * p], a0[2 * p + 1]};
  }
  asm volatile("" : "+v"(prev));
  const float h0 = QL.Hi[0] * jl0 + QL.Hi[3] * jl1 + QL.Hi[4] * jl2;
  const float h1 = QL.Hi[3] * jl0 + QL.Hi[1] * jl1 + QL.Hi[5] * jl2;
  const float h2 = QL.Hi[4] * jl0 + QL.Hi[5] * jl1 + QL.Hi[2] * jl2;
  // round 4, v40.  (a) The diagonal J W = Jb . a0 + jl . mq_L without the own-leg selects: with fb = Jb - T_L jl (above) and
  // mq_L = H_L^-1 jl - T_L^T a0 it is jl . (H_L^-1 jl) + fb . a0 -- nine multiply-adds (three of them on pairs) instead of 26 instructions.
  // (b) "H_L^-1 jl for the own leg, 0 for the others" as a multiply-add with a 0 / 1 factor per leg instead of twelve selects (kOwnLegFactor:
  // the one-wave unit only).
  const v2f dg = fb2[0] * a02[0] + fb2[1] * a02[1] + fb2[2] * a02[2];    // first: fb dies here
  float dgs = dg.x + dg.y;
  asm volatile("" : "+v"(dgs));
  const float own0 = leg == 0 ? 1.0f : 0.0f, own1 = leg == 1 ? 1.0f : 0.0f, own2 = leg == 2 ? 1.0f : 0.0f, own3 = leg == 3 ? 1.0f : 0.0f;
#pragma unroll
  for (int L4 = 0; L4 < 4; L4++) {
    const v2f* TK = reinterpret_cast<const v2f*>(&S.leg[L4].T[0][0]);
    const float own = L4 == 0 ? own0 : (L4 == 1 ? own1 : (L4 == 2 ? own2 : own3));
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const v2f t2 = TK[3 * k] * a02[0] + TK[3 * k + 1] * a02[1] + TK[3 * k + 2] * a02[2];
      const float hk = k == 0 ? h0 : (k == 1 ? h1 : h2);
      mq[3 * L4 + k] = kOwnLegFactor ? fmaf(own, hk, -(t2.x + t2.y)) : ((L4 == leg ? hk : 0.0f) - (t2.x + t2.y));   // the same value either way
    }
  }
  const float diag = fmaf(jl0, h0, fmaf(jl1, h1, fmaf(jl2, h2, dgs)));
  // an inactive row stores too (zeros: its Jacobian is zero; its impulse stays zero anyway)
#pragma unroll
  for (int i = 0; i < 6; i++) S.ph.sub.W[slot][i] = a0[i];
#pragma unroll
  for (int i = 0; i < 12; i++) S.ph.sub.W[slot][6 + i] = mq[i];
#pragma unroll
  for (int i = 0; i < 6; i++) R.wa[i] = a0[i];
#pragma unroll
  for (int i = 0; i < 12; i++) R.wq[i] = mq[i];
  // soft row: lambda' = lambda + (rhs - (A lambda)_row - cfm lambda) / (A_rr + cfm) (m_rhs - appliedImpulse m_cfm - deltaVel jacDiagABInv,
  // btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric).  In the Delassus form of the sweeps (diagonal entry of Ac zero) that
  // is the rigid update with 1 / (A_rr + cfm) for 1 / A_rr; only the warm start's (A lambda) needs the extra cfm lambda
  R.jdi = R.active ? __builtin_amdgcn_rcpf(diag + R.cfm) : 0.0f;
  R.rhs *= R.jdi;
  {
    asm volatile("" : "+v"(prev));
    R.lam = (R.active && R.warm >= 0) ? cfg.warmstart_factor * prev : 0.0f;
    R.w = R.cfm * R.lam;
  }
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// f(row) for the joint-limit rows 4..15 whose bit is set in mask, tested leg by leg first (rows 4 + 3 g .. 6 + 3 g belong to leg g):
// a skipped row is a TAKEN branch, 25-30 ticks for a lone wave, and usually one or two of the twelve rows are active -- testing the
// leg's three bits together makes that 5 taken branches instead of 11
template <class F>
__device__ __forceinline__ void for_active_limit_rows(unsigned int mask, F&& f) {
  static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    if (mask & (0x70u << (3 * g))) {
      static_for<4 + 3 * g, 7 + 3 * g>([&](auto rc) __attribute__((always_inline)) {
        if ((mask >> decltype(rc)::value) & 1u) f(rc);
      });
    }
  });
}
// The Gauss-Seidel sweeps over the row slots in solve order (btMultiBodyConstraintSolver::solveSingleIteration),
// Delassus form.  A row lane keeps y = lambda + (rhs - (A lambda)) / diag of its row (the unclamped Gauss-Seidel value),
// and EVERY lane keeps the impulses of all rows (lam[r], equal in all lanes of the robot).  One row update is
//   y' = y - Ac[r] lam[r];   lam[r] = broadcast_from_lane_of_r(clamp(y, lo, hi));   y = y' + Ac[r] lam[r]
// with Ac[r] = -A[row][r] / diag(row) off the diagonal and 0 on it (y of the updated row does not move): four vector
// instructions (fma, v_med3, v_mov_dpp row_newbcast, fma), three of them on the dependent chain; for the unilateral rows
// (contact normals, joint limits: bounds [0, inf)) clamp and broadcast are ONE instruction (v_max_f32_dpp), two on the chain.
// HAS_B: some robot of the wave has an active joint-limit row (bank B is swept too).
// Knee and contact rows are swept unconditionally (a row visited for a robot where it is inactive is a no-op: its
// bounds, 1/diag, lambda and Delassus column are zero); measured 9 % faster than one scalar branch per leg, which also
// stopped the scheduler from overlapping consecutive row updates.
//
// Without the joint-limit bank (HAS_B false: the common case) the whole sweep loop is ONE hand-scheduled block, because a lone wave
// per SIMD pays an issue slot for every instruction, wait states included, and "VALU write -> DPP read of that register" needs two
// of them.  Per bounded row (knee, friction; their bounds are symmetric, -hi .. hi):
//     m = med3(y, -hi, hi);  y -= Ac[r] lam[r];  <filler>;  y += Ac[r] * m(lane of r)      (v_fmac_f32_dpp: broadcast fused)
// and lam[r] = m(lane of r) (v_mov_b32_dpp) is deferred by one row, where it is the filler; the loop counter's s_sub / s_cmp fill
// the two rows that have no deferred broadcast in front of them.  Per contact normal:
//     t = y - Ac[r] lam[r];  lam[r] = max(y(lane of r), 0)  (v_max_f32_dpp);  t += Ac[r] lam[r];  hi (+)= mun[g] lam[r]
// (the friction bound hi = hi_c + sum_g mun[g] lam[16+g] is rebuilt from the fresh normal impulses: mun[g] = mu in the friction
// lanes of toe g, else 0).  67 issue slots per sweep, no idle ones (the compiler's schedule of the generic form below: 110).
__device__ __forceinline__ void pgs_sweeps_bank_a(int iters, const Row& A, const float (&Ac)[kMaxRows], float (&lam)[kMaxRows]) {
  static_assert(kRPW == 4, "row_newbcast needs 16 lanes per robot");
  float y = A.lam + fmaf(-A.w, A.jdi, A.rhs);
  float hi = fmaf(A.mu_e, A.lam_n, A.hi_c);
  const float hic = A.hi_c;
  float mun0 = A.nrm_slot == 16 ? A.mu_e : 0.0f, mun1 = A.nrm_slot == 17 ? A.mu_e : 0.0f;
  float mun2 = A.nrm_slot == 18 ? A.mu_e : 0.0f, mun3 = A.nrm_slot == 19 ? A.mu_e : 0.0f;
  float zero = 0.0f, t0, m0, m1;
  int it = __builtin_amdgcn_readfirstlane(iters);
  if (it <= 0) return;
#define ORR_RB(S) " row_newbcast:" #S " row_mask:0xf bank_mask:0xf\n\t"
#define ORR_ROW(M, R, FILL, S)                                      \
  "v_med3_f32 %[" #M "], %[y], -%[hi], %[hi]\n\t"                   \
  "v_fma_f32 %[y], -%[a" #R "], %[l" #R "], %[y]\n\t"               \
  FILL                                                              \
  "v_fmac_f32_dpp %[y], %[" #M "], %[a" #R "]" ORR_RB(S)
#define ORR_LAM(R, M, S) "v_mov_b32_dpp %[l" #R "], %[" #M "]" ORR_RB(S)
#define ORR_NRM(TO, FROM, R, S, HI)                                 \
  "v_fma_f32 %[" #TO "], -%[a" #R "], %[l" #R "], %[" #FROM "]\n\t" \
  HI                                                                \
  "v_max_f32_dpp %[l" #R "], %[" #FROM "], %[zero]" ORR_RB(S)       \
  "v_fmac_f32 %[" #TO "], %[a" #R "], %[l" #R "]\n\t"
  // one sweep; F0 / F20: fillers of rows 0 and 20 (the rows without a deferred broadcast of this sweep in front of them), TAIL: the
  // broadcast of row 27's impulse, or nothing when the next sweep's row 0 takes it as its filler
#define ORR_SWEEP(F0, F20, TAIL)                                    \
      ORR_ROW(m0, 0, F0, 0)                                         \
      ORR_ROW(m1, 1, ORR_LAM(0, m0, 0), 1)                          \
      ORR_ROW(m0, 2, ORR_LAM(1, m1, 1), 2)                          \
      ORR_ROW(m1, 3, ORR_LAM(2, m0, 2), 3)                          \
      ORR_NRM(t0, y, 16, 4, ORR_LAM(3, m1, 3))                      \
      ORR_NRM(y, t0, 17, 5, "v_fma_f32 %[hi], %[mun0], %[l16], %[hic]\n\t") \
      ORR_NRM(t0, y, 18, 6, "v_fmac_f32 %[hi], %[mun1], %[l17]\n\t") \
      ORR_NRM(y, t0, 19, 7, "v_fmac_f32 %[hi], %[mun2], %[l18]\n\t") \
      "v_fmac_f32 %[hi], %[mun3], %[l19]\n\t"                      \
      ORR_ROW(m0, 20, F20, 8)                                       \
      ORR_ROW(m1, 21, ORR_LAM(20, m0, 8), 9)                        \
      ORR_ROW(m0, 22, ORR_LAM(21, m1, 9), 10)                       \
      ORR_ROW(m1, 23, ORR_LAM(22, m0, 10), 11)                      \
      ORR_ROW(m0, 24, ORR_LAM(23, m1, 11), 12)                      \
      ORR_ROW(m1, 25, ORR_LAM(24, m0, 12), 13)                      \
      ORR_ROW(m0, 26, ORR_LAM(25, m1, 13), 14)                      \
      ORR_ROW(m1, 27, ORR_LAM(26, m0, 14), 15)                      \
      TAIL
  // three sweeps per loop iteration (a taken branch costs a lone wave as much as five multiply-adds), then the remaining ones singly
  asm("v_mov_b32 %[zero], 0\n\t"
      "s_cmp_lt_u32 %[it], 3\n\t"
      "s_cbranch_scc1 3f\n"
      ".p2align 3\n\t.if ((" ORR_STR(ORR_PARITY) ") >> 12) & 1\n\ts_nop 0\n\t.endif\n"
      "1:\n\t"
      ORR_SWEEP("s_sub_u32 %[it], %[it], 3\n\t", "s_cmp_ge_u32 %[it], 3\n\t", "")
      ORR_SWEEP(ORR_LAM(27, m1, 15), "s_nop 0\n\t", "")
      ORR_SWEEP(ORR_LAM(27, m1, 15), "s_nop 0\n\t", ORR_LAM(27, m1, 15))
      "s_cbranch_scc1 1b\n"
      "3:\n\t"
      "s_cmp_eq_u32 %[it], 0\n\t"
      "s_cbranch_scc1 4f\n"
      "2:\n\t"
      ORR_SWEEP("s_sub_u32 %[it], %[it], 1\n\t", "s_cmp_eq_u32 %[it], 0\n\t", ORR_LAM(27, m1, 15))
      "s_cbranch_scc0 2b\n"
      "4:"
      : [y] "+v"(y), [hi] "+v"(hi), [it] "+s"(it), [zero] "=&v"(zero), [t0] "=&v"(t0), [m0] "=&v"(m0), [m1] "=&v"(m1),
        [l0] "+v"(lam[0]), [l1] "+v"(lam[1]), [l2] "+v"(lam[2]), [l3] "+v"(lam[3]),
        [l16] "+v"(lam[16]), [l17] "+v"(lam[17]), [l18] "+v"(lam[18]), [l19] "+v"(lam[19]),
        [l20] "+v"(lam[20]), [l21] "+v"(lam[21]), [l22] "+v"(lam[22]), [l23] "+v"(lam[23]),
        [l24] "+v"(lam[24]), [l25] "+v"(lam[25]), [l26] "+v"(lam[26]), [l27] "+v"(lam[27])
      : [hic] "v"(hic), [mun0] "v"(mun0), [mun1] "v"(mun1), [mun2] "v"(mun2), [mun3] "v"(mun3),
        [a0] "v"(Ac[0]), [a1] "v"(Ac[1]), [a2] "v"(Ac[2]), [a3] "v"(Ac[3]),
        [a16] "v"(Ac[16]), [a17] "v"(Ac[17]), [a18] "v"(Ac[18]), [a19] "v"(Ac[19]),
        [a20] "v"(Ac[20]), [a21] "v"(Ac[21]), [a22] "v"(Ac[22]), [a23] "v"(Ac[23]),
        [a24] "v"(Ac[24]), [a25] "v"(Ac[25]), [a26] "v"(Ac[26]), [a27] "v"(Ac[27])
      : "scc");
#undef ORR_RB
#undef ORR_ROW
#undef ORR_LAM
#undef ORR_NRM
#undef ORR_SWEEP
}

// The same with the joint-limit bank: every row update also moves yB (bank B's unclamped values), and the joint-limit rows
// (unilateral, bank B) are skipped by a scalar bit test unless some robot of the wave has that limit active.
// 6 issue slots per knee / contact row, 2 + 5 per joint-limit row (2 if skipped).
__device__ __forceinline__ void pgs_sweeps_bank_ab(int iters, unsigned int mask, const Row& A, const Row& B, const float (&Aa)[kMaxRows],
                                                   const float (&Ab)[kMaxRows], float (&lam)[kMaxRows]) {
  static_assert(kRPW == 4, "row_newbcast needs 16 lanes per robot");
  float ya = A.lam + fmaf(-A.w, A.jdi, A.rhs), yb = B.lam + fmaf(-B.w, B.jdi, B.rhs);
  float hi = fmaf(A.mu_e, A.lam_n, A.hi_c);
  const float hic = A.hi_c;
  float mun0 = A.nrm_slot == 16 ? A.mu_e : 0.0f, mun1 = A.nrm_slot == 17 ? A.mu_e : 0.0f;
  float mun2 = A.nrm_slot == 18 ? A.mu_e : 0.0f, mun3 = A.nrm_slot == 19 ? A.mu_e : 0.0f;
  float zero = 0.0f, t0, m;
  int it = __builtin_amdgcn_readfirstlane(iters), stmp;
  const unsigned int msk = __builtin_amdgcn_readfirstlane(mask);
  if (it <= 0) return;
#define ORR_RB(S) " row_newbcast:" #S " row_mask:0xf bank_mask:0xf\n\t"
#define ORR_ROW(R, S)                                               \
  "v_med3_f32 %[m], %[ya], -%[hi], %[hi]\n\t"                       \
  "v_fma_f32 %[ya], -%[a" #R "], %[l" #R "], %[ya]\n\t"             \
  "v_fma_f32 %[yb], -%[b" #R "], %[l" #R "], %[yb]\n\t"             \
  "v_fmac_f32_dpp %[ya], %[m], %[a" #R "]" ORR_RB(S)                \
  "v_fmac_f32_dpp %[yb], %[m], %[b" #R "]" ORR_RB(S)                \
  "v_mov_b32_dpp %[l" #R "], %[m]" ORR_RB(S)
#define ORR_NRM(R, S, HI)                                           \
  "v_fma_f32 %[t0], -%[a" #R "], %[l" #R "], %[ya]\n\t"             \
  "v_fma_f32 %[yb], -%[b" #R "], %[l" #R "], %[yb]\n\t"             \
  "v_max_f32_dpp %[l" #R "], %[ya], %[zero]" ORR_RB(S)              \
  "v_fma_f32 %[ya], %[a" #R "], %[l" #R "], %[t0]\n\t"              \
  "v_fmac_f32 %[yb], %[b" #R "], %[l" #R "]\n\t"                    \
  HI
#define ORR_LIM(R)                                                  \
  "s_bitcmp1_b32 %[msk], " #R "\n\t"                                \
  "s_cbranch_scc0 2f\n\t"                                           \
  "v_fma_f32 %[ya], -%[a" #R "], %[l" #R "], %[ya]\n\t"             \
  "v_fma_f32 %[t0], -%[b" #R "], %[l" #R "], %[yb]\n\t"             \
  "v_max_f32_dpp %[l" #R "], %[yb], %[zero]" ORR_RB(R)              \
  "v_fma_f32 %[yb], %[b" #R "], %[l" #R "], %[t0]\n\t"              \
  "v_fmac_f32 %[ya], %[a" #R "], %[l" #R "]\n"                       \
  "2:\n\t"
  asm("v_mov_b32 %[zero], 0\n"
      ".p2align 3\n\t.if ((" ORR_STR(ORR_PARITY) ") >> 13) & 1\n\ts_nop 0\n\t.endif\n"
      "1:\n\t"
      ORR_ROW(0, 0) ORR_ROW(1, 1) ORR_ROW(2, 2) ORR_ROW(3, 3)
      "s_and_b32 %[tmp], %[msk], 0x70\n\ts_cbranch_scc0 3f\n\t" ORR_LIM(4) ORR_LIM(5) ORR_LIM(6) "3:\n\t"            // leg by leg first
      "s_and_b32 %[tmp], %[msk], 0x380\n\ts_cbranch_scc0 3f\n\t" ORR_LIM(7) ORR_LIM(8) ORR_LIM(9) "3:\n\t"
      "s_and_b32 %[tmp], %[msk], 0x1c00\n\ts_cbranch_scc0 3f\n\t" ORR_LIM(10) ORR_LIM(11) ORR_LIM(12) "3:\n\t"
      "s_and_b32 %[tmp], %[msk], 0xe000\n\ts_cbranch_scc0 3f\n\t" ORR_LIM(13) ORR_LIM(14) ORR_LIM(15) "3:\n\t"
      ORR_NRM(16, 4, "v_fma_f32 %[hi], %[mun0], %[l16], %[hic]\n\t")
      ORR_NRM(17, 5, "v_fmac_f32 %[hi], %[mun1], %[l17]\n\t")
      ORR_NRM(18, 6, "v_fmac_f32 %[hi], %[mun2], %[l18]\n\t")
      ORR_NRM(19, 7, "v_fmac_f32 %[hi], %[mun3], %[l19]\n\t")
      ORR_ROW(20, 8) ORR_ROW(21, 9) ORR_ROW(22, 10) ORR_ROW(23, 11)
      ORR_ROW(24, 12) ORR_ROW(25, 13) ORR_ROW(26, 14) ORR_ROW(27, 15)
      "s_sub_u32 %[it], %[it], 1\n\t"
      "s_cmp_lg_u32 %[it], 0\n\t"
      "s_cbranch_scc1 1b"
      : [ya] "+v"(ya), [yb] "+v"(yb), [hi] "+v"(hi), [it] "+s"(it), [zero] "=&v"(zero), [t0] "=&v"(t0), [m] "=&v"(m), [tmp] "=&s"(stmp),
        [l0] "+v"(lam[0]), [l1] "+v"(lam[1]), [l2] "+v"(lam[2]), [l3] "+v"(lam[3]), [l4] "+v"(lam[4]), [l5] "+v"(lam[5]),
        [l6] "+v"(lam[6]), [l7] "+v"(lam[7]), [l8] "+v"(lam[8]), [l9] "+v"(lam[9]), [l10] "+v"(lam[10]), [l11] "+v"(lam[11]),
        [l12] "+v"(lam[12]), [l13] "+v"(lam[13]), [l14] "+v"(lam[14]), [l15] "+v"(lam[15]),
        [l16] "+v"(lam[16]), [l17] "+v"(lam[17]), [l18] "+v"(lam[18]), [l19] "+v"(lam[19]),
        [l20] "+v"(lam[20]), [l21] "+v"(lam[21]), [l22] "+v"(lam[22]), [l23] "+v"(lam[23]),
        [l24] "+v"(lam[24]), [l25] "+v"(lam[25]), [l26] "+v"(lam[26]), [l27] "+v"(lam[27])
      : [hic] "v"(hic), [mun0] "v"(mun0), [mun1] "v"(mun1), [mun2] "v"(mun2), [mun3] "v"(mun3), [msk] "s"(msk),
        [a0] "v"(Aa[0]), [a1] "v"(Aa[1]), [a2] "v"(Aa[2]), [a3] "v"(Aa[3]), [a4] "v"(Aa[4]), [a5] "v"(Aa[5]), [a6] "v"(Aa[6]),
        [a7] "v"(Aa[7]), [a8] "v"(Aa[8]), [a9] "v"(Aa[9]), [a10] "v"(Aa[10]), [a11] "v"(Aa[11]), [a12] "v"(Aa[12]), [a13] "v"(Aa[13]),
        [a14] "v"(Aa[14]), [a15] "v"(Aa[15]), [a16] "v"(Aa[16]), [a17] "v"(Aa[17]), [a18] "v"(Aa[18]), [a19] "v"(Aa[19]),
        [a20] "v"(Aa[20]), [a21] "v"(Aa[21]), [a22] "v"(Aa[22]), [a23] "v"(Aa[23]), [a24] "v"(Aa[24]), [a25] "v"(Aa[25]),
        [a26] "v"(Aa[26]), [a27] "v"(Aa[27]),
        [b0] "v"(Ab[0]), [b1] "v"(Ab[1]), [b2] "v"(Ab[2]), [b3] "v"(Ab[3]), [b4] "v"(Ab[4]), [b5] "v"(Ab[5]), [b6] "v"(Ab[6]),
        [b7] "v"(Ab[7]), [b8] "v"(Ab[8]), [b9] "v"(Ab[9]), [b10] "v"(Ab[10]), [b11] "v"(Ab[11]), [b12] "v"(Ab[12]), [b13] "v"(Ab[13]),
        [b14] "v"(Ab[14]), [b15] "v"(Ab[15]), [b16] "v"(Ab[16]), [b17] "v"(Ab[17]), [b18] "v"(Ab[18]), [b19] "v"(Ab[19]),
        [b20] "v"(Ab[20]), [b21] "v"(Ab[21]), [b22] "v"(Ab[22]), [b23] "v"(Ab[23]), [b24] "v"(Ab[24]), [b25] "v"(Ab[25]),
        [b26] "v"(Ab[26]), [b27] "v"(Ab[27])
      : "scc");
#undef ORR_RB
#undef ORR_ROW
#undef ORR_NRM
#undef ORR_LIM
}

template <bool HAS_B>
__device__ __forceinline__ void pgs_sweeps(int iters, unsigned int mask, int lane, int sub, Row& A, Row& B,
                                           const float (&AcA)[kMaxRows], const float (&AcB)[kMaxRows], float (&lam)[kMaxRows]) {
#ifndef ORR_GENERIC_PGS
  if constexpr (!HAS_B) pgs_sweeps_bank_a(iters, A, AcA, lam);
  else pgs_sweeps_bank_ab(iters, mask, A, B, AcA, AcB, lam);
  return;
#endif
  float yA = A.lam + fmaf(-A.w, A.jdi, A.rhs), yB = B.lam + fmaf(-B.w, B.jdi, B.rhs);
  float zero;
  asm("v_mov_b32 %0, 0" : "=v"(zero));   // a VGPR operand for the DPP max (opaque, so that it stays in a register)
  float mun[4];  // friction rows: d(bound) / d(normal impulse of their toe)
#pragma unroll
  for (int g = 0; g < 4; g++) mun[g] = A.nrm_slot == 16 + g ? A.mu_e : 0.0f;
  float hiE = fmaf(A.mu_e, A.lam_n, A.hi_c), loE = fmaf(-A.mu_e, A.lam_n, A.lo_c);
  for (int it = 0; it < iters; it++) {
    auto rowA = [&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value, src = r < 4 ? r : r - 12;
      const float old = lam[r];
      const float yp = fmaf(-AcA[r], old, yA);
      float sb;
      if constexpr (r >= 16 && r < 20) sb = dpp_bcast_max0<src>(yA, zero);   // normal row: [0, inf); an inactive one has y == 0
      else sb = bcast_lane<src>(__builtin_amdgcn_fmed3f(yA, loE, hiE), sub);
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
      for_active_limit_rows(mask, [&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const float old = lam[r];
        const float yp = fmaf(-AcB[r], old, yB);
        const float sb = dpp_bcast_max0<r>(yB, zero);                      // joint limit: [0, inf)
        yB = fmaf(AcB[r], sb, yp);
        lam[r] = sb;
        yA = fmaf(AcA[r], sb - old, yA);
      });
    }
    static_for<16, 28>(rowA);
  }
}

// Delassus columns A[row][r] = J_row . W[r] = J_r . W[row] (M^-1 is symmetric), kept in registers already scaled for the
// sweeps: Ac[r] = -A[row][r] / diag(row), 0 on the diagonal; w = (A lambda) of the warm start; lam[r] = warm-start impulse of
// row r (in every lane).  Every row lane holds its OWN impulse response in registers (Row::wa, wq) and receives the Jacobian
// of row r from that row's lane as DPP operands of the multiply-adds (row_newbcast), so a column costs no LDS traffic:
// knee rows have J = e_knee (the column is an element of wq), joint-limit rows J = +-e_joint (one multiply), contact rows
// the full 9 terms (6 base + the 3 joints of their leg, whose index is static per slot).  Knee rows always, joint-limit
// rows one by one, contact rows per leg.
// x, but 0 in lane K of the robot's 16 (compare + select through vcc, two instructions, no lane mask kept in SGPRs)
template <int K>
__device__ __forceinline__ float zero_in_lane(float x, int lane) {
  float o;
  asm("v_cmp_eq_u32 vcc, %2, %1\n\tv_cndmask_b32_e64 %0, %3, 0, vcc" : "=v"(o) : "v"(lane), "n"(K), "v"(x) : "vcc");
  return o;
}
template <bool HAS_B>
__device__ __forceinline__ void delassus_columns(const Shared& S, unsigned int mask, int lane, int sub, Row& A, Row& B, const ContactGeom& G,
                                                 float (&AcA)[kMaxRows], float (&AcB)[kMaxRows], float (&lam)[kMaxRows]) {
#pragma unroll
  for (int r = 0; r < kMaxRows; r++) { AcA[r] = 0.0f; AcB[r] = 0.0f; lam[r] = 0.0f; }
  // bookkeeping of one column once its entries a (this lane's bank-A row) and b (bank-B row) are known
  auto finish = [&](auto rc, float a, float b) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    constexpr bool inB = r >= 4 && r < 16;
    constexpr int src = inB ? r : (r < 4 ? r : r - 12);
    const float l0 = bcast_lane<src>(inB ? B.lam : A.lam, sub);
    lam[r] = l0;
    A.w += a * l0;
    // the diagonal entry of the scaled column is zero (y of the updated row does not move).  `lane == src` as a compare into vcc right
    // here: as a C++ select the compiler keeps one lane mask per row (16 SGPR pairs) alive over the whole sub-step loop, which is what
    // pushed the SGPR file over its limit (46 spilled, every use reloaded with two v_readlane)
    AcA[r] = inB ? -a * A.jdi : zero_in_lane<src>(-a * A.jdi, lane);
    if (r >= 16 && r < 20 && A.nrm_slot == r) A.lam_n = l0;
    if (HAS_B) {
      B.w += b * l0;
      AcB[r] = inB ? zero_in_lane<src>(-b * B.jdi, lane) : -b * B.jdi;
    }
    asm("" : "+v"(AcA[r]), "+v"(AcB[r]));  // keep the scaled value (do not re-derive it inside the sweeps)
  };
  static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {   // knee friction motor: J = unit vector of the knee joint of leg r
    constexpr int r = decltype(rc)::value;
    finish(rc, A.wq[3 * r + 2], HAS_B ? B.wq[3 * r + 2] : 0.0f);
  });
  if (HAS_B) {
    for_active_limit_rows(mask, [&](auto rc) __attribute__((always_inline)) {   // joint limit: J = +-unit vector of joint r - 4
      constexpr int r = decltype(rc)::value;
      const float sg = bcast_lane<r>(B.jl[(r - 4) % 3], sub);
      finish(rc, sg * A.wq[r - 4], sg * B.wq[r - 4]);
    });
  }
  // The contact rows leg by leg, each behind a test of the wave's union mask.  Round 4 measured the alternative (all four legs unconditionally, no zero-initialisation of the columns a skipped leg leaves - a leg is in contact in 60 % of the
  // sub-steps, so the union over four robots misses one in < 3 % of them): 120 instructions fewer in the loop, but 4096 robots
  // 0.2262 -> 0.2261 ms (nothing) and 8192 robots 0.3162 -> 0.3254 ms (+2.9 %: twelve columns in one basic block cost the two-wave build
  // 31 more spilled registers, 10 scratch accesses inside the loop).  Kept off.
  const unsigned int cm = (mask >> 16) & 0xFu;
  static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {      // the three contact rows of leg g at once
    constexpr int g = decltype(gc)::value;
    if ((cm >> g) & 1u) {
      float ax, ay, az, bx = 0.0f, by = 0.0f, bz = 0.0f;
      dpp_contact_triplet<4 + g>(G.rr0, G.rr1, G.rr2, G.c00, G.c01, G.c02, G.c10, G.c11, G.c12, G.c20, G.c21, G.c22, A.wa[0], A.wa[1], A.wa[2],
                                 A.wa[3], A.wa[4], A.wa[5], A.wq[3 * g], A.wq[3 * g + 1], A.wq[3 * g + 2], ax, ay, az);
      if (HAS_B)
        dpp_contact_triplet<4 + g>(G.rr0, G.rr1, G.rr2, G.c00, G.c01, G.c02, G.c10, G.c11, G.c12, G.c20, G.c21, G.c22, B.wa[0], B.wa[1], B.wa[2],
                                   B.wa[3], B.wa[4], B.wa[5], B.wq[3 * g], B.wq[3 * g + 1], B.wq[3 * g + 2], bx, by, bz);
      finish(std::integral_constant<int, 16 + g>{}, az, bz);
      finish(std::integral_constant<int, 20 + 2 * g>{}, ax, bx);
      finish(std::integral_constant<int, 21 + 2 * g>{}, ay, by);
    }
  });
}

// One physics sub-step.  Returns the fall-proxy flag (wave-uniform) when want_fall.
// limit_idle (wave-uniform, kept by the caller across the sub-steps of a launch, 0 at its start): sub-steps for which NO joint of this
// wave can get a joint-limit row.  A joint coordinate moves by at most max_coord_velocity * dt per sub-step (the velocity clamp of the
// integration), so a joint that is m away from the activation distance of its nearer bound stays rowless for floor(m / that) sub-steps;
// the bank-B row setup (12 lanes x ~25 instructions + ballots, every sub-step) is skipped for that long - exactly, not approximately.
template <bool ANCHOR = false>
__device__ static int physics_substep(const KParams& P, Shared& S, const LegConst& K, int lane, int sub, bool want_fall, OwnCoord& X,
                                      int& limit_idle, AnchorState* AS = nullptr, bool anchor_robot = false) {
  const orr_config& cfg = P.cfg;
  const float dt = cfg.sim_dt, inv_dt = 1.0f / cfg.sim_dt, erp_dt = cfg.contact_erp / cfg.sim_dt;
  // termination-only collision proxies (imitation_task.py:536-546): read once per env step, at its last sub-step, from the device table
  // (lane i = proxy i); the loads are issued here and consumed after the leg dynamics, which cover their round trip
  int fp_body = 0, fp_n = 0;
  float fp_x = 0.0f, fp_y = 0.0f, fp_z = 0.0f, fp_r = 0.0f;
  if (want_fall) {
    const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
    static_assert(ORR_MAX_FALL_PROXIES <= kLanes, "one fall proxy per lane");
    fp_body = mc->fall_body[lane]; fp_x = mc->fall_pos[lane][0]; fp_y = mc->fall_pos[lane][1]; fp_z = mc->fall_pos[lane][2]; fp_r = mc->fall_radius[lane];
    fp_n = mc->num_fall;
  }
  BaseFactor BF;
  leg_dynamics(P, S, K, lane, BF);  // -> link poses, leg solves, unconstrained velocities u*
  WSYNC();
  PT(3);
  // the proxy count comes from the device table with the proxies: its first use stays HERE (left alone, the compiler hoists the cheap
  // `lane < fp_n` up to the load and waits out the global-memory round trip on the spot, in front of the leg dynamics)
  asm volatile("" : "+v"(fp_n));
  int fall = 0;
  if (want_fall) {
    bool hit = false;
    if (lane < fp_n) {
      const int b = fp_body;
      const float* Rw = b == 0 ? S.Rb : S.ph.sub.dyn.lc[b - 1].Rw;
      const float oz = b == 0 ? S.s[O(POS) + 2] : S.ph.sub.dyn.lc[b - 1].ow[2];
      const float wz = Rw[6] * fp_x + Rw[7] * fp_y + Rw[8] * fp_z;
      hit = (oz + wz - fp_r) < cfg.contact_margin;
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
  ContactGeom G, Gunused;
  if constexpr (ANCHOR) row_setup_bank_a_anchor(S, cfg, rowlane ? (lane < 4 ? lane : lane + 12) : 0, rowlane, dt, inv_dt, erp_dt, A, G, AS, anchor_robot);
  else {
  row_setup_bank_a(S, cfg, rowlane ? (lane < 4 ? lane : lane + 12) : 0, rowlane, dt, inv_dt, erp_dt, A, G);
  }
  unsigned long long balB = 0ull;
  if (limit_idle > 0) {
    limit_idle--;
    B = Row{};                  // never read: without an active joint-limit row in the wave the bank-B paths below are not taken
    B.active = false;
  } else {
    float margin = 1e30f;
    row_setup<1>(S, cfg, (rowlane && lane >= 4) ? lane : 4, rowlane && lane >= 4, dt, inv_dt, erp_dt, B, Gunused, &margin);
    balB = __ballot(B.active);
    // sub-steps that can be skipped from here: the minimum over the wave's joints of floor(margin / (vmax dt)) - 1 (one sub-step of
    // slack against rounding), found with five ballots (powers of two up to 16 are enough: the check itself is cheap)
    const float steps_f = (rowlane && lane >= 4) ? margin * __builtin_amdgcn_rcpf(cfg.max_coord_velocity * dt) - 1.0f : 1e30f;
    int idle = 0;
    if (__ballot(steps_f < 1.0f) == 0ull) idle = 1;
    if (__ballot(steps_f < 2.0f) == 0ull) idle = 2;
    if (__ballot(steps_f < 4.0f) == 0ull) idle = 4;
    if (__ballot(steps_f < 8.0f) == 0ull) idle = 8;
    if (__ballot(steps_f < 16.0f) == 0ull) idle = 16;
    limit_idle = idle;
  }
  const unsigned long long balA = __ballot(A.active);
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
  row_response(S, cfg, A, rowlane ? (lane < 4 ? lane : lane + 12) : 0, BF);
  if (anyB) row_response(S, cfg, B, (rowlane && lane >= 4) ? lane : kMaxRows, BF);   // lanes without a joint-limit row: dump slot
  WSYNC();
  PT(6);
  // Delassus columns, then the Gauss-Seidel sweeps; two instantiations: with and without the joint-limit bank
  float AcA[kMaxRows], AcB[kMaxRows], lam[kMaxRows];
  if (anyB) {
    delassus_columns<true>(S, mask, lane, sub, A, B, G, AcA, AcB, lam);
    PT(7);
    pgs_sweeps<true>(cfg.solver_iters, mask, lane, sub, A, B, AcA, AcB, lam);
    PT(34);    // sweeps with the joint-limit bank, separately (tools/phase_cycles.py)
  } else {
    delassus_columns<false>(S, mask, lane, sub, A, B, G, AcA, AcB, lam);
    PT(7);
    pgs_sweeps<false>(cfg.solver_iters, mask, lane, sub, A, B, AcA, AcB, lam);
  }
  PT(8);
  // contact impulses are remembered for the next sub-step's warm start (0 for open contacts)
  {  // warm-start slot 3 leg + d: normal (slot 16 + leg), then the two friction rows (20 + 2 leg, 21 + 2 leg); lam[] is the same in
     // every lane of the robot, all of them store it
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
      for_active_limit_rows(mask, add_row);
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
    // sin(h) / |w| and cos(h) by their series: h < 0.2 (orr_create checks max_coord_velocity * sim_dt), exact to float precision there
    const float sc = 0.5f * dt * fmaf(h2, fmaf(h2, fmaf(h2, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
    const float ch = fmaf(h2, fmaf(h2, fmaf(h2, fmaf(h2, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
    const float dq[4] = {w0 * sc, w1 * sc, w2 * sc, ch};
    float qn[4];
    qmul(dq, &S.s[O(QUAT)], qn);
    const float nn = rsq(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    // coordinates: the internal joint angle a = jdir (q - joff) advances by dt v, i.e. q by jdir dt v (jdir = +-1); the owner lane
    // integrates its register copy (OwnCoord) and publishes it.  Every lane stores five words without a divergent `if` (dump slots
    // where it has nothing to store): velocity and coordinate of DOF l, of DOF 16 + l (lanes 0, 1), one quaternion component (lanes 0..3)
    X.x0 = fmaf(dt * v0, K.int_c0, X.x0);
    X.x1 = fmaf(dt * v1, K.int_c1, X.x1);
    float* const dump = &S.ustar[22];
    float* const pv0 = &S.s[lane < 3 ? O(ANGVEL) + lane : (lane < 6 ? O(LINVEL) + lane - 3 : O(QD) + lane - 6)];
    float* const px0 = lane < 3 ? dump : &S.s[lane < 6 ? O(POS) + lane - 3 : O(Q) + lane - 6];
    float* const pv1 = lane < 2 ? &S.s[O(QD) + 10 + lane] : dump;
    float* const px1 = lane < 2 ? &S.s[O(Q) + 10 + lane] : dump;
    float* const pq = lane < 4 ? &S.s[O(QUAT) + lane] : dump;
    WSYNC();
    *pv0 = lane < 6 ? v0 : v0 * K.int_c0;
    *px0 = X.x0;
    *pv1 = v1 * K.int_c1;
    *px1 = X.x1;
    *pq = pick4(lane, qn[0], qn[1], qn[2], qn[3]) * nn;
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
    const float nn = rsq(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
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
