// orr_task.h -- motion clips, imitation reward / observation / termination, sensors, reset (ImitationTask, sensors, Minitaur.reset)
// (device code of libopenroborl_hip.so, included by orr_kernels.hip after orr_device.h; see DESIGN.md sections 3-5)
#pragma once

// ================================================================================================
// reference-motion sampling (task/motion_data.py:417-509,591-633,682-718)
// ================================================================================================
struct Sample {
  int f0, f1, count;
  float blend, phase;
};
__device__ __forceinline__ double clip_phase_d(const DevClip& c, double t) {  // motion_data.py:210-232
  double ph = t / c.dur_d;
  if (c.flags & ORR_CLIP_WRAP) ph -= floor(ph);
  else ph = fmin(fmax(ph, 0.0), 1.0);
  return ph;
}
__device__ __forceinline__ float clip_phase(const DevClip& c, double t) { return (float)clip_phase_d(c, t); }
__device__ __forceinline__ Sample clip_index(const DevClip& c, double t) {  // motion_data.py:234-253,682-718
  Sample s;
  const bool wrap = c.flags & ORR_CLIP_WRAP;
  s.count = (int)floor(t / c.dur_d);
  if (!wrap) s.count = s.count < 0 ? 0 : (s.count > 1 ? 1 : s.count);
  const double ph = clip_phase_d(c, t);
  s.phase = (float)ph;
  if (!wrap && t <= 0.0) { s.f0 = 0; s.f1 = 0; s.blend = 0.0f; }
  else if (!wrap && t >= c.dur_d) { s.f0 = c.F - 1; s.f1 = c.F - 1; s.blend = 0.0f; }
  else {
    s.f0 = (int)(ph * (c.F - 1));
    s.f0 = s.f0 > c.F - 1 ? c.F - 1 : s.f0;
    s.f1 = s.f0 + 1 < c.F - 1 ? s.f0 + 1 : c.F - 1;
    const double nt = ph * c.dur_d, t0 = s.f0 * c.dt_d, t1 = s.f1 * c.dt_d;
    s.blend = s.f1 == s.f0 ? 0.0f : (float)((nt - t0) / (t1 - t0));
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
// Two halves, so that the round trip of the frame loads can be covered by independent work of the caller: sample_poses_issue
// computes the frame indices and issues the loads (into registers), sample_poses_finish stages and blends them.
struct PoseLoads {
  float lo[11], hi[11], v0[2], v1[2];   // frame rows: word `lane` and, for lanes 0..2, word 16 + lane; frame velocities of lane 0's time
  Sample sm;
};
__device__ static void sample_poses_issue(const KParams& P, Shared& S, int lane, double t_lane, PoseLoads& L, int pt_slot = 32) {
  constexpr int nt = 5;   // update time + the four target times (compile-time: the staging arrays below must stay in registers)
  const DevClip& c = S.clip;
  const Sample sm = clip_index(c, lane < nt ? t_lane : 0.0);
  L.sm = sm;
  if (lane < nt) { S.ph.end.red[2 * lane] = __int_as_float(sm.f0); S.ph.end.red[2 * lane + 1] = __int_as_float(sm.f1); }
  WSYNC();
  PT(pt_slot);
  {
    // The clip pointers come out of a table, i.e. as generic pointers: loads through them would be FLAT instructions, which also
    // count on the LDS counter, so every LDS access in between would wait for the previous frame to arrive (13 serialised round
    // trips).  They point to global memory (orr_set_motion takes device pointers): say so, fetch all rows at once (frame row = 19
    // words: lane i and, for lanes 0..2, word 16 + i).
    typedef const float __attribute__((address_space(1))) * gptr;
    const gptr frames = (gptr)c.frames, vels = (gptr)c.vels;
    constexpr int kMaxE = 2 * nt;
    const int w1 = lane < 3 ? 16 + lane : lane;
#pragma unroll
    for (int e = 0; e < kMaxE; e++) {
      const int f = __float_as_int(S.ph.end.red[e]);
      L.lo[e] = frames[f * 19 + lane]; L.hi[e] = frames[f * 19 + w1];
    }
    L.lo[kMaxE] = frames[lane]; L.hi[kMaxE] = frames[w1];                       // frame 0 (warm-up heading)
    const int f0 = __float_as_int(S.ph.end.red[0]), f1 = __float_as_int(S.ph.end.red[1]);
    const int w2 = lane < 2 ? 16 + lane : lane;
    L.v0[0] = vels[f0 * 18 + lane]; L.v0[1] = vels[f0 * 18 + w2];
    L.v1[0] = vels[f1 * 18 + lane]; L.v1[1] = vels[f1 * 18 + w2];
  }
  WSYNC();   // red[] may be reused by the caller from here on
}
__device__ static void sample_poses_finish(const KParams& P, Shared& S, int lane, double t_lane, bool with_vel, const PoseLoads& L, int pt_slot = 32) {
  constexpr int nt = 5, kMaxE = 2 * nt;
  const DevClip& c = S.clip;
  const bool warm_ep = geti(S, O(WARMUP)) != 0;
  const Sample sm = L.sm;
#pragma unroll
  for (int e = 0; e <= kMaxE; e++) {
    S.ph.end.frames[e][lane] = L.lo[e];
    if (lane < 3) S.ph.end.frames[e][16 + lane] = L.hi[e];
  }
  if (with_vel) {
    S.ph.end.fvel[0][lane] = L.v0[0]; S.ph.end.fvel[1][lane] = L.v1[0];
    if (lane < 2) { S.ph.end.fvel[0][16 + lane] = L.v0[1]; S.ph.end.fvel[1][16 + lane] = L.v1[1]; }
  }
  WSYNC();
  PT(pt_slot + 1);
  if (lane < nt) {
    const bool warm_pose = warm_ep && t_lane >= -(double)P.cfg.warmup_time && t_lane < 0.0;
    float out[19];
    if (warm_pose) {
      // default pose rotated to the heading of frame(0) (imitation_task.py:985-1009, 1245-1252); warm-up episodes only (cold table)
      const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
      const float ipos[3] = {mc->init_pos[0], mc->init_pos[1], mc->init_pos[2]};
      const float* fr0 = S.ph.end.frames[10];
      float dr[4], pp[3], qq[4], q0[4] = {fr0[3], fr0[4], fr0[5], fr0[6]};
      const float dh = qheading(q0) - qheading(S.m.init_quat);
      q_about_z(dh, dr);
      qrot(ipos, dr, pp);
      qmul(dr, S.m.init_quat, qq);
      out[0] = pp[0]; out[1] = pp[1]; out[2] = pp[2];
      out[3] = qq[0]; out[4] = qq[1]; out[5] = qq[2]; out[6] = qq[3];
#pragma unroll
      for (int i = 0; i < 12; i++) out[7 + i] = mc->default_joints[i];
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
__device__ static void sample_poses(const KParams& P, Shared& S, int lane, double t_lane, bool with_vel, int pt_slot = 32) {
  PoseLoads L;
  sample_poses_issue(P, S, lane, t_lane, L, pt_slot);
  sample_poses_finish(P, S, lane, t_lane, with_vel, L, pt_slot);
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

__device__ __forceinline__ double motion_time(const KParams& P, const Shared& S) {  // imitation_task.py:831-848
  double t = geti(S, O(STATE_ACTION_COUNTER)) * S.clip.sim_dt_d + (double)S.s[O(TIME_OFFSET)];
  if (geti(S, O(WARMUP))) t -= (double)P.cfg.warmup_time;   // 0.25: exact in float32
  return t;
}

// build the 76-d target observation into obs76 (LDS) from S.ph.end.pose[1..4] (already origin-offset) -- imitation_task.py:254-301.
// The control observation S.co must be current (it is after the last sub-step's ring push and after reset_robot's local blend).
__device__ static void target_obs(const KParams& P, const float* rec, Shared& S, int lane, float* obs76) {
  if (kLanes != 16) ctrl_obs(P, rec, S, lane);
  if (lane >= 1 && lane <= 4) {
    float rpy[3];
    euler_from_quat(&S.co[12], rpy);
    // robot.get_base_orientation (minitaur.py:630-638) = quaternion of the delayed rpy; its heading is the
    // direction of the rotated x axis = atan2(sin(yaw) cos(pitch), cos(yaw) cos(pitch))
    float sy, cy, spch, cpch;
    joint_sincos(rpy[1], &spch, &cpch);
    joint_sincos(rpy[2], &sy, &cy);
    const float heading = atan2_bf(sy * cpch, cy * cpch);
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
// eff_replay (parity replay only, else NULL): [2][8][3] link positions that replace the forward kinematics
__device__ static float calc_reward(const KParams& P, Shared& S, int lane, const float* eff_replay = nullptr) {
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
  if (eff_replay) {
    WSYNC();
    for (int i = lane; i < 48; i += kLanes) (&S.ph.end.ee[0][0][0])[i] = eff_replay[i];
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
    S.ph.end.red[lane] = e;
    WSYNC();
#pragma unroll
    for (int k = 0; k < 8; k++) ee_err += S.ph.end.red[k];
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
// uni_replay (parity replay only, else NULL): 28 draws in [0, 1) that replace the Philox stream
__device__ static void reset_robot(const KParams& P, float* rec, Shared& S, int lane, bool valid, long long total_step_count, float* obs,
                                   const float* uni_replay = nullptr) {
  const orr_config& c = P.cfg;
  // every reset starts a new episode = a new RNG stream (robot, episode)
  const uint32_t robot = (uint32_t)geti(S, O(ROBOT_INDEX)), ep = (uint32_t)geti(S, O(EPISODE_IDX)) + 1u;
  WSYNC();
  if (lane == 0) seti(S, O(EPISODE_IDX), (int)ep);
  // 1-2. default pose at the grid slot, counters, ring, filter (minitaur.py:246-268, 465-483).  The per-motor reset constants come from
  // the cold table: all lanes load (clamped index), no divergent `if` around the loads
  const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
  const int lm = lane < 12 ? lane : 0, l3 = lane < 3 ? lane : 0;
  const int rj = mc->joint_of_motor[lm];
  const float r_q0 = mc->init_motor_angles[lm] + mc->motor_offset[lm], r_p0 = mc->init_pos[l3];
  if (lane < 3) {
    S.s[O(POS) + lane] = r_p0 + (lane < 2 ? S.s[O(GRID_OFFSET) + lane] : 0.0f);
    S.s[O(LINVEL) + lane] = 0.0f; S.s[O(ANGVEL) + lane] = 0.0f;
  }
  if (lane < 4) S.s[O(QUAT) + lane] = S.m.init_quat[lane];
  if (lane < 12) {
    const int j = rj;
    S.s[O(Q) + j] = r_q0;  // no direction factor (minitaur.py:481)
    S.s[O(QD) + j] = 0.0f;
    S.s[O(LAST_ACTION) + lane] = 0.0f; S.s[O(ACTION) + lane] = 0.0f; S.s[O(FILTER_ACTION) + lane] = 0.0f; S.s[O(LAMBDA) + lane] = 0.0f;
    S.s[O(XHIST) + lane] = 0.0f; S.s[O(XHIST) + 12 + lane] = 0.0f; S.s[O(YHIST) + lane] = 0.0f; S.s[O(YHIST) + 12 + lane] = 0.0f;
  }
  if (lane == 0) {
    seti(S, O(RING_LEN), 0); seti(S, O(RING_HEAD), ORR_RING_DEPTH - 1);
    seti(S, O(STATE_ACTION_COUNTER), 0); seti(S, O(STEP_COUNTER), 0); seti(S, O(FILTER_VALID), 0);
    seti(S, O(EP_STEP), 0);   // DONE_REASON keeps the reason the PREVIOUS episode ended with until the next step overwrites it (env.stats())
    S.s[O(EP_RETURN)] = 0.0f;
  }
  WSYNC();
  PT(16);
  // Order of the stages below: what needs global memory is started first (the episode's draws fix the start time, hence the clip
  // frames; the model's mass table), the stages that need nothing from memory run while those loads are in flight.
  // 4a. all 28 draws of the episode (0..25 randomiser, 26 ref-state-init, 27 time offset) = 7 Philox blocks: lane b < 7 evaluates
  // block b once and parks its four numbers in LDS
  float* draws = S.ph.end.red + 24;    // 28 words
  static_assert(kLanes == 16, "reset_robot / sample_poses assume 16 lanes per robot");
  if (lane < 7) {
    float u4[4];
    if (uni_replay) { u4[0] = uni_replay[4 * lane]; u4[1] = uni_replay[4 * lane + 1]; u4[2] = uni_replay[4 * lane + 2]; u4[3] = uni_replay[4 * lane + 3]; }
    else philox_block(c.seed, robot, ep, (uint32_t)lane, u4);
    draws[4 * lane] = u4[0]; draws[4 * lane + 1] = u4[1]; draws[4 * lane + 2] = u4[2]; draws[4 * lane + 3] = u4[3];
  }
  WSYNC();
  PT(24);
  // 5a. task reset (imitation_task.py:183-199, 694-732, 1103-1110): start time -> frame loads issued
  const DevClip& clip = S.clip;
  {
    const float u1 = draws[26], u2 = draws[27];
    const bool ref_init = u1 < c.ref_state_init_prob;
    const bool warm = (!ref_init) && c.warmup_time > 0.0f;
    if (lane == 0) {
      seti(S, O(WARMUP), warm ? 1 : 0);
      S.s[O(TIME_OFFSET)] = warm ? u2 * c.warmup_time : u2 * (float)clip.dur_d;
      S.s[O(ORIGIN_POS)] = 0.0f; S.s[O(ORIGIN_POS) + 1] = 0.0f; S.s[O(ORIGIN_POS) + 2] = 0.0f;
      S.s[O(ORIGIN_ROT)] = 0.0f; S.s[O(ORIGIN_ROT) + 1] = 0.0f; S.s[O(ORIGIN_ROT) + 2] = 0.0f; S.s[O(ORIGIN_ROT) + 3] = 1.0f;
    }
    WSYNC();
  }
  const double t = motion_time(P, S);
  const double step_dt = S.clip.sim_dt_d * c.action_repeat;
  double tl = t;
  {  // lanes 1..4: the four target times.  Selects over the four scalars: indexing the kernel argument with the lane makes the
     // compiler read it from memory with a vector load, whose wait also drains every store and atomic issued before it
    const int k = lane - 1;
    const int steps = (k & 2) ? ((k & 1) ? c.tar_frame_steps[3] : c.tar_frame_steps[2]) : ((k & 1) ? c.tar_frame_steps[1] : c.tar_frame_steps[0]);
    if (lane >= 1 && lane <= 4) tl = t + steps * step_dt;
  }
  PT(20);
  PoseLoads PL;
  sample_poses_issue(P, S, lane, tl, PL, 26);     // uses red[0..9] until it returns
  PT(19);
  // ring entries #1 and #2 of the new episode are kept (LDS) so that the control observations of the reset are blended from them
  // directly: reading the ring back would be a store -> load round trip through memory each time
  float* e1 = S.ph.end.red;            // 20 words each
  receive_obs(P, rec, S, lane, valid, e1);  // ring entry #1
  // 3. sensor histories <- 3 copies of the current readings (minitaur.py:270-271; sensor_wrappers.py:122-129)
  PT(17);
  WSYNC();
  for (int i = lane; i < 19; i += kLanes) S.co[i] = e1[i];   // one entry in the ring: _get_delay_obs returns it (minitaur.py:345-346)
  WSYNC();
  PT(29);
  sensors_push(S, lane, true);
  const float e1_keep[2] = {e1[lane], e1[lane < 3 ? 16 + lane : lane]};   // ring entry #1: words lane and (lanes 0..2) 16 + lane
  PT(18);
  // 4b. randomiser (controllable_env_randomizer_from_config.py:92-122), sorted-name draw order:
  //    inertia 2 | joint friction 8 | latency 1 | lateral friction 1 | mass 2 | motor strength 12
  if (c.flags & ORR_FLAG_RANDOMIZER) {
    for (int i = lane; i < 26; i += kLanes) {
      const float u = draws[i];
      if (i < 2) S.s[O(INERTIA_RATIO) + i] = 0.5f + u * 1.0f;
      else if (i < 10) { if (((i - 2) & 1) == 0) S.s[O(KNEE_FRICTION) + ((i - 2) >> 1)] = u * 0.05f; }
      else if (i == 10) S.s[O(LATENCY)] = u * 0.04f;
      else if (i == 11) S.s[O(FOOT_MU)] = 0.5f + u * 0.75f;
      else if (i < 14) S.s[O(MASS_RATIO) + i - 12] = 0.8f + u * 0.4f;
      else S.s[O(STRENGTH) + i - 14] = 0.8f + u * 0.4f;
    }
    WSYNC();   // the new mass / inertia ratios take effect in the next launch's load_leg_const (no staged mass table any more)
  }
  PT(25);
  // 5b. the reference poses of the start time
  sample_poses_finish(P, S, lane, tl, true, PL, 26);
  PT(21);
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
  PT(22);
  float* e2 = S.ph.end.red + 56;         // ring entry #2; entry #1 was saved in registers below before red[] was reused
  receive_obs(P, rec, S, lane, valid, e2);  // ring entry #2 (imitation_task.py:792)
  {
    // control observation with two entries in the ring (Minitaur._get_delay_obs, minitaur.py:336-357): latency <= 0 -> newest;
    // int(latency / dt) + 1 >= 2 -> the OLDEST entry (#1, the default pose: SURVEY 8a quirk 3); else blend newest / #1
    const float lat = S.s[O(LATENCY)], dt = c.sim_dt;
    const int n = (int)(lat / dt);
    const float al = (lat - n * dt) / dt;
    WSYNC();
    for (int i = lane; i < 19; i += kLanes) {
      const float newest = e2[i], oldest = i < 16 ? e1_keep[0] : e1_keep[1];
      S.co[i] = lat <= 0.0f ? newest : (n + 1 >= 2 ? oldest : (1.0f - al) * newest + al * oldest);
    }
    WSYNC();
  }
  // 7. observation = histories from step 3 + target observation (quadruped_gym_env.py:100-102; wrapper_env.py:101-105)
  if (lane == 0) seti(S, O(MAX_EP_STEPS), time_limit(c, total_step_count));
  if (lane < 12) obs[lane] = S.s[O(IMU_HIST) + lane];
  for (int i = lane; i < 36; i += kLanes) { obs[12 + i] = S.s[O(LASTACT_HIST) + i]; obs[48 + i] = S.s[O(MOTORANG_HIST) + i]; }
  PT(23);
  target_obs(P, rec, S, lane, obs + ORR_PROPRIO_DIM);
  PT(30);
}
