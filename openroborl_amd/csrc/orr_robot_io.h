// orr_robot_io.h -- per-robot record load / store and the latency ring (Minitaur.receive_obs / _get_delay_obs)
// (device code of libopenroborl_hip.so, included by orr_kernels.hip after orr_device.h; see DESIGN.md sections 3-5)
#pragma once

// ================================================================================================
// load / store of the per-robot record
// ================================================================================================
// The state head moves in 16-byte pieces (global_load_dwordx4 / ds_write_b128 and back): 5 instead of 20 memory instructions per lane
// at either end of a launch, where all waves of the chip load / store their records at about the same time.
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ static void load_robot(const KParams& P, const float* rec, Shared& S, int lane) {
  {
    static_assert(kHead % 4 == 0 && ORR_STATE_STRIDE % 4 == 0, "16-byte pieces");
    typedef const f4 __attribute__((address_space(1))) * g4ptr;
    const g4ptr src = (g4ptr)reinterpret_cast<const f4*>(rec);
    f4* dst = reinterpret_cast<f4*>(S.s);
    constexpr int kQ = kHead / 4, kIt = (kQ + kLanes - 1) / kLanes;
    f4 tmp[kIt];
#pragma unroll
    for (int k = 0; k < kIt; k++) { const int q = lane + k * kLanes; tmp[k] = src[q < kQ ? q : kQ - 1]; }
#pragma unroll
    for (int k = 0; k < kIt; k++) { const int q = lane + k * kLanes; if (q < kQ) dst[q] = tmp[k]; }
  }
  WSYNC();
  const DevModel& gm = P.tab->model[geti(S, O(ROBOT_TYPE))];
  {
    static_assert(sizeof(DevClip) % 4 == 0 && sizeof(DevClip) / 4 <= kLanes, "one word of the clip header per lane");
    const unsigned int* cg = reinterpret_cast<const unsigned int*>(&P.tab->clip[geti(S, O(CLIP_ID))]);
    unsigned int* cl = reinterpret_cast<unsigned int*>(&S.clip);
    if (lane < (int)(sizeof(DevClip) / 4)) cl[lane] = cg[lane];
  }
  {
    // the hot part of the model (512 B = two 16-byte pieces per lane), in flight together; the cold part stays in the device table
    // (ModelCold) and the mass properties go straight into the registers of load_leg_const.  The table lives in global memory:
    // global loads (address space 1), not FLAT
    typedef const f4 __attribute__((address_space(1))) * g4ptr;
    static_assert(sizeof(ModelHot) % (16 * kLanes) == 0 && alignof(DevModel) >= 16, "whole 16-byte rows of the hot model per lane");
    const g4ptr mp = (g4ptr)reinterpret_cast<const f4*>(&gm.hot);
    f4* dst = reinterpret_cast<f4*>(&S.m);
    constexpr int kIter = (int)(sizeof(ModelHot) / 16) / kLanes;
    f4 tmp[kIter];
#pragma unroll
    for (int k = 0; k < kIter; k++) tmp[k] = mp[lane + k * kLanes];
#pragma unroll
    for (int k = 0; k < kIter; k++) dst[lane + k * kLanes] = tmp[k];
  }
  WSYNC();
}

__device__ static void store_robot(float* rec, const Shared& S, int lane, bool valid) {
  if (valid) {
    typedef f4 __attribute__((address_space(1))) * g4ptr;
    const g4ptr dst = (g4ptr)reinterpret_cast<f4*>(rec);
    const f4* src = reinterpret_cast<const f4*>(S.s);
    constexpr int kQ = O(RING) / 4;                 // whole 16-byte pieces of the head; the remaining words one by one
#pragma unroll
    for (int k = 0; k < (kQ + kLanes - 1) / kLanes; k++) { const int q = lane + k * kLanes; if (q < kQ) dst[q] = src[q]; }
    if (lane < O(RING) - 4 * kQ) rec[4 * kQ + lane] = S.s[4 * kQ + lane];
  }
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
#pragma unroll
  for (int i = 0; i < 9; i++) S.Rb[i] = Rb[i];   // the same in every lane: all of them store it (no divergent `if`)
}

// Minitaur.receive_obs + get_true_obs (minitaur.py:304-334): push the true observation.  entry: optional copy of the pushed
// entry (20 words, LDS) for callers that build the control observation without reading the ring back (reset_robot).
__device__ static void receive_obs(const KParams& P, float* rec, Shared& S, int lane, bool valid, float* entry = nullptr) {
  const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
  const int head = (geti(S, O(RING_HEAD)) + 1) % ORR_RING_DEPTH, len = geti(S, O(RING_LEN));
  float rel[4], Rb[9], rate[3];
  base_rotation(S, lane, rel, Rb);
  mtv3(Rb, &S.s[O(ANGVEL)], rate);  // get_true_base_rpy_rate (minitaur.py:640-672): angular velocity in the base frame
  for (int i = lane; i < ORR_RING_ENTRY; i += kLanes) {
    float val = 0.0f;
    if (i < 12) {
      int j = mc->joint_of_motor[i];
      val = (S.s[O(Q) + j] - mc->motor_offset[i]) * mc->motor_dir[i];  // get_true_motor_angles (:543-553)
    } else if (i < 16) {
      val = i == 12 ? rel[0] : (i == 13 ? rel[1] : (i == 14 ? rel[2] : rel[3]));
    } else if (i < 19) {
      val = i == 16 ? rate[0] : (i == 17 ? rate[1] : rate[2]);
    }
    if (valid) rec[O(RING) + head * ORR_RING_ENTRY + i] = val;
    if (entry) entry[i] = val;
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
// co_own (optional): receives the control-observation word this lane computes (S.co[lane]: the delayed motor angle of motor `lane`
// for lanes < 12), so that the PD law of the next sub-step need not read it back from LDS
__device__ __forceinline__ void ring_push_and_ctrl_obs(float* rec, Shared& S, int lane, bool valid, const RingFetch& F, RingCursor& C,
                                                       float mang, float* co_own = nullptr) {
  static_assert(ORR_RING_ENTRY == 20, "lane mapping below assumes 20-word entries");
  C.head = ring_wrap_up(C.head + 1);
  C.len = C.len + 1 > ORR_RING_DEPTH ? ORR_RING_DEPTH : C.len + 1;
  float rel[4], Rb[9], rate[3];
  base_rotation(S, lane, rel, Rb);
  mtv3(Rb, &S.s[O(ANGVEL)], rate);  // get_true_base_rpy_rate (minitaur.py:640-672): angular velocity in the base frame
  // word `lane`: motor angles 0..11 (get_true_motor_angles, :543-553), relative quaternion 12..15;
  // word 16 + lane (lanes 0..3): rate 16..18, pad 19
  const float va = lane < 12 ? mang : pick4(lane, rel[0], rel[1], rel[2], rel[3]);
  const float vb = pick4(lane, rate[0], rate[1], rate[2], 0.0f);   // lanes >= 4: never stored
  const float a0 = F.new0 ? va : F.e0[0], a1 = F.new1 ? va : F.e1[0];
  const float b0 = F.new0 ? vb : F.e0[1], b1 = F.new1 ? vb : F.e1[1];
  const float co_a = F.same ? a0 : (1.0f - F.al) * a0 + F.al * a1;
  S.co[lane] = co_a;
  if (co_own) *co_own = co_a;
  S.co[lane < 3 ? 16 + lane : 19] = F.same ? b0 : (1.0f - F.al) * b0 + F.al * b1;   // lanes >= 3: the pad word (no divergent `if`)
  WSYNC();
  // the push is stored AFTER the prefetched entries were consumed: loads and stores share one in-order counter (vmcnt) on this
  // target, so a wait for the (long finished) prefetch behind a fresh store would sit out the store's whole round trip
  float* dst = rec + O(RING) + C.head * ORR_RING_ENTRY;
  if (valid) {
    dst[lane] = va;
    dst[lane < 4 ? 16 + lane : lane] = lane < 4 ? vb : va;   // lanes >= 4 repeat their first store (one branch instead of two)
  }
}
