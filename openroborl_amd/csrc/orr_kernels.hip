// orr_kernels.hip -- HIP kernels (gfx950) + C-ABI of the vectorised quadruped imitation env.
//
// Hot path replaced: WrapperEnv.step / reset (wrapper_env.py:58-107) -> LocomotionGymEnv._step /
// reset (quadruped_gym_env.py:63-104,213-239) -> Minitaur (minitaur.py) + ImitationTask
// (imitation_task.py) + pybullet.stepSimulation.  One launch = one env step for all robots of
// this device: 33 physics sub-steps, observation, reward, termination, optional auto-reset.
// Specification of every stage: DESIGN.md section 4; CPU restatement: oracle/orr_oracle.c.
// Device code by phase: orr_device.h (LDS image, math, DPP helpers), orr_robot_io.h (record load / store, latency
// ring), orr_physics.h (one physics sub-step), orr_task.h (motion clips, reward, observation, reset); this file holds
// the two kernels and the C-ABI.
// Three translation units are built from this file (the third, orr_kernels_anchor.hip, holds the friction-anchor variants: see launch_step_anchor).  The main one (everything) is compiled with the instruction-level-parallelism
// scheduler: one wave per SIMD, ~300 registers, nothing to hide latency but the wave's own independent instructions.  The second
// one (orr_kernels_w2.hip: #define ORR_TU_STEP_W2 + #include of this file) holds ONLY the two-waves-per-SIMD instantiation of the step
// kernel and is compiled with an occupancy-minded scheduler (-Os + iterative-maxocc since the end of round 4: openroborl_amd/_lib.py HIPCC_FLAGS_W2; the compiler's default before): at
// 256 registers that variant spills, and the ILP schedule's longer live ranges cost it 8 % (0.382 vs 0.352 ms at 8192 robots; the default
// scheduler costs the one-wave variant 9 %).
#if defined(ORR_TU_STEP_W2) || defined(ORR_TU_STEP_ANCHOR)
#undef ORR_PHASE_TIMERS      // the development timers live in the main translation unit only
#define ORR_TU_SECONDARY 1   // a unit that holds only instantiations of the step kernel and their launchers
#endif
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// The constraint rows need A0^-1 (Jb - T_L jl): every row lane solves with the Cholesky factor of A0, which stays in registers (15 entries + 6 reciprocal pivots, Chol6Pk) from
// the leg dynamics on (-54 instructions per sub-step there: no unit-column solve, no A0^-1 through LDS; +15 per bank in the row
// response).  4096 robots 0.2338 -> 0.2295 ms; the two-wave unit (256 registers, spilling) is neutral (8192 robots 0.3265 vs 0.3255 ms)
// and takes it too, so that both variants of the kernel keep giving the same bits.
#include "orr_device.h"

// Development aid (tools/phase_cycles.py): -DORR_PHASE_TIMERS makes lane 0 of one wave accumulate shader-clock cycles
// per phase (Shared::pt_acc) and add them to g_phase_cycles at the end of the launch.
#ifdef ORR_PHASE_TIMERS
__device__ long long g_phase_cycles[orr::kPhaseSlots];   // 0..15: phases of the step, 16..23: stages of reset_robot, 24..: finer marks inside the reset
__device__ long long g_wave_phases[2048 * 40];   // per wave of the last launch: its own phase totals (tools/wave_phases.py)
__device__ long long g_wave_timeline[4 * 2048];   // per wave of the last launch: realtime start, realtime end, shader cycles, reset flag
#define PT_INIT() do { if (threadIdx.x == 0) { S.pt_t0 = clock64(); S.pt_r0 = wall_clock64(); } if (threadIdx.x == 0) { for (int i_ = 0; i_ < orr::kPhaseSlots; i_++) S.pt_acc[i_] = 0; S.pt_last = clock64(); } } while (0)
#define PT(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); S.pt_acc[k] += t_ - S.pt_last; S.pt_last = clock64(); } } while (0)
#define PT_TIMELINE(flag) do { if (threadIdx.x == 0 && blockIdx.x < 2048) { g_wave_timeline[4 * blockIdx.x] = S.pt_r0; g_wave_timeline[4 * blockIdx.x + 1] = wall_clock64(); g_wave_timeline[4 * blockIdx.x + 2] = clock64() - S.pt_t0; g_wave_timeline[4 * blockIdx.x + 3] = ((flag) & 0xFF) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 40); } } while (0)   /* bits 8..39: HW_REG_HW_ID (wave, simd, cu, sh, se), 40..43: HW_REG_XCC_ID */
#define PT_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) for (int i_ = 0; i_ < orr::kPhaseSlots; i_++) atomicAdd((unsigned long long*)&g_phase_cycles[i_], (unsigned long long)S.pt_acc[i_]); \
                        if (threadIdx.x == 0 && blockIdx.x < 2048) for (int i_ = 0; i_ < orr::kPhaseSlots; i_++) g_wave_phases[blockIdx.x * 40 + i_] = S.pt_acc[i_]; } while (0)
#else
#define PT_INIT()
#define PT(k)
#define PT_FLUSH()
#define PT_TIMELINE(flag)
#endif

// Development aid (tools/wave_times.py): -DORR_WAVE_TIMELINE makes every wave of the step kernel - BOTH variants, product code otherwise -
// record when it started and ended (100 MHz realtime counter), its shader cycles and the hardware slot it ran on (nothing else is
// instrumented: two s_memrealtime / s_memtime pairs and one 32-byte store per wave)
#ifdef ORR_WAVE_TIMELINE
#define WT_INIT() const long long wt_r0 = wall_clock64(), wt_c0 = clock64()
#define WT_STORE(flag) do { if ((threadIdx.x & 63u) == 0 && P.wave_times) { long long* w_ = P.wave_times + 4 * (size_t)wave_id; w_[0] = wt_r0; w_[1] = wall_clock64(); w_[2] = clock64() - wt_c0; \
    w_[3] = ((flag) & 0xFF) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 40); } } while (0)
#else
#define WT_INIT()
#define WT_STORE(flag)
#endif

// Development aid (tools/dual_contact.py): -DORR_COUNT_DUAL_CONTACT counts, per leg and sub-step, how often the toe sphere and the shank
// sphere of a lower leg are within the contact margin / penetrating at the same time (the engine makes ONE contact point per leg, Bullet
// one per touching shape: DESIGN.md section 9).  One-wave kernel only.
#if defined(ORR_COUNT_DUAL_CONTACT) && !defined(ORR_TU_SECONDARY)
__device__ unsigned long long g_dual_contact[8];
// (called inside the contact branch of row_setup, where lanes 0..3 of every robot are inactive: the first ACTIVE lane adds the wave's count)
#define ORR_DUAL_COUNT(k, cond) do { const unsigned long long b_ = __ballot(cond), act_ = __ballot(1); \
    if (b_ && (int)(threadIdx.x & 63u) == __ffsll((long long)act_) - 1) atomicAdd(&g_dual_contact[k], (unsigned long long)__popcll(b_)); } while (0)
#else
#define ORR_DUAL_COUNT(k, cond)
#endif

using namespace orr;

#define O(name) ORR_OFF_##name

#include "orr_robot_io.h"
#include "orr_physics.h"
#include "orr_task.h"

// ================================================================================================
// kernels
// ================================================================================================
// lane group bookkeeping shared by the kernels: `sub` = which robot of this wave, `lane` = lane within the robot
// WPB = wavefronts per workgroup (each wave is an independent quad of robots; nothing is shared between the waves of a workgroup)
#define ORR_PROLOGUE() ORR_PROLOGUE_W(1)
#define ORR_PROLOGUE_W(WPB)                                                              \
  __shared__ Shared Sarr[kRPW * (WPB)];                                                  \
  const int wtid = (WPB) > 1 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;              \
  const int wave_in_wg = (WPB) > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0; \
  const int wave_id = (int)blockIdx.x * (WPB) + wave_in_wg;                              \
  const int sub = wtid / kLanes, lane = wtid % kLanes;                                   \
  Shared& S = Sarr[wave_in_wg * kRPW + sub];                                             \
  float* obs = S.ph.end.obs;                                                             \
  const int robot_raw = wave_id * kRPW + sub;                                            \
  const bool in_range = robot_raw < P.cfg.num_robots;                                    \
  const int robot = in_range ? robot_raw : 0; /* a padding lane group shadows robot 0 and never stores */ \
  float* rec = P.state + (size_t)robot * ORR_STATE_STRIDE

#ifndef ORR_TU_SECONDARY
__global__ __launch_bounds__(64) void orr_reset_kernel(KParams P, const uint8_t* mask, float* obs_out, const float* uniforms) {
  ORR_PROLOGUE();
  const bool valid = in_range && !(mask && !mask[robot]);
  load_robot(P, rec, S, lane);
  const long long total = P.counters[ORR_CNT_TOTAL_STEP_COUNT];
  reset_robot(P, rec, S, lane, valid, total, obs, uniforms ? uniforms + (size_t)robot * 28 : nullptr);
  WSYNC();
  store_robot(rec, S, lane, valid);
  // a new episode: no cached contact points (ANCHOR, ANCHOR_VALID: 28 words behind the ring).  Unconditional: friction anchors may be switched
  // on (orr_set_model) after this reset, and a caller-bound record need not have been zeroed
  if (valid)
    for (int i = lane; i < ORR_OFFEND_ANCHOR_VALID - ORR_OFF_ANCHOR + 1; i += kLanes) rec[O(ANCHOR) + i] = 0.0f;
  if (obs_out && valid)
    for (int i = lane; i < ORR_OBS_DIM; i += kLanes) obs_out[(size_t)robot * ORR_OBS_DIM + i] = obs[i];
}
#endif  // !ORR_TU_SECONDARY

// mode 0: full env step.  mode 1 (debug / parity of row C): nsub physics sub-steps with the given
// motor torques (actions = torques), no robot or task logic.  mode 2 (parity of everything BUT row C): a full env step in
// which the physics sub-step is replaced by the recorded states of ReplayArgs, the end-effector reward reads recorded link
// positions and the fall flag is given -- the device-side counterpart of the oracle's replay mode, fed with the fixtures that the
// reference's own Python produced (tests/test_gpu_golden_task.py).
// WPE = waves per SIMD the kernel is compiled for.  WPE 1: up to 512 VGPRs (~300 used), one wave on each of the 1024 SIMDs = 4096 robots
// resident at once: the best a batch of <= 4096 robots can do.  WPE 2 (<= 256 VGPRs, 28 of them spilled, one scratch access inside the sub-step loop; LDS 19.3 KB per wave, so
// eight waves fit a CU): for larger batches.  A lone wave issues one vector instruction per ~5 cycles, the SIMD can take one per 2:
// two co-resident waves of this kernel take 1.12x as long as one alone (tools/wave_pairing.py), i.e. 1.8x the throughput per SIMD,
// where the WPE-1 kernel would run the second thousand waves after the first.  orr_step picks the variant from the batch size and the
// device's CU count (ORR_STEP_WAVES_PER_EU = 1 | 2 overrides, for measurements).
// (min, max) waves per SIMD are pinned to the same value: with a higher maximum this LLVM's iterative-ilp scheduler tries
// occupancy-improving reschedules once the kernel fits 256 VGPRs and then crashes in the register allocator.
#ifndef ORR_WAVES_PER_EU
#define ORR_WAVES_PER_EU 1   // development builds (-DORR_WAVES_PER_EU=2) force every instantiation to that occupancy
#endif
#ifndef ORR_WPB
#define ORR_WPB 1            // wavefronts per workgroup of the one-wave-per-SIMD env step (tuning experiments: 2, 4)
#endif
template <int MODE, int WPE>
constexpr int step_wpb() { return MODE == 0 && WPE == 1 ? ORR_WPB : 1; }
// ANCHOR (ABI v5): the variant for robot types with orr_model::friction_anchor - Bullet's cached toe contact points (orr_physics.h:
// AnchorState).  Same source; its own instantiations (one wave per SIMD whatever the batch size: an optional physics feature, not the
// measured path), so that the default kernels carry nothing of it.
template <int MODE, int WPE = ORR_WAVES_PER_EU, bool ANCHOR = false>
__global__ __launch_bounds__((MODE == 0 && WPE == 1 ? 64 * ORR_WPB : 64)) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void orr_step_kernel(KParams P, const float* actions, float* obs_out, float* reward_out,
                                                      uint8_t* done_out, int nsub, ReplayArgs RP) {
  ORR_PROLOGUE_W((step_wpb<MODE, WPE>()));
  const bool valid = in_range;
  const orr_config& c = P.cfg;
  PT_INIT();
  WT_INIT();
  // curriculum counter as of the start of the launch (the last wave of the previous launch folded that launch's episodes in):
  // read here, far ahead of its only use (the time limit of an episode that starts in this launch)
  const long long total_snapshot = P.counters[ORR_CNT_TOTAL_STEP_COUNT];
  load_robot(P, rec, S, lane);
  {
    // Non-finite guard, entry half: a NaN in the INCOMING rigid state (POS..QD) does not survive the step - the velocity clamp (+-100,
    // v_med3) and the branch-free inverse trigonometric functions turn NaNs into finite numbers - so it is recorded here, in a spare
    // word of the LDS image behind the state head (never stored), and ORed into the exit half of the guard (ORR_DONE_NAN below).
    static_assert(kHead > ORR_OFF_RING, "spare LDS word behind the state head (the head = everything in front of the ring)");
    bool bad_in = false;
    for (int i = lane; i < 37; i += kLanes) bad_in = bad_in || !(fabsf(S.s[O(POS) + i]) < 1e30f);
    const bool any_bad = ((__ballot(bad_in) >> (sub * kLanes)) & ((1ull << (kLanes - 1)) * 2ull - 1ull)) != 0ull;
    if (lane == 0) S.s[kHead - 1] = any_bad ? 1.0f : 0.0f;
  }
  // impulse-response table: stale rows are multiplied by zero impulses, so they only have to be finite
  for (int i = lane; i < kMaxRows * kWStride; i += kLanes) (&S.ph.sub.W[0][0])[i] = 0.0f;
  WSYNC();
  LegConst K;
  load_leg_const(P, S, lane, K);
  {
    float rel[4], Rb[9];
    base_rotation(S, lane, rel, Rb);  // Shared::Rb for the first sub-step; the ring push keeps it current afterwards
  }
  // friction anchors (ANCHOR variant only): the cached contact point of the lane's leg, from the record's words behind the ring
  AnchorState AS = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0};
  bool anchor_robot = false;
  const int aleg = lane < 4 ? lane : (lane < 8 ? lane - 4 : (lane - 8) >> 1);   // leg of the lane's bank-A row (knee, normal, friction)
  if constexpr (ANCHOR) {
    static_assert(kLanes == 16, "ANCHOR: aleg indexes the record's four cached points by the 16-lane row layout");
    anchor_robot = model_cold(P, geti(S, O(ROBOT_TYPE)))->friction_anchor != 0;
    const float* an = rec + O(ANCHOR) + 6 * aleg;
    AS.la[0] = an[0]; AS.la[1] = an[1]; AS.la[2] = an[2]; AS.wb[0] = an[3]; AS.wb[1] = an[4]; AS.wb[2] = an[5];
    AS.valid = anchor_robot ? __float_as_int(rec[O(ANCHOR_VALID) + aleg]) : 0;
  }
  auto store_anchor = [&]() __attribute__((always_inline)) {      // the normal-row lanes (4..7) own the record's words
    if constexpr (ANCHOR) {
      if (valid && lane >= 4 && lane < 8) {
        float* an = rec + O(ANCHOR) + 6 * aleg;
        an[0] = AS.la[0]; an[1] = AS.la[1]; an[2] = AS.la[2]; an[3] = AS.wb[0]; an[4] = AS.wb[1]; an[5] = AS.wb[2];
        rec[O(ANCHOR_VALID) + aleg] = __int_as_float(AS.valid);
      }
    }
  };
  PT(0);

  if (MODE == 1) {
    if (lane < 12) {
      const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
      const int j = mc->joint_of_motor[lane];
      S.tau[j] = mc->tau_sign_motor[lane] * actions[(size_t)robot * 12 + lane];
    }
    WSYNC();
    int fall = 0;
    OwnCoord X;
    load_own_coord(S, lane, X);
    int limit_idle = 0;
    for (int s = 0; s < nsub; s++) {
      fall = physics_substep<ANCHOR>(P, S, K, lane, sub, true, X, limit_idle, ANCHOR ? &AS : nullptr, anchor_robot);
      float rel[4], Rb[9];
      base_rotation(S, lane, rel, Rb);
      WSYNC();
    }
    if (valid && lane == 0 && done_out) done_out[robot] = (uint8_t)fall;
    store_robot(rec, S, lane, valid);
    store_anchor();
    return;
  }

  // per-motor constants of the PD loop (lane = motor; lanes 12..15 repeat motor 0), from the cold model table into registers: issued
  // here, first used after the control observation's ring reads
  const int ml = lane < 12 ? lane : 0;
  const ColdPtr mc = model_cold(P, geti(S, O(ROBOT_TYPE)));
  const int mj = mc->joint_of_motor[ml];
  const float m_off = mc->motor_offset[ml], m_dir = mc->motor_dir[ml], m_kp = mc->kp[ml], m_kd = mc->kd[ml];
  const float m_tsign = mc->tau_sign_motor[ml], m_init = mc->init_motor_angles[ml];
  // ---- set_act (minitaur.py:280-285): offset, last action, Butterworth filter ----
  ctrl_obs(P, rec, S, lane);
  // Non-finite guard, action half: the +-0.2 rad clip of the motor command (fmin / fmax) would silently DROP a NaN action while the
  // last-action sensor and the filter history keep it.  A non-finite action is recorded like a non-finite incoming state (ORR_DONE_NAN
  // at the end of the step) and replaced by 0, so that nothing non-finite enters the record.
  float act_in = actions[(size_t)robot * 12 + ml];
  {
    const bool bad_act = !(fabsf(act_in) < 1e30f);
    if (bad_act) act_in = 0.0f;
    const bool any_bad = ((__ballot(bad_act) >> (sub * kLanes)) & ((1ull << (kLanes - 1)) * 2ull - 1ull)) != 0ull;
    if (lane == 0 && any_bad) S.s[kHead - 1] = 1.0f;
  }
  if (lane < 12) {
    const float act = act_in + m_init;
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
  // the step's filtered target and the motor gain, in registers over the sub-steps (lane = motor)
  const float m_gain = m_tsign * S.s[O(STRENGTH) + ml];
  const float m_target = S.s[O(ACTION) + ml], m_prev = S.s[O(FILTER_ACTION) + ml];
  const bool m_has_prev = geti(S, O(FILTER_VALID)) != 0;
  int action_counter = geti(S, O(STATE_ACTION_COUNTER));
  RingCursor ring = {geti(S, O(RING_HEAD)), geti(S, O(RING_LEN))};
  OwnCoord X;
  load_own_coord(S, lane, X);
  int limit_idle = 0;      // see physics_substep
  // Two waves per SIMD: VALU issue is arbitrated by priority, then AGE - the older wave of a SIMD runs nearly unimpeded, the younger on
  // the leftover slots, and when the older one has finished the younger runs on alone at a lone wave's pace (half the SIMD idle).  The
  // two waves of a SIMD come from consecutive dispatch rounds (workgroup b: round b / #SIMDs), so raising the priority of the even rounds
  // in even sub-steps and of the odd rounds in odd sub-steps lets them take turns at being the favoured one and finish together.
  // 8192 robots: 0.349 -> 0.331 ms (-5.3 %, round 3; turns of 2 or 4 sub-steps or a second flip in the middle of a sub-step were no
  // better then: 0.332 / 0.332 / 0.335).  Round 4, final code: turns of FOUR sub-steps 0.3015 -> 0.2991 ms (2: 0.2995, 8 / 16: 0.3032 /
  // 0.3027; without the alternation 0.317; profiles/r04_ab25..28_8192.log).
  const int prio_phase = WPE == 2 ? (int)(((unsigned)wave_id / (unsigned)(P.simds > 0 ? P.simds : 1)) & 1u) : 0;
  // Taking turns pairs the dispatch rounds (0, 1), (2, 3) ...: with an ODD number of rounds the last one has no partner of its own, and the
  // plain age order - the oldest wave of a SIMD runs at nearly a lone wave's pace, the next one moves up when it ends - is the better
  // pipeline (12288 robots = 3 rounds: 0.4925 -> 0.4693 ms without the turns; 16384 / 32768 robots = 4 / 8 rounds: 0.5767 / 1.103 ms with
  // them against 0.5847 / 1.109 without; profiles/r04_ab34_large.log)
  const bool prio_turns = WPE == 2 && ((((unsigned)gridDim.x * (unsigned)step_wpb<MODE, WPE>() + (unsigned)(P.simds > 0 ? P.simds : 1) - 1u) /
                                        (unsigned)(P.simds > 0 ? P.simds : 1)) & 1u) == 0u;
  // What the PD law of a sub-step reads - the delayed angle of the lane's motor (control observation), the joint's true angle and rate -
  // is produced at the END of the previous sub-step: the control-observation word by this very lane, angle and rate by the integration
  // (read from LDS there anyway, for the ring entry).  Carried over in registers, the top of the loop has no LDS round
  // trip of its own: a lone wave has nothing to overlap one with there.  Same values, bit for bit.  4096 robots 0.2235 -> 0.2205 ms.
  // The two-wave build lost 1 % with it when it was introduced (8192 robots 0.3130 -> 0.3160 ms: three more registers across the sub-step)
  // and takes it since the end of round 4: a wave pair runs 1.33 x the LONE time of its build whatever the scheduling (DESIGN.md section
  // 10), and on the final code the carry is worth 0.3014 -> 0.2991 ms (profiles/r04_ab39_8192.log).  Measured and NOT kept in the one-wave build:
  // carrying the base rotation the same way (nine words that leg_dynamics reads back from LDS: 0.2206, neutral), and issuing the loads
  // of the leg dynamics' first reads (base rotation / velocity, own joint angle, the leg's joint rates: 19 registers) in front of the
  // PD law (0.2208 -> 0.2223: worse).
  float co_own = S.co[ml], qm_c = (S.s[O(Q) + mj] - m_off) * m_dir, qdm_c = S.s[O(QD) + mj] * m_dir;
  for (int sstep = 0; sstep < c.action_repeat; sstep++) {
    // priority turns of the two-wave variant (measured: profiles/r04_ab25..30_8192.log): a wave is favoured for kPrioTurn sub-steps at a time; the turns
    // start kPrioOffset sub-steps early, i.e. the younger wave of a SIMD leads with a turn of three and has the last two sub-steps (8192 robots
    // 0.3012 -> 0.2998 ms; offsets 2 / 3 / 4 (= the older wave leads): 0.3030 / 0.3042 / 0.3046; equal priority for the last 4 / 8 sub-steps:
    // 0.3026 / 0.3014, not kept).  kPrioEqualFrom: equal priority from that sub-step on; 1000 = never (action_repeat is 33), the comparison
    // stays so that the unit's code is the measured one
    constexpr int kPrioTurn = 4, kPrioHi = 1, kPrioOffset = 1, kPrioEqualFrom = 1000;
    if (WPE == 2 && prio_turns) { if (sstep < kPrioEqualFrom && ((((sstep + kPrioOffset) / kPrioTurn) ^ prio_phase) & 1)) __builtin_amdgcn_s_setprio(kPrioHi); else __builtin_amdgcn_s_setprio(0); }
    if (kLanes != 16 && sstep > 0) ctrl_obs(P, rec, S, lane);
    {  // every lane (no divergent `if`: it would cost more than it skips); lanes 12..15 repeat motor 0 and store into dump slots
      const float lerp = (float)(sstep + 1) * inv_repeat;  // process_action (minitaur.py:438-460)
      const bool carry = kLanes == 16;
      const float cur = map_pi(carry ? co_own : S.co[ml]);
      const float prev = m_has_prev ? m_prev : cur;
      float cmd = prev + lerp * (m_target - prev);
      cmd = fminf(fmaxf(cmd, cur - c.max_angle_change), cur + c.max_angle_change);  // _clip_motor_commands (:706-723)
      const float qm = carry ? qm_c : (S.s[O(Q) + mj] - m_off) * m_dir;  // pd latency 0 (:359-363)
      const float qdm = carry ? qdm_c : S.s[O(QD) + mj] * m_dir;
      // MotorModel.convert_to_torque, POSITION mode (minitaur_motor.py:163-171)
      S.tau[lane < 12 ? mj : lane] = m_gain * (-1.0f * (m_kp * (qm - cmd)) - m_kd * qdm);
    }
    WSYNC();
    action_counter++;  // robot_step bookkeeping (minitaur.py:287-293); written back after the loop
    PT(2);
    if (kLanes == 16) {  // receive_obs, then the control observation of the next sub-step / of get_obs
      RingFetch F;
      ring_prefetch(rlat, rec, ring, lane, F);
      if constexpr (MODE == 2) {
        const size_t slot = (size_t)robot * c.action_repeat + sstep;
        if (lane < 12 && valid) RP.tau_out[slot * 12 + lane] = S.tau[mj] * m_tsign;  // motor torque, motor order
        WSYNC();
        for (int i = lane; i < 37; i += kLanes) S.s[O(POS) + i] = RP.traj[slot * 37 + i];        // POS QUAT LINVEL ANGVEL Q QD
        WSYNC();
        fall = RP.fall[robot];
      } else
      fall = physics_substep<ANCHOR>(P, S, K, lane, sub, sstep == c.action_repeat - 1, X, limit_idle, ANCHOR ? &AS : nullptr, anchor_robot);
      qm_c = (S.s[O(Q) + mj] - m_off) * m_dir;
      qdm_c = S.s[O(QD) + mj] * m_dir;
      ring_push_and_ctrl_obs(rec, S, lane, valid, F, ring, qm_c, &co_own);
    } else {
      fall = physics_substep<ANCHOR>(P, S, K, lane, sub, sstep == c.action_repeat - 1, X, limit_idle, ANCHOR ? &AS : nullptr, anchor_robot);
      receive_obs(P, rec, S, lane, valid);
    }
    PT(10);
  }
  // past its sub-step loop (reward, observation, reset, store: few vector instructions between long memory waits) a wave issues ahead of its
  // partner: it costs the partner next to nothing and shortens the tail (8192 robots: -0.1 %; with -Os for this unit -0.4 %, profiles/r04_ab29_8192.log)
  if (WPE == 2) __builtin_amdgcn_s_setprio(3);
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
  // the frames of the new reference poses are fetched while the reward is computed (their round trip to L2 is not waited for)
  const DevClip& clip = S.clip;
  const double t = motion_time(P, S);                       // f64: see DevClip
  const double step_dt = clip.sim_dt_d * c.action_repeat;
  double tl = t;
  {  // lanes 1..4: the four target times.  Selects over the four scalars: indexing the kernel argument with the lane makes the
     // compiler read it from memory with a vector load, whose wait also drains every store and atomic issued before it
    const int k = lane - 1;
    const int steps = (k & 2) ? ((k & 1) ? c.tar_frame_steps[3] : c.tar_frame_steps[2]) : ((k & 1) ? c.tar_frame_steps[1] : c.tar_frame_steps[0]);
    if (lane >= 1 && lane <= 4) tl = t + steps * step_dt;
  }
  PoseLoads PL;
  sample_poses_issue(P, S, lane, tl, PL);
  float rew = calc_reward(P, S, lane, MODE == 2 ? RP.eff + (size_t)robot * 48 : nullptr);
  sample_poses_finish(P, S, lane, tl, true, PL);
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
  // the slot is only defined where the atomic was issued (lane 0 of a robot whose episode ended) and only read there.  A frozen
  // nondeterministic value instead of an uninitialised variable: reading it is defined behaviour in every lane, and unlike a constant
  // it gives the compiler nothing to merge with the atomic's result (a merged value made it wait for the atomic right away)
  unsigned long long log_slot = __builtin_nondeterministic_value(log_slot);
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
    if (!(clip.flags & ORR_CLIP_WRAP) && t >= clip.dur_d) reason |= ORR_DONE_MOTION_OVER;  // is_motion_over (:224-233)
    bool bad = false;
    for (int i = lane; i < 37; i += kLanes) bad = bad || !(fabsf(S.s[O(POS) + i]) < 1e30f);
    if (((__ballot(bad) >> (sub * kLanes)) & ((1ull << (kLanes - 1)) * 2ull - 1ull)) != 0ull || S.s[kHead - 1] != 0.0f) reason |= ORR_DONE_NAN;
    if (!(fabsf(rew) < 1e30f)) reason |= ORR_DONE_NAN;
    if (reason & ORR_DONE_NAN) rew = 0.0f;      // whatever was computed from a non-finite state is not a reward
    const int ep_step = geti(S, O(EP_STEP)) + 1;  // quadruped_gym_env.py:237
    if (ep_step >= geti(S, O(MAX_EP_STEPS))) reason |= ORR_DONE_TIME_LIMIT;
    // episode log (imitation_runners.py:185-197): the slot comes from a returning atomic on a counter shared by the whole device (a
    // round trip of several microseconds).  It is issued HERE, as soon as the end of the episode is known, and consumed after the
    // reset: the observation, the target observation and the first stages of the reset run while it is in flight
    if (lane == 0 && valid && reason != 0) log_slot = atomicAdd((unsigned long long*)&P.counters[ORR_CNT_EPISODES], 1ull);
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
  if (reason != 0) {
    float log_ret = 0.0f, log_len = 0.0f;
    const bool logs = lane == 0 && valid;
    const unsigned long long slot = log_slot;
    if (lane == 0) {
      log_ret = S.s[O(EP_RETURN)]; log_len = (float)geti(S, O(EP_STEP));
      S.s[O(LAST_EP_RETURN)] = log_ret;
      seti(S, O(LAST_EP_LEN), geti(S, O(EP_STEP)));
    }
    WSYNC();
    if (c.flags & ORR_FLAG_AUTO_RESET) {
      PT(31);
      reset_robot(P, rec, S, lane, valid, total_snapshot, obs);
      if constexpr (ANCHOR) AS = AnchorState{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0};   // a new episode: no cached contact points
    }
    if (logs && P.ep_log) {
      if (slot < (unsigned long long)P.ep_log_cap) {
        P.ep_log[2 * slot] = log_ret;
        P.ep_log[2 * slot + 1] = log_len;
      } else {
        atomicAdd((unsigned long long*)&P.counters[ORR_CNT_EPLOG_DROPPED], 1ull);
      }
    }
  }
  WSYNC();
  PT(14);
  store_robot(rec, S, lane, valid);
  store_anchor();
  if (valid) {   // the observation, in 16-byte pieces like the record (40 per robot)
    typedef f4 __attribute__((address_space(1))) * g4ptr;
    static_assert(ORR_OBS_DIM % 4 == 0, "16-byte pieces");
    const g4ptr od = (g4ptr)reinterpret_cast<f4*>(obs_out + (size_t)robot * ORR_OBS_DIM);
    const f4* os = reinterpret_cast<const f4*>(obs);
#pragma unroll
    for (int k = 0; k < (ORR_OBS_DIM / 4 + kLanes - 1) / kLanes; k++) { const int q = lane + k * kLanes; if (q < ORR_OBS_DIM / 4) od[q] = os[q]; }
  }
  PT(15);
  PT_FLUSH();
  // One counter update per WAVE (the compiler's own atomic combining is switched off, see _lib.HIPCC_FLAGS: it makes the issuing
  // lane wait for the returned value on the spot): thread 0 adds the wave's finished episodes to the launch accumulator and takes
  // a ticket for the wave's robots; the last wave to finish folds the launch's done count into the curriculum counter
  // (wrapper_env.py:82-83).
  const unsigned long long fin_mask = __ballot(valid && lane == 0 && reason != 0), val_mask = __ballot(valid && lane == 0);
  if (wtid == 0) {
    const unsigned long long nfin = (unsigned long long)__popcll(fin_mask), nval = (unsigned long long)__popcll(val_mask);
    if (nfin) atomicAdd((unsigned long long*)&P.counters[ORR_CNT_DONE_ACCUM], nfin);
    __threadfence();  // this wave's DONE_ACCUM / episode-log writes are visible before its ticket is
    const unsigned long long ticket = atomicAdd((unsigned long long*)&P.counters[ORR_CNT_TICKET], nval);
    if (ticket + nval == (unsigned long long)P.cfg.num_robots) {
      const unsigned long long nd = atomicExch((unsigned long long*)&P.counters[ORR_CNT_DONE_ACCUM], 0ull);
      atomicAdd((unsigned long long*)&P.counters[ORR_CNT_TOTAL_STEP_COUNT], nd);
      atomicAdd((unsigned long long*)&P.counters[ORR_CNT_TOTAL_TIMESTEPS], (unsigned long long)P.cfg.num_robots);
      atomicExch((unsigned long long*)&P.counters[ORR_CNT_TICKET], 0ull);
    }
  }
  PT_TIMELINE((long long)((fin_mask & 1ull) | ((fin_mask >> 15) & 2ull) | ((fin_mask >> 30) & 4ull) | ((fin_mask >> 45) & 8ull)));   // one bit per robot of the wave
  WT_STORE((long long)((fin_mask & 1ull) | ((fin_mask >> 15) & 2ull) | ((fin_mask >> 30) & 4ull) | ((fin_mask >> 45) & 8ull)));
}

namespace orr {
// launcher of the two-waves-per-SIMD instantiation, defined in the second translation unit (see the top of this file)
hipError_t launch_step_w2(const KParams& P, int waves, hipStream_t stream, const float* actions, float* obs, float* reward, uint8_t* done);
// launchers of the friction-anchor instantiations (env step; debug physics), defined in the third translation unit
// (orr_kernels_anchor.hip): kept out of the main unit, whose code generation for the DEFAULT kernels moves when further instantiations
// share its functions (round 5: +6 instructions per sub-step, +0.7 % run time with the anchor variants compiled alongside)
hipError_t launch_step_anchor(const KParams& P, int waves, hipStream_t stream, const float* actions, float* obs, float* reward, uint8_t* done);
hipError_t launch_physics_anchor(const KParams& P, int waves, hipStream_t stream, const float* torques, uint8_t* fall, int nsub);
}
#ifdef ORR_TU_STEP_ANCHOR
namespace orr {
hipError_t launch_step_anchor(const KParams& P, int waves, hipStream_t stream, const float* actions, float* obs, float* reward, uint8_t* done) {
  hipLaunchKernelGGL((orr_step_kernel<0, 1, true>), dim3(waves), dim3(64), 0, stream, P, actions, obs, reward, done, 0, ReplayArgs{});
  return hipGetLastError();
}
hipError_t launch_physics_anchor(const KParams& P, int waves, hipStream_t stream, const float* torques, uint8_t* fall, int nsub) {
  hipLaunchKernelGGL((orr_step_kernel<1, 1, true>), dim3(waves), dim3(64), 0, stream, P, torques, nullptr, nullptr, fall, nsub, ReplayArgs{});
  return hipGetLastError();
}
}  // namespace orr
#elif defined(ORR_TU_STEP_W2)
namespace orr {
hipError_t launch_step_w2(const KParams& P, int waves, hipStream_t stream, const float* actions, float* obs, float* reward, uint8_t* done) {
  hipLaunchKernelGGL((orr_step_kernel<0, 2>), dim3(waves), dim3(64), 0, stream, P, actions, obs, reward, done, 0, ReplayArgs{});
  return hipGetLastError();
}
}  // namespace orr
#else   // ---- everything below: main translation unit only ----


// Rollout boundary (agents/ppo_imitation.py:405-423): pack this rank's episode log into the fixed-size float64 payload of the
// all-gather -- [n_listed, total_timesteps, n_dropped, n_episodes, sum_ret, sum_len, ret[K], len[K]] -- and clear the log, in
// ONE launch of one workgroup (the log holds at most a few ten thousand (return, length) pairs).
__global__ __launch_bounds__(1024) void orr_eplog_pack_kernel(long long* counters, const float* ep_log, int cap_log, double total_timesteps,
                                                              int K, double* out) {
  __shared__ double red_r[1024], red_l[1024];
  const int tid = threadIdx.x;
  const long long cnt_all = counters[ORR_CNT_EPISODES], dropped = counters[ORR_CNT_EPLOG_DROPPED];
  const long long cnt = cnt_all < (long long)cap_log ? cnt_all : (long long)cap_log;   // logged (the rest was counted as dropped)
  const float2* log2 = reinterpret_cast<const float2*>(ep_log);
  double sr = 0.0, sl = 0.0;
  for (long long i0 = tid; i0 < cnt; i0 += 4096) {   // four independent loads in flight per thread
    float2 e[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const long long i = i0 + 1024 * u; e[u] = i < cnt ? log2[i] : make_float2(0.0f, 0.0f); }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long long i = i0 + 1024 * u;
      sr += (double)e[u].x; sl += (double)e[u].y;
      if (i < cnt && i < K) { out[6 + i] = (double)e[u].x; out[6 + K + i] = (double)e[u].y; }
    }
  }
  for (long long i = cnt + tid; i < K; i += 1024) { out[6 + i] = 0.0; out[6 + K + i] = 0.0; }
  red_r[tid] = sr; red_l[tid] = sl;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (tid < w) { red_r[tid] += red_r[tid + w]; red_l[tid] += red_l[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) {
    const long long listed = cnt < (long long)K ? cnt : (long long)K;
    out[0] = (double)listed; out[1] = total_timesteps; out[2] = (double)(dropped + (cnt - listed)); out[3] = (double)cnt;
    out[4] = red_r[0]; out[5] = red_l[0];
    counters[ORR_CNT_EPISODES] = 0; counters[ORR_CNT_EPLOG_DROPPED] = 0;   // every thread read them before the first barrier
  }
}

// Policy-free stress input of SURVEY.md section 8d (i): action = (reference joint pose one control step ahead, taken from the first
// target frame of the observation, mapped joint -> motor space with the robot's own table) - INIT_MOTOR_ANGLES + noise, clipped to
// the action space (imitation_runners.py:140-143).  One thread per (robot, motor), ONE launch whatever the mix of robot types: a
// heterogeneous batch needs a per-robot permutation, which as tensor operations is a copy + a batched GEMM (three launches).
__global__ __launch_bounds__(256) void orr_stress_actions_kernel(const DevTables* __restrict__ tab, const float* __restrict__ state,
                                                                 const float* __restrict__ obs, const float* __restrict__ noise,
                                                                 float* __restrict__ actions, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 12) return;
  const int r = i / 12, m = i - 12 * r;
  const int type = __float_as_int(state[(size_t)r * ORR_STATE_STRIDE + ORR_OFF_ROBOT_TYPE]);
  const ModelCold& C = tab->model[type].cold;
  const float tar = obs[(size_t)r * ORR_OBS_DIM + 84 + 7 + C.joint_of_motor[m]];
  const float a = (tar - C.motor_offset[m]) * C.motor_dir[m] - C.init_motor_angles[m] + noise[i];
  actions[i] = fminf(fmaxf(a, -6.2831853f), 6.2831853f);
}

// ================================================================================================
// C-ABI (include/openroborl_hip.h)
// ================================================================================================
struct orr_handle {
  orr_config cfg;
  int simds;          // SIMDs of the device (4 per CU): a batch of more waves than that runs the two-waves-per-SIMD variant of the step kernel
  int force_wpe;      // ORR_STEP_WAVES_PER_EU (0 = automatic)
  uint32_t anchor_types;   // bit t = robot type t has orr_model::friction_anchor: launches run the ANCHOR variant of the step kernel
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

// The ABI carries times as float32; the reference computes with the DECIMAL constants of its sources in float64 (FrameDuration
// 0.01667, sim_time_step 0.001).  The shortest decimal (<= 7 significant digits) that rounds to the given float, else its exact value.
static double dec7(float x) {
  char b[40];
  snprintf(b, sizeof(b), "%.7g", (double)x);
  const double d = strtod(b, nullptr);
  return (float)d == x ? d : (double)x;
}

extern "C" {

#ifndef ORR_SOURCE_HASH
#define ORR_SOURCE_HASH "unknown"
#endif
const char* orr_last_error(void) { return g_err; }
int32_t orr_abi_version(void) { return ORR_ABI_VERSION; }
// "ORR_SRC_HASH=<hex>" is also what the host loader scans the FILE for (openroborl_amd/_lib.py: a stale-build check that must not dlopen)
static const char g_src_hash[] = "ORR_SRC_HASH=" ORR_SOURCE_HASH;
const char* orr_source_hash(void) { return g_src_hash + 13; }
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
  // the quaternion update uses series for sin / cos of half the rotation of one sub-step (exact to float precision below 0.2 rad):
  // |w| <= sqrt(3) max_coord_velocity after the coordinate-velocity clamp
  if (!(cfg->max_coord_velocity > 0.0f) || !(cfg->sim_dt > 0.0f) || 0.5f * 1.7320508f * cfg->max_coord_velocity * cfg->sim_dt >= 0.2f)
    return fail(-1, "orr_create: max_coord_velocity * sim_dt too large (the base may turn at most 0.4 rad per sub-step)");
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
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    h->simds = 4 * cus;
    const char* f = getenv("ORR_STEP_WAVES_PER_EU");
    h->force_wpe = (f && (f[0] == '1' || f[0] == '2') && f[1] == 0) ? f[0] - '0' : 0;
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

int32_t orr_set_seed(orr_handle* h, uint64_t seed) {
  if (!h) return fail(-1, "orr_set_seed: null handle");
  h->cfg.seed = seed;  // read by the next launch: the RNG is counter-based, keyed by (seed, robot index, episode index)
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
  // built in a local entry: a rejected model leaves the handle (host table, device table, anchor bit) exactly as it was
  DevModel d;
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
  ModelCold& Cd = d.cold;
  for (int i = 0; i < 3; i++) Cd.init_pos[i] = m->init_pos[i];
  for (int i = 0; i < 4; i++) H.init_quat[i] = m->init_quat[i];
  for (int i = 0; i < 12; i++) {
    const int j = m->joint_of_motor[i];
    Cd.init_motor_angles[i] = m->init_motor_angles[i];
    Cd.motor_dir[i] = m->motor_dir[i];
    Cd.motor_offset[i] = m->motor_offset[i];
    Cd.joint_of_motor[i] = j;
    Cd.kp[i] = m->kp[i];
    Cd.kd[i] = m->kd[i];
    if (fabsf(fabsf(m->motor_dir[i]) - 1.0f) > 1e-6f) return fail(-1, "orr_set_model: motor_dir must be +1 or -1");
    H.jdir[j] = m->motor_dir[i] * axsgn[j];
    H.joff[j] = m->motor_offset[i];
    Cd.tau_sign[j] = axsgn[j];
    Cd.tau_sign_motor[i] = axsgn[j];
    Cd.default_joints[i] = (m->init_motor_angles[i] + m->motor_offset[i]) * m->motor_dir[i];
  }
  for (int j = 0; j < 12; j++) {
    for (int k = 0; k < 3; k++) { Cd.link_com[j][k] = m->link_com[j][k]; H.joint_pos[j][k] = m->joint_pos[j][k]; }
    // limits are given for the kinematic angle; the internal angle is axis_sign times it
    H.joint_lo[j] = axsgn[j] > 0 ? m->joint_lo[j] : -m->joint_hi[j];
    H.joint_hi[j] = axsgn[j] > 0 ? m->joint_hi[j] : -m->joint_lo[j];
    if (!(H.joint_hi[j] - H.joint_lo[j] >= 2.0f * h->cfg.limit_activation))
      return fail(-1, "orr_set_model: joint range must be at least 2 * limit_activation");
  }
  for (int l = 0; l < 4; l++)
    for (int k = 0; k < 3; k++) { H.toe_pos[l][k] = m->toe_pos[l][k]; H.lower_com[l][k] = m->lower_com[l][k]; H.shank_pos[l][k] = m->shank_pos[l][k]; }
  if (!(m->shank_radius >= 0.0f)) return fail(-1, "orr_set_model: shank_radius must be >= 0");
  H.toe_radius = m->toe_radius;
  H.shank_radius = m->shank_radius;
  Cd.foot_friction = m->foot_friction;
  Cd.friction_anchor = m->friction_anchor != 0;
  if (m->friction_anchor && !(h->cfg.friction_erp >= 0.0f && h->cfg.friction_erp <= 1.0f)) return fail(-1, "orr_set_model: friction_anchor needs 0 <= orr_config::friction_erp <= 1");
  Cd.num_fall = m->num_fall_proxies;
  if (m->contact_stiffness > 0.0f) {
    if (!(m->contact_damping >= 0.0f)) return fail(-1, "orr_set_model: contact_damping must be >= 0");
    // btMultiBodyConstraintSolver::setupMultiBodyContactConstraint: cfm = 1 / (dt k + d), erp = dt k / (dt k + d); cfm *= 1 / dt
    const double dtk = (double)h->cfg.sim_dt * m->contact_stiffness, denom = dtk + m->contact_damping;
    H.contact_cfm = (float)(1.0 / denom / h->cfg.sim_dt);
    H.contact_erp_dt = (float)(dtk / denom / h->cfg.sim_dt);
  } else {
    H.contact_cfm = 0.0f;
    H.contact_erp_dt = h->cfg.contact_erp / h->cfg.sim_dt;
  }
  for (int i = 0; i < ORR_MAX_FALL_PROXIES; i++) {
    Cd.fall_body[i] = m->fall_body[i];
    Cd.fall_radius[i] = m->fall_radius[i];
    for (int k = 0; k < 3; k++) Cd.fall_pos[i][k] = m->fall_pos[i][k];
  }
  Cd.mass[0] = m->base_mass;
  Cd.group[0] = 0;
  for (int k = 0; k < 6; k++) { Cd.inertia[0][k] = m->base_inertia[k]; Cd.inertia_pa[0][k] = 0.0f; }
  for (int j = 0; j < 12; j++) {
    Cd.mass[j + 1] = m->link_mass[j];
    Cd.group[j + 1] = m->link_group[j];
    for (int k = 0; k < 6; k++) { Cd.inertia[j + 1][k] = m->link_inertia[j][k]; Cd.inertia_pa[j + 1][k] = m->link_inertia_pa[j][k]; }
  }
  HIPCHK(hipMemcpy(&h->tab_dev->model[robot_type], &d, sizeof(DevModel), hipMemcpyHostToDevice), "orr_set_model: hipMemcpy");
  h->tab_host.model[robot_type] = d;
  if (m->friction_anchor) h->anchor_types |= 1u << robot_type; else h->anchor_types &= ~(1u << robot_type);
  return 0;
}

int32_t orr_set_motion(orr_handle* h, int32_t clip_id, const float* frames_dev, const float* frame_vels_dev, int32_t num_frames,
                       double frame_dt, int32_t clip_flags, const float cycle_delta[4]) {
  if (!h || !frames_dev || !frame_vels_dev || !cycle_delta) return fail(-1, "orr_set_motion: null argument");
  if (clip_id < 0 || clip_id >= ORR_MAX_CLIPS) return fail(-1, "orr_set_motion: clip_id out of range");
  if (num_frames < 2) return fail(-1, "orr_set_motion: need at least 2 frames");
  if (!(frame_dt > 0.0)) return fail(-1, "orr_set_motion: Frame duration must be positive.");
  DevClip c;
  c.frames = frames_dev; c.vels = frame_vels_dev; c.F = num_frames; c.flags = clip_flags;
  c.dt_d = frame_dt; c.dur_d = c.dt_d * (num_frames - 1); c.sim_dt_d = dec7(h->cfg.sim_dt);
  c.cdp[0] = cycle_delta[0]; c.cdp[1] = cycle_delta[1]; c.cdp[2] = cycle_delta[2]; c.cdh = cycle_delta[3];
  h->tab_host.clip[clip_id] = c;
  HIPCHK(hipMemcpy(&h->tab_dev->clip[clip_id], &c, sizeof(DevClip), hipMemcpyHostToDevice), "orr_set_motion: hipMemcpy");
  return 0;
}

int32_t orr_bind(orr_handle* h, void* state_dev, int64_t* counters_dev, float* ep_log_dev, int32_t ep_log_capacity) {
  if (!h || !state_dev || !counters_dev) return fail(-1, "orr_bind: null argument (state and counters are required)");
  if (((uintptr_t)state_dev & 15u) != 0) return fail(-1, "orr_bind: the state buffer must be 16-byte aligned (records move in 16-byte pieces)");
  h->state = (float*)state_dev;
  h->counters = (long long*)counters_dev;
  h->ep_log = ep_log_dev;
  h->ep_log_cap = ep_log_dev ? ep_log_capacity : 0;
  return 0;
}

#ifdef ORR_WAVE_TIMELINE
static long long* g_wave_times_dev = nullptr;
static int g_wave_times_cap = 0;
#endif
static KParams make_params(const orr_handle* h) {
  KParams P;
  P.cfg = h->cfg;
  for (int i = 0; i < 3; i++) { P.fb[i] = h->fb[i]; P.fa[i] = h->fa[i]; }
  P.tab = h->tab_dev;
  P.state = h->state;
  P.counters = h->counters;
  P.ep_log = h->ep_log;
  P.ep_log_cap = h->ep_log_cap;
  P.simds = h->simds;
  P.anchor_on = h->anchor_types != 0;
#ifdef ORR_WAVE_TIMELINE
  P.wave_times = g_wave_times_dev;
#endif
  return P;
}

int32_t orr_reset(orr_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!h || !h->state) return fail(-1, "orr_reset: handle not bound");
  hipLaunchKernelGGL(orr_reset_kernel, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h), mask_dev, obs_dev,
                     (const float*)nullptr);
  HIPCHK(hipGetLastError(), "orr_reset: launch");
  return 0;
}

int32_t orr_step(orr_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream) {
  if (!h || !h->state) return fail(-1, "orr_step: handle not bound");
  if (!actions_dev || !obs_dev || !reward_dev || !done_dev) return fail(-1, "orr_step: null buffer");
  if (((uintptr_t)obs_dev & 15u) != 0) return fail(-1, "orr_step: the observation buffer must be 16-byte aligned (it is written in 16-byte pieces)");
  const int waves = (h->cfg.num_robots + kRPW - 1) / kRPW;
  const bool two = h->force_wpe ? h->force_wpe == 2 : waves > h->simds;
  if (h->anchor_types) {   // some robot type has friction anchors: the ANCHOR variant (one wave per SIMD, any batch size)
    HIPCHK(launch_step_anchor(make_params(h), waves, (hipStream_t)stream, actions_dev, obs_dev, reward_dev, done_dev), "orr_step: launch (friction anchors)");
  } else if (two) {
    HIPCHK(launch_step_w2(make_params(h), waves, (hipStream_t)stream, actions_dev, obs_dev, reward_dev, done_dev), "orr_step: launch (two waves per SIMD)");
  } else {
    // <0> = <0, ORR_WAVES_PER_EU>: one wave per SIMD in the shipped build; development builds (-DORR_WAVES_PER_EU=2 with the timers of
    // this translation unit, tools/wave_pairing.py) get their instrumented two-wave kernel through this path with ORR_STEP_WAVES_PER_EU=1
    constexpr int wpb = step_wpb<0, ORR_WAVES_PER_EU>();
    hipLaunchKernelGGL((orr_step_kernel<0>), dim3((waves + wpb - 1) / wpb), dim3(64 * wpb), 0, (hipStream_t)stream, make_params(h), actions_dev, obs_dev,
                       reward_dev, done_dev, 0, ReplayArgs{});
    HIPCHK(hipGetLastError(), "orr_step: launch");
  }
  return 0;
}

// parity / debug entry point (not part of the drop-in surface): nsub physics sub-steps with fixed motor torques
int32_t orr_debug_physics(orr_handle* h, const float* torques_dev, uint8_t* fall_dev, int32_t nsub, void* stream) {
  if (!h || !h->state || !torques_dev) return fail(-1, "orr_debug_physics: bad argument");
  if (h->anchor_types) {
    HIPCHK(launch_physics_anchor(make_params(h), (h->cfg.num_robots + kRPW - 1) / kRPW, (hipStream_t)stream, torques_dev, fall_dev, nsub), "orr_debug_physics: launch (friction anchors)");
    return 0;
  }
  hipLaunchKernelGGL(orr_step_kernel<1>, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h), torques_dev,
                     nullptr, nullptr, fall_dev, nsub, ReplayArgs{});
  HIPCHK(hipGetLastError(), "orr_debug_physics: launch");
  return 0;
}

int32_t orr_episode_stats(orr_handle* h, double total_timesteps, int32_t capacity, double* out_dev, void* stream) {
  if (!h || !h->counters || !h->ep_log || !out_dev) return fail(-1, "orr_episode_stats: needs a bound episode log and an output buffer");
  if (capacity < 1) return fail(-1, "orr_episode_stats: capacity must be >= 1");
  hipLaunchKernelGGL(orr_eplog_pack_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, h->counters, h->ep_log, h->ep_log_cap, total_timesteps,
                     (int)capacity, out_dev);
  HIPCHK(hipGetLastError(), "orr_episode_stats: launch");
  return 0;
}

// measurement input (not part of the drop-in surface): the policy-free stress actions of SURVEY.md section 8d (i), see the kernel
int32_t orr_stress_actions(orr_handle* h, const float* obs_dev, const float* noise_dev, float* actions_dev, void* stream) {
  if (!h || !h->state) return fail(-1, "orr_stress_actions: handle not bound");
  if (!obs_dev || !noise_dev || !actions_dev) return fail(-1, "orr_stress_actions: null buffer");
  const int n = h->cfg.num_robots;
  hipLaunchKernelGGL(orr_stress_actions_kernel, dim3((n * 12 + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->tab_dev, h->state, obs_dev,
                     noise_dev, actions_dev, n);
  HIPCHK(hipGetLastError(), "orr_stress_actions: launch");
  return 0;
}

// parity / debug entry points (not part of the drop-in surface): the env step / reset with the physics engine, the link
// positions, the contact flag and the random draws REPLAYED from buffers (see ReplayArgs in orr_device.h)
int32_t orr_debug_replay_step(orr_handle* h, const float* actions_dev, const float* traj_dev, const float* eff_dev, const uint8_t* fall_dev,
                              float* obs_dev, float* reward_dev, uint8_t* done_dev, float* tau_out_dev, void* stream) {
  if (!h || !h->state) return fail(-1, "orr_debug_replay_step: handle not bound");
  if (!actions_dev || !traj_dev || !eff_dev || !fall_dev || !obs_dev || !reward_dev || !done_dev || !tau_out_dev)
    return fail(-1, "orr_debug_replay_step: null buffer");
  if (h->cfg.flags & ORR_FLAG_AUTO_RESET) return fail(-1, "orr_debug_replay_step: needs a handle without ORR_FLAG_AUTO_RESET");
  ReplayArgs rp{traj_dev, eff_dev, fall_dev, tau_out_dev, nullptr};
  hipLaunchKernelGGL(orr_step_kernel<2>, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h), actions_dev,
                     obs_dev, reward_dev, done_dev, 0, rp);
  HIPCHK(hipGetLastError(), "orr_debug_replay_step: launch");
  return 0;
}
int32_t orr_debug_replay_reset(orr_handle* h, const float* uniforms_dev, float* obs_dev, void* stream) {
  if (!h || !h->state || !uniforms_dev) return fail(-1, "orr_debug_replay_reset: bad argument");
  hipLaunchKernelGGL(orr_reset_kernel, dim3((h->cfg.num_robots + kRPW - 1) / kRPW), dim3(64), 0, (hipStream_t)stream, make_params(h),
                     (const uint8_t*)nullptr, obs_dev, uniforms_dev);
  HIPCHK(hipGetLastError(), "orr_debug_replay_reset: launch");
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

#if defined(ORR_COUNT_DUAL_CONTACT)
// development aid: read (and optionally clear) the toe / shank contact counters of the -DORR_COUNT_DUAL_CONTACT build
int orr_debug_dual_contact(unsigned long long* out8, int reset) {
  HIPCHK(hipDeviceSynchronize(), "orr_debug_dual_contact: sync");
  HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dual_contact), 8 * sizeof(unsigned long long)), "orr_debug_dual_contact: read");
  if (reset) {
    unsigned long long z[8] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dual_contact), z, sizeof(z)), "orr_debug_dual_contact: clear");
  }
  return 0;
}
#endif
#ifdef ORR_WAVE_TIMELINE
// development aid: per-wave timeline of the last step launch of either variant (4 words per wave: realtime start / end in 100 MHz ticks,
// shader cycles, bits 0..7 mask of the robots that finished an episode | bits 8..39 HW_ID (wave slot, SIMD, CU, SE) | bits 40..43 XCC id).
// The first call (waves > 0, out may be null) allocates the device buffer; launches after it are recorded.
int orr_debug_wave_times(long long* out, int waves) {
  HIPCHK(hipDeviceSynchronize(), "orr_debug_wave_times: sync");
  if (waves > g_wave_times_cap) {
    if (g_wave_times_dev) HIPCHK(hipFree(g_wave_times_dev), "orr_debug_wave_times: free");
    HIPCHK(hipMalloc((void**)&g_wave_times_dev, (size_t)waves * 4 * sizeof(long long)), "orr_debug_wave_times: alloc");
    HIPCHK(hipMemset(g_wave_times_dev, 0, (size_t)waves * 4 * sizeof(long long)), "orr_debug_wave_times: clear");
    g_wave_times_cap = waves;
    return 0;
  }
  if (out) HIPCHK(hipMemcpy(out, g_wave_times_dev, (size_t)waves * 4 * sizeof(long long), hipMemcpyDeviceToHost), "orr_debug_wave_times: read");
  return 0;
}
#endif
#ifdef ORR_PHASE_TIMERS
// development aid: read (and optionally clear) the per-phase cycle totals of the instrumented wave
int orr_debug_phase_cycles(long long* out40, int reset) {
  HIPCHK(hipDeviceSynchronize(), "orr_debug_phase_cycles: sync");
  HIPCHK(hipMemcpyFromSymbol(out40, HIP_SYMBOL(g_phase_cycles), kPhaseSlots * sizeof(long long)), "orr_debug_phase_cycles: read");
  if (reset) {
    long long z[kPhaseSlots] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)), "orr_debug_phase_cycles: clear");
  }
  return 0;
}
// development aid: per-wave timeline of the last step launch (4 words per wave: realtime start / end in 100 MHz ticks, shader cycles,
// bits 0..7 mask of the robots that finished an episode, bits 8..39 HW_ID register of the wave (SIMD / CU / SE), bits 40..43 XCC id)
int orr_debug_wave_phases(long long* out, int waves) {
  HIPCHK(hipDeviceSynchronize(), "orr_debug_wave_phases: sync");
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_phases), (size_t)waves * 40 * sizeof(long long)), "orr_debug_wave_phases: read");
  return 0;
}
int orr_debug_wave_timeline(long long* out, int waves) {
  HIPCHK(hipDeviceSynchronize(), "orr_debug_wave_timeline: sync");
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_timeline), (size_t)waves * 4 * sizeof(long long)), "orr_debug_wave_timeline: read");
  return 0;
}
#endif

}  // extern "C"
#endif  // main translation unit
