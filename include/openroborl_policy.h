/* openroborl_policy.h -- C-ABI of the fused policy / value forward pass ("policy inference
 * in the loop on device").
 *
 * Replaces, for all N robots of a shard in ONE launch, what the reference does robot by robot with batch 1:
 *   PPOImitation runner: `policy.step(ob.reshape(-1, *ob.shape))` per robot   (agents/imitation_runners.py:88-92)
 *   ImitationPolicy: actor 160 -> 512 -> 256 -> 12 and critic 160 -> 512 -> 256 -> 1, ReLU (run.py:101-105;
 *   agents/imitation_policies.py:44-51), fixed-std diagonal Gaussian, action = mean + std * N(0, 1)
 *   (agents/imitation_policies.py:96-107), clipped to the action bounds by the runner (imitation_runners.py:140-143).
 *
 * Arithmetic: f32 inputs, f32 accumulation on the matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf
 * chain), i.e. the precision of the reference's float32 TensorFlow graph.  All pointers are device pointers; calls are
 * asynchronous on the caller's stream; return 0 or a negative code with text in orr_last_error() (openroborl_hip.h).
 */
#ifndef OPENROBORL_POLICY_H
#define OPENROBORL_POLICY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORR_POLICY_OBS_DIM 160
#define ORR_POLICY_H0 512
#define ORR_POLICY_H1 256
#define ORR_POLICY_ACT_DIM 12

/* Number of floats of the packed ("fragment-major") image of a K x N weight matrix: N is padded to a multiple of 16,
 * K must be a multiple of 16. */
int64_t orr_policy_packed_size(int32_t k, int32_t n);

/* Repack a row-major [K][N] float32 weight matrix (the stable-baselines `model/<net>_fc<i>/w:0` layout, fan_in x fan_out)
 * into the layout the forward kernel streams: for output-column tile nt (16 columns), k-group kg (16 rows), lane l, j < 4:
 *   out[((nt * K/16 + kg) * 64 + l) * 4 + j] = W[16 kg + 4 j + (l >> 4)][16 nt + (l & 15)]      (0 beyond column N)
 * so that one 16-byte load per lane yields its B operands of four consecutive MFMA k-steps. */
int32_t orr_policy_pack(const float* w_dev, int32_t k, int32_t n, float* out_dev, void* stream);

typedef struct orr_policy_net {
  /* packed weights (orr_policy_pack) and plain biases of the actor ("pi") and the critic ("vf") */
  const float* w0_pi; const float* b0_pi;   /* 160 x 512 */
  const float* w1_pi; const float* b1_pi;   /* 512 x 256 */
  const float* w2_pi; const float* b2_pi;   /* 256 x 12  (bias: 12 floats) */
  const float* w0_vf; const float* b0_vf;   /* 160 x 512 */
  const float* w1_vf; const float* b1_vf;   /* 512 x 256 */
  const float* w2_vf; const float* b2_vf;   /* 256 x 1   (bias: 1 float) */
} orr_policy_net;

/* One forward pass for n robots.
 *   obs      [n][160]  observation (the env's obs tensor)
 *   noise    [n][12]   standard normal samples, or NULL for the deterministic action (mean)
 *   action   [n][12]   clip(mean + std * noise, -clip, +clip)    -> what env.step() takes
 *   raw      [n][12]   mean + std * noise (unclipped; what the learner's log-probability uses), may be NULL
 *   value    [n]       critic output, may be NULL
 *   mean     [n][12]   actor output, may be NULL */
int32_t orr_policy_forward(const orr_policy_net* net, const float* obs, int32_t n, const float* noise, float std, float clip,
                           float* action, float* raw, float* value, float* mean, void* stream);

/* Per-robot GAE(lambda) over a [T][N] rollout segment + per-robot advantage standardisation, one launch
 * (reference: add_vtarg_and_adv, agents/ppo_imitation.py:68-93, and the per-robot normalisation :329-338; device
 * layout and the deliberate use of each robot's own done flags: openroborl_amd/rollout.py).
 *   rewards, vpred [T][N] float32; dones [T][N] uint8 (1 = the episode ended at that step);
 *   bootstrap [N] value after the last step (NULL = 0, the reference's choice)
 *   adv [T][N]: advantages, standardised per robot ((a - mean) / (population std + eps)) when normalize != 0
 *   ret [T][N]: TD(lambda) targets = raw advantage + vpred */
int32_t orr_gae(const float* rewards, const float* vpred, const uint8_t* dones, const float* bootstrap, int32_t t, int32_t n,
                float gamma, float lam, int32_t normalize, float eps, float* adv, float* ret, void* stream);

/* The same with flags.  ORR_GAE_LEGACY_INDEX reproduces the reference's recursion for num_robot > 1 bit for bit in its INDEXING:
 * add_vtarg_and_adv reads `episode_starts[(step*num_robot+i) + (1+i)]` (agents/ppo_imitation.py:88), i.e. the episode-start flag of
 * robot (2i+1) mod N at step t + (2i+1)/N instead of robot i's own at t+1 (identical only for N = 1).  For users who compare learning
 * curves with the reference at its default num_robot: 2.  first_starts [N] uint8 = episode_starts of the segment's first step
 * (NULL = all 1: a segment that begins with fresh episodes). */
#define ORR_GAE_NORMALIZE 1
#define ORR_GAE_LEGACY_INDEX 2
int32_t orr_gae_flags(const float* rewards, const float* vpred, const uint8_t* dones, const uint8_t* first_starts, const float* bootstrap,
                      int32_t t, int32_t n, float gamma, float lam, int32_t flags, float eps, float* adv, float* ret, void* stream);

#ifdef __cplusplus
}
#endif
#endif
