/* openroborl_learner.h -- C-ABI of the hand-written pieces of the PPO update ("PPO learner in
 * PyTorch-ROCm"; host side: openroborl_amd/learner_hip.py).
 *
 * The reference builds the update as a TF1 graph (agents/ppo_imitation.py:156-258: clipped surrogate + value loss on the two
 * 160 -> 512 -> 256 -> {12, 1} ReLU networks of agents/imitation_policies.py:44-51) and applies it with MpiAdam
 * (stable_baselines/common/mpi_adam.py:40-62).  Here the dense contractions stay library GEMMs (hipBLASLt through torch); what the
 * autograd graph spends around them - ~70 elementwise / reduction launches per minibatch, more time than the GEMMs themselves - is
 * these launches:
 *   orr_ppo_head       loss terms, d loss / d mean, d loss / d value, bias gradients of the two output layers
 *   orr_head_backward  gradient through an output layer (fan-out 12 or 1) + ReLU mask + bias gradient of the layer below + the
 *                      output layer's weight gradient
 *   orr_colsum_finish  the column sums behind all bias gradients of a minibatch in one launch
 *   orr_relu_backward  ReLU mask in place + bias gradient
 *   orr_adam_step      Adam on the flat parameter vector, step counter on the device (the whole update replays as one hipGraph)
 * All pointers are device pointers, float32, row-major and 16-byte aligned; calls are asynchronous on the caller's stream; every
 * reduction runs in a fixed order (no float atomics: the same inputs give the same bits).  Return 0 or a negative code with text in
 * orr_last_error() (openroborl_hip.h).
 */
#ifndef OPENROBORL_LEARNER_H
#define OPENROBORL_LEARNER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Floats of scratch one call below needs for a minibatch of m rows and a layer of c columns.  Calls that finish their sums
 * themselves (gb given) can share one buffer: they run one after the other on the stream.  Deferred calls (gb = NULL) leave their
 * per-workgroup partial sums in the buffer for orr_colsum_finish and need one buffer each. */
int64_t orr_learner_workspace_floats(int32_t m, int32_t c);
/* Rows of partial sums a deferred call leaves for m samples. */
int32_t orr_learner_partial_rows(int32_t m);

#define ORR_PPO_BATCH_COLS 16
/* Loss head for a minibatch of m samples (agents/ppo_imitation.py:196-214: ratio = exp(logp - old logp), surrogate
 * -mean(min(ratio A, clip(ratio, 1 - c, 1 + c) A)), value loss mean((v - tdlamret)^2); fixed-std diagonal Gaussian,
 * agents/imitation_policies.py:96-107).
 *   mean  [m][12], value [m]    outputs of the two networks
 *   batch [m][16]               per sample: raw action (12), old log-probability, advantage, TD(lambda) return, unused
 *   g_mean [m][12], g_value [m] d loss / d mean, d loss / d value for loss = surrogate + vf_coef * value loss (the means over m included)
 *   gb_mean [12], gb_value [1]  column sums of g_mean / g_value = the output layers' bias gradients
 *   stats [2]                   surrogate, value loss (means over the minibatch) */
int32_t orr_ppo_head(const float* mean, const float* value, const float* batch, int32_t m, float std, float clip, float vf_coef,
                     float* g_mean, float* g_value, float* gb_mean, float* gb_value, float* stats, float* workspace, void* stream);

/* g [m][c] (gradient w.r.t. a ReLU layer's output h [m][c]) becomes the gradient w.r.t. its pre-activation, in place: g *= (h > 0);
 * gb [c] = its column sums (the layer's bias gradient).  c: a multiple of 4 with 1024 % c == 0.
 * gb = NULL defers the sum: workspace then holds [orr_learner_partial_rows(m)][c] partial sums for orr_colsum_finish. */
int32_t orr_relu_backward(float* g, const float* h, int32_t m, int32_t c, float* gb, float* workspace, void* stream);

/* The layer below an OUTPUT layer of fan-out k = 12 or 1 (w [c][k] as stored, gy [m][k] = d loss / d output): no GEMM with N = k.
 *   gz [m][c] = (gy . w^T) * (h > 0)    gradient w.r.t. the pre-activation of the hidden layer h [m][c]
 *   gb [c]    = column sums of gz       that layer's bias gradient
 *   gw [c][k] = h^T . gy                the output layer's weight gradient
 * gb = gw = NULL defers both sums: workspace holds [rows][c] partials of gb followed by [rows][c * k] partials of gw. */
int32_t orr_head_backward(const float* gy, int32_t k, const float* w, const float* h, int32_t m, int32_t c, float* gz, float* gb, float* gw,
                          float* workspace, void* stream);

/* out [cols] = sum over `rows` rows of partials [rows][cols], for up to 12 jobs in one launch, in a fixed order (e.g. the six bias
 * gradients, the two output-layer weight gradients and the four split weight gradients of a minibatch). */
#define ORR_COLSUM_MAX_JOBS 12
typedef struct orr_colsum_job {
  const float* partials;
  float* out;
  int32_t rows;
  int32_t cols;
} orr_colsum_job;
int32_t orr_colsum_finish(const orr_colsum_job* jobs, int32_t n_jobs, void* stream);

/* One Adam step on n parameters; t = state[0] + 1, the gradient is multiplied by grad_scale first (1 / world size after an
 * all-reduce sum: mpi_adam.py:51-53).
 *   flags = 0: torch.optim.Adam's arithmetic,  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 *   flags = ORR_ADAM_MPI_EPSILON: the reference's MpiAdam (stable_baselines/common/mpi_adam.py:55-62),
 *                                 p -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)      (eps not scaled by sqrt(1 - b2^t))
 * state [2] int32, zero-initialised by the caller once: state[0] counts the steps taken (incremented by the launch itself, so that a
 * captured launch can be replayed), state[1] is internal. */
#define ORR_ADAM_MPI_EPSILON 1
int32_t orr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float grad_scale, int32_t flags, int32_t* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif
