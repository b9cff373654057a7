// orr_policy.hip -- fused actor / critic forward pass for all robots of a shard (include/openroborl_policy.h).
//
// Reference: one `policy.step(ob)` per robot with batch 1 (agents/imitation_runners.py:88-92) on the MlpPolicy-style nets
// of agents/imitation_policies.py:44-51,96-107 (run.py:101-105: layers [512, 256], ReLU).  Here one launch does
//   h0 = relu(obs W0 + b0);  h1 = relu(h0 W1 + b1);  out = h1 W2 + b2        for the actor and the critic,
//   action = clip(mean + std * noise)
// on the matrix cores with f32 inputs and f32 accumulation (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain).
//
// Mapping: a workgroup (4 wavefronts, one per SIMD) owns 16 robots (rows).  N = 4096 robots -> 256 workgroups = one per CU.
// The 16 x 160 observation tile and the hidden activations (16 x 1024, 16 x 512) live in LDS; the weights are streamed
// from L2 in a fragment-major packing (orr_policy_pack) so that one 16-byte load per lane is the B operand of four
// consecutive k-steps; each wave owns a quarter of every layer's output columns and keeps its accumulators in registers.
// MFMA count per wave: 640 (layer 0) + 1024 (layer 1) + 64 (heads) of 32 cycles each = 23 us at 2.4 GHz; the weight
// stream is 1.7 MB per workgroup out of L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openroborl_policy.h"

int orr_fail(int code, const char* msg, hipError_t e);  // orr_kernels.hip

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kObs = ORR_POLICY_OBS_DIM, kH0 = ORR_POLICY_H0, kH1 = ORR_POLICY_H1, kAct = ORR_POLICY_ACT_DIM;
constexpr int kRows = 16;                       // robots per workgroup
constexpr int kObsStride = kObs + 1;            // odd strides: the 16 rows of an A fragment fall into distinct LDS banks
constexpr int kH0Stride = 2 * kH0 + 1;          // actor | critic hidden 0
constexpr int kH1Stride = 2 * kH1 + 1;          // actor | critic hidden 1

__global__ void pack_kernel(const float* __restrict__ w, int K, int N, float* __restrict__ out, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 3), l = (int)((i >> 2) & 63);
  const long long t = i >> 8;                   // nt * (K / 16) + kg
  const int kgs = K / 16, kg = (int)(t % kgs), nt = (int)(t / kgs);
  const int k = 16 * kg + 4 * j + (l >> 4), c = 16 * nt + (l & 15);
  out[i] = c < N ? w[(size_t)k * N + c] : 0.0f;
}

struct Params {
  orr_policy_net net;
  const float* obs;
  const float* noise;
  float* action;
  float* raw;
  float* value;
  float* mean;
  int n;
  float std, clip;
};

// acc[t] += A(16 x 16 k-group kg, from LDS) * B(tile t, k-group kg): four k-steps of v_mfma_f32_16x16x4_f32 per tile.
// A fragment of k-step j: lane l holds A[l & 15][16 kg + 4 j + (l >> 4)];  B fragment: element j of the packed float4.
template <int TILES>
__device__ __forceinline__ void mma_group(f32x4 (&acc)[TILES], const float (&a)[4], const f32x4 (&b)[TILES]) {
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int t = 0; t < TILES; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[t][j], acc[t], 0, 0, 0);
}

// One layer for this wave: TILES column tiles starting at tile index `nt0` of a packed K x (16 * ntiles) matrix.
// `act` points at column 0 of the A operand in LDS (row stride `stride`).  The weight stream is software-pipelined
// DEPTH k-groups deep (a k-group = 4 MFMA k-steps = one 16-byte load per lane and tile); the A fragments of the next
// k-group are read from LDS before the MFMAs of the current one.  K / 16 must be a multiple of DEPTH.
template <int TILES, int K, int DEPTH>
__device__ __forceinline__ void layer(f32x4 (&acc)[TILES], const float* __restrict__ wp, int nt0, const float* act, int stride, int lane) {
  constexpr int KG = K / 16;
  static_assert(KG % DEPTH == 0, "k-groups must be a multiple of the pipeline depth");
  const f32x4* w4 = reinterpret_cast<const f32x4*>(wp) + (size_t)nt0 * KG * 64 + lane;   // tile t, k-group g: w4[(t * KG + g) * 64]
  const float* arow = act + (lane & 15) * stride + (lane >> 4);                             // k-group g, k-step j: arow[16 g + 4 j]
  f32x4 b[DEPTH][TILES];
  float a[2][4];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++)
#pragma unroll
    for (int t = 0; t < TILES; t++) b[d][t] = w4[(t * KG + d) * 64];
#pragma unroll
  for (int j = 0; j < 4; j++) a[0][j] = arow[4 * j];
#pragma unroll 1
  for (int g0 = 0; g0 < KG; g0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const int g = g0 + d;
      if (g + DEPTH - 1 < KG) {
#pragma unroll
        for (int t = 0; t < TILES; t++) b[(d + DEPTH - 1) % DEPTH][t] = w4[(t * KG + g + DEPTH - 1) * 64];
      }
      if (g + 1 < KG) {
#pragma unroll
        for (int j = 0; j < 4; j++) a[(d + 1) & 1][j] = arow[16 * (g + 1) + 4 * j];
      }
      mma_group<TILES>(acc, a[d & 1], b[d]);
    }
  }
}

// C/D layout of the 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + register
template <int TILES>
__device__ __forceinline__ void store_relu(const f32x4 (&acc)[TILES], const float* __restrict__ bias, int nt0, float* dst, int stride, int lane) {
  const int col = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const float b = bias[16 * (nt0 + t) + col];
#pragma unroll
    for (int r = 0; r < 4; r++) dst[(r0 + r) * stride + 16 * (nt0 + t) + col] = fmaxf(acc[t][r] + b, 0.0f);
  }
}

__global__ __launch_bounds__(256) void forward_kernel(Params P) {
  __shared__ float s_obs[kRows * kObsStride];
  __shared__ float s_h0[kRows * kH0Stride];
  __shared__ float s_h1[kRows * kH1Stride];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = blockIdx.x * kRows;
  // observation tile (rows beyond n read as zero)
  for (int i = tid; i < kRows * kObs; i += 256) {
    const int r = i / kObs, c = i - r * kObs;
    s_obs[r * kObsStride + c] = (row0 + r) < P.n ? P.obs[(size_t)(row0 + r) * kObs + c] : 0.0f;
  }
  __syncthreads();
  // ---- layer 0: 160 -> 512, actor and critic; this wave: column tiles [8 wave, 8 wave + 8) of each ----
  {
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = f32x4{0, 0, 0, 0};
    layer<8, kObs, 2>(acc, P.net.w0_pi, 8 * wave, s_obs, kObsStride, lane);
    store_relu<8>(acc, P.net.b0_pi, 8 * wave, s_h0, kH0Stride, lane);
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = f32x4{0, 0, 0, 0};
    layer<8, kObs, 2>(acc, P.net.w0_vf, 8 * wave, s_obs, kObsStride, lane);
    store_relu<8>(acc, P.net.b0_vf, 8 * wave, s_h0 + kH0, kH0Stride, lane);
  }
  __syncthreads();
  // ---- layer 1: 512 -> 256; this wave: column tiles [4 wave, 4 wave + 4) of each net ----
  {
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = f32x4{0, 0, 0, 0};
    layer<4, kH0, 4>(acc, P.net.w1_pi, 4 * wave, s_h0, kH0Stride, lane);
    store_relu<4>(acc, P.net.b1_pi, 4 * wave, s_h1, kH1Stride, lane);
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = f32x4{0, 0, 0, 0};
    layer<4, kH0, 4>(acc, P.net.w1_vf, 4 * wave, s_h0 + kH0, kH0Stride, lane);
    store_relu<4>(acc, P.net.b1_vf, 4 * wave, s_h1 + kH1, kH1Stride, lane);
  }
  __syncthreads();
  // ---- heads: wave 0 the actor (12 of 16 columns), wave 1 the critic (1 of 16 columns) ----
  if (wave >= 2) return;
  f32x4 acc[1] = {f32x4{0, 0, 0, 0}};
  const int col = lane & 15, r0 = 4 * (lane >> 4);
  if (wave == 0) {
    layer<1, kH1, 4>(acc, P.net.w2_pi, 0, s_h1, kH1Stride, lane);
    if (col < kAct) {
      const float b = P.net.b2_pi[col];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int robot = row0 + r0 + r;
        if (robot < P.n) {
          const size_t o = (size_t)robot * kAct + col;
          const float mu = acc[0][r] + b;
          const float a = P.noise ? mu + P.std * P.noise[o] : mu;
          P.action[o] = fminf(fmaxf(a, -P.clip), P.clip);
          if (P.raw) P.raw[o] = a;
          if (P.mean) P.mean[o] = mu;
        }
      }
    }
  } else {
    layer<1, kH1, 4>(acc, P.net.w2_vf, 0, s_h1 + kH1, kH1Stride, lane);
    if (col == 0 && P.value) {
      const float b = P.net.b2_vf[0];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int robot = row0 + r0 + r;
        if (robot < P.n) P.value[robot] = acc[0][r] + b;
      }
    }
  }
}

// One thread per robot: backward recursion over the segment, then mean / variance / standardisation of its own T values.
// legacy (ORR_GAE_LEGACY_INDEX): the recursion's "nonterminal" is read where the reference reads it, episode_starts[(step * N + i) + (1 + i)]
// of the flat [T * N] (+ N x False) array (agents/ppo_imitation.py:88) - robot i's own next-step flag only for N = 1, otherwise the flag of
// robot (2 i + 1) mod N at step t + (2 i + 1) / N.  episode_starts[t'][j] = done[t' - 1][j], and first_starts[j] for t' = 0.  The
// one-step value target still uses the robot's own flag (nextvpreds is filled per robot by the runner, imitation_runners.py:174-188).
__global__ void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ vpred, const uint8_t* __restrict__ dones,
                           const uint8_t* __restrict__ first_starts, const float* __restrict__ bootstrap, int T, int N, float gamma,
                           float lam, int flags, float eps, float* __restrict__ adv, float* __restrict__ ret) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const bool normalize = flags & ORR_GAE_NORMALIZE, legacy = flags & ORR_GAE_LEGACY_INDEX;
  float last = 0.0f, next_v = bootstrap ? bootstrap[n] : 0.0f, sum = 0.0f;
  for (int k = T - 1; k >= 0; k--) {
    const size_t o = (size_t)k * N + n;
    const float own = dones[o] ? 0.0f : 1.0f, v = vpred[o];
    float nonterminal = own;
    if (legacy) {
      const long long f = (long long)k * N + 2 * n + 1;
      const int ts = (int)(f / N), js = (int)(f % N);
      const bool start = ts >= T ? false : (ts == 0 ? (first_starts ? first_starts[js] != 0 : true) : dones[(size_t)(ts - 1) * N + js] != 0);
      nonterminal = start ? 0.0f : 1.0f;
    }
    const float delta = rewards[o] + gamma * next_v * own - v;
    last = delta + gamma * lam * nonterminal * last;
    adv[o] = last;
    ret[o] = last + v;
    sum += last;
    next_v = v;
  }
  if (!normalize) return;
  const float mean = sum / (float)T;
  float var = 0.0f;
  for (int k = 0; k < T; k++) { const float d = adv[(size_t)k * N + n] - mean; var += d * d; }
  const float inv = 1.0f / (sqrtf(var / (float)T) + eps);
  for (int k = 0; k < T; k++) { const size_t o = (size_t)k * N + n; adv[o] = (adv[o] - mean) * inv; }
}

}  // namespace

extern "C" {

int64_t orr_policy_packed_size(int32_t k, int32_t n) {
  if (k <= 0 || n <= 0 || (k % 16) != 0) return -1;
  return (int64_t)((n + 15) / 16) * (k / 16) * 256;
}

int32_t orr_policy_pack(const float* w_dev, int32_t k, int32_t n, float* out_dev, void* stream) {
  const long long total = orr_policy_packed_size(k, n);
  if (!w_dev || !out_dev || total < 0) return orr_fail(-1, "orr_policy_pack: bad argument (K must be a multiple of 16)", hipSuccess);
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_dev, (int)k, (int)n, out_dev, total);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_policy_pack: launch", e);
  return 0;
}

int32_t orr_policy_forward(const orr_policy_net* net, const float* obs, int32_t n, const float* noise, float std, float clip,
                           float* action, float* raw, float* value, float* mean, void* stream) {
  if (!net || !obs || !action || n < 0) return orr_fail(-1, "orr_policy_forward: bad argument", hipSuccess);
  const float* const* p = reinterpret_cast<const float* const*>(net);
  for (int i = 0; i < 12; i++)
    if (!p[i]) return orr_fail(-1, "orr_policy_forward: orr_policy_net has a null pointer", hipSuccess);
  if (n == 0) return 0;
  Params P;
  P.net = *net; P.obs = obs; P.noise = noise; P.action = action; P.raw = raw; P.value = value; P.mean = mean;
  P.n = n; P.std = std; P.clip = clip;
  hipLaunchKernelGGL(forward_kernel, dim3((unsigned)((n + kRows - 1) / kRows)), dim3(256), 0, (hipStream_t)stream, P);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_policy_forward: launch", e);
  return 0;
}

int32_t orr_gae_flags(const float* rewards, const float* vpred, const uint8_t* dones, const uint8_t* first_starts, const float* bootstrap,
                      int32_t t, int32_t n, float gamma, float lam, int32_t flags, float eps, float* adv, float* ret, void* stream) {
  if (!rewards || !vpred || !dones || !adv || !ret || t <= 0 || n < 0) return orr_fail(-1, "orr_gae: bad argument", hipSuccess);
  if (flags & ~(ORR_GAE_NORMALIZE | ORR_GAE_LEGACY_INDEX)) return orr_fail(-1, "orr_gae: unknown flag", hipSuccess);
  if (n == 0) return 0;
  hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rewards, vpred, dones, first_starts,
                     bootstrap, (int)t, (int)n, gamma, lam, (int)flags, eps, adv, ret);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_gae: launch", e);
  return 0;
}

int32_t orr_gae(const float* rewards, const float* vpred, const uint8_t* dones, const float* bootstrap, int32_t t, int32_t n,
                float gamma, float lam, int32_t normalize, float eps, float* adv, float* ret, void* stream) {
  return orr_gae_flags(rewards, vpred, dones, nullptr, bootstrap, t, n, gamma, lam, normalize ? ORR_GAE_NORMALIZE : 0, eps, adv, ret, stream);
}

}  // extern "C"
