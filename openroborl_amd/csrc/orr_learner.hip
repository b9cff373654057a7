// orr_learner.hip -- the non-GEMM part of one PPO minibatch update (include/openroborl_learner.h).
//
// Reference: the TF1 graph of agents/ppo_imitation.py:156-258 (loss :196-208) and MpiAdam (stable_baselines/common/mpi_adam.py:40-62).
// The six forward and ten backward contractions of the two 160 -> 512 -> 256 -> {12, 1} networks are library GEMMs issued by the
// host side (openroborl_amd/learner_hip.py); everything between them is here.  All of it is HBM-bound streaming over [m][c] float32
// activations (m = 16384, c = 512: 33.5 MB per tensor), so the rules are: touch every activation once, 16 bytes per lane, no float
// atomics (column sums go through per-block partials that a second small launch adds in a fixed order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openroborl_learner.h"

int orr_fail(int code, const char* msg, hipError_t e);  // orr_kernels.hip

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 32;     // activation rows per workgroup: m = 16384 -> 512 workgroups (two per CU), partials [512][c]
constexpr int kHeadCols = 16;         // head partials per workgroup: 12 actor bias gradients, 1 critic, surrogate sum, value-loss sum, spare
constexpr int kAct = 12;

__device__ inline float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}

// ---- loss head: one thread per sample --------------------------------------------------------------------------------------
// d(-min(r A, clip(r) A)) / dr as autograd forms it (ppo.PPO.update): inside the clip range both branches are the same value and
// together pass -A; outside it the clipped branch is constant, so -A passes only while the unclipped branch is the smaller one.
__global__ __launch_bounds__(kThreads) void ppo_head_kernel(const float* __restrict__ mean, const float* __restrict__ value,
                                                            const float* __restrict__ batch, int M, float inv_var, float log_norm, float clip,
                                                            float vf_coef, float inv_m, float* __restrict__ g_mean,
                                                            float* __restrict__ g_value, float* __restrict__ partials) {
  __shared__ float red[kThreads / 64][kHeadCols];
  const int i = blockIdx.x * kThreads + threadIdx.x;
  float acc[kHeadCols - 1];
#pragma unroll
  for (int k = 0; k < kHeadCols - 1; k++) acc[k] = 0.0f;
  if (i < M) {
    const f4* b4 = reinterpret_cast<const f4*>(batch + (size_t)i * ORR_PPO_BATCH_COLS);
    const f4* m4 = reinterpret_cast<const f4*>(mean + (size_t)i * kAct);
    const f4 a0 = b4[0], a1 = b4[1], a2 = b4[2], rest = b4[3];
    const f4 m0 = m4[0], m1 = m4[1], m2 = m4[2];
    float d[kAct] = {a0.x - m0.x, a0.y - m0.y, a0.z - m0.z, a0.w - m0.w, a1.x - m1.x, a1.y - m1.y, a1.z - m1.z, a1.w - m1.w,
                     a2.x - m2.x, a2.y - m2.y, a2.z - m2.z, a2.w - m2.w};
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < kAct; k++) sq += d[k] * d[k];
    const float old_logp = rest.x, A = rest.y, ret = rest.z;
    const float logp = -0.5f * sq * inv_var + log_norm;
    const float ratio = expf(logp - old_logp);
    const float lo = 1.0f - clip, hi = 1.0f + clip;
    const float s1 = ratio * A, s2 = fminf(fmaxf(ratio, lo), hi) * A;
    const bool inside = ratio >= lo && ratio <= hi;
    const float dr = (inside || s1 < s2) ? -A : 0.0f;
    const float gl = dr * ratio * inv_m * inv_var;            // d loss / d logp  x  1 / var
    float gm[kAct];
#pragma unroll
    for (int k = 0; k < kAct; k++) { gm[k] = gl * d[k]; acc[k] = gm[k]; }
    f4* o4 = reinterpret_cast<f4*>(g_mean + (size_t)i * kAct);
    o4[0] = f4{gm[0], gm[1], gm[2], gm[3]}; o4[1] = f4{gm[4], gm[5], gm[6], gm[7]}; o4[2] = f4{gm[8], gm[9], gm[10], gm[11]};
    const float dv = value[i] - ret;
    const float gv = vf_coef * 2.0f * dv * inv_m;
    g_value[i] = gv;
    acc[12] = gv; acc[13] = -fminf(s1, s2); acc[14] = dv * dv;
  }
#pragma unroll
  for (int k = 0; k < kHeadCols - 1; k++) {
    const float s = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < kHeadCols)
    partials[(size_t)blockIdx.x * kHeadCols + threadIdx.x] =
        threadIdx.x < kHeadCols - 1 ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.0f;
}

__global__ __launch_bounds__(64) void ppo_head_finish_kernel(const float* __restrict__ partials, int nblk, float inv_m, float* __restrict__ gb_mean,
                                                             float* __restrict__ gb_value, float* __restrict__ stats) {
  // 16 columns x 4 interleaved row groups, combined in a fixed order
  __shared__ float red[4][kHeadCols];
  const int col = threadIdx.x & 15, part = threadIdx.x >> 4;
  float s = 0.0f;
  for (int b = part; b < nblk; b += 4) s += partials[(size_t)b * kHeadCols + col];
  red[part][col] = s;
  __syncthreads();
  if (part == 0) {
    const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    if (col < kAct) gb_mean[col] = t;
    else if (col == 12) gb_value[0] = t;
    else if (col == 13) stats[0] = t * inv_m;
    else if (col == 14) stats[1] = t * inv_m;
  }
}

// ---- ReLU mask (+ the gradient through an output layer of fan-out K) + per-block column sums ------------------------------
// A workgroup owns kRowsPerBlock rows and all c columns; a thread owns four adjacent columns (one 16-byte access per tensor and
// row) of every (256 / (c/4))-th row.  K = 0: g holds the incoming gradient and is masked in place.  K > 0 (the layer below an
// output layer of fan-out K): the incoming gradient is gy . w^T formed on the fly, and since h and gy are in registers anyway the
// output layer's weight gradient h^T gy is accumulated as well (a library GEMM with N = 12 or 1 and K = the batch runs at 1 % of
// the chip: 100 us for 0.1 GFLOP) - wpart [block][c][K].
template <int K>
__global__ __launch_bounds__(kThreads) void relu_backward_kernel(float* __restrict__ g, const float* __restrict__ h, const float* __restrict__ gy,
                                                                 const float* __restrict__ w, int M, int C, float* __restrict__ partials,
                                                                 float* __restrict__ wpart) {
  constexpr int KK = K > 0 ? K : 1;
  __shared__ f4 red[kThreads];
  __shared__ float wred[K > 0 ? kThreads * 4 * KK : 1];
  const int c4n = C >> 2, side = kThreads / c4n;
  const int c4 = threadIdx.x % c4n, rsub = threadIdx.x / c4n;
  const int row0 = blockIdx.x * kRowsPerBlock;
  float wr[4][KK], gw[4][KK];
  if (K > 0) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int k = 0; k < K; k++) { wr[q][k] = w[(size_t)(4 * c4 + q) * K + k]; gw[q][k] = 0.0f; }
  }
  f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
  for (int r = rsub; r < kRowsPerBlock; r += side) {
    const int i = row0 + r;
    if (i >= M) break;
    const size_t o = (size_t)i * C + 4 * c4;
    const f4 hv = *reinterpret_cast<const f4*>(h + o);
    f4 gv;
    if (K > 0) {
      float y[KK];
      if (K % 4 == 0) {
#pragma unroll
        for (int k = 0; k < K / 4; k++) {
          const f4 t = reinterpret_cast<const f4*>(gy + (size_t)i * K)[k];
          y[4 * k] = t.x; y[4 * k + 1] = t.y; y[4 * k + 2] = t.z; y[4 * k + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < K; k++) y[k] = gy[(size_t)i * K + k];
      }
      float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < K; k++) {
#pragma unroll
        for (int q = 0; q < 4; q++) s[q] = fmaf(y[k], wr[q][k], s[q]);
        gw[0][k] = fmaf(hv.x, y[k], gw[0][k]); gw[1][k] = fmaf(hv.y, y[k], gw[1][k]);
        gw[2][k] = fmaf(hv.z, y[k], gw[2][k]); gw[3][k] = fmaf(hv.w, y[k], gw[3][k]);
      }
      gv = f4{s[0], s[1], s[2], s[3]};
    } else {
      gv = *reinterpret_cast<const f4*>(g + o);
    }
    gv.x = hv.x > 0.0f ? gv.x : 0.0f; gv.y = hv.y > 0.0f ? gv.y : 0.0f;
    gv.z = hv.z > 0.0f ? gv.z : 0.0f; gv.w = hv.w > 0.0f ? gv.w : 0.0f;
    *reinterpret_cast<f4*>(g + o) = gv;
    acc += gv;
  }
  red[threadIdx.x] = acc;
  if (K > 0) {
    // [rsub][q * K + k][c4]: consecutive lanes, consecutive words
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int k = 0; k < K; k++) wred[(rsub * 4 * KK + q * KK + k) * c4n + c4] = gw[q][k];
  }
  __syncthreads();
  if (rsub == 0) {
    f4 t = red[c4];
    for (int s = 1; s < side; s++) t += red[s * c4n + c4];
    *reinterpret_cast<f4*>(partials + (size_t)blockIdx.x * C + 4 * c4) = t;
  }
  if (K > 0) {
    // every thread finishes 4 K / side of this workgroup's c x K sums; wpart[block][(4 c4 + q) * K + k]
    float* o = wpart + (size_t)blockIdx.x * C * K;
    for (int e = rsub; e < 4 * K; e += side) {
      float t = wred[e * c4n + c4];
      for (int s = 1; s < side; s++) t += wred[(s * 4 * KK + e) * c4n + c4];
      o[(size_t)(4 * c4 + e / K) * K + (e % K)] = t;
    }
  }
}

// out[col] = sum over the nblk partial rows of a job: 16 columns per workgroup x 16 interleaved row groups (a thread adds
// nblk / 16 values through four independent accumulators), combined in a fixed order.  Up to 12 jobs per launch; a workgroup finds
// its job in the prefix table of workgroup counts.
constexpr int kMaxJobs = ORR_COLSUM_MAX_JOBS;
constexpr int kFewRows = 16;
struct FinishJobs {
  const float* partials[kMaxJobs];
  float* out[kMaxJobs];
  int nblk[kMaxJobs], cols[kMaxJobs], first[kMaxJobs + 1];
};
__global__ __launch_bounds__(kThreads) void colsum_finish_kernel(FinishJobs J, int n_jobs) {
  __shared__ float red[16][17];
  int job = 0;
  while (job + 1 < n_jobs && (int)blockIdx.x >= J.first[job + 1]) job++;
  const float* __restrict__ partials = J.partials[job];
  const int nblk = J.nblk[job], C = J.cols[job];
  if (nblk <= kFewRows) {      // few rows of many columns (the 16 partial products of a split weight gradient): a thread per column
    const int col = ((int)blockIdx.x - J.first[job]) * kThreads + threadIdx.x;
    if (col >= C) return;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int b = 0; b < nblk; b++) s[b & 3] += partials[(size_t)b * C + col];
    J.out[job][col] = (s[0] + s[1]) + (s[2] + s[3]);
    return;
  }
  const int lc = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int col = ((int)blockIdx.x - J.first[job]) * 16 + lc;
  float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (col < C) {
    int b = part;
    for (; b + 48 < nblk; b += 64) {
#pragma unroll
      for (int u = 0; u < 4; u++) s[u] += partials[(size_t)(b + 16 * u) * C + col];
    }
    for (int u = 0; b < nblk; b += 16, u++) s[u & 3] += partials[(size_t)b * C + col];
  }
  red[part][lc] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (part == 0 && col < C) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; q++) t += red[q][lc];
    J.out[job][col] = t;
  }
}

// ---- Adam on the flat parameter vector ----------------------------------------------------------------------------------------
// Every workgroup reads the step count when it starts; the LAST one to finish (ticket) advances it, so a captured launch replays.
constexpr int kAdamThreads = 64;     // 434 k parameters = 1700 one-wave workgroups: enough of them in flight to cover the load latency
__global__ __launch_bounds__(kAdamThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float gscale,
                                                        int flags, int* __restrict__ state) {
  const int t = __atomic_load_n(&state[0], __ATOMIC_RELAXED) + 1;
  const float bc1 = 1.0f - powf(b1, (float)t), bc2 = 1.0f - powf(b2, (float)t);
  const float sq2 = sqrtf(bc2);
  const float step = (flags & ORR_ADAM_MPI_EPSILON) ? lr * sq2 / bc1 : lr / bc1;
  const float vs = (flags & ORR_ADAM_MPI_EPSILON) ? 1.0f : 1.0f / sq2;
  const long long i4 = ((long long)blockIdx.x * kAdamThreads + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    f4 gg = *reinterpret_cast<const f4*>(g + i4) * gscale;
    f4 mm = *reinterpret_cast<f4*>(m + i4), vv = *reinterpret_cast<f4*>(v + i4), pp = *reinterpret_cast<f4*>(p + i4);
    mm = mm + (gg - mm) * (1.0f - b1);
    vv = vv * b2 + gg * gg * (1.0f - b2);
    pp.x -= step * mm.x / (sqrtf(vv.x) * vs + eps); pp.y -= step * mm.y / (sqrtf(vv.y) * vs + eps);
    pp.z -= step * mm.z / (sqrtf(vv.z) * vs + eps); pp.w -= step * mm.w / (sqrtf(vv.w) * vs + eps);
    *reinterpret_cast<f4*>(m + i4) = mm; *reinterpret_cast<f4*>(v + i4) = vv; *reinterpret_cast<f4*>(p + i4) = pp;
  } else {
    for (long long i = i4; i < n; i++) {
      const float gg = g[i] * gscale;
      const float mm = m[i] + (gg - m[i]) * (1.0f - b1), vv = v[i] * b2 + gg * gg * (1.0f - b2);
      m[i] = mm; v[i] = vv;
      p[i] -= step * mm / (sqrtf(vv) * vs + eps);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = atomicAdd(&state[1], 1);
    if (ticket == (int)gridDim.x - 1) {
      __atomic_store_n(&state[0], t, __ATOMIC_RELAXED);
      __atomic_store_n(&state[1], 0, __ATOMIC_RELAXED);
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int nblocks_rows(int m) { return (m + kRowsPerBlock - 1) / kRowsPerBlock; }
inline bool cols_ok(int c) { return c >= 4 && (c % 4) == 0 && (c / 4) <= kThreads && (kThreads % (c / 4)) == 0; }

int launch_finish(FinishJobs& J, int n_jobs, hipStream_t st, const char* who) {
  J.first[0] = 0;
  for (int j = 0; j < n_jobs; j++) J.first[j + 1] = J.first[j] + (J.nblk[j] <= kFewRows ? (J.cols[j] + kThreads - 1) / kThreads : (J.cols[j] + 15) / 16);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)J.first[n_jobs]), dim3(kThreads), 0, st, J, n_jobs);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, who, e);
  return 0;
}

}  // namespace

extern "C" {

int32_t orr_learner_partial_rows(int32_t m) { return m > 0 ? nblocks_rows(m) : -1; }

int64_t orr_learner_workspace_floats(int32_t m, int32_t c) {
  if (m <= 0 || c <= 0) return -1;
  const int64_t rows = nblocks_rows(m), head = (m + kThreads - 1) / kThreads;
  const int64_t a = rows * (int64_t)c * (1 + kAct), b = head * kHeadCols;     // orr_head_backward: [rows][c] + [rows][c][12]
  return a > b ? a : b;
}

int32_t orr_ppo_head(const float* mean, const float* value, const float* batch, int32_t m, float std, float clip, float vf_coef,
                     float* g_mean, float* g_value, float* gb_mean, float* gb_value, float* stats, float* workspace, void* stream) {
  if (!mean || !value || !batch || !g_mean || !g_value || !gb_mean || !gb_value || !stats || !workspace || m <= 0 || !(std > 0.0f))
    return orr_fail(-1, "orr_ppo_head: bad argument", hipSuccess);
  if (!aligned16(mean) || !aligned16(batch) || !aligned16(g_mean)) return orr_fail(-1, "orr_ppo_head: buffers must be 16-byte aligned", hipSuccess);
  const float var = std * std;
  const float log_norm = -0.5f * (float)kAct * logf(2.0f * 3.14159265358979323846f * var);
  const int nblk = (m + kThreads - 1) / kThreads;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ppo_head_kernel, dim3((unsigned)nblk), dim3(kThreads), 0, st, mean, value, batch, (int)m, 1.0f / var, log_norm, clip, vf_coef,
                     1.0f / (float)m, g_mean, g_value, workspace);
  hipLaunchKernelGGL(ppo_head_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, nblk, 1.0f / (float)m, gb_mean, gb_value, stats);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_ppo_head: launch", e);
  return 0;
}

int32_t orr_relu_backward(float* g, const float* h, int32_t m, int32_t c, float* gb, float* workspace, void* stream) {
  if (!g || !h || !workspace || m <= 0) return orr_fail(-1, "orr_relu_backward: bad argument", hipSuccess);
  if (!cols_ok(c)) return orr_fail(-1, "orr_relu_backward: c must be a multiple of 4 that divides 1024", hipSuccess);
  if (!aligned16(g) || !aligned16(h) || !aligned16(workspace)) return orr_fail(-1, "orr_relu_backward: buffers must be 16-byte aligned", hipSuccess);
  const int nblk = nblocks_rows(m);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(relu_backward_kernel<0>, dim3((unsigned)nblk), dim3(kThreads), 0, st, g, h, (const float*)nullptr, (const float*)nullptr, (int)m,
                     (int)c, workspace, (float*)nullptr);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_relu_backward: launch", e);
  if (!gb) return 0;                                 // deferred: the caller sums workspace [rows][c] with orr_colsum_finish
  FinishJobs J{};
  J.partials[0] = workspace; J.out[0] = gb; J.nblk[0] = nblk; J.cols[0] = c;
  return launch_finish(J, 1, st, "orr_relu_backward: launch");
}

int32_t orr_head_backward(const float* gy, int32_t k, const float* w, const float* h, int32_t m, int32_t c, float* gz, float* gb, float* gw,
                          float* workspace, void* stream) {
  if (!gy || !w || !h || !gz || !workspace || m <= 0) return orr_fail(-1, "orr_head_backward: bad argument", hipSuccess);
  if (k != 12 && k != 1) return orr_fail(-1, "orr_head_backward: fan-out must be 12 or 1", hipSuccess);
  if (!cols_ok(c)) return orr_fail(-1, "orr_head_backward: c must be a multiple of 4 that divides 1024", hipSuccess);
  if ((gb == nullptr) != (gw == nullptr)) return orr_fail(-1, "orr_head_backward: gb and gw must both be given or both be deferred", hipSuccess);
  if (!aligned16(gy) || !aligned16(h) || !aligned16(gz) || !aligned16(workspace))
    return orr_fail(-1, "orr_head_backward: buffers must be 16-byte aligned", hipSuccess);
  const int nblk = nblocks_rows(m);
  hipStream_t st = (hipStream_t)stream;
  float* wpart = workspace + (size_t)nblk * c;
  if (k == 12)
    hipLaunchKernelGGL(relu_backward_kernel<12>, dim3((unsigned)nblk), dim3(kThreads), 0, st, gz, h, gy, w, (int)m, (int)c, workspace, wpart);
  else
    hipLaunchKernelGGL(relu_backward_kernel<1>, dim3((unsigned)nblk), dim3(kThreads), 0, st, gz, h, gy, w, (int)m, (int)c, workspace, wpart);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_head_backward: launch", e);
  if (!gb) return 0;                                 // deferred: workspace [rows][c] then [rows][c * k]
  FinishJobs J{};
  J.partials[0] = workspace; J.out[0] = gb; J.nblk[0] = nblk; J.cols[0] = c;
  J.partials[1] = wpart; J.out[1] = gw; J.nblk[1] = nblk; J.cols[1] = c * k;
  return launch_finish(J, 2, st, "orr_head_backward: launch");
}

int32_t orr_colsum_finish(const orr_colsum_job* jobs, int32_t n_jobs, void* stream) {
  if (!jobs || n_jobs < 1 || n_jobs > ORR_COLSUM_MAX_JOBS) return orr_fail(-1, "orr_colsum_finish: 1 to 12 jobs", hipSuccess);
  FinishJobs J{};
  for (int j = 0; j < n_jobs; j++) {
    if (!jobs[j].partials || !jobs[j].out || jobs[j].rows < 1 || jobs[j].cols < 1) return orr_fail(-1, "orr_colsum_finish: bad job", hipSuccess);
    J.partials[j] = jobs[j].partials; J.out[j] = jobs[j].out; J.nblk[j] = jobs[j].rows; J.cols[j] = jobs[j].cols;
  }
  return launch_finish(J, n_jobs, (hipStream_t)stream, "orr_colsum_finish: launch");
}

int32_t orr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float grad_scale, int32_t flags, int32_t* state, void* stream) {
  if (!p || !g || !m || !v || !state || n <= 0) return orr_fail(-1, "orr_adam_step: bad argument", hipSuccess);
  if (flags & ~ORR_ADAM_MPI_EPSILON) return orr_fail(-1, "orr_adam_step: unknown flag", hipSuccess);
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return orr_fail(-1, "orr_adam_step: buffers must be 16-byte aligned", hipSuccess);
  const long long per_block = 4LL * kAdamThreads;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kAdamThreads), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr,
                     beta1, beta2, eps, grad_scale, (int)flags, (int*)state);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return orr_fail(-2, "orr_adam_step: launch", e);
  return 0;
}

}  // extern "C"
