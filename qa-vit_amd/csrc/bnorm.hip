// BatchNorm2d (+ optional exact GELU) of the CNN stem on channel-last rows [M = B*H*W, C]
// (HQAViT_CIFAR100.py:753,760,768,775 -- nn.BatchNorm2d(32/64/128/256), two of them followed by nn.GELU()).
// HBM-bound column statistics: two passes over the activation per direction.
//   forward : stats (per-channel sum / sum of squares about a pivot)  ->  apply (normalise, affine, GELU; running stats)
//   backward: stats (sum dy', sum dy'*xhat with dy' = dy * gelu'(z))   ->  apply (dx; dgamma / dbeta accumulate)
// A thread owns VEC consecutive channels of a row (one 16-byte load) and always the SAME channels, so the column
// partials are registers; a workgroup folds them through LDS and leaves 2*C fp32 atomics.  Grids are capped at 256
// workgroups: the flush, not the streaming, was what bounded the LayerNorm backward (norm.hip).
#include "common.cuh"
#include <stdlib.h>
#include "../../include/qavit.h"
#include "launch.h"

namespace qv {

namespace {

template <typename T> struct BV;
template <> struct BV<bf16> { static constexpr int N = 8; typedef bf16x8 type; };
template <> struct BV<float> { static constexpr int N = 4; typedef f32x4 type; };

constexpr int BN_MAXC = 2048;
constexpr int BN_U = 4, BN_US = 8;                         // rows a thread has in flight: apply passes (thousands of workgroups), statistics passes (<= 256)

// ws[0..C) = sum (x - pivot), ws[C..2C) = sum (x - pivot)^2, ws[2C..3C) = the pivot used (the running mean BEFORE this
// step's update: any pivot is exact, one near the mean avoids the E[x^2] - mean^2 cancellation)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* x, const float* pivot, float* ws, int M, int C, int rows_per_wg) {
  constexpr int VEC = BV<T>::N;
  typedef typename BV<T>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float sred[];     // [2][C]
  const int tpr = C / VEC, rpp = 256 / tpr;
  const int cg = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sred[i] = 0.f;
  if (blockIdx.x == 0) for (int i = threadIdx.x; i < C; i += 256) ws[2 * C + i] = pivot ? pivot[i] : 0.f;
  float pv[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { pv[j] = pivot ? pivot[cg * VEC + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  const int mb = blockIdx.x * rows_per_wg;
  const int me = (mb + rows_per_wg < M) ? mb + rows_per_wg : M;
  // BN_US row loads in flight per thread: with one, a workgroup keeps 4 KB on the wire and the pass runs at a quarter of the memory rate
  for (int m0 = mb + r0; m0 < me; m0 += rpp * BN_US) {
    vec_t v[BN_US];
#pragma unroll
    for (int u = 0; u < BN_US; ++u) {
      const int m = m0 + u * rpp;
      v[u] = *reinterpret_cast<const vec_t*>(x + (size_t)(m < me ? m : m0) * C + cg * VEC);
    }
#pragma unroll
    for (int u = 0; u < BN_US; ++u) {
      if (m0 + u * rpp < me) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float d = to_f<T>(v[u][j]) - pv[j]; s1[j] += d; s2[j] += d * d; }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VEC; ++j) { atomicAdd(sred + cg * VEC + j, s1[j]); atomicAdd(sred + C + cg * VEC + j, s2[j]); }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomic_add_f(ws + i, sred[i]);
}

// training: mean / rstd from ws (and block 0 updates the running statistics and the saved stats); eval: from running stats
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* x, T* y, const float* gamma, const float* beta, const float* ws,
                                                       float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                                                       float momentum, float eps, int act, int training, int M, int C, int64_t Mtot) {
  constexpr int VEC = BV<T>::N;
  typedef typename BV<T>::type vec_t;
  const int tpr = C / VEC, rpp = 256 / tpr;
  const int cg = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
  const float invM = 1.f / (float)Mtot;                  // Mtot = rows behind the statistics (all ranks' rows when they were all-reduced)
  float mu[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cg * VEC + j;
    float mean, var;
    if (training) {
      const float p = ws[2 * C + c];
      const float a1 = ws[c] * invM;
      mean = p + a1;
      var = fmaxf(ws[C + c] * invM - a1 * a1, 0.f);
    } else { mean = running_mean[c]; var = running_var[c]; }
    const float rstd = rsqrtf(var + eps);
    mu[j] = mean; sc[j] = rstd * gamma[c]; sh[j] = beta[c];
    if (training && blockIdx.x == 0 && r0 == 0) {
      save_mean[c] = mean; save_rstd[c] = rstd;
      if (running_mean) {
        const float unb = Mtot > 1 ? var * ((float)Mtot / (float)(Mtot - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
      }
    }
  }
  for (int m0 = blockIdx.x * rpp * BN_U + r0; m0 < M; m0 += gridDim.x * rpp * BN_U) {
    vec_t v[BN_U];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int m = m0 + u * rpp;
      v[u] = *reinterpret_cast<const vec_t*>(x + (size_t)(m < M ? m : m0) * C + cg * VEC);
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int m = m0 + u * rpp;
      if (m >= M) break;
      vec_t o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float z = (to_f<T>(v[u][j]) - mu[j]) * sc[j] + sh[j];
        if (act) z = gelu_f(z);
        o[j] = from_f<T>(z);
      }
      *reinterpret_cast<vec_t*>(y + (size_t)m * C + cg * VEC) = o;
    }
  }
}

// ws[0..C) = sum dy', ws[C..2C) = sum dy' * xhat
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* dy, const T* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                                           float* ws, int act, int M, int C, int rows_per_wg) {
  constexpr int VEC = BV<T>::N;
  typedef typename BV<T>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float sred[];
  const int tpr = C / VEC, rpp = 256 / tpr;
  const int cg = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
  for (int i = threadIdx.x; i < 2 * C; i += 256) sred[i] = 0.f;
  float mu[VEC], rs[VEC], ga[VEC], be[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cg * VEC + j;
    mu[j] = mean[c]; rs[j] = rstd[c]; ga[j] = gamma[c]; be[j] = beta[c]; s1[j] = 0.f; s2[j] = 0.f;
  }
  const int mb = blockIdx.x * rows_per_wg;
  const int me = (mb + rows_per_wg < M) ? mb + rows_per_wg : M;
  for (int m0 = mb + r0; m0 < me; m0 += rpp * BN_US) {
    vec_t xv[BN_US], gv[BN_US];
#pragma unroll
    for (int u = 0; u < BN_US; ++u) {
      const int m = m0 + u * rpp;
      const size_t off = (size_t)(m < me ? m : m0) * C + cg * VEC;
      xv[u] = *reinterpret_cast<const vec_t*>(x + off);
      gv[u] = *reinterpret_cast<const vec_t*>(dy + off);
    }
#pragma unroll
    for (int u = 0; u < BN_US; ++u) {
      if (m0 + u * rpp < me) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xh = (to_f<T>(xv[u][j]) - mu[j]) * rs[j];
          float g = to_f<T>(gv[u][j]);
          if (act) g *= gelu_grad_f(xh * ga[j] + be[j]);
          s1[j] += g; s2[j] += g * xh;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VEC; ++j) { atomicAdd(sred + cg * VEC + j, s1[j]); atomicAdd(sred + C + cg * VEC + j, s2[j]); }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomic_add_f(ws + i, sred[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* dy, const T* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                                           const float* ws, T* dx, float* dgamma, float* dbeta, int act, int training, int M, int C,
                                                           int64_t Mtot, const float* ws_param) {
  constexpr int VEC = BV<T>::N;
  typedef typename BV<T>::type vec_t;
  const int tpr = C / VEC, rpp = 256 / tpr;
  const int cg = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
  const float invM = 1.f / (float)Mtot;
  float mu[VEC], rs[VEC], ga[VEC], be[VEC], k1[VEC], k2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cg * VEC + j;
    mu[j] = mean[c]; rs[j] = rstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
    k1[j] = training ? ws[c] * invM : 0.f;             // eval mode: the statistics are constants, dx = gamma * rstd * dy'
    k2[j] = training ? ws[C + c] * invM : 0.f;
    if (blockIdx.x == 0 && r0 == 0) {
      if (dbeta) atomic_add_f(dbeta + c, ws_param[c]);          // parameter gradients: THIS rank's sums (the gradient all-reduce adds the others)
      if (dgamma) atomic_add_f(dgamma + c, ws_param[C + c]);
    }
  }
  for (int m0 = blockIdx.x * rpp * BN_U + r0; m0 < M; m0 += gridDim.x * rpp * BN_U) {
    vec_t xv[BN_U], gv[BN_U];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int m = m0 + u * rpp;
      const size_t off = (size_t)(m < M ? m : m0) * C + cg * VEC;
      xv[u] = *reinterpret_cast<const vec_t*>(x + off);
      gv[u] = *reinterpret_cast<const vec_t*>(dy + off);
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int m = m0 + u * rpp;
      if (m >= M) break;
      vec_t o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float xh = (to_f<T>(xv[u][j]) - mu[j]) * rs[j];
        float g = to_f<T>(gv[u][j]);
        if (act) g *= gelu_grad_f(xh * ga[j] + be[j]);
        o[j] = from_f<T>(ga[j] * rs[j] * (g - k1[j] - xh * k2[j]));
      }
      *reinterpret_cast<vec_t*>(dx + (size_t)m * C + cg * VEC) = o;
    }
  }
}

template <typename T>
bool bn_shape_ok(const void* p0, const void* p1, const void* p2, int C) {
  constexpr int VEC = BV<T>::N;
  if (C % VEC || C > BN_MAXC) return false;
  const int tpr = C / VEC;
  if (tpr > 256 || 256 % tpr) return false;
  return ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
}

inline int bn_apply_cap() {
  static const int cap = getenv("QAVIT_BN_APPLY_WGS") ? atoi(getenv("QAVIT_BN_APPLY_WGS")) : 1024;
  return cap;
}

inline void stats_grid(int M, int C, int vec, int& grid, int& rows_per_wg) {
  const int rpp = 256 / (C / vec);
  grid = (M + 4 * rpp - 1) / (4 * rpp);                 // >= 4 passes per workgroup
  static const int cap = getenv("QAVIT_BN_STATS_WGS") ? atoi(getenv("QAVIT_BN_STATS_WGS")) : 256;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  rows_per_wg = (M + grid - 1) / grid;
  rows_per_wg = (rows_per_wg + rpp - 1) / rpp * rpp;
  grid = (M + rows_per_wg - 1) / rows_per_wg;
}

template <typename T>
int bn_fwd_t(const void* x, void* y, int M, int C, const float* gamma, const float* beta, float* rm, float* rv, float momentum, float eps,
             int act, float* save_mean, float* save_rstd, float* ws, int training, int phase, int64_t Mtot, hipStream_t st) {
  constexpr int VEC = BV<T>::N;
  if (!bn_shape_ok<T>(x, y, x, C)) return set_error(QAVIT_EINVAL, "bn_fwd: C must be a multiple of the 16-byte vector with 256 % (C/vec) == 0, 16-byte aligned rows");
  if (Mtot <= 0) Mtot = M;
  if (training && phase != 2) {
    zero_f32(ws, (size_t)3 * C, st);
    int grid, rows;
    stats_grid(M, C, VEC, grid, rows);
    hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(grid), dim3(256), (size_t)2 * C * sizeof(float), st, (const T*)x, rm, ws, M, C, rows);
  }
  if (phase == 1) return check_launch("bn_fwd(stats)");
  const int rpp = 256 / (C / VEC);
  int g2 = (M + rpp * BN_U - 1) / (rpp * BN_U);
  if (g2 > bn_apply_cap()) g2 = bn_apply_cap();
  hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(g2), dim3(256), 0, st, (const T*)x, (T*)y, gamma, beta, ws, rm, rv, save_mean, save_rstd,
                     momentum, eps, act, training, M, C, Mtot);
  return check_launch("bn_fwd");
}

template <typename T>
int bn_bwd_t(const void* dy, const void* x, int M, int C, const float* gamma, const float* beta, const float* mean, const float* rstd, int act,
             int training, void* dx, float* dgamma, float* dbeta, float* ws, int phase, int64_t Mtot, const float* ws_param, hipStream_t st) {
  constexpr int VEC = BV<T>::N;
  if (!bn_shape_ok<T>(x, dy, dx, C)) return set_error(QAVIT_EINVAL, "bn_bwd: unsupported channel count / alignment");
  if (Mtot <= 0) Mtot = M;
  if (!ws_param) ws_param = ws;
  if (phase != 2) {
    zero_f32(ws, (size_t)2 * C, st);
    int grid, rows;
    stats_grid(M, C, VEC, grid, rows);
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T>), dim3(grid), dim3(256), (size_t)2 * C * sizeof(float), st, (const T*)dy, (const T*)x, gamma, beta, mean, rstd,
                       ws, act, M, C, rows);
  }
  if (phase == 1) return check_launch("bn_bwd(stats)");
  const int rpp = 256 / (C / VEC);
  int g2 = (M + rpp * BN_U - 1) / (rpp * BN_U);
  if (g2 > bn_apply_cap()) g2 = bn_apply_cap();
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g2), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, beta, mean, rstd, ws, (T*)dx, dgamma, dbeta, act, training, M, C, Mtot, ws_param);
  return check_launch("bn_bwd");
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_bn_fwd(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int act,
                            float* save_mean, float* save_rstd, float* ws, int training, int phase, int64_t M_total, void* stream) {
  if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || phase < 0 || phase > 2) return set_error(QAVIT_EINVAL, "bn_fwd: bad arguments");
  if (training && (!save_mean || !save_rstd || !ws)) return set_error(QAVIT_EINVAL, "bn_fwd: training needs save_mean / save_rstd / ws[3*C]");
  if (!training && (!running_mean || !running_var)) return set_error(QAVIT_EINVAL, "bn_fwd: eval needs the running statistics");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32) return bn_fwd_t<float>(x, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, act, save_mean, save_rstd, ws, training, phase, M_total, st);
  if (dtype == QAVIT_BF16) return bn_fwd_t<bf16>(x, y, M, C, gamma, beta, running_mean, running_var, momentum, eps, act, save_mean, save_rstd, ws, training, phase, M_total, st);
  return set_error(QAVIT_EINVAL, "bn_fwd: unknown dtype");
}

extern "C" int qavit_bn_bwd(int dtype, const void* dy, const void* x, int M, int C, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_rstd, int act, int training, void* dx, float* dgamma, float* dbeta,
                            float* ws, int phase, int64_t M_total, const float* ws_param, void* stream) {
  if (phase < 0 || phase > 2) return set_error(QAVIT_EINVAL, "bn_bwd: phase must be 0, 1 or 2");
  if (!dy || !x || !dx || !gamma || !beta || !save_mean || !save_rstd || !ws || M <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "bn_bwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32) return bn_bwd_t<float>(dy, x, M, C, gamma, beta, save_mean, save_rstd, act, training, dx, dgamma, dbeta, ws, phase, M_total, ws_param, st);
  if (dtype == QAVIT_BF16) return bn_bwd_t<bf16>(dy, x, M, C, gamma, beta, save_mean, save_rstd, act, training, dx, dgamma, dbeta, ws, phase, M_total, ws_param, st);
  return set_error(QAVIT_EINVAL, "bn_bwd: unknown dtype");
}
