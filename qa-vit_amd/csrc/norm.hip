// nn.LayerNorm over the channel axis, forward and backward.  HBM-bound streaming kernels: one wavefront
// per row, 64 lanes stride the C <= 1024 channels, statistics by wave shuffles (no LDS in forward).
#include "common.cuh"
#include <stdlib.h>
#include "../../include/qavit.h"
#include "launch.h"

namespace qv {

constexpr int LN_MAX_C = 1024;

template <typename T, int LN_MAX_PER_LANE>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* x, T* y, const float* gamma, const float* beta, float eps,
                                                            int rows, int C, float* mean_o, float* rstd_o,
                                                            const float* add, int add_rows, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const T* xr = x + (size_t)row * C;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < C) ? to_f<T>(xr[c]) : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) * invC;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      const float dlt = (c < C) ? v[i] - mean : 0.f;
      s2 += dlt * dlt;
    }
    const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
    T* yr = y + (size_t)row * C;
    const float* ar = add ? add + (size_t)(row % add_rows) * C : nullptr;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
        if (act) o = gelu_f(o);
        if (ar) o += ar[c];
        yr[c] = from_f<T>(o);
      }
    }
    if (lane == 0) {
      if (mean_o) mean_o[row] = mean;
      if (rstd_o) rstd_o[row] = rstd;
    }
  }
}

// mean / rstd only (the statistics half of a LayerNorm whose normalisation is fused into the consumer GEMM)
template <typename T, int LN_MAX_PER_LANE>
__device__ __forceinline__ void row_stats_body(const T* x, float eps, int rows, int C, float* mean_o, float* rstd_o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const T* xr = x + (size_t)row * C;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < C) ? to_f<T>(xr[c]) : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) * invC;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      const float dlt = (c < C) ? v[i] - mean : 0.f;
      s2 += dlt * dlt;
    }
    const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  }
}

template <typename T, int LN_MAX_PER_LANE>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* x, float eps, int rows, int C, float* mean_o, float* rstd_o) {
  row_stats_body<T, LN_MAX_PER_LANE>(x, eps, rows, C, mean_o, rstd_o);
}
// up to 4 same-shape inputs in one grid (blockIdx.y = input): the four branch norms of a QuadAttentionBlock
struct Ptr4 { const void* x[4]; float* mean[4]; float* rstd[4]; };
template <typename T, int LN_MAX_PER_LANE>
__global__ __launch_bounds__(256) void row_stats_multi_kernel(Ptr4 P, float eps, int rows, int C) {
  row_stats_body<T, LN_MAX_PER_LANE>(reinterpret_cast<const T*>(P.x[blockIdx.y]), eps, rows, C, P.mean[blockIdx.y], P.rstd[blockIdx.y]);
}

template <typename T, int LN_MAX_PER_LANE>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                            const float* rstd, T* dx, float* dgamma, float* dbeta,
                                                            int rows, int C, float* dadd, int add_rows, const float* beta, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  float pg[LN_MAX_PER_LANE], pb[LN_MAX_PER_LANE];
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const T* xr = x + (size_t)row * C;
    const T* gr = dy + (size_t)row * C;
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAX_PER_LANE], g[LN_MAX_PER_LANE];
    float c1 = 0.f, c2 = 0.f;
    float* dar = dadd ? dadd + (size_t)(row % add_rows) * C : nullptr;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        float d = to_f<T>(gr[c]);
        if (dar) atomic_add_f(dar + c, d);                          // y = act(LN(x)) + add: the add sees dy itself
        xh[i] = (to_f<T>(xr[c]) - mu) * rs;
        if (act) d *= gelu_grad_f(xh[i] * gamma[c] + beta[c]);      // y = gelu(LN(x)): gradient wrt the LN output
        g[i] = d * gamma[c];
        pg[i] += d * xh[i];
        pb[i] += d;
        c1 += g[i] * xh[i];
        c2 += g[i];
      } else { xh[i] = 0.f; g[i] = 0.f; }
    }
    c1 = wave_sum(c1) * invC;
    c2 = wave_sum(c2) * invC;
    T* dxr = dx + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < C) dxr[c] = from_f<T>(rs * (g[i] - c2 - xh[i] * c1));
    }
  }
  // reduce the 4 waves' partial dgamma / dbeta through LDS, one atomic per column per workgroup
  __shared__ float red[2][4][64];
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    if (64 * i >= C) break;
    __syncthreads();
    red[0][wave][lane] = pg[i];
    red[1][wave][lane] = pb[i];
    __syncthreads();
    if (wave == 0) {
      const int c = lane + 64 * i;
      if (c < C) {
        if (dgamma) atomic_add_f(dgamma + c, red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
        if (dbeta) atomic_add_f(dbeta + c, red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
      }
    }
  }
}

template <typename T> struct V4sel;
template <> struct V4sel<float> { typedef f32x4 type; };
template <> struct V4sel<bf16> { typedef bf16x4 type; };

// dadd[(row % add_rows), c] += dy[row, c]: the gradient of the broadcast add (pos_embed) is the batch sum of dy.  The flat view has
// period P = add_rows * C; a thread owns 4 consecutive elements of the period and walks a slice of the repeats (coalesced 8 / 16-byte
// loads), then adds once -- gridDim.y same-address atomics per element instead of one per row.
template <typename T>
__global__ __launch_bounds__(256) void ln_dadd_kernel(const T* dy, float* dadd, long total, int P, int reps_per_slice) {
  typedef typename V4sel<T>::type v4;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= P) return;
  const long r0 = (long)blockIdx.y * reps_per_slice;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (int r = 0; r < reps_per_slice; ++r) {
    const long idx = (r0 + r) * P + e;
    if (idx + 3 < total) {
      const v4 v = *reinterpret_cast<const v4*>(dy + idx);
      a0 += to_f<T>(v[0]); a1 += to_f<T>(v[1]); a2 += to_f<T>(v[2]); a3 += to_f<T>(v[3]);
    }
  }
  atomic_add_f(dadd + e, a0); atomic_add_f(dadd + e + 1, a1); atomic_add_f(dadd + e + 2, a2); atomic_add_f(dadd + e + 3, a3);
}

}  // namespace qv

using namespace qv;

// ---- vector variant (C % 4 == 0, 16-byte aligned rows for bf16 x4 = 8 B / fp32 x4 = 16 B): lane owns channels
// 4*lane .. 4*lane+3 (+256 per extra pass) -> 8/16-byte loads instead of 2/4-byte ones
template <typename T> struct V4;
template <> struct V4<float> { typedef f32x4 type; };
template <> struct V4<bf16> { typedef bf16x4 type; };

// forward / statistics, vector form: lane owns channels 4 lane .. + 3 (+ 256 per extra pass), RB rows of a wave are loaded before the
// first reduction.  STATS: mean / rstd only (qavit_row_stats).  Same two-pass arithmetic as the scalar kernels above.
template <typename T, int NP, int RB, int NW, bool STATS>
__device__ __forceinline__ void layernorm_fwd_v4_body(const T* x, T* y, const float* gamma, const float* beta, float eps, int rows, int C,
                                                      float* mean_o, float* rstd_o, const float* add, int add_rows, int act) {
  typedef typename V4<T>::type v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  float gm[NP][4], bt[NP][4];
  if (!STATS) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = 4 * lane + 256 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) { gm[i][j] = (c + j < C) ? gamma[c + j] : 0.f; bt[i][j] = (c + j < C) ? beta[c + j] : 0.f; }
    }
  }
  for (int row0 = (blockIdx.x * NW + wave) * RB; row0 < rows; row0 += gridDim.x * NW * RB) {
    v4 xv[RB][NP];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < C) xv[r][i] = *reinterpret_cast<const v4*>(x + (size_t)row * C + c);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool live = row0 + r < rows;
      float v[NP][4];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = 4 * lane + 256 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[i][j] = (c < C) ? to_f<T>(xv[r][i][j]) : 0.f; s += v[i][j]; }
      }
      const float mean = wave_sum(s) * invC;
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = 4 * lane + 256 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float dlt = (c < C) ? v[i][j] - mean : 0.f; s2 += dlt * dlt; }
      }
      const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
      if (!live) continue;                                  // uniform per wave
      if (lane == 0) {
        if (mean_o) mean_o[row0 + r] = mean;
        if (rstd_o) rstd_o[row0 + r] = rstd;
      }
      if (!STATS) {
        const float* ar = add ? add + (size_t)((row0 + r) % add_rows) * C : nullptr;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int c = 4 * lane + 256 * i;
          if (c < C) {
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (ar) av = *reinterpret_cast<const f32x4*>(ar + c);
            v4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float t = (v[i][j] - mean) * rstd * gm[i][j] + bt[i][j];
              if (act) t = gelu_f(t);
              o[j] = from_f<T>(t + av[j]);
            }
            *reinterpret_cast<v4*>(y + (size_t)(row0 + r) * C + c) = o;
          }
        }
      }
    }
  }
}
template <typename T, int NP, int RB, int NW, bool STATS>
__global__ __launch_bounds__(64 * NW) void layernorm_fwd_v4_kernel(const T* x, T* y, const float* gamma, const float* beta, float eps, int rows, int C,
                                                                   float* mean_o, float* rstd_o, const float* add, int add_rows, int act) {
  layernorm_fwd_v4_body<T, NP, RB, NW, STATS>(x, y, gamma, beta, eps, rows, C, mean_o, rstd_o, add, add_rows, act);
}
struct Ptr4v { const void* x[4]; float* mean[4]; float* rstd[4]; };
template <typename T, int NP, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void row_stats_v4_multi_kernel(Ptr4v P, float eps, int rows, int C) {
  layernorm_fwd_v4_body<T, NP, RB, NW, true>(reinterpret_cast<const T*>(P.x[blockIdx.y]), nullptr, nullptr, nullptr, eps, rows, C, P.mean[blockIdx.y],
                                             P.rstd[blockIdx.y], nullptr, 0, 0);
}

// launch helper: -> true when the vector form applies (C % 4 == 0, C <= 1024, vector-aligned rows)
template <bool STATS>
static bool ln_fwd_v4_launch(int dtype, const void* x, void* y, const float* gamma, const float* beta, float eps, int rows, int C, float* mean, float* rstd,
                             const float* add, int add_rows, int act, hipStream_t st) {
  static const int on = getenv("QAVIT_LNF_V4") ? atoi(getenv("QAVIT_LNF_V4")) : 1;
  if (!on || C % 4 || C > 1024 || (dtype != QAVIT_F32 && dtype != QAVIT_BF16)) return false;
  const size_t vec = dtype == QAVIT_F32 ? 16 : 8;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % vec || (add && (reinterpret_cast<uintptr_t>(add) & 15))) return false;
  constexpr int NW = 4, RB = 4;
  int grid = (rows + NW * RB - 1) / (NW * RB);
  if (grid > 2048) grid = 2048;
  const int np = (C + 255) / 256;
#define LNF(T_, NP_) hipLaunchKernelGGL((layernorm_fwd_v4_kernel<T_, NP_, (NP_ <= 2 ? RB : 2), NW, STATS>), dim3(grid), dim3(64 * NW), 0, st, (const T_*)x, (T_*)y, gamma, beta, eps, rows, C, mean, rstd, add, add_rows, act)
  if (dtype == QAVIT_F32) { if (np == 1) LNF(float, 1); else if (np == 2) LNF(float, 2); else LNF(float, 4); }
  else { if (np == 1) LNF(bf16, 1); else if (np == 2) LNF(bf16, 2); else LNF(bf16, 4); }
#undef LNF
  return true;
}

// More gradients of the same normalised tensor (a LayerNorm output that feeds several consumers): dy = dy + sum ex.p[q], summed in fp32 on load
// instead of by a k-way sum launch in front of this kernel.
struct LnDyExtra { const void* p[4]; int n; };

template <typename T, int NP, int RB, int NW, bool EXTRA = false>
__device__ __forceinline__ void layernorm_bwd_v4_body(const T* dy, const T* x, const float* gamma, const float* mean,
                                                      const float* rstd, T* dx, float* dgamma, float* dbeta, int rows, int C, const float* beta, int act,
                                                      const T* dres = nullptr, float* parts = nullptr, LnDyExtra ex = LnDyExtra{{nullptr, nullptr, nullptr, nullptr}, 0}) {
  // RB rows per wave per iteration: all their loads are issued before the first reduction, so a wave keeps
  // 2*RB*NP vector loads in flight instead of 2 (the row loop is a pure load -> reduce -> store latency chain).
  typedef typename V4<T>::type v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  float pg[NP][4], pb[NP][4], gm[NP][4], bt[NP][4];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = 4 * lane + 256 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) { pg[i][j] = 0.f; pb[i][j] = 0.f; gm[i][j] = (c + j < C) ? gamma[c + j] : 0.f; bt[i][j] = (act && c + j < C) ? beta[c + j] : 0.f; }
  }
  for (int row0 = (blockIdx.x * NW + wave) * RB; row0 < rows; row0 += gridDim.x * NW * RB) {
    v4 xv[RB][NP], dv[RB][NP], rv[RB][NP];
    v4 ev[EXTRA ? 4 : 1][RB][NP];
    float mu[RB], rs[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int row = row0 + r < rows ? row0 + r : rows - 1;       // clamp: tail rows are loaded twice, stored once
      mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < C) {
          xv[r][i] = *reinterpret_cast<const v4*>(x + (size_t)row * C + c);
          dv[r][i] = *reinterpret_cast<const v4*>(dy + (size_t)row * C + c);
          if (dres) rv[r][i] = *reinterpret_cast<const v4*>(dres + (size_t)row * C + c);      // uniform: the other gradient that meets this one at x
          if (EXTRA) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < ex.n) ev[q][r][i] = *reinterpret_cast<const v4*>(reinterpret_cast<const T*>(ex.p[q]) + (size_t)row * C + c);   // uniform
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool live = row0 + r < rows;
      float xh[NP][4], g[NP][4];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < C) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float d = live ? to_f<T>(dv[r][i][j]) : 0.f;
            if (EXTRA) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (q < ex.n && live) d += to_f<T>(ev[q][r][i][j]);
            }
            xh[i][j] = (to_f<T>(xv[r][i][j]) - mu[r]) * rs[r];
            if (act) d *= gelu_grad_f(xh[i][j] * gm[i][j] + bt[i][j]);
            g[i][j] = d * gm[i][j];
            pg[i][j] += d * xh[i][j];
            pb[i][j] += d;
            c1 += g[i][j] * xh[i][j];
            c2 += g[i][j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) { xh[i][j] = 0.f; g[i][j] = 0.f; }
        }
      }
      c1 = wave_sum(c1) * invC;
      c2 = wave_sum(c2) * invC;
      if (live) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int c = 4 * lane + 256 * i;
          if (c < C) {
            v4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = from_f<T>(rs[r] * (g[i][j] - c2 - xh[i][j] * c1) + (dres ? to_f<T>(rv[r][i][j]) : 0.f));
            *reinterpret_cast<v4*>(dx + (size_t)(row0 + r) * C + c) = o;
          }
        }
      }
    }
  }
  __shared__ float red[2][NW][256];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][wave][4 * lane + j] = pg[i][j]; red[1][wave][4 * lane + j] = pb[i][j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 512; t += 64 * NW) {          // 256 channels x {gamma, beta}
      const int which = t >> 8, cc = t & 255, c = cc + 256 * i;
      float* dst = which ? dbeta : dgamma;
      if (c < C && (dst || parts)) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += red[which][w][cc];
        // parts: this workgroup's row of [gridDim.x][2][C] partial sums (plain store; qavit_ln_param_reduce folds them later) --
        // gridDim.x same-address float atomics per channel were ~45 % of this kernel's time
        if (parts) parts[((size_t)blockIdx.x * 2 + which) * C + c] = sacc;
        else atomic_add_f(dst + c, sacc);
      }
    }
  }
}

template <typename T, int NP, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_v4_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                               const float* rstd, T* dx, float* dgamma, float* dbeta, int rows, int C, const float* beta, int act,
                                                               const T* dres, float* parts) {
  layernorm_bwd_v4_body<T, NP, RB, NW>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, beta, act, dres, parts);
}
template <typename T, int NP, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_sum_kernel(const T* dy, LnDyExtra ex, const T* x, const float* gamma, const float* mean,
                                                                    const float* rstd, T* dx, float* dgamma, float* dbeta, int rows, int C, const T* dres, float* parts) {
  layernorm_bwd_v4_body<T, NP, RB, NW, true>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, nullptr, 0, dres, parts, ex);
}
struct LnBwd4 { const void* dy[4]; const void* x[4]; const float* gamma[4]; const float* mean[4]; const float* rstd[4]; void* dx[4]; float* dgamma[4]; float* dbeta[4]; float* parts[4]; };
template <typename T, int NP, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_v4_multi_kernel(LnBwd4 P, int rows, int C) {
  const int i = blockIdx.y;
  layernorm_bwd_v4_body<T, NP, RB, NW>(reinterpret_cast<const T*>(P.dy[i]), reinterpret_cast<const T*>(P.x[i]), P.gamma[i], P.mean[i], P.rstd[i],
                                       reinterpret_cast<T*>(P.dx[i]), P.dgamma[i], P.dbeta[i], rows, C, nullptr, 0, nullptr, P.parts[i]);
}

// SplitFusion's closing pair, blend then LayerNorm (HQAViT_CIFAR100.py:953-965): mixed = s0*a + s1*(t + dropout(h)), s = softmax(fw);
// y = LayerNorm(mixed).  One launch each way instead of two: the forward forms `mixed` while it loads the row (and writes it for the
// backward), the backward turns the LayerNorm input gradient into (da, dt, dh) and the two blend-weight sums in the registers that hold
// it.  Arithmetic and rounding points are those of mix3_vec_kernel (csrc/runtime.hip) followed by the LayerNorm kernels above -- the
// gradient of `mixed` is rounded to T where the two-launch chain stored it.
// GATED (r, g given; a, da unused): the gate in front of the blend is formed in the same registers -- a = t + sigmoid(g) * r
// (HQAViT_CIFAR100.py:945-949, rounded to T where gate_mix_kernel stored it) -- so `a` is never written, and the backward returns ONE gradient
// for t (the gate's pass-through da plus the blend's dt: the two are s0 * dm and s1 * dm of the same dm) next to dr, dg, dh.
struct LnMix3 {
  const void* a; const void* t; const void* h; const float* fw; void* mixed;      // forward: mixed is written; backward: read (the LayerNorm input)
  void* da; void* dt; void* dh; float* dfw;
  float p; int site; const int64_t* rng;
  const void* r; const void* g; void* dr; void* dg;
};
__device__ __forceinline__ void softmax2(const float* fw, float& w0, float& w1) {
  const float mx = fmaxf(fw[0], fw[1]);
  const float e0 = __expf(fw[0] - mx), e1 = __expf(fw[1] - mx), s = e0 + e1;
  w0 = e0 / s; w1 = e1 / s;
}

template <typename T, int RB, int NW, bool GATED = false>
__global__ __launch_bounds__(64 * NW) void mix3_ln_fwd_kernel(LnMix3 mx, T* y, const float* gamma, const float* beta, float eps, int rows, int C,
                                                              float* mean_o, float* rstd_o) {
  typedef typename V4<T>::type v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  const int c = 4 * lane;
  const bool cl = c < C;
  float w0, w1;
  softmax2(mx.fw, w0, w1);
  const uint32_t key = mx.p > 0.f ? rng_key(mx.rng, mx.site) : 0u;
  const float inv = mx.p > 0.f ? 1.f / (1.f - mx.p) : 1.f;
  const T* a = reinterpret_cast<const T*>(mx.a);
  const T* t = reinterpret_cast<const T*>(mx.t);
  const T* h = reinterpret_cast<const T*>(mx.h);
  T* mo = reinterpret_cast<T*>(mx.mixed);
  float gm[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { gm[j] = (c + j < C) ? gamma[c + j] : 0.f; bt[j] = (c + j < C) ? beta[c + j] : 0.f; }
  for (int row0 = (blockIdx.x * NW + wave) * RB; row0 < rows; row0 += gridDim.x * NW * RB) {
    v4 av[RB], tv[RB], hv[RB], gv[GATED ? RB : 1];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int row = row0 + r < rows ? row0 + r : rows - 1;
      if (cl) {
        if (GATED) {
          av[r] = *reinterpret_cast<const v4*>(reinterpret_cast<const T*>(mx.r) + (size_t)row * C + c);
          gv[r] = *reinterpret_cast<const v4*>(reinterpret_cast<const T*>(mx.g) + (size_t)row * C + c);
        } else {
          av[r] = *reinterpret_cast<const v4*>(a + (size_t)row * C + c);
        }
        tv[r] = *reinterpret_cast<const v4*>(t + (size_t)row * C + c);
        hv[r] = *reinterpret_cast<const v4*>(h + (size_t)row * C + c);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool live = row0 + r < rows;
      const int row = live ? row0 + r : rows - 1;
      float v[4];
      v4 m;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cl) {
          const float f = mx.p > 0.f ? drop_factor(key, (uint32_t)row * (uint32_t)C + (uint32_t)(c + j), mx.p, inv) : 1.f;
          const float b = to_f<T>(from_f<T>(to_f<T>(tv[r][j]) + to_f<T>(from_f<T>(to_f<T>(hv[r][j]) * f))));
          float aj = to_f<T>(av[r][j]);
          if (GATED) {
            const float sg = 1.f / (1.f + __expf(-to_f<T>(gv[r][j])));
            aj = to_f<T>(from_f<T>(to_f<T>(tv[r][j]) + sg * aj));
          }
          m[j] = from_f<T>(w0 * aj + w1 * b);
          v[j] = to_f<T>(m[j]);
        } else v[j] = 0.f;
        s += v[j];
      }
      const float mean = wave_sum(s) * invC;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float dlt = cl ? v[j] - mean : 0.f; s2 += dlt * dlt; }
      const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
      if (!live) continue;                                  // uniform per wave
      if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
      if (cl) {
        *reinterpret_cast<v4*>(mo + (size_t)row * C + c) = m;
        v4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = from_f<T>((v[j] - mean) * rstd * gm[j] + bt[j]);
        *reinterpret_cast<v4*>(y + (size_t)row * C + c) = o;
      }
    }
  }
}

template <typename T, int RB, int NW, bool GATED = false>
__global__ __launch_bounds__(64 * NW) void mix3_ln_bwd_kernel(const T* dy, LnMix3 mx, const float* gamma, const float* mean, const float* rstd,
                                                              float* dgamma, float* dbeta, int rows, int C, float* parts) {
  typedef typename V4<T>::type v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  const int c = 4 * lane;
  const bool cl = c < C;
  float w0, w1;
  softmax2(mx.fw, w0, w1);
  const uint32_t key = mx.p > 0.f ? rng_key(mx.rng, mx.site) : 0u;
  const float inv = mx.p > 0.f ? 1.f / (1.f - mx.p) : 1.f;
  const T* a = reinterpret_cast<const T*>(mx.a);
  const T* t = reinterpret_cast<const T*>(mx.t);
  const T* h = reinterpret_cast<const T*>(mx.h);
  const T* x = reinterpret_cast<const T*>(mx.mixed);
  T* da = reinterpret_cast<T*>(mx.da);
  T* dt = reinterpret_cast<T*>(mx.dt);
  T* dh = reinterpret_cast<T*>(mx.dh);
  float pg[4], pb[4], gm[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { pg[j] = 0.f; pb[j] = 0.f; gm[j] = (c + j < C) ? gamma[c + j] : 0.f; }
  float p0 = 0.f, p1 = 0.f;
  for (int row0 = (blockIdx.x * NW + wave) * RB; row0 < rows; row0 += gridDim.x * NW * RB) {
    v4 xv[RB], dv[RB], av[RB], tv[RB], hv[RB], gv[GATED ? RB : 1];
    float mu[RB], rs[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int row = row0 + r < rows ? row0 + r : rows - 1;       // clamp: tail rows are loaded twice, stored once
      mu[r] = mean[row]; rs[r] = rstd[row];
      if (cl) {
        xv[r] = *reinterpret_cast<const v4*>(x + (size_t)row * C + c);
        dv[r] = *reinterpret_cast<const v4*>(dy + (size_t)row * C + c);
        if (GATED) {
          av[r] = *reinterpret_cast<const v4*>(reinterpret_cast<const T*>(mx.r) + (size_t)row * C + c);
          gv[r] = *reinterpret_cast<const v4*>(reinterpret_cast<const T*>(mx.g) + (size_t)row * C + c);
        } else {
          av[r] = *reinterpret_cast<const v4*>(a + (size_t)row * C + c);
        }
        tv[r] = *reinterpret_cast<const v4*>(t + (size_t)row * C + c);
        hv[r] = *reinterpret_cast<const v4*>(h + (size_t)row * C + c);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool live = row0 + r < rows;
      const int row = live ? row0 + r : rows - 1;
      float xh[4], g[4];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cl) {
          const float d = live ? to_f<T>(dv[r][j]) : 0.f;
          xh[j] = (to_f<T>(xv[r][j]) - mu[r]) * rs[r];
          g[j] = d * gm[j];
          pg[j] += d * xh[j];
          pb[j] += d;
          c1 += g[j] * xh[j];
          c2 += g[j];
        } else { xh[j] = 0.f; g[j] = 0.f; }
      }
      c1 = wave_sum(c1) * invC;
      c2 = wave_sum(c2) * invC;
      if (live && cl) {
        v4 x0, x1, x2, x3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gmix = to_f<T>(from_f<T>(rs[r] * (g[j] - c2 - xh[j] * c1)));      // d(mixed), rounded where the LayerNorm-backward launch stored it
          const float f = mx.p > 0.f ? drop_factor(key, (uint32_t)row * (uint32_t)C + (uint32_t)(c + j), mx.p, inv) : 1.f;
          const float b = to_f<T>(from_f<T>(to_f<T>(tv[r][j]) + to_f<T>(from_f<T>(to_f<T>(hv[r][j]) * f))));
          x0[j] = from_f<T>(gmix * w0);
          x1[j] = from_f<T>(gmix * w1);
          x2[j] = from_f<T>(to_f<T>(x1[j]) * f);
          float aj = to_f<T>(av[r][j]);
          if (GATED) {                                        // gate_mix_kernel's backward on dy = da (x0), and the one gradient of t
            const float rj = aj, sg = 1.f / (1.f + __expf(-to_f<T>(gv[r][j]))), d = to_f<T>(x0[j]);
            aj = to_f<T>(from_f<T>(to_f<T>(tv[r][j]) + sg * rj));
            x1[j] = from_f<T>(d + to_f<T>(x1[j]));
            x0[j] = from_f<T>(d * sg);
            x3[j] = from_f<T>(d * rj * sg * (1.f - sg));
          }
          p0 += gmix * aj; p1 += gmix * b;
        }
        if (GATED) {
          *reinterpret_cast<v4*>(reinterpret_cast<T*>(mx.dr) + (size_t)row * C + c) = x0;
          *reinterpret_cast<v4*>(reinterpret_cast<T*>(mx.dg) + (size_t)row * C + c) = x3;
        } else {
          *reinterpret_cast<v4*>(da + (size_t)row * C + c) = x0;
        }
        *reinterpret_cast<v4*>(dt + (size_t)row * C + c) = x1;
        *reinterpret_cast<v4*>(dh + (size_t)row * C + c) = x2;
      }
    }
  }
  __shared__ float red[2][NW][256];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][wave][4 * lane + j] = pg[j]; red[1][wave][4 * lane + j] = pb[j]; }
  __syncthreads();
  for (int q = threadIdx.x; q < 512; q += 64 * NW) {            // 256 channels x {gamma, beta}
    const int which = q >> 8, cc = q & 255;
    float* dst = which ? dbeta : dgamma;
    if (cc < C && (dst || parts)) {
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sacc += red[which][w][cc];
      if (parts) parts[((size_t)blockIdx.x * 2 + which) * C + cc] = sacc;      // the partial-row layout of layernorm_bwd_v4_body
      else atomic_add_f(dst + cc, sacc);
    }
  }
  if (mx.dfw) {                                              // the blend weights through their softmax
    __syncthreads();
    const float s0 = wave_sum(p0), s1 = wave_sum(p1);
    if (lane == 0) { red[0][wave][0] = s0; red[1][wave][0] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { d0 += red[0][w][0]; d1 += red[1][w][0]; }
      const float dot = d0 * w0 + d1 * w1;
      // with partial rows: this workgroup's [4] after the gridDim.x LayerNorm rows (reduce descriptor C = 2: a fixed-order fold, the two
      // logits' gradients are the same bits on every run); else one pair of atomics per workgroup
      if (parts) *reinterpret_cast<f32x4*>(parts + (size_t)gridDim.x * 2 * C + (size_t)blockIdx.x * 4) = f32x4{w0 * (d0 - dot), w1 * (d1 - dot), 0.f, 0.f};
      else { atomic_add_f(mx.dfw + 0, w0 * (d0 - dot)); atomic_add_f(mx.dfw + 1, w1 * (d1 - dot)); }
    }
  }
}

// LayerNorm backward of a LayerNorm-prologue Linear with a NARROW output (TokenLearner's score Linear: 192 -> 16), fused with that
// Linear's input-gradient GEMM: dxn[row][c] = sum_k dz[row][k] W[k][c] is a 16-deep dot product per element, cheaper to redo in the
// registers of the thread that owns (row, c) than to write a [rows, C] matrix from a GEMM launch and read it back here
// (65536 x 192: 25 MB out + 25 MB in + a launch, per block).  W's column slice of the lane (KZ x 4 values) stays in registers.
template <int KZ, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_lin_kernel(const bf16* dz, int ldz, const bf16* W, int ldw, const bf16* x, const float* gamma,
                                                                    const float* mean, const float* rstd, bf16* dx, float* dgamma, float* dbeta,
                                                                    int rows, int C, const bf16* dres, float* parts) {
  static_assert(KZ % 8 == 0, "dz rows are read as 16-byte vectors");
  __shared__ __attribute__((aligned(16))) float wl[KZ][256];          // W as fp32, one 16-byte read per (k, lane); also the flush scratch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  const int c = 4 * lane;
  const bool cl = c < C;                                   // C <= 256: one 4-column slice per lane
  for (int i = threadIdx.x; i < KZ * 256; i += 64 * NW) {
    const int k = i >> 8, cc = i & 255;
    wl[k][cc] = cc < C ? (float)W[(size_t)k * ldw + cc] : 0.f;
  }
  float pg[4], pb[4], gm[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { pg[j] = 0.f; pb[j] = 0.f; gm[j] = cl ? gamma[c + j] : 0.f; }
  __syncthreads();
  for (int row0 = (blockIdx.x * NW + wave) * RB; row0 < rows; row0 += gridDim.x * NW * RB) {
    bf16x4 xv[RB], rv[RB];
    bf16x8 zv[RB][KZ / 8];
    float mu[RB], rs[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int row = row0 + r < rows ? row0 + r : rows - 1;       // clamp: tail rows are loaded twice, stored once
      mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
      for (int v = 0; v < KZ / 8; ++v) zv[r][v] = *reinterpret_cast<const bf16x8*>(dz + (size_t)row * ldz + 8 * v);   // same address in every lane
      if (cl) {
        xv[r] = *reinterpret_cast<const bf16x4*>(x + (size_t)row * C + c);
        if (dres) rv[r] = *reinterpret_cast<const bf16x4*>(dres + (size_t)row * C + c);
      }
    }
    f32x2 d2[RB][2];                                            // (d[0], d[1]), (d[2], d[3]): the dot products run as v_pk_fma_f32
#pragma unroll
    for (int r = 0; r < RB; ++r) { d2[r][0] = f32x2{0.f, 0.f}; d2[r][1] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < KZ; ++k) {                             // one weight read serves the RB rows in flight
      const f32x4 wk = *reinterpret_cast<const f32x4*>(&wl[k][c]);
      const f32x2 w01 = f32x2{wk[0], wk[1]}, w23 = f32x2{wk[2], wk[3]};
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float z = (float)zv[r][k >> 3][k & 7];
        const f32x2 zz = f32x2{z, z};
        d2[r][0] = __builtin_elementwise_fma(zz, w01, d2[r][0]);
        d2[r][1] = __builtin_elementwise_fma(zz, w23, d2[r][1]);
      }
      if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most four weight reads ahead: all sixteen hoisted cost 64 registers (spills)
    }
    float d[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r) { d[r][0] = d2[r][0][0]; d[r][1] = d2[r][0][1]; d[r][2] = d2[r][1][0]; d[r][3] = d2[r][1][1]; }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const bool live = row0 + r < rows;
      float xh[4], g[4];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dd = (live && cl) ? d[r][j] : 0.f;
        xh[j] = cl ? ((float)xv[r][j] - mu[r]) * rs[r] : 0.f;
        g[j] = dd * gm[j];
        pg[j] += dd * xh[j];
        pb[j] += dd;
        c1 += g[j] * xh[j];
        c2 += g[j];
      }
      c1 = wave_sum(c1) * invC;
      c2 = wave_sum(c2) * invC;
      if (live && cl) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(rs[r] * (g[j] - c2 - xh[j] * c1) + (dres ? (float)rv[r][j] : 0.f));
        *reinterpret_cast<bf16x4*>(dx + (size_t)(row0 + r) * C + c) = o;
      }
    }
  }
  static_assert(KZ >= 2 * NW, "the flush folds 2 x NW rows of 256 floats in the weight tile");
  __syncthreads();
  float (*red)[NW][256] = reinterpret_cast<float (*)[NW][256]>(&wl[0][0]);
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][wave][4 * lane + j] = pg[j]; red[1][wave][4 * lane + j] = pb[j]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 512; t += 64 * NW) {
    const int which = t >> 8, cc = t & 255;
    float* dst = which ? dbeta : dgamma;
    if (cc < C && (dst || parts)) {
      float sacc = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) sacc += red[which][wv][cc];
      if (parts) parts[((size_t)blockIdx.x * 2 + which) * C + cc] = sacc;
      else atomic_add_f(dst + cc, sacc);
    }
  }
}

// dgamma / dbeta += sum over the partial rows a layernorm_bwd launch left in ``parts`` ([nparts][2][C]): one workgroup per LayerNorm,
// threads = (group, 4-float vector of the 2C-wide row), groups stride over the rows with four loads in flight, LDS fold, then ONE
// atomic per channel (atomic because a shared parameter may collect from several LayerNorm calls).
// NARROW descriptors (C <= 4: scalar layer scales, blend / fusion logits) are folded by ONE workgroup whatever nparts is, rows in a fixed
// order: their sums are bit-reproducible from run to run (the wide ones are, too, up to LNR_SLICE partial rows).  C == 1: rows are
// [dgamma, dbeta, pad, pad]; C == 2: [dgamma0, dgamma1, dbeta0, dbeta1].
struct LnReduceGroup { int n; qavit_ln_reduce_desc d[48]; };
constexpr int LNR_SLICE = 32;                              // partial rows per workgroup (blockIdx.y = slice)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(LnReduceGroup G) {
  __shared__ __attribute__((aligned(16))) float fold[4096];
  const qavit_ln_reduce_desc d = G.d[blockIdx.x];
  const bool narrow = d.C <= 4;
  const int p0 = narrow ? 0 : blockIdx.y * LNR_SLICE;
  if (p0 >= d.nparts || (narrow && blockIdx.y)) return;   // uniform per workgroup
  const int p1 = narrow ? d.nparts : (p0 + LNR_SLICE < d.nparts ? p0 + LNR_SLICE : d.nparts);
  const int W = d.C == 1 ? 4 : 2 * d.C, V = W >> 2;
  const int groups = narrow ? 64 / V : 1024 / V;           // narrow: 64 / 32 threads sweep the rows, one serial fold of their sums
  const size_t RS = d.stride > 0 ? (size_t)d.stride : (size_t)W;          // floats between partial rows
  const int g = threadIdx.x / V, v = threadIdx.x - g * V;
  if (g < groups) {
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* base = d.parts + 4 * v;
    int p = p0 + g;
    for (; p + 3 * groups < p1; p += 4 * groups) {
      f32x4 t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const f32x4*>(base + (size_t)(p + q * groups) * RS);
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q][0] += t[q][0]; a[q][1] += t[q][1]; a[q][2] += t[q][2]; a[q][3] += t[q][3]; }
    }
    for (; p < p1; p += groups) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(base + (size_t)p * RS);
      a[0][0] += t[0]; a[0][1] += t[1]; a[0][2] += t[2]; a[0][3] += t[3];
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (a[0][j] + a[1][j]) + (a[2][j] + a[3][j]);
    *reinterpret_cast<f32x4*>(fold + g * W + 4 * v) = o;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < W; c += 1024) {
    float t = 0.f;
    for (int q = 0; q < groups; ++q) t += fold[q * W + c];
    if (d.C == 1) { if (c < 2) { float* dst = c ? d.dbeta : d.dgamma; if (dst) atomic_add_f(dst, t); } continue; }
    float* dst = c < d.C ? d.dgamma : d.dbeta;
    if (dst) atomic_add_f(dst + (c < d.C ? c : c - d.C), t);
  }
}

static int ln_pl(int C) { const int p = (C + 63) / 64; return p <= 1 ? 1 : p <= 2 ? 2 : p <= 3 ? 3 : p <= 4 ? 4 : p <= 8 ? 8 : 16; }

#define LN_DISPATCH(PL, CALL)            \
  switch (PL) {                          \
    case 1: { constexpr int P = 1; CALL; } break;   \
    case 2: { constexpr int P = 2; CALL; } break;   \
    case 3: { constexpr int P = 3; CALL; } break;   \
    case 4: { constexpr int P = 4; CALL; } break;   \
    case 8: { constexpr int P = 8; CALL; } break;   \
    default: { constexpr int P = 16; CALL; } break; \
  }

extern "C" int qavit_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta,
                                   float eps, int rows, int C, float* mean, float* rstd,
                                   const float* add, int add_rows, int act, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "layernorm_fwd: bad arguments");
  if (C > LN_MAX_C) return set_error(QAVIT_EINVAL, "layernorm_fwd: C > 1024 unsupported");
  if (add && add_rows <= 0) return set_error(QAVIT_EINVAL, "layernorm_fwd: add_rows must be positive");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ln_fwd_v4_launch<false>(dtype, x, y, gamma, beta, eps, rows, C, mean, rstd, add, add_rows, act, st)) return check_launch("layernorm_fwd");
  int grid = (rows + 3) / 4;
  if (grid > 4096) grid = 4096;
  const int pl = ln_pl(C);
  if (dtype == QAVIT_F32) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((layernorm_fwd_kernel<float, P>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y, gamma, beta, eps, rows, C, mean, rstd, add, add_rows, act))
  } else if (dtype == QAVIT_BF16) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, P>), dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)y, gamma, beta, eps, rows, C, mean, rstd, add, add_rows, act))
  } else return set_error(QAVIT_EINVAL, "layernorm_fwd: unknown dtype");
  return check_launch("layernorm_fwd");
}

extern "C" int qavit_row_stats(int dtype, const void* x, float eps, int rows, int C, float* mean, float* rstd, void* stream) {
  if (!x || !mean || !rstd || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "row_stats: bad arguments");
  if (C > LN_MAX_C) return set_error(QAVIT_EINVAL, "row_stats: C > 1024 unsupported");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ln_fwd_v4_launch<true>(dtype, x, const_cast<void*>(x), nullptr, nullptr, eps, rows, C, mean, rstd, nullptr, 0, 0, st)) return check_launch("row_stats");
  int grid = (rows + 3) / 4;
  if (grid > 4096) grid = 4096;
  const int pl = ln_pl(C);
  if (dtype == QAVIT_F32) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((row_stats_kernel<float, P>), dim3(grid), dim3(256), 0, st, (const float*)x, eps, rows, C, mean, rstd))
  } else if (dtype == QAVIT_BF16) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((row_stats_kernel<bf16, P>), dim3(grid), dim3(256), 0, st, (const bf16*)x, eps, rows, C, mean, rstd))
  } else return set_error(QAVIT_EINVAL, "row_stats: unknown dtype");
  return check_launch("row_stats");
}

extern "C" int qavit_layernorm_bwd_parts(int rows, int C) {
  (void)C;
  static const int cap = getenv("QAVIT_LNB_GRID") ? atoi(getenv("QAVIT_LNB_GRID")) : 256;
  int grid = (rows + 63) / 64;            // 16 waves x 4 rows per workgroup and pass
  if (grid > cap) grid = cap;
  return grid < 1 ? 1 : grid;
}

extern "C" int qavit_layernorm_bwd_sum(int dtype, int n_dy, const void* const* dy, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int C, const void* dres, float* part_ws,
                                       void* stream) {
  if (!dy || n_dy < 1 || n_dy > 5 || !x || !gamma || !mean || !rstd || !dx || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "layernorm_bwd_sum: bad arguments");
  const size_t esz = dtype == QAVIT_F32 ? 4 : 2;
  if ((dtype != QAVIT_F32 && dtype != QAVIT_BF16) || C % 4 || C > 256) return set_error(QAVIT_EINVAL, "layernorm_bwd_sum: fp32 / bf16, C % 4 == 0, C <= 256");
  uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) | (dres ? reinterpret_cast<uintptr_t>(dres) : 0);
  LnDyExtra ex{{nullptr, nullptr, nullptr, nullptr}, n_dy - 1};
  for (int i = 0; i < n_dy; ++i) {
    if (!dy[i]) return set_error(QAVIT_EINVAL, "layernorm_bwd_sum: null gradient");
    al |= reinterpret_cast<uintptr_t>(dy[i]);
    if (i > 0) ex.p[i - 1] = dy[i];
  }
  if (al % (4 * esz) || (part_ws && (reinterpret_cast<uintptr_t>(part_ws) & 15))) return set_error(QAVIT_EINVAL, "layernorm_bwd_sum: vector-aligned operands");
  constexpr int NW = 16;
  const int grid = qavit_layernorm_bwd_parts(rows, C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32)
    hipLaunchKernelGGL((layernorm_bwd_sum_kernel<float, 1, 2, NW>), dim3(grid), dim3(64 * NW), 0, st, (const float*)dy[0], ex, (const float*)x, gamma, mean, rstd,
                       (float*)dx, dgamma, dbeta, rows, C, (const float*)dres, part_ws);
  else
    hipLaunchKernelGGL((layernorm_bwd_sum_kernel<bf16, 1, 2, NW>), dim3(grid), dim3(64 * NW), 0, st, (const bf16*)dy[0], ex, (const bf16*)x, gamma, mean, rstd,
                       (bf16*)dx, dgamma, dbeta, rows, C, (const bf16*)dres, part_ws);
  return check_launch("layernorm_bwd_sum");
}

extern "C" int qavit_mix3_ln_supported(int dtype, int C) { return (dtype == QAVIT_BF16 || dtype == QAVIT_F32) && C % 4 == 0 && C <= 256; }

static int mix3_ln_check(const char* what, int dtype, int rows, int C, float drop_p, const int64_t* rng, uintptr_t al, const float* part_ws) {
  if (!qavit_mix3_ln_supported(dtype, C)) return set_error(QAVIT_EINVAL, "mix3_ln: fp32 / bf16, C % 4 == 0, C <= 256");
  if (rows <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !rng)) return set_error(QAVIT_EINVAL, what);
  if ((int64_t)rows * C > 0xffffffffll) return set_error(QAVIT_EINVAL, "mix3_ln: the dropout index is 32 bits");
  const size_t esz = dtype == QAVIT_F32 ? 4 : 2;
  if (al % (4 * esz) || (part_ws && (reinterpret_cast<uintptr_t>(part_ws) & 15))) return set_error(QAVIT_EINVAL, "mix3_ln: vector-aligned operands");
  return QAVIT_OK;
}

static int mix3_ln_fwd_launch(int dtype, const LnMix3& mx, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd, int rows, int C,
                              hipStream_t st, bool gated) {
  constexpr int NW = 4, RB = 2;                              // (four rows in flight per wave: 83 registers in the gated form, 0.01 ms per step slower)
  int grid = (rows + NW * RB - 1) / (NW * RB);
  if (grid > 2048) grid = 2048;
#define MLF(T_, G_) hipLaunchKernelGGL((mix3_ln_fwd_kernel<T_, RB, NW, G_>), dim3(grid), dim3(64 * NW), 0, st, mx, (T_*)y, gamma, beta, eps, rows, C, mean, rstd)
  if (dtype == QAVIT_F32) { if (gated) MLF(float, true); else MLF(float, false); }
  else { if (gated) MLF(bf16, true); else MLF(bf16, false); }
#undef MLF
  return check_launch("mix3_ln_fwd");
}

// partial rows of the backward (the LayerNorm parameter gradients): 8 waves x 4 rows (fp32: 2) per workgroup and pass, two workgroups per CU
extern "C" int qavit_mix3_ln_bwd_parts(int rows, int C) {
  (void)C;
  int grid = (rows + 31) / 32;
  if (grid > 512) grid = 512;
  return grid < 1 ? 1 : grid;
}

static int mix3_ln_bwd_launch(int dtype, const void* dy, const LnMix3& mx, const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                              int rows, int C, float* part_ws, hipStream_t st, bool gated) {
  constexpr int NW = 8;
  const int grid = qavit_mix3_ln_bwd_parts(rows, C);
#define MLB(T_, RB_, G_) hipLaunchKernelGGL((mix3_ln_bwd_kernel<T_, RB_, NW, G_>), dim3(grid), dim3(64 * NW), 0, st, (const T_*)dy, mx, gamma, mean, rstd, dgamma, dbeta, rows, C, part_ws)
  if (dtype == QAVIT_F32) { if (gated) MLB(float, 2, true); else MLB(float, 2, false); }
  else { if (gated) MLB(bf16, 2, true); else MLB(bf16, 4, false); }
#undef MLB
  return check_launch("mix3_ln_bwd");
}

extern "C" int qavit_mix3_ln_fwd(int dtype, const void* a, const void* t, const void* h, const float* fw, float drop_p, int drop_site, const int64_t* rng,
                                 void* mixed, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd, int rows, int C,
                                 void* stream) {
  if (!a || !t || !h || !fw || !mixed || !gamma || !beta || !y || !mean || !rstd) return set_error(QAVIT_EINVAL, "mix3_ln_fwd: bad arguments");
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(mixed) |
                       reinterpret_cast<uintptr_t>(y);
  const int rc = mix3_ln_check("mix3_ln_fwd: bad arguments", dtype, rows, C, drop_p, rng, al, nullptr);
  if (rc) return rc;
  LnMix3 mx{a, t, h, fw, mixed, nullptr, nullptr, nullptr, nullptr, drop_p, drop_site, rng, nullptr, nullptr, nullptr, nullptr};
  return mix3_ln_fwd_launch(dtype, mx, gamma, beta, eps, y, mean, rstd, rows, C, reinterpret_cast<hipStream_t>(stream), false);
}

extern "C" int qavit_mix3_ln_bwd(int dtype, const void* dy, const void* a, const void* t, const void* h, const float* fw, float drop_p, int drop_site,
                                 const int64_t* rng, const void* mixed, const float* gamma, const float* mean, const float* rstd, void* da, void* dt,
                                 void* dh, float* dfw, float* dgamma, float* dbeta, int rows, int C, float* part_ws, void* stream) {
  if (!dy || !a || !t || !h || !fw || !mixed || !gamma || !mean || !rstd || !da || !dt || !dh) return set_error(QAVIT_EINVAL, "mix3_ln_bwd: bad arguments");
  const uintptr_t al = reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(h) |
                       reinterpret_cast<uintptr_t>(mixed) | reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(dt) | reinterpret_cast<uintptr_t>(dh);
  const int rc = mix3_ln_check("mix3_ln_bwd: bad arguments", dtype, rows, C, drop_p, rng, al, part_ws);
  if (rc) return rc;
  LnMix3 mx{a, t, h, fw, const_cast<void*>(mixed), da, dt, dh, dfw, drop_p, drop_site, rng, nullptr, nullptr, nullptr, nullptr};
  return mix3_ln_bwd_launch(dtype, dy, mx, gamma, mean, rstd, dgamma, dbeta, rows, C, part_ws, reinterpret_cast<hipStream_t>(stream), false);
}

extern "C" int qavit_gate_mix3_ln_fwd(int dtype, const void* t, const void* r, const void* g, const void* h, const float* fw, float drop_p, int drop_site,
                                      const int64_t* rng, void* mixed, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                                      int rows, int C, void* stream) {
  if (!t || !r || !g || !h || !fw || !mixed || !gamma || !beta || !y || !mean || !rstd) return set_error(QAVIT_EINVAL, "gate_mix3_ln_fwd: bad arguments");
  const uintptr_t al = reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(h) |
                       reinterpret_cast<uintptr_t>(mixed) | reinterpret_cast<uintptr_t>(y);
  const int rc = mix3_ln_check("gate_mix3_ln_fwd: bad arguments", dtype, rows, C, drop_p, rng, al, nullptr);
  if (rc) return rc;
  LnMix3 mx{nullptr, t, h, fw, mixed, nullptr, nullptr, nullptr, nullptr, drop_p, drop_site, rng, r, g, nullptr, nullptr};
  return mix3_ln_fwd_launch(dtype, mx, gamma, beta, eps, y, mean, rstd, rows, C, reinterpret_cast<hipStream_t>(stream), true);
}

extern "C" int qavit_gate_mix3_ln_bwd(int dtype, const void* dy, const void* t, const void* r, const void* g, const void* h, const float* fw, float drop_p,
                                      int drop_site, const int64_t* rng, const void* mixed, const float* gamma, const float* mean, const float* rstd,
                                      void* dt, void* dr, void* dg, void* dh, float* dfw, float* dgamma, float* dbeta, int rows, int C, float* part_ws,
                                      void* stream) {
  if (!dy || !t || !r || !g || !h || !fw || !mixed || !gamma || !mean || !rstd || !dt || !dr || !dg || !dh)
    return set_error(QAVIT_EINVAL, "gate_mix3_ln_bwd: bad arguments");
  const uintptr_t al = reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(g) |
                       reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(mixed) | reinterpret_cast<uintptr_t>(dt) | reinterpret_cast<uintptr_t>(dr) |
                       reinterpret_cast<uintptr_t>(dg) | reinterpret_cast<uintptr_t>(dh);
  const int rc = mix3_ln_check("gate_mix3_ln_bwd: bad arguments", dtype, rows, C, drop_p, rng, al, part_ws);
  if (rc) return rc;
  LnMix3 mx{nullptr, t, h, fw, const_cast<void*>(mixed), nullptr, dt, dh, dfw, drop_p, drop_site, rng, r, g, dr, dg};
  return mix3_ln_bwd_launch(dtype, dy, mx, gamma, mean, rstd, dgamma, dbeta, rows, C, part_ws, reinterpret_cast<hipStream_t>(stream), true);
}

extern "C" int qavit_layernorm_bwd_lin_supported(int dtype, int KZ, int C) { return dtype == QAVIT_BF16 && KZ == 16 && C % 4 == 0 && C <= 256; }

extern "C" int qavit_layernorm_bwd_lin(int dtype, const void* dz, int ldz, const void* W, int ldw, int KZ, const void* x, const float* gamma,
                                       const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int C,
                                       const void* dres, float* part_ws, void* stream) {
  if (!dz || !W || !x || !gamma || !mean || !rstd || !dx || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "layernorm_bwd_lin: bad arguments");
  if (!qavit_layernorm_bwd_lin_supported(dtype, KZ, C)) return set_error(QAVIT_EINVAL, "layernorm_bwd_lin: bf16, 16 Linear outputs, C % 4 == 0, C <= 256 only");
  if (ldz % 8 || ldw % 4 || ldz < KZ || ldw < C ||
      ((reinterpret_cast<uintptr_t>(dz) & 15) | (reinterpret_cast<uintptr_t>(W) & 7) | (reinterpret_cast<uintptr_t>(x) & 7) | (reinterpret_cast<uintptr_t>(dx) & 7) |
       (dres ? reinterpret_cast<uintptr_t>(dres) & 7 : 0) | (part_ws ? reinterpret_cast<uintptr_t>(part_ws) & 15 : 0)))
    return set_error(QAVIT_EINVAL, "layernorm_bwd_lin: operand alignment / leading dimensions");
  constexpr int NW = 8;                                      // four rows in flight per wave (eight measured slower: 34 vs 31.5 us at 65536 rows)
  const int grid = qavit_layernorm_bwd_parts(rows, C);       // the same partial-row count as qavit_layernorm_bwd
  hipLaunchKernelGGL((layernorm_bwd_lin_kernel<16, 4, NW>), dim3(grid), dim3(64 * NW), 0, reinterpret_cast<hipStream_t>(stream), (const bf16*)dz, ldz,
                     (const bf16*)W, ldw, (const bf16*)x, gamma, mean, rstd, (bf16*)dx, dgamma, dbeta, rows, C, (const bf16*)dres, part_ws);
  return check_launch("layernorm_bwd_lin");
}

extern "C" int qavit_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                                   const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                   int rows, int C, float* dadd, int add_rows, const float* beta, int act, const void* dres, float* part_ws, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "layernorm_bwd: bad arguments");
  if (act && !beta) return set_error(QAVIT_EINVAL, "layernorm_bwd: the fused-GELU gradient needs beta");
  if (C > LN_MAX_C) return set_error(QAVIT_EINVAL, "layernorm_bwd: C > 1024 unsupported");
  if (dadd && add_rows <= 0) return set_error(QAVIT_EINVAL, "layernorm_bwd: add_rows must be positive");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t esz0 = dtype == QAVIT_F32 ? 4 : 2;
  if (dadd && C % 4 == 0 && reinterpret_cast<uintptr_t>(dy) % (4 * esz0) == 0 && (dtype == QAVIT_F32 || dtype == QAVIT_BF16)) {
    // batch sum of dy in a launch of its own; the LayerNorm gradient below then runs without dadd (vector kernel)
    const long total = (long)rows * C;
    const int P = add_rows * C;
    const int reps = (int)((total + P - 1) / P);
    const int gx = (P / 4 + 255) / 256;
    int slices = 512 / gx;
    if (slices < 1) slices = 1;
    if (slices > reps) slices = reps;
    const int per = (reps + slices - 1) / slices;
    slices = (reps + per - 1) / per;
    if (dtype == QAVIT_F32) hipLaunchKernelGGL((ln_dadd_kernel<float>), dim3(gx, slices), dim3(256), 0, st, (const float*)dy, dadd, total, P, per);
    else hipLaunchKernelGGL((ln_dadd_kernel<bf16>), dim3(gx, slices), dim3(256), 0, st, (const bf16*)dy, dadd, total, P, per);
    dadd = nullptr;
  }
  int grid = (rows + 3) / 4;
  if (grid > 512) grid = 512;     // every workgroup ends with 2*C same-address atomics: keep the flush small
  const size_t esz = dtype == QAVIT_F32 ? 4 : 2;
  const bool v4ok = !dadd && (C % 4 == 0) && C <= 1024 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) % (4 * esz) == 0);
  if (dres && (!v4ok || (reinterpret_cast<uintptr_t>(dres) % (4 * esz)) != 0))
    return set_error(QAVIT_EINVAL, "layernorm_bwd: dres needs C % 4 == 0, no dadd and vector-aligned operands");
  if (part_ws && (!v4ok || (reinterpret_cast<uintptr_t>(part_ws) & 15)))
    return set_error(QAVIT_EINVAL, "layernorm_bwd: part_ws needs C % 4 == 0, vector-aligned operands and a 16-byte aligned workspace");
  if (v4ok) {
    const int np = (C + 255) / 256;
    constexpr int NW = 16;
    grid = qavit_layernorm_bwd_parts(rows, C);
#define LNV(T_, NP_) hipLaunchKernelGGL((layernorm_bwd_v4_kernel<T_, NP_, (NP_ <= 2 ? 4 : 2), NW>), dim3(grid), dim3(64 * NW), 0, st, (const T_*)dy, (const T_*)x, gamma, mean, rstd, (T_*)dx, dgamma, dbeta, rows, C, beta, act, (const T_*)dres, part_ws)
    if (dtype == QAVIT_F32) { if (np == 1) LNV(float, 1); else if (np == 2) LNV(float, 2); else LNV(float, 4); }
    else if (dtype == QAVIT_BF16) { if (np == 1) LNV(bf16, 1); else if (np == 2) LNV(bf16, 2); else LNV(bf16, 4); }
    else return set_error(QAVIT_EINVAL, "layernorm_bwd: unknown dtype");
#undef LNV
    return check_launch("layernorm_bwd");
  }
  const int pl = ln_pl(C);
  if (dtype == QAVIT_F32) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((layernorm_bwd_kernel<float, P>), dim3(grid), dim3(256), 0, st, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, dgamma, dbeta, rows, C, dadd, add_rows, beta, act))
  } else if (dtype == QAVIT_BF16) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((layernorm_bwd_kernel<bf16, P>), dim3(grid), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (bf16*)dx, dgamma, dbeta, rows, C, dadd, add_rows, beta, act))
  } else return set_error(QAVIT_EINVAL, "layernorm_bwd: unknown dtype");
  return check_launch("layernorm_bwd");
}


extern "C" int qavit_row_stats_multi(int dtype, int n, const void* const* x, float eps, int rows, int C, float* const* mean, float* const* rstd, void* stream) {
  if (!x || !mean || !rstd || n <= 0 || n > 4 || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "row_stats_multi: bad arguments (1 <= n <= 4)");
  if (C > LN_MAX_C) return set_error(QAVIT_EINVAL, "row_stats_multi: C > 1024 unsupported");
  Ptr4 ptrs;
  for (int i = 0; i < n; ++i) {
    if (!x[i] || !mean[i] || !rstd[i]) return set_error(QAVIT_EINVAL, "row_stats_multi: null operand");
    ptrs.x[i] = x[i]; ptrs.mean[i] = mean[i]; ptrs.rstd[i] = rstd[i];
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    const size_t vec = dtype == QAVIT_F32 ? 16 : 8;
    bool v4ok = C % 4 == 0 && C <= 512 && (dtype == QAVIT_F32 || dtype == QAVIT_BF16);
    for (int i = 0; i < n && v4ok; ++i) v4ok = reinterpret_cast<uintptr_t>(x[i]) % vec == 0;
    if (v4ok) {
      Ptr4v P;
      for (int i = 0; i < n; ++i) { P.x[i] = x[i]; P.mean[i] = mean[i]; P.rstd[i] = rstd[i]; }
      constexpr int NW = 4, RB = 4;
      int g = (rows + NW * RB - 1) / (NW * RB);
      if (g > 2048 / n) g = 2048 / n;
      const int np = (C + 255) / 256;
      if (dtype == QAVIT_F32) { if (np == 1) hipLaunchKernelGGL((row_stats_v4_multi_kernel<float, 1, RB, NW>), dim3(g, n), dim3(64 * NW), 0, st, P, eps, rows, C);
                                else hipLaunchKernelGGL((row_stats_v4_multi_kernel<float, 2, RB, NW>), dim3(g, n), dim3(64 * NW), 0, st, P, eps, rows, C); }
      else { if (np == 1) hipLaunchKernelGGL((row_stats_v4_multi_kernel<bf16, 1, RB, NW>), dim3(g, n), dim3(64 * NW), 0, st, P, eps, rows, C);
             else hipLaunchKernelGGL((row_stats_v4_multi_kernel<bf16, 2, RB, NW>), dim3(g, n), dim3(64 * NW), 0, st, P, eps, rows, C); }
      return check_launch("row_stats_multi");
    }
  }
  int grid = (rows + 3) / 4;
  if (grid > 4096 / n) grid = 4096 / n;
  const int pl = ln_pl(C);
  if (dtype == QAVIT_F32) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((row_stats_multi_kernel<float, P>), dim3(grid, n), dim3(256), 0, st, ptrs, eps, rows, C))
  } else if (dtype == QAVIT_BF16) {
    LN_DISPATCH(pl, hipLaunchKernelGGL((row_stats_multi_kernel<bf16, P>), dim3(grid, n), dim3(256), 0, st, ptrs, eps, rows, C))
  } else return set_error(QAVIT_EINVAL, "row_stats_multi: unknown dtype");
  return check_launch("row_stats_multi");
}

extern "C" int qavit_layernorm_bwd_multi(int dtype, int n, const void* const* dy, const void* const* x, const float* const* gamma,
                                         const float* const* mean, const float* const* rstd, void* const* dx,
                                         float* const* dgamma, float* const* dbeta, int rows, int C, float* const* part_ws, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || n <= 0 || n > 4 || rows <= 0 || C <= 0)
    return set_error(QAVIT_EINVAL, "layernorm_bwd_multi: bad arguments (1 <= n <= 4)");
  const size_t esz = dtype == QAVIT_F32 ? 4 : 2;
  bool v4ok = (C % 4 == 0) && C <= 512;
  for (int i = 0; i < n && v4ok; ++i)
    v4ok = ((reinterpret_cast<uintptr_t>(x[i]) | reinterpret_cast<uintptr_t>(dy[i]) | reinterpret_cast<uintptr_t>(dx[i])) % (4 * esz)) == 0;
  if (!v4ok || (dtype != QAVIT_F32 && dtype != QAVIT_BF16)) {          // odd shapes: one launch each through the general entry point
    for (int i = 0; i < n; ++i) {
      const int rc = qavit_layernorm_bwd(dtype, dy[i], x[i], gamma[i], mean[i], rstd[i], dx[i], dgamma[i], dbeta[i], rows, C, nullptr, 0, nullptr, 0, nullptr,
                                         part_ws ? part_ws[i] : nullptr, stream);
      if (rc) return rc;
    }
    return QAVIT_OK;
  }
  LnBwd4 P;
  for (int i = 0; i < n; ++i) { P.dy[i] = dy[i]; P.x[i] = x[i]; P.gamma[i] = gamma[i]; P.mean[i] = mean[i]; P.rstd[i] = rstd[i]; P.dx[i] = dx[i]; P.dgamma[i] = dgamma[i]; P.dbeta[i] = dbeta[i]; P.parts[i] = part_ws ? part_ws[i] : nullptr; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int NW = 16;
  const int grid = qavit_layernorm_bwd_parts(rows, C);
  const int np = (C + 255) / 256;
#define LNVM(T_, NP_) hipLaunchKernelGGL((layernorm_bwd_v4_multi_kernel<T_, NP_, 4, NW>), dim3(grid, n), dim3(64 * NW), 0, st, P, rows, C)
  if (dtype == QAVIT_F32) { if (np == 1) LNVM(float, 1); else LNVM(float, 2); }
  else { if (np == 1) LNVM(bf16, 1); else LNVM(bf16, 2); }
#undef LNVM
  return check_launch("layernorm_bwd_multi");
}

extern "C" int qavit_ln_param_reduce(const qavit_ln_reduce_desc* d, int n, void* stream) {
  if (!d || n <= 0) return set_error(QAVIT_EINVAL, "ln_param_reduce: empty list");
  for (int i = 0; i < n; ++i) {
    const bool narrow = d[i].C == 1 || d[i].C == 2;        // rows of 4 floats
    if (!d[i].parts || d[i].nparts <= 0 || d[i].C <= 0 || (d[i].C % 4 && !narrow) || d[i].C > 2048 || (reinterpret_cast<uintptr_t>(d[i].parts) & 15) ||
        d[i].stride < 0 || d[i].stride % 4 || (d[i].stride > 0 && d[i].stride < (narrow ? 4 : 2 * d[i].C)))
      return set_error(QAVIT_EINVAL, "ln_param_reduce: bad descriptor (C % 4 == 0 or C in {1, 2}, C <= 2048, 16-byte aligned partial sums, stride % 4 == 0)");
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int done = 0; done < n; done += 48) {
    LnReduceGroup G;
    G.n = n - done < 48 ? n - done : 48;
    int maxp = 1;
    for (int i = 0; i < G.n; ++i) { G.d[i] = d[done + i]; if (d[done + i].nparts > maxp) maxp = d[done + i].nparts; }
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(G.n, (maxp + LNR_SLICE - 1) / LNR_SLICE), dim3(1024), 0, st, G);
  }
  return check_launch("ln_param_reduce");
}
