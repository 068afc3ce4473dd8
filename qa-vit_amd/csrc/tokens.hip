// Per-image token-axis kernels: TokenLearner mixing, TokenUpMix, MSDA landmark gather+pool.
// One 256-thread workgroup per image; matrices are tiny (N,M <= 256), so operands sit in LDS / L2 and the
// products run as 16x16 MFMA tiles (mma_lds.cuh) distributed over the 4 waves.
#include "common.cuh"
#include <type_traits>
#include "mma_lds.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "tokens_shared.h"

namespace qv {

// column-wise (over n) softmax statistics helper: 256 threads, M columns, parts = 256 / M threads per column.
// red must hold 256 floats.
__device__ __forceinline__ float col_reduce_max(float v, float* red, int M, int parts) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  const int m = t % M;
  float r = -INFINITY;
  for (int p = 0; p < parts; ++p) r = fmaxf(r, red[p * M + m]);
  return r;
}
__device__ __forceinline__ float col_reduce_sum(float v, float* red, int M, int parts) {
  const int t = threadIdx.x;
  __syncthreads();
  red[t] = v;
  __syncthreads();
  const int m = t % M;
  float r = 0.f;
  for (int p = 0; p < parts; ++p) r += red[p * M + m];
  return r;
}

// ------------------------------------------------------------------------------------------------
// TokenLearner: p = softmax_n(scores[b,:,m]); xc[b,m,:] = sum_n p[n,m] x[b,n,:]
// ------------------------------------------------------------------------------------------------
template <typename T, bool BF>
__global__ __launch_bounds__(256) void tokmix_fwd_kernel(const T* scores, const T* x, T* p_out, T* xc, int B, int N, int M, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* P = sm;                 // [N][M]
  float* red = sm + N * M;       // [256]
  const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6;
  const int parts = 256 / M;     // M divides 256 (checked on host)
  const int m = t % M, part = t / M;
  const T* sc = scores + (size_t)b * N * M;
  float mx = -INFINITY;
  for (int n = part; n < N; n += parts) { const float v = to_f<T>(sc[n * M + m]); P[n * M + m] = v; mx = fmaxf(mx, v); }
  mx = col_reduce_max(mx, red, M, parts);
  float s = 0.f;
  for (int n = part; n < N; n += parts) { const float e = __expf(P[n * M + m] - mx); P[n * M + m] = e; s += e; }
  s = col_reduce_sum(s, red, M, parts);
  const float inv = 1.f / s;
  for (int n = part; n < N; n += parts) {
    const float v = P[n * M + m] * inv;
    P[n * M + m] = v;
    p_out[(size_t)b * N * M + n * M + m] = from_f<T>(v);
  }
  __syncthreads();
  const T* xb = x + (size_t)b * N * C;
  T* ob = xc + (size_t)b * M * C;
  const int mt_n = (M + 15) / 16, ct_n = (C + 15) / 16;
  for (int tile = wave; tile < mt_n * ct_n; tile += 4) {
    const int mt = tile / ct_n, ct = tile - mt * ct_n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mma_tile<BF>(P + mt * 16, 1, M, M - mt * 16, xb + ct * 16, C, 1, C - ct * 16, N, acc);
    tile_to_global<T>(ob + (size_t)mt * 16 * C + ct * 16, C, M - mt * 16, C - ct * 16, acc);
  }
}

// dx[n,:] = sum_m p[n,m] dxc[m,:];  dP[n,m] = x[n,:].dxc[m,:];  dscores = p * (dP - sum_n p*dP)
template <typename T, bool BF>
__global__ __launch_bounds__(256) void tokmix_bwd_kernel(const T* p_in, const T* x, const T* dxc, T* dx, T* dscores, int B, int N, int M, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* P = sm;                  // [N][M]
  float* dP = sm + N * M;         // [N][M]
  float* red = sm + 2 * N * M;    // [256]
  const int b = blockIdx.x, t = threadIdx.x, wave = t >> 6;
  const T* pb = p_in + (size_t)b * N * M;
  const T* xb = x + (size_t)b * N * C;
  const T* gb = dxc + (size_t)b * M * C;
  for (int i = t; i < N * M; i += 256) P[i] = to_f<T>(pb[i]);
  __syncthreads();
  const int nt_n = (N + 15) / 16, ct_n = (C + 15) / 16, mt_n = (M + 15) / 16;
  T* dxb = dx + (size_t)b * N * C;
  for (int tile = wave; tile < nt_n * ct_n; tile += 4) {
    const int nt = tile / ct_n, ct = tile - nt * ct_n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mma_tile<BF>(P + nt * 16 * M, M, 1, N - nt * 16, gb + ct * 16, C, 1, C - ct * 16, M, acc);
    tile_to_global<T>(dxb + (size_t)nt * 16 * C + ct * 16, C, N - nt * 16, C - ct * 16, acc);
  }
  for (int tile = wave; tile < nt_n * mt_n; tile += 4) {
    const int nt = tile / mt_n, mt = tile - nt * mt_n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mma_tile<BF>(xb + (size_t)nt * 16 * C, C, 1, N - nt * 16, gb + (size_t)mt * 16 * C, 1, C, M - mt * 16, C, acc);
    tile_to_f32<false>(dP + nt * 16 * M + mt * 16, M, 1, N - nt * 16, M - mt * 16, acc);
  }
  __syncthreads();
  const int parts = 256 / M;
  const int m = t % M, part = t / M;
  float dot = 0.f;
  for (int n = part; n < N; n += parts) dot += P[n * M + m] * dP[n * M + m];
  dot = col_reduce_sum(dot, red, M, parts);
  for (int n = part; n < N; n += parts)
    dscores[(size_t)b * N * M + n * M + m] = from_f<T>(P[n * M + m] * (dP[n * M + m] - dot));
}

// ------------------------------------------------------------------------------------------------
// TokenUpMix: up[n,c] = sum_m W[n,m] xc[m,c] + bias[n]; y = LN_c(up)
// ------------------------------------------------------------------------------------------------
constexpr int UP_ROWS = 64;   // rows of `up` resident in LDS at a time

template <typename T, bool BF>
__device__ __forceinline__ void upmix_chunk(const float* W, const float* bias, const T* xcb, float* up, int n0, int rows, int N, int M, int C) {
  const int wave = threadIdx.x >> 6;
  const int rt_n = (rows + 15) / 16, ct_n = (C + 15) / 16;
  for (int tile = wave; tile < rt_n * ct_n; tile += 4) {
    const int rt = tile / ct_n, ct = tile - rt * ct_n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mma_tile<BF>(W + (size_t)(n0 + rt * 16) * M, M, 1, rows - rt * 16, xcb + ct * 16, C, 1, C - ct * 16, M, acc);
    const int col = tile_col();
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = rt * 16 + tile_row(reg);
      if (row < rows && ct * 16 + col < C) up[row * C + ct * 16 + col] = acc[reg] + bias[n0 + row];
    }
  }
}

template <typename T, bool BF>
__global__ __launch_bounds__(256) void upmix_fwd_kernel(const T* xc, const float* W, const float* bias, const float* gamma, const float* beta,
                                                        float eps, T* y, float* mean_o, float* rstd_o, int B, int N, int M, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* up = sm;                // [UP_ROWS][C]
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T* xcb = xc + (size_t)b * M * C;
  const float invC = 1.f / (float)C;
  for (int n0 = 0; n0 < N; n0 += UP_ROWS) {
    const int rows = (N - n0 < UP_ROWS) ? N - n0 : UP_ROWS;
    __syncthreads();
    upmix_chunk<T, BF>(W, bias, xcb, up, n0, rows, N, M, C);
    __syncthreads();
    for (int r = wave; r < rows; r += 4) {
      const float* ur = up + r * C;
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += ur[c];
      const float mean = wave_sum(s) * invC;
      float s2 = 0.f;
      for (int c = lane; c < C; c += 64) { const float d = ur[c] - mean; s2 += d * d; }
      const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
      T* yr = y + ((size_t)b * N + n0 + r) * C;
      for (int c = lane; c < C; c += 64) yr[c] = from_f<T>((ur[c] - mean) * rstd * gamma[c] + beta[c]);
      if (lane == 0) { mean_o[(size_t)b * N + n0 + r] = mean; rstd_o[(size_t)b * N + n0 + r] = rstd; }
    }
  }
}

// bwd: dup = LN'(dy); dxc[m,c] = sum_n W[n,m] dup[n,c]; dW[n,m] += sum_c dup[n,c] xc[m,c]; dbias[n] += sum_c dup[n,c]
template <typename T, bool BF>
__global__ __launch_bounds__(256) void upmix_bwd_kernel(const T* dy, const T* xc, const float* W, const float* bias, const float* gamma,
                                                        const float* mean, const float* rstd, T* dxc, float* dW, float* dbias,
                                                        float* dgamma, float* dbeta, int B, int N, int M, int C, int lds_dw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* up = sm;                          // [UP_ROWS][C]  (becomes dup in place)
  float* dxa = up + UP_ROWS * C;           // [M][C] accumulator for dxc of the current image
  float* pg = dxa + M * C;                 // [C] dgamma partial
  float* pb = pg + C;                      // [C] dbeta partial
  float* dba = pb + C;                     // [N] dbias partial
  float* dwa = dba + N;                    // [N][M] (only when lds_dw)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float invC = 1.f / (float)C;
  for (int i = t; i < 2 * C + N; i += 256) pg[i] = 0.f;
  if (lds_dw) for (int i = t; i < N * M; i += 256) dwa[i] = 0.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const T* xcb = xc + (size_t)b * M * C;
    __syncthreads();
    for (int i = t; i < M * C; i += 256) dxa[i] = 0.f;
    for (int n0 = 0; n0 < N; n0 += UP_ROWS) {
      const int rows = (N - n0 < UP_ROWS) ? N - n0 : UP_ROWS;
      __syncthreads();
      upmix_chunk<T, BF>(W, bias, xcb, up, n0, rows, N, M, C);
      __syncthreads();
      // LayerNorm backward per row, dup written in place
      for (int r = wave; r < rows; r += 4) {
        float* ur = up + r * C;
        const size_t row = (size_t)b * N + n0 + r;
        const float mu = mean[row], rs = rstd[row];
        const T* gr = dy + row * C;
        float c1 = 0.f, c2 = 0.f;
        for (int c = lane; c < C; c += 64) {
          const float d = to_f<T>(gr[c]);
          const float xh = (ur[c] - mu) * rs;
          const float g = d * gamma[c];
          c1 += g * xh; c2 += g;
        }
        c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
        float rowsum = 0.f;
        for (int c = lane; c < C; c += 64) {
          const float d = to_f<T>(gr[c]);
          const float xh = (ur[c] - mu) * rs;
          const float g = d * gamma[c];
          atomicAdd(pg + c, d * xh);       // LDS atomics: 4 waves share the [C] partials
          atomicAdd(pb + c, d);
          const float du = rs * (g - c2 - xh * c1);
          ur[c] = du;
          rowsum += du;
        }
        rowsum = wave_sum(rowsum);
        if (lane == 0) dba[n0 + r] += rowsum;   // row r is owned by this wave
      }
      __syncthreads();
      // dxc += W[n0:n0+rows]^T . dup
      {
        const int mt_n = (M + 15) / 16, ct_n = (C + 15) / 16;
        for (int tile = wave; tile < mt_n * ct_n; tile += 4) {
          const int mt = tile / ct_n, ct = tile - mt * ct_n;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mma_tile<BF>(W + (size_t)n0 * M + mt * 16, 1, M, M - mt * 16, up + ct * 16, C, 1, C - ct * 16, rows, acc);
          tile_to_f32<true>(dxa + mt * 16 * C + ct * 16, C, 1, M - mt * 16, C - ct * 16, acc);
        }
        // dW[n0+r, m] += dup[r,:] . xc[m,:]
        const int rt_n = (rows + 15) / 16;
        for (int tile = wave; tile < rt_n * mt_n; tile += 4) {
          const int rt = tile / mt_n, mt = tile - rt * mt_n;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mma_tile<BF>(up + rt * 16 * C, C, 1, rows - rt * 16, xcb + (size_t)mt * 16 * C, 1, C, M - mt * 16, C, acc);
          const int col = tile_col();
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = rt * 16 + tile_row(reg);
            if (row < rows && mt * 16 + col < M) {
              if (lds_dw) dwa[(n0 + row) * M + mt * 16 + col] += acc[reg];
              else atomic_add_f(dW + (size_t)(n0 + row) * M + mt * 16 + col, acc[reg]);
            }
          }
        }
      }
    }
    __syncthreads();
    T* ob = dxc + (size_t)b * M * C;
    for (int i = t; i < M * C; i += 256) ob[i] = from_f<T>(dxa[i]);
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) { atomic_add_f(dgamma + c, pg[c]); atomic_add_f(dbeta + c, pb[c]); }
  if (dbias) for (int n = t; n < N; n += 256) atomic_add_f(dbias + n, dba[n]);
  if (lds_dw) for (int i = t; i < N * M; i += 256) atomic_add_f(dW + i, dwa[i]);
}

// ------------------------------------------------------------------------------------------------
// MSDA landmark tokens: y[b,j,:] = mean_s x[b, idx[j*stride+s], :]
// ------------------------------------------------------------------------------------------------
// I = uint32_t whenever the element count allows: 64-bit div/mod per element costs more than the gather itself
template <typename T, typename I>
__global__ __launch_bounds__(256) void gather_pool_fwd_kernel(const T* x, const int32_t* idx, T* y, int B, int N, int NP, int stride, int C) {
  const I total = (I)B * NP * C;
  const float inv = 1.f / (float)stride;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const int c = (int)(i % (I)C);
    const int j = (int)((i / (I)C) % (I)NP);
    const int b = (int)(i / ((I)C * NP));
    float s = 0.f;
    for (int k = 0; k < stride; ++k) s += to_f<T>(x[((size_t)b * N + idx[j * stride + k]) * C + c]);
    y[i] = from_f<T>(s * inv);
  }
}
template <typename T, typename I>
__global__ __launch_bounds__(256) void gather_pool_bwd_kernel(const T* dy, const int32_t* idx, T* dx, int B, int N, int NP, int stride, int C) {
  const I total = (I)B * N * C;
  const float inv = 1.f / (float)stride;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const int c = (int)(i % (I)C);
    const int n = (int)((i / (I)C) % (I)N);
    const int b = (int)(i / ((I)C * N));
    float s = 0.f;
    for (int e = 0; e < NP * stride; ++e)
      if (idx[e] == n) s += to_f<T>(dy[((size_t)b * NP + e / stride) * C + c]);
    dx[i] = from_f<T>(s * inv);
  }
}

// Same result with the scatter turned into a gather once per workgroup: thread n scans the index list (staged in LDS) in
// order and keeps the (at most 4) pooled rows that read token n -- a token appears once per dilation -- so each output
// vector is the sum of <= 4 rows instead of a scan of all NP*stride entries per element.  Deterministic (list order =
// index order).  A token with more than 4 readers makes its workgroup take the scan for every element.
template <typename T>
__global__ __launch_bounds__(256) void gather_pool_bwd2_kernel(const T* dy, const int32_t* idx, T* dx, int B, int N, int NP, int stride, int C, int ldx) {
  constexpr int VEC = 16 / sizeof(T);
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) int ism[];
  int* sidx = ism;                 // [E]
  int* src = ism + NP * stride;    // [N][4]
  __shared__ int over;
  const int E = NP * stride;
  if (threadIdx.x == 0) over = 0;
  for (int e = threadIdx.x; e < E; e += 256) sidx[e] = idx[e];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {
    int k = 0, loc[4] = {-1, -1, -1, -1};
    for (int e = 0; e < E; ++e)
      if (sidx[e] == n) { if (k < 4) loc[k] = e / stride; ++k; }
#pragma unroll
    for (int q = 0; q < 4; ++q) src[n * 4 + q] = loc[q];
    if (k > 4) over = 1;
  }
  __syncthreads();
  const bool slow = over != 0;
  const float inv = 1.f / (float)stride;
  const int CV = C / VEC;
  const uint32_t total = (uint32_t)B * N * CV;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int cv = (int)(i % (uint32_t)CV);
    const int n = (int)((i / (uint32_t)CV) % (uint32_t)N);
    const int b = (int)(i / ((uint32_t)CV * N));
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (!slow) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = src[n * 4 + q];
        if (r >= 0) {
          const vec_t v = *reinterpret_cast<const vec_t*>(dy + ((size_t)b * NP + r) * C + cv * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += to_f<T>(v[j]);
        }
      }
    } else {
      for (int e = 0; e < E; ++e)
        if (sidx[e] == n) {
          const vec_t v = *reinterpret_cast<const vec_t*>(dy + ((size_t)b * NP + e / stride) * C + cv * VEC);
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[j] += to_f<T>(v[j]);
        }
    }
    vec_t o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = from_f<T>(acc[j] * inv);
    *reinterpret_cast<vec_t*>(dx + ((size_t)b * N + n) * ldx + cv * VEC) = o;
  }
}

}  // namespace qv

using namespace qv;

static bool pow2_le_256(int m) { return m > 0 && m <= 256 && (256 % m) == 0; }

extern "C" int qavit_tokmix_fwd(int dtype, const void* scores, const void* x, void* p, void* xc, int B, int N, int M, int C, void* stream) {
  if (!scores || !x || !p || !xc || B <= 0 || N <= 0 || C <= 0 || !pow2_le_256(M)) return set_error(QAVIT_EINVAL, "tokmix_fwd: bad arguments (M must divide 256)");
  const size_t smem = ((size_t)N * M + 256) * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "tokmix_fwd: N*M too large for LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_BF16) {
    const int took = qv::tokmix_bf16_try(false, scores, x, nullptr, p, xc, B, N, M, C, st);
    if (took < 0) return took;
    if (took == 1) return check_launch("tokmix_fwd(bf16)");
  }
  if (dtype == QAVIT_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokmix_fwd_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tokmix_fwd_kernel<float, false>), dim3(B), dim3(256), smem, st, (const float*)scores, (const float*)x, (float*)p, (float*)xc, B, N, M, C);
  } else if (dtype == QAVIT_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokmix_fwd_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tokmix_fwd_kernel<bf16, true>), dim3(B), dim3(256), smem, st, (const bf16*)scores, (const bf16*)x, (bf16*)p, (bf16*)xc, B, N, M, C);
  } else return set_error(QAVIT_EINVAL, "tokmix_fwd: unknown dtype");
  return check_launch("tokmix_fwd");
}

extern "C" int qavit_tokmix_bwd(int dtype, const void* p, const void* x, const void* dxc, void* dx, void* dscores, int B, int N, int M, int C, void* stream) {
  if (!p || !x || !dxc || !dx || !dscores || B <= 0 || N <= 0 || C <= 0 || !pow2_le_256(M)) return set_error(QAVIT_EINVAL, "tokmix_bwd: bad arguments (M must divide 256)");
  const size_t smem = ((size_t)2 * N * M + 256) * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "tokmix_bwd: N*M too large for LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_BF16) {
    const int took = qv::tokmix_bf16_try(true, p, x, dxc, dx, dscores, B, N, M, C, st);
    if (took < 0) return took;
    if (took == 1) return check_launch("tokmix_bwd(bf16)");
  }
  if (dtype == QAVIT_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokmix_bwd_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tokmix_bwd_kernel<float, false>), dim3(B), dim3(256), smem, st, (const float*)p, (const float*)x, (const float*)dxc, (float*)dx, (float*)dscores, B, N, M, C);
  } else if (dtype == QAVIT_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokmix_bwd_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tokmix_bwd_kernel<bf16, true>), dim3(B), dim3(256), smem, st, (const bf16*)p, (const bf16*)x, (const bf16*)dxc, (bf16*)dx, (bf16*)dscores, B, N, M, C);
  } else return set_error(QAVIT_EINVAL, "tokmix_bwd: unknown dtype");
  return check_launch("tokmix_bwd");
}

extern "C" int qavit_upmix_fwd(int dtype, const void* xc, const float* W, const float* bias, const float* gamma, const float* beta,
                               float eps, void* y, float* mean, float* rstd, int B, int N, int M, int C, void* stream) {
  if (!xc || !W || !bias || !gamma || !beta || !y || !mean || !rstd || B <= 0 || N <= 0 || M <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "upmix_fwd: bad arguments");
  const size_t smem = (size_t)UP_ROWS * C * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "upmix_fwd: C too large for LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_BF16) {
    const int took = qv::upmix_bf16_try(false, nullptr, xc, W, bias, gamma, beta, eps, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, B, N, M, C, st);
    if (took < 0) return took;
    if (took == 1) return check_launch("upmix_fwd(bf16)");
  }
  if (dtype == QAVIT_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upmix_fwd_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((upmix_fwd_kernel<float, false>), dim3(B), dim3(256), smem, st, (const float*)xc, W, bias, gamma, beta, eps, (float*)y, mean, rstd, B, N, M, C);
  } else if (dtype == QAVIT_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upmix_fwd_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((upmix_fwd_kernel<bf16, true>), dim3(B), dim3(256), smem, st, (const bf16*)xc, W, bias, gamma, beta, eps, (bf16*)y, mean, rstd, B, N, M, C);
  } else return set_error(QAVIT_EINVAL, "upmix_fwd: unknown dtype");
  return check_launch("upmix_fwd");
}

extern "C" int qavit_upmix_bwd_parts(int dtype, int B, int N, int M, int C) {
  return (dtype == QAVIT_BF16 && B > 0) ? qv::upmix_bf16_parts(B, N, M, C) : 0;
}

extern "C" int qavit_upmix_bwd(int dtype, const void* dy, const void* xc, const float* W, const float* bias, const float* gamma,
                               const float* mean, const float* rstd, void* dxc, float* dW, float* dbias, float* dgamma, float* dbeta,
                               int B, int N, int M, int C, void* stream) {
  return qavit_upmix_bwd_p(dtype, dy, xc, W, bias, gamma, mean, rstd, dxc, dW, dbias, dgamma, dbeta, B, N, M, C, nullptr, stream);
}

extern "C" int qavit_upmix_fwd_sa_supported(int dtype, int N, int M, int C) {
  return dtype == QAVIT_BF16 && C == 192 && ((N == 64 && M == 16) || (N == 256 && M == 64));
}

extern "C" int qavit_upmix_fwd_sa(int dtype, const void* x, const void* u, const float* sa_gamma, float dp_p, int dp_site, const int64_t* rng, void* xc,
                                  const float* W, const float* bias, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                                  int B, int N, int M, int C, void* stream) {
  if (!qavit_upmix_fwd_sa_supported(dtype, N, M, C)) return set_error(QAVIT_EINVAL, "upmix_fwd_sa: bf16, 64 -> 16 or 256 -> 64 tokens, C = 192 only");
  if (!x || !u || !xc || !W || !bias || !gamma || !beta || !y || !mean || !rstd || B <= 0) return set_error(QAVIT_EINVAL, "upmix_fwd_sa: bad arguments");
  if (dp_p < 0.f || dp_p >= 1.f || (dp_p > 0.f && !rng)) return set_error(QAVIT_EINVAL, "upmix_fwd_sa: drop-path needs rng");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(xc) | reinterpret_cast<uintptr_t>(y)) & 7)
    return set_error(QAVIT_EINVAL, "upmix_fwd_sa: operand alignment");
  const int took = qv::upmix_bf16_try(false, nullptr, x, W, bias, gamma, beta, eps, y, mean, rstd, nullptr, nullptr, nullptr, nullptr, B, N, M, C,
                                      reinterpret_cast<hipStream_t>(stream), nullptr, u, xc, sa_gamma, nullptr, dp_p, dp_site, rng);
  if (took < 0) return took;
  if (took != 1) return set_error(QAVIT_EINVAL, "upmix_fwd_sa: shape not covered");
  return check_launch("upmix_fwd_sa");
}

extern "C" int qavit_upmix_bwd_sa_supported(int dtype, int N, int M, int C) { return dtype == QAVIT_BF16 && N == 64 && M == 16 && C == 192; }

extern "C" int qavit_upmix_bwd_sa(int dtype, const void* dy, const void* xc, const float* W, const float* bias, const float* gamma,
                                  const float* mean, const float* rstd, void* dxc, float* dW, float* dbias, float* dgamma, float* dbeta,
                                  int B, int N, int M, int C, float* parts, const void* u, void* du, const float* sa_gamma, float* sa_dgamma,
                                  float dp_p, int dp_site, const int64_t* rng, void* stream) {
  if (!qavit_upmix_bwd_sa_supported(dtype, N, M, C)) return set_error(QAVIT_EINVAL, "upmix_bwd_sa: bf16, 64 -> 16 tokens, C = 192 only");
  if (!dy || !xc || !W || !bias || !gamma || !mean || !rstd || !dxc || !dW || !dgamma || !dbeta || !u || !du || B <= 0)
    return set_error(QAVIT_EINVAL, "upmix_bwd_sa: bad arguments");
  if (dp_p < 0.f || dp_p >= 1.f || (dp_p > 0.f && !rng)) return set_error(QAVIT_EINVAL, "upmix_bwd_sa: drop-path needs rng");
  if (((reinterpret_cast<uintptr_t>(xc) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(dxc)) & 7) ||
      (parts && (reinterpret_cast<uintptr_t>(parts) & 15)))
    return set_error(QAVIT_EINVAL, "upmix_bwd_sa: operand alignment");
  const int took = qv::upmix_bf16_try(true, dy, xc, W, bias, gamma, nullptr, 0.f, dxc, const_cast<float*>(mean), const_cast<float*>(rstd), dW, dbias, dgamma, dbeta,
                                      B, N, M, C, reinterpret_cast<hipStream_t>(stream), parts, u, du, sa_gamma, sa_dgamma, dp_p, dp_site, rng);
  if (took < 0) return took;
  if (took != 1) return set_error(QAVIT_EINVAL, "upmix_bwd_sa: shape not covered");
  return check_launch("upmix_bwd_sa");
}

extern "C" int qavit_upmix_bwd_p(int dtype, const void* dy, const void* xc, const float* W, const float* bias, const float* gamma,
                                 const float* mean, const float* rstd, void* dxc, float* dW, float* dbias, float* dgamma, float* dbeta,
                                 int B, int N, int M, int C, float* parts, void* stream) {
  if (parts && (qavit_upmix_bwd_parts(dtype, B, N, M, C) == 0 || (reinterpret_cast<uintptr_t>(parts) & 15) || (reinterpret_cast<uintptr_t>(xc) & 7)))
    return set_error(QAVIT_EINVAL, "upmix_bwd: partial rows asked for a shape / dtype / alignment without that path (qavit_upmix_bwd_parts() == 0)");
  if (!dy || !xc || !W || !bias || !gamma || !mean || !rstd || !dxc || !dW || !dgamma || !dbeta || B <= 0 || N <= 0 || M <= 0 || C <= 0)
    return set_error(QAVIT_EINVAL, "upmix_bwd: bad arguments");
  const int lds_dw = ((size_t)N * M <= 4096) ? 1 : 0;
  const size_t smem = ((size_t)UP_ROWS * C + (size_t)M * C + 2 * C + N + (lds_dw ? (size_t)N * M : 0)) * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "upmix_bwd: problem too large for LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int grid = B < 512 ? B : 512;
  if (dtype == QAVIT_BF16) {
    const int took = qv::upmix_bf16_try(true, dy, xc, W, bias, gamma, nullptr, 0.f, dxc, const_cast<float*>(mean), const_cast<float*>(rstd), dW, dbias, dgamma, dbeta, B, N, M, C, st, parts);
    if (took < 0) return took;
    if (took == 1) return check_launch("upmix_bwd(bf16)");
    if (parts) return set_error(QAVIT_EINVAL, "upmix_bwd: partial rows not available");
  }
  if (dtype == QAVIT_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upmix_bwd_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((upmix_bwd_kernel<float, false>), dim3(grid), dim3(256), smem, st, (const float*)dy, (const float*)xc, W, bias, gamma, mean, rstd, (float*)dxc, dW, dbias, dgamma, dbeta, B, N, M, C, lds_dw);
  } else if (dtype == QAVIT_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upmix_bwd_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((upmix_bwd_kernel<bf16, true>), dim3(grid), dim3(256), smem, st, (const bf16*)dy, (const bf16*)xc, W, bias, gamma, mean, rstd, (bf16*)dxc, dW, dbias, dgamma, dbeta, B, N, M, C, lds_dw);
  } else return set_error(QAVIT_EINVAL, "upmix_bwd: unknown dtype");
  return check_launch("upmix_bwd");
}

extern "C" int qavit_gather_pool_fwd(int dtype, const void* x, const int32_t* idx, void* y, int B, int N, int NP, int stride, int C, void* stream) {
  if (!x || !idx || !y || B <= 0 || N <= 0 || NP <= 0 || stride <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "gather_pool_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t total = (int64_t)B * NP * C;
  int grid = (int)((total + 1023) / 1024); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
  const bool small = (int64_t)B * (NP > N ? NP : N) * C < 0x7fffffffLL;
  if (dtype == QAVIT_F32) {
    if (small) hipLaunchKernelGGL((gather_pool_fwd_kernel<float, uint32_t>), dim3(grid), dim3(256), 0, st, (const float*)x, idx, (float*)y, B, N, NP, stride, C);
    else hipLaunchKernelGGL((gather_pool_fwd_kernel<float, int64_t>), dim3(grid), dim3(256), 0, st, (const float*)x, idx, (float*)y, B, N, NP, stride, C);
  } else if (dtype == QAVIT_BF16) {
    if (small) hipLaunchKernelGGL((gather_pool_fwd_kernel<bf16, uint32_t>), dim3(grid), dim3(256), 0, st, (const bf16*)x, idx, (bf16*)y, B, N, NP, stride, C);
    else hipLaunchKernelGGL((gather_pool_fwd_kernel<bf16, int64_t>), dim3(grid), dim3(256), 0, st, (const bf16*)x, idx, (bf16*)y, B, N, NP, stride, C);
  }
  else return set_error(QAVIT_EINVAL, "gather_pool_fwd: unknown dtype");
  return check_launch("gather_pool_fwd");
}

extern "C" int qavit_gather_pool_bwd(int dtype, const void* dy, const int32_t* idx, void* dx, int B, int N, int NP, int stride, int C, void* stream) {
  return qavit_gather_pool_bwd_ld(dtype, dy, idx, dx, C, B, N, NP, stride, C, stream);
}

extern "C" int qavit_gather_pool_bwd_ld(int dtype, const void* dy, const int32_t* idx, void* dx, int ldx, int B, int N, int NP, int stride, int C, void* stream) {
  if (!dy || !idx || !dx || B <= 0 || N <= 0 || NP <= 0 || stride <= 0 || C <= 0 || ldx < C) return set_error(QAVIT_EINVAL, "gather_pool_bwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t total = (int64_t)B * N * C;
  int grid = (int)((total + 1023) / 1024); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
  const bool small = (int64_t)B * (NP > N ? NP : N) * C < 0x7fffffffLL;
  {
    const int vec = dtype == QAVIT_BF16 ? 8 : 4;
    const size_t lds = ((size_t)NP * stride + (size_t)N * 4) * sizeof(int);
    const bool al = !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15);
    if (small && al && C % vec == 0 && ldx % vec == 0 && lds <= 96 * 1024 && (dtype == QAVIT_BF16 || dtype == QAVIT_F32)) {
      int64_t nv = (int64_t)B * N * (C / vec);
      int g2 = (int)((nv + 1023) / 1024); if (g2 > 512) g2 = 512; if (g2 < 1) g2 = 1;
      if (dtype == QAVIT_BF16) hipLaunchKernelGGL((gather_pool_bwd2_kernel<bf16>), dim3(g2), dim3(256), lds, st, (const bf16*)dy, idx, (bf16*)dx, B, N, NP, stride, C, ldx);
      else hipLaunchKernelGGL((gather_pool_bwd2_kernel<float>), dim3(g2), dim3(256), lds, st, (const float*)dy, idx, (float*)dx, B, N, NP, stride, C, ldx);
      return check_launch("gather_pool_bwd");
    }
  }
  if (ldx != C) return set_error(QAVIT_EINVAL, "gather_pool_bwd: a row stride needs the vector kernel (16-byte aligned rows, C and ldx multiples of the vector)");
  if (dtype == QAVIT_F32) {
    if (small) hipLaunchKernelGGL((gather_pool_bwd_kernel<float, uint32_t>), dim3(grid), dim3(256), 0, st, (const float*)dy, idx, (float*)dx, B, N, NP, stride, C);
    else hipLaunchKernelGGL((gather_pool_bwd_kernel<float, int64_t>), dim3(grid), dim3(256), 0, st, (const float*)dy, idx, (float*)dx, B, N, NP, stride, C);
  } else if (dtype == QAVIT_BF16) {
    if (small) hipLaunchKernelGGL((gather_pool_bwd_kernel<bf16, uint32_t>), dim3(grid), dim3(256), 0, st, (const bf16*)dy, idx, (bf16*)dx, B, N, NP, stride, C);
    else hipLaunchKernelGGL((gather_pool_bwd_kernel<bf16, int64_t>), dim3(grid), dim3(256), 0, st, (const bf16*)dy, idx, (bf16*)dx, B, N, NP, stride, C);
  }
  else return set_error(QAVIT_EINVAL, "gather_pool_bwd: unknown dtype");
  return check_launch("gather_pool_bwd");
}
