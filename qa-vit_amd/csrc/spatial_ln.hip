// nn.LayerNorm([C, H, W]) of the ConvNeXt-Tiny style stem (HQAViTv2_CIFAR100.py:766, :777, :791) on channel-last tokens.
// One "row" is a whole sample: E = H*W*C elements (4096 .. 16384), normalised together, with an element-wise affine
// whose parameters keep the reference's [C][H*W] layout (token (n, c) reads weight[c*N + n]: a 16..64 KB table that
// stays in L2).  HBM-bound: x is read once (forward) / x and dy once (backward) and held in registers.
//   forward : one workgroup per sample; sum -> mean, centred sum of squares -> rstd (both from registers).
//   backward: <= 256 workgroups, each looping over samples; a thread always owns the same element positions, so the
//             parameter gradients are registers that leave as one fp32 atomic each at the end.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"

namespace qv {

namespace {

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// element e of a thread: chunk k (of EPT/4), vector slot j:  e = (k*256 + tid)*4 + j
template <typename T, int EPT>
__device__ __forceinline__ void load_row(const T* p, float (&v)[EPT]) {
#pragma unroll
  for (int k = 0; k < EPT / 4; ++k) {
    const int e = (k * 256 + threadIdx.x) * 4;
    if constexpr (sizeof(T) == 4) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(p + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * k + j] = q[j];
    } else {
      const bf16x4 q = *reinterpret_cast<const bf16x4*>(p + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * k + j] = (float)q[j];
    }
  }
}
template <typename T, int EPT>
__device__ __forceinline__ void store_row(T* p, const float (&v)[EPT]) {
#pragma unroll
  for (int k = 0; k < EPT / 4; ++k) {
    const int e = (k * 256 + threadIdx.x) * 4;
    if constexpr (sizeof(T) == 4) {
      f32x4 q;
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = v[4 * k + j];
      *reinterpret_cast<f32x4*>(p + e) = q;
    } else {
      bf16x4 q;
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = (bf16)v[4 * k + j];
      *reinterpret_cast<bf16x4*>(p + e) = q;
    }
  }
}
// parameter index of element e = n*C + c  ->  c*N + n
__device__ __forceinline__ int pidx(int e, int N, int C) { const int n = e / C; return (e - n * C) * N + n; }

template <typename T, int EPT>
__global__ __launch_bounds__(256) void sln_fwd_kernel(const T* x, const float* w, const float* b, T* y, float* mean_o, float* rstd_o,
                                                      int B, int N, int C, float eps) {
  __shared__ float red[4];
  const int E = N * C;
  const float invE = 1.f / (float)E;
  float wv[EPT], bv[EPT];
#pragma unroll
  for (int k = 0; k < EPT / 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pi = pidx((k * 256 + threadIdx.x) * 4 + j, N, C);
      wv[4 * k + j] = w[pi]; bv[4 * k + j] = b[pi];
    }
  for (int s = blockIdx.x; s < B; s += gridDim.x) {
    float v[EPT];
    load_row<T, EPT>(x + (size_t)s * E, v);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) a += v[i];
    const float mean = block_sum256(a, red) * invE;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(block_sum256(q, red) * invE + eps);
#pragma unroll
    for (int i = 0; i < EPT; ++i) v[i] = (v[i] - mean) * rstd * wv[i] + bv[i];
    store_row<T, EPT>(y + (size_t)s * E, v);
    if (threadIdx.x == 0) { mean_o[s] = mean; rstd_o[s] = rstd; }
  }
}

template <typename T, int EPT>
__global__ __launch_bounds__(256) void sln_bwd_kernel(const T* dy, const T* x, const float* w, const float* mean, const float* rstd, T* dx,
                                                      float* dw, float* db, int B, int N, int C) {
  __shared__ float red[4];
  const int E = N * C;
  const float invE = 1.f / (float)E;
  float wv[EPT], gw[EPT], gb[EPT];
#pragma unroll
  for (int k = 0; k < EPT / 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wv[4 * k + j] = w[pidx((k * 256 + threadIdx.x) * 4 + j, N, C)];
      gw[4 * k + j] = 0.f; gb[4 * k + j] = 0.f;
    }
  for (int s = blockIdx.x; s < B; s += gridDim.x) {
    float xv[EPT], g[EPT];
    load_row<T, EPT>(x + (size_t)s * E, xv);
    load_row<T, EPT>(dy + (size_t)s * E, g);
    const float mu = mean[s], rs = rstd[s];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const float xh = (xv[i] - mu) * rs;
      gw[i] += g[i] * xh; gb[i] += g[i];
      xv[i] = xh;
      g[i] *= wv[i];
      c1 += g[i] * xh; c2 += g[i];
    }
    c1 = block_sum256(c1, red) * invE;
    c2 = block_sum256(c2, red) * invE;
#pragma unroll
    for (int i = 0; i < EPT; ++i) g[i] = rs * (g[i] - c2 - xv[i] * c1);
    store_row<T, EPT>(dx + (size_t)s * E, g);
  }
#pragma unroll
  for (int k = 0; k < EPT / 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pi = pidx((k * 256 + threadIdx.x) * 4 + j, N, C);
      atomic_add_f(dw + pi, gw[4 * k + j]);
      atomic_add_f(db + pi, gb[4 * k + j]);
    }
}

// ---- ConvNeXt layer scale: y = x + droppath(gamma[c] * u)  (HQAViTv2_CIFAR100.py:744-748) on rows [M, C] ----
// A thread owns VEC consecutive channels (one 16-byte access) and always the same ones, so dgamma partials are registers.
template <typename T> struct LV;
template <> struct LV<bf16> { static constexpr int N = 8; typedef bf16x8 type; };
template <> struct LV<float> { static constexpr int N = 4; typedef f32x4 type; };

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void chan_scale_kernel(const T* a0, const T* u, const float* gamma, T* out, float* dgamma, int M, int C,
                                                         float dp_p, int dp_site, int dp_rows, const int64_t* rng) {
  constexpr int VEC = LV<T>::N;
  typedef typename LV<T>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float sred[];     // [rpp][C]  (backward only)
  const int tpr = C / VEC, rpp = 256 / tpr;
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  const uint32_t key = dp_p > 0.f ? rng_key(rng, dp_site) : 0u;
  const float inv = dp_p > 0.f ? 1.f / (1.f - dp_p) : 1.f;
  float gm[VEC], part[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { gm[j] = gamma[tc * VEC + j]; part[j] = 0.f; }
  for (int r = blockIdx.x * rpp + tr; r < M; r += gridDim.x * rpp) {
    const float f = dp_p > 0.f ? drop_factor(key, (uint32_t)(r / dp_rows), dp_p, inv) : 1.f;
    const size_t o = (size_t)r * C + tc * VEC;
    const vec_t av = *reinterpret_cast<const vec_t*>(a0 + o);
    const vec_t uv = *reinterpret_cast<const vec_t*>(u + o);
    vec_t ov;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (BWD) {                       // a0 = dy: du = dy * f * gamma,  dgamma += dy * f * u
        const float g = (float)av[j] * f;
        ov[j] = (T)(g * gm[j]);
        part[j] += g * (float)uv[j];
      } else {                         // a0 = x
        ov[j] = (T)((float)av[j] + f * gm[j] * (float)uv[j]);
      }
    }
    *reinterpret_cast<vec_t*>(out + o) = ov;
  }
  if (BWD) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) sred[tr * C + tc * VEC + j] = part[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float s_ = 0.f;
      for (int q = 0; q < rpp; ++q) s_ += sred[q * C + c];
      atomic_add_f(dgamma + c, s_);
    }
  }
}

template <typename T>
bool chan_ok(const void* a, const void* b, const void* c, int C) {
  constexpr int VEC = LV<T>::N;
  if (C % VEC || C / VEC > 256 || 256 % (C / VEC)) return false;
  return !((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15);
}

template <typename T>
int chan_launch(bool bwd, const void* a0, const void* u, const float* gamma, void* out, float* dgamma, int M, int C,
                float dp_p, int dp_site, int dp_rows, const int64_t* rng, hipStream_t st) {
  if (!chan_ok<T>(a0, u, out, C)) return set_error(QAVIT_EINVAL, "chan_scale_add: C must tile 16-byte vectors with 256 % (C/vec) == 0, 16-byte aligned rows");
  const int rpp = 256 / (C / LV<T>::N);
  int grid = (M + rpp - 1) / rpp;
  const int cap = bwd ? 256 : 2048;
  if (grid > cap) grid = cap;
  if (!bwd) hipLaunchKernelGGL((chan_scale_kernel<T, false>), dim3(grid), dim3(256), 0, st, (const T*)a0, (const T*)u, gamma, (T*)out, (float*)nullptr, M, C, dp_p, dp_site, dp_rows, rng);
  else hipLaunchKernelGGL((chan_scale_kernel<T, true>), dim3(grid), dim3(256), (size_t)rpp * C * sizeof(float), st, (const T*)a0, (const T*)u, gamma, (T*)out, dgamma, M, C, dp_p, dp_site, dp_rows, rng);
  return check_launch(bwd ? "chan_scale_add_bwd" : "chan_scale_add_fwd");
}

template <typename T>
int sln_launch(bool bwd, const void* a0, const void* x, const float* w, const float* b, void* o, float* mean, float* rstd, float* dw, float* db,
               int B, int N, int C, float eps, hipStream_t st) {
  const int ept = N * C / 256;
  const int gf = B < 2048 ? B : 2048, gb = B < 256 ? B : 256;
#define SLN(EPT_)                                                                                                                      \
  if (!bwd) hipLaunchKernelGGL((sln_fwd_kernel<T, EPT_>), dim3(gf), dim3(256), 0, st, (const T*)x, w, b, (T*)o, mean, rstd, B, N, C, eps); \
  else hipLaunchKernelGGL((sln_bwd_kernel<T, EPT_>), dim3(gb), dim3(256), 0, st, (const T*)a0, (const T*)x, w, mean, rstd, (T*)o, dw, db, B, N, C);
  if (ept == 16) { SLN(16) } else if (ept == 32) { SLN(32) } else { SLN(64) }
#undef SLN
  return check_launch(bwd ? "spatial_ln_bwd" : "spatial_ln_fwd");
}

bool sln_ok(int N, int C) {
  const int E = N * C;
  return N > 0 && C > 0 && C % 4 == 0 && (E == 4096 || E == 8192 || E == 16384);
}

}  // namespace
}  // namespace qv

using namespace qv;

extern "C" int qavit_spatial_ln_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                    int B, int N, int C, float eps, void* stream) {
  if (!x || !w || !b || !y || !mean || !rstd || B <= 0) return set_error(QAVIT_EINVAL, "spatial_ln_fwd: bad arguments");
  if (!sln_ok(N, C)) return set_error(QAVIT_EINVAL, "spatial_ln_fwd: N*C must be 4096, 8192 or 16384 with C % 4 == 0");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return set_error(QAVIT_EINVAL, "spatial_ln_fwd: 16-byte aligned rows");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32) return sln_launch<float>(false, nullptr, x, w, b, y, mean, rstd, nullptr, nullptr, B, N, C, eps, st);
  if (dtype == QAVIT_BF16) return sln_launch<bf16>(false, nullptr, x, w, b, y, mean, rstd, nullptr, nullptr, B, N, C, eps, st);
  return set_error(QAVIT_EINVAL, "spatial_ln_fwd: unknown dtype");
}

extern "C" int qavit_spatial_ln_bwd(int dtype, const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                                    void* dx, float* dw, float* db, int B, int N, int C, void* stream) {
  if (!dy || !x || !w || !mean || !rstd || !dx || !dw || !db || B <= 0) return set_error(QAVIT_EINVAL, "spatial_ln_bwd: bad arguments");
  if (!sln_ok(N, C)) return set_error(QAVIT_EINVAL, "spatial_ln_bwd: N*C must be 4096, 8192 or 16384 with C % 4 == 0");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15)
    return set_error(QAVIT_EINVAL, "spatial_ln_bwd: 16-byte aligned rows");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32) return sln_launch<float>(true, dy, x, w, nullptr, dx, const_cast<float*>(mean), const_cast<float*>(rstd), dw, db, B, N, C, 0.f, st);
  if (dtype == QAVIT_BF16) return sln_launch<bf16>(true, dy, x, w, nullptr, dx, const_cast<float*>(mean), const_cast<float*>(rstd), dw, db, B, N, C, 0.f, st);
  return set_error(QAVIT_EINVAL, "spatial_ln_bwd: unknown dtype");
}

extern "C" int qavit_chan_scale_add_fwd(int dtype, const void* x, const void* u, const float* gamma, void* y, int rows, int C,
                                        float dp_p, int dp_site, int dp_rows, const int64_t* rng, void* stream) {
  if (!x || !u || !gamma || !y || rows <= 0 || C <= 0 || dp_rows <= 0 || (dp_p > 0.f && !rng) || dp_p >= 1.f)
    return set_error(QAVIT_EINVAL, "chan_scale_add_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32) return chan_launch<float>(false, x, u, gamma, y, nullptr, rows, C, dp_p, dp_site, dp_rows, rng, st);
  if (dtype == QAVIT_BF16) return chan_launch<bf16>(false, x, u, gamma, y, nullptr, rows, C, dp_p, dp_site, dp_rows, rng, st);
  return set_error(QAVIT_EINVAL, "chan_scale_add_fwd: unknown dtype");
}

extern "C" int qavit_chan_scale_add_bwd(int dtype, const void* dy, const void* u, const float* gamma, void* du, float* dgamma,
                                        int rows, int C, float dp_p, int dp_site, int dp_rows, const int64_t* rng, void* stream) {
  if (!dy || !u || !gamma || !du || !dgamma || rows <= 0 || C <= 0 || dp_rows <= 0 || (dp_p > 0.f && !rng) || dp_p >= 1.f)
    return set_error(QAVIT_EINVAL, "chan_scale_add_bwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_F32) return chan_launch<float>(true, dy, u, gamma, du, dgamma, rows, C, dp_p, dp_site, dp_rows, rng, st);
  if (dtype == QAVIT_BF16) return chan_launch<bf16>(true, dy, u, gamma, du, dgamma, rows, C, dp_p, dp_site, dp_rows, rng, st);
  return set_error(QAVIT_EINVAL, "chan_scale_add_bwd: unknown dtype");
}
