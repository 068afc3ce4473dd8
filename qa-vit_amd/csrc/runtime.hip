// Error state, version, and the small streaming kernels (weight packing, rng, dropout, token mean,
// hybrid fuse, scale-add, patchify, optimizer).
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include <stdio.h>
#include <string.h>

namespace qv {

static thread_local char g_err[256] = "";

int set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return QAVIT_ELAUNCH;
  }
  return QAVIT_OK;
}

__global__ void zero_f32_kernel(float* p, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
void zero_f32(float* p, size_t n, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
}

static inline int blocks_for(int64_t n, int per_block, int cap = 4096) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------
// weight packing: dst = cast(src), dstT = cast(src)^T.  One workgroup column per 32x32 tile.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const qavit_pack_desc* descs, int tiles_per_desc) {
  const qavit_pack_desc d = descs[blockIdx.y];
  __shared__ float tile[32][33];
  const int tcols = (d.cols + 31) / 32, trows = (d.rows + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int t = blockIdx.x; t < trows * tcols; t += tiles_per_desc) {
    const int tr = t / tcols, tc = t - tr * tcols;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
      const int r = tr * 32 + ty + i, c = tc * 32 + tx;
      const float v = (r < d.rows && c < d.cols) ? d.src[(size_t)r * d.cols + c] : 0.f;
      tile[ty + i][tx] = v;
      if (d.dst && r < d.rows && c < d.cols) {
        size_t o = (size_t)r * d.cols + c;
        if (d.pad == 1)                                    // MFMA fragment order (include/qavit.h, qavit_pack_desc)
          o = ((size_t)((r >> 4) * (d.cols >> 5) + (c >> 5)) * 64 + (size_t)(((c & 31) >> 3) * 16 + (r & 15))) * 8 + (c & 7);
        else if (d.pad == 3)                               // fragment order of the transpose: row of src^T = c, k index = r
          o = ((size_t)((c >> 4) * (d.rows >> 5) + (r >> 5)) * 64 + (size_t)(((r & 31) >> 3) * 16 + (c & 15))) * 8 + (r & 7);
        reinterpret_cast<T*>(d.dst)[o] = from_f<T>(v);
      }
    }
    __syncthreads();
    if (d.dstT) {
#pragma unroll
      for (int i = 0; i < 32; i += 8) {
        const int c = tc * 32 + ty + i, r = tr * 32 + tx;   // output row = c, output col = r
        if (r < d.rows && c < d.cols) reinterpret_cast<T*>(d.dstT)[(size_t)c * d.ldT + r] = from_f<T>(tile[tx][ty + i]);
      }
    }
  }
}

__global__ void rng_advance_kernel(int64_t* rng) { rng[1] += 1; }

// one 100 MHz wall-clock reading, written when the stream reaches this point (tools/chain_stamps.py: where the two chains of a step are, unprofiled)
__global__ void stamp_kernel(uint64_t* dst) { *dst = wall_clock64(); }

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* x, T* y, int64_t n, float p, int site, const int64_t* rng) {
  const uint32_t key = rng_key(rng, site);
  const float inv = 1.f / (1.f - p);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = from_f<T>(to_f<T>(x[i]) * drop_factor(key, (uint32_t)i, p, inv));
}

// y[b,c] = mean_n x[b,n,c]
template <typename T>
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const T* x, T* y, int B, int N, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += to_f<T>(x[((size_t)b * N + n) * C + c]);
    y[(size_t)b * C + c] = from_f<T>(s / (float)N);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const T* dy, T* dx, int B, int N, int C) {
  const int b = blockIdx.x;
  const float inv = 1.f / (float)N;
  for (int i = threadIdx.x; i < N * C; i += blockDim.x) {
    const int c = i % C;
    dx[(size_t)b * N * C + i] = from_f<T>(to_f<T>(dy[(size_t)b * C + c]) * inv);
  }
}

// HybridFusion: y = x * softmax(fw)[col / Cb]
__device__ __forceinline__ void softmax_small(const float* fw, int nb, float* w) {
  float mx = -INFINITY;
  for (int i = 0; i < nb; ++i) mx = fmaxf(mx, fw[i]);
  float s = 0.f;
  for (int i = 0; i < nb; ++i) { w[i] = __expf(fw[i] - mx); s += w[i]; }
  for (int i = 0; i < nb; ++i) w[i] /= s;
}
template <typename T>
__global__ __launch_bounds__(256) void hybrid_fwd_kernel(const T* x, const float* fw, T* y, int64_t n, int nb, int Cb) {
  float w[8];
  softmax_small(fw, nb, w);
  const int C = nb * Cb;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = from_f<T>(to_f<T>(x[i]) * w[(int)(i % C) / Cb]);
}
// dx = dy * w[b];  ds[b] = sum dy*x over branch b;  dfw[j] += sum_b ds[b] * w[b] * (delta_bj - w[j])
template <typename T>
__global__ __launch_bounds__(256) void hybrid_bwd_kernel(const T* dy, const T* x, const float* fw, T* dx, float* dfw,
                                                         int64_t n, int nb, int Cb, float* parts) {
  float w[8], part[8];
  softmax_small(fw, nb, w);
  for (int i = 0; i < 8; ++i) part[i] = 0.f;
  const int C = nb * Cb;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % C) / Cb;
    const float g = to_f<T>(dy[i]);
    dx[i] = from_f<T>(g * w[b]);
    const float t = g * to_f<T>(x[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) part[j] += (j == b) ? t : 0.f;
  }
  __shared__ float red[8][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = wave_sum(part[j]);
    if (lane == 0) red[j][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ds[8];
    for (int j = 0; j < nb; ++j) ds[j] = red[j][0] + red[j][1] + red[j][2] + red[j][3];
    float dot = 0.f;
    for (int j = 0; j < nb; ++j) dot += ds[j] * w[j];
    if (parts) {                                            // this workgroup's row of [grid][8] contributions (qavit_ln_param_reduce, C = 4)
      for (int j = 0; j < 8; ++j) parts[(size_t)blockIdx.x * 8 + j] = (j < nb && j < 4) ? w[j] * (ds[j] - dot) : 0.f;
    } else
      for (int j = 0; j < nb; ++j) atomic_add_f(dfw + j, w[j] * (ds[j] - dot));
  }
}

// ---- 16-byte-vector forms (C and Cb multiples of the vector, aligned bases): 32-bit index math only -- the scalar
// kernels above spend most of their time in the 64-bit `i % C` / `i / C` per element ----
template <typename T>
__global__ __launch_bounds__(256) void hybrid_fwd_vec_kernel(const T* x, const float* fw, T* y, uint32_t nvec, int nb, int Cb, int C) {
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  float w[8];
  softmax_small(fw, nb, w);
  const uint32_t vpr = C / VEC;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const uint32_t cv = v % vpr;
    const float wb = w[(cv * VEC) / Cb];
    const vec_t xv = *reinterpret_cast<const vec_t*>(x + (size_t)v * VEC);
    vec_t o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = from_f<T>(to_f<T>(xv[j]) * wb);
    *reinterpret_cast<vec_t*>(y + (size_t)v * VEC) = o;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void hybrid_bwd_vec_kernel(const T* dy, const T* x, const float* fw, T* dx, float* dfw,
                                                             uint32_t nvec, int nb, int Cb, int C, float* parts) {
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  float w[8], part[8];
  softmax_small(fw, nb, w);
  for (int i = 0; i < 8; ++i) part[i] = 0.f;
  const uint32_t vpr = C / VEC;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const uint32_t cv = v % vpr;
    const int b = (int)((cv * VEC) / Cb);
    const vec_t gv = *reinterpret_cast<const vec_t*>(dy + (size_t)v * VEC);
    const vec_t xv = *reinterpret_cast<const vec_t*>(x + (size_t)v * VEC);
    const float wb = w[b];
    vec_t o;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { const float g = to_f<T>(gv[j]); o[j] = from_f<T>(g * wb); t += g * to_f<T>(xv[j]); }
    *reinterpret_cast<vec_t*>(dx + (size_t)v * VEC) = o;
#pragma unroll
    for (int j = 0; j < 8; ++j) part[j] += (j == b) ? t : 0.f;
  }
  __shared__ float red[8][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s_ = wave_sum(part[j]);
    if (lane == 0) red[j][wave] = s_;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ds[8];
    for (int j = 0; j < nb; ++j) ds[j] = red[j][0] + red[j][1] + red[j][2] + red[j][3];
    float dot = 0.f;
    for (int j = 0; j < nb; ++j) dot += ds[j] * w[j];
    if (parts) {                                            // this workgroup's row of [grid][8] contributions (qavit_ln_param_reduce, C = 4)
      for (int j = 0; j < 8; ++j) parts[(size_t)blockIdx.x * 8 + j] = (j < nb && j < 4) ? w[j] * (ds[j] - dot) : 0.f;
    } else
      for (int j = 0; j < nb; ++j) atomic_add_f(dfw + j, w[j] * (ds[j] - dot));
  }
}
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void scale_add_vec_kernel(const T* a0, const T* u, const float* gamma, T* out, float* dgamma, uint32_t nvec, int C,
                                                            float dp_p, int dp_site, int dp_rows, const int64_t* rng, float* parts) {
  // forward: a0 = x, out = y = x + f*gamma*u.   backward: a0 = dy, out = du = dy*f*gamma, dgamma += sum dy*f*u.
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  const float gm = gamma ? gamma[0] : 1.f;
  const uint32_t key = dp_p > 0.f ? rng_key(rng, dp_site) : 0u;
  const float inv = dp_p > 0.f ? 1.f / (1.f - dp_p) : 1.f;
  const uint32_t vpr = C / VEC;
  float part = 0.f;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    float f = 1.f;
    if (dp_p > 0.f) f = drop_factor(key, (v / vpr) / (uint32_t)dp_rows, dp_p, inv);
    const vec_t av = *reinterpret_cast<const vec_t*>(a0 + (size_t)v * VEC);
    const vec_t uv = *reinterpret_cast<const vec_t*>(u + (size_t)v * VEC);
    vec_t o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (!BWD) o[j] = from_f<T>(to_f<T>(av[j]) + f * gm * to_f<T>(uv[j]));
      else { const float g = to_f<T>(av[j]) * f; o[j] = from_f<T>(g * gm); part += g * to_f<T>(uv[j]); }
    }
    *reinterpret_cast<vec_t*>(out + (size_t)v * VEC) = o;
  }
  if (BWD && (dgamma || parts)) {
    __shared__ float red[4];
    const float s_ = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s_;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = red[0] + red[1] + red[2] + red[3];
      // parts: this workgroup's row of [grid][4] (plain store; qavit_ln_param_reduce, C = 1, folds the rows in a fixed order: the scalar's
      // gradient is then the same bits on every run -- one float atomic per workgroup on one address is not)
      if (parts) *reinterpret_cast<f32x4*>(parts + (size_t)blockIdx.x * 4) = f32x4{t, 0.f, 0.f, 0.f};
      else atomic_add_f(dgamma, t);
    }
  }
}

// SplitFusion's learnable blend (HQAViT_CIFAR100.py:959-963): y = s0*a + s1*b, s = softmax(fw[2]).
// bwd: da = s0*dy, db = s1*dy, dfw += softmax-jacobian of (sum dy*a, sum dy*b) -- per-workgroup partials are linear in the
// sums, so each workgroup adds its own contribution.  (The stock mul/sum chain reduces with a memset-initialised
// semaphore buffer; memset nodes of a replayed hipGraph were observed to race: garbage fusion-weight gradients.)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void mix2_vec_kernel(const T* a, const T* b, const T* dy, const float* fw, T* o0, T* o1, float* dfw, uint32_t nvec, float* parts) {
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  float w[8];
  softmax_small(fw, 2, w);
  float p0 = 0.f, p1 = 0.f;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const vec_t av = *reinterpret_cast<const vec_t*>(a + (size_t)v * VEC);
    const vec_t bv = *reinterpret_cast<const vec_t*>(b + (size_t)v * VEC);
    vec_t x0, x1;
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) x0[j] = from_f<T>(w[0] * to_f<T>(av[j]) + w[1] * to_f<T>(bv[j]));
      *reinterpret_cast<vec_t*>(o0 + (size_t)v * VEC) = x0;
    } else {
      const vec_t gv = *reinterpret_cast<const vec_t*>(dy + (size_t)v * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float g = to_f<T>(gv[j]);
        x0[j] = from_f<T>(g * w[0]); x1[j] = from_f<T>(g * w[1]);
        p0 += g * to_f<T>(av[j]); p1 += g * to_f<T>(bv[j]);
      }
      *reinterpret_cast<vec_t*>(o0 + (size_t)v * VEC) = x0;
      *reinterpret_cast<vec_t*>(o1 + (size_t)v * VEC) = x1;
    }
  }
  if (BWD && (dfw || parts)) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float s0 = wave_sum(p0), s1 = wave_sum(p1);
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float d0 = red[0][0] + red[0][1] + red[0][2] + red[0][3], d1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      const float dot = d0 * w[0] + d1 * w[1];
      if (parts) *reinterpret_cast<f32x4*>(parts + (size_t)blockIdx.x * 4) = f32x4{w[0] * (d0 - dot), w[1] * (d1 - dot), 0.f, 0.f};   // [grid][4]: reduce desc C = 2
      else { atomic_add_f(dfw + 0, w[0] * (d0 - dot)); atomic_add_f(dfw + 1, w[1] * (d1 - dot)); }
    }
  }
}

// The same blend with SplitFusion's second operand built in place (HQAViT_CIFAR100.py:953-963): y = s0*a + s1*(t + dropout(h)).
// One launch instead of dropout + add + blend forward, and instead of blend-backward + dropout-backward.
// bwd: da = s0*dy, dt = s1*dy, dh = dt * mask, dfw through the softmax from (sum dy*a, sum dy*(t + dropout(h))).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void mix3_vec_kernel(const T* a, const T* t, const T* h, const T* dy, const float* fw, T* o0, T* o1, T* o2, float* dfw,
                                                       uint32_t nvec, float p, int site, const int64_t* rng, float* parts) {
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  float w[8];
  softmax_small(fw, 2, w);
  const uint32_t key = p > 0.f ? rng_key(rng, site) : 0u;
  const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float p0 = 0.f, p1 = 0.f;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const vec_t av = *reinterpret_cast<const vec_t*>(a + (size_t)v * VEC);
    const vec_t tv = *reinterpret_cast<const vec_t*>(t + (size_t)v * VEC);
    const vec_t hv = *reinterpret_cast<const vec_t*>(h + (size_t)v * VEC);
    float f[VEC], b[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      f[j] = p > 0.f ? drop_factor(key, v * (uint32_t)VEC + (uint32_t)j, p, inv) : 1.f;
      // the unfused chain rounds dropout(h) and the sum to T before the blend: keep its values
      b[j] = to_f<T>(from_f<T>(to_f<T>(tv[j]) + to_f<T>(from_f<T>(to_f<T>(hv[j]) * f[j]))));
    }
    vec_t x0, x1, x2;
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) x0[j] = from_f<T>(w[0] * to_f<T>(av[j]) + w[1] * b[j]);
      *reinterpret_cast<vec_t*>(o0 + (size_t)v * VEC) = x0;
    } else {
      const vec_t gv = *reinterpret_cast<const vec_t*>(dy + (size_t)v * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float g = to_f<T>(gv[j]);
        x0[j] = from_f<T>(g * w[0]);
        x1[j] = from_f<T>(g * w[1]);
        x2[j] = from_f<T>(to_f<T>(x1[j]) * f[j]);
        p0 += g * to_f<T>(av[j]); p1 += g * b[j];
      }
      *reinterpret_cast<vec_t*>(o0 + (size_t)v * VEC) = x0;
      *reinterpret_cast<vec_t*>(o1 + (size_t)v * VEC) = x1;
      *reinterpret_cast<vec_t*>(o2 + (size_t)v * VEC) = x2;
    }
  }
  if (BWD && (dfw || parts)) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float s0 = wave_sum(p0), s1 = wave_sum(p1);
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float d0 = red[0][0] + red[0][1] + red[0][2] + red[0][3], d1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      const float dot = d0 * w[0] + d1 * w[1];
      if (parts) *reinterpret_cast<f32x4*>(parts + (size_t)blockIdx.x * 4) = f32x4{w[0] * (d0 - dot), w[1] * (d1 - dot), 0.f, 0.f};   // [grid][4]: reduce desc C = 2
      else { atomic_add_f(dfw + 0, w[0] * (d0 - dot)); atomic_add_f(dfw + 1, w[1] * (d1 - dot)); }
    }
  }
}

// Device-side CutMix / MixUp of the input batch (train_epoch, HQAViT_CIFAR100.py:1381-1399).  plan[6] (device):
// mode (0 none, 1 cutmix, 2 mixup), lambda, x1, y1, x2, y2.  out[b] = in[b] with the box pasted from in[perm[b]]
// (cutmix) or lam*in[b] + (1-lam)*in[perm[b]] (mixup).  One thread per 4 pixels of a row (W % 4 == 0).
__global__ __launch_bounds__(256) void mix_apply_kernel(const float* x, const int64_t* perm, const float* plan, float* out, int B, int C, int H, int W) {
  const int mode = (int)plan[0];
  const float lam = plan[1];
  const int x1 = (int)plan[2], y1 = (int)plan[3], x2 = (int)plan[4], y2 = (int)plan[5];
  const uint32_t w4 = W / 4, per_img = (uint32_t)C * H * w4, total = (uint32_t)B * per_img;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t b = i / per_img, r = i - b * per_img;
    const uint32_t xx = (r % w4) * 4, yy = (r / w4) % H;
    const size_t o = (size_t)b * per_img * 4 + (size_t)r * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + o);
    f32x4 v = a;
    if (mode != 0) {
      const size_t o2 = (size_t)perm[b] * per_img * 4 + (size_t)r * 4;
      const f32x4 c = *reinterpret_cast<const f32x4*>(x + o2);
      if (mode == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = lam * a[j] + (1.f - lam) * c[j];
      } else if ((int)yy >= y1 && (int)yy < y2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int px = (int)xx + j; v[j] = (px >= x1 && px < x2) ? c[j] : a[j]; }
      }
    }
    *reinterpret_cast<f32x4*>(out + o) = v;
  }
}

// Uniform random permutation of 0..B-1 (the torch.randperm of train_epoch :1383,1395) without a library sort: random
// 32-bit keys from the counter RNG (seed, step, site, index), bitonic sort of (key, index) pairs in LDS by one workgroup.
// Ties (probability ~B^2 / 2^33) are broken by index, so the result is always a permutation.
__global__ __launch_bounds__(256) void rand_perm_kernel(int64_t* perm, int B, int P, const int64_t* rng, int site) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sk[];     // keys [P], then indices [P]
  uint32_t* si = sk + P;
  const uint32_t key = rng_key(rng, site);
  for (int i = threadIdx.x; i < P; i += 256) {
    sk[i] = i < B ? (mix32((uint32_t)i * 0x9E3779B9U ^ key) >> 1) : 0xFFFFFFFFu;
    si[i] = (uint32_t)i;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const uint32_t ka = sk[i], kb = sk[l], ia = si[i], ib = si[l];
          const bool gt = ka > kb || (ka == kb && ia > ib);
          if (gt == up) { sk[i] = kb; sk[l] = ka; si[i] = ib; si[l] = ia; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < B; i += 256) perm[i] = (int64_t)si[i];
}

// SplitFusion gate (HQAViT_CIFAR100.py:945-949): y = t + sigmoid(g) * r.   bwd: dt = dy, dr = dy*s, dg = dy*r*s*(1-s).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gate_mix_kernel(const T* t, const T* r, const T* g, const T* dy, T* o0, T* o1, uint32_t nvec) {
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const vec_t rv = *reinterpret_cast<const vec_t*>(r + (size_t)v * VEC);
    const vec_t gv = *reinterpret_cast<const vec_t*>(g + (size_t)v * VEC);
    vec_t a, b;
    if (!BWD) {
      const vec_t tv = *reinterpret_cast<const vec_t*>(t + (size_t)v * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float sg = 1.f / (1.f + __expf(-to_f<T>(gv[j])));
        a[j] = from_f<T>(to_f<T>(tv[j]) + sg * to_f<T>(rv[j]));
      }
      *reinterpret_cast<vec_t*>(o0 + (size_t)v * VEC) = a;
    } else {
      const vec_t dv = *reinterpret_cast<const vec_t*>(dy + (size_t)v * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float sg = 1.f / (1.f + __expf(-to_f<T>(gv[j]))), d = to_f<T>(dv[j]);
        a[j] = from_f<T>(d * sg);
        b[j] = from_f<T>(d * to_f<T>(rv[j]) * sg * (1.f - sg));
      }
      *reinterpret_cast<vec_t*>(o0 + (size_t)v * VEC) = a;
      *reinterpret_cast<vec_t*>(o1 + (size_t)v * VEC) = b;
    }
  }
}

// out = sum of k same-shape tensors (k <= 8): the gradient fan-in of a tensor with k consumers as ONE pass
// (autograd would issue k-1 pairwise adds, each reading two tensors and writing one)
struct SumPtrs { const void* p[8]; };
template <typename T>
__global__ __launch_bounds__(256) void sum_k_kernel(SumPtrs P, int k, T* out, uint32_t nvec) {
  constexpr int VEC = Vec<T>::N;
  typedef typename Vec<T>::type vec_t;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    vec_t x[8];                                           // all k inputs requested before the first add (a run-time loop of load + add is k dependent round trips)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < k) x[i] = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(P.p[i]) + (size_t)v * VEC);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < k) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += to_f<T>(x[i][j]);
      }
    vec_t o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = from_f<T>(acc[j]);
    *reinterpret_cast<vec_t*>(out + (size_t)v * VEC) = o;
  }
}

template <typename T>
static bool vec_ok(int C, int Cb, int64_t n, const void* p0, const void* p1, const void* p2) {
  constexpr int VEC = Vec<T>::N;
  return C % VEC == 0 && Cb % VEC == 0 && n / VEC < 0x7fffffffLL &&
         ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
}

// y = x + droppath(gamma * u)
template <typename T>
__global__ __launch_bounds__(256) void scale_add_fwd_kernel(const T* x, const T* u, const float* gamma, T* y, int64_t n, int C,
                                                            float dp_p, int dp_site, int dp_rows, const int64_t* rng) {
  const float gm = gamma ? gamma[0] : 1.f;
  const uint32_t key = dp_p > 0.f ? rng_key(rng, dp_site) : 0u;
  const float inv = dp_p > 0.f ? 1.f / (1.f - dp_p) : 1.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float f = gm;
    if (dp_p > 0.f) f *= drop_factor(key, (uint32_t)((i / C) / dp_rows), dp_p, inv);
    y[i] = from_f<T>(to_f<T>(x[i]) + f * to_f<T>(u[i]));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const T* dy, const T* u, const float* gamma, T* du, float* dgamma,
                                                            int64_t n, int C, float dp_p, int dp_site, int dp_rows, const int64_t* rng, float* parts) {
  const float gm = gamma ? gamma[0] : 1.f;
  const uint32_t key = dp_p > 0.f ? rng_key(rng, dp_site) : 0u;
  const float inv = dp_p > 0.f ? 1.f / (1.f - dp_p) : 1.f;
  float part = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float f = 1.f;
    if (dp_p > 0.f) f = drop_factor(key, (uint32_t)((i / C) / dp_rows), dp_p, inv);
    const float g = to_f<T>(dy[i]) * f;
    du[i] = from_f<T>(g * gm);
    part += g * to_f<T>(u[i]);
  }
  if (dgamma || parts) {
    __shared__ float red[4];
    const float s = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = red[0] + red[1] + red[2] + red[3];
      if (parts) *reinterpret_cast<f32x4*>(parts + (size_t)blockIdx.x * 4) = f32x4{t, 0.f, 0.f, 0.f};
      else atomic_add_f(dgamma, t);
    }
  }
}

// cols[(b*Hp*Wp + py*Wp + px), c*p*p + dy*p + dx] = img[b, c, py*p+dy, px*p+dx]
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* img, T* cols, int B, int Cin, int H, int W, int p) {
  const int Hp = H / p, Wp = W / p, K = Cin * p * p;
  const int64_t total = (int64_t)B * Cin * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % Cin);
    const int b = (int)(i / ((int64_t)W * H * Cin));
    const int py = y / p, dy = y - py * p, px = x / p, dx = x - px * p;
    if (py < Hp && px < Wp)
      cols[((size_t)b * Hp * Wp + py * Wp + px) * K + c * p * p + dy * p + dx] = from_f<T>(img[i]);
  }
}

// ------------------------------------------------------------------------------------------------
// optimizer: l2 norm + fused AdamW (decoupled weight decay, bias-corrected; torch.optim.AdamW semantics)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* g, int64_t n, float* partial) {
  float s = 0.f;
  const bool vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  const int64_t n4 = vec ? (n >> 2) : 0;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + step < n4; i += 2 * step) {
    const f32x4 a = g4[i], b = g4[i + step];
    s += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]) + (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
  }
  for (; i < n4; i += step) { const f32x4 a = g4[i]; s += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]); }
  for (int64_t j = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += step) s += g[j] * g[j];
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void l2_final_kernel(const float* partial, int nparts, float* out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += partial[i];
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    out[0] = nrm;
    const float mx = out[1];                       // running maximum over calls; a NaN sticks
    out[1] = (nrm > mx || nrm != nrm) ? nrm : mx;
  }
}
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, const uint8_t* skip, int64_t n,
                                                    const float* lr_dev, float b1, float b2, float eps, float wd,
                                                    const float* step_dev, const float* gnorm_dev, float max_norm) {
  const float lr = lr_dev[0], step = step_dev[0];
  float clip = 1.f;
  if (gnorm_dev && max_norm > 0.f) { clip = max_norm / (gnorm_dev[0] + 1e-6f); if (clip > 1.f) clip = 1.f; }
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (skip && skip[i]) continue;
    const float gi = g[i] * clip;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
  }
}

}  // namespace qv

using namespace qv;

extern "C" int qavit_version(void) { return 1; }
extern "C" const char* qavit_last_error(void) { return g_err; }

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, NAME)                   \
  do {                                                                 \
    if ((dtype) == QAVIT_F32) { CALL_F32; }                            \
    else if ((dtype) == QAVIT_BF16) { CALL_BF16; }                     \
    else return set_error(QAVIT_EINVAL, NAME ": unknown dtype");       \
  } while (0)

extern "C" int qavit_pack_weights(int dtype, const qavit_pack_desc* descs_dev, int n_desc, int max_elems, void* stream) {
  if (!descs_dev || n_desc <= 0) return set_error(QAVIT_EINVAL, "pack_weights: empty descriptor table");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int tiles = (max_elems + 1023) / 1024;
  if (tiles < 1) tiles = 1;
  if (tiles > 64) tiles = 64;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((pack_kernel<float>), dim3(tiles, n_desc), dim3(256), 0, st, descs_dev, tiles),
             hipLaunchKernelGGL((pack_kernel<bf16>), dim3(tiles, n_desc), dim3(256), 0, st, descs_dev, tiles), "pack_weights");
  return check_launch("pack_weights");
}

extern "C" int qavit_rng_advance(int64_t* rng, void* stream) {
  if (!rng) return set_error(QAVIT_EINVAL, "rng_advance: null");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), rng);
  return check_launch("rng_advance");
}

extern "C" int qavit_stamp(uint64_t* dst, void* stream) {
  if (!dst) return set_error(QAVIT_EINVAL, "stamp: null");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), dst);
  return check_launch("stamp");
}

extern "C" int qavit_dropout(int dtype, const void* x, void* y, int64_t n, float p, int site, const int64_t* rng, void* stream) {
  if (!x || !y || !rng || n <= 0 || p < 0.f || p >= 1.f) return set_error(QAVIT_EINVAL, "dropout: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = blocks_for(n, 1024);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((dropout_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, n, p, site, rng),
             hipLaunchKernelGGL((dropout_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n, p, site, rng), "dropout");
  return check_launch("dropout");
}

extern "C" int qavit_token_mean_fwd(int dtype, const void* x, void* y, int B, int N, int C, void* stream) {
  if (!x || !y || B <= 0 || N <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "token_mean_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((token_mean_fwd_kernel<float>), dim3(B), dim3(256), 0, st, (const float*)x, (float*)y, B, N, C),
             hipLaunchKernelGGL((token_mean_fwd_kernel<bf16>), dim3(B), dim3(256), 0, st, (const bf16*)x, (bf16*)y, B, N, C), "token_mean_fwd");
  return check_launch("token_mean_fwd");
}
extern "C" int qavit_token_mean_bwd(int dtype, const void* dy, void* dx, int B, int N, int C, void* stream) {
  if (!dy || !dx || B <= 0 || N <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "token_mean_bwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((token_mean_bwd_kernel<float>), dim3(B), dim3(256), 0, st, (const float*)dy, (float*)dx, B, N, C),
             hipLaunchKernelGGL((token_mean_bwd_kernel<bf16>), dim3(B), dim3(256), 0, st, (const bf16*)dy, (bf16*)dx, B, N, C), "token_mean_bwd");
  return check_launch("token_mean_bwd");
}

extern "C" int qavit_hybrid_fuse_fwd(int dtype, const void* x, const float* fw, void* y, int rows, int nb, int Cb, void* stream) {
  if (!x || !fw || !y || rows <= 0 || nb <= 0 || nb > 8 || Cb <= 0) return set_error(QAVIT_EINVAL, "hybrid_fuse_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)rows * nb * Cb;
  if (dtype == QAVIT_BF16 && vec_ok<bf16>(nb * Cb, Cb, n, x, y, x)) {
    const uint32_t nvec = (uint32_t)(n / 8);
    hipLaunchKernelGGL((hybrid_fwd_vec_kernel<bf16>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const bf16*)x, fw, (bf16*)y, nvec, nb, Cb, nb * Cb);
    return check_launch("hybrid_fuse_fwd");
  }
  if (dtype == QAVIT_F32 && vec_ok<float>(nb * Cb, Cb, n, x, y, x)) {
    const uint32_t nvec = (uint32_t)(n / 4);
    hipLaunchKernelGGL((hybrid_fwd_vec_kernel<float>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const float*)x, fw, (float*)y, nvec, nb, Cb, nb * Cb);
    return check_launch("hybrid_fuse_fwd");
  }
  const int g = blocks_for(n, 1024);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((hybrid_fwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, fw, (float*)y, n, nb, Cb),
             hipLaunchKernelGGL((hybrid_fwd_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)x, fw, (bf16*)y, n, nb, Cb), "hybrid_fuse_fwd");
  return check_launch("hybrid_fuse_fwd");
}
extern "C" int qavit_hybrid_fuse_bwd(int dtype, const void* dy, const void* x, const float* fw, void* dx, float* dfw,
                                     int rows, int nb, int Cb, float* part_ws, int* nparts, void* stream) {
  if (!dy || !x || !fw || !dx || !dfw || rows <= 0 || nb <= 0 || nb > 8 || Cb <= 0) return set_error(QAVIT_EINVAL, "hybrid_fuse_bwd: bad arguments");
  if (part_ws && (!nparts || (reinterpret_cast<uintptr_t>(part_ws) & 15))) return set_error(QAVIT_EINVAL, "hybrid_fuse_bwd: part_ws needs nparts and 16-byte alignment");
  if (nb > 4) part_ws = nullptr;                            // rows of [8] = (4 logits, 4 unused): more branches keep the atomics
  if (nparts) *nparts = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)rows * nb * Cb;
  if (dtype == QAVIT_BF16 && vec_ok<bf16>(nb * Cb, Cb, n, x, dy, dx)) {
    const uint32_t nvec = (uint32_t)(n / 8);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((hybrid_bwd_vec_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, fw, (bf16*)dx, dfw, nvec, nb, Cb, nb * Cb, part_ws);
    return check_launch("hybrid_fuse_bwd");
  }
  if (dtype == QAVIT_F32 && vec_ok<float>(nb * Cb, Cb, n, x, dy, dx)) {
    const uint32_t nvec = (uint32_t)(n / 4);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((hybrid_bwd_vec_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)x, fw, (float*)dx, dfw, nvec, nb, Cb, nb * Cb, part_ws);
    return check_launch("hybrid_fuse_bwd");
  }
  const int g = blocks_for(n, 4096, 1024);
  if (part_ws) *nparts = g;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((hybrid_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)x, fw, (float*)dx, dfw, n, nb, Cb, part_ws),
             hipLaunchKernelGGL((hybrid_bwd_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, fw, (bf16*)dx, dfw, n, nb, Cb, part_ws), "hybrid_fuse_bwd");
  return check_launch("hybrid_fuse_bwd");
}

extern "C" int qavit_gate_mix_fwd(int dtype, const void* t, const void* r, const void* g, void* y, int64_t n, void* stream) {
  if (!t || !r || !g || !y || n <= 0) return set_error(QAVIT_EINVAL, "gate_mix_fwd: bad arguments");
  const int vec = dtype == QAVIT_BF16 ? 8 : 4;
  if (((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y)) & 15) || n % vec || n / vec >= 0x7fffffffLL)
    return set_error(QAVIT_EINVAL, "gate_mix_fwd: 16-byte aligned operands, element count a multiple of the 16-byte vector");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t nvec = (uint32_t)(n / vec);
  if (dtype == QAVIT_BF16) hipLaunchKernelGGL((gate_mix_kernel<bf16, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const bf16*)t, (const bf16*)r, (const bf16*)g, (const bf16*)nullptr, (bf16*)y, (bf16*)nullptr, nvec);
  else if (dtype == QAVIT_F32) hipLaunchKernelGGL((gate_mix_kernel<float, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const float*)t, (const float*)r, (const float*)g, (const float*)nullptr, (float*)y, (float*)nullptr, nvec);
  else return set_error(QAVIT_EINVAL, "gate_mix_fwd: unknown dtype");
  return check_launch("gate_mix_fwd");
}
extern "C" int qavit_gate_mix_bwd(int dtype, const void* dy, const void* r, const void* g, void* dr, void* dg, int64_t n, void* stream) {
  if (!dy || !r || !g || !dr || !dg || n <= 0) return set_error(QAVIT_EINVAL, "gate_mix_bwd: bad arguments");
  const int vec = dtype == QAVIT_BF16 ? 8 : 4;
  if (((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dr) | reinterpret_cast<uintptr_t>(dg)) & 15) || n % vec || n / vec >= 0x7fffffffLL)
    return set_error(QAVIT_EINVAL, "gate_mix_bwd: 16-byte aligned operands, element count a multiple of the 16-byte vector");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t nvec = (uint32_t)(n / vec);
  if (dtype == QAVIT_BF16) hipLaunchKernelGGL((gate_mix_kernel<bf16, true>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const bf16*)nullptr, (const bf16*)r, (const bf16*)g, (const bf16*)dy, (bf16*)dr, (bf16*)dg, nvec);
  else if (dtype == QAVIT_F32) hipLaunchKernelGGL((gate_mix_kernel<float, true>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const float*)nullptr, (const float*)r, (const float*)g, (const float*)dy, (float*)dr, (float*)dg, nvec);
  else return set_error(QAVIT_EINVAL, "gate_mix_bwd: unknown dtype");
  return check_launch("gate_mix_bwd");
}

extern "C" int qavit_sum_k(int dtype, const void* const* xs, int k, void* out, int64_t n, void* stream) {
  if (!xs || !out || k <= 0 || k > 8 || n <= 0) return set_error(QAVIT_EINVAL, "sum_k: bad arguments (1 <= k <= 8)");
  const int vec = dtype == QAVIT_BF16 ? 8 : 4;
  SumPtrs P;
  uintptr_t al = reinterpret_cast<uintptr_t>(out);
  for (int i = 0; i < k; ++i) { if (!xs[i]) return set_error(QAVIT_EINVAL, "sum_k: null operand"); P.p[i] = xs[i]; al |= reinterpret_cast<uintptr_t>(xs[i]); }
  if ((al & 15) || n % vec || n / vec >= 0x7fffffffLL) return set_error(QAVIT_EINVAL, "sum_k: 16-byte aligned operands, element count a multiple of the 16-byte vector");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t nvec = (uint32_t)(n / vec);
  if (dtype == QAVIT_BF16) hipLaunchKernelGGL((sum_k_kernel<bf16>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, P, k, (bf16*)out, nvec);
  else if (dtype == QAVIT_F32) hipLaunchKernelGGL((sum_k_kernel<float>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, P, k, (float*)out, nvec);
  else return set_error(QAVIT_EINVAL, "sum_k: unknown dtype");
  return check_launch("sum_k");
}

extern "C" int qavit_rand_perm(int64_t* perm, int B, const int64_t* rng, int site, void* stream) {
  if (!perm || !rng || B <= 0 || B > 16384) return set_error(QAVIT_EINVAL, "rand_perm: 1 <= B <= 16384");
  int P = 1;
  while (P < B) P <<= 1;
  hipLaunchKernelGGL(rand_perm_kernel, dim3(1), dim3(256), (size_t)2 * P * sizeof(uint32_t), reinterpret_cast<hipStream_t>(stream), perm, B, P, rng, site);
  return check_launch("rand_perm");
}

extern "C" int qavit_mix_apply(const float* x, const int64_t* perm, const float* plan, float* out, int B, int C, int H, int W, void* stream) {
  if (!x || !perm || !plan || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return set_error(QAVIT_EINVAL, "mix_apply: bad arguments");
  if (W % 4 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) || x == out || (int64_t)B * C * H * W >= 0x7fffffffLL)
    return set_error(QAVIT_EINVAL, "mix_apply: W must be a multiple of 4, 16-byte aligned fp32 NCHW batches, out != x");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t nv = (int64_t)B * C * H * W / 4;
  hipLaunchKernelGGL(mix_apply_kernel, dim3(blocks_for(nv, 512, 2048)), dim3(256), 0, st, x, perm, plan, out, B, C, H, W);
  return check_launch("mix_apply");
}

extern "C" int qavit_mix2_fwd(int dtype, const void* a, const void* b, const float* fw, void* y, int64_t n, void* stream) {
  if (!a || !b || !fw || !y || n <= 0) return set_error(QAVIT_EINVAL, "mix2_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_BF16 && vec_ok<bf16>(8, 8, n, a, b, y) && n % 8 == 0) {
    const uint32_t nvec = (uint32_t)(n / 8);
    hipLaunchKernelGGL((mix2_vec_kernel<bf16, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)nullptr, fw, (bf16*)y, (bf16*)nullptr, (float*)nullptr, nvec, (float*)nullptr);
  } else if (dtype == QAVIT_F32 && vec_ok<float>(4, 4, n, a, b, y) && n % 4 == 0) {
    const uint32_t nvec = (uint32_t)(n / 4);
    hipLaunchKernelGGL((mix2_vec_kernel<float, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)nullptr, fw, (float*)y, (float*)nullptr, (float*)nullptr, nvec, (float*)nullptr);
  } else return set_error(QAVIT_EINVAL, "mix2_fwd: element count must be a multiple of the 16-byte vector, 16-byte aligned operands");
  return check_launch("mix2_fwd");
}
extern "C" int qavit_mix3_fwd(int dtype, const void* a, const void* t, const void* h, const float* fw, void* y, int64_t n,
                              float drop_p, int drop_site, const int64_t* rng, void* stream) {
  if (!a || !t || !h || !fw || !y || n <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !rng)) return set_error(QAVIT_EINVAL, "mix3_fwd: bad arguments");
  if (n > 0xffffffffll) return set_error(QAVIT_EINVAL, "mix3_fwd: the dropout index is 32 bits");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (dtype == QAVIT_BF16 && al && n % 8 == 0) {
    const uint32_t nvec = (uint32_t)(n / 8);
    hipLaunchKernelGGL((mix3_vec_kernel<bf16, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const bf16*)a, (const bf16*)t, (const bf16*)h, (const bf16*)nullptr, fw,
                       (bf16*)y, (bf16*)nullptr, (bf16*)nullptr, (float*)nullptr, nvec, drop_p, drop_site, rng, (float*)nullptr);
  } else if (dtype == QAVIT_F32 && al && n % 4 == 0) {
    const uint32_t nvec = (uint32_t)(n / 4);
    hipLaunchKernelGGL((mix3_vec_kernel<float, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const float*)a, (const float*)t, (const float*)h, (const float*)nullptr, fw,
                       (float*)y, (float*)nullptr, (float*)nullptr, (float*)nullptr, nvec, drop_p, drop_site, rng, (float*)nullptr);
  } else return set_error(QAVIT_EINVAL, "mix3_fwd: element count must be a multiple of the 16-byte vector, 16-byte aligned operands");
  return check_launch("mix3_fwd");
}

extern "C" int qavit_mix3_bwd(int dtype, const void* dy, const void* a, const void* t, const void* h, const float* fw, void* da, void* dt, void* dh,
                              float* dfw, int64_t n, float drop_p, int drop_site, const int64_t* rng, float* part_ws, int* nparts, void* stream) {
  if (!dy || !a || !t || !h || !fw || !da || !dt || !dh || n <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !rng))
    return set_error(QAVIT_EINVAL, "mix3_bwd: bad arguments");
  if (part_ws && (!nparts || (reinterpret_cast<uintptr_t>(part_ws) & 15))) return set_error(QAVIT_EINVAL, "mix3_bwd: part_ws needs nparts and 16-byte alignment");
  if (nparts) *nparts = 0;
  if (n > 0xffffffffll) return set_error(QAVIT_EINVAL, "mix3_bwd: the dropout index is 32 bits");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool al = ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(h) |
                    reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(dt) | reinterpret_cast<uintptr_t>(dh)) & 15) == 0;
  if (dtype == QAVIT_BF16 && al && n % 8 == 0) {
    const uint32_t nvec = (uint32_t)(n / 8);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((mix3_vec_kernel<bf16, true>), dim3(g), dim3(256), 0, st, (const bf16*)a, (const bf16*)t, (const bf16*)h, (const bf16*)dy, fw,
                       (bf16*)da, (bf16*)dt, (bf16*)dh, dfw, nvec, drop_p, drop_site, rng, part_ws);
  } else if (dtype == QAVIT_F32 && al && n % 4 == 0) {
    const uint32_t nvec = (uint32_t)(n / 4);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((mix3_vec_kernel<float, true>), dim3(g), dim3(256), 0, st, (const float*)a, (const float*)t, (const float*)h, (const float*)dy, fw,
                       (float*)da, (float*)dt, (float*)dh, dfw, nvec, drop_p, drop_site, rng, part_ws);
  } else return set_error(QAVIT_EINVAL, "mix3_bwd: element count must be a multiple of the 16-byte vector, 16-byte aligned operands");
  return check_launch("mix3_bwd");
}

extern "C" int qavit_mix2_bwd(int dtype, const void* dy, const void* a, const void* b, const float* fw, void* da, void* db, float* dfw, int64_t n,
                              float* part_ws, int* nparts, void* stream) {
  if (!dy || !a || !b || !fw || !da || !db || n <= 0) return set_error(QAVIT_EINVAL, "mix2_bwd: bad arguments");
  if (part_ws && (!nparts || (reinterpret_cast<uintptr_t>(part_ws) & 15))) return set_error(QAVIT_EINVAL, "mix2_bwd: part_ws needs nparts and 16-byte alignment");
  if (nparts) *nparts = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == QAVIT_BF16 && vec_ok<bf16>(8, 8, n, a, b, dy) && vec_ok<bf16>(8, 8, n, da, db, dy) && n % 8 == 0) {
    const uint32_t nvec = (uint32_t)(n / 8);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((mix2_vec_kernel<bf16, true>), dim3(g), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)dy, fw, (bf16*)da, (bf16*)db, dfw, nvec, part_ws);
  } else if (dtype == QAVIT_F32 && vec_ok<float>(4, 4, n, a, b, dy) && vec_ok<float>(4, 4, n, da, db, dy) && n % 4 == 0) {
    const uint32_t nvec = (uint32_t)(n / 4);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((mix2_vec_kernel<float, true>), dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)dy, fw, (float*)da, (float*)db, dfw, nvec, part_ws);
  } else return set_error(QAVIT_EINVAL, "mix2_bwd: element count must be a multiple of the 16-byte vector, 16-byte aligned operands");
  return check_launch("mix2_bwd");
}

extern "C" int qavit_scale_add_fwd(int dtype, const void* x, const void* u, const float* gamma, void* y, int rows, int C,
                                   float dp_p, int dp_site, int dp_rows, const int64_t* rng, void* stream) {
  if (!x || !u || !y || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "scale_add_fwd: bad arguments");
  if (dp_p > 0.f && (!rng || dp_rows <= 0)) return set_error(QAVIT_EINVAL, "scale_add_fwd: drop-path needs rng and rows-per-sample");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)rows * C;
  if (dtype == QAVIT_BF16 && vec_ok<bf16>(C, C, n, x, u, y)) {
    const uint32_t nvec = (uint32_t)(n / 8);
    hipLaunchKernelGGL((scale_add_vec_kernel<bf16, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const bf16*)x, (const bf16*)u, gamma, (bf16*)y, (float*)nullptr, nvec, C, dp_p, dp_site, dp_rows, rng, (float*)nullptr);
    return check_launch("scale_add_fwd");
  }
  if (dtype == QAVIT_F32 && vec_ok<float>(C, C, n, x, u, y)) {
    const uint32_t nvec = (uint32_t)(n / 4);
    hipLaunchKernelGGL((scale_add_vec_kernel<float, false>), dim3(blocks_for(nvec, 512, 2048)), dim3(256), 0, st, (const float*)x, (const float*)u, gamma, (float*)y, (float*)nullptr, nvec, C, dp_p, dp_site, dp_rows, rng, (float*)nullptr);
    return check_launch("scale_add_fwd");
  }
  const int g = blocks_for(n, 1024);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((scale_add_fwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)u, gamma, (float*)y, n, C, dp_p, dp_site, dp_rows, rng),
             hipLaunchKernelGGL((scale_add_fwd_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)x, (const bf16*)u, gamma, (bf16*)y, n, C, dp_p, dp_site, dp_rows, rng), "scale_add_fwd");
  return check_launch("scale_add_fwd");
}
extern "C" int qavit_scale_add_bwd(int dtype, const void* dy, const void* u, const float* gamma, void* du, float* dgamma,
                                   int rows, int C, float dp_p, int dp_site, int dp_rows, const int64_t* rng, float* part_ws, int* nparts, void* stream) {
  if (!dy || !u || !du || rows <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "scale_add_bwd: bad arguments");
  if (part_ws && (!nparts || (reinterpret_cast<uintptr_t>(part_ws) & 15))) return set_error(QAVIT_EINVAL, "scale_add_bwd: part_ws needs nparts and 16-byte alignment");
  if (nparts) *nparts = 0;
  if (dp_p > 0.f && (!rng || dp_rows <= 0)) return set_error(QAVIT_EINVAL, "scale_add_bwd: drop-path needs rng and rows-per-sample");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)rows * C;
  if (dtype == QAVIT_BF16 && vec_ok<bf16>(C, C, n, dy, u, du)) {
    const uint32_t nvec = (uint32_t)(n / 8);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((scale_add_vec_kernel<bf16, true>), dim3(g), dim3(256), 0, st, (const bf16*)dy, (const bf16*)u, gamma, (bf16*)du, dgamma, nvec, C, dp_p, dp_site, dp_rows, rng, part_ws);
    return check_launch("scale_add_bwd");
  }
  if (dtype == QAVIT_F32 && vec_ok<float>(C, C, n, dy, u, du)) {
    const uint32_t nvec = (uint32_t)(n / 4);
    const int g = blocks_for(nvec, 1024, 512);
    if (part_ws) *nparts = g;
    hipLaunchKernelGGL((scale_add_vec_kernel<float, true>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)u, gamma, (float*)du, dgamma, nvec, C, dp_p, dp_site, dp_rows, rng, part_ws);
    return check_launch("scale_add_bwd");
  }
  const int g = blocks_for(n, 4096, 1024);
  if (part_ws) *nparts = g;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((scale_add_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dy, (const float*)u, gamma, (float*)du, dgamma, n, C, dp_p, dp_site, dp_rows, rng, part_ws),
             hipLaunchKernelGGL((scale_add_bwd_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)dy, (const bf16*)u, gamma, (bf16*)du, dgamma, n, C, dp_p, dp_site, dp_rows, rng, part_ws), "scale_add_bwd");
  return check_launch("scale_add_bwd");
}

extern "C" int qavit_patchify(int dtype, const float* img, void* cols, int B, int Cin, int H, int W, int p, void* stream) {
  if (!img || !cols || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p)
    return set_error(QAVIT_EINVAL, "patchify: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    // patchify IS im2col with kernel = stride = p and no padding, in the same (c, dy, dx) column order: images that fit LDS take the per-image
    // kernel of dwconv.hip (the image staged with coalesced 16-byte loads, 16-byte row pieces out) instead of one 2-byte store per element
    const int K = Cin * p * p, vn = dtype == QAVIT_F32 ? 4 : 8;
    const size_t esz = dtype == QAVIT_F32 ? 4 : 2;
    if ((dtype == QAVIT_F32 || dtype == QAVIT_BF16) && K % vn == 0 && (size_t)Cin * H * W * 4 <= 96 * 1024 && (Cin * H * W) % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(cols)) & 15) == 0 && (((size_t)K * esz) & 15) == 0)
      return qavit_im2col_ld(dtype, img, 1, cols, K, B, Cin, H, W, p, p, 0, stream);
  }
  const int g = blocks_for((int64_t)B * Cin * H * W, 1024);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((patchify_kernel<float>), dim3(g), dim3(256), 0, st, img, (float*)cols, B, Cin, H, W, p),
             hipLaunchKernelGGL((patchify_kernel<bf16>), dim3(g), dim3(256), 0, st, img, (bf16*)cols, B, Cin, H, W, p), "patchify");
  return check_launch("patchify");
}

extern "C" int qavit_l2norm(const float* g, int64_t n, float* partial, float* out, void* stream) {
  if (!g || !partial || !out || n <= 0) return set_error(QAVIT_EINVAL, "l2norm: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = blocks_for(n, 4096, 1024);   // partial must hold 1024 floats
  hipLaunchKernelGGL(l2_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial);
  hipLaunchKernelGGL(l2_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, out);
  return check_launch("l2norm");
}

// Per-tensor gradient clip of selected segments of the flat gradient buffer: g *= min(1, clip / (||g|| + 1e-6)).
// blockIdx.x = segment, blockIdx.y = chunk of it (the 4x-MLP weights of the stem are 262 k elements: one workgroup per
// segment took 210 us).  Pass 1 leaves sum g^2 per segment in ws[seg]; pass 2 rescales and the last chunk of a segment
// to finish (ticket in ws[nseg + seg]) zeroes both words, so the workspace is clean for the next step without a memset.
__global__ __launch_bounds__(256) void local_clip_norm_kernel(const float* g, const int64_t* seg, float* ws) {
  __shared__ float red[4];
  const float* p = g + seg[2 * blockIdx.x];
  const int64_t n = seg[2 * blockIdx.x + 1];
  float s = 0.f;
  // 16-byte loads, four in flight per thread (segments start on 256-byte boundaries of the flat buffer): as a scalar loop the 262 k-element
  // segments were 64 dependent-looking iterations per thread and the launch took 26 us at the very end of the step
  const bool vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  const int64_t n4 = vec ? (n >> 2) : 0;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
  const int64_t step = (int64_t)gridDim.y * 256;
  int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x;
  for (; i + 3 * step < n4; i += 4 * step) {
    const f32x4 a = p4[i], b = p4[i + step], c = p4[i + 2 * step], d = p4[i + 3 * step];
    s += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]) + (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]) +
         (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]) + (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
  for (; i < n4; i += step) { const f32x4 a = p4[i]; s += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]); }
  for (int64_t j = (n4 << 2) + (int64_t)blockIdx.y * 256 + threadIdx.x; j < n; j += step) { const float v = p[j]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float t = red[0] + red[1] + red[2] + red[3]; if (t != 0.f) atomic_add_f(ws + blockIdx.x, t); }
}
__global__ __launch_bounds__(256) void local_clip_scale_kernel(float* g, const int64_t* seg, float* ws, int nseg, float clip) {
  float* p = g + seg[2 * blockIdx.x];
  const int64_t n = seg[2 * blockIdx.x + 1];
  const float sc = fminf(clip / (sqrtf(ws[blockIdx.x]) + 1e-6f), 1.f);
  if (sc < 1.f) {
    const bool vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    const int64_t n4 = vec ? (n >> 2) : 0;
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.y * 256) { f32x4 a = p4[i]; a[0] *= sc; a[1] *= sc; a[2] *= sc; a[3] *= sc; p4[i] = a; }
    for (int64_t j = (n4 << 2) + (int64_t)blockIdx.y * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.y * 256) p[j] *= sc;
  }
  __syncthreads();                                         // every thread of this chunk has read ws[seg]
  if (threadIdx.x == 0) {
    int* ticket = reinterpret_cast<int*>(ws + nseg) + blockIdx.x;
    if (atomicAdd(ticket, 1) == (int)gridDim.y - 1) { ws[blockIdx.x] = 0.f; *ticket = 0; }
  }
}

extern "C" int qavit_local_clip(float* g, const int64_t* seg, int nseg, float clip, float* ws, void* stream) {
  if (!g || !seg || !ws || nseg <= 0 || !(clip > 0.f)) return set_error(QAVIT_EINVAL, "local_clip: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(nseg, 16);
  hipLaunchKernelGGL(local_clip_norm_kernel, grid, dim3(256), 0, st, g, seg, ws);
  hipLaunchKernelGGL(local_clip_scale_kernel, grid, dim3(256), 0, st, g, seg, ws, nseg, clip);
  return check_launch("local_clip");
}

// dst_a = src_a, dst_b = src_b (same length): the forward-time snapshot of the bank's K and V rows in one launch
// (two clone() calls were two memcpy nodes of the step graph each time a branch reads the bank).
__global__ __launch_bounds__(256) void copy2_kernel(const float* a, const float* b, float* da, float* db, int n4) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * n4; i += gridDim.x * 256) {
    const bool second = i >= n4;
    const int j = second ? i - n4 : i;
    reinterpret_cast<f32x4*>(second ? db : da)[j] = reinterpret_cast<const f32x4*>(second ? b : a)[j];
  }
}
extern "C" int qavit_copy2(const float* src_a, const float* src_b, float* dst_a, float* dst_b, int64_t n, void* stream) {
  if (!src_a || !src_b || !dst_a || !dst_b || n <= 0 || n % 4 || n / 4 >= 0x3fffffffLL) return set_error(QAVIT_EINVAL, "copy2: bad arguments");
  if ((reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(src_b) | reinterpret_cast<uintptr_t>(dst_a) | reinterpret_cast<uintptr_t>(dst_b)) & 15)
    return set_error(QAVIT_EINVAL, "copy2: 16-byte aligned fp32 buffers");
  const int n4 = (int)(n / 4);
  hipLaunchKernelGGL(copy2_kernel, dim3(blocks_for(2 * (int64_t)n4, 256, 1024)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src_a, src_b, dst_a, dst_b, n4);
  return check_launch("copy2");
}

extern "C" int qavit_adamw(float* p, const float* g, float* m, float* v, const uint8_t* skip, int64_t n,
                           const float* lr_dev, float beta1, float beta2, float eps, float wd,
                           const float* step_dev, const float* gnorm_dev, float max_norm, void* stream) {
  if (!p || !g || !m || !v || !lr_dev || !step_dev || n <= 0) return set_error(QAVIT_EINVAL, "adamw: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = blocks_for(n, 1024);
  hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, st, p, g, m, v, skip, n, lr_dev, beta1, beta2, eps, wd, step_dev, gnorm_dev, max_norm);
  return check_launch("adamw");
}
