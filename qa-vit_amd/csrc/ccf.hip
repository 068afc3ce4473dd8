// CCF-FFN middle: h2 = LN2( scale * dwconv3x3( LN1(h) ) + bias ) on channel-last tokens [B, Hs*Ws, C].
// HBM-bound: every activation element is read once and written once; the 3x3 depthwise stencil, both
// LayerNorms and (in backward) all parameter-gradient partial sums stay in LDS.  One workgroup per image.
#include "common.cuh"
#include <stdlib.h>
#include "../../include/qavit.h"
#include "launch.h"

namespace qv {

template <typename T> struct V4;
template <> struct V4<float> { typedef f32x4 type; };
template <> struct V4<bf16> { typedef bf16x4 type; };

// n elements (n % 4 == 0) of an image from global memory to an fp32 LDS tile: four vector loads per thread in flight before the first
// store (the element-wise `for (i = tid; ...) lds[i] = g[i]` with a run-time trip count was one dependent round trip per iteration)
template <typename T, int NTH>
__device__ __forceinline__ void stage_image_f32(float* dst, const T* src, int n) {
  typedef typename V4<T>::type v4;
  const v4* sv = reinterpret_cast<const v4*>(src);
  const int nv = n >> 2;
  for (int base = threadIdx.x; base < nv; base += 4 * NTH) {
    v4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = base + j * NTH; if (i < nv) r[j] = sv[i]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = base + j * NTH;
      if (i < nv) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[4 * i + q] = to_f<T>(r[j][q]);
      }
    }
  }
}

constexpr int F_LN = 1, F_BIAS = 2, F_SCALE = 4;

// row-wise LN over C of buf[N][C] in place; optional stats out
__device__ __forceinline__ void ln_rows(float* buf, int N, int C, const float* g, const float* b, float eps, float* mean_o, float* rstd_o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  for (int r = wave; r < N; r += (int)(blockDim.x >> 6)) {
    float* row = buf + r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c];
    const float mean = wave_sum(s) * invC;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = row[c] - mean; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
    for (int c = lane; c < C; c += 64) row[c] = (row[c] - mean) * rstd * g[c] + b[c];
    if (lane == 0) { if (mean_o) mean_o[r] = mean; if (rstd_o) rstd_o[r] = rstd; }
  }
}

// t[n][c] = scale[c] * sum_{dy,dx} w[c][dy][dx] * a[(y+dy-1, x+dx-1)][c] + bias[c]
__device__ __forceinline__ void dwconv_rows(const float* a, float* t, const float* w, const float* cbias, const float* cscale,
                                            int Hs, int Ws, int C, float* raw) {
  const int N = Hs * Ws;
  for (int i = threadIdx.x; i < N * C; i += (int)blockDim.x) {
    const int n = i / C, c = i - n * C;
    const int y = n / Ws, x = n - y * Ws;
    float s = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = y + dy - 1, xx = x + dx - 1;
        if (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) s += w[c * 9 + dy * 3 + dx] * a[(yy * Ws + xx) * C + c];
      }
    if (cbias) s += cbias[c];
    if (raw) raw[i] = s;
    if (cscale) s *= cscale[c];
    t[i] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(512) void ccf_fwd_kernel(qavit_ccf_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.Hs * p.Ws, C = p.C;
  float* a = sm;               // [N][C]
  float* t = sm + N * C;       // [N][C]
  const T* h = reinterpret_cast<const T*>(p.h);
  T* out = reinterpret_cast<T*>(p.out);
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < N * C; i += (int)blockDim.x) a[i] = to_f<T>(h[(size_t)b * N * C + i]);
    __syncthreads();
    if (p.flags & F_LN) { ln_rows(a, N, C, p.g1, p.b1, p.eps, p.mean1 + (size_t)b * N, p.rstd1 + (size_t)b * N); __syncthreads(); }
    dwconv_rows(a, t, p.w, (p.flags & F_BIAS) ? p.cbias : nullptr, (p.flags & F_SCALE) ? p.cscale : nullptr, p.Hs, p.Ws, C, nullptr);
    __syncthreads();
    if (p.flags & F_LN) { ln_rows(t, N, C, p.g2, p.b2, p.eps, p.mean2 + (size_t)b * N, p.rstd2 + (size_t)b * N); __syncthreads(); }
    for (int i = threadIdx.x; i < N * C; i += (int)blockDim.x) out[(size_t)b * N * C + i] = from_f<T>(t[i]);
  }
}

// Forward on the (wave = rows, lane = channels) mapping with the parameters in registers (C <= 64 * CP): the generic kernel above
// fetches every tap weight and LayerNorm parameter from global memory inside its loops and divides to find (row, channel).
template <typename T, int CP, int NW, int HS, int WS>
__global__ __launch_bounds__(64 * NW) void ccf_fwd2_kernel(qavit_ccf_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hs = HS ? HS : p.Hs, Ws = WS ? WS : p.Ws;      // HS / WS != 0: the map size is a compile-time constant (4 x 4 learned tokens)
  const int N = Hs * Ws, C = p.C;
  float* a = sm;               // [N][C] input, then LN1 output
  float* t = sm + N * C;       // [N][C] conv output
  const T* h = reinterpret_cast<const T*>(p.h);
  T* out = reinterpret_cast<T*>(p.out);
  const bool ln = p.flags & F_LN, hb = p.flags & F_BIAS, hs = p.flags & F_SCALE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  float g1v[CP], b1v[CP], g2v[CP], b2v[CP], csv[CP], cbv[CP], wt[CP][9];
#pragma unroll
  for (int k = 0; k < CP; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < C;
    g1v[k] = (ok && ln) ? p.g1[c] : 0.f; b1v[k] = (ok && ln) ? p.b1[c] : 0.f;
    g2v[k] = (ok && ln) ? p.g2[c] : 0.f; b2v[k] = (ok && ln) ? p.b2[c] : 0.f;
    cbv[k] = (ok && hb) ? p.cbias[c] : 0.f;
    csv[k] = (ok && hs) ? p.cscale[c] : 1.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) wt[k][q] = ok ? p.w[c * 9 + q] : 0.f;
  }
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    __syncthreads();
    stage_image_f32<T, 64 * NW>(a, h + (size_t)b * N * C, N * C);           // (C % 4 == 0 and vector-aligned images: checked by the launcher)
    __syncthreads();
    if (ln) {
      for (int r = wave; r < N; r += NW) {
        float v[CP], s1 = 0.f;
#pragma unroll
        for (int k = 0; k < CP; ++k) { const int c = lane + 64 * k; v[k] = c < C ? a[r * C + c] : 0.f; s1 += v[k]; }
        const float mean = wave_sum(s1) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CP; ++k) { const int c = lane + 64 * k; const float d = c < C ? v[k] - mean : 0.f; s2 += d * d; }
        const float rstd = rsqrtf(wave_sum(s2) * invC + p.eps);
#pragma unroll
        for (int k = 0; k < CP; ++k) { const int c = lane + 64 * k; if (c < C) a[r * C + c] = (v[k] - mean) * rstd * g1v[k] + b1v[k]; }
        if (lane == 0) { p.mean1[(size_t)b * N + r] = mean; p.rstd1[(size_t)b * N + r] = rstd; }
      }
      __syncthreads();
    }
    for (int r = wave; r < N; r += NW) {
      const int y = r / Ws, x = r - y * Ws;
      float v[CP], s1 = 0.f;
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        v[k] = 0.f;
        if (c < C) {
          float sacc = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int yy = y + dy - 1, xx = x + dx - 1;
              if (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) sacc += wt[k][dy * 3 + dx] * a[(yy * Ws + xx) * C + c];
            }
          v[k] = (sacc + cbv[k]) * csv[k];
        }
        s1 += v[k];
      }
      T* orow = out + ((size_t)b * N + r) * C;
      if (ln) {                                              // LN2 on the row this wave just produced
        const float mean = wave_sum(s1) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CP; ++k) { const int c = lane + 64 * k; const float d = c < C ? v[k] - mean : 0.f; s2 += d * d; }
        const float rstd = rsqrtf(wave_sum(s2) * invC + p.eps);
#pragma unroll
        for (int k = 0; k < CP; ++k) { const int c = lane + 64 * k; if (c < C) orow[c] = from_f<T>((v[k] - mean) * rstd * g2v[k] + b2v[k]); }
        if (lane == 0) { p.mean2[(size_t)b * N + r] = mean; p.rstd2[(size_t)b * N + r] = rstd; }
      } else {
#pragma unroll
        for (int k = 0; k < CP; ++k) { const int c = lane + 64 * k; if (c < C) orow[c] = from_f<T>(v[k]); }
      }
    }
  }
  (void)t;
}

// LN backward over rows of LDS buffers: xin holds the LN INPUT rows, gy the output gradient (overwritten by dx).
// pg/pb: [C] LDS partial sums (LDS atomics across the 4 waves)
__device__ __forceinline__ void ln_rows_bwd(const float* xin, float* gy, int N, int C, const float* g, const float* mean, const float* rstd,
                                            float* pg, float* pb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  for (int r = wave; r < N; r += 4) {
    const float* xr = xin + r * C;
    float* gr = gy + r * C;
    const float mu = mean[r], rs = rstd[r];
    float c1 = 0.f, c2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] - mu) * rs, gg = gr[c] * g[c];
      c1 += gg * xh; c2 += gg;
    }
    c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
    for (int c = lane; c < C; c += 64) {
      const float d = gr[c];
      const float xh = (xr[c] - mu) * rs;
      atomicAdd(pg + c, d * xh);
      atomicAdd(pb + c, d);
      gr[c] = rs * (d * g[c] - c2 - xh * c1);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ccf_bwd_kernel(qavit_ccf_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.Hs * p.Ws, C = p.C;
  float* hin = sm;                 // [N][C] raw input h (LN1 input)
  float* a = hin + N * C;          // [N][C] LN1 output
  float* raw = a + N * C;          // [N][C] conv (+bias) before scale
  float* t = raw + N * C;          // [N][C] LN2 input, then gradients in place
  float* part = t + N * C;         // parameter partials: dg1,db1,dg2,db2 [C each], dcb [C], dcs [C], dw [9C]
  float* pg1 = part, *pb1 = part + C, *pg2 = part + 2 * C, *pb2 = part + 3 * C, *pcb = part + 4 * C, *pcs = part + 5 * C, *pw = part + 6 * C;
  const T* h = reinterpret_cast<const T*>(p.h);
  const T* dout = reinterpret_cast<const T*>(p.d_out);
  T* dh = reinterpret_cast<T*>(p.d_h);
  const bool ln = p.flags & F_LN, hb = p.flags & F_BIAS, hs = p.flags & F_SCALE;
  for (int i = threadIdx.x; i < 15 * C; i += 256) part[i] = 0.f;
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < N * C; i += 256) { const float v = to_f<T>(h[(size_t)b * N * C + i]); hin[i] = v; a[i] = v; }
    __syncthreads();
    if (ln) { ln_rows(a, N, C, p.g1, p.b1, p.eps, nullptr, nullptr); __syncthreads(); }
    dwconv_rows(a, t, p.w, hb ? p.cbias : nullptr, hs ? p.cscale : nullptr, p.Hs, p.Ws, C, raw);
    __syncthreads();
    // gradient wrt LN2 output -> wrt t (in place in a scratch: reuse `a`?  a is still needed for dw) -> use t after copying LN2 input
    // step 1: gy := dout ; LN2 backward needs its input t: keep t as xin and put gy into `hin`-independent buffer `raw`? raw is needed for dscale.
    // => fold dscale / dbias first using raw, then reuse raw as the gradient buffer.
    // load dout into registers-per-element order: we process element-wise passes over LDS buffers.
    // pass A: gy (into raw2 = raw after consuming raw) requires dt first, so: gy -> tmp in `a2`:
    // To keep LDS at 4 buffers we stage gy into `t`'s twin only after LN2-bwd; implement LN2-bwd with gy read from global.
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const float invC = 1.f / (float)C;
      for (int r = wave; r < N; r += 4) {
        const T* gr = dout + ((size_t)b * N + r) * C;
        float* tr = t + r * C;
        if (ln) {
          const float mu = p.mean2[(size_t)b * N + r], rs = p.rstd2[(size_t)b * N + r];
          float c1 = 0.f, c2 = 0.f;
          for (int c = lane; c < C; c += 64) {
            const float xh = (tr[c] - mu) * rs, gg = to_f<T>(gr[c]) * p.g2[c];
            c1 += gg * xh; c2 += gg;
          }
          c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC;
          for (int c = lane; c < C; c += 64) {
            const float d = to_f<T>(gr[c]);
            const float xh = (tr[c] - mu) * rs;
            atomicAdd(pg2 + c, d * xh);
            atomicAdd(pb2 + c, d);
            tr[c] = rs * (d * p.g2[c] - c2 - xh * c1);          // dt
          }
        } else {
          for (int c = lane; c < C; c += 64) tr[c] = to_f<T>(gr[c]);
        }
      }
    }
    __syncthreads();
    // dt -> dconv (gradient wrt conv+bias output), dscale, dbias
    for (int i = threadIdx.x; i < N * C; i += 256) {
      const int c = i % C;
      const float dt = t[i];
      if (hs) { atomicAdd(pcs + c, dt * raw[i]); t[i] = dt * p.cscale[c]; }
      if (hb) atomicAdd(pcb + c, t[i]);
    }
    __syncthreads();
    // dw[c][dy][dx] += sum_n dconv[n][c] * a[n + off][c] ;  da[n'][c] = sum_{dy,dx} w[c][dy][dx] dconv[n' - off][c]  (into raw)
    for (int i = threadIdx.x; i < N * C; i += 256) {
      const int n = i / C, c = i - n * C;
      const int y = n / p.Ws, x = n - y * p.Ws;
      const float dc = t[i];
      float da = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int yy = y + dy - 1, xx = x + dx - 1;           // forward tap read by output (y,x)
          if (yy >= 0 && yy < p.Hs && xx >= 0 && xx < p.Ws) atomicAdd(pw + c * 9 + dy * 3 + dx, dc * a[(yy * p.Ws + xx) * C + c]);
          const int yo = y - dy + 1, xo = x - dx + 1;           // outputs that read input (y,x) through tap (dy,dx)
          if (yo >= 0 && yo < p.Hs && xo >= 0 && xo < p.Ws) da += p.w[c * 9 + dy * 3 + dx] * t[(yo * p.Ws + xo) * C + c];
        }
      raw[i] = da;
    }
    __syncthreads();
    if (ln) {
      const float* m1 = p.mean1 + (size_t)b * N;
      const float* r1 = p.rstd1 + (size_t)b * N;
      ln_rows_bwd(hin, raw, N, C, p.g1, m1, r1, pg1, pb1);
      __syncthreads();
    }
    for (int i = threadIdx.x; i < N * C; i += 256) dh[(size_t)b * N * C + i] = from_f<T>(raw[i]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    if (ln) { atomic_add_f(p.dg1 + c, pg1[c]); atomic_add_f(p.db1 + c, pb1[c]); atomic_add_f(p.dg2 + c, pg2[c]); atomic_add_f(p.db2 + c, pb2[c]); }
    if (hb && p.dcbias) atomic_add_f(p.dcbias + c, pcb[c]);
    if (hs && p.dcscale) atomic_add_f(p.dcscale + c, pcs[c]);
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) atomic_add_f(p.dw + i, pw[i]);
}


// Backward with NO LDS atomics.  Every phase uses one mapping -- wave w owns rows w, w+4, ...; lane l owns channels
// l, l+64, ... (CP of them) -- so each parameter-gradient partial (LN gammas/betas, conv bias/scale, the 9 taps) is a
// register of the thread that owns the channel, summed over the rows and images the thread visits, and folded across
// the 4 waves through LDS once at the end.  (The first version issued 13 same-address LDS atomics per element.)
template <typename T, int CP, int NW, int HS, int WS>
__global__ __launch_bounds__(64 * NW) void ccf_bwd2_kernel(qavit_ccf_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hs = HS ? HS : p.Hs, Ws = WS ? WS : p.Ws;      // HS / WS != 0: the map size is a compile-time constant (4 x 4 learned tokens)
  const int N = Hs * Ws, C = p.C;
  float* hin = sm;                 // [N][C] raw input h (LN1 input)
  float* a = hin + N * C;          // [N][C] LN1 output
  float* raw = a + N * C;          // [N][C] conv (+bias) before scale; later d(LN1 output), then dh
  float* t = raw + N * C;          // [N][C] LN2 input, then d(conv output)
  float* dsm = t + N * C;          // [N][C] the incoming gradient d_out
  float* st = dsm + N * C;         // [4][N] mean1, rstd1, mean2, rstd2 of the image's rows
  const T* h = reinterpret_cast<const T*>(p.h);
  const T* dout = reinterpret_cast<const T*>(p.d_out);
  T* dh = reinterpret_cast<T*>(p.d_h);
  const bool ln = p.flags & F_LN, hb = p.flags & F_BIAS, hs = p.flags & F_SCALE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float invC = 1.f / (float)C;
  float rg1[CP], rb1[CP], rg2[CP], rb2[CP], rcb[CP], rcs[CP], rw[CP][9], wt[CP][9], g1v[CP], g2v[CP], csv[CP], b1v[CP], cbv[CP];
#pragma unroll
  for (int k = 0; k < CP; ++k) {
    const int c = lane + 64 * k;
    const bool ok = c < C;
    rg1[k] = rb1[k] = rg2[k] = rb2[k] = rcb[k] = rcs[k] = 0.f;
    g1v[k] = (ok && ln) ? p.g1[c] : 0.f;
    g2v[k] = (ok && ln) ? p.g2[c] : 0.f;
    b1v[k] = (ok && ln) ? p.b1[c] : 0.f;
    cbv[k] = (ok && hb) ? p.cbias[c] : 0.f;
    csv[k] = (ok && hs) ? p.cscale[c] : 1.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) { rw[k][q] = 0.f; wt[k][q] = ok ? p.w[c * 9 + q] : 0.f; }
  }
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    __syncthreads();
    // Everything the image needs from memory is requested HERE, in one round trip: its h and d_out tiles (16-byte / 8-byte vectors) and
    // the saved statistics of its rows.  Fetched where they are used -- statistics at the top of each row iteration of three loops, a
    // d_out row inside the LayerNorm-backward loop -- they were five or six dependent round trips per image on a workgroup whose waves
    // have nothing else to run.
    {
      typedef typename V4<T>::type v4;
      const v4* hv = reinterpret_cast<const v4*>(h + (size_t)b * N * C);
      const v4* dv = reinterpret_cast<const v4*>(dout + (size_t)b * N * C);
      const int nv = N * C / 4;
      for (int base = threadIdx.x; base < nv; base += 2 * 64 * NW) {      // two pieces of each tile per thread in flight before the first store
        v4 x4[2], d4[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int i = base + j * 64 * NW; if (i < nv) { x4[j] = hv[i]; d4[j] = dv[i]; } }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = base + j * 64 * NW;
          if (i < nv) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { hin[4 * i + q] = to_f<T>(x4[j][q]); dsm[4 * i + q] = to_f<T>(d4[j][q]); }
          }
        }
      }
      if (ln)
        for (int i = threadIdx.x; i < N; i += 64 * NW) {
          st[i] = p.mean1[(size_t)b * N + i]; st[N + i] = p.rstd1[(size_t)b * N + i];
          st[2 * N + i] = p.mean2[(size_t)b * N + i]; st[3 * N + i] = p.rstd2[(size_t)b * N + i];
        }
    }
    __syncthreads();
    // recompute of the forward on the same (wave = rows, lane = channels) mapping with the parameters in registers: the saved
    // statistics make LN1 elementwise, and the stencil reads one channel of the neighbour rows.  (The generic helpers above
    // fetch every tap weight from global memory inside the tap loop and divide to find (row, channel): 3/4 of this kernel's time.)
    for (int r = wave; r < N; r += NW) {
      float mu = 0.f, rs = 1.f;
      if (ln) { mu = st[r]; rs = st[N + r]; }
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        if (c < C) { const float v = hin[r * C + c]; a[r * C + c] = ln ? (v - mu) * rs * g1v[k] + b1v[k] : v; }
      }
    }
    __syncthreads();
    for (int r = wave; r < N; r += NW) {
      const int y = r / Ws, x = r - y * Ws;
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          float sacc = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int yy = y + dy - 1, xx = x + dx - 1;
              if (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) sacc += wt[k][dy * 3 + dx] * a[(yy * Ws + xx) * C + c];
            }
          sacc += cbv[k];
          raw[r * C + c] = sacc;
          t[r * C + c] = sacc * csv[k];
        }
      }
    }
    wave_sync();                                        // LN2 backward below reads only the rows this wave just wrote
    // LN2 backward, then d(conv output) = dt * scale, with dscale / dbias partials -- one pass over the owned elements
    for (int r = wave; r < N; r += NW) {
      const float* gr = dsm + r * C;
      float* tr = t + r * C;
      const float* rr = raw + r * C;
      float d[CP], xh[CP];
      float c1 = 0.f, c2 = 0.f;
      float mu = 0.f, rs = 1.f;
      if (ln) { mu = st[2 * N + r]; rs = st[3 * N + r]; }
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        d[k] = 0.f; xh[k] = 0.f;
        if (c < C) {
          d[k] = gr[c];
          xh[k] = (tr[c] - mu) * rs;
          const float gg = d[k] * g2v[k];
          c1 += gg * xh[k]; c2 += gg;
        }
      }
      if (ln) { c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC; }
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          float dt = d[k];
          if (ln) {
            rg2[k] += d[k] * xh[k];
            rb2[k] += d[k];
            dt = rs * (d[k] * g2v[k] - c2 - xh[k] * c1);
          }
          if (hs) { rcs[k] += dt * rr[c]; dt *= csv[k]; }
          rcb[k] += dt;
          tr[c] = dt;
        }
      }
    }
    __syncthreads();
    // conv backward: tap partials in registers, d(LN1 output) into raw
    for (int r = wave; r < N; r += NW) {
      const int y = r / Ws, x = r - y * Ws;
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          const float dc = t[r * C + c];
          float da = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int yy = y + dy - 1, xx = x + dx - 1;           // forward tap read by output (y,x)
              if (yy >= 0 && yy < Hs && xx >= 0 && xx < Ws) rw[k][dy * 3 + dx] += dc * a[(yy * Ws + xx) * C + c];
              const int yo = y - dy + 1, xo = x - dx + 1;           // outputs that read input (y,x) through tap (dy,dx)
              if (yo >= 0 && yo < Hs && xo >= 0 && xo < Ws) da += wt[k][dy * 3 + dx] * t[(yo * Ws + xo) * C + c];
            }
          raw[r * C + c] = da;
        }
      }
    }
    wave_sync();                                        // LN1 backward reads only the rows this wave just wrote
    for (int r = wave; r < N; r += NW) {
      const float* xr = hin + r * C;
      const float* gr = raw + r * C;
      float d[CP], xh[CP];
      float c1 = 0.f, c2 = 0.f, mu = 0.f, rs = 1.f;
      if (ln) { mu = st[r]; rs = st[N + r]; }
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        d[k] = 0.f; xh[k] = 0.f;
        if (c < C) {
          d[k] = gr[c];
          xh[k] = (xr[c] - mu) * rs;
          const float gg = d[k] * g1v[k];
          c1 += gg * xh[k]; c2 += gg;
        }
      }
      if (ln) { c1 = wave_sum(c1) * invC; c2 = wave_sum(c2) * invC; }
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          float dx = d[k];
          if (ln) {
            rg1[k] += d[k] * xh[k];
            rb1[k] += d[k];
            dx = rs * (d[k] * g1v[k] - c2 - xh[k] * c1);
          }
          dh[((size_t)b * N + r) * C + c] = from_f<T>(dx);
        }
      }
    }
  }
  // fold the NW waves: part[wave][15][C] over the (now dead) image buffers, then one atomic per parameter element
  __syncthreads();
  float* part = sm;
#pragma unroll
  for (int k = 0; k < CP; ++k) {
    const int c = lane + 64 * k;
    if (c < C) {
      float* pw_ = part + (size_t)wave * 15 * C;
      pw_[0 * C + c] = rg1[k]; pw_[1 * C + c] = rb1[k]; pw_[2 * C + c] = rg2[k]; pw_[3 * C + c] = rb2[k];
      pw_[4 * C + c] = rcb[k]; pw_[5 * C + c] = rcs[k];
#pragma unroll
      for (int q = 0; q < 9; ++q) pw_[6 * C + c * 9 + q] = rw[k][q];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 15 * C; i += 64 * NW) {
    float v = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) v += part[w_ * 15 * C + i];
    if (p.parts) {                                    // one row of partial sums per workgroup; 512-way same-address atomics otherwise
      const int which = i / C;
      float keep = v;
      if (which < 4 && !ln) keep = 0.f;
      p.parts[(size_t)blockIdx.x * 15 * C + i] = keep;
      continue;
    }
    const int which = i / C;
    if (which >= 6) { atomic_add_f(p.dw + (i - 6 * C), v); continue; }
    const int c = i - which * C;
    if (which < 4) {
      if (!ln) continue;
      float* dst = which == 0 ? p.dg1 : which == 1 ? p.db1 : which == 2 ? p.dg2 : p.db2;
      atomic_add_f(dst + c, v);
    } else if (which == 4) { if (hb && p.dcbias) atomic_add_f(p.dcbias + c, v); }
    else { if (hs && p.dcscale) atomic_add_f(p.dcscale + c, v); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 4 x 4 maps (the 16 learned tokens of every C100 block), C <= 128 even: ONE WAVE PER IMAGE, EVERYTHING IN REGISTERS.  Lane l owns the
// channel pair (2l, 2l + 1) at all 16 positions: the depthwise 3 x 3 stencil and its backward are per-lane arithmetic on 32 registers (no
// LDS, no barrier), a LayerNorm row is a wave sum over the lanes, and the parameter-gradient partials are registers of the lane that owns
// the channel.  The LDS kernels above walk the image five times through fp32 LDS tiles with a barrier between the phases: 37 us backward and
// 16 us forward for 3 MB of activations at B = 1024 (one image per SIMD), VALU- and LDS-issue bound.
template <typename T> struct V2;
template <> struct V2<float> { typedef f32x2 type; };
template <> struct V2<bf16> { typedef __attribute__((ext_vector_type(2))) __bf16 type; };

template <typename T>
__device__ __forceinline__ void ccf3_load_rows(const T* src, int C, int c0, bool act, float (&v)[16][2]) {
  typedef typename V2<T>::type v2;
  v2 raw[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) raw[r] = *reinterpret_cast<const v2*>(src + (size_t)r * C + c0);      // inactive lanes read the pair at c0 = 0
#pragma unroll
  for (int r = 0; r < 16; ++r) { v[r][0] = act ? to_f<T>(raw[r][0]) : 0.f; v[r][1] = act ? to_f<T>(raw[r][1]) : 0.f; }
}

struct Ccf3Par { float g1[2], b1[2], g2[2], b2[2], cb[2], cs[2], wt[2][9]; };
__device__ __forceinline__ void ccf3_params(const qavit_ccf_args& p, int c0, bool act, Ccf3Par& q) {
  const bool ln = p.flags & F_LN, hb = p.flags & F_BIAS, hs = p.flags & F_SCALE;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = c0 + k;
    q.g1[k] = (act && ln) ? p.g1[c] : 0.f; q.b1[k] = (act && ln) ? p.b1[c] : 0.f;
    q.g2[k] = (act && ln) ? p.g2[c] : 0.f; q.b2[k] = (act && ln) ? p.b2[c] : 0.f;
    q.cb[k] = (act && hb) ? p.cbias[c] : 0.f;
    q.cs[k] = (act && hs) ? p.cscale[c] : (act ? 1.f : 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) q.wt[k][t] = act ? p.w[c * 9 + t] : 0.f;
  }
}
// conv(+bias) of the 4 x 4 map held per lane: raw[r][k] = sum_taps wt * a[neighbour] + cb
__device__ __forceinline__ void ccf3_conv(const float (&a)[16][2], const Ccf3Par& q, float (&raw)[16][2]) {
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float sacc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            if (yy >= 0 && yy < 4 && xx >= 0 && xx < 4) sacc += q.wt[k][dy * 3 + dx] * a[yy * 4 + xx][k];
          }
        raw[y * 4 + x][k] = sacc + q.cb[k];
      }
}

template <typename T>
__global__ __launch_bounds__(256) void ccf_fwd3_kernel(qavit_ccf_args p) {
  const int C = p.C, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool act = 2 * lane < C, ln = p.flags & F_LN;
  const int c0 = act ? 2 * lane : 0;
  const float invC = 1.f / (float)C;
  Ccf3Par q;
  ccf3_params(p, c0, act, q);
  const T* h = reinterpret_cast<const T*>(p.h);
  T* out = reinterpret_cast<T*>(p.out);
  typedef typename V2<T>::type v2;
  for (int b = blockIdx.x + (int)gridDim.x * wave; b < p.B; b += (int)gridDim.x * 4) {
    float a[16][2];
    ccf3_load_rows<T>(h + (size_t)b * 16 * C, C, c0, act, a);
    // No lane-dependent branch inside the row loops: a store under `if (lane == 0)` per row cuts the unrolled loop into sixteen basic blocks and
    // the rows' wave sums -- independent chains -- then run one after the other.  Row r's statistics are parked in lane r and stored once.
    float m1 = 0.f, s1 = 0.f, m2 = 0.f, s2 = 0.f;
    if (ln) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float mean = wave_sum(a[r][0] + a[r][1]) * invC;
        const float d0 = act ? a[r][0] - mean : 0.f, d1 = act ? a[r][1] - mean : 0.f;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * invC + p.eps);
        a[r][0] = act ? d0 * rstd * q.g1[0] + q.b1[0] : 0.f;
        a[r][1] = act ? d1 * rstd * q.g1[1] + q.b1[1] : 0.f;
        m1 = lane == r ? mean : m1; s1 = lane == r ? rstd : s1;
      }
    }
    float v[16][2];
    ccf3_conv(a, q, v);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v0 = v[r][0] * q.cs[0], v1 = v[r][1] * q.cs[1];
      if (ln) {
        const float mean = wave_sum(v0 + v1) * invC;
        const float d0 = act ? v0 - mean : 0.f, d1 = act ? v1 - mean : 0.f;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * invC + p.eps);
        v0 = d0 * rstd * q.g2[0] + q.b2[0]; v1 = d1 * rstd * q.g2[1] + q.b2[1];
        m2 = lane == r ? mean : m2; s2 = lane == r ? rstd : s2;
      }
      v[r][0] = v0; v[r][1] = v1;
    }
    if (act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { v2 o; o[0] = from_f<T>(v[r][0]); o[1] = from_f<T>(v[r][1]); *reinterpret_cast<v2*>(out + ((size_t)b * 16 + r) * C + c0) = o; }
    }
    if (ln && lane < 16) {
      p.mean1[(size_t)b * 16 + lane] = m1; p.rstd1[(size_t)b * 16 + lane] = s1;
      p.mean2[(size_t)b * 16 + lane] = m2; p.rstd2[(size_t)b * 16 + lane] = s2;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ccf_bwd3_kernel(qavit_ccf_args p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [4 waves][15 C]: the fold of the parameter-gradient partials
  const int C = p.C, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool act = 2 * lane < C, ln = p.flags & F_LN, hb = p.flags & F_BIAS, hs = p.flags & F_SCALE;
  const int c0 = act ? 2 * lane : 0;
  const float invC = 1.f / (float)C;
  Ccf3Par q;
  ccf3_params(p, c0, act, q);
  const T* h = reinterpret_cast<const T*>(p.h);
  const T* dout = reinterpret_cast<const T*>(p.d_out);
  T* dh = reinterpret_cast<T*>(p.d_h);
  typedef typename V2<T>::type v2;
  float rg1[2] = {0.f, 0.f}, rb1[2] = {0.f, 0.f}, rg2[2] = {0.f, 0.f}, rb2[2] = {0.f, 0.f}, rcb[2] = {0.f, 0.f}, rcs[2] = {0.f, 0.f}, rw[2][9];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int t = 0; t < 9; ++t) rw[k][t] = 0.f;
  for (int b = blockIdx.x + (int)gridDim.x * wave; b < p.B; b += (int)gridDim.x * 4) {
    float hv[16][2], d[16][2];
    ccf3_load_rows<T>(h + (size_t)b * 16 * C, C, c0, act, hv);
    ccf3_load_rows<T>(dout + (size_t)b * 16 * C, C, c0, act, d);
    // recompute of the forward from the saved row statistics (uniform per wave: scalar loads)
    float a[16][2], raw[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float mu = 0.f, rs = 1.f;
      if (ln) { mu = p.mean1[(size_t)b * 16 + r]; rs = p.rstd1[(size_t)b * 16 + r]; }
#pragma unroll
      for (int k = 0; k < 2; ++k) a[r][k] = !act ? 0.f : ln ? (hv[r][k] - mu) * rs * q.g1[k] + q.b1[k] : hv[r][k];
    }
    ccf3_conv(a, q, raw);
    // LayerNorm 2 backward, then d(conv output) = dt * scale with the dscale / dbias partials; d becomes dc
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float mu = 0.f, rs = 1.f;
      if (ln) { mu = p.mean2[(size_t)b * 16 + r]; rs = p.rstd2[(size_t)b * 16 + r]; }
      float xh[2], gg[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) { xh[k] = act ? (raw[r][k] * q.cs[k] - mu) * rs : 0.f; gg[k] = d[r][k] * q.g2[k]; }
      float c1 = 0.f, c2 = 0.f;
      if (ln) { c1 = wave_sum(gg[0] * xh[0] + gg[1] * xh[1]) * invC; c2 = wave_sum(gg[0] + gg[1]) * invC; }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float dt = d[r][k];
        if (ln) { rg2[k] += d[r][k] * xh[k]; rb2[k] += d[r][k]; dt = act ? rs * (gg[k] - c2 - xh[k] * c1) : 0.f; }
        if (hs) { rcs[k] += dt * raw[r][k]; dt *= q.cs[k]; }
        rcb[k] += dt;
        d[r][k] = dt;
      }
    }
    // convolution backward: tap partials, d(LN1 output) into raw
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float dc = d[y * 4 + x][k];
          float da = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int yy = y + dy - 1, xx = x + dx - 1;           // forward tap read by output (y, x)
              if (yy >= 0 && yy < 4 && xx >= 0 && xx < 4) rw[k][dy * 3 + dx] += dc * a[yy * 4 + xx][k];
              const int yo = y - dy + 1, xo = x - dx + 1;           // outputs that read input (y, x) through tap (dy, dx)
              if (yo >= 0 && yo < 4 && xo >= 0 && xo < 4) da += q.wt[k][dy * 3 + dx] * d[yo * 4 + xo][k];
            }
          raw[y * 4 + x][k] = da;
        }
    // LayerNorm 1 backward and the store
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float mu = 0.f, rs = 1.f;
      if (ln) { mu = p.mean1[(size_t)b * 16 + r]; rs = p.rstd1[(size_t)b * 16 + r]; }
      float xh[2], gg[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) { xh[k] = act ? (hv[r][k] - mu) * rs : 0.f; gg[k] = raw[r][k] * q.g1[k]; }
      float c1 = 0.f, c2 = 0.f;
      if (ln) { c1 = wave_sum(gg[0] * xh[0] + gg[1] * xh[1]) * invC; c2 = wave_sum(gg[0] + gg[1]) * invC; }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float da = raw[r][k];
        if (ln) { rg1[k] += da * xh[k]; rb1[k] += da; raw[r][k] = rs * (gg[k] - c2 - xh[k] * c1); }
      }
    }
    if (act) {                                           // one branch for the sixteen stores, none inside the row loop (see ccf_fwd3_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) { v2 o; o[0] = from_f<T>(raw[r][0]); o[1] = from_f<T>(raw[r][1]); *reinterpret_cast<v2*>(dh + ((size_t)b * 16 + r) * C + c0) = o; }
    }
  }
  // fold the four waves (same row layout as ccf_bwd2_kernel: g1 | b1 | g2 | b2 | conv bias | conv scale | taps [c][9])
  if (act) {
    float* pw_ = sm + (size_t)wave * 15 * C;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = c0 + k;
      pw_[0 * C + c] = rg1[k]; pw_[1 * C + c] = rb1[k]; pw_[2 * C + c] = rg2[k]; pw_[3 * C + c] = rb2[k];
      pw_[4 * C + c] = rcb[k]; pw_[5 * C + c] = rcs[k];
#pragma unroll
      for (int t = 0; t < 9; ++t) pw_[6 * C + c * 9 + t] = rw[k][t];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 15 * C; i += 256) {
    const float v = (sm[i] + sm[15 * C + i]) + (sm[2 * 15 * C + i] + sm[3 * 15 * C + i]);
    const int which = i / C;
    if (p.parts) { p.parts[(size_t)blockIdx.x * 15 * C + i] = (which < 4 && !ln) ? 0.f : v; continue; }
    if (which >= 6) { atomic_add_f(p.dw + (i - 6 * C), v); continue; }
    const int c = i - which * C;
    if (which < 4) {
      if (!ln) continue;
      float* dst = which == 0 ? p.dg1 : which == 1 ? p.db1 : which == 2 ? p.dg2 : p.db2;
      atomic_add_f(dst + c, v);
    } else if (which == 4) { if (hb && p.dcbias) atomic_add_f(p.dcbias + c, v); }
    else { if (hs && p.dcscale) atomic_add_f(p.dcscale + c, v); }
  }
}

}  // namespace qv

using namespace qv;

static int ccf_validate(const qavit_ccf_args* a, bool bwd) {
  if (!a || !a->h || !a->w || a->B <= 0 || a->Hs <= 0 || a->Ws <= 0 || a->C <= 0) return set_error(QAVIT_EINVAL, "ccf_mid: bad arguments");
  if ((a->flags & F_LN) && (!a->g1 || !a->b1 || !a->g2 || !a->b2 || !a->mean1 || !a->rstd1 || !a->mean2 || !a->rstd2))
    return set_error(QAVIT_EINVAL, "ccf_mid: LayerNorm flag set but norm arguments missing");
  if ((a->flags & F_BIAS) && !a->cbias) return set_error(QAVIT_EINVAL, "ccf_mid: bias flag set but bias missing");
  if ((a->flags & F_SCALE) && !a->cscale) return set_error(QAVIT_EINVAL, "ccf_mid: scale flag set but scale missing");
  if (!bwd && !a->out) return set_error(QAVIT_EINVAL, "ccf_mid_fwd: null output");
  if (bwd && (!a->d_out || !a->d_h || !a->dw)) return set_error(QAVIT_EINVAL, "ccf_mid_bwd: null gradient buffers");
  if (bwd && (a->flags & F_LN) && (!a->dg1 || !a->db1 || !a->dg2 || !a->db2)) return set_error(QAVIT_EINVAL, "ccf_mid_bwd: null norm gradient buffers");
  return QAVIT_OK;
}

// the register kernels: 4 x 4 map, an even C <= 128, channel pairs loadable as one 4- / 8-byte vector
static bool ccf3_ok(const qavit_ccf_args* a, bool bwd) {
  if (!(a->Hs == 4 && a->Ws == 4 && a->C % 2 == 0 && a->C >= 2 && a->C <= 128 && (a->dtype == QAVIT_F32 || a->dtype == QAVIT_BF16))) return false;
  const uintptr_t al = a->dtype == QAVIT_F32 ? 7 : 3;
  uintptr_t bits = reinterpret_cast<uintptr_t>(a->h);
  bits |= bwd ? (reinterpret_cast<uintptr_t>(a->d_out) | reinterpret_cast<uintptr_t>(a->d_h)) : reinterpret_cast<uintptr_t>(a->out);
  return (bits & al) == 0;
}

extern "C" int qavit_ccf_mid_fwd(const qavit_ccf_args* a, void* stream) {
  int rc = ccf_validate(a, false);
  if (rc) return rc;
  const size_t smem = (size_t)2 * a->Hs * a->Ws * a->C * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "ccf_mid_fwd: image tile too large for LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = a->B < 2048 ? a->B : 2048;
  static const int reg3 = getenv("QAVIT_CCF_REG") ? atoi(getenv("QAVIT_CCF_REG")) : 1;
  if (reg3 && ccf3_ok(a, false)) {                         // 4 x 4 maps: a wave per image, registers only
    int g3 = (a->B + 3) / 4;
    if (g3 > 2048) g3 = 2048;
    if (a->dtype == QAVIT_F32) hipLaunchKernelGGL((ccf_fwd3_kernel<float>), dim3(g3), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL((ccf_fwd3_kernel<bf16>), dim3(g3), dim3(256), 0, st, *a);
    return check_launch("ccf_mid_fwd(reg)");
  }
  const int threads = a->Hs * a->Ws >= 64 ? 512 : 256;    // 64-token maps fill one CU per image: 8 waves share the rows
  static const int fwd2 = getenv("QAVIT_CCF_FWD2") ? atoi(getenv("QAVIT_CCF_FWD2")) : 1;
  const size_t vecb = a->dtype == QAVIT_F32 ? 16 : 8;      // the register-partial kernels stage images with 4-element vector loads
  if (fwd2 && a->C <= 256 && a->C % 4 == 0 && reinterpret_cast<uintptr_t>(a->h) % vecb == 0 && (a->dtype == QAVIT_F32 || a->dtype == QAVIT_BF16)) {
    const int cp = (a->C + 63) / 64;
#define CCFF(T_, CP_) { if (threads == 512) { \
                          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_fwd2_kernel<T_, CP_, 8, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                          hipLaunchKernelGGL((ccf_fwd2_kernel<T_, CP_, 8, 0, 0>), dim3(grid), dim3(512), smem, st, *a); \
                        } else if (a->Hs == 4 && a->Ws == 4 && smem <= 48 * 1024) { \
                          hipLaunchKernelGGL((ccf_fwd2_kernel<T_, CP_, 4, 4, 4>), dim3(grid), dim3(256), smem, st, *a); \
                        } else { \
                          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_fwd2_kernel<T_, CP_, 4, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                          hipLaunchKernelGGL((ccf_fwd2_kernel<T_, CP_, 4, 0, 0>), dim3(grid), dim3(256), smem, st, *a); } }
    if (a->dtype == QAVIT_F32) { if (cp == 1) CCFF(float, 1) else if (cp == 2) CCFF(float, 2) else if (cp == 3) CCFF(float, 3) else CCFF(float, 4) }
    else { if (cp == 1) CCFF(bf16, 1) else if (cp == 2) CCFF(bf16, 2) else if (cp == 3) CCFF(bf16, 3) else CCFF(bf16, 4) }
#undef CCFF
    return check_launch("ccf_mid_fwd");
  }
  if (a->dtype == QAVIT_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((ccf_fwd_kernel<float>), dim3(grid), dim3(threads), smem, st, *a);
  } else if (a->dtype == QAVIT_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_fwd_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((ccf_fwd_kernel<bf16>), dim3(grid), dim3(threads), smem, st, *a);
  } else return set_error(QAVIT_EINVAL, "ccf_mid_fwd: unknown dtype");
  return check_launch("ccf_mid_fwd");
}

extern "C" int qavit_ccf_bwd_parts(int B) {
  static const int cgrid = getenv("QAVIT_CCF_BWD_GRID") ? atoi(getenv("QAVIT_CCF_BWD_GRID")) : 512;
  return B < cgrid ? B : cgrid;
}

extern "C" int qavit_ccf_mid_bwd(const qavit_ccf_args* a, void* stream) {
  int rc = ccf_validate(a, true);
  if (rc) return rc;
  // four image tiles + the register-partial kernel's fifth (d_out) and its row statistics + the fold scratch
  const size_t smem = ((size_t)5 * a->Hs * a->Ws * a->C + 4 * (size_t)a->Hs * a->Ws + 15 * (size_t)a->C) * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "ccf_mid_bwd: image tile too large for LDS");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = qavit_ccf_bwd_parts(a->B);
  if (a->parts && !(a->C <= 256 && a->Hs * a->Ws >= 15 && a->C % 8 == 0 && (reinterpret_cast<uintptr_t>(a->parts) & 15) == 0))
    return set_error(QAVIT_EINVAL, "ccf_mid_bwd: parts needs C <= 256, C % 8 == 0, >= 15 tokens and a 16-byte aligned workspace");
  static const int reg3 = getenv("QAVIT_CCF_REG") ? atoi(getenv("QAVIT_CCF_REG")) : 1;
  if (reg3 && ccf3_ok(a, true) && (!a->parts || (reinterpret_cast<uintptr_t>(a->parts) & 15) == 0)) {
    const size_t sm3 = (size_t)4 * 15 * a->C * sizeof(float);
    if (a->dtype == QAVIT_F32) hipLaunchKernelGGL((ccf_bwd3_kernel<float>), dim3(grid), dim3(256), sm3, st, *a);
    else hipLaunchKernelGGL((ccf_bwd3_kernel<bf16>), dim3(grid), dim3(256), sm3, st, *a);
    return check_launch("ccf_mid_bwd(reg)");
  }
  const size_t vecb = a->dtype == QAVIT_F32 ? 16 : 8;
  if (a->C <= 256 && a->Hs * a->Ws >= 15 && a->C % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(a->h) | reinterpret_cast<uintptr_t>(a->d_out)) % vecb == 0) {   // register-partial kernel (its wave fold needs 60*C floats of the image buffers; vector image loads)
    const int cp = (a->C + 63) / 64;
#define CCF2(T_, CP_) { if (a->Hs * a->Ws >= 64) { \
                          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_bwd2_kernel<T_, CP_, 8, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                          hipLaunchKernelGGL((ccf_bwd2_kernel<T_, CP_, 8, 0, 0>), dim3(grid), dim3(512), smem, st, *a); \
                        } else if (a->Hs == 4 && a->Ws == 4 && smem <= 48 * 1024) { \
                          hipLaunchKernelGGL((ccf_bwd2_kernel<T_, CP_, 4, 4, 4>), dim3(grid), dim3(256), smem, st, *a); \
                        } else { \
                          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_bwd2_kernel<T_, CP_, 4, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                          hipLaunchKernelGGL((ccf_bwd2_kernel<T_, CP_, 4, 0, 0>), dim3(grid), dim3(256), smem, st, *a); } }
    if (a->dtype == QAVIT_F32) { if (cp == 1) CCF2(float, 1) else if (cp == 2) CCF2(float, 2) else if (cp == 3) CCF2(float, 3) else CCF2(float, 4) }
    else if (a->dtype == QAVIT_BF16) { if (cp == 1) CCF2(bf16, 1) else if (cp == 2) CCF2(bf16, 2) else if (cp == 3) CCF2(bf16, 3) else CCF2(bf16, 4) }
    else return set_error(QAVIT_EINVAL, "ccf_mid_bwd: unknown dtype");
#undef CCF2
    return check_launch("ccf_mid_bwd");
  }
  if (a->dtype == QAVIT_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((ccf_bwd_kernel<float>), dim3(grid), dim3(256), smem, st, *a);
  } else if (a->dtype == QAVIT_BF16) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ccf_bwd_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((ccf_bwd_kernel<bf16>), dim3(grid), dim3(256), smem, st, *a);
  } else return set_error(QAVIT_EINVAL, "ccf_mid_bwd: unknown dtype");
  return check_launch("ccf_mid_bwd");
}
