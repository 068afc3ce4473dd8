// Depthwise k x k convolution (stride 1, pad k/2) on channel-last tokens [B, H*W, C]: the dw3x3 / dw5x5 / dw7x7
// of the CNN lateral path (ConvNeXtBlock.dwconv HQAViT_CIFAR100.py:722, LMFAdapter.dwconv_3x3/_5x5 :811-812).
// HBM-bound: one pass over x (and dy); the image tile of 64 channels lives in LDS, each thread owns one
// channel column (lane = channel -> conflict-free LDS, coalesced HBM) and a strip of tokens.
// Weight / bias gradients are accumulated in registers over all images a workgroup visits, reduced through
// LDS, and flushed with one fp32 atomic per element per workgroup.
#include <type_traits>
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include <stdlib.h>

namespace qv {

constexpr int DW_CH = 64;   // channels per workgroup tile

template <typename T, int KS>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* x, const float* w, const float* bias, T* y, int B, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = H * W;
  float* xs = sm;                      // [N][DW_CH]
  float* ws = sm + N * DW_CH;          // [KS*KS][DW_CH]
  const int c0 = blockIdx.x * DW_CH;
  const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c = c0 + cl;
  const bool cok = c < C;
  for (int i = threadIdx.x; i < KS * KS * DW_CH; i += 256) {
    const int t = i / DW_CH, cc = i - t * DW_CH;
    ws[i] = (c0 + cc < C) ? w[(size_t)(c0 + cc) * KS * KS + t] : 0.f;
  }
  const float bv = (bias && cok) ? bias[c] : 0.f;
  constexpr int R = KS / 2;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    __syncthreads();
    for (int n = tg; n < N; n += 4) xs[n * DW_CH + cl] = cok ? to_f<T>(x[((size_t)b * N + n) * C + c]) : 0.f;
    __syncthreads();
    for (int n = tg; n < N; n += 4) {
      const int yy = n / W, xx = n - yy * W;
      float s = bv;
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        const int y2 = yy + dy - R;
        if (y2 < 0 || y2 >= H) continue;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
          const int x2 = xx + dx - R;
          if (x2 >= 0 && x2 < W) s += ws[(dy * KS + dx) * DW_CH + cl] * xs[(y2 * W + x2) * DW_CH + cl];
        }
      }
      if (cok) y[((size_t)b * N + n) * C + c] = from_f<T>(s);
    }
  }
}

template <typename T, int KS>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* dy, const T* x, const float* w, T* dx, float* dw, float* dbias,
                                                         int B, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = H * W;
  float* xs = sm;                       // [N][DW_CH]
  float* gs = xs + N * DW_CH;           // [N][DW_CH]  dy
  float* ws = gs + N * DW_CH;           // [KS*KS][DW_CH]
  float* red = ws + KS * KS * DW_CH;    // [4][DW_CH] cross-strip reduction scratch
  const int c0 = blockIdx.x * DW_CH;
  const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c = c0 + cl;
  const bool cok = c < C;
  for (int i = threadIdx.x; i < KS * KS * DW_CH; i += 256) {
    const int t = i / DW_CH, cc = i - t * DW_CH;
    ws[i] = (c0 + cc < C) ? w[(size_t)(c0 + cc) * KS * KS + t] : 0.f;
  }
  constexpr int R = KS / 2;
  float aw[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) aw[i] = 0.f;
  float ab = 0.f;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    __syncthreads();
    for (int n = tg; n < N; n += 4) {
      const size_t o = ((size_t)b * N + n) * C + c;
      xs[n * DW_CH + cl] = cok ? to_f<T>(x[o]) : 0.f;
      gs[n * DW_CH + cl] = cok ? to_f<T>(dy[o]) : 0.f;
    }
    __syncthreads();
    for (int n = tg; n < N; n += 4) {
      const int yy = n / W, xx = n - yy * W;
      const float g = gs[n * DW_CH + cl];
      ab += g;
      float s = 0.f;
#pragma unroll
      for (int dyy = 0; dyy < KS; ++dyy) {
#pragma unroll
        for (int dxx = 0; dxx < KS; ++dxx) {
          const int y2 = yy + dyy - R, x2 = xx + dxx - R;      // input read by output n through tap (dyy,dxx)
          if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) aw[dyy * KS + dxx] += g * xs[(y2 * W + x2) * DW_CH + cl];
          const int yo = yy - dyy + R, xo = xx - dxx + R;      // output that reads input n through tap (dyy,dxx)
          if (yo >= 0 && yo < H && xo >= 0 && xo < W) s += ws[(dyy * KS + dxx) * DW_CH + cl] * gs[(yo * W + xo) * DW_CH + cl];
        }
      }
      if (cok) dx[((size_t)b * N + n) * C + c] = from_f<T>(s);
    }
  }
  // reduce the 4 token strips, then one atomic per (channel, tap) per workgroup
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) {
    __syncthreads();
    red[tg * DW_CH + cl] = aw[i];
    __syncthreads();
    if (tg == 0 && cok) atomic_add_f(dw + (size_t)c * KS * KS + i, red[cl] + red[DW_CH + cl] + red[2 * DW_CH + cl] + red[3 * DW_CH + cl]);
  }
  if (dbias) {
    __syncthreads();
    red[tg * DW_CH + cl] = ab;
    __syncthreads();
    if (tg == 0 && cok) atomic_add_f(dbias + c, red[cl] + red[DW_CH + cl] + red[2 * DW_CH + cl] + red[3 * DW_CH + cl]);
  }
}

// ------------------------------------------------------------------------------------------------
// 8x8 feature maps (the CIFAR lateral path): the whole map of one channel fits a lane's registers.  One lane =
// one channel of one image; taps, rows and columns are compile-time loops, so every index is a register name and
// the border tests fold away -- no LDS traffic inside the tap loops.  4 waves of a workgroup take different
// images of the same 64 channels.
// ------------------------------------------------------------------------------------------------
template <typename T, int KS>
__global__ __launch_bounds__(256) void dwconv_fwd8_kernel(const T* x, const float* w, const float* bias, T* y, int B, int C) {
  constexpr int HW = 8, N = 64, R = KS / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;                          // clamped: loads stay unconditional
  float wt[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) wt[t] = w[(size_t)cc * KS * KS + t];
  const float bv = bias ? bias[cc] : 0.f;
  for (int b = blockIdx.y * 4 + wave; b < B; b += gridDim.y * 4) {
    float xi[N];
#pragma unroll
    for (int n = 0; n < N; ++n) xi[n] = to_f<T>(x[((size_t)b * N + n) * C + cc]);
#pragma unroll
    for (int yy = 0; yy < HW; ++yy) {
      float o[HW];
#pragma unroll
      for (int xx = 0; xx < HW; ++xx) {
        float s_ = bv;
#pragma unroll
        for (int dyy = 0; dyy < KS; ++dyy) {
          const int y2 = yy + dyy - R;
          if (y2 < 0 || y2 >= HW) continue;
#pragma unroll
          for (int dxx = 0; dxx < KS; ++dxx) {
            const int x2 = xx + dxx - R;
            if (x2 < 0 || x2 >= HW) continue;
            s_ += wt[dyy * KS + dxx] * xi[y2 * HW + x2];
          }
        }
        o[xx] = s_;
      }
      if (cok) {
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) y[((size_t)b * N + yy * HW + xx) * C + c] = from_f<T>(o[xx]);
      }
    }
  }
}

// Backward: dx first (needs dy + taps), then dw with x streamed one ROW at a time -- x row y2 meets dy row yy through
// tap row dyy = y2 - yy + R -- so only 8 values of x are live next to dy[64] and the 49 tap sums: no scratch spills
// at 7x7.  The tap sums of the 4 waves are folded through LDS in ONE pass and leave as contiguous fp32 atomics
// (one [64 channels x taps] block per workgroup).
template <typename T, int KS>
__global__ __launch_bounds__(256) void dwconv_bwd8_kernel(const T* dy, const T* x, const float* w, T* dx, float* dw, float* dbias, int B, int C) {
  constexpr int HW = 8, N = 64, R = KS / 2, KK = KS * KS;
  __shared__ float red[4][64 * KK + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64, c = c0 + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;
  float aw[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) aw[t] = 0.f;
  float ab = 0.f;
  for (int b = blockIdx.y * 4 + wave; b < B; b += gridDim.y * 4) {
    float g[N];
#pragma unroll
    for (int n = 0; n < N; ++n) g[n] = to_f<T>(dy[((size_t)b * N + n) * C + cc]);
    {
      float wt[KK];
#pragma unroll
      for (int t = 0; t < KK; ++t) wt[t] = w[(size_t)cc * KK + t];      // L1/L2-resident; keeps 49 registers free in the dw phase
      // dx = full correlation of dy with the flipped taps
#pragma unroll
      for (int yy = 0; yy < HW; ++yy) {
        float o[HW];
#pragma unroll
        for (int xx = 0; xx < HW; ++xx) {
          float s_ = 0.f;
#pragma unroll
          for (int dyy = 0; dyy < KS; ++dyy) {
            const int yo = yy - dyy + R;
            if (yo < 0 || yo >= HW) continue;
#pragma unroll
            for (int dxx = 0; dxx < KS; ++dxx) {
              const int xo = xx - dxx + R;
              if (xo < 0 || xo >= HW) continue;
              s_ += wt[dyy * KS + dxx] * g[yo * HW + xo];
            }
          }
          o[xx] = s_;
        }
        if (cok) {
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) dx[((size_t)b * N + yy * HW + xx) * C + c] = from_f<T>(o[xx]);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) ab += g[n];
    // dw[dyy][dxx] += sum_{yy,xx} dy[yy][xx] * x[yy+dyy-R][xx+dxx-R], one x row at a time
#pragma unroll
    for (int y2 = 0; y2 < HW; ++y2) {
      float xr[HW];
#pragma unroll
      for (int xx = 0; xx < HW; ++xx) xr[xx] = to_f<T>(x[((size_t)b * N + y2 * HW + xx) * C + cc]);
#pragma unroll
      for (int yy = 0; yy < HW; ++yy) {
        const int dyy = y2 - yy + R;
        if (dyy < 0 || dyy >= KS) continue;
#pragma unroll
        for (int dxx = 0; dxx < KS; ++dxx) {
          float s_ = 0.f;
#pragma unroll
          for (int xx = 0; xx < HW; ++xx) {
            const int x2 = xx + dxx - R;
            if (x2 < 0 || x2 >= HW) continue;
            s_ += g[yy * HW + xx] * xr[x2];
          }
          aw[dyy * KS + dxx] += s_;
        }
      }
    }
  }
  // fold the 4 waves: red[wave][channel-in-chunk * KK + tap] mirrors dw's own layout for this 64-channel block
#pragma unroll
  for (int t = 0; t < KK; ++t) red[wave][lane * KK + t] = cok ? aw[t] : 0.f;
  red[wave][64 * KK + lane] = cok ? ab : 0.f;
  __syncthreads();
  const int live = (C - c0 < 64 ? C - c0 : 64);
  for (int i = threadIdx.x; i < live * KK; i += 256)
    atomic_add_f(dw + (size_t)c0 * KK + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  if (dbias && threadIdx.x < live) {
    const int i = 64 * KK + threadIdx.x;
    atomic_add_f(dbias + c0 + threadIdx.x, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// ------------------------------------------------------------------------------------------------
// Maps whose sides are multiples of 8 (Tiny-ImageNet: 16x16): the same lane-per-channel scheme on 8x8 OUTPUT tiles.
// A wave takes one (image, tile); the (8+2R)-wide halo rows are streamed one at a time -- input row r meets output row
// yy through tap row r - yy -- so 64 outputs, the taps and one halo row are all that is live.  Halo cells outside the
// map are clamped loads zeroed by a wave-uniform select.  FLIP correlates with the flipped taps (dx of the forward).
// ------------------------------------------------------------------------------------------------
template <typename T, int HR>
__device__ __forceinline__ void dw_halo_row(const T* src, float (&xr)[HR], int b, int gy, int x0, int H, int W, int ld, int cc) {
  const bool rok = gy >= 0 && gy < H;
  const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
#pragma unroll
  for (int ci = 0; ci < HR; ++ci) {
    const int gx = x0 + ci;
    const bool ok = rok && gx >= 0 && gx < W;
    const int gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    const float v = to_f<T>(src[(((size_t)b * H + gyc) * W + gxc) * ld + cc]);
    xr[ci] = ok ? v : 0.f;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The tap loops run on PAIRS of adjacent outputs: (o[xx], o[xx+1]) += w * (x[xx+d], x[xx+d+1]) is one v_pk_fma_f32 --
// two fp32 FMAs per lane per issue slot -- against the row kept as even-aligned and odd-aligned register pairs.
template <int HR>
struct RowPairs {
  f32x2 e[HR / 2];          // (x[2j], x[2j+1])
  f32x2 o[HR / 2];          // (x[2j+1], x[2j+2])   (last one's .y unused when HR is even)
  __device__ __forceinline__ void set(const float (&x)[HR]) {
#pragma unroll
    for (int j = 0; j < HR / 2; ++j) {
      e[j] = f32x2{x[2 * j], x[2 * j + 1]};
      o[j] = f32x2{x[2 * j + 1], (2 * j + 2 < HR) ? x[2 * j + 2] : 0.f};
    }
  }
  __device__ __forceinline__ f32x2 at(int k) const { return (k & 1) ? o[k >> 1] : e[k >> 1]; }     // (x[k], x[k+1]), k constant
};

// One halo row of a map that IS the 8x8 tile, from the tile held in registers (everything outside is zero; indices are constants)
template <int HR, int R>
__device__ __forceinline__ void dw_reg_row(const float (*pre)[8], float (&xr)[HR], int r) {
  const int rr = r - R;
#pragma unroll
  for (int ci = 0; ci < HR; ++ci) xr[ci] = (rr >= 0 && rr < 8 && ci >= R && ci < R + 8) ? pre[rr < 0 ? 0 : (rr > 7 ? 7 : rr)][ci >= R && ci < R + 8 ? ci - R : 0] : 0.f;
}

// PRE: the map is one 8x8 tile already in registers (``pre``): no loads here at all -- the caller requested the whole tile at once,
// one memory round trip per tile instead of one per streamed row.
template <typename T, int KS, bool FLIP, bool CAP = false, bool PRE = false>
__device__ __forceinline__ void dw_tile_rows(const T* src, const float (&wt)[KS * KS], f32x2 (&o)[8][4], int b, int y0, int x0, int H, int W, int ld, int cc,
                                             f32x2 (*cap)[4] = nullptr, const float (*pre)[8] = nullptr) {
  constexpr int TS = 8, R = KS / 2, HR = TS + 2 * R;
  float xr[HR], xn[HR];
  if constexpr (!PRE) dw_halo_row<T, HR>(src, xr, b, y0, x0, H, W, ld, cc);
#pragma unroll
  for (int r = 0; r < HR; ++r) {
    if constexpr (PRE) {
      dw_reg_row<HR, R>(pre, xr, r);
    } else {
      if (r + 1 < HR) dw_halo_row<T, HR>(src, xn, b, y0 + r + 1, x0, H, W, ld, cc);     // one row ahead, no further
      asm volatile("" ::: "memory");
    }
    RowPairs<HR> rp;
    rp.set(xr);
    if constexpr (CAP) {                                     // the tile's own cells pass by once: keep them (the caller's second use of src)
      if (r >= R && r < R + TS) {
#pragma unroll
        for (int p = 0; p < TS / 2; ++p) cap[r - R][p] = f32x2{xr[R + 2 * p], xr[R + 2 * p + 1]};
      }
    }
    const bool row_in = (y0 + r >= 0) && (y0 + r < H);       // rows outside the map are all zero: nothing to add
#pragma unroll
    for (int yy = 0; yy < TS; ++yy) {
      const int dyy = r - yy;
      if (dyy < 0 || dyy >= KS || !row_in) continue;
#pragma unroll
      for (int dxx = 0; dxx < KS; ++dxx) {
        const int tap = FLIP ? (KS - 1 - dyy) * KS + (KS - 1 - dxx) : dyy * KS + dxx;
        const f32x2 w2 = f32x2{wt[tap], wt[tap]};
#pragma unroll
        for (int p = 0; p < TS / 2; ++p) o[yy][p] = __builtin_elementwise_fma(w2, rp.at(2 * p + dxx), o[yy][p]);
      }
    }
    if constexpr (!PRE) {
#pragma unroll
      for (int ci = 0; ci < HR; ++ci) xr[ci] = xn[ci];
    } else {
      __builtin_amdgcn_sched_barrier(0);                     // rows in order: with every operand in registers the scheduler would interleave them all
    }
  }
}

// the 8x8 tile of channel cc of image b (rows ld elements apart): 64 loads in flight together
template <typename T>
__device__ __forceinline__ void dw_load_tile8(const T* src, float (&tl)[8][8], int b, int ld, int cc) {
  T raw[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) raw[i] = src[((size_t)b * 64 + i) * ld + cc];
#pragma unroll
  for (int i = 0; i < 64; ++i) tl[i >> 3][i & 7] = to_f<T>(raw[i]);
}

// ONE8: the map IS one 8x8 tile (H = W = 8 known at compile time: out-of-map rows / columns and their loads fold away)
template <typename T, int KS, bool ONE8>
__global__ __launch_bounds__(256) void dwconv_fwdt_kernel(const T* x, const float* w, const float* bias, T* y, int B, int H_, int W_, int C, int ldy,
                                                          T* xc, int ldc) {
  constexpr int TS = 8, R = KS / 2, KK = KS * KS;
  const int H = ONE8 ? 8 : H_, W = ONE8 ? 8 : W_;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x * 64 + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;
  float wt[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) wt[t] = w[(size_t)cc * KK + t];
  const float bv = bias ? bias[cc] : 0.f;
  const int TX = W / TS, TPI = (H / TS) * TX, units = B * TPI;
  for (int u = blockIdx.y * 4 + wave; u < units; u += gridDim.y * 4) {
    const int b = ONE8 ? u : u / TPI, t = u - b * TPI, ty = ONE8 ? 0 : t / TX, tx = ONE8 ? 0 : t - ty * TX;
    f32x2 o[TS][TS / 2];
#pragma unroll
    for (int yy = 0; yy < TS; ++yy)
#pragma unroll
      for (int p = 0; p < TS / 2; ++p) o[yy][p] = f32x2{bv, bv};
    if constexpr (ONE8) {
      float xt[8][8];
      dw_load_tile8<T>(x, xt, b, C, cc);
      dw_tile_rows<T, KS, false, false, true>(x, wt, o, b, -R, -R, 8, 8, C, cc, nullptr, xt);
    } else {
      dw_tile_rows<T, KS, false>(x, wt, o, b, ty * TS - R, tx * TS - R, H, W, C, cc);
    }
    if (cok) {
#pragma unroll
      for (int yy = 0; yy < TS; ++yy)
#pragma unroll
        for (int xx = 0; xx < TS; ++xx) y[(((size_t)b * H + ty * TS + yy) * W + tx * TS + xx) * ldy + c] = from_f<T>(o[yy][xx >> 1][xx & 1]);
    }
    if (xc && cok) {                                         // uniform: the tile's input cells once more, into another column slice (qavit_dwconv_fwd_ld2)
#pragma unroll
      for (int yy = 0; yy < TS; ++yy)
#pragma unroll
        for (int xx = 0; xx < TS; ++xx) {
          const size_t cell = ((size_t)b * H + ty * TS + yy) * W + tx * TS + xx;
          xc[cell * ldc + c] = x[cell * C + c];
        }
    }
  }
}

template <typename T, int KS, bool ONE8>
__global__ __launch_bounds__(256) void dwconv_bwdt_kernel(const T* dy, const T* x, const float* w, T* dx, float* dw, float* dbias, int B, int H_, int W_, int C,
                                                          int lddy, const T* dadd, int lddadd) {
  constexpr int TS = 8, R = KS / 2, KK = KS * KS, HR = TS + 2 * R;
  constexpr bool RT = ONE8 && KS <= 5;                      // both tiles of a unit in registers (7x7: 49 taps + 49 tap sums leave no room for
  constexpr bool RT1 = ONE8;                                //  the x tile: its rows are streamed; the dy tile is in registers for every k)
  const int H = ONE8 ? 8 : H_, W = ONE8 ? 8 : W_;
  __shared__ float red[4][64 * KK + 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = blockIdx.x * 64, c = c0 + lane;
  const bool cok = c < C;
  const int cc = cok ? c : 0;
  float aw[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) aw[t] = 0.f;
  float ab = 0.f;
  const int TX = W / TS, TPI = (H / TS) * TX, units = B * TPI;
  for (int u = blockIdx.y * 4 + wave; u < units; u += gridDim.y * 4) {
    const int b = ONE8 ? u : u / TPI, t = u - b * TPI, ty = ONE8 ? 0 : t / TX, tx = ONE8 ? 0 : t - ty * TX;
    f32x2 g[TS][TS / 2];
    float gt[RT1 ? 8 : 1][8], xt[RT ? 8 : 1][8];           // the dy and x tiles, every load of the unit requested up front
    if constexpr (RT1) dw_load_tile8<T>(dy, gt, b, lddy, cc);
    if constexpr (RT) dw_load_tile8<T>(x, xt, b, C, cc);
    {
      // the addend's 64 cells are requested before the tap arithmetic when the registers allow (k <= 5), after it otherwise
      constexpr bool EARLY = KS <= 5;
      T da[EARLY ? TS * TS : 1];
      if (EARLY && dadd && cok) {
#pragma unroll
        for (int yy = 0; yy < TS; ++yy)
#pragma unroll
          for (int xx = 0; xx < TS; ++xx) da[EARLY ? yy * TS + xx : 0] = dadd[(((size_t)b * H + ty * TS + yy) * W + tx * TS + xx) * lddadd + c];
      }
      float wt[KK];
#pragma unroll
      for (int i = 0; i < KK; ++i) wt[i] = w[(size_t)cc * KK + i];
      f32x2 o[TS][TS / 2];
#pragma unroll
      for (int yy = 0; yy < TS; ++yy)
#pragma unroll
        for (int p = 0; p < TS / 2; ++p) o[yy][p] = f32x2{0.f, 0.f};
      if constexpr (RT1) {
        dw_tile_rows<T, KS, true, false, true>(dy, wt, o, b, -R, -R, 8, 8, lddy, cc, nullptr, gt);
#pragma unroll
        for (int yy = 0; yy < TS; ++yy)
#pragma unroll
          for (int p = 0; p < TS / 2; ++p) g[yy][p] = f32x2{gt[yy][2 * p], gt[yy][2 * p + 1]};
      } else {
        dw_tile_rows<T, KS, true, true>(dy, wt, o, b, ty * TS - R, tx * TS - R, H, W, lddy, cc, g);
      }
      if (cok) {
        if (dadd) {       // another gradient that meets this one at x (may be dx itself: every lane reads exactly the cells it then writes)
#pragma unroll
          for (int yy = 0; yy < TS; ++yy)
#pragma unroll
            for (int xx = 0; xx < TS; ++xx)
              o[yy][xx >> 1][xx & 1] += to_f<T>(EARLY ? da[EARLY ? yy * TS + xx : 0] : dadd[(((size_t)b * H + ty * TS + yy) * W + tx * TS + xx) * lddadd + c]);
        }
#pragma unroll
        for (int yy = 0; yy < TS; ++yy)
#pragma unroll
          for (int xx = 0; xx < TS; ++xx) dx[(((size_t)b * H + ty * TS + yy) * W + tx * TS + xx) * C + c] = from_f<T>(o[yy][xx >> 1][xx & 1]);
      }
    }
#pragma unroll
    for (int yy = 0; yy < TS; ++yy)                          // g: the dy tile, kept from the stream above
#pragma unroll
      for (int p = 0; p < TS / 2; ++p) ab += g[yy][p][0] + g[yy][p][1];
    // dw[dyy][dxx] += sum_{yy,xx} dy[yy][xx] * x[yy+dyy-R][xx+dxx-R]: halo rows of x streamed against the dy tile
    float xr[HR], xn[HR];
    if constexpr (!RT) dw_halo_row<T, HR>(x, xr, b, ty * TS - R, tx * TS - R, H, W, C, cc);
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      if constexpr (RT) {
        dw_reg_row<HR, R>(xt, xr, r);
      } else {
        if (r + 1 < HR) dw_halo_row<T, HR>(x, xn, b, ty * TS - R + r + 1, tx * TS - R, H, W, C, cc);
        asm volatile("" ::: "memory");
      }
      RowPairs<HR> rp;
      rp.set(xr);
      const bool row_in = (ty * TS - R + r >= 0) && (ty * TS - R + r < H);
#pragma unroll
      for (int yy = 0; yy < TS; ++yy) {
        const int dyy = r - yy;
        if (dyy < 0 || dyy >= KS || !row_in) continue;
#pragma unroll
        for (int dxx = 0; dxx < KS; ++dxx) {
          f32x2 s2 = f32x2{0.f, 0.f};
#pragma unroll
          for (int p = 0; p < TS / 2; ++p) s2 = __builtin_elementwise_fma(g[yy][p], rp.at(2 * p + dxx), s2);
          aw[dyy * KS + dxx] += s2[0] + s2[1];
        }
      }
      if constexpr (!RT) {
#pragma unroll
        for (int ci = 0; ci < HR; ++ci) xr[ci] = xn[ci];
      } else {
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KK; ++t) red[wave][lane * KK + t] = cok ? aw[t] : 0.f;
  red[wave][64 * KK + lane] = cok ? ab : 0.f;
  __syncthreads();
  const int live = (C - c0 < 64 ? C - c0 : 64);
  for (int i = threadIdx.x; i < live * KK; i += 256)
    atomic_add_f(dw + (size_t)c0 * KK + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  if (dbias && threadIdx.x < live) {
    const int i = 64 * KK + threadIdx.x;
    atomic_add_f(dbias + c0 + threadIdx.x, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// Row strides (in elements) of the operands that may be column slices of a wider buffer: y forward, dy and the addend backward.
struct DwStrides { int ldy; int lddy; const void* dadd; int lddadd; void* xc = nullptr; int ldc = 0; };
static inline bool tiled_ok(int H, int W) { return H % 8 == 0 && W % 8 == 0; }

template <typename T, int KS>
static int launch_dw(bool bwd, const void* a0, const void* a1, const float* w, const float* bias, void* o0, float* dw, float* dbias,
                     int B, int H, int W, int C, hipStream_t st, const DwStrides& S) {
  const int N = H * W;
  const int ld0 = bwd ? S.lddy : S.ldy;                      // the strided operand: y forward, dy backward
  const bool plain = ld0 == C && !S.dadd && !S.xc;
  const int chunks = (C + DW_CH - 1) / DW_CH;
  static const int pk8 = getenv("QAVIT_DW8_PK") ? atoi(getenv("QAVIT_DW8_PK")) : 1;      // 8x8 maps on the packed-FMA tile kernels (0: the scalar-FMA dwconv_fwd8 / bwd8)
  if ((pk8 || !plain) && H == 8 && W == 8) {
    int gy = (B + 3) / 4;
    static const int w8 = getenv("QAVIT_DWT8_BWD_WGS") ? atoi(getenv("QAVIT_DWT8_BWD_WGS")) : 512;
    const int cap = (bwd ? w8 : 1024) / chunks > 0 ? (bwd ? w8 : 1024) / chunks : 1;
    if (gy > cap) gy = cap;
    if (!bwd) hipLaunchKernelGGL((dwconv_fwdt_kernel<T, KS, true>), dim3(chunks, gy), dim3(256), 0, st, (const T*)a0, w, bias, (T*)o0, B, H, W, C, S.ldy, (T*)S.xc, S.ldc);
    else hipLaunchKernelGGL((dwconv_bwdt_kernel<T, KS, true>), dim3(chunks, gy), dim3(256), 0, st, (const T*)a0, (const T*)a1, w, (T*)o0, dw, dbias, B, H, W, C,
                            S.lddy, (const T*)S.dadd, S.lddadd);
    return check_launch("dwconv8(pk)");
  }
  if (!plain && !(tiled_ok(H, W))) return set_error(QAVIT_EINVAL, "dwconv: row strides / addend need the tile kernels (sides multiples of 8)");
  if (!bwd && H == 8 && W == 8) {
    int gy8 = (B + 3) / 4;
    const int cap = 1024 / chunks > 0 ? 1024 / chunks : 1;
    if (gy8 > cap) gy8 = cap;
    hipLaunchKernelGGL((dwconv_fwd8_kernel<T, KS>), dim3(chunks, gy8), dim3(256), 0, st, (const T*)a0, w, bias, (T*)o0, B, C);
    return check_launch("dwconv_fwd8");
  }
  static const bool tiled = !(getenv("QAVIT_DW_TILED") && atoi(getenv("QAVIT_DW_TILED")) == 0);
  if ((tiled || !plain) && H % 8 == 0 && W % 8 == 0 && H * W > 64) {
    const int units = B * (H / 8) * (W / 8);
    int gy = (units + 3) / 4;
    static const int wt_ = getenv("QAVIT_DWT_BWD_WGS") ? atoi(getenv("QAVIT_DWT_BWD_WGS")) : 256;
    const int cap = (bwd ? wt_ : 2048) / chunks > 0 ? (bwd ? wt_ : 2048) / chunks : 1;
    if (gy > cap) gy = cap;
    if (!bwd) {
      hipLaunchKernelGGL((dwconv_fwdt_kernel<T, KS, false>), dim3(chunks, gy), dim3(256), 0, st, (const T*)a0, w, bias, (T*)o0, B, H, W, C, S.ldy, (T*)S.xc, S.ldc);
      return check_launch("dwconv_fwdt");
    }
    hipLaunchKernelGGL((dwconv_bwdt_kernel<T, KS, false>), dim3(chunks, gy), dim3(256), 0, st, (const T*)a0, (const T*)a1, w, (T*)o0, dw, dbias, B, H, W, C,
                       S.lddy, (const T*)S.dadd, S.lddadd);
    return check_launch("dwconv_bwdt");
  }
  if (!bwd) {
    const size_t smem = ((size_t)N * DW_CH + KS * KS * DW_CH) * sizeof(float);
    if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "dwconv_fwd: feature map too large for LDS");
    int gy = B < 2048 / chunks ? B : 2048 / chunks;
    if (gy < 1) gy = 1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_fwd_kernel<T, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((dwconv_fwd_kernel<T, KS>), dim3(chunks, gy), dim3(256), smem, st, (const T*)a0, w, bias, (T*)o0, B, H, W, C);
    return check_launch("dwconv_fwd");
  }
  if (H == 8 && W == 8) {
    int gy8 = (B + 3) / 4;
    static const int wgs8 = getenv("QAVIT_DW8_WGS") ? atoi(getenv("QAVIT_DW8_WGS")) : 256;
    const int cap = wgs8 / chunks > 0 ? wgs8 / chunks : 1;
    if (gy8 > cap) gy8 = cap;
    hipLaunchKernelGGL((dwconv_bwd8_kernel<T, KS>), dim3(chunks, gy8), dim3(256), 0, st, (const T*)a0, (const T*)a1, w, (T*)o0, dw, dbias, B, C);
    return check_launch("dwconv_bwd8");
  }
  const size_t smem = ((size_t)2 * N * DW_CH + KS * KS * DW_CH + 4 * DW_CH) * sizeof(float);
  if (smem > 160 * 1024) return set_error(QAVIT_EINVAL, "dwconv_bwd: feature map too large for LDS");
  int gy = B < 1024 / chunks ? B : 1024 / chunks;
  if (gy < 1) gy = 1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_bwd_kernel<T, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((dwconv_bwd_kernel<T, KS>), dim3(chunks, gy), dim3(256), smem, st, (const T*)a0, (const T*)a1, w, (T*)o0, dw, dbias, B, H, W, C);
  return check_launch("dwconv_bwd");
}

template <typename T>
static int dispatch_dw(int ks, bool bwd, const void* a0, const void* a1, const float* w, const float* bias, void* o0, float* dw, float* dbias,
                       int B, int H, int W, int C, hipStream_t st, const DwStrides& S) {
  switch (ks) {
    case 3: return launch_dw<T, 3>(bwd, a0, a1, w, bias, o0, dw, dbias, B, H, W, C, st, S);
    case 5: return launch_dw<T, 5>(bwd, a0, a1, w, bias, o0, dw, dbias, B, H, W, C, st, S);
    case 7: return launch_dw<T, 7>(bwd, a0, a1, w, bias, o0, dw, dbias, B, H, W, C, st, S);
    default: return set_error(QAVIT_EINVAL, "dwconv: kernel size must be 3, 5 or 7");
  }
}

// ------------------------------------------------------------------------------------------------
// im2col / col2im for the two strided 3x3 stem convolutions (HQAViT_CIFAR100.py:752, :759): the conv becomes a
// GEMM over rows (b, oy, ox) with K = Cin*k*k in the nn.Conv2d weight's own (c, dy, dx) order.
// ------------------------------------------------------------------------------------------------
template <typename T, bool NCHW_F32, typename I>
__global__ __launch_bounds__(256) void im2col_kernel(const void* src_, T* cols, int ld, int B, int Cin, int H, int W, int k, int stride, int pad, int Ho, int Wo) {
  const int Kc = Cin * k * k;
  const I total = (I)B * Ho * Wo * ld;                // rows of ld >= Kc elements; the pad columns are written as zeros
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const int col = (int)(i % (I)ld);
    const I row = i / (I)ld;
    if (col >= Kc) { cols[i] = from_f<T>(0.f); continue; }
    const int ox = (int)(row % (I)Wo), oy = (int)((row / (I)Wo) % (I)Ho), b = (int)(row / ((I)Wo * Ho));
    const int c = col / (k * k), r = col - c * k * k, dy = r / k, dx = r - dy * k;
    const int y = oy * stride + dy - pad, x = ox * stride + dx - pad;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      if (NCHW_F32) v = reinterpret_cast<const float*>(src_)[(((size_t)b * Cin + c) * H + y) * W + x];
      else v = to_f<T>(reinterpret_cast<const T*>(src_)[((size_t)b * H * W + y * W + x) * Cin + c]);
    }
    cols[i] = from_f<T>(v);
  }
}

// dx[b, y*W+x, c] = sum over taps of dcols[(b,oy,ox), c*k*k + dy*k + dx]  (channel-last destination)
template <typename T, typename I>
__global__ __launch_bounds__(256) void col2im_kernel(const T* dcols, T* dx, int B, int Cin, int H, int W, int k, int stride, int pad, int Ho, int Wo) {
  const int Kc = Cin * k * k;
  const I total = (I)B * H * W * Cin;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const int c = (int)(i % (I)Cin);
    const I pix = i / (I)Cin;
    const int x = (int)(pix % (I)W), y = (int)((pix / (I)W) % (I)H), b = (int)(pix / ((I)W * H));
    float s = 0.f;
    for (int dy = 0; dy < k; ++dy) {
      const int ty = y + pad - dy;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int dxx = 0; dxx < k; ++dxx) {
        const int tx = x + pad - dxx;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= Wo) continue;
        s += to_f<T>(dcols[(((size_t)b * Ho + oy) * Wo + ox) * Kc + c * k * k + dy * k + dxx]);
      }
    }
    dx[i] = from_f<T>(s);
  }
}

// ------------------------------------------------------------------------------------------------
// Per-image versions for channel-last sources: a workgroup stages the image's whole source tile (im2col) or its whole dcols block (col2im)
// in LDS with coalesced 16-byte loads -- all of a thread's loads in flight before its first LDS store -- and then produces 16-byte pieces
// of output rows from LDS.  The element-per-thread kernels above walk the (c, dy, dx) column order with 2-byte accesses 18 bytes apart:
// 62 us to write 37.7 MB of im2col rows and 74 us to fold them back at B = 1024, on the lateral path's critical tail.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Vec8;
template <> struct Vec8<bf16> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Vec8<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T, bool NCHW_F32>
__global__ __launch_bounds__(256) void im2col_img_kernel(const void* src_, T* cols, int ld, int Cin, int H, int W, int k, int stride, int pad, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  typedef typename std::conditional<NCHW_F32, float, T>::type S;        // source element: the fp32 NCHW image or channel-last tokens in T
  S* xs = reinterpret_cast<S*>(smraw);                         // the image as it lies in memory: [Cin][H][W] or [H*W][Cin]
  typedef typename Vec8<S>::type svec;
  typedef typename Vec8<T>::type vec;
  constexpr int SN = Vec8<S>::N, VN = Vec8<T>::N;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Kc = Cin * k * k, kk = k * k;
  const int nsrc = H * W * Cin / SN;
  const svec* sv = reinterpret_cast<const svec*>(reinterpret_cast<const S*>(src_) + (size_t)b * H * W * Cin);
  for (int base = tid; base < nsrc; base += 8 * 256) {
    svec r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = base + 256 * j; if (i < nsrc) r[j] = sv[i]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = base + 256 * j; if (i < nsrc) reinterpret_cast<svec*>(xs)[i] = r[j]; }
  }
  __syncthreads();
  const int ppr = ld / VN;                                     // pieces per output row
  const int npieces = Ho * Wo * ppr;
  T* out = cols + (size_t)b * Ho * Wo * ld;
  for (int p = tid; p < npieces; p += 256) {
    const int row = p / ppr, c0 = (p - row * ppr) * VN;
    const int oy = row / Wo, ox = row - oy * Wo;
    vec o;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int col = c0 + j;
      float v = 0.f;
      if (col < Kc) {
        const int c = col / kk, r = col - c * kk, dy = r / k, dx = r - dy * k;
        const int y = oy * stride + dy - pad, x = ox * stride + dx - pad;
        if (y >= 0 && y < H && x >= 0 && x < W) v = NCHW_F32 ? (float)xs[(c * H + y) * W + x] : to_f<S>(xs[(y * W + x) * Cin + c]);
      }
      o[j] = from_f<T>(v);
    }
    *reinterpret_cast<vec*>(out + (size_t)row * ld + c0) = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im_img_kernel(const T* dcols, T* dx, int Cin, int H, int W, int k, int stride, int pad, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  T* ds = reinterpret_cast<T*>(smraw);                         // [Ho*Wo][Kc]
  typedef typename Vec8<T>::type vec;
  constexpr int VN = Vec8<T>::N;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Kc = Cin * k * k, kk = k * k;
  const int nsrc = Ho * Wo * Kc / VN;
  const vec* sv = reinterpret_cast<const vec*>(dcols + (size_t)b * Ho * Wo * Kc);
  for (int base = tid; base < nsrc; base += 12 * 256) {
    vec r[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { const int i = base + 256 * j; if (i < nsrc) r[j] = sv[i]; }
#pragma unroll
    for (int j = 0; j < 12; ++j) { const int i = base + 256 * j; if (i < nsrc) reinterpret_cast<vec*>(ds)[i] = r[j]; }
  }
  __syncthreads();
  const int cpp = Cin / VN;                                    // pieces per pixel
  const int npieces = H * W * cpp;
  T* out = dx + (size_t)b * H * W * Cin;
  for (int p = tid; p < npieces; p += 256) {
    const int pix = p / cpp, c0 = (p - pix * cpp) * VN;
    const int y = pix / W, x = pix - y * W;
    float acc[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < k; ++dy) {
      const int ty = y + pad - dy;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int dxx = 0; dxx < k; ++dxx) {
        const int tx = x + pad - dxx;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= Wo) continue;
        const T* rowp = ds + (oy * Wo + ox) * Kc + dy * k + dxx;
#pragma unroll
        for (int j = 0; j < VN; ++j) acc[j] += to_f<T>(rowp[(c0 + j) * kk]);
      }
    }
    vec o;
#pragma unroll
    for (int j = 0; j < VN; ++j) o[j] = from_f<T>(acc[j]);
    *reinterpret_cast<vec*>(out + (size_t)pix * Cin + c0) = o;
  }
}

}  // namespace qv

using namespace qv;

extern "C" int qavit_im2col(int dtype, const void* src, int src_nchw_f32, void* cols, int B, int Cin, int H, int W, int k, int stride, int pad, void* stream) {
  return qavit_im2col_ld(dtype, src, src_nchw_f32, cols, Cin * k * k, B, Cin, H, W, k, stride, pad, stream);
}

extern "C" int qavit_im2col_ld(int dtype, const void* src, int src_nchw_f32, void* cols, int ld, int B, int Cin, int H, int W, int k, int stride, int pad, void* stream) {
  if (!src || !cols || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0 || ld < Cin * k * k) return set_error(QAVIT_EINVAL, "im2col: bad arguments");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int64_t total = (int64_t)B * Ho * Wo * ld;
  int grid = (int)((total + 1023) / 1024); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    // an image that fits LDS: the per-image kernel (coalesced staging, 16-byte row pieces)
    const size_t esz = dtype == QAVIT_F32 ? 4 : 2, ssz = src_nchw_f32 ? 4 : esz;
    const int vn = dtype == QAVIT_F32 ? 4 : 8, sn = ssz == 4 ? 4 : 8;
    const size_t smem = (size_t)H * W * Cin * ssz;
    if ((dtype == QAVIT_F32 || dtype == QAVIT_BF16) && smem <= 96 * 1024 && (H * W * Cin) % sn == 0 && ld % vn == 0 &&
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(cols)) & 15) == 0 && (smem & 15) == 0 && (((size_t)ld * esz) & 15) == 0) {
#define IM2IMG(T_, N_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(im2col_img_kernel<T_, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                            hipLaunchKernelGGL((im2col_img_kernel<T_, N_>), dim3(B), dim3(256), smem, st, src, (T_*)cols, ld, Cin, H, W, k, stride, pad, Ho, Wo); } while (0)
      if (dtype == QAVIT_F32) { if (src_nchw_f32) IM2IMG(float, true); else IM2IMG(float, false); }
      else { if (src_nchw_f32) IM2IMG(bf16, true); else IM2IMG(bf16, false); }
#undef IM2IMG
      return check_launch("im2col(image)");
    }
  }
  const bool small = total < 0x7fffffffLL && (int64_t)B * Cin * H * W < 0x7fffffffLL;
#define IM2COL(T_, N_) do { if (small) hipLaunchKernelGGL((im2col_kernel<T_, N_, uint32_t>), dim3(grid), dim3(256), 0, st, src, (T_*)cols, ld, B, Cin, H, W, k, stride, pad, Ho, Wo); \
                            else hipLaunchKernelGGL((im2col_kernel<T_, N_, int64_t>), dim3(grid), dim3(256), 0, st, src, (T_*)cols, ld, B, Cin, H, W, k, stride, pad, Ho, Wo); } while (0)
  if (dtype == QAVIT_F32) {
    if (src_nchw_f32) IM2COL(float, true); else IM2COL(float, false);
  } else if (dtype == QAVIT_BF16) {
    if (src_nchw_f32) IM2COL(bf16, true); else IM2COL(bf16, false);
  } else return set_error(QAVIT_EINVAL, "im2col: unknown dtype");
#undef IM2COL
  return check_launch("im2col");
}

extern "C" int qavit_col2im(int dtype, const void* dcols, void* dx, int B, int Cin, int H, int W, int k, int stride, int pad, void* stream) {
  if (!dcols || !dx || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0) return set_error(QAVIT_EINVAL, "col2im: bad arguments");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int64_t total = (int64_t)B * H * W * Cin;
  int grid = (int)((total + 1023) / 1024); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    const size_t esz = dtype == QAVIT_F32 ? 4 : 2;
    const int vn = dtype == QAVIT_F32 ? 4 : 8;
    const size_t smem = (size_t)Ho * Wo * Cin * k * k * esz;
    if ((dtype == QAVIT_F32 || dtype == QAVIT_BF16) && smem <= 128 * 1024 && Cin % vn == 0 && (Ho * Wo * Cin * k * k) % vn == 0 && (smem & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(dcols) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 && (((size_t)H * W * Cin * esz) & 15) == 0) {
      if (dtype == QAVIT_F32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(col2im_img_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipLaunchKernelGGL((col2im_img_kernel<float>), dim3(B), dim3(256), smem, st, (const float*)dcols, (float*)dx, Cin, H, W, k, stride, pad, Ho, Wo);
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(col2im_img_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipLaunchKernelGGL((col2im_img_kernel<bf16>), dim3(B), dim3(256), smem, st, (const bf16*)dcols, (bf16*)dx, Cin, H, W, k, stride, pad, Ho, Wo);
      }
      return check_launch("col2im(image)");
    }
  }
  const bool small = total < 0x7fffffffLL && (int64_t)B * Ho * Wo * Cin * k * k < 0x7fffffffLL;
  if (dtype == QAVIT_F32) {
    if (small) hipLaunchKernelGGL((col2im_kernel<float, uint32_t>), dim3(grid), dim3(256), 0, st, (const float*)dcols, (float*)dx, B, Cin, H, W, k, stride, pad, Ho, Wo);
    else hipLaunchKernelGGL((col2im_kernel<float, int64_t>), dim3(grid), dim3(256), 0, st, (const float*)dcols, (float*)dx, B, Cin, H, W, k, stride, pad, Ho, Wo);
  } else if (dtype == QAVIT_BF16) {
    if (small) hipLaunchKernelGGL((col2im_kernel<bf16, uint32_t>), dim3(grid), dim3(256), 0, st, (const bf16*)dcols, (bf16*)dx, B, Cin, H, W, k, stride, pad, Ho, Wo);
    else hipLaunchKernelGGL((col2im_kernel<bf16, int64_t>), dim3(grid), dim3(256), 0, st, (const bf16*)dcols, (bf16*)dx, B, Cin, H, W, k, stride, pad, Ho, Wo);
  } else return set_error(QAVIT_EINVAL, "col2im: unknown dtype");
  return check_launch("col2im");
}

extern "C" int qavit_dwconv_fwd_ld(int dtype, const void* x, const float* w, const float* bias, void* y, int ldy, int B, int H, int W, int C, int ks, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ldy < C) return set_error(QAVIT_EINVAL, "dwconv_fwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const DwStrides S{ldy, C, nullptr, 0};
  if (dtype == QAVIT_F32) return dispatch_dw<float>(ks, false, x, nullptr, w, bias, y, nullptr, nullptr, B, H, W, C, st, S);
  if (dtype == QAVIT_BF16) return dispatch_dw<bf16>(ks, false, x, nullptr, w, bias, y, nullptr, nullptr, B, H, W, C, st, S);
  return set_error(QAVIT_EINVAL, "dwconv_fwd: unknown dtype");
}

extern "C" int qavit_dwconv_fwd_ld2(int dtype, const void* x, const float* w, const float* bias, void* y, int ldy, void* xcopy, int ldc,
                                    int B, int H, int W, int C, int ks, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ldy < C || !xcopy || ldc < C) return set_error(QAVIT_EINVAL, "dwconv_fwd_ld2: bad arguments");
  if (!tiled_ok(H, W)) return set_error(QAVIT_EINVAL, "dwconv_fwd_ld2: map sides must be multiples of 8");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DwStrides S{ldy, C, nullptr, 0};
  S.xc = xcopy; S.ldc = ldc;
  if (dtype == QAVIT_F32) return dispatch_dw<float>(ks, false, x, nullptr, w, bias, y, nullptr, nullptr, B, H, W, C, st, S);
  if (dtype == QAVIT_BF16) return dispatch_dw<bf16>(ks, false, x, nullptr, w, bias, y, nullptr, nullptr, B, H, W, C, st, S);
  return set_error(QAVIT_EINVAL, "dwconv_fwd_ld2: unknown dtype");
}

extern "C" int qavit_dwconv_fwd(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int ks, void* stream) {
  return qavit_dwconv_fwd_ld(dtype, x, w, bias, y, C, B, H, W, C, ks, stream);
}

extern "C" int qavit_dwconv_bwd_ld(int dtype, const void* dy, int lddy, const void* x, const float* w, void* dx, const void* dadd, int lddadd,
                                   float* dw, float* dbias, int B, int H, int W, int C, int ks, void* stream) {
  if (!dy || !x || !w || !dx || !dw || B <= 0 || H <= 0 || W <= 0 || C <= 0 || lddy < C || (dadd && lddadd < C))
    return set_error(QAVIT_EINVAL, "dwconv_bwd: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const DwStrides S{C, lddy, dadd, dadd ? lddadd : 0};
  if (dtype == QAVIT_F32) return dispatch_dw<float>(ks, true, dy, x, w, nullptr, dx, dw, dbias, B, H, W, C, st, S);
  if (dtype == QAVIT_BF16) return dispatch_dw<bf16>(ks, true, dy, x, w, nullptr, dx, dw, dbias, B, H, W, C, st, S);
  return set_error(QAVIT_EINVAL, "dwconv_bwd: unknown dtype");
}

extern "C" int qavit_dwconv_bwd(int dtype, const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias,
                                int B, int H, int W, int C, int ks, void* stream) {
  return qavit_dwconv_bwd_ld(dtype, dy, C, x, w, dx, nullptr, 0, dw, dbias, B, H, W, C, ks, stream);
}
