// TokenLearner (HQAViT_CIFAR100.py:971-1002) as ONE launch each way:
//   scores = Linear(LayerNorm(x)) [N, M];  P = softmax over the N tokens;  xc = P^T x [M, C]
// The unfused chain read the [B*N, C] token matrix three times forward (row statistics + LayerNorm-prologue GEMM with 16 outputs,
// then the mixing kernel) and four times backward (mixing backward, the 16-deep input-gradient product inside the LayerNorm-backward
// launch, the deferred weight-gradient GEMM with LayerNorm-on-load), in launches whose own work is one memory round trip.  Here an
// image's N x C tile is read ONCE per direction: statistics, the normalised operand, both products and the softmax run on LDS tiles of
// the image (frag16.cuh: v_mfma_f32_16x16x16_bf16 on row-major bf16 tiles), and the backward also forms the score Linear's weight /
// bias gradient and the LayerNorm parameter gradients from the tiles it holds, leaving them as ONE partial row per workgroup for the
// pass's reduce launch (qavit_ln_param_reduce) -- the weight-gradient flush at the end of backward no longer re-reads the token matrix.
// Rounding points are those of the launches this replaces (bf16 LayerNorm output, bf16 scores, bf16 P, bf16 dscores).
#include "common.cuh"
#include "frag16.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include <stdlib.h>

#ifdef QAVIT_TL_STAMPS        // diagnostic build only (tools/tl_stamps.py): s_memtime at the phase boundaries of the backward, 40 words per workgroup
__device__ unsigned long long qv_tl_stamps[1024 * 40];
#define TLSTAMP(k) do { if (threadIdx.x == 0 && (k) < 40) qv_tl_stamps[(size_t)(blockIdx.x & 1023) * 40 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int qavit_tl_stamps(void* host_dst, int nwg) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(qv_tl_stamps), (size_t)nwg * 40 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#else
#define TLSTAMP(k) do { } while (0)
#endif

namespace qv {

namespace {

template <int NT_, int MT_, int CT_>
struct TlLds {
  static constexpr int NT = NT_, MT = MT_, CT = CT_;
  static constexpr int N = 16 * NT, M = 16 * MT, C = 16 * CT;
  static constexpr int LDC = C + 4, LDM = M + 4;             // bf16 row strides: 8-byte aligned rows (frag16.cuh)
  // bf16 element offsets of the tiles
  static constexpr int xs = 0;                               // [N][LDC] the image's tokens
  static constexpr int xn = xs + N * LDC;                    // [N][LDC] LayerNorm output, rounded to bf16 as the GEMM prologue rounded it
  static constexpr int wl = xn + N * LDC;                    // [M][LDC] score weight (compute-dtype copy)
  static constexpr int p = wl + M * LDC;                     // [N][LDM] probabilities
  static constexpr int fwd_end = p + N * LDM;
  static constexpr int dz = fwd_end;                         // [N][LDM] score gradients (backward)
  static constexpr int dxc = dz + N * LDM;                   // [M][LDC] incoming gradient tile (backward)
  static constexpr int bwd_end = dxc + M * LDC;
  static constexpr int NTW = (NT + 3) / 4;                   // row tiles per wave (wave w owns tiles w, w + 4, ...)
  static constexpr int TPW = (MT * CT + 3) / 4;              // (m-tile, c-tile) output tiles per wave
  static constexpr int CPL = C / 4;                          // channels per lane when 4 lanes share a row
  static constexpr int PR = M * C + M + 2 * C;               // floats per partial row of the backward: [dW | dbias | dgamma | dbeta]
  static constexpr size_t fwd_bytes = (size_t)((fwd_end + 7) / 8 * 8) * 2 + (size_t)(2 * C + 2 * 4 * M) * 4;
  static constexpr size_t bwd_bytes = (size_t)((bwd_end + 7) / 8 * 8) * 2 + (size_t)(2 * C + 4 * M + 4 * 2 * C) * 4;
};

// rows of one 16-row tile, in two halves so that the NEXT image's rows are in flight while the current image is worked on:
//   request: global -> registers (4 lanes per row, 16-byte pieces: a lane owns C / 4 consecutive channels of its row)
//   commit : registers -> row statistics (or the saved ones) -> the raw and the normalised LDS tiles
// Every lane of the wave takes part (group_sum<4> is a DPP reduction over the 4 lanes of a row).
template <class L>
struct TlRows { bf16x8 raw[L::CPL / 8]; float mean, rstd; };

template <class L, bool STATS_IN>
__device__ __forceinline__ void tl_rows_request(TlRows<L>& q, const bf16* xrow0, int nt, const float* mean_in, const float* rstd_in) {
  const int lane = threadIdx.x & 63, rr = lane >> 2, part = lane & 3;
  const int row = 16 * nt + rr;
  const bf16* src = xrow0 + (size_t)row * L::C + part * L::CPL;
#pragma unroll
  for (int j = 0; j < L::CPL / 8; ++j) q.raw[j] = *reinterpret_cast<const bf16x8*>(src + 8 * j);
  if (STATS_IN) { q.mean = mean_in[row]; q.rstd = rstd_in[row]; }
}

template <class L, bool STATS_IN>
__device__ __forceinline__ void tl_rows_commit(const TlRows<L>& q, bf16* sm, const float* prm, int nt, float eps, float* mean_out, float* rstd_out) {
  constexpr int CPL = L::CPL;
  const int lane = threadIdx.x & 63, rr = lane >> 2, part = lane & 3;
  const int row = 16 * nt + rr;
  float mean = q.mean, rstd = q.rstd;
  float v[CPL];
#pragma unroll
  for (int j = 0; j < CPL / 8; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[8 * j + e] = (float)q.raw[j][e];
  if (!STATS_IN) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s += v[j];
    mean = group_sum<4>(s) * (1.f / (float)L::C);
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { const float d = v[j] - mean; s2 += d * d; }
    rstd = rsqrtf(group_sum<4>(s2) * (1.f / (float)L::C) + eps);
    if (part == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  }
  bf16* dxs = sm + L::xs + row * L::LDC + part * CPL;
  bf16* dxn = sm + L::xn + row * L::LDC + part * CPL;
  const float* ga = prm + part * CPL;
  const float* be = prm + L::C + part * CPL;
#pragma unroll
  for (int j = 0; j < CPL; j += 4) {
    bf16x4 r4, n4;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(ga + j);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(be + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r4[e] = q.raw[(j + e) >> 3][(j + e) & 7];
      n4[e] = (bf16)((v[j + e] - mean) * rstd * g4[e] + b4[e]);
    }
    *reinterpret_cast<bf16x4*>(dxs + j) = r4;
    *reinterpret_cast<bf16x4*>(dxn + j) = n4;
  }
}

template <class L>
__device__ __forceinline__ void tl_stage_weight(bf16* sm, float* prm, const bf16* W, const float* g, const float* b) {
  constexpr int CH = L::C / 4;
  for (int i = threadIdx.x; i < L::M * CH; i += 256) {
    const int m = i / CH, ch = i - m * CH;
    *reinterpret_cast<bf16x4*>(sm + L::wl + m * L::LDC + 4 * ch) = *reinterpret_cast<const bf16x4*>(W + (size_t)m * L::C + 4 * ch);
  }
  for (int i = threadIdx.x; i < L::C; i += 256) { prm[i] = g[i]; prm[L::C + i] = b[i]; }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// forward.  Workgroup = 4 waves, images b = blockIdx.x, + gridDim.x, ...; wave w owns the row tiles w, w + 4, ...
// ------------------------------------------------------------------------------------------------------------------------------------
template <int NT, int MT, int CT>
__global__ __launch_bounds__(256) void tl_fwd_kernel(qavit_tl_args a) {
  using L = TlLds<NT, MT, CT>;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* prm = reinterpret_cast<float*>(smraw + (size_t)((L::fwd_end + 7) / 8 * 8) * 2);     // [2][C] LayerNorm gamma, beta
  float* red = prm + 2 * L::C;                                                                // [2][4][M]: column max / sum per wave
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, q4 = lane >> 4;
  const bf16* x = reinterpret_cast<const bf16*>(a.x);
  bf16* p_out = reinterpret_cast<bf16*>(a.p);
  bf16* xc = reinterpret_cast<bf16*>(a.xc);
  tl_stage_weight<L>(sm, prm, reinterpret_cast<const bf16*>(a.W), a.ln_g, a.ln_b);
  float bias_m[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bias_m[mt] = a.bias ? a.bias[16 * mt + col] : 0.f;
  __syncthreads();
  // the wave's rows of the NEXT image travel in registers while this image is worked on (one exposed memory round trip per workgroup,
  // not per image)
  TlRows<L> rq[L::NTW];
  auto request = [&](int b) {
#pragma unroll
    for (int i = 0; i < L::NTW; ++i) {
      const int nt = wave + 4 * i;
      if (nt < NT) tl_rows_request<L, false>(rq[i], x + (size_t)b * L::N * L::C, nt, nullptr, nullptr);
    }
  };
  if ((int)blockIdx.x < a.B) request(blockIdx.x);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
#pragma unroll
    for (int i = 0; i < L::NTW; ++i) {
      const int nt = wave + 4 * i;
      if (nt < NT) tl_rows_commit<L, false>(rq[i], sm, prm, nt, a.eps, a.mean + (size_t)b * L::N, a.rstd + (size_t)b * L::N);
    }
    if (b + (int)gridDim.x < a.B) request(b + gridDim.x);
    wave_sync();                                             // a wave's score tiles read its own rows only
    // ---- scores of the wave's row tiles: acc[r] = S[n = 16 nt + 4 q4 + r][m = 16 mt + col], rounded to bf16 as the GEMM stored them ----
    float sc[L::NTW][MT][4];
#pragma unroll
    for (int i = 0; i < L::NTW; ++i) {
      const int nt = wave + 4 * i;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (nt < NT) {
#pragma unroll
          for (int kt = 0; kt < CT; ++kt) acc = mma16(rowfrag(sm + L::xn, L::LDC, 16 * nt, 16 * kt), rowfrag(sm + L::wl, L::LDC, 16 * mt, 16 * kt), acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[i][mt][r] = nt < NT ? (float)(bf16)(acc[r] + bias_m[mt]) : -INFINITY;
      }
    }
    // ---- softmax over the N tokens of each column m: registers -> the 4 lanes of a column -> the 4 waves ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < L::NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[i][mt][r]);
      mx = rows4_max(mx);
      if (q4 == 0) red[wave * L::M + 16 * mt + col] = mx;
    }
    __syncthreads();
    float inv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = 16 * mt + col;
      const float gm = fmaxf(fmaxf(red[m], red[L::M + m]), fmaxf(red[2 * L::M + m], red[3 * L::M + m]));
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < L::NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(sc[i][mt][r] - gm); sc[i][mt][r] = e; s += e; }
      s = rows4_sum(s);
      if (q4 == 0) red[(4 + wave) * L::M + m] = s;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = 16 * mt + col;
      inv[mt] = 1.f / ((red[4 * L::M + m] + red[5 * L::M + m]) + (red[6 * L::M + m] + red[7 * L::M + m]));
    }
#pragma unroll
    for (int i = 0; i < L::NTW; ++i) {
      const int nt = wave + 4 * i;
      if (nt < NT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 16 * nt + 4 * q4 + r, m = 16 * mt + col;
            const bf16 pv = (bf16)(sc[i][mt][r] * inv[mt]);
            sm[L::p + n * L::LDM + m] = pv;
            p_out[((size_t)b * L::N + n) * L::M + m] = pv;
          }
      }
    }
    __syncthreads();                                         // P and every wave's token rows are in LDS
    // ---- xc = P^T x, operands swapped: acc[r] = xc[m = 16 mt + col][c = 16 ct + 4 q4 + r] -- one 8-byte row segment per lane ----
#pragma unroll
    for (int j = 0; j < L::TPW; ++j) {
      const int tile = wave + 4 * j;
      if (tile < MT * CT) {
        const int mt = tile / CT, ct = tile - mt * CT;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc = mma16(trfrag(sm + L::xs, L::LDC, nt * 16, ct * 16), trfrag(sm + L::p, L::LDM, nt * 16, mt * 16), acc);
        bf16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (bf16)acc[r];
        *reinterpret_cast<bf16x4*>(xc + ((size_t)b * L::M + 16 * mt + col) * L::C + 16 * ct + 4 * q4) = o4;
      }
    }
    __syncthreads();                                         // the tiles are free for the next image
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// backward.  dxc [B, M, C] -> dx [B, N, C] and this workgroup's partial row [dW (M x C) | dbias (M) | dgamma (C) | dbeta (C)]:
//   dP = x dxc^T;  dz = P (dP - sum_n P dP)  (softmax over tokens);  dxn = dz W;  dx = LayerNorm_backward(dxn) + P dxc
//   dW += dz^T LN(x);  dbias += colsum(dz);  dgamma += colsum(dxn * xhat);  dbeta += colsum(dxn)
// Row n of a wave's tile sits on the 4 lanes {col, col + 16, col + 32, col + 48} in the accumulators of the swapped products
// (acc[ct][r] = v[n = 16 nt + col][c = 16 ct + 4 q4 + r]), so the two row sums of the LayerNorm backward are in-lane sums + rows4_sum.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int NT, int MT, int CT>
__global__ __launch_bounds__(256) void tl_bwd_kernel(qavit_tl_bwd_args a) {
  using L = TlLds<NT, MT, CT>;
  static_assert(L::NTW == 1, "the backward holds one row tile per wave in registers (N = 64)");
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* prm = reinterpret_cast<float*>(smraw + (size_t)((L::bwd_end + 7) / 8 * 8) * 2);     // [2][C] LayerNorm gamma, beta
  float* red = prm + 2 * L::C;                                                                // [4][M]: per-wave column sums of P dP
  float* gred = red + 4 * L::M;                                                               // [4][2][C]: the final fold of dgamma / dbeta
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, q4 = lane >> 4;
  const bf16* x = reinterpret_cast<const bf16*>(a.x);
  const bf16* dxcg = reinterpret_cast<const bf16*>(a.dxc);
  const bf16* pg_ = reinterpret_cast<const bf16*>(a.p);
  bf16* dx = reinterpret_cast<bf16*>(a.dx);
  tl_stage_weight<L>(sm, prm, reinterpret_cast<const bf16*>(a.W), a.ln_g, a.ln_b);
  f32x4 dWacc[L::TPW];
#pragma unroll
  for (int j = 0; j < L::TPW; ++j) dWacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float pgam[CT][4], pbet[CT][4], dbacc[MT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { pgam[ct][r] = 0.f; pbet[ct][r] = 0.f; }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) dbacc[mt] = 0.f;
  const float invC = 1.f / (float)L::C;
  const int nt = wave;                                       // this wave's row tile
  TLSTAMP(0);
  // everything the NEXT image needs from memory -- the wave's token rows and their statistics, the thread's pieces of the dxc tile and of
  // P, the statistics of the lane's LayerNorm-backward row -- is requested while the current image is worked on: in-kernel stamps of the
  // first version put 38 % of the kernel into the staging section, i.e. into one exposed memory round trip per image
  constexpr int CH = L::C / 4, DXP = (L::M * CH + 255) / 256, PP = (L::N * L::M / 4 + 255) / 256;
  TlRows<L> rq;
  bf16x4 dq[DXP], pq[PP];
  float mu_q, rs_q;
  auto request = [&](int b) {
    const size_t r0 = (size_t)b * L::N;
    tl_rows_request<L, true>(rq, x + r0 * L::C, nt, a.mean + r0, a.rstd + r0);
    const bf16* src = dxcg + (size_t)b * L::M * L::C;
#pragma unroll
    for (int j = 0; j < DXP; ++j) {
      const int i = t + 256 * j, m = i / CH, ch = i - m * CH;
      if (i < L::M * CH) dq[j] = *reinterpret_cast<const bf16x4*>(src + (size_t)m * L::C + 4 * ch);
    }
    const bf16* ps = pg_ + r0 * L::M;
#pragma unroll
    for (int j = 0; j < PP; ++j) { const int i = t + 256 * j; if (i < L::N * L::M / 4) pq[j] = *reinterpret_cast<const bf16x4*>(ps + 4 * i); }
    mu_q = a.mean[r0 + 16 * nt + col];
    rs_q = a.rstd[r0 + 16 * nt + col];
  };
  if ((int)blockIdx.x < a.B) request(blockIdx.x);
  __syncthreads();
  TLSTAMP(1);
  int img_ = 0;
  for (int b = blockIdx.x; b < a.B; b += gridDim.x, ++img_) {
    const size_t row0 = (size_t)b * L::N;
    // ---- commit: this wave's token rows (raw + normalised), the dxc tile and P ----
    tl_rows_commit<L, true>(rq, sm, prm, nt, 0.f, nullptr, nullptr);
#pragma unroll
    for (int j = 0; j < DXP; ++j) {
      const int i = t + 256 * j, m = i / CH, ch = i - m * CH;
      if (i < L::M * CH) *reinterpret_cast<bf16x4*>(sm + L::dxc + m * L::LDC + 4 * ch) = dq[j];
    }
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int i = t + 256 * j, n = (4 * i) / L::M, m = 4 * i - n * L::M;
      if (i < L::N * L::M / 4) *reinterpret_cast<bf16x4*>(sm + L::p + n * L::LDM + m) = pq[j];
    }
    const float mu_n = mu_q, rs_n = rs_q;                    // the row statistics of the lane's LayerNorm-backward row (n = 16 nt + col)
    if (b + (int)gridDim.x < a.B) request(b + gridDim.x);
    TLSTAMP(2 + 8 * img_);
    __syncthreads();
    TLSTAMP(3 + 8 * img_);
    // ---- dP tile and the softmax backward: acc[r] = dP[n = 16 nt + 4 q4 + r][m = 16 mt + col] ----
    float dP[MT][4], pv[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc = mma16(rowfrag(sm + L::xs, L::LDC, 16 * nt, 16 * ct), rowfrag(sm + L::dxc, L::LDC, 16 * mt, 16 * ct), acc);
      float dot = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[mt][r] = (float)sm[L::p + (16 * nt + 4 * q4 + r) * L::LDM + 16 * mt + col];
        dP[mt][r] = acc[r];
        dot += pv[mt][r] * acc[r];
      }
      dot = rows4_sum(dot);
      if (q4 == 0) red[wave * L::M + 16 * mt + col] = dot;
    }
    TLSTAMP(4 + 8 * img_);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = 16 * mt + col;
      const float dot = (red[m] + red[L::M + m]) + (red[2 * L::M + m] + red[3 * L::M + m]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bf16 d = (bf16)(pv[mt][r] * (dP[mt][r] - dot));
        sm[L::dz + (16 * nt + 4 * q4 + r) * L::LDM + m] = d;
        dbacc[mt] += (float)d;
      }
    }
    TLSTAMP(5 + 8 * img_);
    __syncthreads();                                         // every wave's dz rows (the weight gradient contracts over all N)
    TLSTAMP(6 + 8 * img_);
    // ---- dxn = dz W (swapped: acc[ct][r] = dxn[n = 16 nt + col][c = 16 ct + 4 q4 + r]) and the LayerNorm backward of row n ----
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[ct] = mma16(trfrag(sm + L::wl, L::LDC, 16 * mt, 16 * ct), rowfrag(sm + L::dz, L::LDM, 16 * nt, 16 * mt), acc[ct]);
    }
    TLSTAMP(7 + 8 * img_);
    const bf16* xrow = sm + L::xs + (16 * nt + col) * L::LDC + 4 * q4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xrow + 16 * ct);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(prm + 16 * ct + 4 * q4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = ((float)xv[r] - mu_n) * rs_n, d = acc[ct][r], g = d * g4[r];
        pgam[ct][r] += d * xh;
        pbet[ct][r] += d;
        s1 += g * xh;
        s2 += g;
      }
    }
    s1 = rows4_sum(s1) * invC;
    s2 = rows4_sum(s2) * invC;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const bf16x4 xv = *reinterpret_cast<const bf16x4*>(xrow + 16 * ct);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(prm + 16 * ct + 4 * q4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = ((float)xv[r] - mu_n) * rs_n;
        acc[ct][r] = rs_n * (acc[ct][r] * g4[r] - s2 - xh * s1);
      }
      // + the mixing product's gradient of x, P dxc, accumulated onto the LayerNorm-backward tile by the MFMAs themselves
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[ct] = mma16(trfrag(sm + L::dxc, L::LDC, 16 * mt, 16 * ct), rowfrag(sm + L::p, L::LDM, 16 * nt, 16 * mt), acc[ct]);
      bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (bf16)acc[ct][r];
      *reinterpret_cast<bf16x4*>(dx + (row0 + 16 * nt + col) * L::C + 16 * ct + 4 * q4) = o4;
    }
    TLSTAMP(8 + 8 * img_);
    // ---- dW[m][c] += sum_n dz[n][m] xn[n][c]: acc[r] = dW[m = 16 mt + 4 q4 + r][c = 16 ct + col] ----
#pragma unroll
    for (int j = 0; j < L::TPW; ++j) {
      const int tile = wave + 4 * j;
      if (tile < MT * CT) {
        const int mt = tile / CT, ct = tile - mt * CT;
#pragma unroll
        for (int k = 0; k < NT; ++k) dWacc[j] = mma16(trfrag(sm + L::dz, L::LDM, 16 * k, 16 * mt), trfrag(sm + L::xn, L::LDC, 16 * k, 16 * ct), dWacc[j]);
      }
    }
    TLSTAMP(9 + 8 * img_);
    __syncthreads();                                         // the tiles are free for the next image
  }
  TLSTAMP(36);
  // ---- this workgroup's partial row (plain stores; folded with every other kernel's by the pass's reduce launch) ----
  float* prow = a.parts + (size_t)blockIdx.x * L::PR;
#pragma unroll
  for (int j = 0; j < L::TPW; ++j) {
    const int tile = wave + 4 * j;
    if (tile < MT * CT) {
      const int mt = tile / CT, ct = tile - mt * CT;
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[(size_t)(16 * mt + 4 * q4 + r) * L::C + 16 * ct + col] = dWacc[j][r];
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float v = rows4_sum(dbacc[mt]);
    if (q4 == 0) red[wave * L::M + 16 * mt + col] = v;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sg = grp_sum<16>(pgam[ct][r]), sb = grp_sum<16>(pbet[ct][r]);      // over the 16 rows (col lanes) of the wave's tile
      if (col == 0) { gred[(wave * 2 + 0) * L::C + 16 * ct + 4 * q4 + r] = sg; gred[(wave * 2 + 1) * L::C + 16 * ct + 4 * q4 + r] = sb; }
    }
  __syncthreads();
  for (int i = t; i < L::M; i += 256) prow[L::M * L::C + i] = (red[i] + red[L::M + i]) + (red[2 * L::M + i] + red[3 * L::M + i]);
  for (int i = t; i < 2 * L::C; i += 256) {
    const int which = i / L::C, c = i - which * L::C;
    prow[L::M * L::C + L::M + i] = (gred[(0 * 2 + which) * L::C + c] + gred[(1 * 2 + which) * L::C + c]) + (gred[(2 * 2 + which) * L::C + c] + gred[(3 * 2 + which) * L::C + c]);
  }
  TLSTAMP(37);
}

int tl_grid(int B, bool bwd) {
  static const int gf = getenv("QAVIT_TL_FWD_GRID") ? atoi(getenv("QAVIT_TL_FWD_GRID")) : 512;
  static const int gb = getenv("QAVIT_TL_BWD_GRID") ? atoi(getenv("QAVIT_TL_BWD_GRID")) : 256;
  const int g = bwd ? gb : gf;
  return B < g ? B : (g < 1 ? 1 : g);
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_tl_supported(int dtype, int N, int M, int C) { return dtype == QAVIT_BF16 && N == 64 && M == 16 && C == 192; }

extern "C" int qavit_tl_fwd(const qavit_tl_args* a, void* stream) {
  if (!a || !a->x || !a->ln_g || !a->ln_b || !a->W || !a->p || !a->xc || !a->mean || !a->rstd || a->B <= 0)
    return set_error(QAVIT_EINVAL, "tl_fwd: null operand");
  if (!qavit_tl_supported(QAVIT_BF16, a->N, a->M, a->C)) return set_error(QAVIT_EINVAL, "tl_fwd: bf16, N = 64, M = 16, C = 192 only (qavit_tl_supported)");
  if ((reinterpret_cast<uintptr_t>(a->x) & 15) || (reinterpret_cast<uintptr_t>(a->W) & 7) || (reinterpret_cast<uintptr_t>(a->xc) & 7))
    return set_error(QAVIT_EINVAL, "tl_fwd: x 16-byte aligned, W and xc 8-byte aligned");
  using L = TlLds<4, 1, 12>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tl_fwd_kernel<4, 1, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((tl_fwd_kernel<4, 1, 12>), dim3(tl_grid(a->B, false)), dim3(256), L::fwd_bytes, reinterpret_cast<hipStream_t>(stream), *a);
  return check_launch("tl_fwd");
}

extern "C" int qavit_tl_bwd_parts(int B, int N, int M) { (void)N; (void)M; return B > 0 ? tl_grid(B, true) : 0; }

extern "C" int qavit_tl_bwd(const qavit_tl_bwd_args* a, void* stream) {
  if (!a || !a->dxc || !a->x || !a->p || !a->mean || !a->rstd || !a->ln_g || !a->ln_b || !a->W || !a->dx || !a->parts || a->B <= 0)
    return set_error(QAVIT_EINVAL, "tl_bwd: null operand");
  if (!qavit_tl_supported(QAVIT_BF16, a->N, a->M, a->C)) return set_error(QAVIT_EINVAL, "tl_bwd: bf16, N = 64, M = 16, C = 192 only (qavit_tl_supported)");
  if ((reinterpret_cast<uintptr_t>(a->x) & 15) || (reinterpret_cast<uintptr_t>(a->W) & 7) || (reinterpret_cast<uintptr_t>(a->dxc) & 7) ||
      (reinterpret_cast<uintptr_t>(a->p) & 7) || (reinterpret_cast<uintptr_t>(a->dx) & 7) || (reinterpret_cast<uintptr_t>(a->parts) & 15))
    return set_error(QAVIT_EINVAL, "tl_bwd: x / parts 16-byte aligned, W, dxc, p, dx 8-byte aligned");
  using L = TlLds<4, 1, 12>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tl_bwd_kernel<4, 1, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((tl_bwd_kernel<4, 1, 12>), dim3(tl_grid(a->B, true)), dim3(256), L::bwd_bytes, reinterpret_cast<hipStream_t>(stream), *a);
  return check_launch("tl_bwd");
}
