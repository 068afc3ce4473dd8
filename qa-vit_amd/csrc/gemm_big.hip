// bf16 gemm_nt for the "fat" problems of the path (N*K too large for a resident weight slice: the CCF-FFN
// 256<->1024 linears, the patch / lateral projections, the fused qkv):  C[M,N] = epi( pro(A)[M,K] . B[N,K]^T + bias ).
//
// Classic K-loop tiling, sized for a CDNA4 CU: a workgroup owns a BM x BN output tile with BN = the whole N where
// it fits (<= 256), so every A element is read from HBM ONCE (the resident-slice kernel in gemm.hip re-reads A once
// per 32/64-column slice -- 8x for N = 256, and with a_mode 2 it also repeats the GELU'/dropout transform 8x).
// A and B stream through LDS in 64-wide K chunks with a register prefetch of the next chunk issued before the
// MFMAs of the current one.  4 waves form a 2 x 2 grid, each accumulating (BM/32) x (BN/32) 16x16 tiles of
// v_mfma_f32_16x16x32_bf16.  Column blocks of one row tile are placed on the same XCD (workgroup ids are dealt
// round-robin over the 8 XCDs) so their shared A rows hit that XCD's L2.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "gemm_shared.h"
#include <stdlib.h>

namespace qv {

namespace {

constexpr int BK = 64;
constexpr int LDT = BK + 8;   // 144-byte rows: 16-byte aligned, 8 consecutive rows cover all 32 banks

struct EpiKeys {
  uint32_t drop, dp;
  float inv_keep, dp_inv;
};

// 16 consecutive output columns of row m starting at column n (v holds the fp32 accumulators)
template <int EPI>
__device__ __forceinline__ void epilogue16(const qavit_gemm_args& g, const EpiKeys& ek, int m, int n, float (&v)[16]) {
  bf16* C = reinterpret_cast<bf16*>(g.C);
  const bf16* Rr = reinterpret_cast<const bf16*>(g.R);
  bf16* Zo = reinterpret_cast<bf16*>(g.Z);
  const int nv = (g.N - n < 16) ? (g.N - n) : 16;
  const bool full = (nv == 16) && (n % 8 == 0);
  if (g.bias) {
    if (nv == 16 && ((reinterpret_cast<uintptr_t>(g.bias + n) & 15) == 0)) {
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(g.bias + n + i);
        v[i] += t[0]; v[i + 1] += t[1]; v[i + 2] += t[2]; v[i + 3] += t[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) if (i < nv) v[i] += g.bias[n + i];
    }
  }
  float zv[16];
  if (EPI == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) zv[i] = v[i];
    if (g.act == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = gelu_f(v[i]);
    }
    if (g.drop_p > 0.f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] *= drop_factor(ek.drop, (uint32_t)m * (uint32_t)g.N + (uint32_t)(n + i), g.drop_p, ek.inv_keep);
    }
    float rowf = g.scale;
    if (g.dp_p > 0.f) rowf *= drop_factor(ek.dp, (uint32_t)(m / g.dp_rows), g.dp_p, ek.dp_inv);
    if (rowf != 1.f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] *= rowf;
    }
    if (Rr) {
      if (full && (g.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(Rr) & 15) == 0)) {
#pragma unroll
        for (int i = 0; i < 16; i += 8) {
          const bf16x8 t = *reinterpret_cast<const bf16x8*>(Rr + (size_t)m * g.ldr + n + i);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[i + j] += (float)t[j];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i < nv) v[i] += (float)Rr[(size_t)m * g.ldr + n + i];
      }
    }
  }
  bf16* crow = C + (size_t)m * g.ldc + n;
  if (full && (g.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0)) {
#pragma unroll
    for (int i = 0; i < 16; i += 8) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)v[i + j];
      *reinterpret_cast<bf16x8*>(crow + i) = o;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < nv) crow[i] = (bf16)v[i];
  }
  if (EPI == 1 && Zo) {
    bf16* zrow = Zo + (size_t)m * g.ldz + n;
    if (full && (g.ldz % 8 == 0) && ((reinterpret_cast<uintptr_t>(Zo) & 15) == 0)) {
#pragma unroll
      for (int i = 0; i < 16; i += 8) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)zv[i + j];
        *reinterpret_cast<bf16x8*>(zrow + i) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) if (i < nv) zrow[i] = (bf16)zv[i];
    }
  }
}

// Preconditions (checked by gemm_nt_big_try): K % 32 == 0; A / a_Z / a_out / B rows 16-byte aligned.
template <int BM_, int BN_, int AMODE, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_big_kernel(qavit_gemm_args g, int n_tiles_m, int ncb, int stats_in) {
  constexpr int TM = BM_ / 32, TN = BN_ / 32;       // 16x16 tiles per wave (rows, cols)
  constexpr int WN = BN_ / 2;                       // columns per wave
  constexpr int WLD = WN + 4;                       // epilogue scratch row stride (floats)
  constexpr int AV = BM_ * (BK / 8) / 256;          // 16-byte vectors per thread per chunk
  constexpr int BV = BN_ * (BK / 8) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* As = reinterpret_cast<bf16*>(smem);         // [BM][LDT]
  bf16* Bs = As + BM_ * LDT;                        // [BN][LDT]
  float* Gs = reinterpret_cast<float*>(Bs + BN_ * LDT);   // [2][K] LayerNorm gamma / beta (AMODE 1)

  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int cb = jj % ncb, tm = (jj / ncb) * 8 + xcd;
  if (tm >= n_tiles_m) return;                      // uniform per workgroup, before any barrier
  const int m0 = tm * BM_, n0 = cb * BN_;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fq = lane >> 4;

  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* A2 = AMODE == 0 ? reinterpret_cast<const bf16*>(g.A2) : nullptr;      // two-source A: plain prologue only
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  const bf16* Zin = reinterpret_cast<const bf16*>(g.a_Z);
  bf16* Aout = reinterpret_cast<bf16*>(g.a_out);
  constexpr bool bwd = AMODE == 2;
  const bool use_z = bwd && Zin && g.a_act;
  const bool write_aout = bwd && Aout && cb == 0;

  uint32_t key_adrop = 0, key_adp = 0;
  EpiKeys ek{0u, 0u, 1.f, 1.f};
  if ((AMODE == 2 || EPI == 1) && g.rng) {
    key_adrop = rng_key(g.rng, g.a_drop_site);
    key_adp = rng_key(g.rng, g.a_dp_site);
    ek.drop = rng_key(g.rng, g.drop_site);
    ek.dp = rng_key(g.rng, g.dp_site);
  }
  const float a_inv_keep = g.a_drop_p > 0.f ? 1.f / (1.f - g.a_drop_p) : 1.f;
  const float a_dp_inv = g.a_dp_p > 0.f ? 1.f / (1.f - g.a_dp_p) : 1.f;
  ek.inv_keep = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  ek.dp_inv = g.dp_p > 0.f ? 1.f / (1.f - g.dp_p) : 1.f;

  // this thread's staging slots: vector idx = tid + i*256 -> row idx/8, 8-element column group idx%8
  const int sv = tid & 7, sr = tid >> 3;            // rows advance by 32 per slot
  float mu[AV], rs[AV], rowf[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int m = m0 + sr + 32 * i;
    mu[i] = 0.f; rs[i] = 0.f; rowf[i] = 1.f;
    if (m < g.M) {
      if (AMODE == 1 && !stats_in) { mu[i] = g.ln_mean[m]; rs[i] = g.ln_rstd[m]; }
      if (bwd) {
        rowf[i] = g.a_scale;
        if (g.a_dp_p > 0.f) rowf[i] *= drop_factor(key_adp, (uint32_t)(m / g.a_dp_rows), g.a_dp_p, a_dp_inv);
      }
    }
  }
  if (AMODE == 1) {
    for (int k = tid; k < g.K; k += 256) { Gs[k] = g.ln_gamma[k]; Gs[g.K + k] = g.ln_beta[k]; }
    if (stats_in) {
      // a_mode 3: the row statistics are computed here (two passes over this workgroup's A rows, which the K loop reads
      // again from L1/L2) instead of by a row_stats launch in front of the GEMM; the first column block writes them out
      // for the backward.  The 8 threads that share a row (sv = tid & 7) are consecutive lanes.
      const float invK = 1.f / (float)g.K;
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const int m = m0 + sr + 32 * i;
        const int mc = m < g.M ? m : g.M - 1;
        float s = 0.f;
        for (int k0 = 0; k0 < g.K; k0 += BK) {
          const int k = k0 + sv * 8;
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(A + (size_t)mc * g.lda + (k < g.K ? k : 0));
          if (k < g.K) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)x[j];
          }
        }
        s = group_sum<8>(s);
        const float mean = s * invK;
        float q = 0.f;
        for (int k0 = 0; k0 < g.K; k0 += BK) {
          const int k = k0 + sv * 8;
          const bf16x8 x = *reinterpret_cast<const bf16x8*>(A + (size_t)mc * g.lda + (k < g.K ? k : 0));
          if (k < g.K) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)x[j] - mean; q += d * d; }
          }
        }
        q = group_sum<8>(q);
        const float rstd = rsqrtf(q * invK + g.ln_eps);
        mu[i] = mean; rs[i] = rstd;
        if (cb == 0 && sv == 0 && m < g.M) { g.ln_mean[m] = mean; g.ln_rstd[m] = rstd; }
      }
    }
  }

  bf16x8 pa[AV], pz[AV], pb[BV];
  auto prefetch = [&](int k0) {
    const int k = k0 + sv * 8;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int m = m0 + sr + 32 * i;
      bf16x8 x, z;
#pragma unroll
      for (int j = 0; j < 8; ++j) { x[j] = (bf16)0.f; z[j] = (bf16)0.f; }
      if (m < g.M && k < g.K) {
        if (A2 && k >= g.a2_k0) x = *reinterpret_cast<const bf16x8*>(A2 + (size_t)m * g.lda2 + (k - g.a2_k0));     // uniform per chunk (a2_k0 % 64 == 0)
        else x = *reinterpret_cast<const bf16x8*>(A + (size_t)m * g.lda + k);
        if (use_z) z = *reinterpret_cast<const bf16x8*>(Zin + (size_t)m * g.a_ldz + k);
      }
      pa[i] = x;
      pz[i] = z;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int n = n0 + sr + 32 * i;
      bf16x8 x;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (bf16)0.f;
      if (n < g.N && k < g.K) x = *reinterpret_cast<const bf16x8*>(B + (size_t)n * g.ldb + k);
      pb[i] = x;
    }
  };
  auto commit = [&](int k0) {
    const int k = k0 + sv * 8;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int m = m0 + sr + 32 * i;
      bf16x8 o = pa[i];
      if (AMODE == 1 && m < g.M && k < g.K) {
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
          const f32x4 ga = *reinterpret_cast<const f32x4*>(Gs + k + j);
          const f32x4 be = *reinterpret_cast<const f32x4*>(Gs + g.K + k + j);
#pragma unroll
          for (int q = 0; q < 4; ++q) o[j + q] = (bf16)(((float)o[j + q] - mu[i]) * rs[i] * ga[q] + be[q]);
        }
      }
      if (bwd && m < g.M && k < g.K) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)o[j] * rowf[i];
        if (g.a_drop_p > 0.f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] *= drop_factor(key_adrop, (uint32_t)m * (uint32_t)g.K + (uint32_t)(k + j), g.a_drop_p, a_inv_keep);
        }
        if (use_z) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] *= gelu_grad_f((float)pz[i][j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)f[j];
        if (write_aout) *reinterpret_cast<bf16x8*>(Aout + (size_t)m * g.a_ldo + k) = o;
      }
      *reinterpret_cast<bf16x8*>(As + (sr + 32 * i) * LDT + sv * 8) = o;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) *reinterpret_cast<bf16x8*>(Bs + (sr + 32 * i) * LDT + sv * 8) = pb[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LayerNorm-backward epilogue (EPI 2): what its first row group reads from memory -- the LayerNorm input, up to two addends, the
  // residual, the row statistics: 4 x the output's bytes -- is requested HERE, before the K loop, and the next group's while the current
  // one is worked on: as the epilogue's own loads they stood alone at the end of every workgroup (K = 960, M = 16384: 22.6 us without
  // the epilogue, 36 us with it).  Row tiles of 128 rows keep the plain form (their accumulators leave no registers for it).
  constexpr bool LNPRE = EPI == 2 && BM_ <= 64;
  constexpr int LN_NV = (EPI == 2) ? BN_ / 64 : 1;
  bf16x8 lx[LN_NV], la0[LN_NV], la1[LN_NV], lr[LN_NV];
  float lmu = 0.f, lrs = 0.f;
  auto ln_request = [&](int i) {
    const int srow = tid >> 3, c0 = (tid & 7) * (BN_ / 8);
    const int m = m0 + ((srow >> 4) * TM + i) * 16 + (srow & 15);
    const size_t mc = (size_t)(m < g.M ? m : g.M - 1);
    lmu = g.e_mean[mc]; lrs = g.e_rstd[mc];
#pragma unroll
    for (int v = 0; v < LN_NV; ++v) {
      const int c = c0 + 8 * v;
      lx[v] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(g.e_x) + mc * BN_ + c);
      if (g.e_add0) la0[v] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(g.e_add0) + mc * BN_ + c);
      if (g.e_add1) la1[v] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(g.e_add1) + mc * BN_ + c);
      if (g.R) lr[v] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(g.R) + mc * g.ldr + c);
    }
  };
  if (LNPRE) ln_request(0);

  prefetch(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    __syncthreads();                                  // previous chunk's MFMAs are done with As/Bs (and Gs is staged)
    commit(k0);
    __syncthreads();
    if (k0 + BK < g.K) prefetch(k0 + BK);
    const int nf = (g.K - k0 < BK) ? (g.K - k0) / 32 : BK / 32;
    for (int kf = 0; kf < nf; ++kf) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(As + ((wm * TM + i) * 16 + fr) * LDT + kf * 32 + fq * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + ((wn * TN + j) * 16 + fr) * LDT + kf * 32 + fq * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                                    // As/Bs are dead: reuse them as per-wave epilogue scratch
  if constexpr (EPI == 2) {
    // LAYERNORM-BACKWARD EPILOGUE (qavit_gemm_args.e_x): the tile holds whole rows of dy = the gradient of a LayerNorm's output (BN_ ==
    // N, one column block).  32 rows at a time -- the two row groups the 2 x 2 wave grid holds per step -- go through an fp32 LDS tile;
    // 8 threads share a row (BN_ / 8 consecutive columns each), so the two row means are in-lane sums + one DPP reduction over 8
    // adjacent lanes.  A thread keeps the same columns for every row group: its dgamma / dbeta column partials stay in registers until
    // the end, where they meet in the LDS tile and leave as this row tile's partial row (or as float atomics).
    constexpr int SLD = BN_ + 4, CPT = BN_ / 8, NV = CPT / 8;
    static_assert(BN_ % 64 == 0 && (BN_ == 128 || BN_ == 192 || BN_ == 256), "LayerNorm width = one column block of 128, 192 or 256");
    float* S = reinterpret_cast<float*>(smem);                 // [32][SLD]
    float* Gl = S + 32 * SLD;                                   // [BN_] gamma
    for (int c = tid; c < BN_; c += 256) Gl[c] = g.e_gamma[c];
    const int srow = tid >> 3, seg = tid & 7, c0 = seg * CPT;
    const bf16* X = reinterpret_cast<const bf16*>(g.e_x);
    const bf16* A0 = reinterpret_cast<const bf16*>(g.e_add0);
    const bf16* A1 = reinterpret_cast<const bf16*>(g.e_add1);
    const bf16* Rr = reinterpret_cast<const bf16*>(g.R);
    bf16* Cc = reinterpret_cast<bf16*>(g.C);
    float pg[CPT], pb[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) { pg[q] = 0.f; pb[q] = 0.f; }
    const float invN = 1.f / (float)BN_;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(wm * 16 + fq * 4 + r) * SLD + wn * WN + j * 16 + fr] = acc[i][j][r];
      __syncthreads();
      const int m = m0 + ((srow >> 4) * TM + i) * 16 + (srow & 15);
      const bool live = m < g.M;
      const size_t mc = (size_t)(live ? m : g.M - 1);
      if (!LNPRE) ln_request(i);
      const float mu = lmu, rs = lrs;
      bf16x8 rres[NV];
      float d[CPT], xh[CPT];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = c0 + 8 * v;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(S + srow * SLD + c), t1 = *reinterpret_cast<const f32x4*>(S + srow * SLD + c + 4);
        const bf16x8 xv = lx[v];
        rres[v] = lr[v];
        float dd[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        if (A0) { const bf16x8 av = la0[v];
#pragma unroll
                  for (int q = 0; q < 8; ++q) dd[q] += (float)av[q]; }
        if (A1) { const bf16x8 av = la1[v];
#pragma unroll
                  for (int q = 0; q < 8; ++q) dd[q] += (float)av[q]; }
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gl + c), g1 = *reinterpret_cast<const f32x4*>(Gl + c + 4);
        const float gm[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float dy = live ? dd[q] : 0.f, xx = ((float)xv[q] - mu) * rs, gg = dy * gm[q];
          d[8 * v + q] = gg;                                   // dy * gamma
          xh[8 * v + q] = xx;
          s1 += gg * xx;
          s2 += gg;
          pg[8 * v + q] += dy * xx;
          pb[8 * v + q] += dy;
        }
      }
      if (LNPRE && i + 1 < TM) ln_request(i + 1);             // the next row group's operands travel while this one is finished
      s1 = group_sum<8>(s1) * invN;
      s2 = group_sum<8>(s2) * invN;
      if (live) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = c0 + 8 * v;
          float o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = rs * (d[8 * v + q] - s2 - xh[8 * v + q] * s1);
          if (Rr) { const bf16x8 rv = rres[v];
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] += (float)rv[q]; }
          bf16x8 ov;
#pragma unroll
          for (int q = 0; q < 8; ++q) ov[q] = (bf16)o[q];
          *reinterpret_cast<bf16x8*>(Cc + mc * g.ldc + c) = ov;
        }
      }
      __syncthreads();                                        // the tile is free for the next row group
    }
    // column sums of the partials over the 32 row-threads of each segment, one array at a time through the LDS tile
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int q = 0; q < CPT; ++q) S[srow * SLD + c0 + q] = which ? pb[q] : pg[q];
      __syncthreads();
      for (int c = tid; c < BN_; c += 256) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += S[r * SLD + c];
        if (g.e_parts) g.e_parts[((size_t)tm * 2 + which) * BN_ + c] = t;
        else { float* dst = which ? g.e_dbeta : g.e_dgamma; if (dst) atomic_add_f(dst + c, t); }
      }
      __syncthreads();
    }
    return;
  }
  float* Ws = reinterpret_cast<float*>(smem) + wave * 16 * WLD;
  constexpr int CG = WN / 16;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ws[(fq * 4 + r) * WLD + j * 16 + fr] = acc[i][j][r];
    wave_sync();
    for (int idx = lane; idx < 16 * CG; idx += 64) {
      const int r = idx / CG, cg = idx - r * CG;
      const int m = m0 + (wm * TM + i) * 16 + r;
      const int n = n0 + wn * WN + cg * 16;
      if (m < g.M && n < g.N) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(Ws + r * WLD + cg * 16 + q);
          v[q] = t[0]; v[q + 1] = t[1]; v[q + 2] = t[2]; v[q + 3] = t[3];
        }
        epilogue16<EPI>(g, ek, m, n, v);
      }
    }
    wave_sync();
  }
}

template <int BM_, int BN_, int AMODE, int EPI>
int big_launch(const qavit_gemm_args& g, hipStream_t st) {
  const int n_tiles_m = (g.M + BM_ - 1) / BM_, ncb = (g.N + BN_ - 1) / BN_;
  size_t smem = (size_t)(BM_ + BN_) * LDT * 2 + (AMODE == 1 ? (size_t)2 * g.K * 4 : 0);
  const size_t scratch = EPI == 2 ? (size_t)(32 * (BN_ + 4) + BN_) * 4 : (size_t)4 * 16 * (BN_ / 2 + 4) * 4;
  if (smem < scratch) smem = scratch;
  if (smem > 150 * 1024) return -100;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_big_kernel<BM_, BN_, AMODE, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int grid = (n_tiles_m + 7) / 8 * 8 * ncb;
  hipLaunchKernelGGL((gemm_nt_big_kernel<BM_, BN_, AMODE, EPI>), dim3(grid), dim3(256), smem, st, g, n_tiles_m, ncb, g.a_mode == 3 ? 1 : 0);
  return QAVIT_OK;
}

// row-tile height the dispatch below picks for a problem of M rows and `ncb` column blocks (also the partial-row count of the
// LayerNorm-backward epilogue: one row per row tile)
int big_bm(int M, int ncb, int bn, int K = 0) {
  static const int bm32_below = getenv("QAVIT_BIG_BM32") ? atoi(getenv("QAVIT_BIG_BM32")) : 448;
  if (bn == 256) return 128;
  if ((long)((M + 127) / 128) * ncb >= 448) return 128;
  // a long contraction re-reads the whole weight once per row tile: from K = 768 the 64-row tile wins although it leaves one workgroup per
  // CU (the fan node's K = 960 GEMM at 16384 rows: 22.6 -> 21.8 us plain, 30.0 -> 26.4 us with the LayerNorm-backward epilogue)
  if (K >= 768) return 64;
  return (long)((M + 63) / 64) * ncb < bm32_below ? 32 : 64;
}

template <int BM_, int BN_>
int big_lnbwd(const qavit_gemm_args& g, hipStream_t st) {
  if constexpr (BN_ == 256 && BM_ != 128) return -100;
  else return g.a_mode == 2 ? big_launch<BM_, BN_, 2, 2>(g, st) : big_launch<BM_, BN_, 0, 2>(g, st);
}

template <int BM_, int BN_>
int big_modes(const qavit_gemm_args& g, hipStream_t st) {
  const bool full = g.Z || g.act || g.drop_p > 0.f || g.dp_p > 0.f || g.R || g.scale != 1.f;
  if (g.a_mode == 1 || g.a_mode == 3) return full ? big_launch<BM_, BN_, 1, 1>(g, st) : big_launch<BM_, BN_, 1, 0>(g, st);
  if (g.a_mode == 2) return full ? big_launch<BM_, BN_, 2, 1>(g, st) : big_launch<BM_, BN_, 2, 0>(g, st);
  return full ? big_launch<BM_, BN_, 0, 1>(g, st) : big_launch<BM_, BN_, 0, 0>(g, st);
}

}  // namespace

bool gemm_nt_lnbwd_shape_ok(int dtype, int M, int N, int K, int a_mode) {
  return dtype == QAVIT_BF16 && M >= 1024 && (N == 128 || N == 192 || N == 256) && K >= 64 && K % 32 == 0 && (a_mode == 0 || a_mode == 2);
}
int gemm_nt_lnbwd_parts(int M, int N, int K) { return (M + big_bm(M, 1, N, K) - 1) / big_bm(M, 1, N, K); }

// the LayerNorm-backward epilogue (qavit_gemm_args.e_x): only this kernel has it.  1 = launched, < 0 error (never "not applicable":
// the caller asked qavit_gemm_nt_lnbwd_supported first)
int gemm_nt_big_lnbwd(const qavit_gemm_args& g, hipStream_t st) {
  if (!gemm_nt_lnbwd_shape_ok(g.dtype, g.M, g.N, g.K, g.a_mode)) return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm-backward epilogue: bf16, M >= 1024, N in {128, 192, 256}, K >= 64, K % 32 == 0, a_mode 0 or 2");
  auto al = [](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; };
  if (!g.e_mean || !g.e_rstd || !g.e_gamma || g.A2 || g.bias || g.Z || g.act || g.drop_p > 0.f || g.dp_p > 0.f || g.scale != 1.f)
    return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm-backward epilogue takes no bias / activation / dropout / scale and needs mean, rstd, gamma");
  if (!al(g.A, g.lda) || !al(g.B, g.ldb) || !al(g.C, g.ldc) || !al(g.e_x, 8) || (g.e_add0 && !al(g.e_add0, 8)) || (g.e_add1 && !al(g.e_add1, 8)) ||
      (g.R && !al(g.R, g.ldr)) || (g.e_parts && (reinterpret_cast<uintptr_t>(g.e_parts) & 15)))
    return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm-backward epilogue: 16-byte aligned operands, leading dimensions % 8");
  if (g.a_mode == 2 && ((g.a_Z && g.a_act && !al(g.a_Z, g.a_ldz)) || (g.a_out && !al(g.a_out, g.a_ldo))))
    return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm-backward epilogue: a_Z / a_out alignment");
  const int bm = big_bm(g.M, 1, g.N, g.K);
  int rc;
  if (g.N == 256) rc = big_lnbwd<128, 256>(g, st);
  else if (g.N == 192) rc = bm == 128 ? big_lnbwd<128, 192>(g, st) : (bm == 32 ? big_lnbwd<32, 192>(g, st) : big_lnbwd<64, 192>(g, st));
  else rc = bm == 128 ? big_lnbwd<128, 128>(g, st) : (bm == 32 ? big_lnbwd<32, 128>(g, st) : big_lnbwd<64, 128>(g, st));
  if (rc == -100) return set_error(QAVIT_EINVAL, "gemm_nt: LayerNorm-backward epilogue: tile does not fit LDS");
  if (rc != QAVIT_OK) return rc;
  rc = check_launch("gemm_nt(big, LayerNorm backward)");
  return rc == QAVIT_OK ? 1 : rc;
}

// returns 1 = launched, 0 = not applicable (caller falls back to the resident-slice kernel), < 0 error
int gemm_nt_big_try(const qavit_gemm_args& g_in, hipStream_t st) {
  qavit_gemm_args g = g_in;
  static long thresh = -1, force_bn = 0, wide_n = 0, stats_ncb = 3;
  if (thresh < 0) {
    const char* e0 = getenv("QAVIT_BIG_STATS_NCB"); stats_ncb = e0 ? atol(e0) : 3;
    const char* e = getenv("QAVIT_GEMM_BIG"); thresh = e ? atol(e) : 64L * 128L;
    e = getenv("QAVIT_BIG_BN"); force_bn = e ? atol(e) : 0;
    e = getenv("QAVIT_BIG_WIDE_N"); wide_n = e ? atol(e) : 0;
  }
  if (thresh == 0 || (long)g.N * g.K < thresh || g.N < 64 || g.K < 96 || g.M < 1024) return 0;
  if (g.K % 32) return 0;
  auto al = [](const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; };
  if (!al(g.A, g.lda) || !al(g.B, g.ldb)) return 0;
  if (g.A2 && (g.a_mode != 0 || g.a2_k0 % 64 || g.a2_k0 <= 0 || g.a2_k0 >= g.K || !al(g.A2, g.lda2))) return 0;
  if (g.a_mode == 2) {
    if (g.a_Z && g.a_act && !al(g.a_Z, g.a_ldz)) return 0;
    if (g.a_out && !al(g.a_out, g.a_ldo)) return 0;
  }
  // column block: the whole N when the A-side transform is expensive (a_mode 2 would repeat it per block), otherwise
  // 128-wide blocks (64 accumulator registers -> two workgroups per CU) unless N is a multiple of 192 only
  int bn;
  if (g.a_mode == 2 && g.N <= 256) bn = g.N > 192 ? 256 : (g.N > 128 ? 192 : 128);
  else if (wide_n > 0 && g.N >= wide_n && g.N % 256 == 0) bn = 256;   // fat FFN layers (N = 1024): half the column blocks, so half the A re-reads and prologue repeats
  else if (g.N % 128 == 0) bn = 128;
  else if (g.N % 192 == 0) bn = 192;
  else bn = g.N > 192 ? 256 : (g.N > 128 ? 192 : 128);
  if (force_bn) bn = (int)force_bn;
  const int ncb = (g.N + bn - 1) / bn;
  if (g.a_mode == 3 && stats_ncb > 0 && ncb >= stats_ncb && g.lda == g.K) {
    // every column block of a row tile would recompute that tile's LayerNorm statistics in its prologue (two dependent passes
    // over its A rows before the first MFMA): with several column blocks one row_stats launch in front is cheaper
    const int rc0 = qavit_row_stats(g.dtype, g.A, g.ln_eps, g.M, g.K, g.ln_mean, g.ln_rstd, reinterpret_cast<void*>(st));
    if (rc0) return rc0;
    g.a_mode = 1;
  }
  int rc;
  // fewer than ~1.75 workgroups per CU at 64-row tiles (rows = B*16 learned tokens): 32-row tiles put two workgroups on a
  // CU so one's global-load latency hides behind the other's MFMAs (big_bm)
  const int bm_pick = big_bm(g.M, ncb, bn, g.K);
  const bool bm128 = bm_pick == 128, bm32 = bm_pick == 32;
  if (bn == 256) rc = big_modes<128, 256>(g, st);                      // 64-row tiles would starve the MFMA pipe here
  else if (bn == 192) rc = bm128 ? big_modes<128, 192>(g, st) : (bm32 ? big_modes<32, 192>(g, st) : big_modes<64, 192>(g, st));
  else rc = bm128 ? big_modes<128, 128>(g, st) : (bm32 ? big_modes<32, 128>(g, st) : big_modes<64, 128>(g, st));
  if (rc == -100) return 0;
  if (rc != QAVIT_OK) return rc;
  rc = check_launch("gemm_nt(big)");
  return rc == QAVIT_OK ? 1 : rc;
}

}  // namespace qv
