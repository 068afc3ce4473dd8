// bf16 weight-gradient GEMM  C[N,K] += A[M,N]^T . B[M,K]  (+ colsum[N] += sum_m A)  with WIDE output tiles.
//
// dW of a Linear(192,192) at M = 16k rows reads 12.6 MB and produces a 192x192 result: the work is all reduction.
// The 64x64-tile kernel in gemm.hip re-read A and B three times each and spent a barrier pair per 8 MFMAs; here a
// workgroup owns a (32*IN) x (32*JN) tile -- the WHOLE dW when N, K <= 192 -- so each operand row is read once and a
// 64-row chunk feeds IN*JN*2 MFMAs per wave between barriers.  Operands are staged row-major ([m][n], [m][k]: 16-byte
// coalesced loads, register-prefetched one chunk ahead); the m-contiguous MFMA fragments come from
// ds_read_b64_tr_b16.  Row strides are 16 elements over the tile width, i.e. 8 banks (mod 32) per row, so the four
// rows of a transposed read fall in disjoint bank groups.  Partial results of the M-splits are added with fp32
// atomics; up to TNW_GROUP problems of one tile class share a launch (the dW GEMMs of a backward pass are independent).
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "gemm_shared.h"
#include <stdlib.h>

namespace qv {

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

__device__ __forceinline__ bf16x8 trf(const bf16* tile, int ld, int m0, int c0) {
  // lane l: rows m0 + 8*(l>>4) + j (j = 0..7), column c0 + (l & 15)
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const bf16* a0 = tile + (m0 + 8 * g + q) * ld + c0 + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(a0 + 4 * ld));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

constexpr int MC = 64;            // rows of M per staged chunk
constexpr int TNW_GROUP = 24;

struct TnwGroup {
  int n;
  int wg_start[TNW_GROUP + 1];
  int tn[TNW_GROUP], tk[TNW_GROUP], rows[TNW_GROUP];
  qavit_gemm_tn_args p[TNW_GROUP];
};

constexpr int pow2ceil(int v) { return v <= 4 ? 4 : v <= 8 ? 8 : v <= 16 ? 16 : 32; }

// staging geometry of one operand with W columns: column group cg (8 columns) is FIXED per thread, so LayerNorm
// gamma/beta and the column sums live in registers
template <int W>
struct Stage {
  static constexpr int CG = W / 8;              // live column groups
  static constexpr int CGS = pow2ceil(CG);      // slots (power of two)
  static constexpr int RP = 256 / CGS;          // rows per pass
  static constexpr int PASS = MC / RP;
  static constexpr int LD = W + 16;
};

// Preconditions (gemm_tn_wide checks them, other problems take the generic kernel): A, B 16-byte aligned,
// lda, ldb, N, K multiples of 8.  The staging loads are UNCONDITIONAL (row / column indices are clamped into the
// operand and the out-of-range vectors are zeroed when they are committed to LDS): a load under a lane-dependent
// branch makes the compiler wait for it at the join, which serialises the chunk's 16 loads into 16 round trips.
template <int IN, int JN>
__device__ __forceinline__ void tn_wide_body(const qavit_gemm_tn_args& g, bf16* At, bf16* Bt, int bx, int by, int bz, int rows_per_split) {
  constexpr int TNW = 32 * IN, TKW = 32 * JN;
  typedef Stage<TNW> SA;
  typedef Stage<TKW> SB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave & 1, wb = wave >> 1;
  const int n0 = bx * TNW, k0 = by * TKW;
  const int mbeg = bz * rows_per_split;
  const int mend = (mbeg + rows_per_split < g.M) ? mbeg + rows_per_split : g.M;
  if (mbeg >= mend) return;                            // uniform per workgroup
  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  const bool ln = g.ln_mean != nullptr;
  const bool want_csum = g.colsum && by == 0;
  const float* mean_p = ln ? g.ln_mean : g.C;          // any readable fp32 address when LayerNorm is off
  const float* rstd_p = ln ? g.ln_rstd : g.C;

  const int cga = tid & (SA::CGS - 1), rga = tid / SA::CGS;
  const int cgb = tid & (SB::CGS - 1), rgb = tid / SB::CGS;
  const int na = n0 + 8 * cga, kb = k0 + 8 * cgb;
  const bool a_live = cga < SA::CG && na < g.N, b_live = cgb < SB::CG && kb < g.K;
  const bf16* a_col = A + (a_live ? na : 0);
  const bf16* b_col = B + (b_live ? kb : 0);

  float gam[8], bet[8], csum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    csum[j] = 0.f;
    gam[j] = (ln && b_live) ? g.ln_gamma[kb + j] : 0.f;
    bet[j] = (ln && b_live) ? g.ln_beta[kb + j] : 0.f;
  }

  f32x4 acc[IN][JN];
#pragma unroll
  for (int i = 0; i < IN; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 pa[SA::PASS], pb[SB::PASS];
  float pmu[SB::PASS], prs[SB::PASS];
  auto prefetch = [&](int mc) {
#pragma unroll
    for (int h = 0; h < SA::PASS; ++h) {
      int m = mc + rga + SA::RP * h;
      m = m < mend ? m : mend - 1;
      pa[h] = *reinterpret_cast<const bf16x8*>(a_col + (size_t)m * g.lda);
    }
#pragma unroll
    for (int h = 0; h < SB::PASS; ++h) {
      int m = mc + rgb + SB::RP * h;
      m = m < mend ? m : mend - 1;
      pb[h] = *reinterpret_cast<const bf16x8*>(b_col + (size_t)m * g.ldb);
      const int mi = ln ? m : 0;
      pmu[h] = mean_p[mi];
      prs[h] = rstd_p[mi];
    }
  };

  prefetch(mbeg);
  for (int mc = mbeg; mc < mend; mc += MC) {
    __syncthreads();
    if (cga < SA::CG) {
#pragma unroll
      for (int h = 0; h < SA::PASS; ++h) {
        const bool ok = a_live && (mc + rga + SA::RP * h < mend);
        bf16x8 v = pa[h];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : (bf16)0.f;
        *reinterpret_cast<bf16x8*>(At + (rga + SA::RP * h) * SA::LD + 8 * cga) = v;
        if (want_csum) {
#pragma unroll
          for (int j = 0; j < 8; ++j) csum[j] += (float)v[j];
        }
      }
    }
    if (cgb < SB::CG) {
#pragma unroll
      for (int h = 0; h < SB::PASS; ++h) {
        const bool ok = b_live && (mc + rgb + SB::RP * h < mend);
        bf16x8 v = pb[h];
        if (ln) {                                        // uniform
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (bf16)(((float)v[j] - pmu[h]) * prs[h] * gam[j] + bet[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : (bf16)0.f;
        *reinterpret_cast<bf16x8*>(Bt + (rgb + SB::RP * h) * SB::LD + 8 * cgb) = v;
      }
    }
    __syncthreads();
    if (mc + MC < mend) prefetch(mc + MC);
#pragma unroll
    for (int kf = 0; kf < MC / 32; ++kf) {
      bf16x8 af[IN], bfr[JN];
#pragma unroll
      for (int i = 0; i < IN; ++i) af[i] = trf(At, SA::LD, kf * 32, (wa * IN + i) * 16);
#pragma unroll
      for (int j = 0; j < JN; ++j) bfr[j] = trf(Bt, SB::LD, kf * 32, (wb * JN + j) * 16);
#pragma unroll
      for (int i = 0; i < IN; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < IN; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wa * IN + i) * 16 + fq * 4 + r, k = k0 + (wb * JN + j) * 16 + fr;
        if (n < g.N && k < g.K) atomic_add_f(g.C + (size_t)n * g.ldc + k, acc[i][j][r]);
      }
  if (want_csum) {                                   // uniform per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(At);       // [TNW]
    for (int i = tid; i < TNW; i += 256) red[i] = 0.f;
    __syncthreads();
    if (cga < SA::CG) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(red + 8 * cga + j, csum[j]);
    }
    __syncthreads();
    for (int i = tid; i < TNW; i += 256) if (n0 + i < g.N) atomic_add_f(g.colsum + n0 + i, red[i]);
  }
}

template <int IN, int JN>
__global__ __launch_bounds__(256) void gemm_tn_wide_kernel(TnwGroup G) {
  __shared__ __attribute__((aligned(16))) bf16 At[MC * Stage<32 * IN>::LD];   // [m][n]
  __shared__ __attribute__((aligned(16))) bf16 Bt[MC * Stage<32 * JN>::LD];   // [m][k]
  const int bid = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int j = 1; j < TNW_GROUP; ++j) if (j < G.n && bid >= G.wg_start[j]) i = j;
  const int local = bid - G.wg_start[i];
  const int tiles = G.tn[i] * G.tk[i];
  const int bz = local / tiles, t = local - bz * tiles;
  const int by = t / G.tn[i], bx = t - by * G.tn[i];
  tn_wide_body<IN, JN>(G.p[i], At, Bt, bx, by, bz, G.rows[i]);
}

int tile_class(int n, int which = 0) {           // 32-column units per workgroup tile: 1, 2, 4 or 6
  static int force[2] = {-1, -1};
  if (force[0] < 0) {
    const char* e = getenv("QAVIT_TN_CN"); force[0] = e ? atoi(e) : 0;
    e = getenv("QAVIT_TN_CK"); force[1] = e ? atoi(e) : 0;
  }
  if (n > 128 && force[which]) return force[which];
  if (n <= 32) return 1;
  if (n <= 64) return 2;
  if (which == 1) {                // K side: <= 128-wide tiles keep the accumulators at <= 96 registers (two workgroups per CU)
    static int k96 = -1;
    if (k96 < 0) { const char* e = getenv("QAVIT_TN_K96"); k96 = e ? atoi(e) : 0; }   // measured: no gain (more classes = more launches)
    if (k96 && n % 96 == 0 && n % 128 != 0) return 3;     // 96, 192, 576: exact 96-wide tiles instead of padded 128-wide ones
    return 4;
  }
  if (n <= 128) return 4;
  if (n <= 192) return 6;
  const int p128 = (n + 127) / 128 * 128, p192 = (n + 191) / 192 * 192;
  return p128 < p192 ? 4 : 6;
}

template <int IN, int JN>
void launch_class(const qavit_gemm_tn_args* const* probs, int n, hipStream_t st) {
  constexpr int TNW = 32 * IN, TKW = 32 * JN;
  static int target = -1;
  if (target < 0) { const char* e = getenv("QAVIT_TN_WGS"); target = e ? atoi(e) : 256; }
  int done = 0;
  while (done < n) {
    const int cnt = (n - done < TNW_GROUP) ? (n - done) : TNW_GROUP;
    // rows per split: long chains amortise the atomic epilogue; shorten them until the launch fills the chip
    int rows = 4096;
    for (;;) {
      long wg = 0;
      for (int i = 0; i < cnt; ++i) {
        const qavit_gemm_tn_args& g = *probs[done + i];
        wg += (long)((g.N + TNW - 1) / TNW) * ((g.K + TKW - 1) / TKW) * ((g.M + rows - 1) / rows);
      }
      if (wg >= target || rows <= 128) break;
      rows >>= 1;
    }
    TnwGroup G;
    G.n = cnt;
    int wg = 0;
    for (int i = 0; i < cnt; ++i) {
      const qavit_gemm_tn_args& g = *probs[done + i];
      G.p[i] = g;
      G.tn[i] = (g.N + TNW - 1) / TNW; G.tk[i] = (g.K + TKW - 1) / TKW;
      int r = g.splits > 0 ? (g.M + g.splits - 1) / g.splits : rows;
      r = (r + MC - 1) / MC * MC;
      G.rows[i] = r;
      G.wg_start[i] = wg;
      wg += G.tn[i] * G.tk[i] * ((g.M + r - 1) / r);
    }
    G.wg_start[cnt] = wg;
    hipLaunchKernelGGL((gemm_tn_wide_kernel<IN, JN>), dim3(wg), dim3(256), 0, st, G);
    done += cnt;
  }
}

typedef void (*class_fn)(const qavit_gemm_tn_args* const*, int, hipStream_t);
template <int IN> class_fn pick_j(int jc) {
  switch (jc) {
    case 1: return launch_class<IN, 1>;
    case 2: return launch_class<IN, 2>;
    case 3: return launch_class<IN, 3>;
    default: return launch_class<IN, 4>;
  }
}
class_fn pick(int ic, int jc) {
  switch (ic) {
    case 1: return pick_j<1>(jc);
    case 2: return pick_j<2>(jc);
    case 4: return pick_j<4>(jc);
    default: return pick_j<6>(jc);
  }
}

}  // namespace

// All problems must be bf16 and validated by the caller.  Launches them grouped by tile class.
bool gemm_tn_wide_ok(const qavit_gemm_tn_args& g) {
  return g.dtype == QAVIT_BF16 && g.N % 8 == 0 && g.K % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 &&
         ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) == 0;
}

int gemm_tn_wide(const qavit_gemm_tn_args* a, int n, hipStream_t st) {
  static const int ncls[4] = {1, 2, 4, 6}, kcls[4] = {1, 2, 3, 4};
  const qavit_gemm_tn_args* sel[256];
  for (int ci = 0; ci < 4; ++ci)
    for (int cj = 0; cj < 4; ++cj) {
      int cnt = 0;
      for (int i = 0; i < n; ++i) {
        if (gemm_tn_wide_ok(a[i]) && tile_class(a[i].N, 0) == ncls[ci] && tile_class(a[i].K, 1) == kcls[cj]) {
          sel[cnt++] = a + i;
          if (cnt == 256) { pick(ncls[ci], kcls[cj])(sel, cnt, st); cnt = 0; }
        }
      }
      if (cnt) pick(ncls[ci], kcls[cj])(sel, cnt, st);
    }
  return check_launch("gemm_tn(wide)");
}

}  // namespace qv
