// bf16 weight-gradient GEMM  C[N,K] += A[M,N]^T . B[M,K]  (+ colsum[N] += sum_m A)  with WIDE output tiles.
//
// dW of a Linear(192,192) at M = 16k rows reads 12.6 MB and produces a 192x192 result: the work is all reduction.
// The 64x64-tile kernel in gemm.hip re-read A and B three times each and spent a barrier pair per 8 MFMAs; here a
// workgroup owns a (32*IN) x (32*JN) tile -- the WHOLE dW when N, K <= 192 -- so each operand row is read once and a
// 64-row chunk feeds IN*JN*2 MFMAs per wave between barriers.  Operands are staged row-major ([m][n], [m][k]: 16-byte
// coalesced loads, register-prefetched one chunk ahead); the m-contiguous MFMA fragments come from
// ds_read_b64_tr_b16.  Row strides are 16 elements over the tile width, i.e. 8 banks (mod 32) per row, so the four
// rows of a transposed read fall in disjoint bank groups.
//
// Scheduling is stream-K: up to TNW_GROUP problems of one tile class share a launch (the dW GEMMs of a backward pass
// are independent); the launch's work is the list of (problem, tile, 64- or 128-row chunk) units in that order, cut
// into EQUAL contiguous ranges, one per workgroup, with as many workgroups as the chip holds at once.  A workgroup
// accumulates in registers while its range stays inside one tile and adds its partial result with fp32 atomics when it
// leaves the tile, so a launch has no tail round (the fixed rows-per-split grid it replaces ran 576 workgroups on 512
// slots: two rounds for 1.1 rounds of work) and the atomics stay at one or two flushes per workgroup.
//
// Tile classes: N side 32 / 64 / 128 / 192 columns, K side 32 / 64 / 96 / 128 columns on four waves (2 x 2), and a
// 192-column K class on eight waves (2 x 4, 128-row chunks) so that the K = 192 layers -- most of the model -- read
// each operand row once (two 128-wide tiles re-read A and spent a third of their MFMAs on padding).
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

constexpr int TNW_GROUP = 24;

struct TnwGroup {
  int n;
  int per;                            // units per workgroup
  int dbg;
  int unit_start[TNW_GROUP + 1];      // prefix sums of tiles * chunks
  int tn[TNW_GROUP], chunks[TNW_GROUP];
  qavit_gemm_tn_args p[TNW_GROUP];
};

constexpr int pow2ceil(int v) { return v <= 4 ? 4 : v <= 8 ? 8 : v <= 16 ? 16 : 32; }

// staging geometry of one operand with W columns: column group cg (8 columns) is FIXED per thread, so LayerNorm
// gamma/beta and the column sums live in registers
template <int NT, int W, int MC>
struct Stage {
  static constexpr int CG = W / 8;              // live column groups
  static constexpr int CGS = pow2ceil(CG);      // slots (power of two)
  static constexpr int RP = NT / CGS;           // rows per pass
  static constexpr int PASS = MC / RP;
  static constexpr int LD = W + 16;
};

// class geometry: waves 2 x WB, a wave owns IN x JN MFMA blocks; MC = rows of M per staged chunk
template <int IN, int JN, int WB>
struct Cls {
  static constexpr int NT = 128 * WB;
  static constexpr int MC = WB == 4 ? 128 : 64;
  static constexpr int TNW = 32 * IN, TKW = 16 * WB * JN;
  typedef Stage<NT, TNW, MC> SA;
  typedef Stage<NT, TKW, MC> SB;
};

// Preconditions (gemm_tn_wide checks them, other problems take the generic kernel): A, B 16-byte aligned,
// lda, ldb, N, K multiples of 8.  The staging loads are UNCONDITIONAL (row / column indices are clamped into the
// operand and the out-of-range vectors are zeroed when they are committed to LDS): a load under a lane-dependent
// branch makes the compiler wait for it at the join, which serialises the chunk's 16 loads into 16 round trips.
template <int IN, int JN, int WB>
__device__ __forceinline__ void tn_wide_body(const qavit_gemm_tn_args& g, bf16* At, bf16* Bt, int bx, int by, int mbeg, int mend, int dbg) {
  typedef Cls<IN, JN, WB> CL;
  constexpr int TNW = CL::TNW, TKW = CL::TKW, MC = CL::MC, NT = CL::NT;
  typedef typename CL::SA SA;
  typedef typename CL::SB SB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave & 1, wb = wave >> 1;
  const int n0 = bx * TNW, k0 = by * TKW;
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
  if (!(dbg & 1))
#pragma unroll
  for (int i = 0; i < IN; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wa * IN + i) * 16 + fq * 4 + r, k = k0 + (wb * JN + j) * 16 + fr;
        if (n < g.N && k < g.K) atomic_add_f(g.C + (size_t)n * g.ldc + k, acc[i][j][r]);
      }
  if (want_csum && !(dbg & 2)) {                     // uniform per workgroup
    // lanes l and l ^ CGS, l ^ 2 CGS ... of a wave hold the same column group (other rows): fold them with shuffles, then the
    // waves through LDS (plain stores; the LDS-atomic version of this spent 16-way same-address conflicts per column)
#pragma unroll
    for (int off = SA::CGS; off < 64; off <<= 1)
#pragma unroll
      for (int j = 0; j < 8; ++j) csum[j] += __shfl_xor(csum[j], off);
    __syncthreads();
    float* red = reinterpret_cast<float*>(At);       // [waves][8 * CGS]
    constexpr int NWV = NT / 64, RW = 8 * (SA::CGS < 64 ? SA::CGS : 64);
    if (lane < SA::CGS && cga < SA::CG) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave * RW + 8 * cga + j] = csum[j];
    }
    __syncthreads();
    for (int i = tid; i < TNW; i += NT) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) t += red[w * RW + i];
      if (n0 + i < g.N) atomic_add_f(g.colsum + n0 + i, t);
    }
    __syncthreads();                                 // At is staged again by the next range of this workgroup
  }
}

template <int IN, int JN, int WB>
__global__ __launch_bounds__(128 * WB) void gemm_tn_wide_kernel(TnwGroup G) {
  typedef Cls<IN, JN, WB> CL;
  __shared__ __attribute__((aligned(16))) bf16 At[CL::MC * CL::SA::LD];   // [m][n]
  __shared__ __attribute__((aligned(16))) bf16 Bt[CL::MC * CL::SB::LD];   // [m][k]
  const int total = G.unit_start[G.n];
  int u = blockIdx.x * G.per;
  int uend = u + G.per;
  if (uend > total) uend = total;
  int i = 0;
  while (u < uend) {                                   // every quantity here is uniform over the workgroup
    while (u >= G.unit_start[i + 1]) ++i;
    const int chunks = G.chunks[i];
    const int local = u - G.unit_start[i];
    const int t = local / chunks, c0 = local - t * chunks;
    int c1 = c0 + (uend - u);
    if (c1 > chunks) c1 = chunks;
    const int tn = G.tn[i];
    const int by = t / tn, bx = t - by * tn;
    const int M = G.p[i].M;
    const int mend = c1 * CL::MC < M ? c1 * CL::MC : M;
    tn_wide_body<IN, JN, WB>(G.p[i], At, Bt, bx, by, c0 * CL::MC, mend, G.dbg);
    u += c1 - c0;
  }
}

// workgroups the chip holds at once, per class (two 4-wave workgroups or one 8-wave workgroup per CU)
int resident_wgs(int wb) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  static int per_cu = -1;
  if (per_cu < 0) { const char* e = getenv("QAVIT_TN_WG_PER_CU"); per_cu = e ? atoi(e) : 0; }
  if (wb == 4) return cus;
  return cus * (per_cu > 0 ? per_cu : 2);
}

int tile_class(int n, int which = 0) {           // N side: 32-column units per tile (1, 2, 4, 6); K side: 32-column units (1, 2, 3, 4) or 6 = the 192-wide eight-wave class
  static int force[2] = {-1, -1};
  if (force[0] < 0) {
    const char* e = getenv("QAVIT_TN_CN"); force[0] = e ? atoi(e) : 0;
    e = getenv("QAVIT_TN_CK"); force[1] = e ? atoi(e) : 0;
  }
  if (n > 128 && force[which]) return force[which];
  if (n <= 32) return 1;
  if (n <= 64) return 2;
  if (which == 1 && n <= 96) return 3;
  if (n <= 128) return 4;
  if (n <= 192) return 6;
  const int p128 = (n + 127) / 128 * 128, p192 = (n + 191) / 192 * 192;
  return p128 < p192 ? 4 : 6;
}

template <int IN, int JN, int WB>
void launch_class(const qavit_gemm_tn_args* const* probs, int n, hipStream_t st) {
  typedef Cls<IN, JN, WB> CL;
  static int min_units = -1;
  if (min_units < 0) { const char* e = getenv("QAVIT_TN_MIN_UNITS"); min_units = e ? atoi(e) : 4; if (min_units < 1) min_units = 1; }
  int done = 0;
  while (done < n) {
    const int cnt = (n - done < TNW_GROUP) ? (n - done) : TNW_GROUP;
    TnwGroup G;
    G.n = cnt;
    G.dbg = 0;
#ifdef QAVIT_TN_FLUSH_EXPERIMENT   // diagnostic build only (QAVIT_EXTRA_HIPCC_FLAGS=-DQAVIT_TN_FLUSH_EXPERIMENT): QAVIT_TN_DBG bit 0 skips the tile flush,
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("QAVIT_TN_DBG"); dbg = e ? atoi(e) : 0; } G.dbg = dbg; }   // bit 1 the column-sum flush -- WRONG results, timing only
#endif
    int units = 0;
    for (int i = 0; i < cnt; ++i) {
      const qavit_gemm_tn_args& g = *probs[done + i];
      G.p[i] = g;
      G.tn[i] = (g.N + CL::TNW - 1) / CL::TNW;
      G.chunks[i] = (g.M + CL::MC - 1) / CL::MC;
      G.unit_start[i] = units;
      units += G.tn[i] * ((g.K + CL::TKW - 1) / CL::TKW) * G.chunks[i];
    }
    for (int i = cnt; i <= TNW_GROUP; ++i) G.unit_start[i] = units;
    // equal contiguous ranges over the resident workgroups; short launches use fewer, longer ranges (each range
    // ends in a tile-sized atomic flush)
    int wgs = resident_wgs(WB);
    if (units < wgs * min_units) wgs = (units + min_units - 1) / min_units;
    G.per = (units + wgs - 1) / wgs;
    wgs = (units + G.per - 1) / G.per;
    hipLaunchKernelGGL((gemm_tn_wide_kernel<IN, JN, WB>), dim3(wgs), dim3(CL::NT), 0, st, G);
    done += cnt;
  }
}

typedef void (*class_fn)(const qavit_gemm_tn_args* const*, int, hipStream_t);
template <int IN> class_fn pick_j(int jc) {
  switch (jc) {
    case 1: return launch_class<IN, 1, 2>;
    case 2: return launch_class<IN, 2, 2>;
    case 3: return launch_class<IN, 3, 2>;
    case 4: return launch_class<IN, 4, 2>;
    default: return launch_class<IN, 3, 4>;
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
  static const int ncls[4] = {1, 2, 4, 6}, kcls[5] = {1, 2, 3, 4, 6};
  const qavit_gemm_tn_args* sel[256];
  for (int ci = 0; ci < 4; ++ci)
    for (int cj = 0; cj < 5; ++cj) {
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
