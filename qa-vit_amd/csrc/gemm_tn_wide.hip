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
// Tile classes: N side 32 / 64 / 128 / 192 / 256 columns, K side 32 / 64 / 96 / 128 columns on four waves (2 x 2), and
// 192- and 256-column K classes on eight waves (2 x 4, 128-row chunks; 64-row chunks under the 256 x 256 tile) so that
// the K = 192 layers -- most of the model -- read each operand row once (two 128-wide tiles re-read A and spent a third
// of their MFMAs on padding) and the lateral path's 256 <-> 1024 layers re-read 4 + 1 times instead of 8 + 2.
//
// With a workspace (qavit_gemm_tn_grouped_ws) ALL problems of a call share one launch whatever their class: gemm_tn_uni_kernel below.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "gemm_shared.h"
#include <stdlib.h>

namespace qv {

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

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

// Operand pointers that reach the kernel through a table in memory (not as kernel arguments) are generic to the compiler, and
// FLAT loads count on lgkmcnt as well as vmcnt: every wait for an LDS fragment would also wait for the chunk prefetch.  These say
// "global" explicitly.
#define QV_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ bf16x8 gload8(const bf16* p) { return *(const QV_AS1 bf16x8*)(p); }
__device__ __forceinline__ float gloadf(const float* p) { return *(const QV_AS1 float*)(p); }
__device__ __forceinline__ void gatomic_add(float* p, float v) { __builtin_amdgcn_global_atomic_fadd_f32((QV_AS1 float*)(p), v); }

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
template <int IN_, int JN_, int WB_>
struct Cls {
  static constexpr int IN = IN_, JN = JN_, WB = WB_;
  static constexpr int NT = 128 * WB;
  static constexpr int MC = (WB == 4 && IN * JN < 24) ? 128 : 64;   // 256 x 256 tile: 16 staged vectors per thread at 128 rows do not fit beside 128 accumulators
  static constexpr int TNW = 32 * IN, TKW = 16 * WB * JN;
  typedef Stage<NT, TNW, MC> SA;
  typedef Stage<NT, TKW, MC> SB;
};

// Geometry of the ONE-LAUNCH kernel (gemm_tn_uni_kernel): every class on eight waves (2 x 4), N side 32 * IN columns (IN = 1, 2, 4, 6,
// 8), K side 64 * JN columns (JN = 1 .. 4).  The chunk is as long as LDS and the staging registers allow -- a skinny tile moves few
// bytes per row, and with one workgroup per CU the bytes in flight per chunk are what hides the memory latency.
constexpr int ucls_mc(int in, int jn) {
  const int w = 32 * in + 64 * jn, row = (w + 32) * 2;
  if (in * jn >= 24) return 64;                      // 96+ accumulator registers: 8 + 8 staged vectors at most
  int mc = 512;
  while (mc * row > 150 * 1024 || (w >= 256 && mc > 128)) mc >>= 1;
  return mc;
}
constexpr int ucls_lds(int in, int jn) { return ucls_mc(in, jn) * (32 * in + 64 * jn + 32) * 2; }
// one (tile, chunk) unit: operand elements / 64 + the chunk's fixed part, times what the class was MEASURED to take per such unit inside the
// one launch (tools/tn_stamps.py, C100 step at B = 1024: the 192-column K classes run 11-22 % over the byte model, the skinniest class 16 %
// under) -- the launch lasts as long as its slowest range, and ranges made of one class only were 19 % over the mean
constexpr int ucls_adj_pct(int in, int jn) {
  if (jn == 3) return in == 2 ? 122 : in == 4 ? 111 : in == 6 ? 120 : 115;
  if (in == 1) return jn == 1 ? 84 : 95;
  if (in == 8 && jn == 4) return 111;
  if (jn == 4 && in <= 4) return 97;
  return 100;
}
constexpr int ucls_cost(int in, int jn) { return (ucls_mc(in, jn) * (32 * in + 64 * jn) / 64 + 64) * ucls_adj_pct(in, jn) / 100; }
constexpr int ucls_lds_max() {
  int m = 0;
  for (int in : {1, 2, 4, 6, 8})
    for (int jn = 1; jn <= 4; ++jn) m = ucls_lds(in, jn) > m ? ucls_lds(in, jn) : m;
  return m;
}

template <int IN_, int JN_>
struct UCls {
  static constexpr int IN = IN_, JN = JN_, WB = 4;
  static constexpr int NT = 512;
  static constexpr int MC = ucls_mc(IN, JN);
  static constexpr int TNW = 32 * IN, TKW = 64 * JN;
  typedef Stage<NT, TNW, MC> SA;
  typedef Stage<NT, TKW, MC> SB;
};

#ifdef QAVIT_TN_STAMPS         // diagnostic build only (tools/tn_stamps.py): per workgroup s_memtime at start / end, segments run, and time + cost per tile class
}  // namespace
}  // namespace qv
__device__ unsigned long long qv_tn_stamps[512 * 48];
extern "C" int qavit_tn_stamps(void* host_dst, int nwg) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(qv_tn_stamps), (size_t)nwg * 48 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
namespace qv {
namespace {
#define TNSTAMP_SET(k, v) do { if (threadIdx.x == 0) qv_tn_stamps[(size_t)(blockIdx.x & 511) * 48 + (k)] = (v); } while (0)
#define TNSTAMP_ADD(k, v) do { if (threadIdx.x == 0) qv_tn_stamps[(size_t)(blockIdx.x & 511) * 48 + (k)] += (v); } while (0)
#else
#define TNSTAMP_SET(k, v) do { } while (0)
#define TNSTAMP_ADD(k, v) do { } while (0)
#endif

// Preconditions (gemm_tn_wide checks them, other problems take the generic kernel): A, B 16-byte aligned,
// lda, ldb, N, K multiples of 8.  The staging loads are UNCONDITIONAL (row / column indices are clamped into the
// operand and the out-of-range vectors are zeroed when they are committed to LDS): a load under a lane-dependent
// branch makes the compiler wait for it at the join, which serialises the chunk's 16 loads into 16 round trips.
template <typename CL>
__device__ __forceinline__ void tn_wide_body(const qavit_gemm_tn_args& g, bf16* At, bf16* Bt, float* gb, int bx, int by, int mbeg, int mend, int dbg) {
  constexpr int IN = CL::IN, JN = CL::JN;
  constexpr int TNW = CL::TNW, TKW = CL::TKW, MC = CL::MC, NT = CL::NT;
  typedef typename CL::SA SA;
  typedef typename CL::SB SB;
  if (mbeg >= mend) return;                            // uniform per workgroup
  // the lane's staging / fragment geometry is recomputed per call ON PURPOSE: in the one-launch kernel twenty of these bodies sit in
  // one problem loop, and hoisting every body's lane-invariant index registers out of that loop (they depend on threadIdx only) cost
  // 150 spilled registers.  The empty volatile asm pins the computation to the call.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave & 1, wb = wave >> 1;
  const int n0 = bx * TNW, k0 = by * TKW;
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

  // LayerNorm-on-load: the tile's gamma / beta live in LDS (gb[0 .. 255] / gb[256 .. 511]), read back per staged vector -- sixteen
  // registers per lane that every body (normalising or not) used to carry through its MFMA loop
  float csum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) csum[j] = 0.f;
  if (ln && rgb == 0 && cgb < SB::CG) {                // visible after the first chunk's barrier; the previous body read gb before ITS last barrier
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gb[8 * cgb + j] = b_live ? gloadf(g.ln_gamma + kb + j) : 0.f;
      gb[256 + 8 * cgb + j] = b_live ? gloadf(g.ln_beta + kb + j) : 0.f;
    }
  }

  f32x4 acc[IN][JN];
#pragma unroll
  for (int i = 0; i < IN; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 pa[SA::PASS], pb[SB::PASS];
  float pmu[SB::PASS], prs[SB::PASS];
  // lane offsets inside a chunk (elements, 32 bits): a whole chunk's loads are then  uniform row pointer + lane offset  -- scalar address
  // arithmetic, no 64-bit multiply per load -- and need no row clamp; only a range's last, partial chunk takes the clamped form below
  const uint32_t offa = 2u * ((uint32_t)rga * (uint32_t)g.lda + (uint32_t)(a_live ? na : 0));      // BYTES: (uniform pointer) + zext(32-bit lane
  const uint32_t offb = 2u * ((uint32_t)rgb * (uint32_t)g.ldb + (uint32_t)(b_live ? kb : 0));      // offset) is the scalar-base load form
  auto prefetch = [&](int mc) {
    if (mc + MC <= mend) {                               // uniform
      const char* Ab = reinterpret_cast<const char*>(A + (size_t)mc * g.lda);
      const char* Bb = reinterpret_cast<const char*>(B + (size_t)mc * g.ldb);
#pragma unroll
      for (int h = 0; h < SA::PASS; ++h) pa[h] = *(const QV_AS1 bf16x8*)(Ab + (size_t)(SA::RP * h) * g.lda * 2 + offa);
#pragma unroll
      for (int h = 0; h < SB::PASS; ++h) pb[h] = *(const QV_AS1 bf16x8*)(Bb + (size_t)(SB::RP * h) * g.ldb * 2 + offb);
      if (ln) {                                          // uniform
        const uint32_t offs = 4u * (uint32_t)rgb;
#pragma unroll
        for (int h = 0; h < SB::PASS; ++h) {
            pmu[h] = *(const QV_AS1 float*)(reinterpret_cast<const char*>(mean_p + mc + SB::RP * h) + offs);
            prs[h] = *(const QV_AS1 float*)(reinterpret_cast<const char*>(rstd_p + mc + SB::RP * h) + offs);
          }
      }
      return;
    }
#pragma unroll
    for (int h = 0; h < SA::PASS; ++h) {
      int m = mc + rga + SA::RP * h;
      m = m < mend ? m : mend - 1;
      pa[h] = gload8(a_col + (size_t)m * g.lda);
    }
#pragma unroll
    for (int h = 0; h < SB::PASS; ++h) {
      int m = mc + rgb + SB::RP * h;
      m = m < mend ? m : mend - 1;
      pb[h] = gload8(b_col + (size_t)m * g.ldb);
      const int mi = ln ? m : 0;
      pmu[h] = gloadf(mean_p + mi);
      prs[h] = gloadf(rstd_p + mi);
    }
  };

  prefetch(mbeg);
  for (int mc = mbeg; mc < mend; mc += MC) {
#ifdef QAVIT_TN_STAMPS
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#else
    __syncthreads();
#endif
    // A whole chunk is committed as loaded: no row of it is past the range, and a column past N or K only feeds output rows / columns the
    // flush does not write (its lane loaded column group 0: finite values).  The per-element selects this path leaves out, with the
    // 64-bit address multiplies of the clamped prefetch, were 3/4 of the staging phase's VALU work, and staging was 37 % + 19 % (waiting
    // for the slowest wave of it) of the 192 x 192 class's time against 2 % waiting for memory (tools/tn_phases.py).
    const bool whole = mc + MC <= mend;                  // uniform
    if (whole) {
      if (cga < SA::CG) {
#pragma unroll
        for (int h = 0; h < SA::PASS; ++h) {
          *reinterpret_cast<bf16x8*>(At + (rga + SA::RP * h) * SA::LD + 8 * cga) = pa[h];
          if (want_csum) {                               // (as one more MFMA per A fragment against a vector of ones instead: the step's list 1.14-1.21 ms against 1.12)
            const u32x4 u = __builtin_bit_cast(u32x4, pa[h]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              csum[2 * q] += __uint_as_float(u[q] << 16);
              csum[2 * q + 1] += __uint_as_float(u[q] & 0xffff0000u);
            }
          }
        }
      }
      if (cgb < SB::CG) {
#pragma unroll
        for (int h = 0; h < SB::PASS; ++h) {
          bf16x8 v = pb[h];
          if (ln) {                                      // uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)(((float)v[j] - pmu[h]) * prs[h] * gb[8 * cgb + j] + gb[256 + 8 * cgb + j]);
          }
          *reinterpret_cast<bf16x8*>(Bt + (rgb + SB::RP * h) * SB::LD + 8 * cgb) = v;
        }
      }
    } else {
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
          for (int j = 0; j < 8; ++j) v[j] = (bf16)(((float)v[j] - pmu[h]) * prs[h] * gb[8 * cgb + j] + gb[256 + 8 * cgb + j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : (bf16)0.f;
        *reinterpret_cast<bf16x8*>(Bt + (rgb + SB::RP * h) * SB::LD + 8 * cgb) = v;
      }
    }
    }
    // the next chunk's loads go out as soon as the staging registers are free, BEFORE the barrier (a __syncthreads would drain them: the
    // barrier below waits for the LDS stores only) -- 3 % of the launch against prefetching behind the barrier.  (Spread over the k-steps
    // of the MFMA phase instead -- a chunk's 128 load instructions are ~2000 clocks of issue on the CU's 64 B/clock vector memory path --
    // they arrive too late: the mixed list of a step 1.28 ms against 1.10.)
    if (mc + MC < mend) prefetch(mc + MC);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef QAVIT_TN_STAMPS
    const unsigned long long s3 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef QAVIT_TN_STAMPS
    const unsigned long long s4 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int kf = 0; kf < MC / 32; ++kf) {
      // B fragments of the k-step up front, A fragments as the MFMAs consume them (the scheduler runs them ahead as far as registers allow:
      // holding all IN of them cost 16 registers the 256-wide classes do not have)
      bf16x8 bfr[JN];
#pragma unroll
      for (int j = 0; j < JN; ++j) bfr[j] = trf(Bt, SB::LD, kf * 32, (wb * JN + j) * 16);
#pragma unroll
      for (int i = 0; i < IN; ++i) {
        const bf16x8 af = trf(At, SA::LD, kf * 32, (wa * IN + i) * 16);
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
      }
    }
#ifdef QAVIT_TN_STAMPS        // wave 0's view: load wait | barrier A | staging (+ prefetch issue, LDS stores landed) | barrier B | MFMA phase (issue)
    TNSTAMP_ADD(44, s1 - s0); TNSTAMP_ADD(45, s2 - s1); TNSTAMP_ADD(46, s3 - s2); TNSTAMP_ADD(47, s4 - s3);
    TNSTAMP_ADD(3, (unsigned long long)__builtin_amdgcn_s_memtime() - s4);
#endif
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
        if (n < g.N && k < g.K) gatomic_add(g.C + (size_t)n * g.ldc + k, acc[i][j][r]);
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
      if (n0 + i < g.N) gatomic_add(g.colsum + n0 + i, t);
    }
    __syncthreads();                                 // At is staged again by the next range of this workgroup
  }
}

template <int IN, int JN, int WB, typename Src>
__device__ __forceinline__ void tn_wide_ranges(const Src& G, int n, int per, int dbg, bf16* At, bf16* Bt, float* gb) {
  typedef Cls<IN, JN, WB> CL;
  const int total = G.unit_start[n];
  int u = blockIdx.x * per;
  int uend = u + per;
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
    tn_wide_body<CL>(G.p[i], At, Bt, gb, bx, by, c0 * CL::MC, mend, dbg);
    u += c1 - c0;
  }
}

template <int IN, int JN, int WB>
__global__ __launch_bounds__(128 * WB) void gemm_tn_wide_kernel(TnwGroup G) {
  typedef Cls<IN, JN, WB> CL;
  __shared__ __attribute__((aligned(16))) bf16 At[CL::MC * CL::SA::LD];   // [m][n]
  __shared__ __attribute__((aligned(16))) bf16 Bt[CL::MC * CL::SB::LD];   // [m][k]
  __shared__ __attribute__((aligned(16))) float gb[512];
  tn_wide_ranges<IN, JN, WB>(G, G.n, G.per, G.dbg, At, Bt, gb);
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE launch for every problem of a backward pass.  A launch per tile class ends each workgroup's range in a tile-sized fp32-atomic
// flush (the chip retires about one atomic dword per L2 channel and clock: ~1.5 TB/s) and a C100 step needed 12-15 class launches
// x 256-512 workgroups x ~1.5 flushes = ~450 MB of atomics, a fifth of the family's time, plus a ramp and a tail per launch.  Here
// the (problem, tile, chunk) units of ALL classes form one list, weighted by the operand bytes a unit moves, cut into equal
// contiguous ranges over one workgroup per CU: ~330 flushes per step instead of ~5000.  The class of the problem a range is in picks
// the body (uniform switch); registers and LDS are those of the largest class, which ran at one workgroup per CU anyway.
constexpr int TNU_MAX = 512;            // problems per launch
constexpr int TNU_PER_WRITE = 28;       // entries a writer launch carries by value (< 4 KB of kernel arguments)

struct TnuEntry {
  qavit_gemm_tn_args p;
  int cls;                              // IN | JN << 8
  int tn, chunks;                       // tiles on the N side, chunks per tile
  int cu;                               // cost of one unit
  int tiles, cb;                        // tiles of the problem; chunks per block (multi-tile problems, see tnu_run), 0 = one tile
};
struct TnuTable {
  int cost_start[TNU_MAX + 1];          // prefix sums of units * cu
  TnuEntry e[TNU_MAX];
};
struct TnuWrite {
  int n, off;
  int cost_start[TNU_PER_WRITE + 1];
  TnuEntry e[TNU_PER_WRITE];
};

static_assert(sizeof(TnuWrite) <= 4096, "a writer launch carries its entries as kernel arguments");
__global__ __launch_bounds__(64) void tnu_table_write_kernel(TnuWrite W, TnuTable* T) {
  const int i = threadIdx.x;
  if (i < W.n) T->e[W.off + i] = W.e[i];
  if (i <= W.n) T->cost_start[W.off + i] = W.cost_start[i];
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename P>
__device__ __forceinline__ P* uni_ptr(P* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<P*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Unit order inside a problem.  One tile: its chunks in order.  Several tiles (an operand wider than the widest tile -- the lateral
// path's 256 <-> 1024 layers, the 768-column fuse, the 576-column qkv): BLOCKS of `cb` chunks, and inside a block tile after tile over the
// same rows, with cb chunks priced at one workgroup's range.  The workgroups that hold the tiles of one row block then walk the same rows at
// the same time: the operand every tile re-reads comes out of the Infinity Cache instead of HBM a second, third, fourth time, and the
// column slices the tiles take from one wide row are fetched together (tile after tile over ALL rows, as before, the slices of a row were
// read a quarter of a launch apart: 1.2x the algorithmic bytes by the counters, and those ranges were the slowest of the launch).
template <int IN, int JN>
__device__ __forceinline__ void tnu_run(const qavit_gemm_tn_args& g, char* smem, int tn, int chunks, int tiles, int cb, int j0, int j1, int dbg) {
  typedef UCls<IN, JN> CL;
  bf16* At = reinterpret_cast<bf16*>(smem);
  bf16* Bt = At + CL::MC * CL::SA::LD;
  while (j0 < j1) {                                     // uniform
    int t, c0, cend;
    if (cb > 0) {
      const int blku = tiles * cb, blk = j0 / blku, rem = j0 - blk * blku, cbase = blk * cb;
      const int cbl = chunks - cbase < cb ? chunks - cbase : cb;      // the last block is the short one
      t = rem / cbl; c0 = cbase + rem - t * cbl; cend = cbase + cbl;
    } else {
      t = j0 / chunks; c0 = j0 - t * chunks; cend = chunks;
    }
    int c1 = c0 + (j1 - j0);
    if (c1 > cend) c1 = cend;
    const int by = t / tn, bx = t - by * tn;
    const int mend = c1 * CL::MC < g.M ? c1 * CL::MC : g.M;
    tn_wide_body<CL>(g, At, Bt, reinterpret_cast<float*>(smem + ucls_lds_max()), bx, by, c0 * CL::MC, mend, dbg);
    j0 += c1 - c0;
  }
}


__global__ __launch_bounds__(512) void gemm_tn_uni_kernel(const TnuTable* __restrict__ T, int n, int per, int dbg) {
  __shared__ __attribute__((aligned(16))) char smem[ucls_lds_max() + 2048];        // + gamma / beta of the tile (tn_wide_body)
  static_assert(ucls_lds_max() % 16 == 0 && ucls_lds_max() + 2048 <= 160 * 1024, "LDS budget");
  const int total = uni(T->cost_start[n]);
  const int lo = blockIdx.x * per;
  int hi = lo + per;
  if (hi > total) hi = total;
#ifdef QAVIT_TN_STAMPS
  for (int k = 0; k < 48; ++k) TNSTAMP_SET(k, 0ull);
  TNSTAMP_SET(0, (unsigned long long)__builtin_amdgcn_s_memtime());
#endif
  if (lo >= hi) return;
  int a = 0, b = n;                                      // cost_start[a] <= lo < cost_start[b]
  while (b - a > 1) {
    const int m = (a + b) >> 1;
    if (uni(T->cost_start[m]) <= lo) a = m; else b = m;
  }
  for (int i = a; i < n; ++i) {
    const int cs = uni(T->cost_start[i]);
    if (cs >= hi) break;
    const TnuEntry* e = &T->e[i];
    const int cu = uni(e->cu), chunks = uni(e->chunks), tn = uni(e->tn), cls = uni(e->cls), tiles = uni(e->tiles), cb = uni(e->cb);
    const int units = (uni(T->cost_start[i + 1]) - cs) / cu;
    const int j0 = lo > cs ? (lo - cs + cu - 1) / cu : 0;
    int j1 = (hi - cs + cu - 1) / cu;
    if (j1 > units) j1 = units;
    if (j0 >= j1) continue;
    qavit_gemm_tn_args g;
    g.dtype = QAVIT_BF16;
    g.M = uni(e->p.M); g.N = uni(e->p.N); g.K = uni(e->p.K);
    g.A = uni_ptr(e->p.A); g.lda = uni64(e->p.lda);
    g.B = uni_ptr(e->p.B); g.ldb = uni64(e->p.ldb);
    g.C = uni_ptr(e->p.C); g.ldc = uni64(e->p.ldc);
    g.colsum = uni_ptr(e->p.colsum);
    g.ln_gamma = uni_ptr(e->p.ln_gamma); g.ln_beta = uni_ptr(e->p.ln_beta);
    g.ln_mean = uni_ptr(e->p.ln_mean); g.ln_rstd = uni_ptr(e->p.ln_rstd);
    g.splits = 0;
#define TNU_CASE(I, J) case (I | (J << 8)): tnu_run<I, J>(g, smem, tn, chunks, tiles, cb, j0, j1, dbg); break;
#ifdef QAVIT_TN_STAMPS
    const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#endif
    switch (cls) {
      TNU_CASE(1, 1) TNU_CASE(1, 2) TNU_CASE(1, 3) TNU_CASE(1, 4)
      TNU_CASE(2, 1) TNU_CASE(2, 2) TNU_CASE(2, 3) TNU_CASE(2, 4)
      TNU_CASE(4, 1) TNU_CASE(4, 2) TNU_CASE(4, 3) TNU_CASE(4, 4)
      TNU_CASE(6, 1) TNU_CASE(6, 2) TNU_CASE(6, 3) TNU_CASE(6, 4)
      TNU_CASE(8, 1) TNU_CASE(8, 2) TNU_CASE(8, 3) TNU_CASE(8, 4)
      default: break;
    }
#undef TNU_CASE
#ifdef QAVIT_TN_STAMPS
    {
      const int in = cls & 255, jn = cls >> 8;
      const int ci = (in == 1 ? 0 : in == 2 ? 1 : in == 4 ? 2 : in == 6 ? 3 : 4) * 4 + (jn - 1);
      TNSTAMP_ADD(4 + ci, (unsigned long long)__builtin_amdgcn_s_memtime() - t_in);
      TNSTAMP_ADD(24 + ci, (unsigned long long)((j1 - j0) * cu));
      TNSTAMP_ADD(2, 1ull);
    }
#endif
  }
  TNSTAMP_SET(1, (unsigned long long)__builtin_amdgcn_s_memtime());
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

int tile_class(int n, int which = 0) {           // 32-column units per tile: N side 1, 2, 4, 6, 8; K side 1, 2, 3, 4 on four waves, 6 / 8 = the 192- / 256-wide eight-wave classes
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
  // wider than one tile: the tile width that pads least; ties go to the wider tile (fewer re-reads of the other operand)
  int best = 4, best_pad = (n + 127) / 128 * 128;
  const int p192 = (n + 191) / 192 * 192, p256 = (n + 255) / 256 * 256;
  if (p192 <= best_pad) { best = 6; best_pad = p192; }
  if (p256 <= best_pad) { best = 8; best_pad = p256; }
  return best;
}

template <int IN, int JN, int WB>
void fill_group(TnwGroup& G, const qavit_gemm_tn_args* const* probs, int cnt, int& units) {
  typedef Cls<IN, JN, WB> CL;
  G.n = cnt;
  G.dbg = 0;
#ifdef QAVIT_TN_FLUSH_EXPERIMENT   // diagnostic build only (QAVIT_EXTRA_HIPCC_FLAGS=-DQAVIT_TN_FLUSH_EXPERIMENT): QAVIT_TN_DBG bit 0 skips the tile flush,
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("QAVIT_TN_DBG"); dbg = e ? atoi(e) : 0; } G.dbg = dbg; }   // bit 1 the column-sum flush -- WRONG results, timing only
#endif
  for (int i = 0; i < cnt; ++i) {
    const qavit_gemm_tn_args& g = *probs[i];
    G.p[i] = g;
    G.tn[i] = (g.N + CL::TNW - 1) / CL::TNW;
    G.chunks[i] = (g.M + CL::MC - 1) / CL::MC;
    G.unit_start[i] = units;
    units += G.tn[i] * ((g.K + CL::TKW - 1) / CL::TKW) * G.chunks[i];
  }
  for (int i = cnt; i <= TNW_GROUP; ++i) G.unit_start[i] = units;
}

// equal contiguous ranges over the resident workgroups; short launches use fewer, longer ranges (each range ends in a
// tile-sized atomic flush)
inline void plan_ranges(int units, int wb, int& per, int& wgs) {
  static int min_units = -1;
  if (min_units < 0) { const char* e = getenv("QAVIT_TN_MIN_UNITS"); min_units = e ? atoi(e) : 4; if (min_units < 1) min_units = 1; }
  wgs = resident_wgs(wb);
  if (units < wgs * min_units) wgs = (units + min_units - 1) / min_units;
  per = (units + wgs - 1) / wgs;
  wgs = (units + per - 1) / per;
}

template <int IN, int JN, int WB>
void launch_class(const qavit_gemm_tn_args* const* probs, int n, hipStream_t st) {
  typedef Cls<IN, JN, WB> CL;
  int done = 0;
  while (done < n) {
    const int cnt = (n - done < TNW_GROUP) ? (n - done) : TNW_GROUP;
    TnwGroup G;
    int units = 0, wgs;
    fill_group<IN, JN, WB>(G, probs + done, cnt, units);
    plan_ranges(units, WB, G.per, wgs);
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
    case 6: return launch_class<IN, 3, 4>;
    default: return launch_class<IN, 4, 4>;
  }
}
class_fn pick(int ic, int jc) {
  switch (ic) {
    case 1: return pick_j<1>(jc);
    case 2: return pick_j<2>(jc);
    case 4: return pick_j<4>(jc);
    case 6: return pick_j<6>(jc);
    default: return pick_j<8>(jc);
  }
}

}  // namespace

// All problems must be bf16 and validated by the caller.  Launches them grouped by tile class.
bool gemm_tn_wide_ok(const qavit_gemm_tn_args& g) {
  // K % 8 != 0 is fine when B's rows are padded to a multiple of 8 (the 16-byte loads stay inside the row; the products of the pad
  // columns land in output columns >= K, which the flush does not write) and nothing is normalised on load
  const bool k_ok = g.K % 8 == 0 || (!g.ln_mean && g.ldb >= (g.K + 7) / 8 * 8);
  return g.dtype == QAVIT_BF16 && g.N % 8 == 0 && k_ok && g.lda % 8 == 0 && g.ldb % 8 == 0 &&
         ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) == 0;
}

size_t gemm_tn_wide_ws_bytes() { return sizeof(TnuTable); }

namespace {
int uni_k_class(int k) {                 // 64-column units of the K-side tile (1 .. 4)
  if (k <= 64) return 1;
  if (k <= 128) return 2;
  if (k <= 192) return 3;
  if (k <= 256) return 4;
  int best = 2, best_pad = (k + 127) / 128 * 128;
  const int p192 = (k + 191) / 192 * 192, p256 = (k + 255) / 256 * 256;
  if (p192 <= best_pad) { best = 3; best_pad = p192; }
  if (p256 <= best_pad) { best = 4; best_pad = p256; }
  return best;
}

// every wide-eligible problem of the list in one launch through the table in `ws`; returns the number of problems it took
// Under stream capture the table is ONE copy node from a pinned host image instead of ceil(n / 28) writer launches (~6 us each, in front
// of the one big launch at the very end of the step).  A captured graph reads the image at every replay, so an image is never reused or
// freed; they come from a small pool allocated on a NON-capturing call (the eager warm-up steps every capture is preceded by) -- pinned
// allocation is not legal while a capture is open -- and a capture that finds the pool empty takes the writer launches.
constexpr int TNU_HOST_IMAGES = 16;
struct TnuHostPool { TnuTable* img[TNU_HOST_IMAGES]; int n, used; bool tried; };
static TnuHostPool& tnu_pool() { static TnuHostPool p{{}, 0, 0, false}; return p; }
static void tnu_pool_fill() {
  TnuHostPool& p = tnu_pool();
  if (p.tried) return;
  p.tried = true;
  for (int i = 0; i < TNU_HOST_IMAGES; ++i) {
    void* h = nullptr;
    if (hipHostMalloc(&h, sizeof(TnuTable), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); break; }
    p.img[p.n++] = reinterpret_cast<TnuTable*>(h);
  }
}

int gemm_tn_uni(const qavit_gemm_tn_args* a, int n, hipStream_t st, void* ws) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("QAVIT_TN_CLASS_LAUNCHES"); off = e ? atoi(e) : 0; }
  if (off || !ws) return 0;
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += gemm_tn_wide_ok(a[i]) ? 1 : 0;
  if (cnt < 2 || cnt > TNU_MAX) return 0;
  TnuTable* T = reinterpret_cast<TnuTable*>(ws);
  static int host_img = -1;
  if (host_img < 0) { const char* e = getenv("QAVIT_TN_HOST_TABLE"); host_img = e ? atoi(e) : 1; }
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap_st) != hipSuccess) { (void)hipGetLastError(); cap_st = hipStreamCaptureStatusNone; }
  const bool capturing = cap_st == hipStreamCaptureStatusActive;
  if (host_img && !capturing) tnu_pool_fill();
  TnuTable* H = nullptr;
  if (host_img && capturing && tnu_pool().used < tnu_pool().n) H = tnu_pool().img[tnu_pool().used];
  // pass 1: the launch's total cost -> the range one workgroup gets (the block length of the multi-tile problems is priced at it)
  int wgs = resident_wgs(4);
  static int cap = -1;                                  // QAVIT_TN_WGS: fewer workgroups than CUs (a launch that runs BESIDE other work)
  if (cap < 0) { const char* e = getenv("QAVIT_TN_WGS"); cap = e ? atoi(e) : 0; }
  if (cap > 0 && cap < wgs) wgs = cap;
  static int blocked = -1;                              // QAVIT_TN_BLOCKED=0: tile after tile over all rows (the round-3 order)
  if (blocked < 0) { const char* e = getenv("QAVIT_TN_BLOCKED"); blocked = e ? atoi(e) : 1; }
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    if (!gemm_tn_wide_ok(a[i])) continue;
    const qavit_gemm_tn_args& g = a[i];
    const int in = tile_class(g.N, 0), jn = uni_k_class(g.K), mc = ucls_mc(in, jn);
    total += (long long)((g.N + 32 * in - 1) / (32 * in)) * ((g.K + 64 * jn - 1) / (64 * jn)) * ((g.M + mc - 1) / mc) * ucls_cost(in, jn);
  }
  if (total > 0x7fffffffLL) return -1;                  // caller falls back to the class launches
  const int min_cost = 4 * ucls_cost(1, 1);             // a range shorter than a few small units is all flush
  int per = (int)((total + wgs - 1) / wgs);
  if (per < min_cost) per = min_cost;
  TnuWrite W;
  W.n = 0; W.off = 0;
  int cost = 0, done = 0;
  auto flush = [&]() {
    W.cost_start[W.n] = cost;
    if (H) {
      for (int i = 0; i < W.n; ++i) H->e[W.off + i] = W.e[i];
      for (int i = 0; i <= W.n; ++i) H->cost_start[W.off + i] = W.cost_start[i];
    } else {
      hipLaunchKernelGGL(tnu_table_write_kernel, dim3(1), dim3(64), 0, st, W, T);
    }
    W.off += W.n; W.n = 0;
  };
  for (int i = 0; i < n; ++i) {
    if (!gemm_tn_wide_ok(a[i])) continue;
    const qavit_gemm_tn_args& g = a[i];
    const int in = tile_class(g.N, 0), jn = uni_k_class(g.K);
    TnuEntry& e = W.e[W.n];
    e.p = g;
    e.cls = in | (jn << 8);
    e.tn = (g.N + 32 * in - 1) / (32 * in);
    const int mc = ucls_mc(in, jn);
    e.chunks = (g.M + mc - 1) / mc;
    e.cu = ucls_cost(in, jn);
    e.tiles = e.tn * ((g.K + 64 * jn - 1) / (64 * jn));
    e.cb = 0;
    if (e.tiles > 1 && blocked) {
      e.cb = (per + e.cu / 2) / e.cu;
      if (e.cb < 1) e.cb = 1;
      if (e.cb >= e.chunks) e.cb = e.chunks;             // one block: tile after tile, as before
    }
    W.cost_start[W.n] = cost;
    cost += e.tiles * e.chunks * e.cu;
    ++done;
    if (++W.n == TNU_PER_WRITE && done < cnt) flush();
  }
  flush();
  if (H) {
    // cost_start[0 .. cnt] and e[0 .. cnt) are all the kernel reads: two copy nodes (the arrays are 2 KB apart in the image when cnt is small)
    if (hipMemcpyAsync(T->cost_start, H->cost_start, sizeof(int) * (size_t)(cnt + 1), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(T->e, H->e, sizeof(TnuEntry) * (size_t)cnt, hipMemcpyHostToDevice, st) != hipSuccess)
      return set_error(QAVIT_ELAUNCH, "gemm_tn: copy node of the problem table");
    ++tnu_pool().used;
  }
  wgs = (cost + per - 1) / per;
  int dbg = 0;
#ifdef QAVIT_TN_FLUSH_EXPERIMENT
  { const char* e = getenv("QAVIT_TN_DBG"); dbg = e ? atoi(e) : 0; }
#endif
  hipLaunchKernelGGL(gemm_tn_uni_kernel, dim3(wgs), dim3(512), 0, st, (const TnuTable*)T, cnt, per, dbg);
  return cnt;
}
}  // namespace


int gemm_tn_wide(const qavit_gemm_tn_args* a, int n, hipStream_t st, void* ws) {
  static const int ncls[5] = {1, 2, 4, 6, 8}, kcls[6] = {1, 2, 3, 4, 6, 8};
  {
    const int rc = gemm_tn_uni(a, n, st, ws);
    if (rc > 0) return check_launch("gemm_tn(one launch)");
    if (rc < -1) return rc;
  }
  const qavit_gemm_tn_args* sel[256];
  // without a workspace (or with QAVIT_TN_CLASS_LAUNCHES=1): one launch per tile class and 24 problems
  for (int ci = 0; ci < 5; ++ci)
    for (int cj = 0; cj < 6; ++cj) {
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
