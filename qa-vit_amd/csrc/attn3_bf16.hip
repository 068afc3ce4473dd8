// bf16 attention core for the SMALL problems of the CIFAR configuration: <= 16 queries and <= 16 token keys per
// (group, head), with the key block boundary on a 16-row tile (Linformer rows / token rows first, then the shared
// bank rows).  Same math and operand images as attn_bf16.hip, restructured around what bounds problems this small
// -- memory latency and LDS footprint, not MFMA rate:
//   * a wave keeps ONE head for its whole life, so the head's shared bank rows (sh_k / sh_v) and the Linformer E
//     matrices are staged into LDS once, not once per problem;
//   * the next problem's q / dO / k / v rows are loaded into registers (unconditional loads, clamped indices) before
//     the current problem's MFMAs and committed to LDS afterwards, so HBM latency overlaps the matrix work;
//   * gradients that sum over problems (shared rows, dE) stay in accumulator registers for the wave's whole life
//     and go to the workspace once -- no fp32 LDS accumulators, and dKf/dVf reuse the Kf/Vf tiles, which cuts the
//     backward LDS slice from 44 KB to 22 KB (3 -> 7 resident waves per CU).
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include "frag16.cuh"

namespace qv {

namespace {

struct A3Lds { int q, p, kf, vf, kt, vt, ek, ev, d_o, ds, total; };

__host__ __device__ inline A3Lds a3_lds(int mode, bool bwd, int NK16, int D16, int KC) {
  A3Lds L;
  const int ldd = D16 + 4, ldk = NK16 + 4, lde = KC + 4;
  int o = 0;
  L.q = o; o += 16 * ldd;
  L.p = o; o += 16 * ldk;
  L.kf = o; o += NK16 * ldd;
  L.vf = o; o += NK16 * ldd;
  L.kt = L.vt = L.ek = L.ev = 0;
  if (mode == 0) {
    L.kt = o; o += 16 * ldd;
    L.vt = o; o += 16 * ldd;
    L.ek = o; o += 16 * lde;
    L.ev = o; o += 16 * lde;
  }
  L.d_o = L.ds = 0;
  if (bwd) {
    L.d_o = o; o += 16 * ldd;
    L.ds = o; o += 16 * ldk;
  }
  L.total = (o + 7) / 8 * 8;
  return L;
}

// 4 accumulator values = 4 consecutive elements of one output row: one 8-byte store
__device__ __forceinline__ void row4_to_lds(bf16* dst, const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  *reinterpret_cast<bf16x4*>(dst) = v;
}
__device__ __forceinline__ void row4_to_global(bf16* dst, const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  *reinterpret_cast<bf16x4*>(dst) = v;
}

__device__ __forceinline__ bool any_nan4(bf16x4 v) {
  return ((float)v[0] != (float)v[0]) | ((float)v[1] != (float)v[1]) | ((float)v[2] != (float)v[2]) | ((float)v[3] != (float)v[3]);
}

// KT0 = number of 16-row key tiles BEFORE the shared rows (Linformer rows in mode 0, token rows in mode 1)
template <int MODE, int NKT, int DT, int KT0, bool BWD>
__global__ __launch_bounds__(64) void attn3_kernel(qavit_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  constexpr int D16 = DT * 16, NK16 = NKT * 16, LDD = D16 + 4, LDK = NK16 + 4, NKo = KT0 * 16;
  const int LDE = a.KC + 4;
  const A3Lds L = a3_lds(MODE, BWD, NK16, D16, a.KC);
  const int lane = threadIdx.x, col = lane & 15, q4 = lane >> 4;
  const int D = a.D, DC = D >> 2, H = a.H;
  const int NK = NKo + a.S;
  const int h = blockIdx.x % H;
  const float scale = rsqrtf((float)D);
  const bf16* qg = reinterpret_cast<const bf16*>(a.q);
  const bf16* ktg = reinterpret_cast<const bf16*>(a.k_tok);
  const bf16* vtg = reinterpret_cast<const bf16*>(a.v_tok);
  bf16* og = reinterpret_cast<bf16*>(a.o);
  const bf16* dog = reinterpret_cast<const bf16*>(a.d_o);
  bf16* dqg = reinterpret_cast<bf16*>(a.dq);
  bf16* dktg = reinterpret_cast<bf16*>(a.dk_tok);
  bf16* dvtg = reinterpret_cast<bf16*>(a.dv_tok);
  bool bad = false;
  const AttnDrop drop = attn_drop_init(a);

  for (int i = lane; i < L.total; i += 64) sm[i] = (bf16)0.f;
  wave_sync();
  // ---------------- per-head constants: shared bank rows, Linformer matrices ----------------
  for (int i = lane; i < a.S * DC; i += 64) {
    const int s = i / DC, ch = i - s * DC;
    const f32x4 k = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)s * H * D + h * D + 4 * ch);
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)s * H * D + h * D + 4 * ch);
    bf16x4 kb, vb;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bad |= (k[j] != k[j]) | (v[j] != v[j]); kb[j] = (bf16)k[j]; vb[j] = (bf16)v[j]; }
    *reinterpret_cast<bf16x4*>(sm + L.kf + (NKo + s) * LDD + 4 * ch) = kb;
    *reinterpret_cast<bf16x4*>(sm + L.vf + (NKo + s) * LDD + 4 * ch) = vb;
  }
  if (MODE == 0) {
    const int EC = a.KC >> 2;
    for (int i = lane; i < a.L * EC; i += 64) {
      const int l = i / EC, ch = i - l * EC;
      const f32x4 ek = *reinterpret_cast<const f32x4*>(a.E_k + (size_t)l * a.KC + 4 * ch);
      const f32x4 ev = *reinterpret_cast<const f32x4*>(a.E_v + (size_t)l * a.KC + 4 * ch);
      bf16x4 kb, vb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { kb[j] = (bf16)ek[j]; vb[j] = (bf16)ev[j]; }
      *reinterpret_cast<bf16x4*>(sm + L.ek + l * LDE + 4 * ch) = kb;
      *reinterpret_cast<bf16x4*>(sm + L.ev + l * LDE + 4 * ch) = vb;
    }
  }

  // accumulators that live across problems: shared-row gradients (key tiles >= KT0) and dE
  f32x4 gK[NKT][DT], gV[NKT][DT], eK[KT0 > 0 ? KT0 : 1], eV[KT0 > 0 ? KT0 : 1];
  if (BWD) {
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
      for (int j = 0; j < DT; ++j) { gK[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; gV[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < (KT0 > 0 ? KT0 : 1); ++i) { eK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; eV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  const int total = a.G * H;
  const bool has_k = a.L > 0;                            // uniform
  bf16x4 pq[DT], pdo[DT], pk[DT], pv[DT];
  // rows of problem `pid` -> registers.  Every load is unconditional: out-of-range chunks re-read chunk 0 and are
  // zeroed at commit (a load under a lane-dependent branch would be waited for at the join).
  auto prefetch = [&](int pid) {
    const int g = pid / H;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      const int i = lane + 64 * k;
      const int ii = i < a.Nq * DC ? i : 0;
      const int r = ii / DC, ch = ii - r * DC;
      const int64_t qr = attn_qrow(a, g, r);
      pq[k] = *reinterpret_cast<const bf16x4*>(qg + qr * a.ldq + h * D + 4 * ch);
      if (BWD) pdo[k] = *reinterpret_cast<const bf16x4*>(dog + qr * a.lddo + h * D + 4 * ch);
    }
    if (has_k) {
#pragma unroll
      for (int k = 0; k < DT; ++k) {
        const int i = lane + 64 * k;
        const int ii = i < a.L * DC ? i : 0;
        const int r = ii / DC, ch = ii - r * DC;
        const int64_t kr = attn_krow(a, g, r);
        pk[k] = *reinterpret_cast<const bf16x4*>(ktg + kr * a.ldk + h * D + 4 * ch);
        pv[k] = *reinterpret_cast<const bf16x4*>(vtg + kr * a.ldv + h * D + 4 * ch);
      }
    }
  };

  int pid = blockIdx.x;
  if (pid < total) prefetch(pid);
  for (; pid < total; pid += gridDim.x) {
    const int g = pid / H;
    wave_sync();                                         // the previous problem's LDS reads are done
    // ---------------- commit the prefetched rows ----------------
    const bf16x4 zero4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
#pragma unroll
    for (int k = 0; k < DT; ++k) {
      const int i = lane + 64 * k;
      if (i < 16 * DC) {
        const int r = i / DC, ch = i - r * DC;
        const bf16x4 v = (i < a.Nq * DC) ? pq[k] : zero4;
        if (!BWD) bad |= any_nan4(v);
        *reinterpret_cast<bf16x4*>(sm + L.q + r * LDD + 4 * ch) = v;
        if (BWD) *reinterpret_cast<bf16x4*>(sm + L.d_o + r * LDD + 4 * ch) = (i < a.Nq * DC) ? pdo[k] : zero4;
        if (has_k) {
          const bool ok = i < a.L * DC;
          const bf16x4 kk = ok ? pk[k] : zero4, vv = ok ? pv[k] : zero4;
          if (!BWD) bad |= any_nan4(kk) | any_nan4(vv);
          *reinterpret_cast<bf16x4*>(sm + (MODE == 0 ? L.kt : L.kf) + r * LDD + 4 * ch) = kk;
          *reinterpret_cast<bf16x4*>(sm + (MODE == 0 ? L.vt : L.vf) + r * LDD + 4 * ch) = vv;
        }
      }
    }
    {
      const int np = pid + gridDim.x;
      prefetch(np < total ? np : pid);                   // always issued: the last round re-reads its own rows
    }
    wave_sync();
    // Every product below is formed TRANSPOSED (operands of the MFMA swapped): a lane's 4 accumulator values are then 4
    // consecutive elements of ONE output row, so each accumulator tile leaves as one 8-byte store (LDS or global)
    // instead of four 2-byte ones, and the softmax reductions run in-lane plus two cross-group shuffles.
    // Accumulator element r of a tile: row = 4*q4 + r of the tile (the output's column index), col = lane & 15.
    if (MODE == 0) {
      // Kf^T[d][j] = sum_l kt[l][d] E_k[l][j]
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
          ak = mma16(trfrag(sm + L.kt, LDD, 0, dt * 16), trfrag(sm + L.ek, LDE, 0, jt * 16), ak);
          av = mma16(trfrag(sm + L.vt, LDD, 0, dt * 16), trfrag(sm + L.ev, LDE, 0, jt * 16), av);
          row4_to_lds(sm + L.kf + (jt * 16 + col) * LDD + dt * 16 + 4 * q4, ak);
          row4_to_lds(sm + L.vf + (jt * 16 + col) * LDD + dt * 16 + 4 * q4, av);
        }
      wave_sync();
    }
    // ---------------- S^T[key][query], softmax over keys on registers ----------------
    f32x4 s[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        acc = mma16(rowfrag(sm + L.kf, LDD, nt * 16, dt * 16), rowfrag(sm + L.q, LDD, 0, dt * 16), acc);
      s[nt] = acc;
    }
    {
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = nt * 16 + 4 * q4 + r < NK;
          s[nt][r] = ok ? s[nt][r] * scale : -INFINITY;
          mx = fmaxf(mx, s[nt][r]);
        }
      mx = rows4_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(s[nt][r] - mx); s[nt][r] = e; sum += e; }
      sum = rows4_sum(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[nt][r] *= inv;
      }
    }
    // dropout on the probabilities: the P tile in LDS (operand of P.V and of dVf) holds P*m, the registers keep P
    f32x4 dm[NKT];
    if (drop.on) {
      const uint32_t pkey = attn_drop_pkey(drop, pid);
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
        f32x4 pd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dm[nt][r] = attn_drop_factor(drop, pkey, col, nt * 16 + 4 * q4 + r);
          pd[r] = s[nt][r] * dm[nt][r];
        }
        row4_to_lds(sm + L.p + col * LDK + nt * 16 + 4 * q4, pd);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) row4_to_lds(sm + L.p + col * LDK + nt * 16 + 4 * q4, s[nt]);
    }
    wave_sync();
    const int64_t my_q = attn_qrow(a, g, col < a.Nq ? col : 0);     // this lane's query row (col) in the global matrices
    if (!BWD) {
      // O^T[d][query] = sum_key Vf[key][d] P[query][key]
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
          acc = mma16(trfrag(sm + L.vf, LDD, nt * 16, dt * 16), rowfrag(sm + L.p, LDK, 0, nt * 16), acc);
        bad |= (acc[0] != acc[0]) | (acc[1] != acc[1]) | (acc[2] != acc[2]) | (acc[3] != acc[3]);
        if (col < a.Nq && dt * 16 + 4 * q4 < D) row4_to_global(og + my_q * a.ldo + h * D + dt * 16 + 4 * q4, acc);
      }
    } else {
      // dP^T[key][query] = sum_d Vf[key][d] dO[query][d] ; dS = P * (dP - rowdot) * scale
      f32x4 dp[NKT];
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          acc = mma16(rowfrag(sm + L.vf, LDD, nt * 16, dt * 16), rowfrag(sm + L.d_o, LDD, 0, dt * 16), acc);
        dp[nt] = acc;
      }
      if (drop.on) {
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[nt][r] *= dm[nt][r];
      }
      {
        float dot = 0.f;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dot += s[nt][r] * dp[nt][r];
        dot = rows4_sum(dot);
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[nt][r] = s[nt][r] * (dp[nt][r] - dot) * scale;
          row4_to_lds(sm + L.ds + col * LDK + nt * 16 + 4 * q4, dp[nt]);
        }
      }
      wave_sync();
      // per-problem key tiles start from zero; the shared-row tiles (nt >= KT0) keep accumulating
#pragma unroll
      for (int nt = 0; nt < KT0; ++nt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { gK[nt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; gV[nt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      // dQ^T[d][query] = sum_key Kf[key][d] dS[query][key] ; gK^T[d][key] += sum_query Q[query][d] dS[query][key] ;
      // gV^T[d][key] += sum_query dO[query][d] P[query][key]
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
          acc = mma16(trfrag(sm + L.kf, LDD, nt * 16, dt * 16), rowfrag(sm + L.ds, LDK, 0, nt * 16), acc);
        if (col < a.Nq && dt * 16 + 4 * q4 < D) row4_to_global(dqg + my_q * a.lddq + h * D + dt * 16 + 4 * q4, acc);
        const s16x4 aq = trfrag(sm + L.q, LDD, 0, dt * 16);
        const s16x4 ao = trfrag(sm + L.d_o, LDD, 0, dt * 16);
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
          gK[nt][dt] = mma16(aq, trfrag(sm + L.ds, LDK, 0, nt * 16), gK[nt][dt]);
          gV[nt][dt] = mma16(ao, trfrag(sm + L.p, LDK, 0, nt * 16), gV[nt][dt]);
        }
      }
      if (KT0 > 0) {
        if (MODE == 1) {
          // token keys: lane = key row (col), 4 consecutive channels per tile
#pragma unroll
          for (int nt = 0; nt < KT0; ++nt) {
            const int key = nt * 16 + col;
            const int64_t kr = attn_krow(a, g, key < a.L ? key : 0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
              if (key < a.L && dt * 16 + 4 * q4 < D) {
                row4_to_global(dktg + kr * a.lddk + h * D + dt * 16 + 4 * q4, gK[nt][dt]);
                row4_to_global(dvtg + kr * a.lddv + h * D + dt * 16 + 4 * q4, gV[nt][dt]);
              }
          }
        } else {
          // Linformer rows: dKf / dVf take over the (now dead) per-problem rows of the Kf / Vf tiles
          wave_sync();
#pragma unroll
          for (int nt = 0; nt < KT0; ++nt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              row4_to_lds(sm + L.kf + (nt * 16 + col) * LDD + dt * 16 + 4 * q4, gK[nt][dt]);
              row4_to_lds(sm + L.vf + (nt * 16 + col) * LDD + dt * 16 + 4 * q4, gV[nt][dt]);
            }
          wave_sync();
          // dk_tok^T[d][l] = sum_j dKf[j][d] E_k[l][j]
          const int64_t kr = attn_krow(a, g, col < a.L ? col : 0);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < KT0; ++jt) {
              ak = mma16(trfrag(sm + L.kf, LDD, jt * 16, dt * 16), rowfrag(sm + L.ek, LDE, 0, jt * 16), ak);
              av = mma16(trfrag(sm + L.vf, LDD, jt * 16, dt * 16), rowfrag(sm + L.ev, LDE, 0, jt * 16), av);
            }
            if (col < a.L && dt * 16 + 4 * q4 < D) {
              row4_to_global(dktg + kr * a.lddk + h * D + dt * 16 + 4 * q4, ak);
              row4_to_global(dvtg + kr * a.lddv + h * D + dt * 16 + 4 * q4, av);
            }
          }
          // dE_k[l][j] += sum_d kt[l][d] dKf[j][d]
#pragma unroll
          for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              eK[jt] = mma16(rowfrag(sm + L.kt, LDD, 0, dt * 16), rowfrag(sm + L.kf, LDD, jt * 16, dt * 16), eK[jt]);
              eV[jt] = mma16(rowfrag(sm + L.vt, LDD, 0, dt * 16), rowfrag(sm + L.vf, LDD, jt * 16, dt * 16), eV[jt]);
            }
        }
      }
    }
  }
  if (BWD) {
    // one partial per wave, in the layout attn_reduce_kernel folds: [dE_k L*KC][dE_v L*KC][dsh_k S*D][dsh_v S*D]
    const int nE = (MODE == 0) ? a.L * a.KC : 0, nS = a.S * D;
    float* ws = a.ws + (size_t)blockIdx.x * (2 * nE + 2 * nS);
    if (MODE == 0) {
#pragma unroll
      for (int jt = 0; jt < KT0; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int l = 4 * q4 + r, j = jt * 16 + col;
          if (l < a.L && j < a.KC) { ws[l * a.KC + j] = eK[jt][r]; ws[nE + l * a.KC + j] = eV[jt][r]; }
        }
    }
#pragma unroll
    for (int nt = KT0; nt < NKT; ++nt)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int srow = (nt - KT0) * 16 + col, cc = dt * 16 + 4 * q4 + r;      // transposed tiles: rows = channels, cols = keys
          if (srow < a.S && cc < D) {
            ws[2 * nE + srow * D + cc] = gK[nt][dt][r];
            ws[2 * nE + nS + srow * D + cc] = gV[nt][dt][r];
          }
        }
  } else {
    if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
  }
}

template <int MODE, int NKT, int DT, int KT0>
int a3_launch(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st) {
  const A3Lds L = a3_lds(MODE, bwd, NKT * 16, DT * 16, a.KC);
  const size_t bytes = (size_t)L.total * 2;
  if (bytes > 64 * 1024) return -100;
  if (bwd) hipLaunchKernelGGL((attn3_kernel<MODE, NKT, DT, KT0, true>), dim3(grid), dim3(64), bytes, st, a);
  else hipLaunchKernelGGL((attn3_kernel<MODE, NKT, DT, KT0, false>), dim3(grid), dim3(64), bytes, st, a);
  return QAVIT_OK;
}

}  // namespace

// 1 = launched, 0 = shape not covered.  The caller (attn_bf16_try) has already checked alignment and strides.
int attn3_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st) {
  if (a.Nq > 16 || a.L > 16 || a.S > 16 || a.S <= 0 || a.D % 4) return 0;
  // 8-byte row-segment stores of the outputs
  auto al8 = [](const void* p, int64_t ld) { return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 7) == 0 && ld % 4 == 0); };
  if (!bwd && !al8(a.o, a.ldo)) return 0;
  if (bwd && (!al8(a.dq, a.lddq) || (a.L > 0 && (!al8(a.dk_tok, a.lddk) || !al8(a.dv_tok, a.lddv))))) return 0;
  int rc = -100;
  if (a.mode == 0 && a.KC == 32 && a.L > 0 && a.D > 32 && a.D <= 48) rc = a3_launch<0, 3, 3, 2>(a, bwd, grid, st);
  else if (a.mode == 1 && a.L == 0 && a.D > 32 && a.D <= 48) rc = a3_launch<1, 1, 3, 0>(a, bwd, grid, st);
  else if (a.mode == 1 && a.L == 16 && a.D <= 16) rc = a3_launch<1, 2, 1, 1>(a, bwd, grid, st);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}

}  // namespace qv
