// bf16 attention core for problems with several 16-query tiles (Nq >= 32: Tiny-ImageNet's 64 learned tokens, QA-ViT at
// 32 px without TokenLearner, the 224-px models).  Same math and tile idioms as attn_bf16.hip, but a (group, head)
// problem belongs to a WORKGROUP of four waves instead of one wave:
//   * the key side (bank rows, token rows, Linformer E, and E^T K / E^T V) is staged / computed once by all 256 threads;
//   * the query tiles are split over the waves -- each wave has its own q / P / dO / dS tiles and needs only wave-level
//     ordering inside its tile loop -- so a 64-query problem runs its four tiles concurrently;
//   * backward: the waves' dKf / dVf accumulators meet in an fp32 LDS tile -- wave 0 stores, waves 1..3 add in turn between
//     barriers (plain ds_read / ds_write: LDS float atomics run a lane at a time and made the first version 5x slower
//     than the one-wave kernel) -- are rounded once to the bf16 operands of the Linformer products (which reuse the
//     Kf / Vf tiles), and those products are split over the waves.
// Partial sums of the bank-row and Linformer-matrix gradients leave as one slice of the workspace per workgroup, folded
// by attn_reduce_kernel (attn.hip) exactly as for the one-wave kernels.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include <stdlib.h>
#include "attn_shared.h"
#include "frag16.cuh"

namespace qv {

namespace {

// 4 accumulator values of a TRANSPOSED product = 4 consecutive elements of one output row: one 8-byte store
__device__ __forceinline__ bf16x4 row4(const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  return v;
}

struct A4Lds {   // offsets in bf16 elements unless noted
  int ldd, ldk, lde;
  int kf, vf, kt, vt, ek, ev;          // shared by the workgroup
  int wq, wp, wdo, wds, wstride;       // per-wave tiles: base + wave * wstride
  int end16;                           // bf16 elements in use (zeroed once)
  int rk, rv, accs;                    // float offsets (from the float view of the base)
  int bytes;
};

__host__ __device__ inline A4Lds a4_lds(const qavit_attn_args& a, bool bwd, int NK16, int D16) {
  A4Lds L;
  const int L16 = (a.L + 15) / 16 * 16;
  L.ldd = D16 + 4; L.ldk = NK16 + 4; L.lde = a.KC + 4;
  int o = 0;
  L.kf = o; o += NK16 * L.ldd;
  L.vf = o; o += NK16 * L.ldd;
  L.kt = L.vt = L.ek = L.ev = 0;
  if (a.mode == 0) {
    L.kt = o; o += L16 * L.ldd;
    L.vt = o; o += L16 * L.ldd;
    L.ek = o; o += L16 * L.lde;
    L.ev = o; o += L16 * L.lde;
  }
  int w = 0;
  L.wq = w; w += 16 * L.ldd;
  L.wp = w; w += 16 * L.ldk;
  L.wdo = L.wds = 0;
  if (bwd) { L.wdo = w; w += 16 * L.ldd; L.wds = w; w += 16 * L.ldk; }
  L.wstride = w;
  L.wq += o; L.wp += o; L.wdo += o; L.wds += o;
  o += 4 * w;
  L.end16 = o;
  o = (o + 7) / 8 * 8;
  int f = o / 2;
  L.rk = L.rv = L.accs = f;
  if (bwd) {
    L.rk = f; f += NK16 * (D16 + 4);          // row stride D16 + 4: the four 4-row groups of a tile hit different banks
    L.rv = f; f += NK16 * (D16 + 4);
    L.accs = f; f += 2 * a.S * a.D;
  }
  L.bytes = f * 4;
  return L;
}

template <int MODE, int NKT, int DT, bool BWD>
__global__ __launch_bounds__(256) void attn4_kernel(qavit_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* smf = reinterpret_cast<float*>(smraw);
  constexpr int NK16 = NKT * 16, D16 = DT * 16;
  const A4Lds L = a4_lds(a, BWD, NK16, D16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q4 = lane >> 4;
  const int D = a.D;
  const int NKo = (MODE == 0) ? a.KC : a.L;
  const int NK = NKo + a.S;
  const int L16 = (a.L + 15) / 16 * 16;
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

  const int nE = (MODE == 0) ? a.L * a.KC : 0;
  const int nS = a.S * D;
  float* Rk = smf + L.rk;
  float* Rv = smf + L.rv;
  float* accSk = smf + L.accs;
  float* accSv = accSk + nS;
  float* wsE = BWD ? a.ws + (size_t)blockIdx.x * (2 * nE + 2 * nS) : nullptr;    // [dE_k | dE_v | dsh_k | dsh_v]
  bf16* Wq = sm + L.wq + wave * L.wstride;
  bf16* Wp = sm + L.wp + wave * L.wstride;
  bf16* Wdo = sm + L.wdo + wave * L.wstride;
  bf16* Wds = sm + L.wds + wave * L.wstride;

  for (int i = tid; i < L.end16; i += 256) sm[i] = (bf16)0.f;     // padded rows / columns must read as 0 in every product
  if (BWD) for (int i = tid; i < 2 * nS; i += 256) accSk[i] = 0.f;

  const int DC = D >> 2;                                 // 4-element chunks per row (D % 4 == 0, checked on the host)
  for (int pid = blockIdx.x; pid < a.G * a.H; pid += gridDim.x) {
    const int g = pid / a.H, h = pid - g * a.H;
    __syncthreads();
    // ---------------- key side, all four waves ----------------
    for (int i = tid; i < a.S * DC; i += 256) {
      const int s = i / DC, ch = i - s * DC;
      const f32x4 k = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)s * a.H * D + h * D + 4 * ch);
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)s * a.H * D + h * D + 4 * ch);
      bf16x4 kb, vb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { bad |= (k[j] != k[j]) | (v[j] != v[j]); kb[j] = (bf16)k[j]; vb[j] = (bf16)v[j]; }
      *reinterpret_cast<bf16x4*>(sm + L.kf + (NKo + s) * L.ldd + 4 * ch) = kb;
      *reinterpret_cast<bf16x4*>(sm + L.vf + (NKo + s) * L.ldd + 4 * ch) = vb;
    }
    {
      const int kdst = (MODE == 0) ? L.kt : L.kf, vdst = (MODE == 0) ? L.vt : L.vf;
      for (int i = tid; i < a.L * DC; i += 256) {
        const int l = i / DC, ch = i - l * DC;
        const int64_t kr = attn_krow(a, g, l);
        const bf16x4 k = *reinterpret_cast<const bf16x4*>(ktg + kr * a.ldk + h * D + 4 * ch);
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(vtg + kr * a.ldv + h * D + 4 * ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= ((float)k[j] != (float)k[j]) | ((float)v[j] != (float)v[j]);
        *reinterpret_cast<bf16x4*>(sm + kdst + l * L.ldd + 4 * ch) = k;
        *reinterpret_cast<bf16x4*>(sm + vdst + l * L.ldd + 4 * ch) = v;
      }
    }
    if (MODE == 0) {
      const int EC = a.KC >> 2;
      for (int i = tid; i < a.L * EC; i += 256) {
        const int l = i / EC, ch = i - l * EC;
        const f32x4 ek = *reinterpret_cast<const f32x4*>(a.E_k + (size_t)l * a.KC + 4 * ch);
        const f32x4 ev = *reinterpret_cast<const f32x4*>(a.E_v + (size_t)l * a.KC + 4 * ch);
        bf16x4 kb, vb;
#pragma unroll
        for (int j = 0; j < 4; ++j) { kb[j] = (bf16)ek[j]; vb[j] = (bf16)ev[j]; }
        *reinterpret_cast<bf16x4*>(sm + L.ek + l * L.lde + 4 * ch) = kb;
        *reinterpret_cast<bf16x4*>(sm + L.ev + l * L.lde + 4 * ch) = vb;
      }
      __syncthreads();
      // Kf[j][d] = sum_l E_k[l][j] kt[l][d]: the (KC/16) x DT output tiles are dealt to the waves
      const int njt = a.KC >> 4;
      for (int t = wave; t < njt * DT; t += 4) {
        const int jt = t / DT, dt = t - jt * DT;
        f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
        for (int l0 = 0; l0 < L16; l0 += 16) {
          ak = mma16(trfrag(sm + L.ek, L.lde, l0, jt * 16), trfrag(sm + L.kt, L.ldd, l0, dt * 16), ak);
          av = mma16(trfrag(sm + L.ev, L.lde, l0, jt * 16), trfrag(sm + L.vt, L.ldd, l0, dt * 16), av);
        }
        acc_to_lds(sm + L.kf, L.ldd, jt * 16, dt * 16, ak);
        acc_to_lds(sm + L.vf, L.ldd, jt * 16, dt * 16, av);
      }
    }
    __syncthreads();

    // ---------------- query tiles: wave w takes tiles w, w+4, ... ----------------
    f32x4 gK[NKT][DT], gV[NKT][DT];
    if (BWD) {
#pragma unroll
      for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int j = 0; j < DT; ++j) { gK[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; gV[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    bool any_tile = false;
    for (int q0 = 16 * wave; q0 < a.Nq; q0 += 64) {
      any_tile = true;
      const int rows = (a.Nq - q0 < 16) ? a.Nq - q0 : 16;
      wave_sync();
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        const int i = lane + 64 * c;
        if (i < 16 * DC) {
          const int r = i / DC, ch = i - r * DC;
          const int rc = r < rows ? r : rows - 1;
          const int64_t qr = attn_qrow(a, g, q0 + rc);
          bf16x4 v = *reinterpret_cast<const bf16x4*>(qg + qr * a.ldq + h * D + 4 * ch), gvv;
#pragma unroll
          for (int j = 0; j < 4; ++j) gvv[j] = (bf16)0.f;
          if (BWD) gvv = *reinterpret_cast<const bf16x4*>(dog + qr * a.lddo + h * D + 4 * ch);
          if (r >= rows) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = (bf16)0.f; gvv[j] = (bf16)0.f; }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) bad |= ((float)v[j] != (float)v[j]);
          *reinterpret_cast<bf16x4*>(Wq + r * L.ldd + 4 * ch) = v;
          if (BWD) *reinterpret_cast<bf16x4*>(Wdo + r * L.ldd + 4 * ch) = gvv;
        }
      }
      wave_sync();
      // scores + softmax on registers
      f32x4 s[NKT];
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          acc = mma16(rowfrag(Wq, L.ldd, 0, dt * 16), rowfrag(sm + L.kf, L.ldd, nt * 16, dt * 16), acc);
        s[nt] = acc;
      }
      float inv_sum[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
          const bool ok = nt * 16 + col < NK;
          s[nt][r] = ok ? s[nt][r] * scale : -INFINITY;
          mx = fmaxf(mx, s[nt][r]);
        }
        mx = grp_max<16>(mx);
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) { const float e = __expf(s[nt][r] - mx); s[nt][r] = e; sum += e; }
        sum = grp_sum<16>(sum);
        inv_sum[r] = 1.f / sum;
      }
      // dropout on the probabilities: the P tile in LDS (operand of P.V and of dVf) holds P*m, the registers keep P
      f32x4 dm[NKT];
      const uint32_t pkey = drop.on ? attn_drop_pkey(drop, pid) : 0u;
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[nt][r] *= inv_sum[r];
        if (drop.on) {
          f32x4 pd;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dm[nt][r] = attn_drop_factor(drop, pkey, q0 + 4 * q4 + r, nt * 16 + col);
            pd[r] = s[nt][r] * dm[nt][r];
          }
          acc_to_lds(Wp, L.ldk, 0, nt * 16, pd);
        } else {
          acc_to_lds(Wp, L.ldk, 0, nt * 16, s[nt]);
        }
      }
      wave_sync();
      const int64_t my_q = attn_qrow(a, g, q0 + (col < rows ? col : 0));      // this lane's query row in the global matrices
      if (!BWD) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt)       // O^T[d][query]: this lane = query row `col`, 4 consecutive d
            acc = mma16(trfrag(sm + L.vf, L.ldd, nt * 16, dt * 16), rowfrag(Wp, L.ldk, 0, nt * 16), acc);
          bad |= (acc[0] != acc[0]) | (acc[1] != acc[1]) | (acc[2] != acc[2]) | (acc[3] != acc[3]);
          if (col < rows && dt * 16 + 4 * q4 < D)
            *reinterpret_cast<bf16x4*>(og + my_q * a.ldo + h * D + dt * 16 + 4 * q4) = row4(acc);
        }
      } else {
        f32x4 dp[NKT];
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            acc = mma16(rowfrag(Wdo, L.ldd, 0, dt * 16), rowfrag(sm + L.vf, L.ldd, nt * 16, dt * 16), acc);
          dp[nt] = acc;
        }
        if (drop.on) {
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[nt][r] *= dm[nt][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dot = 0.f;
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt) dot += s[nt][r] * dp[nt][r];
          dot = grp_sum<16>(dot);
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt) dp[nt][r] = s[nt][r] * (dp[nt][r] - dot) * scale;
        }
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) acc_to_lds(Wds, L.ldk, 0, nt * 16, dp[nt]);
        wave_sync();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt)       // dQ^T[d][query]
            acc = mma16(trfrag(sm + L.kf, L.ldd, nt * 16, dt * 16), rowfrag(Wds, L.ldk, 0, nt * 16), acc);
          if (col < rows && dt * 16 + 4 * q4 < D)
            *reinterpret_cast<bf16x4*>(dqg + my_q * a.lddq + h * D + dt * 16 + 4 * q4) = row4(acc);
          const s16x4 bq = trfrag(Wq, L.ldd, 0, dt * 16);
          const s16x4 bo = trfrag(Wdo, L.ldd, 0, dt * 16);
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt) {     // dKf^T[d][key], dVf^T[d][key]: lane = key `nt*16 + col`, 4 consecutive d
            gK[nt][dt] = mma16(bq, trfrag(Wds, L.ldk, 0, nt * 16), gK[nt][dt]);
            gV[nt][dt] = mma16(bo, trfrag(Wp, L.ldk, 0, nt * 16), gV[nt][dt]);
          }
        }
      }
    }

    if (BWD) {
      // the waves' key-side partials meet in the fp32 tiles: wave 0 (always has tile 0) stores, the others add in turn
      constexpr int RS = D16 + 4;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (wave == w && any_tile) {
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              f32x4* rk = reinterpret_cast<f32x4*>(Rk + (nt * 16 + col) * RS + dt * 16 + 4 * q4);   // key row, 4 consecutive d
              f32x4* rv = reinterpret_cast<f32x4*>(Rv + (nt * 16 + col) * RS + dt * 16 + 4 * q4);
              if (w == 0) { *rk = gK[nt][dt]; *rv = gV[nt][dt]; }
              else {
                f32x4 tk = *rk, tv = *rv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { tk[r] += gK[nt][dt][r]; tv[r] += gV[nt][dt][r]; }
                *rk = tk; *rv = tv;
              }
            }
        }
        __syncthreads();                     // after the last turn every wave is out of its tile loop: Kf / Vf are dead
      }
      for (int i = tid; i < NK16 * D16; i += 256) {
        const int row = i / D16, cc = i - row * D16;
        const float vk = Rk[row * RS + cc], vv = Rv[row * RS + cc];
        sm[L.kf + row * L.ldd + cc] = (bf16)vk;                       // dKf / dVf: operands of the Linformer products
        sm[L.vf + row * L.ldd + cc] = (bf16)vv;
        if (cc < D) {
          if (row >= NKo && row < NK) { accSk[(row - NKo) * D + cc] += vk; accSv[(row - NKo) * D + cc] += vv; }
          if (MODE == 1 && row < NKo) {
            const int64_t kr = attn_krow(a, g, row);
            dktg[kr * a.lddk + h * D + cc] = (bf16)vk;
            dvtg[kr * a.lddv + h * D + cc] = (bf16)vv;
          }
        }
      }
      if (MODE == 0) {
        __syncthreads();
        const int nlt = L16 >> 4, njt = a.KC >> 4;
        // dk_tok[l][d] = sum_j E_k[l][j] dKf[j][d]: (L16/16) x DT tiles over the waves
        for (int t = wave; t < nlt * DT; t += 4) {
          const int lt = t / DT, dt = t - lt * DT;
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
          for (int j0 = 0; j0 < a.KC; j0 += 16) {
            ak = mma16(rowfrag(sm + L.ek, L.lde, lt * 16, j0), trfrag(sm + L.kf, L.ldd, j0, dt * 16), ak);
            av = mma16(rowfrag(sm + L.ev, L.lde, lt * 16, j0), trfrag(sm + L.vf, L.ldd, j0, dt * 16), av);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int l = lt * 16 + 4 * q4 + r, cc = dt * 16 + col;
            if (l < a.L && cc < D) {
              const int64_t kr = attn_krow(a, g, l);
              dktg[kr * a.lddk + h * D + cc] = (bf16)ak[r];
              dvtg[kr * a.lddv + h * D + cc] = (bf16)av[r];
            }
          }
        }
        // dE_k[l][j] += sum_d kt[l][d] dKf[j][d]: (L16/16) x (KC/16) tiles over the waves; a tile always lands on the same
        // wave and lane, so the workspace slice is stored by the first problem and read-modify-written afterwards
        for (int t = wave; t < nlt * njt; t += 4) {
          const int lt = t / njt, jt = t - lt * njt;
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            ak = mma16(rowfrag(sm + L.kt, L.ldd, lt * 16, dt * 16), rowfrag(sm + L.kf, L.ldd, jt * 16, dt * 16), ak);
            av = mma16(rowfrag(sm + L.vt, L.ldd, lt * 16, dt * 16), rowfrag(sm + L.vf, L.ldd, jt * 16, dt * 16), av);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int l = lt * 16 + 4 * q4 + r, j = jt * 16 + col;
            if (l < a.L && j < a.KC) {
              float* ek_ = wsE + l * a.KC + j;
              float* ev_ = wsE + nE + l * a.KC + j;
              if (pid == (int)blockIdx.x) { *ek_ = ak[r]; *ev_ = av[r]; }
              else { *ek_ += ak[r]; *ev_ += av[r]; }
            }
          }
        }
      }
    }
  }
  if (BWD) {
    __syncthreads();
    for (int i = tid; i < 2 * nS; i += 256) wsE[2 * nE + i] = accSk[i];
  } else {
    if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
  }
}

template <int MODE, int NKT, int DT>
int a4_launch(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st) {
  const A4Lds L = a4_lds(a, bwd, NKT * 16, DT * 16);
  if (L.bytes > 160 * 1024) return -100;
  if (bwd) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn4_kernel<MODE, NKT, DT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn4_kernel<MODE, NKT, DT, true>), dim3(grid), dim3(256), L.bytes, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn4_kernel<MODE, NKT, DT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn4_kernel<MODE, NKT, DT, false>), dim3(grid), dim3(256), L.bytes, st, a);
  }
  return QAVIT_OK;
}

}  // namespace

// Called by attn_bf16_try after its alignment checks, for Nq >= 32.  1 = launched, 0 = not covered, < 0 = error.
int attn4_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st) {
  static int use4 = -1;
  if (use4 < 0) { const char* e = getenv("QAVIT_ATTN4"); use4 = e ? atoi(e) : 1; }
  if (!use4 || a.Nq < 32) return 0;
  const int NKo = (a.mode == 0) ? a.KC : a.L;
  const int nkt = (NKo + a.S + 15) / 16, dt = (a.D + 15) / 16;
  // channel-group problems (D = 4) are so small that the backward's barriers cost more than the split tiles win: 64 tokens
  // 231 us here vs 156 us on the one-wave kernel, and the 224-px step is 21.1 ms with them there vs 22.4 ms here
  static int cga_bwd = -1;
  if (cga_bwd < 0) { const char* e = getenv("QAVIT_ATTN4_CGA_BWD"); cga_bwd = e ? atoi(e) : 0; }
  if (bwd && a.mode == 1 && dt == 1 && !cga_bwd) return 0;
  int rc = -100;
  if (a.mode == 0 && nkt <= 3 && dt == 3) rc = a4_launch<0, 3, 3>(a, bwd, grid, st);
  else if (a.mode == 0 && nkt <= 5 && dt == 3) rc = a4_launch<0, 5, 3>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt == 1 && dt == 3) rc = a4_launch<1, 1, 3>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt <= 5 && dt == 1) rc = a4_launch<1, 5, 1>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt <= 14 && dt == 1) rc = a4_launch<1, 14, 1>(a, bwd, grid, st);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}

}  // namespace qv
