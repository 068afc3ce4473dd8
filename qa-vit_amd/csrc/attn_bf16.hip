// bf16 attention core (forward + backward) for the four QA-ViT branches -- the fast path behind qavit_attn_fwd /
// qavit_attn_bwd when dtype == QAVIT_BF16 (the fp32 parity path stays in attn.hip).
//
// One wavefront per (group, head) problem.  Every operand lives in the wave's LDS slice as a row-major bf16 tile
// and feeds v_mfma_f32_16x16x16_bf16 (k = 16: D = 48, 48 keys, 16 window tokens, 32 Linformer rows are all
// multiples of 16, so no k padding).  A fragment whose reduction axis runs along the tile's COLUMNS is one
// ds_read_b64 per lane; one whose reduction axis runs along the ROWS is one ds_read_b64_tr_b16 (hardware
// transpose) -- so Q.Kf^T, P.Vf, P^T.dO, dS^T.Q, E^T.k_tok ... all read the same images with no transposed
// copies.  Softmax runs on the accumulator registers (16-lane row reductions); only P / dS go back to LDS (bf16)
// as the next product's operand.  dKf/dVf accumulate in registers across the query tiles of a problem; dE and
// the shared-row (bank) gradients accumulate in fp32 LDS across the problems a wave visits and leave as one
// partial per wave (folded by attn_reduce_kernel in attn.hip).
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include <stdlib.h>
#include "attn_shared.h"
#include "frag16.cuh"

namespace qv {

// 4 accumulator values of a TRANSPOSED product = 4 consecutive elements of one output row: one 8-byte store
__device__ __forceinline__ bf16x4 row4(const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  return v;
}

struct A2Lds {   // offsets in bf16 elements unless noted
  int ldd, ldk, lde;
  int q, d_o, p, ds, kt, vt, ek, ev, kf, vf, dkf, dvf;
  int end16;                // bf16 elements in use (zeroed once)
  int accf;                 // float offset (from the float view of the base) of the fp32 accumulators
  int bytes;
};

// nk16 / d16: the kernel instantiation's padded key count and head dim (NKT*16, DT*16), not the problem's own
__host__ __device__ inline A2Lds a2_lds(const qavit_attn_args& a, bool bwd, int NK16, int D16) {
  A2Lds L;
  const int L16 = (a.L + 15) / 16 * 16;
  L.ldd = D16 + 4; L.ldk = NK16 + 4; L.lde = a.KC + 4;
  int o = 0;
  L.q = o; o += 16 * L.ldd;
  L.p = o; o += 16 * L.ldk;
  L.kf = o; o += NK16 * L.ldd;
  L.vf = o; o += NK16 * L.ldd;
  L.kt = L.vt = L.ek = L.ev = 0;
  if (a.mode == 0) {
    L.kt = o; o += L16 * L.ldd;
    L.vt = o; o += L16 * L.ldd;
    L.ek = o; o += L16 * L.lde;
    L.ev = o; o += L16 * L.lde;
  }
  L.d_o = L.ds = L.dkf = L.dvf = 0;
  if (bwd) {
    L.d_o = o; o += 16 * L.ldd;
    L.ds = o; o += 16 * L.ldk;
    // dKf / dVf (bf16 operands of the Linformer products) are written after the last query tile, when Kf / Vf are dead:
    // same tiles.  Their padded rows / columns come out 0 (P = dS = 0 there), which is what the next problem expects.
    L.dkf = L.kf;
    L.dvf = L.vf;
  }
  L.end16 = o;
  o = (o + 7) / 8 * 8;                         // 16-byte boundary
  L.accf = o / 2;
  // fp32 accumulators in LDS: the shared-row (bank) gradients only.  dE_k / dE_v accumulate in the wave's own slice of
  // the workspace (attn.hip: attn_ws_per_wave) -- 10 KB less LDS per wave, 4 instead of 2 waves per CU for MSDA at N=64.
  const int fl = bwd ? 2 * a.S * a.D : 0;
  L.bytes = o * 2 + fl * 4;
  return L;
}

template <int MODE, int NKT, int DT, bool BWD>
__global__ __launch_bounds__(64) void attn2_kernel(qavit_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* smf = reinterpret_cast<float*>(smraw);
  const A2Lds L = a2_lds(a, BWD, NKT * 16, DT * 16);
  const int lane = threadIdx.x, col = lane & 15, q4 = lane >> 4;
  const int D = a.D;
  const int NKo = (MODE == 0) ? a.KC : a.L;
  const int NK = NKo + a.S, NK16 = NKT * 16;
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
  float* accSk = smf + L.accf;
  float* accSv = accSk + nS;
  float* wsE = BWD ? a.ws + (size_t)blockIdx.x * (2 * nE + 2 * nS) : nullptr;    // [dE_k | dE_v | dsh_k | dsh_v]
  if (BWD) for (int i = lane; i < 2 * nS; i += 64) accSk[i] = 0.f;

  // zero the padded tiles once: rows / columns beyond the valid extents must read as 0 in every product
  {
    for (int i = lane; i < L.end16; i += 64) sm[i] = (bf16)0.f;
  }

  for (int pid = blockIdx.x; pid < a.G * a.H; pid += gridDim.x) {
    const int g = pid / a.H, h = pid - g * a.H;
    __syncthreads();
    // ---------------- key side (4-element chunks: 16-byte fp32 / 8-byte bf16 loads, 8-byte LDS stores) ----------------
    const int DC = D >> 2;                               // chunks per row (D % 4 == 0, checked on the host)
    for (int i = lane; i < a.S * DC; i += 64) {
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
      for (int i = lane; i < a.L * DC; i += 64) {
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
      for (int i = lane; i < a.L * EC; i += 64) {
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
      // Kf[j][d] = sum_l E_k[l][j] kt[l][d]   (both operands reduce along their rows -> transposed reads)
      for (int jt = 0; jt * 16 < a.KC; ++jt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
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

    f32x4 gK[NKT][DT], gV[NKT][DT];
    if (BWD) {
#pragma unroll
      for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int j = 0; j < DT; ++j) { gK[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; gV[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    // query side: the next tile's q / dO rows are fetched into registers while this tile computes (unconditional loads
    // of clamped rows, zeroed at commit: a load under a lane-dependent branch would be waited for at the join)
    bf16x4 pq[DT], pg[DT];
    auto fetch = [&](int q0) {
      const int rows_n = (a.Nq - q0 < 16) ? a.Nq - q0 : 16;
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        const int i = lane + 64 * c;
        const int r = i / DC, ch = i - r * DC;
        const int rc = r < rows_n ? r : rows_n - 1;
        const int64_t qr = attn_qrow(a, g, q0 + (rc < 0 ? 0 : rc));
        const int chc = (i < 16 * DC) ? ch : 0;
        pq[c] = *reinterpret_cast<const bf16x4*>(qg + qr * a.ldq + h * D + 4 * chc);
        if (BWD) pg[c] = *reinterpret_cast<const bf16x4*>(dog + qr * a.lddo + h * D + 4 * chc);
      }
    };
    fetch(0);
    for (int q0 = 0; q0 < a.Nq; q0 += 16) {
      const int rows = (a.Nq - q0 < 16) ? a.Nq - q0 : 16;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        const int i = lane + 64 * c;
        if (i < 16 * DC) {
          const int r = i / DC, ch = i - r * DC;
          bf16x4 v = pq[c], gvv;
#pragma unroll
          for (int j = 0; j < 4; ++j) gvv[j] = (bf16)0.f;
          if (BWD) gvv = pg[c];
          if (r >= rows) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = (bf16)0.f; gvv[j] = (bf16)0.f; }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) bad |= ((float)v[j] != (float)v[j]);
          *reinterpret_cast<bf16x4*>(sm + L.q + r * L.ldd + 4 * ch) = v;
          if (BWD) *reinterpret_cast<bf16x4*>(sm + L.d_o + r * L.ldd + 4 * ch) = gvv;
        }
      }
      __syncthreads();
      if (q0 + 16 < a.Nq) fetch(q0 + 16);
      // ---------------- scores + softmax on registers ----------------
      f32x4 s[NKT];
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          acc = mma16(rowfrag(sm + L.q, L.ldd, 0, dt * 16), rowfrag(sm + L.kf, L.ldd, nt * 16, dt * 16), acc);
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
          acc_to_lds(sm + L.p, L.ldk, 0, nt * 16, pd);
        } else {
          acc_to_lds(sm + L.p, L.ldk, 0, nt * 16, s[nt]);
        }
      }
      __syncthreads();
      const int64_t my_q = attn_qrow(a, g, q0 + (col < rows ? col : 0));      // this lane's query row in the global matrices
      if (!BWD) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt)       // O^T[d][query]: this lane = query row `col`, 4 consecutive d
            acc = mma16(trfrag(sm + L.vf, L.ldd, nt * 16, dt * 16), rowfrag(sm + L.p, L.ldk, 0, nt * 16), acc);
          bad |= (acc[0] != acc[0]) | (acc[1] != acc[1]) | (acc[2] != acc[2]) | (acc[3] != acc[3]);
          if (col < rows && dt * 16 + 4 * q4 < D)
            *reinterpret_cast<bf16x4*>(og + my_q * a.ldo + h * D + dt * 16 + 4 * q4) = row4(acc);
        }
      } else {
        // dP = dO . Vf^T ; dS = P * (dP - rowdot) * scale
        f32x4 dp[NKT];
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            acc = mma16(rowfrag(sm + L.d_o, L.ldd, 0, dt * 16), rowfrag(sm + L.vf, L.ldd, nt * 16, dt * 16), acc);
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
        for (int nt = 0; nt < NKT; ++nt) acc_to_lds(sm + L.ds, L.ldk, 0, nt * 16, dp[nt]);
        __syncthreads();
        // dQ = dS . Kf ; dKf += dS^T . Q ; dVf += P^T . dO
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt)       // dQ^T[d][query]
            acc = mma16(trfrag(sm + L.kf, L.ldd, nt * 16, dt * 16), rowfrag(sm + L.ds, L.ldk, 0, nt * 16), acc);
          if (col < rows && dt * 16 + 4 * q4 < D)
            *reinterpret_cast<bf16x4*>(dqg + my_q * a.lddq + h * D + dt * 16 + 4 * q4) = row4(acc);
          const s16x4 bq = trfrag(sm + L.q, L.ldd, 0, dt * 16);
          const s16x4 bo = trfrag(sm + L.d_o, L.ldd, 0, dt * 16);
#pragma unroll
          for (int nt = 0; nt < NKT; ++nt) {     // dKf^T[d][key], dVf^T[d][key]: lane = key `nt*16 + col`, 4 consecutive d
            gK[nt][dt] = mma16(bq, trfrag(sm + L.ds, L.ldk, 0, nt * 16), gK[nt][dt]);
            gV[nt][dt] = mma16(bo, trfrag(sm + L.p, L.ldk, 0, nt * 16), gV[nt][dt]);
          }
        }
      }
    }

    if (BWD) {
      __syncthreads();
      // key-side gradients: to LDS (bf16, operands of the Linformer products) and into the shared-row accumulators
#pragma unroll
      for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int row = nt * 16 + col, c0 = dt * 16 + 4 * q4;            // key row, first of this lane's 4 columns
          *reinterpret_cast<bf16x4*>(sm + L.dkf + row * L.ldd + c0) = row4(gK[nt][dt]);
          *reinterpret_cast<bf16x4*>(sm + L.dvf + row * L.ldd + c0) = row4(gV[nt][dt]);
          if (c0 < D) {
            if (row >= NKo && row < NK) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                accSk[(row - NKo) * D + c0 + r] += gK[nt][dt][r];
                accSv[(row - NKo) * D + c0 + r] += gV[nt][dt][r];
              }
            }
            if (MODE == 1 && row < NKo) {
              const int64_t kr = attn_krow(a, g, row);
              *reinterpret_cast<bf16x4*>(dktg + kr * a.lddk + h * D + c0) = row4(gK[nt][dt]);
              *reinterpret_cast<bf16x4*>(dvtg + kr * a.lddv + h * D + c0) = row4(gV[nt][dt]);
            }
          }
        }
      if (MODE == 0) {
        __syncthreads();
        for (int lt = 0; lt * 16 < L16; ++lt) {
          // dk_tok[l][d] = sum_j E_k[l][j] dKf[j][d]
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
            for (int j0 = 0; j0 < a.KC; j0 += 16) {
              ak = mma16(rowfrag(sm + L.ek, L.lde, lt * 16, j0), trfrag(sm + L.dkf, L.ldd, j0, dt * 16), ak);
              av = mma16(rowfrag(sm + L.ev, L.lde, lt * 16, j0), trfrag(sm + L.dvf, L.ldd, j0, dt * 16), av);
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
          // dE_k[l][j] += sum_d kt[l][d] dKf[j][d]
          for (int jt = 0; jt * 16 < a.KC; ++jt) {
            f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              ak = mma16(rowfrag(sm + L.kt, L.ldd, lt * 16, dt * 16), rowfrag(sm + L.dkf, L.ldd, jt * 16, dt * 16), ak);
              av = mma16(rowfrag(sm + L.vt, L.ldd, lt * 16, dt * 16), rowfrag(sm + L.dvf, L.ldd, jt * 16, dt * 16), av);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int l = lt * 16 + 4 * q4 + r, j = jt * 16 + col;
              if (l < a.L && j < a.KC) {
                // the same lane owns the same element in every problem of this wave: plain store, then read-modify-write
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
  }
  if (BWD) {
    __syncthreads();
    for (int i = lane; i < 2 * nS; i += 64) wsE[2 * nE + i] = accSk[i];
  } else {
    if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
  }
}

template <int MODE, int NKT, int DT>
static int a2_launch(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st) {
  const A2Lds L = a2_lds(a, bwd, NKT * 16, DT * 16);
  if (L.bytes > 160 * 1024) return set_error(QAVIT_EINVAL, "attn(bf16): problem does not fit one wave's LDS slice");
  if (bwd) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn2_kernel<MODE, NKT, DT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn2_kernel<MODE, NKT, DT, true>), dim3(grid), dim3(64), L.bytes, st, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn2_kernel<MODE, NKT, DT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn2_kernel<MODE, NKT, DT, false>), dim3(grid), dim3(64), L.bytes, st, a);
  }
  return QAVIT_OK;
}

// returns 1 if this fast path took the launch, 0 if the shape is not covered (caller falls back), <0 on error
int attn_bf16_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st) {
  const int NKo = (a.mode == 0) ? a.KC : a.L;
  const int nkt = (NKo + a.S + 15) / 16, dt = (a.D + 15) / 16;
  if (a.mode == 0 && (a.KC % 16 != 0)) return 0;
  // 8-byte / 16-byte vector staging: head dim and every leading dimension a multiple of 4 elements, aligned bases
  auto al = [](const void* p, size_t n) { return (reinterpret_cast<uintptr_t>(p) % n) == 0; };
  if (a.D % 4 || a.ldq % 4 || (a.L > 0 && (a.ldk % 4 || a.ldv % 4)) || !al(a.q, 8) || !al(a.sh_k, 16) || !al(a.sh_v, 16)) return 0;
  if (a.L > 0 && (!al(a.k_tok, 8) || !al(a.v_tok, 8))) return 0;
  if (a.mode == 0 && (!al(a.E_k, 16) || !al(a.E_v, 16))) return 0;
  if (bwd && (a.lddo % 4 || !al(a.d_o, 8))) return 0;
  // 8-byte row-fragment stores of the transposed products
  if (!bwd && (a.ldo % 4 || !al(a.o, 8))) return 0;
  if (bwd && (a.lddq % 4 || !al(a.dq, 8) || (a.L > 0 && (a.lddk % 4 || a.lddv % 4 || !al(a.dk_tok, 8) || !al(a.dv_tok, 8))))) return 0;
  {
    static int use3 = -1;
    if (use3 < 0) { const char* e = getenv("QAVIT_ATTN3"); use3 = e ? atoi(e) : 1; }
    const int t3 = use3 ? attn3_try(a, bwd, grid, st) : 0;
    if (t3) return t3;
    const int t4 = attn4_try(a, bwd, grid, st);
    if (t4) return t4;
  }
  int rc = -100;
  if (a.mode == 0 && nkt <= 3 && dt == 3) rc = a2_launch<0, 3, 3>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt == 1 && dt == 3) rc = a2_launch<1, 1, 3>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt <= 2 && dt == 1) rc = a2_launch<1, 2, 1>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt <= 5 && dt == 1) rc = a2_launch<1, 5, 1>(a, bwd, grid, st);
  // 224-px shapes (N = 196): 64 Linformer rows + 16 bank rows = 80 keys for SWA / MSDA; 196 tokens + 16 for CGA (D = 4)
  else if (a.mode == 0 && nkt <= 5 && dt == 3) rc = a2_launch<0, 5, 3>(a, bwd, grid, st);
  else if (a.mode == 1 && nkt <= 14 && dt == 1) rc = a2_launch<1, 14, 1>(a, bwd, grid, st);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}

}  // namespace qv
