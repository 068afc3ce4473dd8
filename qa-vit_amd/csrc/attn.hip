// Attention core of the four QA-ViT branches (see include/qavit.h, qavit_attn_args).
// One wavefront per (group g, head h) problem; all operands of the problem live in that wave's LDS slice
// as fp32 and feed 16x16 MFMA tiles (mma_lds.cuh).  Key-side matrices (Linformer-compressed keys + bank
// rows) are built once per problem and reused by every 16-row query tile.
#include "common.cuh"
#include "mma_lds.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include <stdlib.h>

namespace qv {

struct AttnDims {
  int NK, NKo, NKp;   // total keys, own/compressed keys, padded row stride of the score tile
};
__host__ __device__ inline AttnDims attn_dims(const qavit_attn_args& a) {
  AttnDims d;
  d.NKo = (a.mode == 0) ? a.KC : a.L;
  d.NK = d.NKo + a.S;
  d.NKp = d.NK + 4;
  return d;
}

// LDS layout (floats), shared by fwd and bwd
struct AttnLds {
  int q, s, kf, vf, kt, vt, ek, ev;            // fwd
  int d_o, dp, dkf, dvf, acc_ek, acc_ev, acc_shk, acc_shv;   // bwd only
  int total;
  int spill;                                   // bwd, mode 0: E read from global, dE accumulated in the wave's workspace
};
__host__ __device__ inline AttnLds attn_lds_impl(const qavit_attn_args& a, bool bwd, bool spill) {
  const AttnDims d = attn_dims(a);
  AttnLds L; int o = 0;
  L.spill = spill ? 1 : 0;
  L.q = o; o += 16 * a.D;
  L.s = o; o += 16 * d.NKp;
  L.kf = o; o += d.NK * a.D;
  L.vf = o; o += d.NK * a.D;
  L.kt = L.vt = L.ek = L.ev = 0;
  if (a.mode == 0) {
    L.kt = o; o += a.L * a.D;
    L.vt = o; o += a.L * a.D;
    if (!spill) { L.ek = o; o += a.L * a.KC; L.ev = o; o += a.L * a.KC; }
  }
  L.d_o = L.dp = L.dkf = L.dvf = L.acc_ek = L.acc_ev = L.acc_shk = L.acc_shv = 0;
  if (bwd) {
    L.d_o = o; o += 16 * a.D;
    L.dp = o; o += 16 * d.NKp;
    L.dkf = o; o += d.NK * a.D;
    L.dvf = o; o += d.NK * a.D;
    if (a.mode == 0 && !spill) { L.acc_ek = o; o += a.L * a.KC; L.acc_ev = o; o += a.L * a.KC; }
    L.acc_shk = o; o += a.S * a.D;
    L.acc_shv = o; o += a.S * a.D;
  }
  L.total = (o + 3) / 4 * 4;
  return L;
}
// Backward problems whose Linformer matrices do not fit next to everything else (MSDA at 224 px: L=128, KC=64)
// keep E in global memory (L2-resident, shared by every wave) and accumulate dE in place in the wave's workspace.
__host__ __device__ inline AttnLds attn_lds(const qavit_attn_args& a, bool bwd) {
  AttnLds L = attn_lds_impl(a, bwd, false);
  if (bwd && a.mode == 0 && (size_t)L.total * 4 > 160 * 1024) L = attn_lds_impl(a, bwd, true);
  return L;
}

// floats of workspace one wave writes: [dE_k (L*KC) | dE_v | dsh_k (S*D) | dsh_v]
__host__ __device__ inline int64_t attn_ws_per_wave(const qavit_attn_args& a) {
  return (a.mode == 0 ? 2 * (int64_t)a.L * a.KC : 0) + 2 * (int64_t)a.S * a.D;
}

static inline int attn_grid(const qavit_attn_args& a, bool bwd) {
  const int64_t problems = (int64_t)a.G * a.H;
  // 1792 = 7 waves per CU x 256 CUs: the small-problem backward (attn3, 22 KB of LDS per wave) has exactly 7 resident waves per CU,
  // so every workgroup of the launch is resident at once.  At 2048 the last 256 start when the first finish and the launch takes
  // two "rounds" of 2 problems each instead of one round of 2-3 (measured: 1.54 -> 1.32 ms of attention backward per step).
  static const int bwd_cap = getenv("QAVIT_ATTN_BWD_CAP") ? atoi(getenv("QAVIT_ATTN_BWD_CAP")) : 1792;
  static const int fwd_cap = getenv("QAVIT_ATTN_FWD_CAP") ? atoi(getenv("QAVIT_ATTN_FWD_CAP")) : 2048;
  int64_t cap = bwd ? bwd_cap : fwd_cap;          // bwd: bounded so the partial-sum workspace stays small
  int64_t g = problems < cap ? problems : cap;
  g = g / a.H * a.H;                          // a multiple of H: the head of a wave is fixed
  if (g < a.H) g = a.H;
  return (int)g;
}

__device__ __forceinline__ int64_t qrow(const qavit_attn_args& a, int g, int i) {
  if (a.groups_per_b <= 0) return (int64_t)g * a.Nq + i;
  const int b = g / a.groups_per_b, gi = g - b * a.groups_per_b;
  return (int64_t)b * a.q_rows_per_b + (a.q_tbl ? a.q_tbl[gi * a.Nq + i] : gi * a.Nq + i);
}
__device__ __forceinline__ int64_t krow(const qavit_attn_args& a, int g, int l) {
  if (a.groups_per_b <= 0) return (int64_t)g * a.L + l;
  const int b = g / a.groups_per_b, gi = g - b * a.groups_per_b;
  return (int64_t)b * a.k_rows_per_b + (a.k_tbl ? a.k_tbl[gi * a.L + l] : gi * a.L + l);
}

// ---- shared staging: builds Kf / Vf (and keeps kt/vt/ek/ev for bwd) for problem (g,h) ----
template <typename T, bool BF, bool SPILL = false>
__device__ __forceinline__ bool stage_keys(const qavit_attn_args& a, const AttnDims& d, const AttnLds& L, float* sm, int g, int h) {
  const int lane = threadIdx.x;
  const int D = a.D;
  bool bad = false;
  // shared rows
  for (int i = lane; i < a.S * D; i += 64) {
    const int s = i / D, dd = i - s * D;
    const float k = a.sh_k[(size_t)s * a.H * D + h * D + dd];
    const float v = a.sh_v[(size_t)s * a.H * D + h * D + dd];
    bad |= (k != k) | (v != v);
    sm[L.kf + (d.NKo + s) * D + dd] = k;
    sm[L.vf + (d.NKo + s) * D + dd] = v;
  }
  const T* kt = reinterpret_cast<const T*>(a.k_tok);
  const T* vt = reinterpret_cast<const T*>(a.v_tok);
  if (a.mode == 0) {
    for (int i = lane; i < a.L * D; i += 64) {
      const int l = i / D, dd = i - l * D;
      const int64_t kr = krow(a, g, l);
      const float k = to_f<T>(kt[kr * a.ldk + h * D + dd]);
      const float v = to_f<T>(vt[kr * a.ldv + h * D + dd]);
      bad |= (k != k) | (v != v);
      sm[L.kt + i] = k;
      sm[L.vt + i] = v;
    }
    if (!SPILL) for (int i = lane; i < a.L * a.KC; i += 64) { sm[L.ek + i] = a.E_k[i]; sm[L.ev + i] = a.E_v[i]; }
    __syncthreads();
    // Kf[j][d] = sum_l E_k[l][j] * kt[l][d]
    for (int jt = 0; jt * 16 < a.KC; ++jt)
      for (int dt = 0; dt * 16 < D; ++dt) {
        f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
        if (SPILL) {
          ak = mma_tile<BF>(a.E_k + jt * 16, 1, a.KC, a.KC - jt * 16, sm + L.kt + dt * 16, D, 1, D - dt * 16, a.L, ak);
          av = mma_tile<BF>(a.E_v + jt * 16, 1, a.KC, a.KC - jt * 16, sm + L.vt + dt * 16, D, 1, D - dt * 16, a.L, av);
        } else {
          ak = mma_tile<BF>(sm + L.ek + jt * 16, 1, a.KC, a.KC - jt * 16, sm + L.kt + dt * 16, D, 1, D - dt * 16, a.L, ak);
          av = mma_tile<BF>(sm + L.ev + jt * 16, 1, a.KC, a.KC - jt * 16, sm + L.vt + dt * 16, D, 1, D - dt * 16, a.L, av);
        }
        tile_to_f32<false>(sm + L.kf + jt * 16 * D + dt * 16, D, 1, a.KC - jt * 16, D - dt * 16, ak);
        tile_to_f32<false>(sm + L.vf + jt * 16 * D + dt * 16, D, 1, a.KC - jt * 16, D - dt * 16, av);
      }
  } else {
    for (int i = lane; i < a.L * D; i += 64) {
      const int l = i / D, dd = i - l * D;
      const int64_t kr = krow(a, g, l);
      const float k = to_f<T>(kt[kr * a.ldk + h * D + dd]);
      const float v = to_f<T>(vt[kr * a.ldv + h * D + dd]);
      bad |= (k != k) | (v != v);
      sm[L.kf + i] = k;
      sm[L.vf + i] = v;
    }
  }
  __syncthreads();
  return bad;
}

// scores of a 16-row query tile -> probabilities in sm[L.s] (row stride NKp)
// DROP_FWD: the forward's tile leaves as P * dropout factor (the backward keeps P and applies the mask itself)
template <bool BF>
__device__ __forceinline__ void scores_softmax(const qavit_attn_args& a, const AttnDims& d, const AttnLds& L, float* sm, int rows, float scale,
                                               const AttnDrop& drop, uint32_t pkey, int q0, bool drop_fwd) {
  const int D = a.D;
  for (int nt = 0; nt * 16 < d.NK; ++nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mma_tile<BF>(sm + L.q, D, 1, rows, sm + L.kf + nt * 16 * D, 1, D, d.NK - nt * 16, D, acc);
    tile_to_f32<false>(sm + L.s + nt * 16, d.NKp, 1, rows, d.NK - nt * 16, acc, scale);
  }
  __syncthreads();
  {  // 4 lanes per row
    const int lane = threadIdx.x, row = lane >> 2, part = lane & 3;
    float* srow = sm + L.s + row * d.NKp;
    float mx = -INFINITY;
    if (row < rows) for (int j = part; j < d.NK; j += 4) mx = fmaxf(mx, srow[j]);
    mx = group_max<4>(mx);
    float sum = 0.f;
    if (row < rows) for (int j = part; j < d.NK; j += 4) { const float e = __expf(srow[j] - mx); srow[j] = e; sum += e; }
    sum = group_sum<4>(sum);
    const float inv = 1.f / sum;
    if (row < rows) {
      if (drop_fwd && drop.on) for (int j = part; j < d.NK; j += 4) srow[j] *= inv * attn_drop_factor(drop, pkey, q0 + row, j);
      else for (int j = part; j < d.NK; j += 4) srow[j] *= inv;
    }
  }
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ bool load_rows16(const qavit_attn_args& a, int g, int q0, const T* src, int64_t ld, int h, int rows, int D, float* dst) {
  bool bad = false;
  for (int i = threadIdx.x; i < 16 * D; i += 64) {
    const int r = i / D, dd = i - r * D;
    const float v = (r < rows) ? to_f<T>(src[qrow(a, g, q0 + r) * ld + h * D + dd]) : 0.f;
    bad |= (v != v);
    dst[i] = v;
  }
  return bad;
}

// accumulator tile -> rows qrow(g, q0 + row) / krow(g, l0 + row) of a global matrix
template <typename T>
__device__ __forceinline__ void store_qrows(const qavit_attn_args& a, int g, int q0, T* dst, int64_t ld, int coff, int rows, int cols, const f32x4& acc) {
  const int col = tile_col();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = tile_row(reg);
    if (row < rows && col < cols) dst[qrow(a, g, q0 + row) * ld + coff + col] = from_f<T>(acc[reg]);
  }
}
template <typename T>
__device__ __forceinline__ void store_krows(const qavit_attn_args& a, int g, int l0, T* dst, int64_t ld, int coff, int rows, int cols, const f32x4& acc) {
  const int col = tile_col();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = tile_row(reg);
    if (row < rows && col < cols) dst[krow(a, g, l0 + row) * ld + coff + col] = from_f<T>(acc[reg]);
  }
}

template <typename T, bool BF>
__global__ __launch_bounds__(64) void attn_fwd_kernel(qavit_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const AttnDims d = attn_dims(a);
  const AttnLds L = attn_lds(a, false);
  const int D = a.D;
  const float scale = rsqrtf((float)D);
  bool bad = false;
  const AttnDrop drop = attn_drop_init(a);
  const T* q = reinterpret_cast<const T*>(a.q);
  T* o = reinterpret_cast<T*>(a.o);
  for (int pid = blockIdx.x; pid < a.G * a.H; pid += gridDim.x) {
    const int g = pid / a.H, h = pid - g * a.H;
    const uint32_t pkey = attn_drop_pkey(drop, pid);
    __syncthreads();
    bad |= stage_keys<T, BF>(a, d, L, sm, g, h);
    for (int q0 = 0; q0 < a.Nq; q0 += 16) {
      const int rows = (a.Nq - q0 < 16) ? a.Nq - q0 : 16;
      __syncthreads();
      bad |= load_rows16<T>(a, g, q0, q, a.ldq, h, rows, D, sm + L.q);
      __syncthreads();
      scores_softmax<BF>(a, d, L, sm, rows, scale, drop, pkey, q0, true);
      for (int dt = 0; dt * 16 < D; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_tile<BF>(sm + L.s, d.NKp, 1, rows, sm + L.vf + dt * 16, D, 1, D - dt * 16, d.NK, acc);
        bad |= (acc[0] != acc[0]) | (acc[1] != acc[1]) | (acc[2] != acc[2]) | (acc[3] != acc[3]);
        store_qrows<T>(a, g, q0, o, a.ldo, h * D + dt * 16, rows, D - dt * 16, acc);
      }
    }
  }
  if (a.nan_flag && __any(bad) && threadIdx.x == 0) atomicOr(a.nan_flag, 1);
}

template <typename T, bool BF, bool SPILL>
__global__ __launch_bounds__(64) void attn_bwd_kernel(qavit_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const AttnDims d = attn_dims(a);
  const AttnLds L = attn_lds(a, true);
  const int D = a.D;
  const float scale = rsqrtf((float)D);
  const int lane = threadIdx.x;
  const T* q = reinterpret_cast<const T*>(a.q);
  const T* dO = reinterpret_cast<const T*>(a.d_o);
  T* dq = reinterpret_cast<T*>(a.dq);
  T* dkt = reinterpret_cast<T*>(a.dk_tok);
  T* dvt = reinterpret_cast<T*>(a.dv_tok);
  const AttnDrop drop = attn_drop_init(a);
  const int n_e = (a.mode == 0) ? 2 * a.L * a.KC : 0;
  const int n_acc = (SPILL ? 0 : n_e) + 2 * a.S * D;             // accumulators held in LDS (contiguous)
  const int acc0 = (a.mode == 0 && !SPILL) ? L.acc_ek : L.acc_shk;
  for (int i = lane; i < n_acc; i += 64) sm[acc0 + i] = 0.f;
  float* ws = a.ws + (size_t)blockIdx.x * attn_ws_per_wave(a);
  if (SPILL) {                                                   // dE_k | dE_v live in the workspace itself
    for (int i = lane; i < n_e; i += 64) ws[i] = 0.f;
    __threadfence();
  }

  for (int pid = blockIdx.x; pid < a.G * a.H; pid += gridDim.x) {
    const int g = pid / a.H, h = pid - g * a.H;
    __syncthreads();
    stage_keys<T, BF, SPILL>(a, d, L, sm, g, h);
    for (int i = lane; i < 2 * d.NK * D; i += 64) sm[L.dkf + i] = 0.f;     // dkf and dvf are adjacent
    for (int q0 = 0; q0 < a.Nq; q0 += 16) {
      const int rows = (a.Nq - q0 < 16) ? a.Nq - q0 : 16;
      __syncthreads();
      load_rows16<T>(a, g, q0, q, a.ldq, h, rows, D, sm + L.q);
      load_rows16<T>(a, g, q0, dO, a.lddo, h, rows, D, sm + L.d_o);
      __syncthreads();
      scores_softmax<BF>(a, d, L, sm, rows, scale, drop, 0u, q0, false);     // P in sm[L.s]
      // dP~ = dO . Vf^T  (gradient of the DROPPED probabilities P~ = P * m)
      for (int nt = 0; nt * 16 < d.NK; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_tile<BF>(sm + L.d_o, D, 1, rows, sm + L.vf + nt * 16 * D, 1, D, d.NK - nt * 16, D, acc);
        tile_to_f32<false>(sm + L.dp + nt * 16, d.NKp, 1, rows, d.NK - nt * 16, acc);
      }
      __syncthreads();
      {  // dP = m * dP~ ; dS = P * (dP - sum_j P*dP) * scale (in place in dp) ; the P tile becomes P~ for the dVf product
        const int row = lane >> 2, part = lane & 3;
        float* prow = sm + L.s + row * d.NKp;
        float* drow = sm + L.dp + row * d.NKp;
        const uint32_t pkey = attn_drop_pkey(drop, pid);
        float dot = 0.f;
        if (row < rows) {
          if (drop.on) for (int j = part; j < d.NK; j += 4) drow[j] *= attn_drop_factor(drop, pkey, q0 + row, j);
          for (int j = part; j < d.NK; j += 4) dot += prow[j] * drow[j];
        }
        dot = group_sum<4>(dot);
        if (row < rows) for (int j = part; j < d.NK; j += 4) {
          const float pj = prow[j];
          drow[j] = pj * (drow[j] - dot) * scale;
          if (drop.on) prow[j] = pj * attn_drop_factor(drop, pkey, q0 + row, j);
        }
      }
      __syncthreads();
      // dVf += P~^T . dO
      for (int nt = 0; nt * 16 < d.NK; ++nt)
        for (int dt = 0; dt * 16 < D; ++dt) {
          f32x4 av = {0.f, 0.f, 0.f, 0.f};
          av = mma_tile<BF>(sm + L.s + nt * 16, 1, d.NKp, d.NK - nt * 16, sm + L.d_o + dt * 16, D, 1, D - dt * 16, rows, av);
          tile_to_f32<true>(sm + L.dvf + nt * 16 * D + dt * 16, D, 1, d.NK - nt * 16, D - dt * 16, av);
        }
      // dQ = dS . Kf ;  dKf += dS^T . Q
      for (int dt = 0; dt * 16 < D; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_tile<BF>(sm + L.dp, d.NKp, 1, rows, sm + L.kf + dt * 16, D, 1, D - dt * 16, d.NK, acc);
        store_qrows<T>(a, g, q0, dq, a.lddq, h * D + dt * 16, rows, D - dt * 16, acc);
        for (int nt = 0; nt * 16 < d.NK; ++nt) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f};
          ak = mma_tile<BF>(sm + L.dp + nt * 16, 1, d.NKp, d.NK - nt * 16, sm + L.q + dt * 16, D, 1, D - dt * 16, rows, ak);
          tile_to_f32<true>(sm + L.dkf + nt * 16 * D + dt * 16, D, 1, d.NK - nt * 16, D - dt * 16, ak);
        }
      }
    }
    __syncthreads();
    // shared-row gradients (this wave's head slice)
    for (int i = lane; i < a.S * D; i += 64) {
      sm[L.acc_shk + i] += sm[L.dkf + d.NKo * D + i];
      sm[L.acc_shv + i] += sm[L.dvf + d.NKo * D + i];
    }
    if (a.mode == 0) {
      // dk_tok[l][d] = sum_j E_k[l][j] dKf[j][d] ;  dE_k[l][j] += sum_d kt[l][d] dKf[j][d]
      for (int lt = 0; lt * 16 < a.L; ++lt) {
        for (int dt = 0; dt * 16 < D; ++dt) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
          if (SPILL) {
            ak = mma_tile<BF>(a.E_k + lt * 16 * a.KC, a.KC, 1, a.L - lt * 16, sm + L.dkf + dt * 16, D, 1, D - dt * 16, a.KC, ak);
            av = mma_tile<BF>(a.E_v + lt * 16 * a.KC, a.KC, 1, a.L - lt * 16, sm + L.dvf + dt * 16, D, 1, D - dt * 16, a.KC, av);
          } else {
            ak = mma_tile<BF>(sm + L.ek + lt * 16 * a.KC, a.KC, 1, a.L - lt * 16, sm + L.dkf + dt * 16, D, 1, D - dt * 16, a.KC, ak);
            av = mma_tile<BF>(sm + L.ev + lt * 16 * a.KC, a.KC, 1, a.L - lt * 16, sm + L.dvf + dt * 16, D, 1, D - dt * 16, a.KC, av);
          }
          store_krows<T>(a, g, lt * 16, dkt, a.lddk, h * D + dt * 16, a.L - lt * 16, D - dt * 16, ak);
          store_krows<T>(a, g, lt * 16, dvt, a.lddv, h * D + dt * 16, a.L - lt * 16, D - dt * 16, av);
        }
        for (int jt = 0; jt * 16 < a.KC; ++jt) {
          f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
          ak = mma_tile<BF>(sm + L.kt + lt * 16 * D, D, 1, a.L - lt * 16, sm + L.dkf + jt * 16 * D, 1, D, a.KC - jt * 16, D, ak);
          av = mma_tile<BF>(sm + L.vt + lt * 16 * D, D, 1, a.L - lt * 16, sm + L.dvf + jt * 16 * D, 1, D, a.KC - jt * 16, D, av);
          if (SPILL) {     // the same lane owns the same element in every problem: plain read-modify-write
            tile_to_f32<true>(ws + lt * 16 * a.KC + jt * 16, a.KC, 1, a.L - lt * 16, a.KC - jt * 16, ak);
            tile_to_f32<true>(ws + a.L * a.KC + lt * 16 * a.KC + jt * 16, a.KC, 1, a.L - lt * 16, a.KC - jt * 16, av);
          } else {
            tile_to_f32<true>(sm + L.acc_ek + lt * 16 * a.KC + jt * 16, a.KC, 1, a.L - lt * 16, a.KC - jt * 16, ak);
            tile_to_f32<true>(sm + L.acc_ev + lt * 16 * a.KC + jt * 16, a.KC, 1, a.L - lt * 16, a.KC - jt * 16, av);
          }
        }
      }
    } else {
      for (int i = lane; i < a.L * D; i += 64) {
        const int l = i / D, dd = i - l * D;
        const int64_t kr = krow(a, g, l);
        dkt[kr * a.lddk + h * D + dd] = from_f<T>(sm[L.dkf + i]);
        dvt[kr * a.lddv + h * D + dd] = from_f<T>(sm[L.dvf + i]);
      }
    }
  }
  __syncthreads();
  float* wo = SPILL ? ws + n_e : ws;
  for (int i = lane; i < n_acc; i += 64) wo[i] = sm[acc0 + i];
}

// fold the per-wave partials into the gradient buffers: thread = output element, blockIdx.y = slice of 8*H waves
// (coalesced reads across the element axis, one fp32 atomic per thread)
constexpr int RED_WAVES = 8;
__global__ __launch_bounds__(256) void attn_reduce_kernel(qavit_attn_args a, int nwaves) {
  const int nE = (a.mode == 0) ? a.L * a.KC : 0;
  const int nS = a.S * a.D;
  const int per = 2 * nE + 2 * nS;
  const int total = 2 * nE + 2 * nS * a.H;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int w0 = blockIdx.y * RED_WAVES * a.H;            // slices are multiples of H waves: head alignment is kept
  int w1 = w0 + RED_WAVES * a.H;
  if (w1 > nwaves) w1 = nwaves;
  float s = 0.f;
  if (i < 2 * nE) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;                   // independent chains: the loads of a slice are all in flight
    int w = w0;
    for (; w + 3 < w1; w += 4) {
      s += a.ws[(size_t)w * per + i]; s1 += a.ws[(size_t)(w + 1) * per + i];
      s2 += a.ws[(size_t)(w + 2) * per + i]; s3 += a.ws[(size_t)(w + 3) * per + i];
    }
    for (; w < w1; ++w) s += a.ws[(size_t)w * per + i];
    s += s1 + s2 + s3;
    float* dst = (i < nE) ? a.dE_k : a.dE_v;
    if (dst) atomic_add_f(dst + (i < nE ? i : i - nE), s);
  } else {
    int r = i - 2 * nE;                 // [which(2)][h][s][d]
    const int which = r / (nS * a.H); r -= which * nS * a.H;
    const int h = r / nS; r -= h * nS;
    const int srow = r / a.D, dd = r - srow * a.D;
    const size_t off = (size_t)2 * nE + which * nS + srow * a.D + dd;
    float s1 = 0.f;
    int w = w0 + h;
    for (; w + a.H < w1; w += 2 * a.H) { s += a.ws[(size_t)w * per + off]; s1 += a.ws[(size_t)(w + a.H) * per + off]; }
    for (; w < w1; w += a.H) s += a.ws[(size_t)w * per + off];
    s += s1;
    float* dst = which == 0 ? a.dsh_k : a.dsh_v;
    if (dst) atomic_add_f(dst + (size_t)srow * a.H * a.D + h * a.D + dd, s);
  }
}

// x := 0 if flag[0] != 0; the LAST workgroup to have read the flag resets it (flag[1] is the arrival ticket), so the flag is
// clear for the next attention call without a second launch
template <typename T>
__global__ __launch_bounds__(256) void nan_guard_kernel(T* x, int64_t n, int* flag) {
  __shared__ int f_s;
  if (threadIdx.x == 0) {
    f_s = *reinterpret_cast<volatile int*>(flag);
    __threadfence();
    if (atomicAdd(flag + 1, 1) == (int)gridDim.x - 1) { flag[0] = 0; flag[1] = 0; }
  }
  __syncthreads();
  if (f_s == 0) return;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = from_f<T>(0.f);
}

static int attn_validate(const qavit_attn_args* a, bool bwd) {
  if (!a) return set_error(QAVIT_EINVAL, "attn: null args");
  if (a->G <= 0 || a->Nq <= 0 || a->H <= 0 || a->D <= 0 || a->S < 0 || a->L < 0) return set_error(QAVIT_EINVAL, "attn: bad dimensions");
  if (a->mode != 0 && a->mode != 1) return set_error(QAVIT_EINVAL, "attn: mode must be 0 or 1");
  if (a->mode == 0 && (a->KC <= 0 || a->L <= 0 || !a->E_k || !a->E_v)) return set_error(QAVIT_EINVAL, "attn: Linformer mode needs E_k/E_v, L and KC");
  if (!a->q || (!bwd && !a->o) || (a->L > 0 && (!a->k_tok || !a->v_tok)) || (a->S > 0 && (!a->sh_k || !a->sh_v)))
    return set_error(QAVIT_EINVAL, "attn: null operand");
  if (attn_dims(*a).NK <= 0) return set_error(QAVIT_EINVAL, "attn: no keys");
  if (bwd) {
    if (!a->d_o || !a->dq || (a->L > 0 && (!a->dk_tok || !a->dv_tok)) || !a->ws) return set_error(QAVIT_EINVAL, "attn_bwd: null operand");
  }
  return QAVIT_OK;
}

}  // namespace qv

using namespace qv;

extern "C" int64_t qavit_attn_ws_floats(const qavit_attn_args* a) {
  if (!a) return 0;
  return (int64_t)attn_grid(*a, true) * attn_ws_per_wave(*a);
}

static void launch_reduce(const qavit_attn_args& a, int grid, hipStream_t st) {
  const int nE = (a.mode == 0) ? a.L * a.KC : 0;
  const int total = 2 * nE + 2 * a.S * a.D * a.H;
  if (total > 0) {
    const int slices = (grid + RED_WAVES * a.H - 1) / (RED_WAVES * a.H);
    hipLaunchKernelGGL(attn_reduce_kernel, dim3((total + 255) / 256, slices), dim3(256), 0, st, a, grid);
  }
}

template <typename T, bool BF>
static int attn_launch(const qavit_attn_args& a, bool bwd, hipStream_t st) {
  const int grid = attn_grid(a, bwd);
  if (bwd && a.ws_floats < (int64_t)grid * attn_ws_per_wave(a)) return set_error(QAVIT_EINVAL, "attn_bwd: workspace too small");
  if (BF) {
    static int use_fast = -1;
    if (use_fast < 0) { const char* e = getenv("QAVIT_ATTN_GENERIC"); use_fast = (e && atoi(e)) ? 0 : 1; }
    const int took = use_fast ? attn_bf16_try(a, bwd, grid, st) : 0;
    if (took < 0) return took;
    if (took == 1) {
      if (bwd) launch_reduce(a, grid, st);
      return check_launch(bwd ? "attn_bwd(bf16)" : "attn_fwd(bf16)");
    }
  }
  const AttnLds L = attn_lds(a, bwd);
  if ((size_t)L.total * 4 > 160 * 1024) return set_error(QAVIT_EINVAL, "attn: problem does not fit one wave's LDS slice");
  const size_t smem = (size_t)L.total * sizeof(float);
  if (!bwd) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<T, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL((attn_fwd_kernel<T, BF>), dim3(grid), dim3(64), smem, st, a);
    return check_launch("attn_fwd");
  }
  if (L.spill) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<T, BF, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn_bwd_kernel<T, BF, true>), dim3(grid), dim3(64), smem, st, a);
  } else {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<T, BF, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL((attn_bwd_kernel<T, BF, false>), dim3(grid), dim3(64), smem, st, a);
  }
  launch_reduce(a, grid, st);
  return check_launch("attn_bwd");
}

extern "C" int qavit_attn_fwd(const qavit_attn_args* a, void* stream) {
  int rc = attn_validate(a, false);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == QAVIT_F32) return attn_launch<float, false>(*a, false, st);
  if (a->dtype == QAVIT_BF16) return attn_launch<bf16, true>(*a, false, st);
  return set_error(QAVIT_EINVAL, "attn_fwd: unknown dtype");
}

extern "C" int qavit_attn_bwd(const qavit_attn_args* a, void* stream) {
  int rc = attn_validate(a, true);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == QAVIT_F32) return attn_launch<float, false>(*a, true, st);
  if (a->dtype == QAVIT_BF16) return attn_launch<bf16, true>(*a, true, st);
  return set_error(QAVIT_EINVAL, "attn_bwd: unknown dtype");
}

extern "C" int qavit_nan_guard(int dtype, void* x, int64_t n, int* flag, void* stream) {
  if (!x || !flag || n <= 0) return set_error(QAVIT_EINVAL, "nan_guard: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int nb = (int)((n + 2047) / 2048);
  if (nb > 64) nb = 64;          // the common case reads the flag and leaves; every workgroup also takes a ticket on one address
  if (dtype == QAVIT_F32) hipLaunchKernelGGL((nan_guard_kernel<float>), dim3(nb), dim3(256), 0, st, (float*)x, n, flag);
  else if (dtype == QAVIT_BF16) hipLaunchKernelGGL((nan_guard_kernel<bf16>), dim3(nb), dim3(256), 0, st, (bf16*)x, n, flag);
  else return set_error(QAVIT_EINVAL, "nan_guard: unknown dtype");
  return check_launch("nan_guard");
}
