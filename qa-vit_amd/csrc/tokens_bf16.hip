// bf16 fast paths of the per-image token kernels (bank statistics, TokenLearner mixing, TokenUpMix) on LDS bf16
// tiles + v_mfma_f32_16x16x16_bf16 (frag16.cuh).  The generic fp32 / any-shape versions stay in bank.hip and
// tokens.hip; each `*_try` returns 1 when it took the launch, 0 when the shape is not covered.
#include "common.cuh"
#include "frag16.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "tokens_shared.h"

namespace qv {

// Waves of a workgroup run different trip counts here, so no workgroup barrier may sit inside the image loops: each wave
// owns its LDS tiles; LDS operations of one wave execute in issue order, this only stops the compiler reordering them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// bank statistics: U[s][c] += sum_b softmax_n(tn Wg^T + bg)^T tn,  tn = LN_write(LN_branch(tokens))
// One wave per image (4 images in flight per workgroup).  Rows are normalised in registers (4 lanes per row),
// written once to a bf16 LDS tile that then feeds both products; U stays in accumulator registers across all the
// images a wave visits.
// ------------------------------------------------------------------------------------------------
template <int NT, int CT>   // N = 16*NT tokens, C = 16*CT channels
__global__ __launch_bounds__(256) void bank_stats2_kernel(const bf16* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr,
                                                          const float* Wg, const float* bg, float* ws, int B, float eps) {
  constexpr int N = 16 * NT, C = 16 * CT, S = 16;
  constexpr int LDC = C + 4, LDS_ = S + 4;
  constexpr int CPL = C / 4;                          // channels per lane when 4 lanes share a row
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* WgT = reinterpret_cast<bf16*>(smraw);                         // [S][LDC]
  float* prm = reinterpret_cast<float*>(smraw + S * LDC * 2);         // [4][C]: gbr, bbr, gwr, bwr
  bf16* per_wave = reinterpret_cast<bf16*>(smraw + S * LDC * 2 + 4 * C * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16* tn = per_wave + wave * (N * LDC + N * LDS_);                   // [N][LDC]
  bf16* wt = tn + N * LDC;                                             // [N][LDS_]
  const int col = lane & 15, q4 = lane >> 4;

  for (int i = tid; i < S * C; i += 256) { const int s = i / C, c = i - s * C; WgT[s * LDC + c] = (bf16)Wg[i]; }
  for (int i = tid; i < C; i += 256) { prm[i] = gbr[i]; prm[C + i] = bbr[i]; prm[2 * C + i] = gwr[i]; prm[3 * C + i] = bwr[i]; }
  __syncthreads();
  const float bgv = bg[col];
  f32x4 U[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) U[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float invC = 1.f / (float)C;

  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    // ---- two chained LayerNorms, 4 lanes per row, 16 rows per pass ----
    const int rr = lane >> 2, part = lane & 3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16* src = tokens + ((size_t)b * N + nt * 16 + rr) * C + part * CPL;
      float v[CPL];
#pragma unroll
      for (int j = 0; j < CPL; j += 4) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(src + j);
        v[j] = (float)t[0]; v[j + 1] = (float)t[1]; v[j + 2] = (float)t[2]; v[j + 3] = (float)t[3];
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) s += v[j];
        const float mean = group_sum<4>(s) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) { const float d = v[j] - mean; s2 += d * d; }
        const float rstd = rsqrtf(group_sum<4>(s2) * invC + eps);
        const float* ga = prm + (pass * 2) * C + part * CPL;
        const float* be = prm + (pass * 2 + 1) * C + part * CPL;
#pragma unroll
        for (int j = 0; j < CPL; j += 4) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(ga + j);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(be + j);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) v[j + jj] = (v[j + jj] - mean) * rstd * g4[jj] + b4[jj];
        }
      }
      bf16* dst = tn + (nt * 16 + rr) * LDC + part * CPL;
#pragma unroll
      for (int j = 0; j < CPL; j += 4) {
        bf16x4 t;
        t[0] = (bf16)v[j]; t[1] = (bf16)v[j + 1]; t[2] = (bf16)v[j + 2]; t[3] = (bf16)v[j + 3];
        *reinterpret_cast<bf16x4*>(dst + j) = t;
      }
    }
    wave_sync();                                  // per-wave tiles: only this wave's lanes need to see the rows
    // ---- gate logits [N][S] = tn . Wg^T + bg, softmax over the N tokens of each slot ----
    f32x4 lg[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc = {bgv, bgv, bgv, bgv};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc = mma16(rowfrag(tn, LDC, nt * 16, ct * 16), rowfrag(WgT, LDC, 0, ct * 16), acc);
      lg[nt] = acc;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, lg[nt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(lg[nt][r] - mx); lg[nt][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc_to_lds(wt, LDS_, nt * 16, 0, lg[nt], inv);
    wave_sync();
    // ---- U[s][c] += sum_n w[n][s] tn[n][c] ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) U[ct] = mma16(trfrag(wt, LDS_, nt * 16, 0), trfrag(tn, LDC, nt * 16, ct * 16), U[ct]);
    wave_sync();
  }
  // ---- reduce the 4 waves' accumulators and write this workgroup's partial ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(per_wave);            // [4][S*C] aliases the (now idle) per-wave tiles
  float* mine = red + wave * S * C;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[(4 * q4 + r) * C + ct * 16 + col] = U[ct][r];
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * S * C;
  for (int i = tid; i < S * C; i += 256) out[i] = red[i] + red[S * C + i] + red[2 * S * C + i] + red[3 * S * C + i];
}

template <int NT, int CT>
static int bank2_launch(const bf16* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr, const float* Wg,
                        const float* bg, float* ws, int B, int grid, float eps, hipStream_t st) {
  constexpr int N = 16 * NT, C = 16 * CT, S = 16;
  size_t per_wave = (size_t)(N * (C + 4) + N * (S + 4)) * 2;
  size_t tail = 4 * per_wave;
  if (tail < (size_t)4 * S * C * 4) tail = (size_t)4 * S * C * 4;
  const size_t smem = (size_t)S * (C + 4) * 2 + 4 * C * 4 + tail;
  if (smem > 160 * 1024) return -100;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bank_stats2_kernel<NT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((bank_stats2_kernel<NT, CT>), dim3(grid), dim3(256), smem, st, tokens, gbr, bbr, gwr, bwr, Wg, bg, ws, B, eps);
  return QAVIT_OK;
}

int bank_stats_bf16_try(const void* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr, const float* Wg,
                        const float* bg, float* ws, int B, int N, int C, int S, int grid, float eps, hipStream_t st) {
  if (S != 16 || (reinterpret_cast<uintptr_t>(tokens) & 7)) return 0;
  int rc = -100;
  if (C == 192 && N == 16) rc = bank2_launch<1, 12>((const bf16*)tokens, gbr, bbr, gwr, bwr, Wg, bg, ws, B, grid, eps, st);
  else if (C == 192 && N == 64) rc = bank2_launch<4, 12>((const bf16*)tokens, gbr, bbr, gwr, bwr, Wg, bg, ws, B, grid, eps, st);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}

}  // namespace qv
