// bf16 fast paths of the per-image token kernels (bank statistics, TokenLearner mixing, TokenUpMix) on LDS bf16
// tiles + v_mfma_f32_16x16x16_bf16 (frag16.cuh).  The generic fp32 / any-shape versions stay in bank.hip and
// tokens.hip; each `*_try` returns 1 when it took the launch, 0 when the shape is not covered.
#include "common.cuh"
#include <stdlib.h>
#include "frag16.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "tokens_shared.h"

#ifdef QAVIT_TOKEN_STAMPS     // diagnostic build only (tools/token_stamps.py): s_memtime at phase boundaries of the up-mix backward, 32 words per workgroup
__device__ unsigned long long qv_token_stamps[1024 * 32];
#define TSTAMP(k) do { if (threadIdx.x == 0 && (k) < 32) qv_token_stamps[(size_t)(blockIdx.x & 1023) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int qavit_token_stamps(void* host_dst, int nwg) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(qv_token_stamps), (size_t)nwg * 32 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#else
#define TSTAMP(k) do { } while (0)
#endif

namespace qv {

// Waves of a workgroup run different trip counts here, so no workgroup barrier may sit inside the image loops: each wave
// owns its LDS tiles and orders its own LDS traffic with wave_sync() (common.cuh).

// ------------------------------------------------------------------------------------------------
// bank statistics: U[s][c] += sum_b softmax_n(tn Wg^T + bg)^T tn,  tn = LN_write(LN_branch(tokens))
// One wave per image (4 images in flight per workgroup).  Rows are normalised in registers (4 lanes per row),
// written once to a bf16 LDS tile that then feeds both products; U stays in accumulator registers across all the
// images a wave visits.
// ------------------------------------------------------------------------------------------------
// `fx` (flag != NULL): the NaN -> zeros rule of the branch that produced `tokens`, deferred to this launch (qavit_nan_fix): with the flag
// raised a wave rewrites the rows of each image it visits as dropout(bias) -- in memory, for the later readers, and in its registers --
// and zeroes the image's saved attention output, before the statistics are taken; flag / ticket / trip are handled as the rule's own launch does.
template <int NT, int CT>   // N = 16*NT tokens, C = 16*CT channels
__global__ __launch_bounds__(256) void bank_stats2_kernel(bf16* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr,
                                                          const float* Wg, const float* bg, float* ws, int B, float eps, qavit_nan_fix fx) {
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

  // (the flag's broadcast word sits behind the dynamic region: a static __shared__ object on top of the 160 KB opt-in would not fit)
  constexpr size_t PW_B = (size_t)(N * LDC + N * LDS_) * 2;
  constexpr size_t TAIL_B = 4 * PW_B > (size_t)4 * S * C * 4 ? 4 * PW_B : (size_t)4 * S * C * 4;
  volatile int& f_s = *reinterpret_cast<volatile int*>(smraw + (size_t)S * LDC * 2 + 4 * C * 4 + TAIL_B);
  if (tid == 0) {
    int f = 0;
    if (fx.flag) {
      f = *reinterpret_cast<volatile int*>(fx.flag);
      // the read has RETURNED before this workgroup's arrival is counted (no agent-scope fence: on eight XCDs that is an L2 write-back per
      // workgroup, and all that must be ordered is this one load before this one atomic)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (atomicAdd(fx.flag + 1, 1) == (int)gridDim.x - 1) { fx.flag[0] = 0; fx.flag[1] = 0; }     // every workgroup has read the flag by then
    }
    f_s = f;
  }
  for (int i = tid; i < S * C; i += 256) { const int s = i / C, c = i - s * C; WgT[s * LDC + c] = (bf16)Wg[i]; }
  for (int i = tid; i < C; i += 256) { prm[i] = gbr[i]; prm[C + i] = bbr[i]; prm[2 * C + i] = gwr[i]; prm[3 * C + i] = bwr[i]; }
  __syncthreads();
  const bool fixit = f_s != 0;                         // uniform over the grid
  if (fx.flag && fx.trip && blockIdx.x == 0 && tid == 0) *fx.trip = fixit ? 1 : 0;
  const bool fdrop = fixit && fx.drop_p > 0.f && fx.rng != nullptr;
  const uint32_t fkey = fdrop ? rng_key(fx.rng, fx.drop_site) : 0u;
  const float finv = fdrop ? 1.f / (1.f - fx.drop_p) : 1.f;
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
      bf16* src = tokens + ((size_t)b * N + nt * 16 + rr) * C + part * CPL;
      float v[CPL];
      if (fixit) {                                    // rare: the branch output is dropout(proj(0)) = dropout(bias), for this reader and the later ones
        const uint32_t row = (uint32_t)(b * N + nt * 16 + rr);
#pragma unroll
        for (int j = 0; j < CPL; j += 4) {
          bf16x4 t;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int c = part * CPL + j + jj;
            float val = fx.bias[c];
            if (fdrop) val *= drop_factor(fkey, row * (uint32_t)C + (uint32_t)c, fx.drop_p, finv);
            t[jj] = (bf16)val;
            v[j + jj] = (float)t[jj];
          }
          *reinterpret_cast<bf16x4*>(src + j) = t;
        }
      } else {
#pragma unroll
        for (int j = 0; j < CPL; j += 4) {
          const bf16x4 t = *reinterpret_cast<const bf16x4*>(src + j);
          v[j] = (float)t[0]; v[j + 1] = (float)t[1]; v[j + 2] = (float)t[2]; v[j + 3] = (float)t[3];
        }
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
    if (fixit && fx.o_save) {
      bf16* os = reinterpret_cast<bf16*>(fx.o_save) + (size_t)b * N * fx.ldos;
      for (int i = lane; i < N * fx.Co; i += 64) { const int r = i / fx.Co, c = i - r * fx.Co; os[(size_t)r * fx.ldos + c] = (bf16)0.f; }
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
    mx = rows4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(lg[nt][r] - mx); lg[nt][r] = e; sum += e; }
    sum = rows4_sum(sum);
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
                        const float* bg, float* ws, int B, int grid, float eps, hipStream_t st, const qavit_nan_fix& fx) {
  constexpr int N = 16 * NT, C = 16 * CT, S = 16;
  size_t per_wave = (size_t)(N * (C + 4) + N * (S + 4)) * 2;
  size_t tail = 4 * per_wave;
  if (tail < (size_t)4 * S * C * 4) tail = (size_t)4 * S * C * 4;
  const size_t smem = (size_t)S * (C + 4) * 2 + 4 * C * 4 + tail + 16;      // + the NaN-rule flag's broadcast word
  if (smem > 160 * 1024) return -100;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bank_stats2_kernel<NT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((bank_stats2_kernel<NT, CT>), dim3(grid), dim3(256), smem, st, const_cast<bf16*>(tokens), gbr, bbr, gwr, bwr, Wg, bg, ws, B, eps, fx);
  return QAVIT_OK;
}

int bank_stats_bf16_try(const void* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr, const float* Wg,
                        const float* bg, float* ws, int B, int N, int C, int S, int grid, float eps, hipStream_t st, const qavit_nan_fix* fix) {
  if (S != 16 || (reinterpret_cast<uintptr_t>(tokens) & 7)) return 0;
  qavit_nan_fix fx{};
  if (fix) {
    fx = *fix;
    if (fx.o_save && ((reinterpret_cast<uintptr_t>(fx.o_save) & 1) || fx.ldos < fx.Co)) return 0;
  }
  int rc = -100;
  if (C == 192 && N == 16) rc = bank2_launch<1, 12>((const bf16*)tokens, gbr, bbr, gwr, bwr, Wg, bg, ws, B, grid, eps, st, fx);
  else if (C == 192 && N == 64) rc = bank2_launch<4, 12>((const bf16*)tokens, gbr, bbr, gwr, bwr, Wg, bg, ws, B, grid, eps, st, fx);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}


// ------------------------------------------------------------------------------------------------
// TokenUpMix: up[n][c] = sum_m W[n][m] xc[m][c] + bias[n];  y = LN_c(up).   One workgroup per image, one wave per
// 16-row tile; the 192 columns of a row live in 12 accumulator tiles of ONE wave, so the LayerNorm (and, in backward,
// its gradient) is a 16-lane register reduction -- `up` never touches memory.
// ------------------------------------------------------------------------------------------------
// ROWS rows of C bf16 channels from global memory into an LDS tile of row stride LD, by 256 threads, in BATCHES of loads: all the loads of
// a batch are in flight before the first LDS store.  A plain `for (i = tid; ...) lds[i] = g[i]` with a trip count the compiler does not
// unroll is a chain of dependent round trips -- 12 (64 x 192 tile) or 48 (256 x 192) of them at the top of every image.
template <int ROWS, int C, int LD, int BMAX = 12>
__device__ __forceinline__ void stage_rows_batched(bf16* dst, const bf16* src) {
  constexpr int CH = C / 4, TOT = ROWS * CH, PER = (TOT + 255) / 256, BATCH = PER < BMAX ? PER : BMAX;
#pragma unroll
  for (int base = 0; base < PER; base += BATCH) {
    bf16x4 r[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = threadIdx.x + 256 * (base + j);
      if (base + j < PER && (TOT % 256 == 0 || i < TOT)) { const int n = i / CH, ch = i - n * CH; r[j] = *reinterpret_cast<const bf16x4*>(src + (size_t)n * C + 4 * ch); }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = threadIdx.x + 256 * (base + j);
      if (base + j < PER && (TOT % 256 == 0 || i < TOT)) { const int n = i / CH, ch = i - n * CH; *reinterpret_cast<bf16x4*>(dst + n * LD + 4 * ch) = r[j]; }
    }
  }
}

// The block tail in front of TokenUpMix, y = x + droppath(gamma * u) (HQAViT_CIFAR100.py:1085, :1118-1121), differentiated inside the up-mix
// backward: dxc is this kernel's output anyway; with `u` given it also writes du = dxc * f * gamma and adds sum(dxc * f * u) to dgamma (f = the
// image's drop-path factor) -- the elementwise launch that did this re-read dxc and u from memory, once per block.
struct UpScaleAdd { const bf16* u; bf16* du; const float* gamma; float* dgamma; float dp_p; int dp_site; const int64_t* rng; };
// The same block tail in the forward: with `u` given the up-mix forward's `xc` operand is x, the kernel forms xc = x + f * gamma * u while it
// stages the image (f = the image's drop-path factor) and writes it to `xc_out` for the backward -- the elementwise launch in front of it is gone.
struct UpFwdAdd { const bf16* u; bf16* xc_out; const float* gamma; float dp_p; int dp_site; const int64_t* rng; };

template <int ROWS, int C, int LD, int BMAX = 6>
__device__ __forceinline__ void stage_rows_scale_add(bf16* dst, const bf16* x, const bf16* u, bf16* out, float f) {
  constexpr int CH = C / 4, TOT = ROWS * CH, PER = (TOT + 255) / 256, BATCH = PER < BMAX ? PER : BMAX;
#pragma unroll
  for (int base = 0; base < PER; base += BATCH) {
    bf16x4 rx[BATCH], ru[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = threadIdx.x + 256 * (base + j);
      if (base + j < PER && (TOT % 256 == 0 || i < TOT)) {
        const int n = i / CH, ch = i - n * CH;
        rx[j] = *reinterpret_cast<const bf16x4*>(x + (size_t)n * C + 4 * ch);
        ru[j] = *reinterpret_cast<const bf16x4*>(u + (size_t)n * C + 4 * ch);
      }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = threadIdx.x + 256 * (base + j);
      if (base + j < PER && (TOT % 256 == 0 || i < TOT)) {
        const int n = i / CH, ch = i - n * CH;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)((float)rx[j][r] + f * (float)ru[j][r]);      // the expression of scale_add_vec_kernel
        *reinterpret_cast<bf16x4*>(dst + n * LD + 4 * ch) = o;
        *reinterpret_cast<bf16x4*>(out + (size_t)n * C + 4 * ch) = o;
      }
    }
  }
}

template <int NT, int MT, int CT>
struct UpLds {
  static constexpr int N = 16 * NT, M = 16 * MT, C = 16 * CT;
  static constexpr int LDM = M + 4, LDC = C + 4;
  static constexpr int wt = 0;                               // bf16 [N][LDM]
  static constexpr int xc = wt + N * LDM;                    // bf16 [M][LDC]
  static constexpr int dup = xc + M * LDC;                   // bf16 [min(N,64)][LDC]   (backward only: one 4-tile chunk)
  static constexpr int fwd_bf16 = dup;
  static constexpr int bwd_bf16 = dup + (N < 64 ? N : 64) * LDC;
};

template <int NT, int MT, int CT, int BMAX = 16>
__device__ __forceinline__ void up_stage_w(bf16* Wt, const float* W) {
  using L = UpLds<NT, MT, CT>;
  if constexpr (BMAX == 0) {
    for (int i = threadIdx.x; i < L::N * L::M; i += 256) { const int n = i / L::M, m = i - n * L::M; Wt[n * L::LDM + m] = (bf16)W[i]; }
    return;
  }
  constexpr int TOT = L::N * L::M, PER = (TOT + 255) / 256, BATCH = BMAX == 0 ? 1 : (PER < BMAX ? PER : BMAX);     // loads in batches: see stage_rows_batched
#pragma unroll
  for (int base = 0; base < PER; base += BATCH) {
    float r[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) { const int i = threadIdx.x + 256 * (base + j); if (base + j < PER && i < TOT) r[j] = W[i]; }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = threadIdx.x + 256 * (base + j);
      if (base + j < PER && i < TOT) { const int n = i / L::M, m = i - n * L::M; Wt[n * L::LDM + m] = (bf16)r[j]; }
    }
  }
}
template <int NT, int MT, int CT, int BMAX = 12>
__device__ __forceinline__ void up_stage_xc(bf16* Xs, const bf16* xcb) {
  using L = UpLds<NT, MT, CT>;
  if constexpr (BMAX > 0) {
    stage_rows_batched<L::M, L::C, L::LDC, BMAX>(Xs, xcb);
  } else {                                                   // one load in flight: for the kernel that has no registers to spare
    constexpr int CH = L::C / 4;
    for (int i = threadIdx.x; i < L::M * CH; i += 256) {
      const int m = i / CH, ch = i - m * CH;
      *reinterpret_cast<bf16x4*>(Xs + m * L::LDC + 4 * ch) = *reinterpret_cast<const bf16x4*>(xcb + (size_t)m * L::C + 4 * ch);
    }
  }
}
// accumulators of one 16-row tile: acc[ct][r] = up[nt*16 + 4q + r][ct*16 + col]
template <int NT, int MT, int CT>
__device__ __forceinline__ void up_tile(const bf16* Wt, const bf16* Xs, const float* bias, int nt, f32x4 (&acc)[CT]) {
  using L = UpLds<NT, MT, CT>;
  const int q4 = (threadIdx.x & 63) >> 4;
  f32x4 bi;
#pragma unroll
  for (int r = 0; r < 4; ++r) bi[r] = bias[nt * 16 + 4 * q4 + r];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    f32x4 a = bi;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a = mma16(rowfrag(Wt, L::LDM, nt * 16, mt * 16), trfrag(Xs, L::LDC, mt * 16, ct * 16), a);
    acc[ct] = a;
  }
}

template <int NT, int MT, int CT>
__global__ __launch_bounds__(256) void upmix2_fwd_kernel(const bf16* xc, const float* W, const float* bias, const float* gamma, const float* beta,
                                                         float eps, bf16* y, float* mean_o, float* rstd_o, int B, UpFwdAdd fa) {
  using L = UpLds<NT, MT, CT>;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q4 = lane >> 4;
  const bool fa_on = fa.u != nullptr;                          // uniform
  const float fa_gm = fa_on && fa.gamma ? fa.gamma[0] : 1.f;
  const bool fa_dp = fa_on && fa.dp_p > 0.f && fa.rng != nullptr;
  const uint32_t fa_key = fa_dp ? rng_key(fa.rng, fa.dp_site) : 0u;
  const float fa_inv = fa_dp ? 1.f / (1.f - fa.dp_p) : 1.f;
  up_stage_w<NT, MT, CT>(sm + L::wt, W);
  float ga[CT], be[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { ga[ct] = gamma[ct * 16 + col]; be[ct] = beta[ct * 16 + col]; }
  const float invC = 1.f / (float)L::C;
  bf16* yt = sm + L::dup + wave * 16 * L::LDC;                 // this wave's output tile [16][LDC]
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    if (fa_on) {
      float f = fa_gm;
      if (fa_dp) f *= drop_factor(fa_key, (uint32_t)b, fa.dp_p, fa_inv);
      const size_t o = (size_t)b * L::M * L::C;
      stage_rows_scale_add<L::M, L::C, L::LDC>(sm + L::xc, xc + o, fa.u + o, fa.xc_out + o, f);
    } else {
      up_stage_xc<NT, MT, CT>(sm + L::xc, xc + (size_t)b * L::M * L::C);
    }
    __syncthreads();
    for (int nt = wave; nt < NT; nt += 4) {
      f32x4 acc[CT];
      up_tile<NT, MT, CT>(sm + L::wt, sm + L::xc, bias, nt, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) s += acc[ct][r];
        const float mean = grp_sum<16>(s) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { const float d = acc[ct][r] - mean; s2 += d * d; }
        const float rstd = rsqrtf(grp_sum<16>(s2) * invC + eps);
        const size_t row = (size_t)b * L::N + nt * 16 + 4 * q4 + r;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) yt[(4 * q4 + r) * L::LDC + ct * 16 + col] = (bf16)((acc[ct][r] - mean) * rstd * ga[ct] + be[ct]);
        if (col == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
      }
      // the tile's 16 rows out of the wave's LDS slice as whole rows (8-byte pieces): two-byte stores of four rows per instruction were
      // the kernel's store path
      wave_sync();
      {
        constexpr int CH = L::C / 4;
        bf16* ys = y + ((size_t)b * L::N + nt * 16) * L::C;
        for (int i = lane; i < 16 * CH; i += 64) {
          const int n = i / CH, ch = i - n * CH;
          *reinterpret_cast<bf16x4*>(ys + (size_t)n * L::C + 4 * ch) = *reinterpret_cast<const bf16x4*>(yt + n * L::LDC + 4 * ch);
        }
      }
      wave_sync();
    }
  }
}

template <int NT, int MT, int CT>
__global__ __launch_bounds__(256) void upmix2_bwd_kernel(const bf16* dy, const bf16* xc, const float* W, const float* bias, const float* gamma,
                                                         const float* mean, const float* rstd, bf16* dxc, float* dW, float* dbias,
                                                         float* dgamma, float* dbeta, int B, float* parts, UpScaleAdd sa) {
  using L = UpLds<NT, MT, CT>;
  constexpr int NTW = (NT + 3) / 4;                           // row tiles per wave
  const bool sa_on = NT <= 4 && sa.u != nullptr;              // uniform
  const float sa_gm = sa_on ? (sa.gamma ? sa.gamma[0] : 1.f) : 0.f;
  const bool sa_dp = sa_on && sa.dp_p > 0.f && sa.rng != nullptr;
  const uint32_t sa_key = sa_dp ? rng_key(sa.rng, sa.dp_site) : 0u;
  const float sa_inv = sa_dp ? 1.f / (1.f - sa.dp_p) : 1.f;
  float sa_sum = 0.f;
  constexpr int DXT = (MT * CT + 3) / 4;                      // dxc tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* fl = reinterpret_cast<float*>(smraw + (size_t)((L::bwd_bf16 + 7) / 8 * 8) * 2);   // [N] dbias + [4][2][C] gamma/beta partials
  float* dba = fl;
  float* gred = fl + L::N;
  // 256-token variant: the dW accumulators of a wave's four row tiles (64 registers) live in LDS, one 16-byte slot per (tile, lane) in
  // accumulator layout -- owned by that lane alone, so no synchronisation: read, 12 MFMAs, write back.  With them in registers the kernel
  // sat at 512 registers per lane and its LayerNorm section ran through AGPR shuffles (stamps: 2/3 of a tile's time).
  constexpr bool DW_LDS = NT > 4;
  f32x4* dwl = reinterpret_cast<f32x4*>(fl + L::N + 8 * L::C);          // [NTW][MT][256 threads]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q4 = lane >> 4;
  TSTAMP(0);
  up_stage_w<NT, MT, CT, (NT <= 4 ? 16 : 0)>(sm + L::wt, W);
  for (int i = threadIdx.x; i < L::N; i += 256) dba[i] = 0.f;
  int img = 0;                                               // (stamps)
  float ga[CT], pg[CT], pb[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { ga[ct] = gamma[ct * 16 + col]; pg[ct] = 0.f; pb[ct] = 0.f; }
  f32x4 dwacc[DW_LDS ? 1 : NTW][DW_LDS ? 1 : MT];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      if (DW_LDS) dwl[(i * MT + j) * 256 + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
      else dwacc[DW_LDS ? 0 : i][DW_LDS ? 0 : j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const float invC = 1.f / (float)L::C;
  // Software pipeline over (image, row-tile chunk) steps: the NEXT step's dy tile (and, at an image boundary, the next image's xc
  // tile) is requested into registers before this step's products start, and committed to LDS at the top of the next step -- with one
  // wave per SIMD nothing else hides the two global round trips per image (they were ~4 of the ~10 us an image took).
  constexpr int CH = L::C / 4;
  constexpr int DYL = 16 * CH / 64;                          // 8-byte pieces of a 16-row dy tile per lane
  constexpr int XL = (L::M * CH + 255) / 256;                // 8-byte pieces of the xc tile per thread
  constexpr bool PIPE = true;                                // dy tiles (+ row statistics) one step ahead: both variants
  constexpr bool PIPE_XC = NT <= 4;                          // the next image's xc tile too: not in the 256-token variant (24 more registers per lane: spills)
  bf16x4 dyr[PIPE ? DYL : 1], xr[PIPE_XC ? XL : 1];
  float mun[4], rsn[4];                                      // the next tile's row statistics travel with its dy rows (in-stamp timeline: fetched
                                                             // inside the row loop they were four dependent global round trips, 2/3 of an image's time)
  auto req_dy = [&](int bb, int tw) {
    const int nt = wave + 4 * tw;
    if (nt < NT && bb < B) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = (size_t)bb * L::N + nt * 16 + 4 * q4 + r;
        mun[r] = mean[row]; rsn[r] = rstd[row];
      }
      const bf16* dys = dy + ((size_t)bb * L::N + nt * 16) * L::C;
#pragma unroll
      for (int j = 0; j < DYL; ++j) {
        const int i = lane + 64 * j, n = i / CH, ch = i - n * CH;
        dyr[j] = *reinterpret_cast<const bf16x4*>(dys + (size_t)n * L::C + 4 * ch);
      }
    }
  };
  auto req_xc = [&](int bb) {
    if (bb < B) {
      const bf16* xs = xc + (size_t)bb * L::M * L::C;
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < L::M * CH) { const int m = i / CH, ch = i - m * CH; xr[j] = *reinterpret_cast<const bf16x4*>(xs + (size_t)m * L::C + 4 * ch); }
      }
    }
  };
  if (PIPE_XC) req_xc(blockIdx.x);
  if (PIPE) req_dy(blockIdx.x, 0);
  TSTAMP(1);
  for (int b = blockIdx.x; b < B; b += gridDim.x, ++img) {
    __syncthreads();
    TSTAMP(2 + 6 * img);
    if (PIPE_XC) {
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < L::M * CH) { const int m = i / CH, ch = i - m * CH; *reinterpret_cast<bf16x4*>(sm + L::xc + m * L::LDC + 4 * ch) = xr[j]; }
      }
      req_xc(b + gridDim.x);
    } else {
      up_stage_xc<NT, MT, CT, 0>(sm + L::xc, xc + (size_t)b * L::M * L::C);     // (512 registers per lane: the plain loop)
    }
    f32x4 dxa[DXT];
#pragma unroll
    for (int i = 0; i < DXT; ++i) dxa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x4 usa[NT <= 4 ? DXT : 1];                             // the scale-add's u segments of this image's dxc tiles: requested now, used after the products
    if (sa_on) {
#pragma unroll
      for (int i = 0; i < DXT; ++i) {
        const int tile = wave + 4 * i;
        if (tile < MT * CT) {
          const int mt = tile / CT, ct = tile - mt * CT;
          usa[NT <= 4 ? i : 0] = *reinterpret_cast<const bf16x4*>(sa.u + ((size_t)b * L::M + mt * 16 + col) * L::C + ct * 16 + 4 * q4);
        }
      }
    }
    // the row tiles go through LDS four at a time (one per wave); dxc accumulates over the chunks in registers
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      __syncthreads();                                           // xc staged / previous chunk consumed
      const int nt = wave + 4 * tw;
      float muc[4], rsc[4];
      if (nt < NT) {
        // this wave's 16 dy rows: whole rows in 8-byte pieces into its slice of the dup tile (read element-wise below, then overwritten
        // in place by du) -- 48 two-byte global loads per lane were the kernel's load path
        constexpr bool STAGE_DY = true;                          // dy rows always reach the row loop through LDS (8-byte row pieces), never as 2-byte global loads
        if (PIPE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { muc[r] = mun[r]; rsc[r] = rsn[r]; }
          bf16* dtw = sm + L::dup + wave * 16 * L::LDC;
#pragma unroll
          for (int j = 0; j < DYL; ++j) {
            const int i = lane + 64 * j, n = i / CH, ch = i - n * CH;
            *reinterpret_cast<bf16x4*>(dtw + n * L::LDC + 4 * ch) = dyr[j];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {                          // requested before the tile's MFMAs, used after them
            const size_t row = (size_t)b * L::N + nt * 16 + 4 * q4 + r;
            muc[r] = mean[row]; rsc[r] = rstd[row];
          }
          if (STAGE_DY) {
            const bf16* dys = dy + ((size_t)b * L::N + nt * 16) * L::C;
            bf16* dtw = sm + L::dup + wave * 16 * L::LDC;
            for (int i = lane; i < 16 * CH; i += 64) {
              const int n = i / CH, ch = i - n * CH;
              *reinterpret_cast<bf16x4*>(dtw + n * L::LDC + 4 * ch) = *reinterpret_cast<const bf16x4*>(dys + (size_t)n * L::C + 4 * ch);
            }
          }
        }
      }
      if (PIPE) { if (tw + 1 < NTW) req_dy(b, tw + 1); else req_dy(b + gridDim.x, 0); }
      TSTAMP(3 + 6 * img);
      if (nt < NT) {
        constexpr bool STAGE_DY = true;
        f32x4 acc[CT];
        up_tile<NT, MT, CT>(sm + L::wt, sm + L::xc, bias, nt, acc);
        if (STAGE_DY) wave_sync();
        TSTAMP(4 + 6 * img);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t row = (size_t)b * L::N + nt * 16 + 4 * q4 + r;
          const float mu = muc[r], rs = rsc[r];
          float g[CT];
          float c1 = 0.f, c2 = 0.f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float d = STAGE_DY ? (float)sm[L::dup + (wave * 16 + 4 * q4 + r) * L::LDC + ct * 16 + col] : (float)dy[row * L::C + ct * 16 + col];
            const float xh = (acc[ct][r] - mu) * rs;
            acc[ct][r] = xh;
            g[ct] = d * ga[ct];
            pg[ct] += d * xh;
            pb[ct] += d;
            c1 += g[ct] * xh; c2 += g[ct];
          }
          c1 = grp_sum<16>(c1) * invC; c2 = grp_sum<16>(c2) * invC;
          float rsum = 0.f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) { const float du = rs * (g[ct] - c2 - acc[ct][r] * c1); acc[ct][r] = du; rsum += du; }
          rsum = grp_sum<16>(rsum);
          if (col == 0) dba[nt * 16 + 4 * q4 + r] += rsum;       // this row belongs to this wave only
        }
        if (STAGE_DY) wave_sync();                               // every lane has read its dy elements: the slice takes du
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc_to_lds(sm + L::dup, L::LDC, wave * 16, ct * 16, acc[ct]);
      }
      TSTAMP(5 + 6 * img);
      __syncthreads();
      TSTAMP(6 + 6 * img);
      // dxc[m][c] += sum_{n in chunk} W[n][m] dup[n][c]
      constexpr int KT = NT < 4 ? NT : 4;
#pragma unroll
      for (int i = 0; i < DXT; ++i) {
        const int tile = wave + 4 * i;
        if (tile < MT * CT) {
          const int mt = tile / CT, ct = tile - mt * CT;
#pragma unroll
          for (int k = 0; k < KT; ++k)
            if (4 * tw + k < NT)
              dxa[i] = (NT <= 4) ? mma16(trfrag(sm + L::dup, L::LDC, k * 16, ct * 16), trfrag(sm + L::wt, L::LDM, (4 * tw + k) * 16, mt * 16), dxa[i])    // swapped: dxc^T tile
                                 : mma16(trfrag(sm + L::wt, L::LDM, (4 * tw + k) * 16, mt * 16), trfrag(sm + L::dup, L::LDC, k * 16, ct * 16), dxa[i]);
        }
      }
      // dW[n][m] += sum_c dup[n][c] xc[m][c]   (own row tile)
      if (nt < NT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 a = DW_LDS ? dwl[(tw * MT + mt) * 256 + threadIdx.x] : dwacc[DW_LDS ? 0 : tw][DW_LDS ? 0 : mt];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            a = mma16(rowfrag(sm + L::dup, L::LDC, wave * 16, ct * 16), rowfrag(sm + L::xc, L::LDC, mt * 16, ct * 16), a);
          if (DW_LDS) dwl[(tw * MT + mt) * 256 + threadIdx.x] = a; else dwacc[DW_LDS ? 0 : tw][DW_LDS ? 0 : mt] = a;
        }
      }
      TSTAMP(7 + 6 * img);
    }
#pragma unroll
    for (int i = 0; i < DXT; ++i) {
      const int tile = wave + 4 * i;
      if (tile < MT * CT) {
        const int mt = tile / CT, ct = tile - mt * CT;
        if (NT <= 4) {                                          // acc[r] = dxc[m = 16 mt + col][c = 16 ct + 4 q4 + r]: one 8-byte row segment
          bf16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = (bf16)dxa[i][r];
          const size_t off = ((size_t)b * L::M + mt * 16 + col) * L::C + ct * 16 + 4 * q4;
          *reinterpret_cast<bf16x4*>(dxc + off) = o4;
          if (sa_on) {                                          // the scale-add in front of the up-mix, on the rounded dxc the separate launch would read
            const float f = sa_dp ? drop_factor(sa_key, (uint32_t)b, sa.dp_p, sa_inv) : 1.f;
            const bf16x4 u4 = usa[NT <= 4 ? i : 0];
            bf16x4 d4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gv = (float)o4[r] * f;
              d4[r] = (bf16)(gv * sa_gm);
              sa_sum += gv * (float)u4[r];
            }
            *reinterpret_cast<bf16x4*>(sa.du + off) = d4;
          }
        } else {                                                // (the 256-token variant holds 512 registers per lane: its schedule is left alone)
#pragma unroll
          for (int r = 0; r < 4; ++r) dxc[((size_t)b * L::M + mt * 16 + 4 * q4 + r) * L::C + ct * 16 + col] = (bf16)dxa[i][r];
        }
      }
    }
  }
  TSTAMP(30);
  __syncthreads();
  constexpr int PROW = L::N * L::M + L::N + 2 * L::C + 4;     // floats per partial row: [dW | dbias | dgamma | dbeta | scale-add dgamma, 0, 0, 0]
  if (sa_on && (sa.dgamma || (NT <= 4 && parts))) {         // (gred is free until the dgamma / dbeta fold below)
    const float t = wave_sum(sa_sum);
    if (lane == 0) gred[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float tt = gred[0] + gred[1] + gred[2] + gred[3];
      // the layer scale's gradient: the last four floats of this workgroup's partial row (reduce descriptor C = 1, a fixed-order fold:
      // the scalar is the same bits on every run), or one atomic per workgroup on one address
      if (NT <= 4 && parts) *reinterpret_cast<f32x4*>(parts + (size_t)blockIdx.x * PROW + (PROW - 4)) = f32x4{tt, 0.f, 0.f, 0.f};
      else atomic_add_f(sa.dgamma, tt);
    }
    __syncthreads();
  } else if (NT <= 4 && parts && threadIdx.x == 0) {
    *reinterpret_cast<f32x4*>(parts + (size_t)blockIdx.x * PROW + (PROW - 4)) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // flush: dW tiles, dbias, dgamma / dbeta -- as this workgroup's row [dW | dbias | dgamma | dbeta] of `parts` (plain stores; one
  // qavit_ln_param_reduce launch folds the rows of every kernel of the pass), or with float atomics.  The atomics are 256+ deep on each
  // of ~1.5k addresses and serialise at L2: they were the larger half of this kernel and the reason a second workgroup per CU lost.
  // (64-token variant only: the 256-token one sits at 512 registers per lane and keeps its atomics)
  float* prow = (NT <= 4 && parts) ? parts + (size_t)blockIdx.x * PROW : nullptr;
  if constexpr (NT <= 4) {
    float* dwdst = prow ? prow : dW;
    const bool plain = prow != nullptr;                      // uniform
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int nt = wave + 4 * tw;
      if (nt < NT) {
        float* rowp = dwdst + (size_t)(nt * 16 + 4 * q4) * L::M + col;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (plain) rowp[r * L::M + mt * 16] = dwacc[DW_LDS ? 0 : tw][DW_LDS ? 0 : mt][r]; else atomic_add_f(rowp + r * L::M + mt * 16, dwacc[DW_LDS ? 0 : tw][DW_LDS ? 0 : mt][r]);
          }
      }
    }
    if (prow) { for (int i = threadIdx.x; i < L::N; i += 256) prow[L::N * L::M + i] = dba[i]; }
    else if (dbias) for (int i = threadIdx.x; i < L::N; i += 256) atomic_add_f(dbias + i, dba[i]);
  } else {
#pragma unroll
    for (int tw = 0; tw < NTW; ++tw) {
      const int nt = wave + 4 * tw;
      if (nt < NT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 a = DW_LDS ? dwl[(tw * MT + mt) * 256 + threadIdx.x] : dwacc[DW_LDS ? 0 : tw][DW_LDS ? 0 : mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) atomic_add_f(dW + (size_t)(nt * 16 + 4 * q4 + r) * L::M + mt * 16 + col, a[r]);
        }
      }
    }
    if (dbias) for (int i = threadIdx.x; i < L::N; i += 256) atomic_add_f(dbias + i, dba[i]);
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float a = pg[ct], c = pb[ct];
    a = rows4_sum(a);
    c = rows4_sum(c);
    if (q4 == 0) { gred[(wave * 2 + 0) * L::C + ct * 16 + col] = a; gred[(wave * 2 + 1) * L::C + ct * 16 + col] = c; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L::C; i += 256) {
    const float sg = gred[(0 * 2 + 0) * L::C + i] + gred[(1 * 2 + 0) * L::C + i] + gred[(2 * 2 + 0) * L::C + i] + gred[(3 * 2 + 0) * L::C + i];
    const float sb = gred[(0 * 2 + 1) * L::C + i] + gred[(1 * 2 + 1) * L::C + i] + gred[(2 * 2 + 1) * L::C + i] + gred[(3 * 2 + 1) * L::C + i];
    if (NT <= 4 && prow) { prow[L::N * L::M + L::N + i] = sg; prow[L::N * L::M + L::N + L::C + i] = sb; }
    else { atomic_add_f(dgamma + i, sg); atomic_add_f(dbeta + i, sb); }
  }
  TSTAMP(31);
}

// workgroups of the backward launch: one per CU.  With the partial-row flush a second workgroup per CU is no longer slower, and not
// faster either (28.3 vs 31.0 us at B = 1024: the per-workgroup prologue -- W staged as bf16, first loads -- and the flush are a third of
// a two-image workgroup's time); QAVIT_UPMIX_BWD_GRID overrides.
template <int NT, int MT, int CT>
static int up2_bwd_grid(int B, bool parts) {
  (void)parts;
  static const int bgrid = getenv("QAVIT_UPMIX_BWD_GRID") ? atoi(getenv("QAVIT_UPMIX_BWD_GRID")) : 0;
  const int g = bgrid > 0 ? bgrid : 256;
  return B < g ? B : g;
}

template <int NT, int MT, int CT>
static int up2_launch(bool bwd, const void* a0, const void* xc, const float* W, const float* bias, const float* gamma, const float* beta,
                      float eps, void* o0, float* mean, float* rstd, float* dW, float* dbias, float* dgamma, float* dbeta, int B, hipStream_t st,
                      float* parts = nullptr, const UpScaleAdd* sa = nullptr, const UpFwdAdd* fa = nullptr) {
  using L = UpLds<NT, MT, CT>;
  if (!bwd) {
    const size_t smem = (size_t)(L::fwd_bf16 + 64 * L::LDC) * 2;        // + the four waves' output tiles
    if (smem > 150 * 1024) return -100;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upmix2_fwd_kernel<NT, MT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // one image per workgroup while W is small; the 256-token variant stages 64 KB of W per workgroup (four batches of loads): a workgroup per CU,
    // each walking its images, pays that once
    static const int fgrid = getenv("QAVIT_UPMIX_FWD_GRID") ? atoi(getenv("QAVIT_UPMIX_FWD_GRID")) : (NT > 4 ? 256 : 2048);
    hipLaunchKernelGGL((upmix2_fwd_kernel<NT, MT, CT>), dim3(B < fgrid ? B : fgrid), dim3(256), smem, st, (const bf16*)xc, W, bias, gamma, beta, eps,
                       (bf16*)o0, mean, rstd, B, fa ? *fa : UpFwdAdd{nullptr, nullptr, nullptr, 0.f, 0, nullptr});
    return QAVIT_OK;
  }
  const size_t smem = (size_t)((L::bwd_bf16 + 7) / 8 * 8) * 2 + (size_t)(L::N + 8 * L::C) * 4 +
                      (NT > 4 ? (size_t)((NT + 3) / 4) * MT * 256 * 16 : 0);          // + the 256-token variant's dW accumulators
  if (smem > 160 * 1024) return -100;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(upmix2_bwd_kernel<NT, MT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((upmix2_bwd_kernel<NT, MT, CT>), dim3(up2_bwd_grid<NT, MT, CT>(B, parts != nullptr)), dim3(256), smem, st, (const bf16*)a0,
                     (const bf16*)xc, W, bias, gamma, mean, rstd, (bf16*)o0, dW, dbias, dgamma, dbeta, B, parts,
                     sa ? *sa : UpScaleAdd{nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr});
  return QAVIT_OK;
}

// partial rows the bf16 backward writes for this shape (0: the shape takes another kernel, no partial-row path)
int upmix_bf16_parts(int B, int N, int M, int C) {
  if (C != 192) return 0;
  if (N == 64 && M == 16) return up2_bwd_grid<4, 1, 12>(B, true);
  return 0;                                                  // (N == 256: atomics, see the kernel's flush)
}

int upmix_bf16_try(bool bwd, const void* dy, const void* xc, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                   void* out, float* mean, float* rstd, float* dW, float* dbias, float* dgamma, float* dbeta, int B, int N, int M, int C, hipStream_t st,
                   float* parts, const void* sa_u, void* sa_du, const float* sa_gamma, float* sa_dgamma, float sa_dp_p, int sa_dp_site, const int64_t* sa_rng) {
  if ((reinterpret_cast<uintptr_t>(xc) & 7) || C != 192) return 0;
  int rc = -100;
  UpScaleAdd sa{(const bf16*)sa_u, (bf16*)sa_du, sa_gamma, sa_dgamma, sa_dp_p, sa_dp_site, sa_rng};
  UpFwdAdd fa{(const bf16*)sa_u, (bf16*)sa_du, sa_gamma, sa_dp_p, sa_dp_site, sa_rng};     // forward: sa_du = where xc is written
  const bool fwd_sa = !bwd && sa_u;
  if (bwd && sa_u && !(N == 64 && M == 16)) return 0;        // the fused scale-add backward exists in the 64-token variant only
  if (N == 64 && M == 16) rc = up2_launch<4, 1, 12>(bwd, dy, xc, W, bias, gamma, beta, eps, out, mean, rstd, dW, dbias, dgamma, dbeta, B, st, parts,
                                                    bwd && sa_u ? &sa : nullptr, fwd_sa ? &fa : nullptr);
  else if (N == 256 && M == 64) rc = up2_launch<16, 4, 12>(bwd, dy, xc, W, bias, gamma, beta, eps, out, mean, rstd, dW, dbias, dgamma, dbeta, B, st, nullptr,
                                                           nullptr, fwd_sa ? &fa : nullptr);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}


// ------------------------------------------------------------------------------------------------
// TokenLearner mix: p = softmax_n(scores[b,:,m]);  xc[m,:] = sum_n p[n,m] x[n,:].   One workgroup per image; P and
// the x tile are staged once in LDS as bf16 and every product is an MFMA on transposed-read fragments.
// ------------------------------------------------------------------------------------------------
template <int NT, int MT, int CT>
struct MixLds {
  static constexpr int N = 16 * NT, M = 16 * MT, C = 16 * CT;
  static constexpr int LDM = M + 4, LDC = C + 4;
  static constexpr int p = 0;                                // bf16 [N][LDM]
  static constexpr int x = p + N * LDM;                      // bf16 [N][LDC]
  static constexpr int g = x + N * LDC;                      // (forward ends here)
  static constexpr int fwd_bf16 = g;
  // backward: x is staged one 16-row tile per wave, so its region is [4][16][LDC]; dxc follows
  static constexpr int bx = p + N * LDM;                     // bf16 [4][16][LDC]
  static constexpr int bg = bx + 64 * LDC;                   // bf16 [M][LDC]
  static constexpr int bwd_bf16 = bg + M * LDC;
};

__device__ __forceinline__ float colred(float v, float* red, int M, int parts, bool is_max) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  const int m = threadIdx.x % M;
  float r = is_max ? -INFINITY : 0.f;
  for (int q = 0; q < parts; ++q) { const float o = red[q * M + m]; r = is_max ? fmaxf(r, o) : r + o; }
  return r;
}

template <int ROWS, int NT, int MT, int CT>
__device__ __forceinline__ void mix_stage_rows(bf16* dst, const bf16* src) {
  using L = MixLds<NT, MT, CT>;
  stage_rows_batched<ROWS, L::C, L::LDC>(dst, src);
}

template <int NT, int MT, int CT>
__global__ __launch_bounds__(256) void tokmix2_fwd_kernel(const bf16* scores, const bf16* x, bf16* p_out, bf16* xc, int B) {
  using L = MixLds<NT, MT, CT>;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* red = reinterpret_cast<float*>(smraw + (size_t)((L::fwd_bf16 + 7) / 8 * 8) * 2);     // [256]
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, q4 = lane >> 4;
  constexpr int parts = 256 / L::M, RP = L::N / parts;          // rows per thread of its column
  const int m = t % L::M, part = t / L::M;
  mix_stage_rows<L::N, NT, MT, CT>(sm + L::x, x + (size_t)b * L::N * L::C);
  const bf16* sc = scores + (size_t)b * L::N * L::M;
  float v[RP];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < RP; ++i) { v[i] = (float)sc[(part + i * parts) * L::M + m]; mx = fmaxf(mx, v[i]); }
  mx = colred(mx, red, L::M, parts, true);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RP; ++i) { v[i] = __expf(v[i] - mx); s += v[i]; }
  s = colred(s, red, L::M, parts, false);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    const int n = part + i * parts;
    const bf16 pv = (bf16)(v[i] * inv);
    sm[L::p + n * L::LDM + m] = pv;
    p_out[(size_t)b * L::N * L::M + n * L::M + m] = pv;
  }
  __syncthreads();
  for (int tile = wave; tile < MT * CT; tile += 4) {
    const int mt = tile / CT, ct = tile - mt * CT;
    // operands swapped = the transposed product: acc[r] = xc[m = 16 mt + col][c = 16 ct + 4 q4 + r] -- one 8-byte row segment per lane
    // instead of four 2-byte elements of four rows (both operand roles have the same lane layout: idx = lane % 16, k = 4 (lane / 16) + j)
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a = mma16(trfrag(sm + L::x, L::LDC, nt * 16, ct * 16), trfrag(sm + L::p, L::LDM, nt * 16, mt * 16), a);
    bf16x4 o4;
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = (bf16)a[r];
    *reinterpret_cast<bf16x4*>(xc + ((size_t)b * L::M + mt * 16 + col) * L::C + ct * 16 + 4 * q4) = o4;
  }
}

// dx[n,:] = sum_m p[n,m] dxc[m,:];  dP[n,m] = x[n,:].dxc[m,:];  dscores = p * (dP - sum_n p*dP)
template <int NT, int MT, int CT>
__global__ __launch_bounds__(256) void tokmix2_bwd_kernel(const bf16* p_in, const bf16* x, const bf16* dxc, bf16* dx, bf16* dscores, int B) {
  using L = MixLds<NT, MT, CT>;
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  bf16* sm = reinterpret_cast<bf16*>(smraw);
  float* dP = reinterpret_cast<float*>(smraw + (size_t)((L::bwd_bf16 + 7) / 8 * 8) * 2);      // [N][M]
  float* red = dP + L::N * L::M;                                                               // [256]
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, q4 = lane >> 4;
  // this wave's x rows travel one tile ahead in registers: the first tile is requested with the image's dxc / P staging (one round trip
  // instead of two), the next one before the current tile's products (the 256-token variant walks four tiles per wave)
  constexpr int CH = L::C / 4;
  constexpr int XL = 16 * CH / 64;
  bf16x4 xr[XL];
  auto req_x = [&](int nt) {
    if (nt < NT) {
      const bf16* xs = x + ((size_t)b * L::N + nt * 16) * L::C;
#pragma unroll
      for (int j = 0; j < XL; ++j) {
        const int i = lane + 64 * j, n = i / CH, ch = i - n * CH;
        xr[j] = *reinterpret_cast<const bf16x4*>(xs + (size_t)n * L::C + 4 * ch);
      }
    }
  };
  req_x(wave);
  mix_stage_rows<L::M, NT, MT, CT>(sm + L::bg, dxc + (size_t)b * L::M * L::C);
  const bf16* pb = p_in + (size_t)b * L::N * L::M;
  {
    constexpr int TOT = L::N * L::M / 4, PER = (TOT + 255) / 256, BATCH = PER < 16 ? PER : 16;      // P [N][M] as 8-byte pieces, loads in batches
    static_assert(L::M % 4 == 0, "P rows are staged as 8-byte pieces");
#pragma unroll
    for (int base = 0; base < PER; base += BATCH) {
      bf16x4 r[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) { const int i = t + 256 * (base + j); if (base + j < PER && i < TOT) r[j] = *reinterpret_cast<const bf16x4*>(pb + 4 * i); }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int i = t + 256 * (base + j);
        if (base + j < PER && i < TOT) { const int n = (4 * i) / L::M, m = 4 * i - n * L::M; *reinterpret_cast<bf16x4*>(sm + L::p + n * L::LDM + m) = r[j]; }
      }
    }
  }
  __syncthreads();
  bf16* xw = sm + L::bx + wave * 16 * L::LDC;                  // this wave's x tile
  for (int nt = wave; nt < NT; nt += 4) {
    wave_sync();
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      const int i = lane + 64 * j, n = i / CH, ch = i - n * CH;
      *reinterpret_cast<bf16x4*>(xw + n * L::LDC + 4 * ch) = xr[j];
    }
    req_x(nt + 4);
    wave_sync();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};             // operands swapped: acc[r] = dx[n = 16 nt + col][c = 16 ct + 4 q4 + r], 8-byte row segments
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a = mma16(trfrag(sm + L::bg, L::LDC, mt * 16, ct * 16), rowfrag(sm + L::p, L::LDM, nt * 16, mt * 16), a);
      bf16x4 o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (bf16)a[r];
      *reinterpret_cast<bf16x4*>(dx + ((size_t)b * L::N + nt * 16 + col) * L::C + ct * 16 + 4 * q4) = o4;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) a = mma16(rowfrag(xw, L::LDC, 0, ct * 16), rowfrag(sm + L::bg, L::LDC, mt * 16, ct * 16), a);
#pragma unroll
      for (int r = 0; r < 4; ++r) dP[(nt * 16 + 4 * q4 + r) * L::M + mt * 16 + col] = a[r];
    }
  }
  __syncthreads();
  constexpr int parts = 256 / L::M;
  const int m = t % L::M, part = t / L::M;
  float dot = 0.f;
  for (int n = part; n < L::N; n += parts) dot += (float)sm[L::p + n * L::LDM + m] * dP[n * L::M + m];
  dot = colred(dot, red, L::M, parts, false);
  for (int n = part; n < L::N; n += parts)
    dscores[(size_t)b * L::N * L::M + n * L::M + m] = (bf16)((float)sm[L::p + n * L::LDM + m] * (dP[n * L::M + m] - dot));
}

template <int NT, int MT, int CT>
static int mix2_launch(bool bwd, const void* a0, const void* x, const void* dxc, void* o0, void* o1, int B, hipStream_t st) {
  using L = MixLds<NT, MT, CT>;
  if (!bwd) {
    const size_t smem = (size_t)((L::fwd_bf16 + 7) / 8 * 8) * 2 + 256 * 4;
    if (smem > 150 * 1024) return -100;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokmix2_fwd_kernel<NT, MT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tokmix2_fwd_kernel<NT, MT, CT>), dim3(B), dim3(256), smem, st, (const bf16*)a0, (const bf16*)x, (bf16*)o0, (bf16*)o1, B);
    return QAVIT_OK;
  }
  const size_t smem = (size_t)((L::bwd_bf16 + 7) / 8 * 8) * 2 + (size_t)(L::N * L::M + 256) * 4;
  if (smem > 150 * 1024) return -100;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tokmix2_bwd_kernel<NT, MT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((tokmix2_bwd_kernel<NT, MT, CT>), dim3(B), dim3(256), smem, st, (const bf16*)a0, (const bf16*)x, (const bf16*)dxc, (bf16*)o0, (bf16*)o1, B);
  return QAVIT_OK;
}

// forward: a0 = scores, o0 = p, o1 = xc.   backward: a0 = p, dxc, o0 = dx, o1 = dscores.
int tokmix_bf16_try(bool bwd, const void* a0, const void* x, const void* dxc, void* o0, void* o1, int B, int N, int M, int C, hipStream_t st) {
  if ((reinterpret_cast<uintptr_t>(x) & 7) || (bwd && (reinterpret_cast<uintptr_t>(dxc) & 7)) || C != 192) return 0;
  int rc = -100;
  if (N == 64 && M == 16) rc = mix2_launch<4, 1, 12>(bwd, a0, x, dxc, o0, o1, B, st);
  else if (N == 256 && M == 64) rc = mix2_launch<16, 4, 12>(bwd, a0, x, dxc, o0, o1, B, st);
  if (rc == -100) return 0;
  return rc == QAVIT_OK ? 1 : rc;
}

}  // namespace qv
