// GlobalTokenBank.write (train mode, no gradient).  See include/qavit.h.
//   stats : per image  tn = LN_write(LN_branch(tokens));  w = softmax_tokens(tn Wg^T + bg);  U += w^T tn
//           (workgroup partials -> one reduction into acc[S,C])
//   apply : U/B -> clamp -> bank update (+ the Linear(192,192) applied to the 16 mean rows instead of to
//           every token: exact because each softmax column sums to one)
#include "common.cuh"
#include "mma_lds.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "tokens_shared.h"

namespace qv {

constexpr int BANK_MAX_WG = 256;

// NCH = token rows resident in LDS at a time.  NCH >= N: one pass.  Otherwise (N = 196 at 224 px) two passes over the
// chunks: logits of every chunk first (the softmax runs over ALL tokens of the image), then the chunks are staged and
// normalised again for U += w^T tn -- the re-read comes from L2.
template <typename T, bool BF, int NTH>
__global__ __launch_bounds__(NTH) void bank_stats_kernel(const T* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr,
                                                         const float* Wg, const float* bg, float* ws, int B, int N, int C, int S, int NCH, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tn = sm;                    // [NCH][C]
  float* lg = tn + NCH * C;          // [N][S] logits -> weights
  float* U = lg + N * S;             // [S][C] accumulator over this workgroup's images
  float* red = U + S * C;            // [NTH]
  constexpr int NW = NTH / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float invC = 1.f / (float)C;
  for (int i = t; i < S * C; i += NTH) U[i] = 0.f;
  // rows [n0, n0 + rows) of image b -> tn, through the two chained LayerNorms
  auto stage_ln = [&](int b, int n0, int rows) {
    __syncthreads();
    for (int i = t; i < rows * C; i += NTH) tn[i] = to_f<T>(tokens[((size_t)b * N + n0) * C + i]);
    __syncthreads();
    for (int r = wave; r < rows; r += NW) {
      float* row = tn + r * C;
      for (int pass = 0; pass < 2; ++pass) {
        const float* g = pass == 0 ? gbr : gwr;
        const float* bb = pass == 0 ? bbr : bwr;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += row[c];
        const float mean = wave_sum(s) * invC;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = row[c] - mean; s2 += d * d; }
        const float rstd = rsqrtf(wave_sum(s2) * invC + eps);
        for (int c = lane; c < C; c += 64) row[c] = (row[c] - mean) * rstd * g[c] + bb[c];
      }
    }
    __syncthreads();
  };
  // logits[n0 + n][s] = tn[n,:] . Wg[s,:] + bg[s]
  auto logits = [&](int n0, int rows) {
    const int nt_n = (rows + 15) / 16, st_n = (S + 15) / 16;
    for (int tile = wave; tile < nt_n * st_n; tile += NW) {
      const int nt = tile / st_n, stt = tile - nt * st_n;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mma_tile<BF>(tn + nt * 16 * C, C, 1, rows - nt * 16, Wg + (size_t)stt * 16 * C, 1, C, S - stt * 16, C, acc);
      const int col = tile_col();
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = nt * 16 + tile_row(reg);
        if (row < rows && stt * 16 + col < S) lg[(n0 + row) * S + stt * 16 + col] = acc[reg] + bg[stt * 16 + col];
      }
    }
  };
  // U[s][c] += sum_n w[n0 + n][s] tn[n][c]
  auto accum = [&](int n0, int rows) {
    const int st_n = (S + 15) / 16, ct_n = (C + 15) / 16;
    for (int tile = wave; tile < st_n * ct_n; tile += NW) {
      const int stt = tile / ct_n, ct = tile - stt * ct_n;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mma_tile<BF>(lg + n0 * S + stt * 16, 1, S, S - stt * 16, tn + ct * 16, C, 1, C - ct * 16, rows, acc);
      tile_to_f32<true>(U + stt * 16 * C + ct * 16, C, 1, S - stt * 16, C - ct * 16, acc);
    }
  };
  const bool one_pass = NCH >= N;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    for (int n0 = 0; n0 < N; n0 += NCH) {
      const int rows = N - n0 < NCH ? N - n0 : NCH;
      stage_ln(b, n0, rows);
      logits(n0, rows);
    }
    __syncthreads();
    // softmax over tokens per slot: S columns, NTH/S threads per column (S divides 256)
    {
      const int parts = NTH / S, s_ = t % S, part = t / S;
      float mx = -INFINITY;
      for (int n = part; n < N; n += parts) mx = fmaxf(mx, lg[n * S + s_]);
      red[t] = mx; __syncthreads();
      mx = -INFINITY;
      for (int p = 0; p < parts; ++p) mx = fmaxf(mx, red[p * S + s_]);
      __syncthreads();
      float sum = 0.f;
      for (int n = part; n < N; n += parts) { const float e = __expf(lg[n * S + s_] - mx); lg[n * S + s_] = e; sum += e; }
      red[t] = sum; __syncthreads();
      sum = 0.f;
      for (int p = 0; p < parts; ++p) sum += red[p * S + s_];
      const float inv = 1.f / sum;
      for (int n = part; n < N; n += parts) lg[n * S + s_] *= inv;
    }
    __syncthreads();
    if (one_pass) accum(0, N);
    else
      for (int n0 = 0; n0 < N; n0 += NCH) {
        const int rows = N - n0 < NCH ? N - n0 : NCH;
        stage_ln(b, n0, rows);
        accum(n0, rows);
      }
  }
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * S * C;
  for (int i = t; i < S * C; i += NTH) out[i] = U[i];
}

// acc (zero on entry: bank_apply re-zeroes what it consumes) += partials; blockIdx.y = slice of 16 partials
__global__ __launch_bounds__(256) void bank_reduce_kernel(const float* ws, float* acc, int nparts, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w0 = blockIdx.y * 16;
  const int w1 = (w0 + 16 < nparts) ? w0 + 16 : nparts;
  float s = 0.f;
  for (int w = w0; w < w1; ++w) s += ws[(size_t)w * n + i];
  atomic_add_f(acc + i, s);
}

// one workgroup per slot s.  ``parts`` (optional): the per-workgroup partial sums of bank_stats, [nparts][S][C]; the
// slot's row is folded here in a fixed order (threads = (group, 4-column vector), groups stride over the partials, then
// an LDS fold over the groups) -- the single-GPU write is stats -> apply, deterministic, no reduce launch.  Without
// ``parts`` the row comes from ``acc`` (already summed, and all-reduced across ranks when data-parallel).
__global__ __launch_bounds__(1024) void bank_apply_kernel(float* __restrict__ acc, const float* __restrict__ Wc, const float* __restrict__ bc,
                                                          float* __restrict__ bank_k, float* __restrict__ bank_v,
                                                          int64_t* update_count, int S, int C, float inv_batch, int mode,
                                                          const float* __restrict__ parts, int nparts, float* __restrict__ snap_k, float* __restrict__ snap_v) {
  extern __shared__ __attribute__((aligned(16))) float u[];   // [C] + fold scratch [groups][C]
  const int s = blockIdx.x;
  if (parts) {
    float* fold = u + C;
    const int C4 = C >> 2, groups = (int)blockDim.x / C4;
    const int g = threadIdx.x / C4, c4 = threadIdx.x - g * C4;
    if (g < groups) {
      // four independent partial sums: the loads of a thread are a latency chain otherwise (nparts / groups round trips)
      f32x4 a4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* base = parts + (size_t)s * C + 4 * c4;
      const size_t stride = (size_t)S * C;
      int p_ = g;
      for (; p_ + 3 * groups < nparts; p_ += 4 * groups) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(base + (size_t)(p_ + q * groups) * stride);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a4[q][0] += v[q][0]; a4[q][1] += v[q][1]; a4[q][2] += v[q][2]; a4[q][3] += v[q][3]; }
      }
      for (; p_ < nparts; p_ += groups) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p_ * stride);
        a4[0][0] += v[0]; a4[0][1] += v[1]; a4[0][2] += v[2]; a4[0][3] += v[3];
      }
      f32x4 t4;
#pragma unroll
      for (int j = 0; j < 4; ++j) t4[j] = (a4[0][j] + a4[1][j]) + (a4[2][j] + a4[3][j]);
      *reinterpret_cast<f32x4*>(fold + g * C + 4 * c4) = t4;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float t = 0.f;
      for (int q = 0; q < groups; ++q) t += fold[q * C + c];
      u[c] = t * inv_batch;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += blockDim.x) { u[c] = acc[s * C + c] * inv_batch; acc[s * C + c] = 0.f; }   // consumed: zero for the next write
  }
  float rate, cu, cb;
  if (mode == 1) { rate = 0.01f; cu = 0.1f; cb = 1.0f; }
  else { rate = (update_count && update_count[0] >= 1000) ? 0.01f : 0.005f; cu = 0.05f; cb = 0.5f; }
  __syncthreads();
  // update_count += 1 once per write, by the LAST workgroup to get here: every workgroup has read the counter by then
  if (mode == 0 && threadIdx.x == 0) {
    int* ticket = reinterpret_cast<int*>(acc + (size_t)S * C);
    if (atomicAdd(ticket, 1) == S - 1) { update_count[0] += 1; *ticket = 0; }
  }
  // upd_k = Wc u + bc: one WAVE per output row (lanes stride over k: coalesced 256-byte reads of the weight row, then a wave
  // reduction), four rows in flight per wave (their weight loads are issued before the first reduction).
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (int)blockDim.x >> 6;
  for (int c0 = wv; c0 < C; c0 += 4 * nwv) {
    float part[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * nwv;
      part[q] = 0.f;
      if (c < C) {
        const float* wr = Wc + (size_t)c * C;
        for (int k = lane; k < C; k += 64) part[q] += u[k] * wr[k];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) part[q] = wave_sum(part[q]);
    if (lane < 4) {
      const int c = c0 + lane * nwv;
      if (c < C) {
        const float pq = lane == 0 ? part[0] : lane == 1 ? part[1] : lane == 2 ? part[2] : part[3];
        const float dotk = pq + bc[c];
        const float uk = fminf(fmaxf(dotk, -cu), cu);
        const float uv = fminf(fmaxf(u[c], -cu), cu);
        const float nk = bank_k[s * C + c] + rate * uk;
        const float nv = bank_v[s * C + c] + rate * uv;
        const float ok_ = fminf(fmaxf(nk, -cb), cb), ov_ = fminf(fmaxf(nv, -cb), cb);
        bank_k[s * C + c] = ok_;
        bank_v[s * C + c] = ov_;
        if (snap_k) { snap_k[s * C + c] = ok_; snap_v[s * C + c] = ov_; }
      }
    }
  }
}

// The same write for C <= 192 with 1024 threads, built around ONE memory round trip: a wave's weight rows (12 rows x 3 values per lane),
// its old bank values and the thread's share of the partial sums are all requested before anything is consumed -- the kernel above is
// a chain of dependent loads (fold -> u -> weight rows -> bank rows) on 16 workgroups, ~20 us for 0.3 MFLOP, and it sits on the
// forward critical path three times per block (the next branch reads the bank this one writes).
__global__ __launch_bounds__(1024) void bank_apply192_kernel(float* __restrict__ acc, const float* __restrict__ Wc, const float* __restrict__ bc,
                                                             float* __restrict__ bank_k, float* __restrict__ bank_v,
                                                             int64_t* update_count, int S, int C, float inv_batch, int mode,
                                                             const float* __restrict__ parts, int nparts, float* __restrict__ snap_k, float* __restrict__ snap_v) {
  extern __shared__ __attribute__((aligned(16))) float u[];   // [C] + fold scratch [groups][C]
  constexpr int RW = 12, KW = 3, NWV = 16, FMAX = 16;
  const int s = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // provably wave-uniform: the per-row scalars below (bias, old bank values) then take
                                                                        // scalar loads into SGPRs -- 36 registers per lane less (the kernel spilled at 128)
  float w[RW][KW], bcr[RW], okr[RW], ovr[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int c = wv + NWV * i, cc = c < C ? c : 0;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const int k = lane + 64 * j;
      w[i][j] = Wc[(size_t)cc * C + (k < C ? k : 0)];
    }
    bcr[i] = bc[cc]; okr[i] = bank_k[s * C + cc]; ovr[i] = bank_v[s * C + cc];
  }
  const long cnt = (mode == 0 && update_count) ? update_count[0] : 0;
  const int C4 = C >> 2, groups = 1024 / C4;
  const int g = threadIdx.x / C4, c4 = threadIdx.x - g * C4;
  float* fold = u + C;
  if (parts) {
    f32x4 v[FMAX];
    const float* base = parts + (size_t)s * C + 4 * (g < groups ? c4 : 0);
    const size_t stride = (size_t)S * C;
#pragma unroll
    for (int q = 0; q < FMAX; ++q) {
      const int p_ = g + q * groups;
      v[q] = *reinterpret_cast<const f32x4*>(base + (size_t)(p_ < nparts ? p_ : 0) * stride);
    }
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < FMAX; ++q) {
      if (g + q * groups < nparts) { a4[0] += v[q][0]; a4[1] += v[q][1]; a4[2] += v[q][2]; a4[3] += v[q][3]; }
    }
    for (int p_ = g + FMAX * groups; p_ < nparts; p_ += groups) {          // more partials than the unrolled window (not with <= 256 statistics workgroups)
      const f32x4 t = *reinterpret_cast<const f32x4*>(base + (size_t)p_ * stride);
      a4[0] += t[0]; a4[1] += t[1]; a4[2] += t[2]; a4[3] += t[3];
    }
    if (g < groups) *reinterpret_cast<f32x4*>(fold + g * C + 4 * c4) = a4;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 1024) {
      float t = 0.f;
      for (int q = 0; q < groups; ++q) t += fold[q * C + c];
      u[c] = t * inv_batch;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 1024) { u[c] = acc[s * C + c] * inv_batch; acc[s * C + c] = 0.f; }   // consumed: zero for the next write
  }
  float rate, cu, cb;
  if (mode == 1) { rate = 0.01f; cu = 0.1f; cb = 1.0f; }
  else { rate = cnt >= 1000 ? 0.01f : 0.005f; cu = 0.05f; cb = 0.5f; }
  __syncthreads();
  if (mode == 0 && threadIdx.x == 0) {                       // once per write, by the last workgroup: every workgroup has read the counter by then
    int* ticket = reinterpret_cast<int*>(acc + (size_t)S * C);
    if (atomicAdd(ticket, 1) == S - 1) { update_count[0] += 1; *ticket = 0; }
  }
  float uk[KW];
#pragma unroll
  for (int j = 0; j < KW; ++j) { const int k = lane + 64 * j; uk[j] = k < C ? u[k] : 0.f; }
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < KW; ++j) part += uk[j] * w[i][j];
    part = wave_sum(part);
    const int c = wv + NWV * i;
    if (lane == 0 && c < C) {
      const float dk = fminf(fmaxf(part + bcr[i], -cu), cu);
      const float dv = fminf(fmaxf(u[c], -cu), cu);
      const float nk = fminf(fmaxf(okr[i] + rate * dk, -cb), cb), nv = fminf(fmaxf(ovr[i] + rate * dv, -cb), cb);
      bank_k[s * C + c] = nk;
      bank_v[s * C + c] = nv;
      if (snap_k) { snap_k[s * C + c] = nk; snap_v[s * C + c] = nv; }
    }
  }
}

static int bank_grid(int B) { return B < BANK_MAX_WG ? B : BANK_MAX_WG; }

}  // namespace qv

using namespace qv;

extern "C" int64_t qavit_bank_ws_floats(int B, int N, int C, int S) {
  (void)N;
  return (int64_t)bank_grid(B) * S * C;
}

namespace qv {
void branch_nan_fix_launch(void* out, int64_t ldo, int rows, int C, const float* bias, float p, int site, const int64_t* rng, int* flag,
                           int* trip, void* o_save, int64_t ldos, int Co, hipStream_t st);      // branch_fwd.hip
}

extern "C" int qavit_bank_stats(int dtype, const void* tokens, const float* g_branch, const float* b_branch,
                                const float* g_write, const float* b_write, const float* Wg, const float* bg,
                                float* acc, float* ws, int64_t ws_floats, int B, int N, int C, int S, float eps, void* stream) {
  return qavit_bank_stats_nanfix(dtype, const_cast<void*>(tokens), g_branch, b_branch, g_write, b_write, Wg, bg, acc, ws, ws_floats, B, N, C, S, eps, nullptr, stream);
}

extern "C" int qavit_branch_nan_fix(int dtype, void* out, int rows, int C, const qavit_nan_fix* fix, void* stream) {
  if (dtype != QAVIT_BF16 || !out || rows <= 0 || C <= 0 || !fix || !fix->flag || !fix->bias) return set_error(QAVIT_EINVAL, "branch_nan_fix: bf16 rows, flag and bias");
  if (fix->drop_p < 0.f || fix->drop_p >= 1.f || (fix->drop_p > 0.f && !fix->rng)) return set_error(QAVIT_EINVAL, "branch_nan_fix: dropout needs rng");
  qv::branch_nan_fix_launch(out, C, rows, C, fix->bias, fix->drop_p, fix->drop_site, fix->rng, fix->flag, fix->trip, fix->o_save, fix->ldos, fix->Co,
                            reinterpret_cast<hipStream_t>(stream));
  return check_launch("branch_nan_fix");
}

extern "C" int qavit_bank_stats_nanfix(int dtype, void* tokens, const float* g_branch, const float* b_branch,
                                       const float* g_write, const float* b_write, const float* Wg, const float* bg,
                                       float* acc, float* ws, int64_t ws_floats, int B, int N, int C, int S, float eps,
                                       const qavit_nan_fix* fix, void* stream) {
  if (fix && (!fix->flag || !fix->bias || (fix->drop_p > 0.f && !fix->rng) || fix->drop_p < 0.f || fix->drop_p >= 1.f || dtype != QAVIT_BF16))
    return set_error(QAVIT_EINVAL, "bank_stats: the deferred NaN rule needs flag, bias, rng with dropout, bf16 tokens");
  if (!tokens || !g_branch || !b_branch || !g_write || !b_write || !Wg || !bg || !ws) return set_error(QAVIT_EINVAL, "bank_stats: null operand");
  if (B <= 0 || N <= 0 || C <= 0 || S <= 0 || S > 256 || (256 % S) != 0) return set_error(QAVIT_EINVAL, "bank_stats: bad dimensions (S must divide 256)");
  const int grid = bank_grid(B);
  if (ws_floats < (int64_t)grid * S * C) return set_error(QAVIT_EINVAL, "bank_stats: workspace too small");
  const size_t fixed = ((size_t)N * S + (size_t)S * C + 1024) * sizeof(float);
  int NCH = N;
  if (fixed + (size_t)NCH * C * sizeof(float) > 160 * 1024) {
    if (fixed + (size_t)16 * C * sizeof(float) > 160 * 1024) return set_error(QAVIT_EINVAL, "bank_stats: token tile too large for LDS");
    NCH = (int)((160 * 1024 - fixed) / ((size_t)C * sizeof(float))) / 16 * 16;
    const int chunks = (N + NCH - 1) / NCH;                // balance the chunks
    NCH = ((N + chunks - 1) / chunks + 15) / 16 * 16;
  }
  const size_t smem = fixed + (size_t)NCH * C * sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int n_acc = S * C;
  if (dtype == QAVIT_BF16) {
    const int took = bank_stats_bf16_try(tokens, g_branch, b_branch, g_write, b_write, Wg, bg, ws, B, N, C, S, grid, eps, st, fix);
    if (took < 0) return took;
    if (took == 1) {
      if (acc) hipLaunchKernelGGL(bank_reduce_kernel, dim3((n_acc + 255) / 256, (grid + 15) / 16), dim3(256), 0, st, ws, acc, grid, n_acc);
      return check_launch("bank_stats(bf16)");
    }
  }
  if (fix)      // no kernel here carries the deferred rule for this shape: the rule's own launch first
    qv::branch_nan_fix_launch(tokens, C, B * N, C, fix->bias, fix->drop_p, fix->drop_site, fix->rng, fix->flag, fix->trip, fix->o_save, fix->ldos, fix->Co, st);
  // one image per workgroup: a 196-token image (two chunks) gets 16 waves to share its rows and MFMA tiles
#define BANK_LAUNCH(T_, BF_, NTH_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bank_stats_kernel<T_, BF_, NTH_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((bank_stats_kernel<T_, BF_, NTH_>), dim3(grid), dim3(NTH_), smem, st, (const T_*)tokens, g_branch, b_branch, g_write, b_write, Wg, bg, ws, B, N, C, S, NCH, eps); }
  const bool wide = N > 64;
  if (dtype == QAVIT_F32) { if (wide) BANK_LAUNCH(float, false, 1024) else BANK_LAUNCH(float, false, 256) }
  else if (dtype == QAVIT_BF16) { if (wide) BANK_LAUNCH(bf16, true, 1024) else BANK_LAUNCH(bf16, true, 256) }
  else return set_error(QAVIT_EINVAL, "bank_stats: unknown dtype");
#undef BANK_LAUNCH
  const int n = S * C;
  if (acc) hipLaunchKernelGGL(bank_reduce_kernel, dim3((n + 255) / 256, (grid + 15) / 16), dim3(256), 0, st, ws, acc, grid, n);
  return check_launch("bank_stats");
}

extern "C" int qavit_bank_apply(float* acc, const float* Wc, const float* bc, float* bank_k, float* bank_v,
                                int64_t* update_count, int S, int C, float inv_batch, int mode,
                                const float* parts, int nparts, float* snap_k, float* snap_v, void* stream) {
  if (!acc || !Wc || !bc || !bank_k || !bank_v || S <= 0 || C <= 0) return set_error(QAVIT_EINVAL, "bank_apply: bad arguments");
  if (mode == 0 && !update_count) return set_error(QAVIT_EINVAL, "bank_apply: mode 0 needs update_count");
  if ((snap_k == nullptr) != (snap_v == nullptr)) return set_error(QAVIT_EINVAL, "bank_apply: snap_k and snap_v come together");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (parts && (nparts <= 0 || C % 4 || C / 4 > 1024 || (reinterpret_cast<uintptr_t>(parts) & 15)))
    return set_error(QAVIT_EINVAL, "bank_apply: partials need C % 4 == 0, 16-byte alignment");
  if (C <= 192 && C % 4 == 0) {
    const int groups = 1024 / (C / 4);
    hipLaunchKernelGGL(bank_apply192_kernel, dim3(S), dim3(1024), (size_t)(1 + groups) * C * sizeof(float), st, acc, Wc, bc, bank_k, bank_v, update_count, S, C,
                       inv_batch, mode, parts, nparts, snap_k, snap_v);
  } else if (parts) {
    const int groups = 1024 / (C / 4);
    hipLaunchKernelGGL(bank_apply_kernel, dim3(S), dim3(1024), (size_t)(1 + groups) * C * sizeof(float), st, acc, Wc, bc, bank_k, bank_v, update_count, S, C, inv_batch, mode, parts, nparts, snap_k, snap_v);
  } else {
    hipLaunchKernelGGL(bank_apply_kernel, dim3(S), dim3(256), (size_t)C * sizeof(float), st, acc, Wc, bc, bank_k, bank_v, update_count, S, C, inv_batch, mode, (const float*)nullptr, 0, snap_k, snap_v);
  }
  return check_launch("bank_apply");
}
