// BottleneckMLP + residual of a QuadAttentionBlock in ONE launch each way (HQAViT_CIFAR100.py:643-656, :1082-1083):
//   x1 = x + drop_path( dropout( fc2( dropout( GELU( fc1(y) ) ) ) ) ),   fc1: Linear(192 -> 96), fc2: Linear(96 -> 192)
// y = HybridFusion's output (csrc/cfuse.hip), x = the block input.  It replaces two GEMM launches forward (and the 96-wide intermediate's
// trip through memory as a GEMM operand) and two input-gradient GEMM launches backward; weight gradients stay deferred, grouped
// qavit_gemm_tn problems on the operands this kernel writes (z1 / h1 forward, dz1 / dz2 backward).
//
// Rows are independent: a workgroup of 4 waves takes 64 rows, a wave 16 rows end to end -- no barrier after the operands are staged.
// Both weights (72 KB bf16) sit in LDS, row-major with padded rows (conflict-free fragment reads).  The chain is register-resident:
//   z1^T = W1 y^T           16x16x32 MFMAs, "transposed": acc[r] = z1[row = lane % 16][n = 16 nt + 4 (lane / 16) + r]
//   h1   = drop(GELU(z1))   on the accumulator registers; the bf16 quads ARE the B operands of
//   u^T  = W2 h1^T          16x16x16 MFMAs (a 16x16 accumulator quad = 4 consecutive k of one column: frag16.cuh)
//   x1   = x + dp(drop(u))  epilogue on the accumulators, rows leave through the wave's (dead) LDS tile in 16-byte pieces.
// Backward mirrors it with the transposed weights read from the SAME LDS tiles by ds_read_b64_tr_b16:
//   gm = g * dp * drop2 (written once: operand of dW2),  dh1^T = W2^T gm^T,  dz1 = dh1 * drop1 * GELU'(z1) (operand of dW1),  dy^T = W1^T dz1^T.
// Dropout masks follow the qavit_gemm_nt epilogue contract (drop_factor(key(site), row * N + col); drop path: row / dp_rows).
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "frag16.cuh"

namespace qv {

namespace {

constexpr int MC = 192, MH = 96;                           // channels, hidden
constexpr int MROWS = 64, MNW = 4;                         // rows per workgroup, waves
constexpr int LDW1 = MC + 8, LDW2 = MH + 8, LDT = MC + 8;  // padded LDS rows (elements)
constexpr int SM_W1 = 0, SM_W2 = SM_W1 + MH * LDW1 * 2, SM_Y = SM_W2 + MC * LDW2 * 2, SM_X = SM_Y + MROWS * LDT * 2,
              SM_MLP_FWD = SM_X + MROWS * LDT * 2;         // 38400 + 39936 + 25600 + 25600 = 129536
constexpr int SM_Z = SM_X, SM_MLP_BWD = SM_Z + MROWS * LDW2 * 2;      // backward: g tile at SM_Y, z1 tile [64][104] at SM_Z: 117248

__device__ __forceinline__ bf16x4 c4(const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  return v;
}

// both weights: row-major global [96][192] / [192][96] -> padded LDS tiles; all of a thread's pieces in flight before the first LDS store
__device__ __forceinline__ void stage_weights(const bf16* w1, const bf16* w2, bf16* s1, bf16* s2, int tid) {
  constexpr int P1 = MH * (MC / 8) / 256, P2 = MC * (MH / 8) / 256;      // 9 + 9 sixteen-byte pieces per thread
  bf16x8 r1[P1], r2[P2];
#pragma unroll
  for (int it = 0; it < P1; ++it) { const int p = tid + 256 * it; r1[it] = *reinterpret_cast<const bf16x8*>(w1 + (size_t)(p / 24) * MC + 8 * (p % 24)); }
#pragma unroll
  for (int it = 0; it < P2; ++it) { const int p = tid + 256 * it; r2[it] = *reinterpret_cast<const bf16x8*>(w2 + (size_t)(p / 12) * MH + 8 * (p % 12)); }
#pragma unroll
  for (int it = 0; it < P1; ++it) { const int p = tid + 256 * it; *reinterpret_cast<bf16x8*>(s1 + (p / 24) * LDW1 + 8 * (p % 24)) = r1[it]; }
#pragma unroll
  for (int it = 0; it < P2; ++it) { const int p = tid + 256 * it; *reinterpret_cast<bf16x8*>(s2 + (p / 12) * LDW2 + 8 * (p % 12)) = r2[it]; }
}

template <bool SAVE>
__global__ __launch_bounds__(64 * MNW) void mlp2_fwd_kernel(qavit_mlp2_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16* w1s = reinterpret_cast<bf16*>(smraw + SM_W1);
  bf16* w2s = reinterpret_cast<bf16*>(smraw + SM_W2);
  bf16* yt = reinterpret_cast<bf16*>(smraw + SM_Y);
  bf16* xt = reinterpret_cast<bf16*>(smraw + SM_X);
  const int row0 = blockIdx.x * MROWS;
  const bf16* yg = reinterpret_cast<const bf16*>(a.y);
  const bf16* xg = reinterpret_cast<const bf16*>(a.resid);
  // ---- operand tiles: 64 rows x 24 pieces each, rows past M read the last row (their results are not stored) ----
  bf16x8 yr[6], xr[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int p = tid + 256 * it, r = p / 24, c8 = p - r * 24;
    const int64_t gr = row0 + r < a.M ? row0 + r : a.M - 1;
    yr[it] = *reinterpret_cast<const bf16x8*>(yg + gr * a.ldy + 8 * c8);
    xr[it] = *reinterpret_cast<const bf16x8*>(xg + gr * a.ldr + 8 * c8);
  }
  stage_weights(reinterpret_cast<const bf16*>(a.w1_rm), reinterpret_cast<const bf16*>(a.w2_rm), w1s, w2s, tid);
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int p = tid + 256 * it, r = p / 24, c8 = p - r * 24;
    *reinterpret_cast<bf16x8*>(yt + r * LDT + 8 * c8) = yr[it];
    *reinterpret_cast<bf16x8*>(xt + r * LDT + 8 * c8) = xr[it];
  }
  const bool d1 = a.drop1_p > 0.f && a.rng != nullptr, d2 = a.drop2_p > 0.f && a.rng != nullptr, dpo = a.dp_p > 0.f && a.rng != nullptr;
  const uint32_t k1 = d1 ? rng_key(a.rng, a.drop1_site) : 0u, k2 = d2 ? rng_key(a.rng, a.drop2_site) : 0u, kp = dpo ? rng_key(a.rng, a.dp_site) : 0u;
  const float i1 = d1 ? 1.f / (1.f - a.drop1_p) : 1.f, i2 = d2 ? 1.f / (1.f - a.drop2_p) : 1.f, ip = dpo ? 1.f / (1.f - a.dp_p) : 1.f;
  __syncthreads();

  const int wr = 16 * wave;                                // this wave's rows of the tile
  const int64_t grow = (int64_t)row0 + wr + col;           // this lane's row (accumulator column)
  bf16* ytw = yt + wr * LDT;
  bf16* xtw = xt + wr * LDT;
  // ---- z1^T = W1 y^T + b1 ----
  bf16x8 yf[MC / 32];
#pragma unroll
  for (int ks = 0; ks < MC / 32; ++ks) yf[ks] = *reinterpret_cast<const bf16x8*>(ytw + col * LDT + 32 * ks + 8 * q4);
  f32x4 acc1[MH / 16];
#pragma unroll
  for (int nt = 0; nt < MH / 16; ++nt) acc1[nt] = *reinterpret_cast<const f32x4*>(a.b1 + 16 * nt + 4 * q4);
#pragma unroll
  for (int ks = 0; ks < MC / 32; ++ks)
#pragma unroll
    for (int nt = 0; nt < MH / 16; ++nt)
      acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(w1s + (16 * nt + col) * LDW1 + 32 * ks + 8 * q4), yf[ks], acc1[nt], 0, 0, 0);
  // ---- h1 = dropout(GELU(z1)): operand quads of the second product; z1 / h1 rows for backward through the wave's (dead) y tile ----
  wave_sync();                                             // every lane has its y fragments: the tile can take z1 | h1
  bf16x4 hq[MH / 16];
#pragma unroll
  for (int nt = 0; nt < MH / 16; ++nt) {
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = gelu_f(acc1[nt][r]);
      if (d1) v *= drop_factor(k1, (uint32_t)grow * (uint32_t)MH + (uint32_t)(16 * nt + 4 * q4 + r), a.drop1_p, i1);
      h[r] = v;
    }
    hq[nt] = c4(h);
    if (SAVE) {
      *reinterpret_cast<bf16x4*>(ytw + col * LDT + 16 * nt + 4 * q4) = c4(acc1[nt]);
      *reinterpret_cast<bf16x4*>(ytw + col * LDT + MH + 16 * nt + 4 * q4) = hq[nt];
    }
  }
  if (SAVE) {
    wave_sync();
    bf16* zg = reinterpret_cast<bf16*>(a.z1);
    bf16* hg = reinterpret_cast<bf16*>(a.h1);
#pragma unroll
    for (int it = 0; it < 6; ++it) {                        // 16 rows x (12 + 12) pieces
      const int p = lane + 64 * it, r = p / 24, c8 = p - r * 24;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(ytw + r * LDT + 8 * c8);
      if (row0 + wr + r < a.M) {
        if (c8 < 12) *reinterpret_cast<bf16x8*>(zg + (size_t)(row0 + wr + r) * MH + 8 * c8) = v;
        else *reinterpret_cast<bf16x8*>(hg + (size_t)(row0 + wr + r) * MH + 8 * (c8 - 12)) = v;
      }
    }
  }
  // ---- u^T = W2 h1^T + b2;  x1 = x + drop_path(dropout(u)) ----
  const float dpf = dpo ? drop_factor(kp, (uint32_t)(grow / a.dp_rows), a.dp_p, ip) : 1.f;
#pragma unroll
  for (int ct = 0; ct < MC / 16; ++ct) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.b2 + 16 * ct + 4 * q4);
#pragma unroll
    for (int ks = 0; ks < MH / 16; ++ks) acc = mma16(rowfrag(w2s, LDW2, 16 * ct, 16 * ks), as_s16(hq[ks]), acc);
    const bf16x4 x4 = *reinterpret_cast<const bf16x4*>(xtw + col * LDT + 16 * ct + 4 * q4);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[r];
      if (d2) v *= drop_factor(k2, (uint32_t)grow * (uint32_t)MC + (uint32_t)(16 * ct + 4 * q4 + r), a.drop2_p, i2);
      o[r] = (float)x4[r] + v * dpf;
    }
    *reinterpret_cast<bf16x4*>(xtw + col * LDT + 16 * ct + 4 * q4) = c4(o);       // in place: each lane rewrites the quad it read
  }
  wave_sync();
  bf16* og = reinterpret_cast<bf16*>(a.out);
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int p = lane + 64 * it, r = p / 24, c8 = p - r * 24;
    if (row0 + wr + r < a.M) *reinterpret_cast<bf16x8*>(og + (size_t)(row0 + wr + r) * a.ldo + 8 * c8) = *reinterpret_cast<const bf16x8*>(xtw + r * LDT + 8 * c8);
  }
}

__global__ __launch_bounds__(64 * MNW) void mlp2_bwd_kernel(qavit_mlp2_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16* w1s = reinterpret_cast<bf16*>(smraw + SM_W1);
  bf16* w2s = reinterpret_cast<bf16*>(smraw + SM_W2);
  bf16* gt = reinterpret_cast<bf16*>(smraw + SM_Y);
  bf16* zt = reinterpret_cast<bf16*>(smraw + SM_Z);
  const int row0 = blockIdx.x * MROWS;
  const bf16* gg = reinterpret_cast<const bf16*>(a.g);
  const bf16* zg = reinterpret_cast<const bf16*>(a.z1);
  bf16* dz2g = reinterpret_cast<bf16*>(a.dz2);
  const bool d1 = a.drop1_p > 0.f && a.rng != nullptr, d2 = a.drop2_p > 0.f && a.rng != nullptr, dpo = a.dp_p > 0.f && a.rng != nullptr;
  const uint32_t k1 = d1 ? rng_key(a.rng, a.drop1_site) : 0u, k2 = d2 ? rng_key(a.rng, a.drop2_site) : 0u, kp = dpo ? rng_key(a.rng, a.dp_site) : 0u;
  const float i1 = d1 ? 1.f / (1.f - a.drop1_p) : 1.f, i2 = d2 ? 1.f / (1.f - a.drop2_p) : 1.f, ip = dpo ? 1.f / (1.f - a.dp_p) : 1.f;
  bf16x8 gr[6], zr[3];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int p = tid + 256 * it, r = p / 24, c8 = p - r * 24;
    const int64_t g_ = row0 + r < a.M ? row0 + r : a.M - 1;
    gr[it] = *reinterpret_cast<const bf16x8*>(gg + g_ * a.ldg + 8 * c8);
  }
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = tid + 256 * it, r = p / 12, c8 = p - r * 12;
    const int64_t g_ = row0 + r < a.M ? row0 + r : a.M - 1;
    zr[it] = *reinterpret_cast<const bf16x8*>(zg + g_ * MH + 8 * c8);
  }
  stage_weights(reinterpret_cast<const bf16*>(a.w1_rm), reinterpret_cast<const bf16*>(a.w2_rm), w1s, w2s, tid);
  // gm = g * drop_path * dropout2: into the tile and, once, to memory (operand of dW2)
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int p = tid + 256 * it, r = p / 24, c8 = p - r * 24;
    const int64_t g_ = (int64_t)row0 + r;
    bf16x8 v = gr[it];
    if (d2 || dpo) {
      const float dpf = dpo ? drop_factor(kp, (uint32_t)(g_ / a.dp_rows), a.dp_p, ip) : 1.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v[e] * dpf;
        if (d2) f *= drop_factor(k2, (uint32_t)g_ * (uint32_t)MC + (uint32_t)(8 * c8 + e), a.drop2_p, i2);
        v[e] = (bf16)f;
      }
      if (dz2g && g_ < a.M) *reinterpret_cast<bf16x8*>(dz2g + (size_t)g_ * MC + 8 * c8) = v;
    }
    *reinterpret_cast<bf16x8*>(gt + r * LDT + 8 * c8) = v;
  }
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = tid + 256 * it, r = p / 12, c8 = p - r * 12;
    *reinterpret_cast<bf16x8*>(zt + r * LDW2 + 8 * c8) = zr[it];
  }
  __syncthreads();

  const int wr = 16 * wave;
  const int64_t grow = (int64_t)row0 + wr + col;
  bf16* gtw = gt + wr * LDT;
  bf16* ztw = zt + wr * LDW2;
  // ---- dh1^T = W2^T gm^T (contraction over the 192 channels), dz1 = dh1 * dropout1 * GELU'(z1) ----
  f32x4 acc1[MH / 16];
#pragma unroll
  for (int nt = 0; nt < MH / 16; ++nt) acc1[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < MC / 16; ++ct) {
    const s16x4 gf = rowfrag(gtw, LDT, 0, 16 * ct);        // lane: row = col, 4 consecutive channels
#pragma unroll
    for (int nt = 0; nt < MH / 16; ++nt) acc1[nt] = mma16(trfrag(w2s, LDW2, 16 * ct, 16 * nt), gf, acc1[nt]);   // A[n][c] = W2[c][n]
  }
  bf16x4 dq[MH / 16];
#pragma unroll
  for (int nt = 0; nt < MH / 16; ++nt) {
    const bf16x4 z4 = *reinterpret_cast<const bf16x4*>(ztw + col * LDW2 + 16 * nt + 4 * q4);
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc1[nt][r] * gelu_grad_f((float)z4[r]);
      if (d1) v *= drop_factor(k1, (uint32_t)grow * (uint32_t)MH + (uint32_t)(16 * nt + 4 * q4 + r), a.drop1_p, i1);
      d[r] = v;
    }
    dq[nt] = c4(d);
  }
  wave_sync();                                             // every lane has read its z1 quads: the tile takes dz1
#pragma unroll
  for (int nt = 0; nt < MH / 16; ++nt) *reinterpret_cast<bf16x4*>(ztw + col * LDW2 + 16 * nt + 4 * q4) = dq[nt];
  wave_sync();
  bf16* dz1g = reinterpret_cast<bf16*>(a.dz1);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, r = p / 12, c8 = p - r * 12;
    if (row0 + wr + r < a.M) *reinterpret_cast<bf16x8*>(dz1g + (size_t)(row0 + wr + r) * MH + 8 * c8) = *reinterpret_cast<const bf16x8*>(ztw + r * LDW2 + 8 * c8);
  }
  // ---- dy^T = W1^T dz1^T (contraction over the 96 hidden units): rows through the (dead) g tile ----
#pragma unroll
  for (int ct = 0; ct < MC / 16; ++ct) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < MH / 16; ++nt) acc = mma16(trfrag(w1s, LDW1, 16 * nt, 16 * ct), as_s16(dq[nt]), acc);      // A[c][n] = W1[n][c]
    // this wave's g fragments of column tile ct were consumed above (all 12 gf reads precede this loop in program order)
    *reinterpret_cast<bf16x4*>(gtw + col * LDT + 16 * ct + 4 * q4) = c4(acc);
  }
  wave_sync();
  bf16* dyg = reinterpret_cast<bf16*>(a.dy);
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int p = lane + 64 * it, r = p / 24, c8 = p - r * 24;
    if (row0 + wr + r < a.M) *reinterpret_cast<bf16x8*>(dyg + (size_t)(row0 + wr + r) * a.lddy + 8 * c8) = *reinterpret_cast<const bf16x8*>(gtw + r * LDT + 8 * c8);
  }
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_mlp2_supported(int C, int Hd) { return (C == MC && Hd == MH) ? 1 : 0; }

extern "C" int qavit_mlp2_fwd(const qavit_mlp2_args* a, void* stream) {
  if (!a) return set_error(QAVIT_EINVAL, "mlp2: null args");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "mlp2: bf16 only (fp32 runs the two qavit_gemm_nt launches)");
  if (a->C != MC || a->Hd != MH || a->M <= 0) return set_error(QAVIT_EINVAL, "mlp2: built for Linear(192 -> 96) -> Linear(96 -> 192)");
  if (!a->y || !a->resid || !a->w1_rm || !a->w2_rm || !a->b1 || !a->b2 || !a->out) return set_error(QAVIT_EINVAL, "mlp2: null operand");
  if ((a->z1 == nullptr) != (a->h1 == nullptr)) return set_error(QAVIT_EINVAL, "mlp2: z1 and h1 (the backward pass's operands) come together");
  if ((a->drop1_p > 0.f || a->drop2_p > 0.f || a->dp_p > 0.f) && !a->rng) return set_error(QAVIT_EINVAL, "mlp2: dropout requested without rng state");
  if (a->dp_p > 0.f && a->dp_rows <= 0) return set_error(QAVIT_EINVAL, "mlp2: drop path needs rows-per-sample");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al(a->y) || !al(a->resid) || !al(a->w1_rm) || !al(a->w2_rm) || !al(a->b1) || !al(a->b2) || !al(a->out) || (a->z1 && (!al(a->z1) || !al(a->h1))) ||
      a->ldy % 8 || a->ldr % 8 || a->ldo % 8)
    return set_error(QAVIT_EINVAL, "mlp2: operands must be 16-byte aligned with leading dimensions a multiple of 8 elements");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done[2] = {false, false};
  const int grid = (a->M + MROWS - 1) / MROWS;
  if (a->z1) {
    if (!attr_done[1]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp2_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SM_MLP_FWD); attr_done[1] = true; }
    hipLaunchKernelGGL((mlp2_fwd_kernel<true>), dim3(grid), dim3(64 * MNW), SM_MLP_FWD, st, *a);
  } else {
    if (!attr_done[0]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp2_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SM_MLP_FWD); attr_done[0] = true; }
    hipLaunchKernelGGL((mlp2_fwd_kernel<false>), dim3(grid), dim3(64 * MNW), SM_MLP_FWD, st, *a);
  }
  return check_launch("mlp2_fwd");
}

extern "C" int qavit_mlp2_bwd(const qavit_mlp2_bwd_args* a, void* stream) {
  if (!a) return set_error(QAVIT_EINVAL, "mlp2_bwd: null args");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "mlp2_bwd: bf16 only");
  if (a->C != MC || a->Hd != MH || a->M <= 0) return set_error(QAVIT_EINVAL, "mlp2_bwd: built for Linear(192 -> 96) -> Linear(96 -> 192)");
  if (!a->g || !a->z1 || !a->w1_rm || !a->w2_rm || !a->dz1 || !a->dy) return set_error(QAVIT_EINVAL, "mlp2_bwd: null operand");
  if ((a->drop1_p > 0.f || a->drop2_p > 0.f || a->dp_p > 0.f) && !a->rng) return set_error(QAVIT_EINVAL, "mlp2_bwd: dropout requested without rng state");
  if ((a->drop2_p > 0.f || a->dp_p > 0.f) && !a->dz2) return set_error(QAVIT_EINVAL, "mlp2_bwd: dropout / drop path on the output needs dz2 (the masked gradient is the operand of dW2)");
  if (a->dp_p > 0.f && a->dp_rows <= 0) return set_error(QAVIT_EINVAL, "mlp2_bwd: drop path needs rows-per-sample");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al(a->g) || !al(a->z1) || !al(a->w1_rm) || !al(a->w2_rm) || !al(a->dz1) || !al(a->dy) || (a->dz2 && !al(a->dz2)) || a->ldg % 8 || a->lddy % 8)
    return set_error(QAVIT_EINVAL, "mlp2_bwd: operands must be 16-byte aligned with leading dimensions a multiple of 8 elements");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp2_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_MLP_BWD); attr_done = true; }
  hipLaunchKernelGGL(mlp2_bwd_kernel, dim3((a->M + MROWS - 1) / MROWS), dim3(64 * MNW), SM_MLP_BWD, st, *a);
  return check_launch("mlp2_bwd");
}
