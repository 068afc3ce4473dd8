// Fused CHANNEL-GROUP attention branch, forward (EfficientChannelGroupAttention, HQAViT_CIFAR100.py:535-595) for the 16-learned-token
// problems of the CIFAR configuration:  out = dropout( proj( concat_g SDPA_g ) ),  SDPA_g over 4 heads of D = 4 on
// q / k / v = Linear(32 -> 16) of channel group g, keys = [16 tokens ; 16 projected bank rows].  One launch replaces the stacked
// q/k/v GEMM (98304 x 32 -> 48), the attention kernel (24576 problems of 16 x 32 x 4), its NaN guard and the proj GEMM.
//
// The problems are far too small to tile: everything about an image fits one WAVE.  A workgroup of 4 waves takes 4 images (grid =
// B / 4 = one workgroup per CU at B = 1024); the proj weight sits in LDS as 16x16x16 MFMA fragments, the q/k/v weights (shared by
// all groups) in registers.  Per (image, group), with acc[r] = C[4 q4 + r][col] and operand quads = 4 consecutive k of row col:
//   q^T = Wq x_g^T          acc = q[token = col][d = 4 q4 + r]  -> B operand of S^T;  lane group q4 holds exactly head q4's 4 dims
//   k^T = Wk x_g^T          acc = k[token = col][d ..]          -> A operand of S^T
//   v   = x_g Wv^T          acc = v[token = 4 q4 + r][d = col]  -> A operand of O^T
//   S_h^T = K (Q masked to lane group h)^T   (the mask makes one 16-deep MFMA contract over head h's 4 dims only)
//   softmax over the 32 keys on the accumulator registers (+ dropout), P_h^T as B operand of
//   O_h^T = V^T P_h^T       valid in lane group h (d in head h): the four heads' results are merged by lane group
// then the 6 groups' O quads meet in the image's LDS tile and out^T = Wp O^T runs on the LDS fragments.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include "frag16.cuh"

namespace qv {

int cga64_fwd_launch(const qavit_cga_args* a, hipStream_t st);       // cga64.hip: the same branch on 64-token problems
int cga64_bwd_launch(const qavit_cga_bwd_args* a, hipStream_t st);
void branch_nan_fix_launch(void* out, int64_t ldo, int rows, int C, const float* bias, float p, int site, const int64_t* rng, int* flag,
                           int* trip, void* o_save, int64_t ldos, int Co, hipStream_t st);

namespace {

constexpr int CT = 16, CC = 192, CG = 6, CPG = 32, CCG = 16, CH = 4, CD = 4, CS = 16, CO = CG * CCG;   // tokens, channels, groups, ch/group, q dims/group, heads, head dim, bank rows, 96
constexpr int CNI = 4;                                     // images per workgroup
constexpr int CNW = 8;                                     // waves per workgroup: TWO per image (three channel groups each) -- the chain of one
                                                           // image is long and serial (softmax, dropout hashes, lane permutes between small MFMAs),
                                                           // a second wave per SIMD hides its latencies
constexpr int LDX = CC + 8, LDOO = CO + 8;
constexpr int WP_FRAGS = (CC / 16) * (CO / 16);            // 12 x 6 fragments of 512 B
constexpr int SM_WP = 0, SM_XT = WP_FRAGS * 512, SM_OT = SM_XT + CNI * CT * LDX * 2, SM_CGA = SM_OT + CNI * CT * LDOO * 2;   // 36864 + 25600 + 13312 = 75776 bytes

__device__ __forceinline__ bf16x4 cvt4c(const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  return v;
}

__global__ __launch_bounds__(64 * CNW) void cga_fwd_kernel(qavit_cga_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q4 = lane >> 4;
  bf16* swp = reinterpret_cast<bf16*>(smraw + SM_WP);
  const int wimg = wave >> 1, half = wave & 1;             // this wave: image wimg of the tile, channel groups 3 half .. 3 half + 2
  bf16* xt = reinterpret_cast<bf16*>(smraw + SM_XT) + wimg * (CT * LDX);
  bf16* ot = reinterpret_cast<bf16*>(smraw + SM_OT) + wimg * (CT * LDOO);
  const int img_raw = blockIdx.x * CNI + wimg;
  const bool valid = img_raw < a.B;
  const int img = valid ? img_raw : a.B - 1;
  const bf16* xg = reinterpret_cast<const bf16*>(a.x);
  const bf16* wq = reinterpret_cast<const bf16*>(a.wqkv_rm);
  const bf16* wp = reinterpret_cast<const bf16*>(a.wproj_rm);
  bf16* og = reinterpret_cast<bf16*>(a.out);
  bf16* osv = reinterpret_cast<bf16*>(a.o_save);

  // ---- loads: the image's token tile, the proj weight as fragments, q/k/v weights, biases, bank rows ----
  bf16x8 xr[3];                                            // rows 8 half .. + 8 of the image's token tile
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
    xr[it] = *reinterpret_cast<const bf16x8*>(xg + ((size_t)img * CT + row) * a.ldx + 8 * c8);
  }
  bf16x4 wpr[WP_FRAGS / CNW];                              // 9 quads per thread
#pragma unroll
  for (int it = 0; it < WP_FRAGS / CNW; ++it) {
    const int f = it * CNW + wave, ctile = f / (CO / 16), otile = f - ctile * (CO / 16);
    wpr[it] = *reinterpret_cast<const bf16x4*>(wp + (size_t)(16 * ctile + col) * CO + 16 * otile + 4 * q4);
  }
  s16x4 wqf[3][2];                                         // [part][k-step]: lane = output dim d (col), 4 consecutive input channels
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wqf[p][ks] = as_s16(*reinterpret_cast<const bf16x4*>(wq + (size_t)(16 * p + col) * CPG + 16 * ks + 4 * q4));
  f32x4 bq, bk;
  float bv;
  bq = *reinterpret_cast<const f32x4*>(a.bqkv + 4 * q4);
  bk = *reinterpret_cast<const f32x4*>(a.bqkv + CCG + 4 * q4);
  bv = a.bqkv[2 * CCG + col];
  const f32x4 shk4 = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)col * CCG + 4 * q4);      // bank key s = col, dims 4 q4 ..
  float shv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) shv[i] = a.sh_v[(size_t)(4 * q4 + i) * CCG + col];               // bank value rows 4 q4 + i, dim col
  // ---- consumers ----
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
    *reinterpret_cast<bf16x8*>(xt + row * LDX + 8 * c8) = xr[it];
  }
#pragma unroll
  for (int it = 0; it < WP_FRAGS / CNW; ++it) {
    const int f = it * CNW + wave;
    *reinterpret_cast<bf16x4*>(swp + ((size_t)f * 64 + lane) * 4) = wpr[it];
  }
  bool bad = false;
  s16x4 bkA, bvP;
  {
    bf16x4 t1, t2;
#pragma unroll
    for (int i = 0; i < 4; ++i) { bad |= (shk4[i] != shk4[i]) | (shv[i] != shv[i]); t1[i] = (bf16)shk4[i]; t2[i] = (bf16)shv[i]; }
    bkA = as_s16(t1); bvP = as_s16(t2);
  }
  const bool adrop = a.attn_drop_p > 0.f && a.rng != nullptr;
  AttnDrop drop;
  drop.on = adrop;
  drop.p = adrop ? a.attn_drop_p : 0.f;
  drop.inv_keep = adrop ? 1.f / (1.f - a.attn_drop_p) : 1.f;
  drop.key = adrop ? rng_key(a.rng, a.attn_drop_site) : 0u;
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pp = pdrop ? a.proj_drop_p : 0.f, pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;
  const float scale = 0.5f;                                // 1 / sqrt(D = 4)
  const s16x4 zero_s = {0, 0, 0, 0};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                         // the proj fragments and the token tiles are complete

  for (int g = 3 * half; g < 3 * half + 3; ++g) {
    const s16x4 xf0 = rowfrag(xt, LDX, 0, CPG * g), xf1 = rowfrag(xt, LDX, 0, CPG * g + 16);   // lane = token, 4 consecutive channels
    f32x4 aq = bq, ak = bk, av = f32x4{bv, bv, bv, bv};
    aq = mma16(wqf[0][0], xf0, aq); aq = mma16(wqf[0][1], xf1, aq);      // q[token = col][d = 4 q4 + r]
    ak = mma16(wqf[1][0], xf0, ak); ak = mma16(wqf[1][1], xf1, ak);      // k[token = col][d = 4 q4 + r]
    av = mma16(xf0, wqf[2][0], av); av = mma16(xf1, wqf[2][1], av);      // v[token = 4 q4 + r][d = col]
    const s16x4 qb = as_s16(cvt4c(aq)), ka = as_s16(cvt4c(ak)), vp = as_s16(cvt4c(av));
    f32x4 oacc = zero4;
#pragma unroll
    for (int h = 0; h < CH; ++h) {
      const s16x4 qm = (q4 == h) ? qb : zero_s;            // head h's dims only
      f32x4 s0 = mma16(ka, qm, zero4);                     // S^T[key = 4 q4 + r (tokens)][query = col]
      f32x4 s1 = mma16(bkA, qm, zero4);                    // bank keys
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s0[r] *= scale; s1[r] *= scale; mx = fmaxf(mx, fmaxf(s0[r], s1[r])); }
      mx = rows4_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s0[r] = __expf(s0[r] - mx); s1[r] = __expf(s1[r] - mx); sum += s0[r] + s1[r]; }
      sum = rows4_sum(sum);
      const float inv = 1.f / sum;
      if (adrop) {
        const uint32_t pkey = attn_drop_pkey(drop, (img * CG + g) * CH + h);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s0[r] *= inv * attn_drop_factor(drop, pkey, col, 4 * q4 + r);
          s1[r] *= inv * attn_drop_factor(drop, pkey, col, CT + 4 * q4 + r);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[r] *= inv; s1[r] *= inv; }
      }
      f32x4 oh = mma16(vp, as_s16(cvt4c(s0)), zero4);      // O_h^T[d = 4 q4 + r][query = col]: meaningful where d is in head h = lane group h
      oh = mma16(bvP, as_s16(cvt4c(s1)), oh);
      if (q4 == h) oacc = oh;
    }
    bad |= (oacc[0] != oacc[0]) | (oacc[1] != oacc[1]) | (oacc[2] != oacc[2]) | (oacc[3] != oacc[3]);
    const bf16x4 o4 = cvt4c(oacc);                         // O[query = col][16 g + 4 q4 ..]
    *reinterpret_cast<bf16x4*>(ot + col * LDOO + CCG * g + 4 * q4) = o4;
    if (osv && valid) *reinterpret_cast<bf16x4*>(osv + ((size_t)img * CT + col) * CO + CCG * g + 4 * q4) = o4;
  }
  __syncthreads();                                         // both waves of the image have written their groups' O quads

  // ---- out = dropout(O Wp^T + b): out^T[c][t] = Wp[c][:] . O[t][:]; this wave: output tiles 6 half .. + 6, 6 k-steps from the LDS fragments ----
  s16x4 of[CO / 16];
#pragma unroll
  for (int o16 = 0; o16 < CO / 16; ++o16) of[o16] = rowfrag(ot, LDOO, 0, 16 * o16);
  {
    f32x4 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = *reinterpret_cast<const f32x4*>(a.bproj + (6 * half + j) * 16 + 4 * q4);
#pragma unroll
    for (int o16 = 0; o16 < CO / 16; ++o16)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const s16x4 wf = as_s16(*reinterpret_cast<const bf16x4*>(swp + ((size_t)((6 * half + j) * (CO / 16) + o16) * 64 + lane) * 4));
        acc[j] = mma16(wf, of[o16], acc[j]);
      }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int c0 = (6 * half + j) * 16 + 4 * q4;
      if (pdrop) {
        const uint32_t base = (uint32_t)(img * CT + col) * (uint32_t)CC + (uint32_t)c0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] *= drop_factor(pkey_proj, base + r, pp, pinv);
      }
      *reinterpret_cast<bf16x4*>(xt + col * LDX + c0) = cvt4c(acc[j]);      // the token tile is dead: it collects the output rows
    }
  }
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {                        // whole rows out: 16-byte pieces
      const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
      *reinterpret_cast<bf16x8*>(og + ((size_t)img * CT + row) * a.ldo + 8 * c8) = *reinterpret_cast<const bf16x8*>(xt + row * LDX + 8 * c8);
    }
  }
  if (a.nan_flag && __any(bad && valid) && lane == 0) atomicOr(a.nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward, one launch: the proj input gradient, the recomputed attention cores' backward, the q/k/v projections' input gradient.
//   dO = (dout * mask) Wp ;  per (group, head): P recomputed, dP = dO V^T, D = rowsum(P m * dP), dS = P (dP m - D) / 2,
//   dQ = dS K, dK = dS^T Q, dV = (P m)^T dO ;  dx_g = dq Wq + dk Wk + dv Wv
// Contractions over queries (dK, dV) and over keys (dQ) need S-shaped matrices in both orientations: S and dP are formed twice from
// the same operand registers with A and B swapped, the second orientation's statistics arrive by lane permute (as in branch_bwd.hip).
// dq / dk / dv leave as rows of [B*16*6, 48] (operand of the deferred weight-gradient GEMMs), the masked dout as dz (operand of
// dW_proj with the O the forward saved); the bank rows' gradients as one row of 512 partial sums per workgroup.
constexpr int CGA_PART = 2 * CS * CCG;                     // [d sh_k 16 x 16 | d sh_v 16 x 16]
constexpr int SMB_WP = 0, SMB_GT = WP_FRAGS * 512, SMB_XT = SMB_GT + CNI * CT * LDX * 2, SMB_RED = SMB_XT + CNI * CT * LDX * 2,
              SM_CGA_BWD = SMB_RED + CNW * CGA_PART * 4;   // 36864 + 25600 + 25600 + 16384 = 104448 bytes

__global__ __launch_bounds__(64 * CNW) void cga_bwd_kernel(qavit_cga_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the per-wave tile bases and image offsets live in SGPRs (the kernel sat at 256 VGPRs + 20 B of scratch)
  bf16* swp = reinterpret_cast<bf16*>(smraw + SMB_WP);      // Wp^T fragments: (o tile, c tile): lane = o, 4 consecutive c
  const int wimg = wave >> 1, half = wave & 1;             // image wimg of the tile, channel groups 3 half .. 3 half + 2
  bf16* gt = reinterpret_cast<bf16*>(smraw + SMB_GT) + wimg * (CT * LDX);
  bf16* xt = reinterpret_cast<bf16*>(smraw + SMB_XT) + wimg * (CT * LDX);
  float* red = reinterpret_cast<float*>(smraw + SMB_RED);
  const int img_raw = blockIdx.x * CNI + wimg;
  const bool valid = img_raw < a.B;
  const int img = valid ? img_raw : a.B - 1;
  const bf16* xg = reinterpret_cast<const bf16*>(a.x);
  const bf16* gg = reinterpret_cast<const bf16*>(a.dout);
  const bf16* wq = reinterpret_cast<const bf16*>(a.wqkv_rm);
  const bf16* wqT = reinterpret_cast<const bf16*>(a.wqkvT_rm);
  const bf16* wpT = reinterpret_cast<const bf16*>(a.wprojT_rm);
  bf16* dzg = reinterpret_cast<bf16*>(a.dz);
  bf16* dqg = reinterpret_cast<bf16*>(a.dqkv);
  bf16* dxg = reinterpret_cast<bf16*>(a.dx);

  if (a.nan_trip && *a.nan_trip != 0) {
    // the forward tripped the NaN rule: it returned proj(0), nothing flows back through the attention cores (HQAViT_CIFAR100.py:356-357,
    // :394-395).  dz = masked dout (db_proj), dqkv = dx = 0, bank-row partials 0.
    const bool pd = a.proj_drop_p > 0.f && a.rng != nullptr;
    const uint32_t pk = pd ? rng_key(a.rng, a.proj_drop_site) : 0u;
    const float ppp = pd ? a.proj_drop_p : 0.f, piv = pd ? 1.f / (1.f - a.proj_drop_p) : 1.f;
    bf16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (bf16)0.f;
    if (valid) {
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
        if (pd && dzg) {
          bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gg + ((size_t)img * CT + row) * a.lddout + 8 * c8);
          const uint32_t base = (uint32_t)(img * CT + row) * (uint32_t)CC + (uint32_t)(8 * c8);
#pragma unroll
          for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pk, base + e, ppp, piv));
          *reinterpret_cast<bf16x8*>(dzg + ((size_t)img * CT + row) * a.lddz + 8 * c8) = g8;
        }
        *reinterpret_cast<bf16x8*>(dxg + ((size_t)img * CT + row) * a.lddx + 8 * c8) = z8;
      }
      // this wave's half of the image's dqkv rows: 8 tokens x 6 groups x 48 = 2304 contiguous elements
      bf16* drow = dqg + ((size_t)img * CT + 8 * half) * CG * (3 * CCG);
      for (int p = lane; p < 8 * CG * 3 * CCG / 8; p += 64) *reinterpret_cast<bf16x8*>(drow + 8 * p) = z8;
    }
    for (int e = tid; e < CGA_PART; e += 64 * CNW) a.parts[(size_t)blockIdx.x * CGA_PART + e] = 0.f;
    return;
  }
  bf16x8 xr[3], gr[3];
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
    xr[it] = *reinterpret_cast<const bf16x8*>(xg + ((size_t)img * CT + row) * a.ldx + 8 * c8);
    gr[it] = *reinterpret_cast<const bf16x8*>(gg + ((size_t)img * CT + row) * a.lddout + 8 * c8);
  }
  bf16x4 wpr[WP_FRAGS / CNW];
#pragma unroll
  for (int it = 0; it < WP_FRAGS / CNW; ++it) {
    const int f = it * CNW + wave, otile = f / (CC / 16), ctile = f - otile * (CC / 16);
    wpr[it] = *reinterpret_cast<const bf16x4*>(wpT + (size_t)(16 * otile + col) * CC + 16 * ctile + 4 * q4);
  }
  s16x4 wqf[3][2], wtf[3][2];                              // forward operand (lane = d, 4 channels) and its transpose (lane = channel, 4 dims)
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      wqf[p][ks] = as_s16(*reinterpret_cast<const bf16x4*>(wq + (size_t)(16 * p + col) * CPG + 16 * ks + 4 * q4));
      wtf[p][ks] = as_s16(*reinterpret_cast<const bf16x4*>(wqT + (size_t)(16 * ks + col) * (3 * CCG) + 16 * p + 4 * q4));
    }
  const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bqkv + 4 * q4);
  const f32x4 bk = *reinterpret_cast<const f32x4*>(a.bqkv + CCG + 4 * q4);
  const float bqp = a.bqkv[col], bkp = a.bqkv[CCG + col];
  const f32x4 shk4 = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)col * CCG + 4 * q4);      // lane = bank row, 4 dims
  const f32x4 shv4 = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)col * CCG + 4 * q4);
  float shkp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) shkp[i] = a.sh_k[(size_t)(4 * q4 + i) * CCG + col];              // lane = dim, 4 bank rows
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pp = pdrop ? a.proj_drop_p : 0.f, pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
    bf16x8 g8 = gr[it];
    if (pdrop) {
      const uint32_t base = (uint32_t)(img * CT + row) * (uint32_t)CC + (uint32_t)(8 * c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pkey_proj, base + e, pp, pinv));
      if (dzg && valid) *reinterpret_cast<bf16x8*>(dzg + ((size_t)img * CT + row) * a.lddz + 8 * c8) = g8;
    }
    *reinterpret_cast<bf16x8*>(gt + row * LDX + 8 * c8) = g8;
    *reinterpret_cast<bf16x8*>(xt + row * LDX + 8 * c8) = xr[it];
  }
#pragma unroll
  for (int it = 0; it < WP_FRAGS / CNW; ++it) *reinterpret_cast<bf16x4*>(swp + ((size_t)(it * CNW + wave) * 64 + lane) * 4) = wpr[it];
  s16x4 bkA, bvA, bkP;
  {
    bf16x4 t1, t2, t3;
#pragma unroll
    for (int i = 0; i < 4; ++i) { t1[i] = (bf16)shk4[i]; t2[i] = (bf16)shv4[i]; t3[i] = (bf16)shkp[i]; }
    bkA = as_s16(t1); bvA = as_s16(t2); bkP = as_s16(t3);
  }
  const bool adrop = a.attn_drop_p > 0.f && a.rng != nullptr;
  AttnDrop drop;
  drop.on = adrop;
  drop.p = adrop ? a.attn_drop_p : 0.f;
  drop.inv_keep = adrop ? 1.f / (1.f - a.attn_drop_p) : 1.f;
  drop.key = adrop ? rng_key(a.rng, a.attn_drop_site) : 0u;
  const float scale = 0.5f;
  const s16x4 zero_s = {0, 0, 0, 0};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  s16x4 idq;                                               // the 16 x 16 identity as a B operand: element (k = 4 q4 + j, column col)
  {
    bf16x4 t1;
#pragma unroll
    for (int j = 0; j < 4; ++j) t1[j] = (bf16)((4 * q4 + j == col) ? 1.f : 0.f);
    idq = as_s16(t1);
  }
  __syncthreads();

  // ---- dO = gm Wp in both layouts: dob[g] lane = query, 4 dims (B operand);  dop[g] lane = dim, 4 queries ----
  s16x4 dob[CG / 2], dop[CG / 2];                          // this wave's three groups
  {
    s16x4 gf[CC / 16];
#pragma unroll
    for (int ct = 0; ct < CC / 16; ++ct) gf[ct] = rowfrag(gt, LDX, 0, 16 * ct);
#pragma unroll
    for (int g = 0; g < CG / 2; ++g) {
      f32x4 c1 = zero4, c2 = zero4;
#pragma unroll
      for (int ct = 0; ct < CC / 16; ++ct) {
        const s16x4 wf = as_s16(*reinterpret_cast<const bf16x4*>(swp + ((size_t)((3 * half + g) * (CC / 16) + ct) * 64 + lane) * 4));
        c1 = mma16(wf, gf[ct], c1);                        // dO^T[o = 4 q4 + r][t = col]
        c2 = mma16(gf[ct], wf, c2);                        // dO  [t = 4 q4 + r][o = col]
      }
      dob[g] = as_s16(cvt4c(c1));
      dop[g] = as_s16(cvt4c(c2));
    }
  }
  f32x4 dshk = zero4, dshv = zero4;                        // acc[r] = d sh^T[d = 4 q4 + r][s = col], summed over groups

  if (valid)
#pragma unroll
  for (int gl = 0; gl < CG / 2; ++gl) {
    const int g = 3 * half + gl;
    const s16x4 xf0 = rowfrag(xt, LDX, 0, CPG * g), xf1 = rowfrag(xt, LDX, 0, CPG * g + 16);
    f32x4 aq = bq, ak = bk, av = zero4, aqp = f32x4{bqp, bqp, bqp, bqp}, akp = f32x4{bkp, bkp, bkp, bkp};
    {
      const f32x4 bvv = *reinterpret_cast<const f32x4*>(a.bqkv + 2 * CCG + 4 * q4);
      av = bvv;
    }
    aq = mma16(wqf[0][0], xf0, aq); aq = mma16(wqf[0][1], xf1, aq);          // q[token = col][4 dims]
    ak = mma16(wqf[1][0], xf0, ak); ak = mma16(wqf[1][1], xf1, ak);          // k[token = col][4 dims]
    av = mma16(wqf[2][0], xf0, av); av = mma16(wqf[2][1], xf1, av);          // v[token = col][4 dims]
    aqp = mma16(xf0, wqf[0][0], aqp); aqp = mma16(xf1, wqf[0][1], aqp);      // q[token = 4 q4 + r][dim = col]
    akp = mma16(xf0, wqf[1][0], akp); akp = mma16(xf1, wqf[1][1], akp);      // k[token = 4 q4 + r][dim = col]
    const s16x4 qb = as_s16(cvt4c(aq)), ka = as_s16(cvt4c(ak)), va = as_s16(cvt4c(av)), qp = as_s16(cvt4c(aqp)), kp = as_s16(cvt4c(akp));
    f32x4 dq = zero4, dk = zero4, dv = zero4;
#pragma unroll
    for (int h = 0; h < CH; ++h) {
      const bool mine = q4 == h;
      const s16x4 qm = mine ? qb : zero_s, dom = mine ? dob[gl] : zero_s;
      f32x4 sT0 = mma16(ka, qm, zero4), sT1 = mma16(bkA, qm, zero4);        // S^T[key][query = col]
      const f32x4 dT0 = mma16(va, dom, zero4), dT1 = mma16(bvA, dom, zero4);  // dP^T
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) { sT0[r] *= scale; sT1[r] *= scale; mx = fmaxf(mx, fmaxf(sT0[r], sT1[r])); }
      mx = rows4_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { sT0[r] = __expf(sT0[r] - mx); sT1[r] = __expf(sT1[r] - mx); sum += sT0[r] + sT1[r]; }
      sum = rows4_sum(sum);
      const float inv = 1.f / sum;
      const uint32_t pkey = adrop ? attn_drop_pkey(drop, (img * CG + g) * CH + h) : 0u;
      float m0[4], m1[4], dsum = 0.f;
      f32x4 pm0, pm1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        m0[r] = adrop ? attn_drop_factor(drop, pkey, col, 4 * q4 + r) : 1.f;
        m1[r] = adrop ? attn_drop_factor(drop, pkey, col, CT + 4 * q4 + r) : 1.f;
        sT0[r] *= inv; sT1[r] *= inv;                      // P^T
        pm0[r] = sT0[r] * m0[r]; pm1[r] = sT1[r] * m1[r];  // (P m)^T
        dsum += pm0[r] * dT0[r] + pm1[r] * dT1[r];
      }
      dsum = rows4_sum(dsum);                    // D[query = col] = sum_keys (P m) dP
      f32x4 e0, e1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e0[r] = sT0[r] * (dT0[r] * m0[r] - dsum) * scale;  // dS^T
        e1[r] = sT1[r] * (dT1[r] * m1[r] - dsum) * scale;
      }
      const s16x4 dsT0 = as_s16(cvt4c(e0)), dsT1 = as_s16(cvt4c(e1));
      // the other orientation (lane = key, registers = queries): a quad tile in accumulator layout read as an A operand is its own
      // transpose, so one MFMA against the identity gives dS / P m exactly -- no second S / dP, exponentials or dropout hashes
      const s16x4 ds20 = as_s16(cvt4c(mma16(dsT0, idq, zero4))), ds21 = as_s16(cvt4c(mma16(dsT1, idq, zero4)));
      const s16x4 pd20 = as_s16(cvt4c(mma16(as_s16(cvt4c(pm0)), idq, zero4))), pd21 = as_s16(cvt4c(mma16(as_s16(cvt4c(pm1)), idq, zero4)));
      // dQ^T[d][query] = sum_key K[key][d] dS^T[key][query]  (tokens + bank rows); valid where d is in head h = this lane group
      f32x4 t = mma16(kp, dsT0, zero4);
      t = mma16(bkP, dsT1, t);
      if (mine) dq = t;
      // dK^T[d][key] = sum_query Q[query][d] dS[query][key];  dV^T[d][key] = sum_query dO[query][d] (P m)[query][key]
      t = mma16(qp, ds20, zero4);
      if (mine) dk = t;
      t = mma16(dop[gl], pd20, zero4);
      if (mine) dv = t;
      t = mma16(qp, ds21, zero4);                           // bank rows: d sh_k^T[d][s]
      if (mine) { dshk[0] += t[0]; dshk[1] += t[1]; dshk[2] += t[2]; dshk[3] += t[3]; }
      t = mma16(dop[gl], pd21, zero4);
      if (mine) { dshv[0] += t[0]; dshv[1] += t[1]; dshv[2] += t[2]; dshv[3] += t[3]; }
    }
    // dq / dk / dv: value[token = col][d = 4 q4 + r] -> the (image, token, group) row of dqkv
    const bf16x4 dq4 = cvt4c(dq), dk4 = cvt4c(dk), dv4 = cvt4c(dv);
    bf16* drow = dqg + (((size_t)img * CT + col) * CG + g) * (3 * CCG);
    *reinterpret_cast<bf16x4*>(drow + 4 * q4) = dq4;
    *reinterpret_cast<bf16x4*>(drow + CCG + 4 * q4) = dk4;
    *reinterpret_cast<bf16x4*>(drow + 2 * CCG + 4 * q4) = dv4;
    // dx_g^T[c][token] = sum_d Wq[d][c] dq[token][d] + Wk .. + Wv ..
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 c = mma16(wtf[0][ks], as_s16(dq4), zero4);
      c = mma16(wtf[1][ks], as_s16(dk4), c);
      c = mma16(wtf[2][ks], as_s16(dv4), c);
      *reinterpret_cast<bf16x4*>(xt + col * LDX + CPG * g + 16 * ks + 4 * q4) = cvt4c(c);      // in place: group g's columns of the tile (only this wave reads them) become dx
    }
  }
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {                        // whole dx rows out: 16-byte pieces
      const int p = lane + 64 * it, row = 8 * half + p / 24, c8 = p % 24;
      *reinterpret_cast<bf16x8*>(dxg + ((size_t)img * CT + row) * a.lddx + 8 * c8) = *reinterpret_cast<const bf16x8*>(xt + row * LDX + 8 * c8);
    }
  }
  // ---- bank-row gradients: the four waves' sums -> one row of partials ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave * CGA_PART + col * CCG + 4 * q4 + r] = dshk[r];                 // [s][d]
    red[wave * CGA_PART + CS * CCG + col * CCG + 4 * q4 + r] = dshv[r];
  }
  __syncthreads();
  for (int e = tid; e < CGA_PART; e += 64 * CNW) {
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < CNW; ++w) sacc += red[w * CGA_PART + e];
    a.parts[(size_t)blockIdx.x * CGA_PART + e] = sacc;
  }
}

int cga_validate(const qavit_cga_args* a) {
  if (!a) return set_error(QAVIT_EINVAL, "cga: null args");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "cga: the fused channel-group kernel is bf16 only");
  if ((a->T != CT && a->T != 64) || a->C != CC || a->G != CG || a->H != CH || a->D != CD || a->S != CS)
    return set_error(QAVIT_EINVAL, "cga: built for 16 or 64 tokens x 192 channels, 6 groups, 4 heads of 4, 16 bank rows");
  if (a->B <= 0 || !a->x || !a->out || !a->wqkv_rm || !a->wproj_rm || !a->bqkv || !a->bproj || !a->sh_k || !a->sh_v) return set_error(QAVIT_EINVAL, "cga: null operand");
  if (a->nan_trip && !a->nan_flag) return set_error(QAVIT_EINVAL, "cga: nan_trip is written by the NaN-rule launch, which needs nan_flag");
  auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
  if (!al(a->x, 15) || !al(a->out, 15) || !al(a->wqkv_rm, 7) || !al(a->wproj_rm, 7) || !al(a->bqkv, 15) || !al(a->bproj, 15) || !al(a->sh_k, 15) || !al(a->sh_v, 3) ||
      (a->o_save && !al(a->o_save, 7)) || a->ldx % 8 || a->ldo % 8)
    return set_error(QAVIT_EINVAL, "cga: alignment (x 16 bytes / ld % 8; out, weights, o_save 8 bytes / ld % 4; biases and sh_k 16 bytes)");
  return QAVIT_OK;
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_cga_bwd_parts(int B, int T) { return B > 0 ? (T == 64 ? B : (B + CNI - 1) / CNI) : 0; }

extern "C" int qavit_cga_bwd(const qavit_cga_bwd_args* a, void* stream) {
  if (!a) return set_error(QAVIT_EINVAL, "cga_bwd: null args");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "cga_bwd: bf16 only");
  if ((a->T != CT && a->T != 64) || a->C != CC || a->G != CG || a->H != CH || a->D != CD || a->S != CS)
    return set_error(QAVIT_EINVAL, "cga_bwd: built for 16 or 64 tokens x 192 channels, 6 groups, 4 heads of 4, 16 bank rows");
  if (a->B <= 0 || !a->x || !a->dout || !a->wqkv_rm || !a->wqkvT_rm || !a->wprojT_rm || !a->bqkv || !a->sh_k || !a->sh_v || !a->dqkv || !a->dx || !a->parts)
    return set_error(QAVIT_EINVAL, "cga_bwd: null operand");
  if (a->proj_drop_p > 0.f && a->rng && !a->dz) return set_error(QAVIT_EINVAL, "cga_bwd: proj dropout needs dz");
  auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
  if (!al(a->x, 15) || !al(a->dout, 15) || (a->dz && !al(a->dz, 15)) || !al(a->wqkv_rm, 7) || !al(a->wqkvT_rm, 7) || !al(a->wprojT_rm, 7) || !al(a->bqkv, 15) ||
      !al(a->sh_k, 15) || !al(a->sh_v, 15) || !al(a->dqkv, 7) || !al(a->dx, 15) || !al(a->parts, 15) || a->ldx % 8 || a->lddout % 8 || (a->dz && a->lddz % 8) || a->lddx % 8)
    return set_error(QAVIT_EINVAL, "cga_bwd: alignment (activations 16 bytes / ld % 8; weights, dqkv, dx 8 bytes; biases, bank rows, parts 16 bytes)");
  static_assert(CGA_PART == QAVIT_CGA_PARTS_FLOATS, "header constant out of date");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->T == 64) return cga64_bwd_launch(a, st);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cga_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_CGA_BWD);
    attr_done = true;
  }
  hipLaunchKernelGGL(cga_bwd_kernel, dim3((a->B + CNI - 1) / CNI), dim3(64 * CNW), SM_CGA_BWD, st, *a);
  return check_launch("cga_bwd");
}

extern "C" int qavit_cga_supported(int T, int C, int G, int H, int S) { return ((T == CT || T == 64) && C == CC && G == CG && H == CH && S == CS) ? 1 : 0; }

extern "C" int qavit_cga_fwd(const qavit_cga_args* a, void* stream) {
  int rc = cga_validate(a);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->T == 64) return cga64_fwd_launch(a, st);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cga_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SM_CGA);
    attr_done = true;
  }
  hipLaunchKernelGGL(cga_fwd_kernel, dim3((a->B + CNI - 1) / CNI), dim3(64 * CNW), SM_CGA, st, *a);
  if (a->nan_flag && !a->nan_defer)
    branch_nan_fix_launch(a->out, a->ldo, a->B * CT, CC, a->bproj, a->proj_drop_p, a->proj_drop_site, a->rng, a->nan_flag, a->nan_trip, a->o_save, CO, CO, st);
  return check_launch("cga_fwd");
}
