// Fused CHANNEL-GROUP attention branch on 64-token problems (EfficientChannelGroupAttention, HQAViT_IN_Tiny.py:889-925 / QAViT.py; the
// 64 learned tokens of the Tiny-ImageNet configuration, QA-ViT at 32 px): forward and backward, one launch each, same contract as the
// 16-token kernels of cga.hip (qavit_cga_args / qavit_cga_bwd_args with T = 64).
//
// Per image and channel group g (6 groups of 32 channels): q / k / v = Linear(32 -> 16) of the group's slice of the 64 tokens, 4 heads of
// D = 4 over [64 token keys ; 16 projected bank rows] = 80 keys, then proj(96 -> 192) over the concatenated groups.  A (group, head)
// problem is 64 x 80 x 4: still far too small to tile across waves, and the kernel is VALU-bound (softmax exponentials, dropout
// hashes), so the decomposition only has to keep every SIMD busy:
//   forward : one image per workgroup, 12 waves = (group, query-tile pair).  A wave projects k / v of all four token tiles of its group
//             (16 small MFMAs, cheaper than sharing them through LDS) and runs its two 16-query tiles against the five key tiles.
//   backward: one image per workgroup, 6 waves = group.  dK / dV are sums over all 64 queries: one wave walks the four query tiles and
//             keeps them in accumulators (no cross-wave reduction); dQ of a query tile is complete when its tile ends.
// Heads by lane-group masking as in cga.hip: with acc[r] = C[4 q4 + r][col], lane group q4 holds exactly head q4's four dims, so zeroing
// the other lane groups of one operand contracts a 16-deep MFMA over head h only.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "attn_shared.h"
#include "frag16.cuh"

namespace qv {

void branch_nan_fix_launch(void* out, int64_t ldo, int rows, int C, const float* bias, float p, int site, const int64_t* rng, int* flag,
                           int* trip, void* o_save, int64_t ldos, int Co, hipStream_t st);

namespace {

constexpr int WT = 64, WC = 192, WG = 6, WPG = 32, WCG = 16, WH = 4, WS = 16, WO = WG * WCG;     // tokens, channels, groups, ch/group, q dims/group, heads, bank rows, 96
constexpr int QT = WT / 16;                                // query / token tiles
constexpr int NKT = QT + 1;                                // key tiles: tokens + bank rows
constexpr int FW_WAVES = 12, BW_WAVES = 6;
constexpr int LDX6 = WC + 8, LDO6 = WO + 8;
constexpr int WPF = (WC / 16) * (WO / 16);                 // 72 proj-weight fragments of 512 B
constexpr int SMF_WP = 0, SMF_XT = WPF * 512, SMF_OT = SMF_XT + WT * LDX6 * 2, SMF_TOTAL = SMF_OT + WT * LDO6 * 2;      // 36864 + 25600 + 13312 = 75776
constexpr int PARTS = 2 * WS * WCG;                        // [d sh_k 16 x 16 | d sh_v 16 x 16]
constexpr int SMB_WP = 0, SMB_GT = WPF * 512, SMB_XT = SMB_GT + WT * LDX6 * 2, SMB_RED = SMB_XT + WT * LDX6 * 2,
              SMB_TOTAL = SMB_RED + BW_WAVES * PARTS * 4;  // 36864 + 25600 + 25600 + 12288 = 100352

__device__ __forceinline__ bf16x4 cv4(const f32x4& acc) {
  bf16x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (bf16)acc[r];
  return v;
}
__device__ __forceinline__ s16x4 cv4s(const f32x4& acc) { return as_s16(cv4(acc)); }

__global__ __launch_bounds__(64 * FW_WAVES) void cga64_fwd_kernel(qavit_cga_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16* swp = reinterpret_cast<bf16*>(smraw + SMF_WP);
  bf16* xt = reinterpret_cast<bf16*>(smraw + SMF_XT);
  bf16* ot = reinterpret_cast<bf16*>(smraw + SMF_OT);
  const int img = blockIdx.x;
  const bf16* xg = reinterpret_cast<const bf16*>(a.x) + (size_t)img * WT * a.ldx;
  const bf16* wq = reinterpret_cast<const bf16*>(a.wqkv_rm);
  const bf16* wp = reinterpret_cast<const bf16*>(a.wproj_rm);
  bf16* og = reinterpret_cast<bf16*>(a.out) + (size_t)img * WT * a.ldo;
  bf16* osv = reinterpret_cast<bf16*>(a.o_save);

  // ---- loads: the image's token tile, the proj weight as fragments, q/k/v weights, biases, bank rows ----
  bf16x8 xr[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int p = tid + 64 * FW_WAVES * it, row = p / 24, c8 = p - row * 24;
    xr[it] = *reinterpret_cast<const bf16x8*>(xg + (size_t)row * a.ldx + 8 * c8);
  }
  bf16x4 wpr[WPF / FW_WAVES];                              // 6 quads per thread
#pragma unroll
  for (int it = 0; it < WPF / FW_WAVES; ++it) {
    const int f = it * FW_WAVES + wave, ctile = f / (WO / 16), otile = f - ctile * (WO / 16);
    wpr[it] = *reinterpret_cast<const bf16x4*>(wp + (size_t)(16 * ctile + col) * WO + 16 * otile + 4 * q4);
  }
  s16x4 wqf[3][2];                                         // [part][k-step]: lane = output dim d (col), 4 consecutive input channels
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wqf[p][ks] = as_s16(*reinterpret_cast<const bf16x4*>(wq + (size_t)(16 * p + col) * WPG + 16 * ks + 4 * q4));
  const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bqkv + 4 * q4);
  const f32x4 bk = *reinterpret_cast<const f32x4*>(a.bqkv + WCG + 4 * q4);
  const float bv = a.bqkv[2 * WCG + col];
  const f32x4 shk4 = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)col * WCG + 4 * q4);      // bank key s = col, dims 4 q4 ..
  float shv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) shv[i] = a.sh_v[(size_t)(4 * q4 + i) * WCG + col];               // bank value rows 4 q4 + i, dim col
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int p = tid + 64 * FW_WAVES * it, row = p / 24, c8 = p - row * 24;
    *reinterpret_cast<bf16x8*>(xt + row * LDX6 + 8 * c8) = xr[it];
  }
#pragma unroll
  for (int it = 0; it < WPF / FW_WAVES; ++it) *reinterpret_cast<bf16x4*>(swp + ((size_t)(it * FW_WAVES + wave) * 64 + lane) * 4) = wpr[it];
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
  __syncthreads();                                         // the proj fragments and the token tile are complete

  {
    const int g = wave >> 1, qp = wave & 1;                // this wave: channel group g, query tiles 2 qp, 2 qp + 1
    s16x4 ka[QT], vp[QT];                                  // k[token = col][d = 4 q4 + r] / v[token = 4 q4 + r][d = col] of the four token tiles
#pragma unroll
    for (int kt = 0; kt < QT; ++kt) {
      const s16x4 xf0 = rowfrag(xt, LDX6, 16 * kt, WPG * g), xf1 = rowfrag(xt, LDX6, 16 * kt, WPG * g + 16);
      f32x4 ak = bk, av = f32x4{bv, bv, bv, bv};
      ak = mma16(wqf[1][0], xf0, ak); ak = mma16(wqf[1][1], xf1, ak);
      av = mma16(xf0, wqf[2][0], av); av = mma16(xf1, wqf[2][1], av);
      ka[kt] = cv4s(ak); vp[kt] = cv4s(av);
    }
#pragma unroll 1
    for (int qi = 0; qi < 2; ++qi) {
      const int qt = 2 * qp + qi;
      const s16x4 xf0 = rowfrag(xt, LDX6, 16 * qt, WPG * g), xf1 = rowfrag(xt, LDX6, 16 * qt, WPG * g + 16);
      f32x4 aq = bq;
      aq = mma16(wqf[0][0], xf0, aq); aq = mma16(wqf[0][1], xf1, aq);      // q[token = col][d = 4 q4 + r]
      const s16x4 qb = cv4s(aq);
      f32x4 oacc = zero4;
#pragma unroll
      for (int h = 0; h < WH; ++h) {
        const s16x4 qm = (q4 == h) ? qb : zero_s;          // head h's dims only
        f32x4 s[NKT];                                      // S^T[key = 16 kt + 4 q4 + r][query = col]
#pragma unroll
        for (int kt = 0; kt < QT; ++kt) s[kt] = mma16(ka[kt], qm, zero4);
        s[QT] = mma16(bkA, qm, zero4);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kt][r] *= scale; mx = fmaxf(mx, s[kt][r]); }
        mx = rows4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
        sum = rows4_sum(sum);
        const float inv = 1.f / sum;
        if (adrop) {
          const uint32_t pkey = attn_drop_pkey(drop, (img * WG + g) * WH + h);
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] *= inv * attn_drop_factor(drop, pkey, 16 * qt + col, 16 * kt + 4 * q4 + r);
        } else {
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
        }
        f32x4 oh = mma16(bvP, cv4s(s[QT]), zero4);         // O_h^T[d = 4 q4 + r][query = col]: meaningful where d is in head h = lane group h
#pragma unroll
        for (int kt = 0; kt < QT; ++kt) oh = mma16(vp[kt], cv4s(s[kt]), oh);
        if (q4 == h) oacc = oh;
      }
      bad |= (oacc[0] != oacc[0]) | (oacc[1] != oacc[1]) | (oacc[2] != oacc[2]) | (oacc[3] != oacc[3]);
      const bf16x4 o4 = cv4(oacc);                         // O[query = 16 qt + col][16 g + 4 q4 ..]
      *reinterpret_cast<bf16x4*>(ot + (16 * qt + col) * LDO6 + WCG * g + 4 * q4) = o4;
      if (osv) *reinterpret_cast<bf16x4*>(osv + ((size_t)img * WT + 16 * qt + col) * WO + WCG * g + 4 * q4) = o4;
    }
  }
  __syncthreads();                                         // every group's O quads are in the tile; the token tile is dead

  // ---- out = dropout(O Wp^T + b): out^T[c][t] = Wp[c][:] . O[t][:]; this wave: token tile wave & 3, output tiles 4 (wave >> 2) .. + 4 ----
  {
    const int tt = wave & 3, c0t = 4 * (wave >> 2);
    s16x4 of[WO / 16];
#pragma unroll
    for (int o16 = 0; o16 < WO / 16; ++o16) of[o16] = rowfrag(ot, LDO6, 16 * tt, 16 * o16);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = *reinterpret_cast<const f32x4*>(a.bproj + (c0t + j) * 16 + 4 * q4);
#pragma unroll
    for (int o16 = 0; o16 < WO / 16; ++o16)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const s16x4 wf = as_s16(*reinterpret_cast<const bf16x4*>(swp + ((size_t)((c0t + j) * (WO / 16) + o16) * 64 + lane) * 4));
        acc[j] = mma16(wf, of[o16], acc[j]);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c0 = (c0t + j) * 16 + 4 * q4;
      if (pdrop) {
        const uint32_t base = (uint32_t)(img * WT + 16 * tt + col) * (uint32_t)WC + (uint32_t)c0;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] *= drop_factor(pkey_proj, base + r, pp, pinv);
      }
      *reinterpret_cast<bf16x4*>(xt + (16 * tt + col) * LDX6 + c0) = cv4(acc[j]);      // the dead token tile collects the output rows
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {                          // whole rows out: 16-byte pieces
    const int p = tid + 64 * FW_WAVES * it, row = p / 24, c8 = p - row * 24;
    *reinterpret_cast<bf16x8*>(og + (size_t)row * a.ldo + 8 * c8) = *reinterpret_cast<const bf16x8*>(xt + row * LDX6 + 8 * c8);
  }
  if (a.nan_flag && __any(bad) && lane == 0) atomicOr(a.nan_flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward (see cga.hip for the algebra): wave = channel group; loop over the four query tiles.
//   per (query tile, head): S^T, dP^T over the five key tiles -> softmax statistics, D, dS^T -> dQ; then key tile by key tile the second
//   orientation (S, dP -> P m, dS) -> dK, dV accumulators (token tiles) and the bank rows' gradients.
__global__ __launch_bounds__(64 * BW_WAVES) void cga64_bwd_kernel(qavit_cga_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16* swp = reinterpret_cast<bf16*>(smraw + SMB_WP);      // Wp^T fragments: (o tile, c tile): lane = o, 4 consecutive c
  bf16* gt = reinterpret_cast<bf16*>(smraw + SMB_GT);
  bf16* xt = reinterpret_cast<bf16*>(smraw + SMB_XT);
  float* red = reinterpret_cast<float*>(smraw + SMB_RED);
  const int img = blockIdx.x;
  const bf16* xg = reinterpret_cast<const bf16*>(a.x) + (size_t)img * WT * a.ldx;
  const bf16* gg = reinterpret_cast<const bf16*>(a.dout) + (size_t)img * WT * a.lddout;
  const bf16* wq = reinterpret_cast<const bf16*>(a.wqkv_rm);
  const bf16* wqT = reinterpret_cast<const bf16*>(a.wqkvT_rm);
  const bf16* wpT = reinterpret_cast<const bf16*>(a.wprojT_rm);
  bf16* dzg = a.dz ? reinterpret_cast<bf16*>(a.dz) + (size_t)img * WT * a.lddz : nullptr;
  bf16* dqg = reinterpret_cast<bf16*>(a.dqkv) + (size_t)img * WT * WG * (3 * WCG);
  bf16* dxg = reinterpret_cast<bf16*>(a.dx) + (size_t)img * WT * a.lddx;
  const bool pdrop = a.proj_drop_p > 0.f && a.rng != nullptr;
  const uint32_t pkey_proj = pdrop ? rng_key(a.rng, a.proj_drop_site) : 0u;
  const float pp = pdrop ? a.proj_drop_p : 0.f, pinv = pdrop ? 1.f / (1.f - a.proj_drop_p) : 1.f;
  constexpr int NTH = 64 * BW_WAVES;

  if (a.nan_trip && *a.nan_trip != 0) {
    // the forward tripped the NaN rule: it returned proj(0); dz = masked dout (db_proj), dqkv = dx = 0, bank-row partials 0
    bf16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (bf16)0.f;
    for (int p = tid; p < WT * 24; p += NTH) {
      const int row = p / 24, c8 = p - row * 24;
      if (pdrop && dzg) {
        bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gg + (size_t)row * a.lddout + 8 * c8);
        const uint32_t base = (uint32_t)(img * WT + row) * (uint32_t)WC + (uint32_t)(8 * c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pkey_proj, base + e, pp, pinv));
        *reinterpret_cast<bf16x8*>(dzg + (size_t)row * a.lddz + 8 * c8) = g8;
      }
      *reinterpret_cast<bf16x8*>(dxg + (size_t)row * a.lddx + 8 * c8) = z8;
    }
    for (int p = tid; p < WT * WG * 3 * WCG / 8; p += NTH) *reinterpret_cast<bf16x8*>(dqg + 8 * p) = z8;
    for (int e = tid; e < PARTS; e += NTH) a.parts[(size_t)blockIdx.x * PARTS + e] = 0.f;
    return;
  }

  bf16x8 xr[4], gr[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int p = tid + NTH * it, row = p / 24, c8 = p - row * 24;
    xr[it] = *reinterpret_cast<const bf16x8*>(xg + (size_t)row * a.ldx + 8 * c8);
    gr[it] = *reinterpret_cast<const bf16x8*>(gg + (size_t)row * a.lddout + 8 * c8);
  }
  bf16x4 wpr[WPF / BW_WAVES];                              // 12 quads per thread
#pragma unroll
  for (int it = 0; it < WPF / BW_WAVES; ++it) {
    const int f = it * BW_WAVES + wave, otile = f / (WC / 16), ctile = f - otile * (WC / 16);
    wpr[it] = *reinterpret_cast<const bf16x4*>(wpT + (size_t)(16 * otile + col) * WC + 16 * ctile + 4 * q4);
  }
  s16x4 wqf[3][2], wtf[3][2];                              // forward operand (lane = d, 4 channels) and its transpose (lane = channel, 4 dims)
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      wqf[p][ks] = as_s16(*reinterpret_cast<const bf16x4*>(wq + (size_t)(16 * p + col) * WPG + 16 * ks + 4 * q4));
      wtf[p][ks] = as_s16(*reinterpret_cast<const bf16x4*>(wqT + (size_t)(16 * ks + col) * (3 * WCG) + 16 * p + 4 * q4));
    }
  const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bqkv + 4 * q4);
  const f32x4 bk = *reinterpret_cast<const f32x4*>(a.bqkv + WCG + 4 * q4);
  const f32x4 bvv = *reinterpret_cast<const f32x4*>(a.bqkv + 2 * WCG + 4 * q4);
  const float bqp = a.bqkv[col], bkp = a.bqkv[WCG + col];
  const f32x4 shk4 = *reinterpret_cast<const f32x4*>(a.sh_k + (size_t)col * WCG + 4 * q4);      // lane = bank row, 4 dims
  const f32x4 shv4 = *reinterpret_cast<const f32x4*>(a.sh_v + (size_t)col * WCG + 4 * q4);
  float shkp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) shkp[i] = a.sh_k[(size_t)(4 * q4 + i) * WCG + col];              // lane = dim, 4 bank rows
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int p = tid + NTH * it, row = p / 24, c8 = p - row * 24;
    bf16x8 g8 = gr[it];
    if (pdrop) {
      const uint32_t base = (uint32_t)(img * WT + row) * (uint32_t)WC + (uint32_t)(8 * c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = (bf16)((float)g8[e] * drop_factor(pkey_proj, base + e, pp, pinv));
      if (dzg) *reinterpret_cast<bf16x8*>(dzg + (size_t)row * a.lddz + 8 * c8) = g8;
    }
    *reinterpret_cast<bf16x8*>(gt + row * LDX6 + 8 * c8) = g8;
    *reinterpret_cast<bf16x8*>(xt + row * LDX6 + 8 * c8) = xr[it];
  }
#pragma unroll
  for (int it = 0; it < WPF / BW_WAVES; ++it) *reinterpret_cast<bf16x4*>(swp + ((size_t)(it * BW_WAVES + wave) * 64 + lane) * 4) = wpr[it];
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

  const int g = wave;
  // k, v of the four token tiles in the operand layouts the loop needs
  s16x4 ka[QT], va[QT], kp[QT];                            // k[token = col][4 dims], v[token = col][4 dims], k[token = 4 q4 + r][dim = col]
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) {
    const s16x4 xf0 = rowfrag(xt, LDX6, 16 * kt, WPG * g), xf1 = rowfrag(xt, LDX6, 16 * kt, WPG * g + 16);
    f32x4 ak = bk, av = bvv, akp = f32x4{bkp, bkp, bkp, bkp};
    ak = mma16(wqf[1][0], xf0, ak); ak = mma16(wqf[1][1], xf1, ak);
    av = mma16(wqf[2][0], xf0, av); av = mma16(wqf[2][1], xf1, av);
    akp = mma16(xf0, wqf[1][0], akp); akp = mma16(xf1, wqf[1][1], akp);
    ka[kt] = cv4s(ak); va[kt] = cv4s(av); kp[kt] = cv4s(akp);
  }
  f32x4 dk[QT], dv[QT], dshk = zero4, dshv = zero4;        // dK^T / dV^T[d = 4 q4 + r][key token = col] per token tile; d sh^T[d][s = col]
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) { dk[kt] = zero4; dv[kt] = zero4; }

#pragma unroll 1
  for (int qt = 0; qt < QT; ++qt) {
    // q of this query tile in both layouts, dO = gm Wp of this (group, query tile) in both layouts
    s16x4 qb, qp, dob, dop;
    {
      const s16x4 xf0 = rowfrag(xt, LDX6, 16 * qt, WPG * g), xf1 = rowfrag(xt, LDX6, 16 * qt, WPG * g + 16);
      f32x4 aq = bq, aqp = f32x4{bqp, bqp, bqp, bqp};
      aq = mma16(wqf[0][0], xf0, aq); aq = mma16(wqf[0][1], xf1, aq);          // q[token = col][4 dims]
      aqp = mma16(xf0, wqf[0][0], aqp); aqp = mma16(xf1, wqf[0][1], aqp);      // q[token = 4 q4 + r][dim = col]
      qb = cv4s(aq); qp = cv4s(aqp);
      f32x4 c1 = zero4, c2 = zero4;
#pragma unroll
      for (int ct = 0; ct < WC / 16; ++ct) {
        const s16x4 gf = rowfrag(gt, LDX6, 16 * qt, 16 * ct);
        const s16x4 wf = as_s16(*reinterpret_cast<const bf16x4*>(swp + ((size_t)(g * (WC / 16) + ct) * 64 + lane) * 4));
        c1 = mma16(wf, gf, c1);                            // dO^T[o = 4 q4 + r][t = col]
        c2 = mma16(gf, wf, c2);                            // dO  [t = 4 q4 + r][o = col]
      }
      dob = cv4s(c1); dop = cv4s(c2);
    }
    f32x4 dq = zero4;
#pragma unroll 1
    for (int h = 0; h < WH; ++h) {
      const bool mine = q4 == h;
      const s16x4 qm = mine ? qb : zero_s, dom = mine ? dob : zero_s;
      f32x4 sT[NKT], dT[NKT];                              // S^T, dP^T [key = 16 kt + 4 q4 + r][query = col]
#pragma unroll
      for (int kt = 0; kt < QT; ++kt) { sT[kt] = mma16(ka[kt], qm, zero4); dT[kt] = mma16(va[kt], dom, zero4); }
      sT[QT] = mma16(bkA, qm, zero4); dT[QT] = mma16(bvA, dom, zero4);
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sT[kt][r] *= scale; mx = fmaxf(mx, sT[kt][r]); }
      mx = rows4_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sT[kt][r] = __expf(sT[kt][r] - mx); sum += sT[kt][r]; }
      sum = rows4_sum(sum);
      const float inv = 1.f / sum;
      const uint32_t pkey = adrop ? attn_drop_pkey(drop, (img * WG + g) * WH + h) : 0u;
      float dsum = 0.f;
      s16x4 pmq[NKT];                                      // (P m)^T quads, bf16
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x4 pm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float m = adrop ? attn_drop_factor(drop, pkey, 16 * qt + col, 16 * kt + 4 * q4 + r) : 1.f;
          sT[kt][r] *= inv;                                // P^T
          pm[r] = sT[kt][r] * m;
          dT[kt][r] *= m;                                  // dP^T m
          dsum += sT[kt][r] * dT[kt][r];
        }
        pmq[kt] = cv4s(pm);
      }
      dsum = rows4_sum(dsum);                              // D[query = col] = sum_keys (P m) dP
      // dQ^T[d][query] = sum_key K[key][d] dS^T[key][query] (tokens + bank rows); valid where d is in head h = this lane group
      f32x4 t = zero4;
      s16x4 dsq[NKT];                                      // dS^T quads, bf16
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = sT[kt][r] * (dT[kt][r] - dsum) * scale;      // dS^T
        dsq[kt] = cv4s(e);
        t = mma16(kt < QT ? kp[kt < QT ? kt : 0] : bkP, dsq[kt], t);
      }
      if (mine) dq = t;
      // The other orientation (lane = key, registers = queries 4 q4 + r), which the contractions over QUERIES need: a quad tile in
      // accumulator layout read as an A operand is its own transpose, so ONE MFMA against the identity turns (P m)^T / dS^T into
      // P m / dS exactly (bf16 values x 1.0) -- instead of forming S and dP a second time and redoing the exponentials and the
      // dropout hashes, which were ~45 % of this kernel's vector instructions.
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const f32x4 f = mma16(dsq[kt], idq, zero4), pm2 = mma16(pmq[kt], idq, zero4);   // dS / P m [query = 4 q4 + r][key = col]
        // dK^T[d][key] = sum_query Q[query][d] dS[query][key];  dV^T[d][key] = sum_query dO[query][d] (P m)[query][key]
        const f32x4 tk = mma16(qp, cv4s(f), zero4), tv = mma16(dop, cv4s(pm2), zero4);
        if (kt < QT) {
          if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { dk[kt < QT ? kt : 0][r] += tk[r]; dv[kt < QT ? kt : 0][r] += tv[r]; }
          }
        } else if (mine) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { dshk[r] += tk[r]; dshv[r] += tv[r]; }
        }
      }
    }
    // dq[token = 16 qt + col][d = 4 q4 + r] parks in the token tile: this tile's rows of group g's columns are dead (q, k, v of the
    // tile are formed), and a register array indexed by the loop counter would live in scratch
    *reinterpret_cast<bf16x4*>(xt + (16 * qt + col) * LDX6 + WPG * g + 4 * q4) = cv4(dq);
  }
  // dq / dk / dv -> the (image, token, group) rows of dqkv;  dx_g^T[c][token] = sum_d Wq[d][c] dq[token][d] + Wk .. + Wv ..
#pragma unroll
  for (int kt = 0; kt < QT; ++kt) {
    const bf16x4 dk4 = cv4(dk[kt]), dv4 = cv4(dv[kt]);
    const bf16x4 dq4 = *reinterpret_cast<const bf16x4*>(xt + (16 * kt + col) * LDX6 + WPG * g + 4 * q4);   // read before this lane's own dx quads overwrite the slot
    bf16* drow = dqg + ((size_t)(16 * kt + col) * WG + g) * (3 * WCG);
    *reinterpret_cast<bf16x4*>(drow + 4 * q4) = dq4;
    *reinterpret_cast<bf16x4*>(drow + WCG + 4 * q4) = dk4;
    *reinterpret_cast<bf16x4*>(drow + 2 * WCG + 4 * q4) = dv4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 c = mma16(wtf[0][ks], as_s16(dq4), zero4);
      c = mma16(wtf[1][ks], as_s16(dk4), c);
      c = mma16(wtf[2][ks], as_s16(dv4), c);
      *reinterpret_cast<bf16x4*>(xt + (16 * kt + col) * LDX6 + WPG * g + 16 * ks + 4 * q4) = cv4(c);      // in place: group g's columns (only this wave reads them) become dx
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {                          // whole dx rows out: 16-byte pieces
    const int p = tid + NTH * it, row = p / 24, c8 = p - row * 24;
    *reinterpret_cast<bf16x8*>(dxg + (size_t)row * a.lddx + 8 * c8) = *reinterpret_cast<const bf16x8*>(xt + row * LDX6 + 8 * c8);
  }
  // ---- bank-row gradients: the six groups' sums -> one row of partials ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave * PARTS + col * WCG + 4 * q4 + r] = dshk[r];                  // [s][d]
    red[wave * PARTS + WS * WCG + col * WCG + 4 * q4 + r] = dshv[r];
  }
  __syncthreads();
  for (int e = tid; e < PARTS; e += NTH) {
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < BW_WAVES; ++w) sacc += red[w * PARTS + e];
    a.parts[(size_t)blockIdx.x * PARTS + e] = sacc;
  }
}

}  // namespace

// entry points called by cga.hip's qavit_cga_fwd / qavit_cga_bwd for T = 64 (validated there)
int cga64_fwd_launch(const qavit_cga_args* a, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cga64_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMF_TOTAL);
    attr_done = true;
  }
  hipLaunchKernelGGL(cga64_fwd_kernel, dim3(a->B), dim3(64 * FW_WAVES), SMF_TOTAL, st, *a);
  if (a->nan_flag && !a->nan_defer)
    branch_nan_fix_launch(a->out, a->ldo, a->B * WT, WC, a->bproj, a->proj_drop_p, a->proj_drop_site, a->rng, a->nan_flag, a->nan_trip, a->o_save, WO, WO, st);
  return check_launch("cga_fwd");
}

int cga64_bwd_launch(const qavit_cga_bwd_args* a, hipStream_t st) {
  static_assert(PARTS == QAVIT_CGA_PARTS_FLOATS, "header constant out of date");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cga64_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMB_TOTAL);
    attr_done = true;
  }
  hipLaunchKernelGGL(cga64_bwd_kernel, dim3(a->B), dim3(64 * BW_WAVES), SMB_TOTAL, st, *a);
  return check_launch("cga_bwd");
}

}  // namespace qv
