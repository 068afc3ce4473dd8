// QuadAttentionBlock's  HybridFusion( concat_i compress_i( norm_i( branch_i ) ) )  (HQAViT_CIFAR100.py:904-925, :1075-1081), forward, for
// the 16-learned-token problems: four LayerNorms, four Linear(192 -> 48), the concat and the softmax-weighted scaling in ONE launch
// (it replaces the grouped row-statistics launch, the grouped LayerNorm-prologue GEMM and the scaling kernel).  Outputs exactly what
// the unfused forward left for backward: cat (unscaled concat), the scaled y, the four norms' mean / rstd.
//
// A wave owns ONE branch (its 48 x 192 weight in LDS, row-major; gamma / beta in LDS) and walks two images: a token tile [16][192] is
// staged in LDS, read back as operand fragments (lane = token, 4 consecutive channels), normalised in registers (a token's 192 values
// sit in 4 lanes: the row statistics are two lane-permute steps) and multiplied with the weight's row fragments: 36 MFMAs per
// (image, branch), accumulators = z^T[out dim][token] = 8-byte row segments of the outputs.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"
#include "frag16.cuh"

namespace qv {

namespace {

constexpr int FT = 16, FC = 192, FNB = 4, FCB = 48;        // tokens, channels, branches, compressed channels per branch
constexpr int FNW = 8, FIMG = 4;                           // waves per workgroup (4 branches x 2 image slots), images per workgroup
constexpr int FLD = FC + 8;
constexpr int SMF_W = 0, SMF_GB = FNB * FCB * FLD * 2, SMF_X = SMF_GB + FNB * 2 * FC * 4, SMF_FLAG = SMF_X + FNW * FT * FLD * 2, SMF_TOTAL = SMF_FLAG + 16;   // 76800 + 6144 + 51200 (+ the NaN flag's broadcast word) = 134160

__global__ __launch_bounds__(64 * FNW) void cfuse_fwd_kernel(qavit_cfuse_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q4 = lane >> 4;
  const int br = wave & 3, slot = wave >> 2;
  bf16* sw = reinterpret_cast<bf16*>(smraw + SMF_W);        // [4][48][FLD]
  float* sgb = reinterpret_cast<float*>(smraw + SMF_GB);    // [4][2][192]
  bf16* xt = reinterpret_cast<bf16*>(smraw + SMF_X) + wave * (FT * FLD);
  bf16* catg = reinterpret_cast<bf16*>(a.cat);
  bf16* yg = reinterpret_cast<bf16*>(a.y);
  const bf16* xg = reinterpret_cast<const bf16*>(a.x[br]);

  auto load_x = [&](int img, bf16x8* r) {
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int p = lane + 64 * it, row = p / 24, c8 = p % 24;
      r[it] = *reinterpret_cast<const bf16x8*>(xg + ((size_t)img * FT + row) * FC + 8 * c8);
    }
  };
  const int img0 = blockIdx.x * FIMG + 2 * slot;
  bf16x8 xr[6], xn[6];
  load_x(img0 < a.B ? img0 : a.B - 1, xr);
  // the NaN rule of the branch whose output this launch reads next (a.fix, see the header): the flag is read by one thread beside the
  // weight staging below and reaches the others through the barrier that staging needs anyway
  volatile int& f_s = *reinterpret_cast<volatile int*>(smraw + SMF_FLAG);
  if (tid == 0) {
    int f = 0;
    if (a.fix.flag) {
      f = *reinterpret_cast<volatile int*>(a.fix.flag);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the read has returned before this workgroup's arrival is counted
      if (atomicAdd(a.fix.flag + 1, 1) == (int)gridDim.x - 1) { a.fix.flag[0] = 0; a.fix.flag[1] = 0; }
    }
    f_s = f;
  }
  // weights (row-major tiles), gamma / beta, fusion weights
  {
    // the four weight tiles: all nine 16-byte loads of a thread in flight before the first LDS store (a load-store loop with a
    // runtime trip count waits for every load in turn: nine memory round trips before the kernel starts)
    constexpr int WPT = FNB * FCB * (FC / 8) / (64 * FNW);
    static_assert(WPT * 64 * FNW == FNB * FCB * (FC / 8), "weight pieces must divide over the threads");
    bf16x8 wr[WPT];
#pragma unroll
    for (int it = 0; it < WPT; ++it) {
      const int p = tid + it * 64 * FNW;
      const int b = p / (FCB * (FC / 8)), rem = p - b * (FCB * (FC / 8)), row = rem / (FC / 8), c8 = rem - row * (FC / 8);
      wr[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.w_rm[b]) + (size_t)row * FC + 8 * c8);
    }
#pragma unroll
    for (int it = 0; it < WPT; ++it) {
      const int p = tid + it * 64 * FNW;
      const int b = p / (FCB * (FC / 8)), rem = p - b * (FCB * (FC / 8)), row = rem / (FC / 8), c8 = rem - row * (FC / 8);
      *reinterpret_cast<bf16x8*>(sw + ((size_t)b * FCB + row) * FLD + 8 * c8) = wr[it];
    }
  }
  for (int p = tid; p < FNB * 2 * FC; p += 64 * FNW) {
    const int b = p / (2 * FC), rem = p - b * (2 * FC);
    sgb[p] = rem < FC ? a.gamma[b][rem] : a.beta[b][rem - FC];
  }
  float wsc;
  {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < FNB; ++i) mx = fmaxf(mx, a.fw[i]);
    float ssum = 0.f, mine = 0.f;
#pragma unroll
    for (int i = 0; i < FNB; ++i) { const float e = __expf(a.fw[i] - mx); ssum += e; mine = (i == br) ? e : mine; }
    wsc = mine / ssum;
  }
  f32x4 bias4[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) bias4[nt] = a.bias[br] ? *reinterpret_cast<const f32x4*>(a.bias[br] + 16 * nt + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const bool fixit = f_s != 0;                             // uniform over the grid
  if (a.fix.flag && a.fix.trip && blockIdx.x == 0 && tid == 0) *a.fix.trip = fixit ? 1 : 0;
  const bool fix_mine = fixit && br == a.fix_branch;       // uniform per wave
  const bf16* swb = sw + (size_t)br * FCB * FLD;
  const float* gam = sgb + br * 2 * FC;
  const float* bet = gam + FC;
  const float invC = 1.f / (float)FC;

  for (int ii = 0; ii < 2; ++ii) {
    const int img = img0 + ii;
    if (img >= a.B) break;                                 // uniform per wave; no barrier below
    wave_sync();                                           // the previous image's fragment reads are done
    if (fix_mine) {
      // rare: the branch produced NaN somewhere, its whole output is dropout(proj(0)) = dropout(bias) -- for this reader (the image's
      // tile is built from the bias instead of from the rows loaded) and for the backward (written back); the saved attention output is
      // zero.  One piece at a time, straight into the tile: the path must not cost the common one its registers.
      const bool fdrop = a.fix.drop_p > 0.f && a.fix.rng != nullptr;
      const uint32_t fkey = fdrop ? rng_key(a.fix.rng, a.fix.drop_site) : 0u;
      const float finv = fdrop ? 1.f / (1.f - a.fix.drop_p) : 1.f;
      bf16* xw = const_cast<bf16*>(xg);
#pragma unroll 1
      for (int it = 0; it < 6; ++it) {
        const int p = lane + 64 * it, row = p / 24, c8 = p % 24;
        const uint32_t grow = (uint32_t)(img * FT + row);
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = 8 * c8 + j;
          float val = a.fix.bias[c];
          if (fdrop) val *= drop_factor(fkey, grow * (uint32_t)FC + (uint32_t)c, a.fix.drop_p, finv);
          t[j] = (bf16)val;
        }
        *reinterpret_cast<bf16x8*>(xw + ((size_t)img * FT + row) * FC + 8 * c8) = t;
        *reinterpret_cast<bf16x8*>(xt + row * FLD + 8 * c8) = t;
      }
      if (a.fix.o_save) {
        bf16* os = reinterpret_cast<bf16*>(a.fix.o_save) + (size_t)img * FT * a.fix.ldos;
        for (int i = lane; i < FT * a.fix.Co; i += 64) { const int r = i / a.fix.Co, c = i - r * a.fix.Co; os[(size_t)r * a.fix.ldos + c] = (bf16)0.f; }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int p = lane + 64 * it, row = p / 24, c8 = p % 24;
        *reinterpret_cast<bf16x8*>(xt + row * FLD + 8 * c8) = xr[it];
      }
    }
    if (ii == 0 && img + 1 < a.B) load_x(img + 1, xn);      // the second image's rows fly during the first one's arithmetic
    wave_sync();
    bf16x4 xf[FC / 16];
    float s1 = 0.f;
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks) {
      xf[ks] = *reinterpret_cast<const bf16x4*>(xt + col * FLD + 16 * ks + 4 * q4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s1 += (float)xf[ks][j];
    }
    s1 = rows4_sum(s1);
    const float mean = s1 * invC;
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = (float)xf[ks][j] - mean; s2 += d * d; }
    s2 = rows4_sum(s2);
    const float rstd = rsqrtf(s2 * invC + a.eps);
    if (q4 == 0) { a.mean[br][(size_t)img * FT + col] = mean; a.rstd[br][(size_t)img * FT + col] = rstd; }
    f32x4 acc[3] = {bias4[0], bias4[1], bias4[2]};
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gam + 16 * ks + 4 * q4);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bet + 16 * ks + 4 * q4);
      bf16x4 n4;
#pragma unroll
      for (int j = 0; j < 4; ++j) n4[j] = (bf16)(((float)xf[ks][j] - mean) * rstd * g4[j] + b4[j]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[nt] = mma16(rowfrag(swb, FLD, 16 * nt, 16 * ks), as_s16(n4), acc[nt]);   // z^T[n = 16 nt + 4 q4 + r][token = col]
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      bf16x4 z4, y4;
#pragma unroll
      for (int r = 0; r < 4; ++r) { z4[r] = (bf16)acc[nt][r]; y4[r] = (bf16)((float)z4[r] * wsc); }
      const size_t o = ((size_t)img * FT + col) * (FNB * FCB) + br * FCB + 16 * nt + 4 * q4;
      *reinterpret_cast<bf16x4*>(catg + o) = z4;
      *reinterpret_cast<bf16x4*>(yg + o) = y4;
    }
    if (ii == 0) {
#pragma unroll
      for (int it = 0; it < 6; ++it) xr[it] = xn[it];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward in one launch (it replaces the scaling-backward kernel, the grouped input-gradient GEMM and the grouped LayerNorm backward):
//   dcat_i = dy_i * softmax(fw)_i  (written once: operand of the deferred weight-gradient GEMMs),  ds_i = sum dy_i * cat_i,
//   dxn_i = dcat_i W_i  (W_i^T fragments by transposing LDS reads of the same row-major tile),  dx_i = LayerNorm'(dxn_i)
// Per-channel sums over tokens and images (dgamma, dbeta) stay in registers across the wave's images and are folded over the 16 token
// lanes once, at the end; with the fusion-logit gradients they leave as ONE row of partial sums per workgroup.
constexpr int CF_PART = FNB * 2 * FC + 8;                  // [4 x (dgamma 192 | dbeta 192) | dfw 4 (+4 pad)] = 1544 floats
constexpr int SMB_W = 0, SMB_G = FNB * FCB * FLD * 2, SMB_X2 = SMB_G + FNB * FC * 4, SMB_R = SMB_X2 + FNW * FT * FLD * 2,
              SMB_TOTAL = SMB_R + (FNW * 2 * FC + 16) * 4;  // 76800 + 3072 + 51200 + 12352 = 143424 bytes

__global__ __launch_bounds__(64 * FNW) void cfuse_bwd_kernel(qavit_cfuse_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every per-wave base pointer below lives in SGPRs
  const int br = wave & 3, slot = wave >> 2;
  bf16* sw = reinterpret_cast<bf16*>(smraw + SMB_W);
  float* sgam = reinterpret_cast<float*>(smraw + SMB_G);    // [4][192]
  bf16* xt = reinterpret_cast<bf16*>(smraw + SMB_X2) + wave * (FT * FLD);
  float* red = reinterpret_cast<float*>(smraw + SMB_R);     // [8 waves][2][192] + 16 (ds per wave)
  const bf16* dyg = reinterpret_cast<const bf16*>(a.dy);
  const bf16* catg = reinterpret_cast<const bf16*>(a.cat);
  bf16* dcg = reinterpret_cast<bf16*>(a.dcat);
  const bf16* xg = reinterpret_cast<const bf16*>(a.x[br]);
  bf16* dxg = reinterpret_cast<bf16*>(a.dx[br]);

  auto load_x = [&](int img, bf16x8* r) {
#pragma unroll
    for (int it = 0; it < 6; ++it)                          // an image's 16 x 192 tile is contiguous: piece p at 8 p elements
      r[it] = *reinterpret_cast<const bf16x8*>(xg + (size_t)img * (FT * FC) + 8 * (lane + 64 * it));
  };
  const int img0 = blockIdx.x * FIMG + 2 * slot;
  // every global operand of an image is requested in one go (x rows, its slices of dy and cat, the row statistics); the second
  // image's requests go out before the first image's arithmetic
  bf16x8 xr[6];
  bf16x4 dyr[3], ctr[3];
  float mean, rstd;
  auto load_img = [&](int img) {
    load_x(img, xr);
    const size_t rowo = ((size_t)img * FT + col) * (FNB * FCB) + br * FCB + 4 * q4;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      dyr[nt] = *reinterpret_cast<const bf16x4*>(dyg + rowo + 16 * nt);
      ctr[nt] = *reinterpret_cast<const bf16x4*>(catg + rowo + 16 * nt);
    }
    mean = a.mean[br][(size_t)img * FT + col];
    rstd = a.rstd[br][(size_t)img * FT + col];
  };
  load_img(img0 < a.B ? img0 : a.B - 1);
  {
    // the four weight tiles: all nine 16-byte loads of a thread in flight before the first LDS store (a load-store loop with a
    // runtime trip count waits for every load in turn: nine memory round trips before the kernel starts)
    constexpr int WPT = FNB * FCB * (FC / 8) / (64 * FNW);
    static_assert(WPT * 64 * FNW == FNB * FCB * (FC / 8), "weight pieces must divide over the threads");
    bf16x8 wr[WPT];
#pragma unroll
    for (int it = 0; it < WPT; ++it) {
      const int p = tid + it * 64 * FNW;
      const int b = p / (FCB * (FC / 8)), rem = p - b * (FCB * (FC / 8)), row = rem / (FC / 8), c8 = rem - row * (FC / 8);
      wr[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.w_rm[b]) + (size_t)row * FC + 8 * c8);
    }
#pragma unroll
    for (int it = 0; it < WPT; ++it) {
      const int p = tid + it * 64 * FNW;
      const int b = p / (FCB * (FC / 8)), rem = p - b * (FCB * (FC / 8)), row = rem / (FC / 8), c8 = rem - row * (FC / 8);
      *reinterpret_cast<bf16x8*>(sw + ((size_t)b * FCB + row) * FLD + 8 * c8) = wr[it];
    }
  }
  for (int p = tid; p < FNB * FC; p += 64 * FNW) sgam[p] = a.gamma[p / FC][p % FC];
  float wv[FNB], wsc;
  {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < FNB; ++i) mx = fmaxf(mx, a.fw[i]);
    float ssum = 0.f;
#pragma unroll
    for (int i = 0; i < FNB; ++i) { wv[i] = __expf(a.fw[i] - mx); ssum += wv[i]; }
    wsc = 0.f;
#pragma unroll
    for (int i = 0; i < FNB; ++i) { wv[i] /= ssum; wsc = (i == br) ? wv[i] : wsc; }
  }
  __syncthreads();
  const bf16* swb = sw + (size_t)br * FCB * FLD;
  const float* gam = sgam + br * FC;
  const float invC = 1.f / (float)FC;
  // dgamma needs the per-element products, 48 running sums per lane.  dbeta = sum_t dxn[t][c] = sum_n W[n][c] (sum_t dz[t][n]): only
  // the 12 column sums of dz per lane are kept, and the 48 x 192 matrix-vector product is done once per workgroup at the end.
  float pg[FC / 16][4], sdz[3][4], dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < FC / 16; ++ks)
#pragma unroll
    for (int j = 0; j < 4; ++j) pg[ks][j] = 0.f;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) sdz[nt][j] = 0.f;

#pragma unroll 1
  for (int ii = 0; ii < 2; ++ii) {
    const int img = img0 + ii;
    if (img >= a.B) break;                                 // uniform per wave; no barrier inside the loop
    wave_sync();
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int p = lane + 64 * it, row = p / 24, c8 = p % 24;
      *reinterpret_cast<bf16x8*>(xt + row * FLD + 8 * c8) = xr[it];
    }
    const size_t rowo = ((size_t)img * FT + col) * (FNB * FCB) + br * FCB + 4 * q4;
    const float mu = mean, rs = rstd;
    s16x4 dzf[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      bf16x4 z4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = (float)dyr[nt][j];
        z4[j] = (bf16)(g * wsc);
        sdz[nt][j] += (float)z4[j];                          // the rounded value: what the product below multiplies
        dsum += g * (float)ctr[nt][j];
      }
      *reinterpret_cast<bf16x4*>(dcg + rowo + 16 * nt) = z4;
      dzf[nt] = as_s16(z4);
    }
    if (ii == 0 && img + 1 < a.B) load_img(img + 1);        // the second image's operands fly during the first one's arithmetic
    wave_sync();
    // dxn^T[c][token] = sum_n W[n][c] dz[token][n]
    f32x4 acc[FC / 16];
#pragma unroll
    for (int ct = 0; ct < FC / 16; ++ct) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) c = mma16(trfrag(swb, FLD, 16 * nt, 16 * ct), dzf[nt], c);
      acc[ct] = c;                                         // dxn[token = col][c = 16 ct + 4 q4 + r]
      if (ct & 1) __builtin_amdgcn_sched_barrier(0);       // keep the scheduler from hoisting all 36 fragment reads (72 registers) at once
    }
    // LayerNorm backward on the token's row (4 lanes hold it)
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks) {
      const bf16x4 x4 = *reinterpret_cast<const bf16x4*>(xt + col * FLD + 16 * ks + 4 * q4);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gam + 16 * ks + 4 * q4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = ((float)x4[j] - mu) * rs;
        const float d = acc[ks][j];
        const float gg = d * g4[j];
        pg[ks][j] += d * xh;
        c1 += gg * xh;
        c2 += gg;
        acc[ks][j] = gg;
      }
    }
    c1 = rows4_sum(c1);
    c2 = rows4_sum(c2);
    c1 *= invC; c2 *= invC;
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks) {
      const bf16x4 x4 = *reinterpret_cast<const bf16x4*>(xt + col * FLD + 16 * ks + 4 * q4);   // x-hat again from the tile: 48 registers less than keeping it
      bf16x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = (bf16)(rs * (acc[ks][j] - c2 - ((float)x4[j] - mu) * rs * c1));
      *reinterpret_cast<bf16x4*>(xt + col * FLD + 16 * ks + 4 * q4) = o4;          // in place: the tile becomes dx (each lane rewrites what it read)
    }
    wave_sync();
#pragma unroll
    for (int it = 0; it < 6; ++it) {                        // whole rows out: 16-byte pieces, 384 contiguous bytes per row
      const int p = lane + 64 * it, row = p / 24, c8 = p % 24;
      *reinterpret_cast<bf16x8*>(dxg + (size_t)img * (FT * FC) + 8 * p) = *reinterpret_cast<const bf16x8*>(xt + row * FLD + 8 * c8);
    }
  }
  // ---- fold the 16 token lanes, then the two waves of a branch; one row of partials per workgroup ----
#pragma unroll
  for (int ks = 0; ks < FC / 16; ++ks)
#pragma unroll
    for (int j = 0; j < 4; ++j) pg[ks][j] = row16_sum(pg[ks][j]);      // the 16 token lanes = one DPP row
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) sdz[nt][j] = row16_sum(sdz[nt][j]);
  dsum = wave_sum(dsum);
  if (col == 0) {
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 2 + 0) * FC + 16 * ks + 4 * q4 + j] = pg[ks][j];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 2 + 1) * FC + 16 * nt + 4 * q4 + j] = sdz[nt][j];     // the wave's second row: 48 column sums of dz
  }
  if (lane == 0) red[FNW * 2 * FC + wave] = dsum;
  __syncthreads();
  float* out = a.parts + (size_t)blockIdx.x * CF_PART;
  for (int e = tid; e < FNB * 2 * FC; e += 64 * FNW) {
    const int b = e / (2 * FC), rem = e - b * (2 * FC);      // branch b: waves b and b + 4
    float v;
    if (rem < FC) {
      v = red[(b * 2) * FC + rem] + red[((b + 4) * 2) * FC + rem];
    } else {
      const int c = rem - FC;
      const float* s0 = red + (b * 2 + 1) * FC;
      const float* s1 = red + ((b + 4) * 2 + 1) * FC;
      const bf16* wc = sw + (size_t)b * FCB * FLD + c;
      v = 0.f;
#pragma unroll 8
      for (int n = 0; n < FCB; ++n) v += (float)wc[n * FLD] * (s0[n] + s1[n]);
    }
    out[e] = v;
  }
  if (tid == 0) {
    float ds[FNB], dot = 0.f;
#pragma unroll
    for (int b = 0; b < FNB; ++b) { ds[b] = red[FNW * 2 * FC + b] + red[FNW * 2 * FC + b + 4]; dot += ds[b] * wv[b]; }
#pragma unroll
    for (int b = 0; b < FNB; ++b) out[FNB * 2 * FC + b] = wv[b] * (ds[b] - dot);
#pragma unroll
    for (int b = FNB; b < 8; ++b) out[FNB * 2 * FC + b] = 0.f;
  }
}

}  // namespace

}  // namespace qv

using namespace qv;

extern "C" int qavit_compress_fuse_supported(int T, int C, int nb, int Cb) { return (T == FT && C == FC && nb == FNB && Cb == FCB) ? 1 : 0; }

extern "C" int qavit_compress_fuse_fwd(const qavit_cfuse_args* a, void* stream) {
  if (!a) return set_error(QAVIT_EINVAL, "compress_fuse: null args");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "compress_fuse: bf16 only");
  if (a->T != FT || a->C != FC || a->NB != FNB || a->CB != FCB || a->B <= 0) return set_error(QAVIT_EINVAL, "compress_fuse: built for 16 tokens x 192 channels, 4 branches of 48");
  if (!a->fw || !a->cat || !a->y) return set_error(QAVIT_EINVAL, "compress_fuse: null operand");
  for (int i = 0; i < FNB; ++i) {
    if (!a->x[i] || !a->gamma[i] || !a->beta[i] || !a->w_rm[i] || !a->mean[i] || !a->rstd[i]) return set_error(QAVIT_EINVAL, "compress_fuse: null branch operand");
    if ((reinterpret_cast<uintptr_t>(a->x[i]) | reinterpret_cast<uintptr_t>(a->w_rm[i])) & 15) return set_error(QAVIT_EINVAL, "compress_fuse: x and weights must be 16-byte aligned");
    if (a->bias[i] && (reinterpret_cast<uintptr_t>(a->bias[i]) & 15)) return set_error(QAVIT_EINVAL, "compress_fuse: bias must be 16-byte aligned");
  }
  if ((reinterpret_cast<uintptr_t>(a->cat) | reinterpret_cast<uintptr_t>(a->y)) & 7) return set_error(QAVIT_EINVAL, "compress_fuse: outputs must be 8-byte aligned");
  if (a->fix.flag) {
    if (a->fix_branch < 0 || a->fix_branch >= FNB || !a->fix.bias) return set_error(QAVIT_EINVAL, "compress_fuse: the deferred NaN rule needs its branch index and proj bias");
    if (a->fix.drop_p < 0.f || a->fix.drop_p >= 1.f || (a->fix.drop_p > 0.f && !a->fix.rng)) return set_error(QAVIT_EINVAL, "compress_fuse: the deferred NaN rule's dropout needs rng");
    if (a->fix.o_save && (a->fix.Co <= 0 || a->fix.ldos < a->fix.Co)) return set_error(QAVIT_EINVAL, "compress_fuse: the deferred NaN rule's o_save stride");
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cfuse_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMF_TOTAL);
    attr_done = true;
  }
  hipLaunchKernelGGL(cfuse_fwd_kernel, dim3((a->B + FIMG - 1) / FIMG), dim3(64 * FNW), SMF_TOTAL, st, *a);
  return check_launch("compress_fuse_fwd");
}

extern "C" int qavit_compress_fuse_bwd_parts(int B) { return B > 0 ? (B + FIMG - 1) / FIMG : 0; }

extern "C" int qavit_compress_fuse_bwd(const qavit_cfuse_bwd_args* a, void* stream) {
  if (!a) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: null args");
  if (a->dtype != QAVIT_BF16) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: bf16 only");
  if (a->T != FT || a->C != FC || a->NB != FNB || a->CB != FCB || a->B <= 0) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: built for 16 tokens x 192 channels, 4 branches of 48");
  if (!a->fw || !a->dy || !a->cat || !a->dcat || !a->parts) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: null operand");
  for (int i = 0; i < FNB; ++i) {
    if (!a->x[i] || !a->gamma[i] || !a->w_rm[i] || !a->mean[i] || !a->rstd[i] || !a->dx[i]) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: null branch operand");
    if ((reinterpret_cast<uintptr_t>(a->x[i]) | reinterpret_cast<uintptr_t>(a->w_rm[i])) & 15) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: x and weights must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(a->dx[i]) & 15) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: dx must be 16-byte aligned");
  }
  if ((reinterpret_cast<uintptr_t>(a->dy) | reinterpret_cast<uintptr_t>(a->cat) | reinterpret_cast<uintptr_t>(a->dcat)) & 7) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: dy / cat / dcat must be 8-byte aligned");
  if (reinterpret_cast<uintptr_t>(a->parts) & 15) return set_error(QAVIT_EINVAL, "compress_fuse_bwd: parts must be 16-byte aligned");
  static_assert(CF_PART == QAVIT_CFUSE_PARTS_FLOATS, "header constant out of date");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cfuse_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMB_TOTAL);
    attr_done = true;
  }
  hipLaunchKernelGGL(cfuse_bwd_kernel, dim3((a->B + FIMG - 1) / FIMG), dim3(64 * FNW), SMB_TOTAL, st, *a);
  return check_launch("compress_fuse_bwd");
}
