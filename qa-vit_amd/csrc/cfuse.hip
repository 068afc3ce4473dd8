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
constexpr int SMF_W = 0, SMF_GB = FNB * FCB * FLD * 2, SMF_X = SMF_GB + FNB * 2 * FC * 4, SMF_TOTAL = SMF_X + FNW * FT * FLD * 2;   // 76800 + 6144 + 51200 = 134144

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
  // weights (row-major tiles), gamma / beta, fusion weights
  for (int p = tid; p < FNB * FCB * (FC / 8); p += 64 * FNW) {
    const int b = p / (FCB * (FC / 8)), rem = p - b * (FCB * (FC / 8)), row = rem / (FC / 8), c8 = rem - row * (FC / 8);
    *reinterpret_cast<bf16x8*>(sw + ((size_t)b * FCB + row) * FLD + 8 * c8) =
        *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(a.w_rm[b]) + (size_t)row * FC + 8 * c8);
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
  const bf16* swb = sw + (size_t)br * FCB * FLD;
  const float* gam = sgb + br * 2 * FC;
  const float* bet = gam + FC;
  const float invC = 1.f / (float)FC;

  for (int ii = 0; ii < 2; ++ii) {
    const int img = img0 + ii;
    if (img >= a.B) break;                                 // uniform per wave; no barrier below
    wave_sync();                                           // the previous image's fragment reads are done
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int p = lane + 64 * it, row = p / 24, c8 = p % 24;
      *reinterpret_cast<bf16x8*>(xt + row * FLD + 8 * c8) = xr[it];
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
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * invC;
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < FC / 16; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = (float)xf[ks][j] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
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
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cfuse_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMF_TOTAL);
    attr_done = true;
  }
  hipLaunchKernelGGL(cfuse_fwd_kernel, dim3((a->B + FIMG - 1) / FIMG), dim3(64 * FNW), SMF_TOTAL, st, *a);
  return check_launch("compress_fuse_fwd");
}
