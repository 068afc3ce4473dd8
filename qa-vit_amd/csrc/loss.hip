// Label-smoothed cross entropy, forward AND gradient in one launch (the reference's criterion, HQAViT_CIFAR100.py:1373 /
// :1404-1408: nn.CrossEntropyLoss(label_smoothing) on the logits, or lam * CE(y_a) + (1 - lam) * CE(y_b) under MixUp / CutMix).
// torch's composite op is ~16 tiny kernels forward + backward on a [B, 100] matrix; here one workgroup walks the rows:
//   p = softmax(logits_i);  t = (1 - ls) * (lam * onehot(y_a) + (1 - lam) * onehot(y_b)) + ls / C
//   loss = mean_i ( - sum_c t_c log p_c ),   dlogits_i = (p - t) / B          (reduction = 'mean', as torch)
// Deterministic (fixed reduction order), no atomics, no memset.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"

namespace qv {

template <typename T>
__global__ __launch_bounds__(1024) void ce_ls_kernel(const T* logits, const int64_t* ya, const int64_t* yb, const float* lam_dev, float ls,
                                                     int B, int C, float* loss, T* dlogits) {
  __shared__ float red[16];
  const float lam = (yb && lam_dev) ? lam_dev[0] : 1.f;
  const float invB = 1.f / (float)B, uni = ls / (float)C;
  float part = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const T* row = logits + (size_t)i * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, to_f<T>(row[c]));
    float se = 0.f, sl = 0.f;
    for (int c = 0; c < C; ++c) { const float z = to_f<T>(row[c]) - mx; se += __expf(z); sl += z; }
    const float lse = __logf(se);
    const int a = (int)ya[i], b = yb ? (int)yb[i] : a;
    const float la = to_f<T>(row[a]) - mx - lse, lb = to_f<T>(row[b]) - mx - lse;      // log p[y_a], log p[y_b]
    const float mean_logp = sl / (float)C - lse;                                        // (1/C) sum_c log p_c
    part += -(1.f - ls) * (lam * la + (1.f - lam) * lb) - ls * mean_logp;
    if (dlogits) {
      T* drow = dlogits + (size_t)i * C;
      const float inv_se = 1.f / se;
      for (int c = 0; c < C; ++c) {
        float t = uni;
        if (c == a) t += (1.f - ls) * lam;
        if (c == b) t += (1.f - ls) * (1.f - lam);
        drow[c] = from_f<T>((__expf(to_f<T>(row[c]) - mx) * inv_se - t) * invB);
      }
    }
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    loss[0] = s * invB;
  }
}

}  // namespace qv

using namespace qv;

extern "C" int qavit_ce_label_smooth(int dtype, const void* logits, const int64_t* y_a, const int64_t* y_b, const float* lam_dev,
                                     float label_smoothing, int B, int C, float* loss, void* dlogits, void* stream) {
  if (!logits || !y_a || !loss || B <= 0 || C <= 0 || label_smoothing < 0.f || label_smoothing >= 1.f)
    return set_error(QAVIT_EINVAL, "ce_label_smooth: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int threads = B >= 1024 ? 1024 : ((B + 63) / 64) * 64;
  if (dtype == QAVIT_F32)
    hipLaunchKernelGGL((ce_ls_kernel<float>), dim3(1), dim3(threads), 0, st, (const float*)logits, y_a, y_b, lam_dev, label_smoothing, B, C, loss, (float*)dlogits);
  else if (dtype == QAVIT_BF16)
    hipLaunchKernelGGL((ce_ls_kernel<bf16>), dim3(1), dim3(threads), 0, st, (const bf16*)logits, y_a, y_b, lam_dev, label_smoothing, B, C, loss, (bf16*)dlogits);
  else return set_error(QAVIT_EINVAL, "ce_label_smooth: unknown dtype");
  return check_launch("ce_label_smooth");
}
