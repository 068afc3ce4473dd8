// Label-smoothed cross entropy, forward AND gradient in one launch (the reference's criterion, HQAViT_CIFAR100.py:1373 /
// :1404-1408: nn.CrossEntropyLoss(label_smoothing) on the logits, or lam * CE(y_a) + (1 - lam) * CE(y_b) under MixUp / CutMix).
// torch's composite op is ~16 tiny kernels forward + backward on a [B, 100] matrix; here one launch, a wave per row (lanes stride
// the classes: coalesced loads, two wave reductions), 16 rows per workgroup:
//   p = softmax(logits_i);  t = (1 - ls) * (lam * onehot(y_a) + (1 - lam) * onehot(y_b)) + ls / C
//   loss = mean_i ( - sum_c t_c log p_c ),   dlogits_i = (p - t) / B          (reduction = 'mean', as torch)
// The loss is deterministic: workgroups leave their partial sums in ws[1 + block]; the last one to arrive (ticket in ws[0]) adds
// them in block order, writes loss[0] and puts the ticket back to zero -- no float atomics, no memset node.
#include "common.cuh"
#include "../../include/qavit.h"
#include "launch.h"

namespace qv {

constexpr int CE_WAVES = 16;

template <typename T>
__global__ __launch_bounds__(64 * CE_WAVES) void ce_ls_kernel(const T* logits, const int64_t* ya, const int64_t* yb, const float* lam_dev, float ls,
                                                              int B, int C, float* loss, T* dlogits, float* ws) {
  __shared__ float red[CE_WAVES];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float lam = (yb && lam_dev) ? lam_dev[0] : 1.f;
  const float invB = 1.f / (float)B, uni = ls / (float)C;
  float part = 0.f;
  const int i = blockIdx.x * CE_WAVES + wave;
  if (i < B) {                                                   // uniform per wave
    const T* row = logits + (size_t)i * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, to_f<T>(row[c]));
    mx = wave_max(mx);
    float se = 0.f, sl = 0.f;
    for (int c = lane; c < C; c += 64) { const float z = to_f<T>(row[c]) - mx; se += __expf(z); sl += z; }
    se = wave_sum(se); sl = wave_sum(sl);
    const float lse = __logf(se);
    const int a = (int)ya[i], b = yb ? (int)yb[i] : a;
    const float la = to_f<T>(row[a]) - mx - lse, lb = to_f<T>(row[b]) - mx - lse;      // log p[y_a], log p[y_b]
    const float mean_logp = sl / (float)C - lse;                                        // (1/C) sum_c log p_c
    part = -(1.f - ls) * (lam * la + (1.f - lam) * lb) - ls * mean_logp;
    if (dlogits) {
      T* drow = dlogits + (size_t)i * C;
      const float inv_se = 1.f / se;
      for (int c = lane; c < C; c += 64) {
        float t = uni;
        if (c == a) t += (1.f - ls) * lam;
        if (c == b) t += (1.f - ls) * (1.f - lam);
        drow[c] = from_f<T>((__expf(to_f<T>(row[c]) - mx) * inv_se - t) * invB);
      }
    }
  }
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < CE_WAVES; ++w) s += red[w];
    __hip_atomic_store(ws + 1 + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned ticket = atomicAdd(reinterpret_cast<unsigned*>(ws), 1u);
    last = (ticket == gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    float s = 0.f;
    for (unsigned w = 0; w < gridDim.x; ++w) s += __hip_atomic_load(ws + 1 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    loss[0] = s * invB;
    __hip_atomic_store(reinterpret_cast<unsigned*>(ws), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace qv

using namespace qv;

extern "C" int qavit_ce_label_smooth(int dtype, const void* logits, const int64_t* y_a, const int64_t* y_b, const float* lam_dev,
                                     float label_smoothing, int B, int C, float* loss, void* dlogits, float* ws, void* stream) {
  if (!logits || !y_a || !loss || !ws || B <= 0 || C <= 0 || label_smoothing < 0.f || label_smoothing >= 1.f)
    return set_error(QAVIT_EINVAL, "ce_label_smooth: bad arguments");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = (B + CE_WAVES - 1) / CE_WAVES;
  if (dtype == QAVIT_F32)
    hipLaunchKernelGGL((ce_ls_kernel<float>), dim3(grid), dim3(64 * CE_WAVES), 0, st, (const float*)logits, y_a, y_b, lam_dev, label_smoothing, B, C, loss, (float*)dlogits, ws);
  else if (dtype == QAVIT_BF16)
    hipLaunchKernelGGL((ce_ls_kernel<bf16>), dim3(grid), dim3(64 * CE_WAVES), 0, st, (const bf16*)logits, y_a, y_b, lam_dev, label_smoothing, B, C, loss, (bf16*)dlogits, ws);
  else return set_error(QAVIT_EINVAL, "ce_label_smooth: unknown dtype");
  return check_launch("ce_label_smooth");
}
