// bf16 fast paths of the per-image token kernels (tokens_bf16.hip): 1 = launched, 0 = shape not covered, < 0 = error
#pragma once
#include <hip/hip_runtime.h>

namespace qv {
int bank_stats_bf16_try(const void* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr, const float* Wg,
                        const float* bg, float* ws, int B, int N, int C, int S, int grid, float eps, hipStream_t st);
}  // namespace qv
