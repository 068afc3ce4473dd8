// bf16 fast paths of the per-image token kernels (tokens_bf16.hip): 1 = launched, 0 = shape not covered, < 0 = error
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/qavit.h"

namespace qv {
// fix (optional): the deferred NaN-rule rewrite of the branch that produced `tokens` (include/qavit.h qavit_nan_fix), done per image before it is read
int bank_stats_bf16_try(const void* tokens, const float* gbr, const float* bbr, const float* gwr, const float* bwr, const float* Wg,
                        const float* bg, float* ws, int B, int N, int C, int S, int grid, float eps, hipStream_t st, const qavit_nan_fix* fix = nullptr);
int upmix_bf16_try(bool bwd, const void* dy, const void* xc, const float* W, const float* bias, const float* gamma, const float* beta, float eps,
                   void* out, float* mean, float* rstd, float* dW, float* dbias, float* dgamma, float* dbeta, int B, int N, int M, int C, hipStream_t st,
                   float* parts = nullptr,       // bwd: parts != NULL -> upmix_bf16_parts() rows of [dW | dbias | dgamma | dbeta] instead of atomics
                   const void* sa_u = nullptr, void* sa_du = nullptr, const float* sa_gamma = nullptr, float* sa_dgamma = nullptr,   // bwd: the scale-add in front of
                   float sa_dp_p = 0.f, int sa_dp_site = 0, const int64_t* sa_rng = nullptr);                                          // the up-mix, differentiated here
int upmix_bf16_parts(int B, int N, int M, int C);
int tokmix_bf16_try(bool bwd, const void* a0, const void* x, const void* dxc, void* o0, void* o1, int B, int N, int M, int C, hipStream_t st);
}  // namespace qv
