// Row addressing shared by the fp32 (attn.hip) and bf16 (attn_bf16.hip) attention kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/qavit.h"

namespace qv {

__device__ __forceinline__ int64_t attn_qrow(const qavit_attn_args& a, int g, int i) {
  if (a.groups_per_b <= 0) return (int64_t)g * a.Nq + i;
  const int b = g / a.groups_per_b, gi = g - b * a.groups_per_b;
  return (int64_t)b * a.q_rows_per_b + (a.q_tbl ? a.q_tbl[gi * a.Nq + i] : gi * a.Nq + i);
}
__device__ __forceinline__ int64_t attn_krow(const qavit_attn_args& a, int g, int l) {
  if (a.groups_per_b <= 0) return (int64_t)g * a.L + l;
  const int b = g / a.groups_per_b, gi = g - b * a.groups_per_b;
  return (int64_t)b * a.k_rows_per_b + (a.k_tbl ? a.k_tbl[gi * a.L + l] : gi * a.L + l);
}

// bf16 fast path: 1 = launched, 0 = shape not covered (fall back to the generic kernel), < 0 = error
int attn_bf16_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st);
// small-problem variant (<= 16 queries / token keys; attn3_bf16.hip), same return convention; called by attn_bf16_try
int attn3_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st);
// four-waves-per-problem variant for Nq >= 32 (attn4_bf16.hip), same return convention; called by attn_bf16_try
int attn4_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st);

}  // namespace qv
