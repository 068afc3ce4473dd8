// Row addressing shared by the fp32 (attn.hip) and bf16 (attn_bf16.hip) attention kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/qavit.h"
#include "common.cuh"

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

// Attention-probability dropout (qavit_attn_args.drop_p).  One key per launch (seed, step, site), one per (group, head)
// problem, then a hash of (query, key): every attention kernel -- fp32 or bf16, forward or backward, whatever its tiling --
// sees the same mask for the same element.
struct AttnDrop {
  uint32_t key; float p, inv_keep; bool on;
};
__device__ __forceinline__ AttnDrop attn_drop_init(const qavit_attn_args& a) {
  AttnDrop d;
  d.on = a.drop_p > 0.f && a.rng != nullptr;
  d.p = a.drop_p;
  d.inv_keep = d.on ? 1.f / (1.f - a.drop_p) : 1.f;
  d.key = d.on ? rng_key(a.rng, a.drop_site) : 0u;
  return d;
}
__device__ __forceinline__ uint32_t attn_drop_pkey(const AttnDrop& d, int pid) { return mix32(d.key + (uint32_t)pid * 0x9E3779B9U); }
// multiplicative factor of P[query i][key j]: 0 or 1/(1-p)
__device__ __forceinline__ float attn_drop_factor(const AttnDrop& d, uint32_t pkey, int i, int j) {
  return drop_factor(pkey, ((uint32_t)i << 16) | (uint32_t)j, d.p, d.inv_keep);
}

// bf16 fast path: 1 = launched, 0 = shape not covered (fall back to the generic kernel), < 0 = error
int attn_bf16_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st);
// small-problem variant (<= 16 queries / token keys; attn3_bf16.hip), same return convention; called by attn_bf16_try
int attn3_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st);
// four-waves-per-problem variant for Nq >= 32 (attn4_bf16.hip), same return convention; called by attn_bf16_try
int attn4_try(const qavit_attn_args& a, bool bwd, int grid, hipStream_t st);

}  // namespace qv
