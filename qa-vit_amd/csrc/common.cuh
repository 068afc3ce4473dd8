// Shared device helpers for the qavit gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qv {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------------
// scalar conversions (fp32 is the arithmetic type everywhere; T is the storage type)
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }   // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// elements per 16-byte vector
template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct Vec<bf16> { static constexpr int N = 8; typedef bf16x8 type; };

// ------------------------------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane, on the VALU's data-parallel-primitive paths
// (quad permutes, then the half-row and row mirrors): no LDS-pipe permute instructions, which all the waves of a CU share
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]: lane ^ 1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]: lane ^ 2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad of the half row
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror: the other half row
  return v;
}
// lane ^ 16 and lane ^ 32 exchanges by gfx950's v_permlane16_swap / v_permlane32_swap (VALU; with both operands = v the two results
// are {even rows' values in both rows of a pair, odd rows' values ...} resp. {low half, high half}): sums / maxima over the four
// 16-lane rows without ds_bpermute, which goes through the LDS pipe all the waves of a CU share
__device__ __forceinline__ float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows4_sum(float v) { return xor32_sum(xor16_sum(v)); }     // over the 4 lanes {col, col+16, col+32, col+48}
__device__ __forceinline__ float rows4_max(float v) { return xor32_max(xor16_max(v)); }
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  return v;
}
// Reductions over a power-of-two group of `W` adjacent lanes (W <= 64), result in every lane of the group, entirely on the VALU: quad
// permutes and row mirrors (DPP) inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across rows.  The __shfl_xor butterfly these
// replace is one ds_bpermute_b32 per step -- an LDS-pipe round trip of 100+ cycles, log2(W) of them in a dependent chain.
template <int W> __device__ __forceinline__ float group_sum(float v) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group_sum: power of two up to 64");
  if (W >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  if (W >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  if (W >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  if (W >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  if (W >= 32) v = xor16_sum(v);
  if (W >= 64) v = xor32_sum(v);
  return v;
}
template <int W> __device__ __forceinline__ float group_max(float v) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group_max: power of two up to 64");
  if (W >= 2) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  if (W >= 4) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  if (W >= 8) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  if (W >= 16) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  if (W >= 32) v = xor16_max(v);
  if (W >= 64) v = xor32_max(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

// exact-erf GELU (nn.GELU() default) and its derivative
// ------------------------------------------------------------------------------------------------
// Exact (erf) GELU, HQAViT_CIFAR100.py nn.GELU().  erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, the same size as
// the fp32 round-off of 0.5*x*(1+erff(x/sqrt2))): one v_rcp + one v_exp + 6 fma instead of libm's ~50-instruction erff,
// which made the GELU epilogue of the 1024-wide FFN GEMMs VALU-bound.  The negative tail is formed without the
// 1 - (1 - t) cancellation, and exp(-x^2/2) is shared with the density term of the gradient.
__device__ __forceinline__ float gelu_tail(float x, float& e) {      // returns 0.5 * erfc(|x|/sqrt2), e = exp(-x^2/2)
  const float u = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
  e = __expf(-(u * u));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return 0.5f * poly * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float e;
  const float tail = gelu_tail(x, e);
  return x * (x >= 0.f ? 1.0f - tail : tail);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float e;
  const float tail = gelu_tail(x, e);
  const float cdf = x >= 0.f ? 1.0f - tail : tail;
  return cdf + x * (0.39894228040143268f * e);
}

// ------------------------------------------------------------------------------------------------
// counter-based RNG for dropout / drop-path.  rng[0] = seed, rng[1] = step counter (device memory, so a
// captured hipGraph replays with fresh masks).  A mask is a pure function of (seed, step, site, index):
// the backward pass regenerates it instead of storing it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng_key(const int64_t* rng, int site) {
  const uint32_t seed = (uint32_t)rng[0], step = (uint32_t)rng[1];
  return mix32(seed ^ mix32(step * 0x9E3779B9U + (uint32_t)site * 0x85EBCA6BU + 0x68E31DA4U));
}
// uniform in [0,1)
__device__ __forceinline__ float rng_uniform(uint32_t key, uint32_t idx) {
  return (float)(mix32(idx * 0x9E3779B9U ^ key) >> 8) * (1.0f / 16777216.0f);
}
// multiplicative dropout factor: 0 or 1/(1-p).  Elements 2m and 2m+1 share ONE hash (its low / high 16 bits): the 32-bit
// multiplies of mix32 are quarter-rate instructions and a dropout epilogue was paying two of them per element.  The drop
// threshold is p in 1/65536 steps (|p_eff - p| < 1.6e-5, far inside the mask's own sampling noise).
__device__ __forceinline__ uint32_t rng_u16(uint32_t key, uint32_t idx) {
  const uint32_t h = mix32((idx >> 1) * 0x9E3779B9U ^ key);
  return (idx & 1u) ? (h >> 16) : (h & 0xFFFFu);
}
__device__ __forceinline__ float drop_factor(uint32_t key, uint32_t idx, float p, float inv_keep) {
  return rng_u16(key, idx) >= (uint32_t)(p * 65536.0f) ? inv_keep : 0.0f;
}

__device__ __forceinline__ void atomic_add_f(float* p, float v) { atomicAdd(p, v); }

// Orders one wave's own LDS traffic (LDS operations of a wave execute in issue order; this stops the compiler
// from reordering them).  NOT a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace qv
