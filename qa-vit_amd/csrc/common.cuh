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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce over a power-of-two group of `W` adjacent lanes (W <= 64)
template <int W> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------
// exact-erf GELU (nn.GELU() default) and its derivative
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
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
// multiplicative dropout factor: 0 or 1/(1-p)
__device__ __forceinline__ float drop_factor(uint32_t key, uint32_t idx, float p, float inv_keep) {
  return rng_uniform(key, idx) >= p ? inv_keep : 0.0f;
}

__device__ __forceinline__ void atomic_add_f(float* p, float v) { atomicAdd(p, v); }

}  // namespace qv
